// Experiment: does buffer_load ... lds (LDS-DMA) write zeros for out-of-range lanes?  Also checks per-lane gather.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
__global__ void k(const uint32_t* src, uint32_t nbytes, uint32_t* out) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[64 * 4 * 2];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 4 * 2; i += 64) lds[i] = 0xABABABABu;
  __syncthreads();
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  // lanes gather rows in reverse order; odd lanes point out of range
  uint32_t voff = (lane & 1) ? 0x7fff0000u : (uint32_t)(63 - lane) * 16u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
  // second instruction with an SGPR offset of 1024 bytes
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + 256), 16, voff, 1024, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 64 * 4 * 2; i += 64) out[i] = lds[i];
}
int main() {
  std::vector<uint32_t> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = i + 1;
  uint32_t *d, *o;
  hipMalloc(&d, 4096); hipMalloc(&o, 2048);
  hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 2048u, o);
  std::vector<uint32_t> r(512);
  hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
  printf("err=%s\n", hipGetErrorString(hipGetLastError()));
  for (int l = 0; l < 6; ++l) printf("lane %d: %08x %08x %08x %08x | +1024: %08x\n", l, r[l*4], r[l*4+1], r[l*4+2], r[l*4+3], r[256 + l*4]);
  printf("lane 62: %08x  lane 63: %08x\n", r[62*4], r[63*4]);
  return 0;
}
