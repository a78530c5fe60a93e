#!/usr/bin/env python3
"""Device-copy bandwidth by access pattern (scripts/experiments/copy_bw.hip): GB/s of read + write for a buffer of --mb MiB."""
import argparse
import ctypes
import os

import torch

ap = argparse.ArgumentParser()
ap.add_argument("--mb", type=int, default=1024)
a = ap.parse_args()
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcopy_bw.so"))
lib.copy_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
n = a.mb << 20
x = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 255)
y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ms = timed(lambda: y.copy_(x))
print(f"torch copy_                       {2 * n / ms / 1e6:8.0f} GB/s")
names = ["u1", "u4", "u8", "u4 nt", "u8 nt", "u4 chunked", "u8 chunked", "u8 nt chunked"]
for v, name in enumerate(names):
    for blocks in (256, 512, 1024, 2048, 4096, 16384):
        ms = timed(lambda: lib.copy_probe(x.data_ptr(), y.data_ptr(), n, v, blocks, st))
        print(f"{name:16s} blocks={blocks:6d}   {2 * n / ms / 1e6:8.0f} GB/s")
assert torch.equal(x, y)
