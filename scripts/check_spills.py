#!/usr/bin/env python3
"""Build check for conv_gemm.hip: no register-spill traffic inside the K loops of the persistent conv kernels.

A spill reload is a vector-memory operation; waiting for it (vmcnt) also waits for every LDS-DMA in flight, which
serialises the software pipeline.  The script compiles conv_gemm.hip to gfx950 assembly and, for every
conv_gemm_stream_kernel instantiation, counts private-memory instructions in basic blocks of the K loops: loops at depth >= 2 (the
tile loop is depth 1) whose body holds MFMA instructions (the epilogue's own small reduction loops are not K loops).
usage: python scripts/check_spills.py [--all]
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "trajectorydiffusion_amd", "csrc", "conv_gemm.hip")
out = os.path.join(tempfile.gettempdir(), "conv_gemm_gfx950.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-S",
                       "--cuda-device-only", "-o", out, src] + os.environ.get("TDX_EXTRA_FLAGS", "").split(), stderr=subprocess.DEVNULL)
name, depth, bad, total = None, 0, {}, {}
loops = {}   # (kernel, depth-2 loop header) -> [spill instructions, MFMA instructions]
hdr = None
for line in open(out):
    m = re.match(r"^(_Z\w*conv_gemm_stream_kernel\w*):", line)
    if m:
        name, depth, hdr = m.group(1), 0, None
        bad[name], total[name] = 0, 0
        continue
    if name is None:
        continue
    if line.startswith(".Lfunc_end"):
        name = None
        continue
    m = re.search(r"in Loop: Header=(\S+) Depth=(\d+)", line)
    if line.startswith(".LBB") or line.lstrip().startswith("; %bb."):
        depth = int(m.group(2)) if m else 0
        hdr = m.group(1) if m and depth == 2 else (hdr if depth > 2 else None)   # deeper blocks belong to the enclosing depth-2 loop
    if re.match(r"\s+scratch_(load|store)", line):
        total[name] += 1
        if depth >= 2:
            loops.setdefault((name, hdr), [0, 0])[0] += 1
    elif depth >= 2 and re.match(r"\s+v_mfma", line):
        loops.setdefault((name, hdr), [0, 0])[1] += 1
for (k, h), (sp, mf) in loops.items():
    if mf:
        bad[k] += sp
fail = 0
for k in sorted(bad):
    short = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", k).replace("EEEvNS_8ConvArgsE", "")
    if bad[k] or "--all" in sys.argv:
        print(f"{short:60s} spill instructions: {total[k]:4d} total, {bad[k]:4d} inside K loops")
    fail += bad[k] > 0
print(f"{len(bad)} persistent conv kernels, {fail} with spill traffic inside a K loop")
sys.exit(1 if fail else 0)
