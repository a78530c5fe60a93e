#!/usr/bin/env python3
"""Aggregate two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter_collection.csv each) of
`bench.py --steps 1 --warmup 0 --T 3 --no-cpu-baseline --no-graph` into profiles/rN_pmc_traffic.json: HBM bytes of the conv
GEMM kernels per denoise step and per tdx_conv_gemm call (38 calls per step; a call may be two kernel launches).

usage: python scripts/pmc_traffic.py FETCH.csv WRITE.csv OUT.json "B=2048, L=200, bf16" [slices]
B = trajectories per launch; slices = batch slices the sampler ran per denoise step (sampler.auto_shards: 2 at B=4096), so a
denoise step holds 38 x slices tdx_conv_gemm calls and 34 x slices GroupNorm launches.
gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE counts half of wide coalesced reads -> doubled;
both counters are in KiB.  Calibration: groupnorm_act reads and writes every element exactly once."""
import csv, json, sys
fetch_csv, write_csv, out, workload = sys.argv[1:5]
SLICES = int(sys.argv[5]) if len(sys.argv) > 5 else 1
CALLS_PER_STEP, GN_PER_STEP = 38 * SLICES, 34 * SLICES


def load(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        fam = "conv_gemm" if "conv_gemm" in k else ("groupnorm_act" if "groupnorm_act" in k else None)
        if fam is None:
            continue
        a = acc.setdefault(fam, {"launches": 0, "sum_KiB": 0.0})
        a["launches"] += 1
        a["sum_KiB"] += float(r["Counter_Value"])
    return acc


raw = {"FETCH_SIZE": load(fetch_csv, "FETCH_SIZE"), "WRITE_SIZE": load(write_csv, "WRITE_SIZE")}
steps = raw["FETCH_SIZE"]["groupnorm_act"]["launches"] / GN_PER_STEP
rd = 2 * raw["FETCH_SIZE"]["conv_gemm"]["sum_KiB"] * 1024 / steps
wr = raw["WRITE_SIZE"]["conv_gemm"]["sum_KiB"] * 1024 / steps
gn_rd = 2 * raw["FETCH_SIZE"]["groupnorm_act"]["sum_KiB"] * 1024 / steps
gn_wr = raw["WRITE_SIZE"]["groupnorm_act"]["sum_KiB"] * 1024 / steps
B = int(workload.split("B=")[1].split(",")[0])
es = 2 if "bf16" in workload else 4
import datetime
import socket
res = {
    "measured_on": "%s, host %s (scripts/profile_round.sh)" % (datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%d %H:%M UTC"), socket.gethostname()),
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 0 --T 3 --no-cpu-baseline --no-graph",
    "workload": workload, "slices_per_denoise_step": SLICES, "raw": raw, "denoise_steps_covered": steps,
    "gfx950_correction": "FETCH_SIZE doubled; calibration on groupnorm_act: 2*FETCH = %.2f GB/step, WRITE = %.2f GB/step vs %.2f GB algorithmic each"
                         % (gn_rd / 1e9, gn_wr / 1e9, 599040 * es * B * SLICES / 1e9),
    "conv_gemm_kernel_launches_per_step": raw["FETCH_SIZE"]["conv_gemm"]["launches"] / steps,
    "conv_gemm_bytes_per_denoise_step": {"read": rd, "write": wr, "total": rd + wr},
    "conv_gemm_bytes_per_launch_avg": (rd + wr) / CALLS_PER_STEP,
    "conv_gemm_algorithmic_bytes_per_denoise_step": {"read_activations_once": 920976 * es * B * SLICES, "write": 831376 * es * B * SLICES,
                                                     "weights": 47.5e6 * es},
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: res[k] for k in ("denoise_steps_covered", "conv_gemm_bytes_per_denoise_step", "conv_gemm_bytes_per_launch_avg", "gfx950_correction")}))
