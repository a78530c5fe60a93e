#!/usr/bin/env python3
"""Aggregate the rocprofv3 --pmc pass of scripts/pmc_mfma.sh (counter_collection.csv) per kernel family.
usage: python scripts/pmc_mfma.py gpurun_out/r1m/mfma.csv profiles/r1_pmc_mfma_lds.json
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter
sums the 8 XCDs; MI355X_MICROARCH.md, DVFS note).  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles of wave lifetime."""
import collections
import csv
import json
import sys

src, dst = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.Counter()
for r in csv.DictReader(open(src)):
    k = r["Kernel_Name"]
    fam = "conv_gemm" if "conv_gemm" in k else ("groupnorm_act" if "groupnorm" in k else "other")
    acc[fam][r["Counter_Name"]] += float(r["Counter_Value"])
    launches[fam] += r["Counter_Name"] == "SQ_WAVE_CYCLES"
out = {"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT "
                  "SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 1 --warmup 0 --T 3 --no-cpu-baseline --no-fp32-leg --no-graph (TDX_SAMPLER_SERIAL=1)",
       "workload": "B=4096 as 2 slices of 2048 in series, L=200, bf16 (T=3 denoise steps + the roofline pass)", "families": {}}
for fam, c in acc.items():
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    wc = c["SQ_WAVE_CYCLES"] or 1.0
    out["families"][fam] = {
        "kernel_launches": launches[fam], "kernel_cycles_sum": cyc, "raw": dict(c),
        "mfma_utilisation": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024) if cyc else None,
        "lds_bank_conflict_share_of_lds_cycles": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"] if c["SQ_LDS_IDX_ACTIVE"] else None,
        "wave_lifetime_split": {"parked_waitcnt_or_barrier": c["SQ_WAIT_ANY"] / wc, "issue_stalled": c["SQ_WAIT_INST_ANY"] / wc,
                                "issuing": c["SQ_ACTIVE_INST_ANY"] / wc}}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "raw"} for k, v in out["families"].items()}, indent=1))
