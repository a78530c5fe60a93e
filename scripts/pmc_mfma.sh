# MFMA / LDS utilisation counters of the conv kernels (one --pmc pass; run via gpurun): bash scripts/pmc_mfma.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r3}
export TDX_SAMPLER_SERIAL=1   # the sampler's batch slices one after the other on one stream
O=gpurun_out/${R}m; rm -rf $O; mkdir -p $O
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc -- python3 bench.py --steps 1 --warmup 0 --T 3 --no-cpu-baseline --no-fp32-leg --no-other-configs --no-graph > $O/pmc.log 2>&1
find $O/pmc -name "*counter_collection.csv" -exec cp {} $O/mfma.csv \;
rm -rf $O/pmc
ls -la $O; tail -3 $O/pmc.log
