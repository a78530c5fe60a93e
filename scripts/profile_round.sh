set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r3}
O=gpurun_out/$R; rm -rf $O; mkdir -p $O
# the driver's line: C2 bf16 + fp32, and the C3 / C4 / C5 (+ C5 fp32) sub-objects, HBM roofline, CPU baseline
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 100 python scripts/op_breakdown.py > $O/op_breakdown.txt 2>&1
timeout -k 10 100 python scripts/latent_profile.py > $O/c5_op_breakdown.txt 2>&1
# kernel traces: the sampler's two batch slices one after the other on one stream (TDX_SAMPLER_SERIAL), so that a kernel's duration is
# its own and not its share of a GPU it runs on together with the other slice's kernels
export TDX_SAMPLER_SERIAL=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 1 --warmup 0 --T 50 --no-cpu-baseline --no-fp32-leg --no-other-configs --no-graph > $O/kt.log 2>&1
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt32 -- python3 bench.py --dtype fp32 --steps 1 --warmup 0 --T 20 --no-cpu-baseline --no-graph > $O/kt32.log 2>&1
find $O/kt32 -name "*kernel_stats.csv" -exec cp {} $O/fp32_kernel_stats.csv \;
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt5 -- python3 bench.py --workload c5 --steps 1 --warmup 0 --T 50 --no-cpu-baseline --no-graph > $O/kt5.log 2>&1
find $O/kt5 -name "*kernel_stats.csv" -exec cp {} $O/c5_kernel_stats.csv \;
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 1 --warmup 0 --T 3 --no-cpu-baseline --no-fp32-leg --no-other-configs --no-graph > $O/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 1 --warmup 0 --T 3 --no-cpu-baseline --no-fp32-leg --no-other-configs --no-graph > $O/write.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch32 -- python3 bench.py --dtype fp32 --steps 1 --warmup 0 --T 3 --no-cpu-baseline --no-graph > $O/fetch32.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write32 -- python3 bench.py --dtype fp32 --steps 1 --warmup 0 --T 3 --no-cpu-baseline --no-graph > $O/write32.log 2>&1
unset TDX_SAMPLER_SERIAL
timeout -k 10 200 python scripts/train_bench.py --breakdown > $O/train_bench.json 2> $O/train_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktt -- python3 scripts/train_bench.py --no-graph --steps 2 --warmup 1 > $O/ktt.log 2>&1
find $O/ktt -name "*kernel_stats.csv" -exec cp {} $O/train_kernel_stats.csv \;
find $O/fetch -name "*counter_collection.csv" -exec cp {} $O/fetch.csv \;
find $O/write -name "*counter_collection.csv" -exec cp {} $O/write.csv \;
find $O/fetch32 -name "*counter_collection.csv" -exec cp {} $O/fetch32.csv \;
find $O/write32 -name "*counter_collection.csv" -exec cp {} $O/write32.csv \;
python scripts/pmc_traffic.py $O/fetch.csv $O/write.csv $O/pmc_traffic.json "B=2048, L=200, bf16" 2 > $O/pmc_traffic.log 2>&1
python scripts/pmc_traffic.py $O/fetch32.csv $O/write32.csv $O/pmc_traffic_fp32.json "B=2048, L=200, fp32" 2 > $O/pmc_traffic_fp32.log 2>&1
rm -rf $O/kt $O/kt32 $O/kt5 $O/ktt $O/fetch $O/write $O/fetch32 $O/write32
ls -la $O; head -c 600 $O/bench.json
