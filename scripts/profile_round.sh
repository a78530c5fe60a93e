set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r2}
O=gpurun_out/$R; rm -rf $O; mkdir -p $O
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 400 python bench.py --sampler ddim --batch 65536 --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-leg > $O/bench_ddim_B65536.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --workload c5 --steps 1 --warmup 1 > $O/bench_c5.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --workload c5 --dtype fp32 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_c5_fp32.json 2>> $O/bench.err
timeout -k 10 100 python scripts/op_breakdown.py > $O/op_breakdown.txt 2>&1
timeout -k 10 100 python scripts/latent_profile.py > $O/c5_op_breakdown.txt 2>&1
# kernel traces: the sampler's two batch slices one after the other on one stream (TDX_SAMPLER_SERIAL), so that a kernel's duration is
# its own and not its share of a GPU it runs on together with the other slice's kernels
export TDX_SAMPLER_SERIAL=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 1 --warmup 0 --T 50 --no-cpu-baseline --no-fp32-leg --no-graph > $O/kt.log 2>&1
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt5 -- python3 bench.py --workload c5 --steps 1 --warmup 0 --T 50 --no-cpu-baseline --no-graph > $O/kt5.log 2>&1
find $O/kt5 -name "*kernel_stats.csv" -exec cp {} $O/c5_kernel_stats.csv \;
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 1 --warmup 0 --T 3 --no-cpu-baseline --no-fp32-leg --no-graph > $O/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 1 --warmup 0 --T 3 --no-cpu-baseline --no-fp32-leg --no-graph > $O/write.log 2>&1
unset TDX_SAMPLER_SERIAL
timeout -k 10 200 python scripts/train_bench.py --breakdown > $O/train_bench.json 2> $O/train_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktt -- python3 scripts/train_bench.py --no-graph --steps 2 --warmup 1 > $O/ktt.log 2>&1
find $O/ktt -name "*kernel_stats.csv" -exec cp {} $O/train_kernel_stats.csv \;
find $O/fetch -name "*counter_collection.csv" -exec cp {} $O/fetch.csv \;
find $O/write -name "*counter_collection.csv" -exec cp {} $O/write.csv \;
rm -rf $O/kt $O/kt5 $O/ktt $O/fetch $O/write
ls -la $O; head -c 600 $O/bench.json
