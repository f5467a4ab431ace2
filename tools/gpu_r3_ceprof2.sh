cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3ce
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3ce/t6 -o t6 --output-format csv -- python3 $R/bench.py --items 98304 --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-full-schedule --kernel-kinds ragged --kernel-iters 5 > $R/gpurun_out/r3ce/t6.log 2>&1
grep -h "ce_" $R/gpurun_out/r3ce/t6/t6_kernel_stats.csv | cut -d, -f1-4 | cut -c40-160
