cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3ce
cd /tmp && export TMPDIR=/tmp
for sp in 1 0; do
ACATTN_CE_SPLIT=$sp timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3ce/c4s$sp -o c4s$sp --output-format csv -- python3 $R/bench.py --config cfg4 --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs --no-full-schedule --kernel-kinds ragged --kernel-iters 5 > $R/gpurun_out/r3ce/c4s$sp.log 2>&1
grep -h "ce_" $R/gpurun_out/r3ce/c4s$sp/c4s${sp}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
done
