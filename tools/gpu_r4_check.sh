#!/bin/bash
# Round 4: GPU parity suite, then bench.py (default line), then `bench.py --gpus 2` WITHOUT a launcher (rehearsal on one GPU
# over gloo: the spawn path of bench.py) and the same with the rehearsal switch off (must exit non-zero on a 1-GPU box).
set -u
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r4
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q "$@" > $O/pytest_gpu.log 2>&1
rc=$?
tail -n 15 $O/pytest_gpu.log
echo "pytest rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 500 python bench.py > $O/bench_line.json 2> $O/bench.err
rc=$?
echo "bench rc=$rc"; tail -c 3000 $O/bench_line.json; echo
if [ $rc -ne 0 ]; then tail -n 20 $O/bench.err; exit $rc; fi
ACATTN_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs > $O/bench_gpus2_rehearsal.json 2> $O/bench_gpus2_rehearsal.err
echo "rehearsal --gpus 2 rc=$?"; tail -c 600 $O/bench_gpus2_rehearsal.json; echo
timeout -k 10 100 python bench.py --gpus 2 --steps 10 > $O/bench_gpus2_refused.json 2> $O/bench_gpus2_refused.err
echo "--gpus 2 on a 1-GPU box rc=$? (must be non-zero)"; tail -n 2 $O/bench_gpus2_refused.err
