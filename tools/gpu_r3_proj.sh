#!/bin/bash
# round 3: hidden 128 / 256 projections -- parity tests, then the cfg4 / cfg5 step with the fused and the library projections
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r3p
[ -n "$SKIP_TESTS" ] || timeout -k 10 600 python -m pytest tests/test_hip_proj.py tests/test_hip_onehop.py -x -q -m gpu > gpurun_out/r3p/tests.log 2>&1 || { tail -40 gpurun_out/r3p/tests.log; exit 1; }
[ -n "$SKIP_TESTS" ] || tail -3 gpurun_out/r3p/tests.log
for pj in fused library; do
  timeout -k 10 300 python bench.py --config cfg4 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --kernel-kinds ragged --projections $pj > gpurun_out/r3p/cfg4_$pj.log 2>&1
  grep "^{" gpurun_out/r3p/cfg4_$pj.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg4', '$pj', d['ms_per_step'], d['value'])"
done
for pj in fused library; do
  timeout -k 10 300 python bench.py --config cfg5 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --kernel-kinds ragged --projections $pj > gpurun_out/r3p/cfg5_$pj.log 2>&1
  grep "^{" gpurun_out/r3p/cfg5_$pj.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg5', '$pj', d['ms_per_step'], d['value'])"
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/r3p/prof" -o cfg4 --output-format csv -- python "$GRAFT_REPO_ROOT/bench.py" --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --kernel-kinds ragged > "$GRAFT_REPO_ROOT/gpurun_out/r3p/prof.log" 2>&1
echo done
