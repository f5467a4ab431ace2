set -e
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3q; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/r3q/l200" -o l200 --output-format csv -- python3 "$R/bench.py" --seq-len 200 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --kernel-kinds ragged --kernel-iters 20 > "$R/gpurun_out/r3q/l200.log" 2>&1
grep "^{" "$R/gpurun_out/r3q/l200.log" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
