#!/bin/bash
# bench.py --kernel-only per library build (ACATTN_LIB list; "default" = the in-tree build)
for lib in "$@"; do
  if [ "$lib" != "default" ]; then export ACATTN_LIB=$PWD/$lib; else unset ACATTN_LIB; fi
  timeout -k 10 200 python bench.py --kernel-only > gpurun_out/libs_bench.log 2>&1 || { echo "$lib failed"; tail -n 3 gpurun_out/libs_bench.log; continue; }
  echo "$lib: $(tail -n 1 gpurun_out/libs_bench.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["avg_launch_us"], d["roofline_spatial_only"]["avg_launch_us"])')"
done
