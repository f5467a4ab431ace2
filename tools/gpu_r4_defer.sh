#!/bin/bash
# deferred weight-gradient reductions: suites that train through the trainer, then the step with and without
cd ${GRAFT_REPO_ROOT:-.}; O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_backward.py tests/test_hip_model_surface.py tests/test_hip_combined.py tests/test_hip_bert4rec.py tests/test_hip_linear.py tests/test_hip_tail.py -x -q 2>&1 | tail -3 || exit 1
bash tools/gpu_r4_env_ab.sh 2 none ACATTN_NO_DEFER=1
