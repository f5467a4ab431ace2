#!/bin/bash
# round 3: counter passes of the forward kernel (ragged + spatial-only), then of the full-length batch, then kernel
# stats of the forward-only bench and of the training step
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r3
KINDS=ragged,spatial tools/gpu_pmc.sh r3/pmc || exit 1
KINDS=full tools/gpu_pmc.sh r3/pmc_full || exit 1
tools/gpu_profile.sh r3/fwd_stats --kernel-only --kernel-iters 300 || exit 1
tools/gpu_profile.sh r3/step_stats --steps 100 --warmup 10 --no-cpu-baseline --no-full-schedule --no-other-configs || exit 1
echo all done
