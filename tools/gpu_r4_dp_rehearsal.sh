#!/bin/bash
# Round 4: the data-parallel code path on the one-GPU box -- (a) world 1 under torch.distributed.run with RCCL and the
# synchroniser forced on (both collectives; the optimizer launch is its own captured graph now), (b) two and four ranks on ONE
# GPU over gloo through bench.py's own rank launcher (ACATTN_BENCH_REHEARSAL=1), (c) the one-walk mode under (a).
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r4/dp; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { tag=$1; shift; "$@" > $O/$tag.json 2> $O/$tag.err; echo "$tag rc=$?"; python3 -c "
import json,sys
try:
    d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); print('   ', {k:d[k] for k in ('value','n_gpus','rccl_ranks','collective_backend','ms_per_step')}, d['config']['final_losses'], d['config']['backward'])
except Exception as e: print('    no line:', e)"; }
B="--steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --no-full-schedule --kernel-iters 20"
run plain timeout -k 10 200 python bench.py $B
run w1_allreduce timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force-grad-sync $B
run w1_reduce_scatter timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --force-grad-sync --dp-collective reduce_scatter $B
run w1_onewalk timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --force-grad-sync --combined-backward $B
ACATTN_BENCH_REHEARSAL=1 run r2_gloo timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --no-full-schedule --kernel-iters 20
ACATTN_BENCH_REHEARSAL=1 run r4_gloo timeout -k 10 400 python bench.py --gpus 4 --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-full-schedule --kernel-iters 20
ACATTN_BENCH_REHEARSAL=1 run r2_gloo_cfg4 timeout -k 10 400 python bench.py --gpus 2 --config cfg4 --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-full-schedule --kernel-iters 10
