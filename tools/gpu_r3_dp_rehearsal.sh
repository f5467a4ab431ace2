#!/bin/bash
# one-GPU rehearsals of the data-parallel path: (a) world 1 under torchrun with RCCL and the gradient synchroniser forced
# on, both collectives; (b) two ranks on cuda:0 with gloo (ACATTN_BENCH_REHEARSAL=1).  Throughput of (b) means nothing.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3dp
export HSA_ENABLE_IPC_MODE_LEGACY=0
for coll in all_reduce reduce_scatter; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --force-grad-sync --dp-collective $coll --no-cpu-baseline --no-other-configs --kernel-kinds ragged > gpurun_out/r3dp/w1_$coll.log 2>&1
  echo "world 1 nccl $coll rc=$?"; grep "^{" gpurun_out/r3dp/w1_$coll.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['n_gpus'])"
done
for coll in all_reduce reduce_scatter; do
  ACATTN_BENCH_REHEARSAL=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 10 --warmup 3 --dp-collective $coll --no-cpu-baseline --no-other-configs --kernel-kinds ragged > gpurun_out/r3dp/w2_$coll.log 2>&1
  echo "world 2 gloo rehearsal $coll rc=$?"; grep "^{" gpurun_out/r3dp/w2_$coll.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['n_gpus'], d['scaling'])"
done
