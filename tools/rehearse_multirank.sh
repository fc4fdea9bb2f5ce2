#!/bin/bash
# Rehearsal of `bench.py --gpus N` on a ONE-GPU box: N processes share device 0, gloo process group, the sharded solver
# talks through the host-staged test communicator (P2P=0) or the peer-to-peer inbox transport over HIP IPC (P2P=1).
# Timings are meaningless (the ranks time-share one GPU); the point is the control flow and the transports.
set -e
export FEMBRAIN_BENCH_LOCAL_COMM=1
for P2P in 0 1; do for N in 2 4; do
FEMBRAIN_P2P=$P2P timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500+N+10*P2P)) bench.py --gpus $N --steps 3 --warmup 1 --no-field 2>&1 | grep -E '^\{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('P2P=$P2P N=$N', d['config']['exchange'], 'ms/step %.1f us/iter %.1f iters %.0f' % (d['ms_per_step'], d['us_per_cg_iteration'], d['cg_iterations_per_step']), d['exchange_us'])"
done; done
