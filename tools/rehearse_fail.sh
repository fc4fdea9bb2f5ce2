#!/bin/bash
# failure path of bench.py --gpus 2 on a one-GPU box: rank 1 raises inside the timed steps of the 8M-tet leg; rank 0 must
# get out of its exchange (bounded wait), both must skip the leg together, and rank 0 must still print its headline line
export FEMBRAIN_BENCH_LOCAL_COMM=1 FEMBRAIN_LOCAL_TIMEOUT_MS=8000
O=gpurun_out
for P2P in 0 1; do
FEMBRAIN_P2P=$P2P FEMBRAIN_BENCH_INJECT_FAILURE=8m timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29520+P2P)) \
  bench.py --gpus 2 --steps 2 --warmup 1 --no-field > $O/bench_n2_fail$P2P.log 2>&1; echo "N=2 P2P=$P2P injected failure rc=$?"
grep -E '^\{' $O/bench_n2_fail$P2P.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value', d['value'], 'cube111', d.get('cube111'), 'error', d.get('error'))"
done
# the same with the sharded persistent solver attached (every rank on its own half of the CUs)
FEMBRAIN_P2P=1 FEMBRAIN_BENCH_CU_SPLIT=1 FEMBRAIN_BENCH_INJECT_FAILURE=8m timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29523 \
  bench.py --gpus 2 --steps 2 --warmup 1 --no-field > $O/bench_n2_fail_sp.log 2>&1; echo "N=2 CU split, sharded persistent, injected failure rc=$?"
grep -E '^\{' $O/bench_n2_fail_sp.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value', d['value'], d['config']['exchange'][:30], 'cube111', d.get('cube111'), 'error', d.get('error'))"
