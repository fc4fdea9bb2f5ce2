#!/bin/bash
# Round-2 rehearsal of bench.py on a ONE-GPU box (through gpurun): the default one-GPU line, then `--gpus 2` with both
# ranks on device 0 (host-staged communicator, then peer-to-peer inboxes), then the failure path: a rank that dies in the
# middle of the 8M-tet leg must not hang the others and rank 0 must still print its line (with cube111.error or error).
# Timings of the multi-rank runs are meaningless (the ranks time-share one GPU); the point is the control flow.
export FEMBRAIN_BENCH_LOCAL_COMM=1
O=gpurun_out
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "N=1 rc=$?"; tail -c 600 $O/bench_n1.json; echo
for P2P in 0 1; do
FEMBRAIN_P2P=$P2P timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29510+P2P)) \
  bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench_n2_p2p$P2P.log 2>&1; echo "N=2 P2P=$P2P rc=$?"
grep -E '^\{' $O/bench_n2_p2p$P2P.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['exchange'], 'ms/step %.1f' % d['ms_per_step'], d['cg_iterations'], d['config']['sharded_self_check'], 'cube111', d.get('cube111'), d.get('field_tets'), d.get('error'))"
done
# failure injection: rank 1 raises inside the timed steps of the 8M-tet leg
FEMBRAIN_P2P=0 FEMBRAIN_BENCH_INJECT_FAILURE=8m timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29520 \
  bench.py --gpus 2 --steps 2 --warmup 1 --no-field > $O/bench_n2_fail.log 2>&1; echo "N=2 injected failure rc=$?"
grep -E '^\{' $O/bench_n2_fail.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value', d['value'], 'cube111', d.get('cube111'), 'error', d.get('error'))"
