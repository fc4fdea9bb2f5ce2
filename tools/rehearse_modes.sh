export FEMBRAIN_BENCH_LOCAL_COMM=1 FEMBRAIN_P2P=1
for M in 2 3 4; do
FEMBRAIN_XCH_MODE=$M timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600+M)) bench.py --gpus 2 --steps 3 --warmup 1 --no-field 2>&1 | grep -E '^\{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('mode $M', d['config']['exchange'], 'us/iter %.1f iters %.0f' % (d['us_per_cg_iteration'], d['cg_iterations_per_step']))"
done
# the sharded persistent solver beside the exchange modes: both ranks on the one GPU, each on half of its CUs (the persistent grids must be
# resident together); the line's exchange_trials_ms_per_step has "sharded_persistent", config.exchange says which form ran
FEMBRAIN_BENCH_CU_SPLIT=1 FEMBRAIN_PERSIST_TIMEOUT_MS=2000 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29610 \
  bench.py --gpus 2 --steps 3 --warmup 1 --no-field > gpurun_out/bench_n2_sp.log 2>&1; echo "N=2 CU split rc=$?"
grep -E '^\{' gpurun_out/bench_n2_sp.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print('trials', c['exchange_trials_ms_per_step'], '|', c['exchange'][:40], '|', c['pcg'], '| ms/step %.1f us/iter %.1f' % (d['ms_per_step'], d['us_per_cg_iteration']), c['sharded_self_check'], c['exchange_note'], 'cube111', d.get('cube111'))"
# the big leg's own decision about the sharded persistent solver, rehearsed with a mesh whose halves it takes (the 1M-tet cube again)
FEMBRAIN_BENCH_CU_SPLIT=1 FEMBRAIN_BENCH_BIG_WORKLOAD=cube56 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
  bench.py --gpus 2 --steps 2 --warmup 1 --no-field > gpurun_out/bench_n2_sp_big.log 2>&1; echo "N=2 CU split, big leg = cube56 rc=$?"
grep -E '^\{' gpurun_out/bench_n2_sp_big.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['cube111']; print('big leg:', b.get('pcg_kernel'), 'us/iter %.1f' % b.get('us_per_cg_iteration', 0), b.get('sharded_persistent_trial'), b.get('error'))"
