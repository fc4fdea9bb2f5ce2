export FEMBRAIN_BENCH_LOCAL_COMM=1 FEMBRAIN_P2P=1
for M in 2 3 4; do
FEMBRAIN_XCH_MODE=$M timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600+M)) bench.py --gpus 2 --steps 3 --warmup 1 --no-field 2>&1 | grep -E '^\{' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('mode $M', d['config']['exchange'], 'us/iter %.1f iters %.0f' % (d['us_per_cg_iteration'], d['cg_iterations_per_step']))"
done
