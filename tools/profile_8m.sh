#!/bin/bash
# Development aid (GPU box, through gpurun): the two-launch PCG iteration at 8M tets (BASELINE config 5 on one GPU) -- kernel trace and
# the two HBM counters in their own passes; summarised by hand into profiles/r03_cube111_iteration_breakdown.json (tools/summarize_profiles.py for the kernel stats)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --workload cube111 --steps 2 --warmup 1 --no-cpu-baseline --no-field"
export FEMBRAIN_BENCH_SKIP_8M=1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof8_kt -o kt -- python3 $ARGS > $R/gpurun_out/prof8_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof8_fetch -o fetch -- python3 $ARGS > $R/gpurun_out/prof8_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof8_write -o write -- python3 $ARGS > $R/gpurun_out/prof8_write.log 2>&1
ls $R/gpurun_out/prof8_kt $R/gpurun_out/prof8_fetch $R/gpurun_out/prof8_write
