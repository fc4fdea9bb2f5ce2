#!/bin/bash
# Development aid, run on the GPU box through gpurun: kernel trace + two PMC passes (FETCH_SIZE, WRITE_SIZE in their own
# runs, no trace domains) of the default bench workload; raw output under gpurun_out/, summaries are made afterwards by
# tools/kernel_stats.py and tools/summarize_profiles.py.
set -e
export FEMBRAIN_BENCH_SKIP_8M=${FEMBRAIN_BENCH_SKIP_8M:-1}   # the 8M-tet leg has its own kernels (k_spmv<..., true>); profile it with =0
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 5 --warmup 1 --no-cpu-baseline"
# (every pass writes the iteration count of each persistent launch, in launch order: the summaries divide per-launch figures by it)
export FEMBRAIN_BENCH_LAUNCH_LOG=$R/gpurun_out/prof_kt_launches.json
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -o kt -- python3 $ARGS > $R/gpurun_out/prof_kt.log 2>&1
export FEMBRAIN_BENCH_LAUNCH_LOG=$R/gpurun_out/prof_fetch_launches.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -o fetch -- python3 $ARGS --no-field > $R/gpurun_out/prof_fetch.log 2>&1
export FEMBRAIN_BENCH_LAUNCH_LOG=$R/gpurun_out/prof_write_launches.json
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -o write -- python3 $ARGS --no-field > $R/gpurun_out/prof_write.log 2>&1
ls -R $R/gpurun_out/prof_kt $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write | head -30
