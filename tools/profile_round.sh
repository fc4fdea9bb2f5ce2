#!/bin/bash
# Development aid, run on the GPU box through gpurun: kernel trace + two PMC passes (FETCH_SIZE, WRITE_SIZE in their own
# runs, no trace domains) of the default bench workload; raw output under gpurun_out/, summaries are made afterwards by
# tools/kernel_stats.py and tools/summarize_profiles.py.
set -e
export FEMBRAIN_BENCH_SKIP_8M=${FEMBRAIN_BENCH_SKIP_8M:-1}   # the 8M-tet leg has its own kernels (k_spmv<..., true>); profile it with =0
export FEMBRAIN_BENCH_SKIP_LEGS=${FEMBRAIN_BENCH_SKIP_LEGS:-1} # (round 4) the cube27 / blob100k / scrambled legs: other handles' launches of the same kernels would mix into the per-launch sums
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
# (round 4) the field pipeline on the 256^3 sphere grid: kernel trace, then FETCH_SIZE and WRITE_SIZE of its kernels in their own passes
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_poly_kt -o kt -- python3 $R/tools/probe_poly.py > $R/gpurun_out/prof_poly_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_poly_fetch -o fetch -- python3 $R/tools/probe_poly.py > $R/gpurun_out/prof_poly_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_poly_write -o write -- python3 $R/tools/probe_poly.py > $R/gpurun_out/prof_poly_write.log 2>&1
