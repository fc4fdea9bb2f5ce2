#!/bin/bash
# Development aid (through gpurun): kernel trace of fb_fem_resync_delta at the 56^3 cube (tools/probe_resync_delta.py), the last 1 % change listed
# kernel by kernel (tools/show_delta_trace.py)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_delta
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_delta -o kt -- python3 $R/tools/probe_resync_delta.py 56 1 2 > $R/gpurun_out/prof_delta.log 2>&1
cd $R
f=$(find gpurun_out/prof_delta -name 'kt_kernel_trace.csv' | head -1)
cp $f gpurun_out/prof_delta/kt_kernel_trace.csv 2>/dev/null
python3 tools/show_delta_trace.py > gpurun_out/delta_trace.txt 2>&1
tail -70 gpurun_out/delta_trace.txt
