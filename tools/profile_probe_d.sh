#!/bin/bash
# Development aid (through gpurun): kernel trace of tools/probe_numbering.py case d (the 606k-tet Delaunay mesh), assembly and solver kernels listed
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_d
PROBE_ONLY=d rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_d -o kt -- python3 $R/tools/probe_numbering.py 56 > $R/gpurun_out/prof_d.log 2>&1
cd $R
f=$(find gpurun_out/prof_d -name 'kt_kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-90s calls %6s  avg %10.1f us  total %10.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
