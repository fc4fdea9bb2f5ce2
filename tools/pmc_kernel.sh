#!/bin/bash
# Development aid (GPU box): one PMC pass of a python script, per-kernel sums printed for kernels matching a pattern.
# usage: tools/pmc_kernel.sh "<counters>" <kernel regex> <script.py> [args]
set -e
R=$GRAFT_REPO_ROOT
CNT="$1"; PAT="$2"; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_k
rocprofv3 --pmc $CNT --output-format csv -d $R/gpurun_out/pmc_k -o k -- python3 "$@" > $R/gpurun_out/pmc_k.log 2>&1
cd $R
python3 - "$PAT" <<'PY'
import csv, glob, re, sys, collections
pat = re.compile(sys.argv[1])
f = glob.glob("gpurun_out/pmc_k/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if not pat.search(k): continue
    name = re.sub(r"\(.*", "", k)[:60]
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (name, r["Dispatch_Id"])
    if key not in seen: seen.add(key); n[name] += 1
for name, d in acc.items():
    print(name, "dispatches", n[name])
    for c, v in sorted(d.items()): print("   %-28s %.4g per dispatch" % (c, v / n[name]))
PY
