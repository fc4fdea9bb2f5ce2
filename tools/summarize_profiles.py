"""Turns raw rocprofv3 output under gpurun_out/ into the small summaries committed under profiles/ (development aid).

usage: summarize_profiles.py <kernel_stats.csv> <fetch counter csv> <write counter csv> <tag>"""
import collections
import csv
import json
import re
import statistics
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fembrain_amd.lib import source_sha256

stats, fetch, write, tag = sys.argv[1:5]
rows = list(csv.DictReader(open(stats)))
with open("profiles/%s_kernel_stats.csv" % tag, "w") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "calls", "total_ms", "avg_us", "percent", "min_us", "max_us"])
    for r in rows:
        m = re.search(r"(k_\w+(<[^>]*>)?|__amd\w+)", r["Name"])
        name = m.group(1) if m else r["Name"][:40]
        w.writerow([name, r["Calls"], "%.3f" % (float(r["TotalDurationNs"]) / 1e6), "%.2f" % (float(r["AverageNs"]) / 1e3), r["Percentage"],
                    "%.2f" % (float(r["MinNs"]) / 1e3), "%.2f" % (float(r["MaxNs"]) / 1e3)])


def agg(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        m = re.search(r"(k_\w+(<[^>]*>)?|__amd\w+)", r["Kernel_Name"])
        d[m.group(1) if m else r["Kernel_Name"][:40]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                                                       int(r["VGPR_Count"]), int(r["SGPR_Count"]), int(r["LDS_Block_Size"])))
    return d


f, wv = agg(fetch), agg(write)
out = []
for k in f:
    fv = [x for x in f[k] if x[1] > 3000]
    ww = [x for x in wv.get(k, []) if x[1] > 3000]
    if not fv:
        continue
    out.append(dict(kernel=k, launches=len(fv), FETCH_SIZE_KB_median=statistics.median(x[0] for x in fv),
                    WRITE_SIZE_KB_median=statistics.median(x[0] for x in ww) if ww else None,
                    duration_us_median=statistics.median(x[1] for x in fv) / 1e3, vgpr=fv[0][2], sgpr=fv[0][3], lds=fv[0][4]))
with open("profiles/%s_pmc_summary.csv" % tag, "w") as fh:
    wr = csv.DictWriter(fh, fieldnames=list(out[0].keys()))
    wr.writeheader()
    wr.writerows(out)
# the dominant kernel of the PCG: the persistent launch where the handle runs it, else the SpMV of the two-launch iteration
def per_launch(path, pat):
    """counter value and duration of every dispatch of the kernels matching `pat`, in dispatch order"""
    rows = [r for r in csv.DictReader(open(path)) if pat in r["Kernel_Name"].replace(" ", "")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return [(float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows]


pipe = [r for r in out if r["kernel"].replace(" ", "").startswith("k_pcg_pipe<float")]
if pipe:
    # ONE launch = one whole solve; launches differ in their iteration counts, so the counters are divided by the iterations the
    # same launches ran (bench.py wrote them, FEMBRAIN_BENCH_LAUNCH_LOG): bytes and time per PCG iteration
    fl_, wl_ = per_launch(fetch, "k_pcg_pipe<float"), per_launch(write, "k_pcg_pipe<float")
    fit = json.load(open(os.environ.get("FETCH_LAUNCHES", "gpurun_out/prof_fetch_launches.json")))["k_pcg_pipe_launch_iterations"]
    wit = json.load(open(os.environ.get("WRITE_LAUNCHES", "gpurun_out/prof_write_launches.json")))["k_pcg_pipe_launch_iterations"]
    assert len(fit) == len(fl_) and len(wit) == len(wl_), (len(fit), len(fl_), len(wit), len(wl_))
    fetch_kb_it = sum(x[0] for x in fl_) / sum(fit)
    write_kb_it = sum(x[0] for x in wl_) / sum(wit)
    us_it = sum(x[1] for x in fl_) / 1e3 / sum(fit)
    hbm_it = (2 * fetch_kb_it + write_kb_it) * 1024
    json.dump({"kernel": pipe[0]["kernel"], "workload": "cube56 (998,250 tets), f32 matrix",
               "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-field` "
                         "(tools/profile_round.sh); sums over the %d launches of the pass divided by the %d PCG iterations they ran" % (len(fl_), sum(fit)),
               "launches": len(fl_), "iterations": sum(fit), "FETCH_SIZE_KB_per_iteration": fetch_kb_it, "WRITE_SIZE_KB_per_iteration": write_kb_it,
               "us_per_iteration_under_pmc": us_it,
               "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
               "note": "k_pcg_pipe: one launch = one whole solve (all PCG iterations of a step); its traffic per iteration is below the algorithmic "
                       "bytes of an iteration because the vectors and part of the matrix stay in registers / LDS",
               "hbm_bytes_per_unit": hbm_it, "kernel_source_sha256": source_sha256("fem")}, open("profiles/dominant_pmc.json", "w"), indent=1)
    print(open("profiles/dominant_pmc.json").read())
    for r in out:
        print(r["kernel"][:40], r["launches"], "F %.0f KB W %s KB %.1f us" % (r["FETCH_SIZE_KB_median"], r["WRITE_SIZE_KB_median"], r["duration_us_median"]))
    sys.exit(0)
cands = [r for r in out if r["kernel"].replace(" ", "").startswith("k_spmv<float,3")]
sp = max(cands, key=lambda r: r["launches"] * r["duration_us_median"])
hbm = (2 * sp["FETCH_SIZE_KB_median"] + sp["WRITE_SIZE_KB_median"]) * 1024
json.dump({"kernel": sp["kernel"], "workload": "cube56 (998,250 tets), f32 matrix",
           "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-field` (tools/profile_round.sh); per-launch medians over %d launches of at least 3 us" % sp["launches"],
           "FETCH_SIZE_KB": sp["FETCH_SIZE_KB_median"], "WRITE_SIZE_KB": sp["WRITE_SIZE_KB_median"], "duration_us_median": sp["duration_us_median"],
           "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
           "hbm_bytes_per_launch": hbm,
           "kernel_source_sha256": source_sha256("fem")}, open("profiles/dominant_pmc.json", "w"), indent=1)
print(open("profiles/dominant_pmc.json").read())
for r in out:
    print(r["kernel"][:40], r["launches"], "F %.0f KB W %s KB %.1f us" % (r["FETCH_SIZE_KB_median"], r["WRITE_SIZE_KB_median"], r["duration_us_median"]))
