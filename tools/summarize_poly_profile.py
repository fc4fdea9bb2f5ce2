"""Turns the raw rocprofv3 output of the field pipeline (tools/profile_round.sh, last three passes) into profiles/<tag>_poly256_kernel_stats.csv
and profiles/<tag>_poly256_pmc.json (per kernel: launches, median FETCH_SIZE / WRITE_SIZE in KB, bytes with the gfx950 correction).
usage: summarize_poly_profile.py <kernel_stats.csv> <fetch counter csv> <write counter csv> <tag>"""
import collections
import csv
import json
import os
import re
import statistics
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fembrain_amd.lib import source_sha256  # noqa: E402

stats, fetch, write, tag = sys.argv[1:5]


def short(name):
    m = re.search(r"(k_\w+(<[^>]*>)?|__amd\w+)", name)
    return m.group(1) if m else name[:40]


with open("profiles/%s_poly256_kernel_stats.csv" % tag, "w") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "calls", "total_ms", "avg_us", "percent", "min_us", "max_us"])
    for r in csv.DictReader(open(stats)):
        w.writerow([short(r["Name"]), r["Calls"], "%.3f" % (float(r["TotalDurationNs"]) / 1e6), "%.2f" % (float(r["AverageNs"]) / 1e3), r["Percentage"],
                    "%.2f" % (float(r["MinNs"]) / 1e3), "%.2f" % (float(r["MaxNs"]) / 1e3)])


def agg(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[short(r["Kernel_Name"])].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return d


f, wv = agg(fetch), agg(write)
out = {}
for k in sorted(set(f) | set(wv)):
    fk, wk = f.get(k, []), wv.get(k, [])
    if not fk or len(fk) < 5:
        continue
    fe, wr = statistics.median(x[0] for x in fk), statistics.median(x[0] for x in wk) if wk else 0.0
    out[k] = {"launches": len(fk), "FETCH_SIZE_KB": fe, "WRITE_SIZE_KB": wr, "bytes": (2 * fe + wr) * 1024,
              "duration_us_median_under_pmc": statistics.median(x[1] for x in fk) / 1e3}
json.dump({"workload": "sphere.blob, 256^3 grid: sweep + classification + ranks + tet vertices + tet elements (tools/probe_poly.py, 31 pipeline runs)",
           "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact for 16-byte-per-lane stores",
           "kernels": out, "kernel_source_sha256": source_sha256("poly")}, open("profiles/%s_poly256_pmc.json" % tag, "w"), indent=1)
print(open("profiles/%s_poly256_pmc.json" % tag).read())
