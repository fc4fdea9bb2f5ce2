"""Development aid: per-kernel totals from a rocprofv3 (rocpd sqlite) kernel trace.  usage: kernel_stats.py <results.db> [out.csv]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = []
for name, start, end in db.execute("select name, start, end from kernels"):
    m = re.search(r"(k_\w+(<[^>]*>)?|__amd\w+)", name)
    rows.append((m.group(1) if m else name[:48], end - start))
agg = {}
for k, d in rows:
    a = agg.setdefault(k, [0, 0, 1 << 62, 0])
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values()) or 1
lines = ["kernel,calls,total_ms,avg_us,percent,min_us,max_us"]
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append('"%s",%d,%.3f,%.2f,%.2f,%.2f,%.2f' % (k, a[0], a[1] / 1e6, a[1] / a[0] / 1e3, 100.0 * a[1] / tot, a[2] / 1e3, a[3] / 1e3))
text = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
print(text)
