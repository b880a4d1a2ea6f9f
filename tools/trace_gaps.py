"""Dev tool: from a rocprofv3 kernel_trace.csv, report per-kernel average duration and the idle gaps
between consecutive kernels on the stream."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = collections.defaultdict(list)
gaps = []
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void bssm::", "").replace("bssm::", "")
    dur[name].append(e - s)
    if prev_end is not None:
        gaps.append(s - prev_end)
    prev_end = e
tot = sum(sum(v) for v in dur.values())
print("kernels: %d, busy %.3f ms, span %.3f ms" % (len(rows), tot / 1e6, (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("  %-40s n=%6d avg %8.2f us  total %8.2f ms" % (k[:40], len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
g = sorted(gaps)
print("gaps: avg %.2f us, median %.2f us, p90 %.2f us" % (sum(g) / len(g) / 1e3, g[len(g) // 2] / 1e3, g[int(len(g) * 0.9)] / 1e3))
