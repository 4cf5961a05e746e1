"""dev tool: list per-launch durations of the watershed kernels of the LAST frame in a rocprofv3 kernel trace csv."""
import csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last frame = from the last k_ws_info_init on
idx = max(i for i, r in enumerate(rows) if "ws_info_init" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
tot = {}
for r in rows[idx:]:
    n = r["Kernel_Name"].split("(")[0].replace("void tip::", "")[:40]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "ws_emit" in n or len(sys.argv) > 2:
        pass
    print("%8.1f us  +%7.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
    tot[n] = tot.get(n, 0) + (e - s) / 1e3
    if "ws_emit" in n: break
print("--- totals (us)")
for n, v in sorted(tot.items(), key=lambda kv: -kv[1]): print("%8.1f  %s" % (v, n))
