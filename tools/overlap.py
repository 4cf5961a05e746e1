"""Timeline analysis of a rocprofv3 --kernel-trace CSV of the headline leg (bench.py --workload unet): are the network's convolution kernels
running all the time, and do the forward passes of the frames in flight run one after another or on top of each other?
usage: python tools/overlap.py <kernel_trace.csv> [label [warmup steps]]"""
import csv, sys, collections

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Queue_Id", "")))
rows.sort()
t0 = rows[0][0]
firsts = [a for a, b, n, q in rows if "k_unet_conv_first" in n]
# the timed region: the first run of forward passes (starts less than 260 ms apart) that holds warm-up + steps of them; its last `steps`
warmup, steps = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (3, 12)
runs, cur = [], [firsts[0]]
for a in firsts[1:]:
    if a - cur[-1] > 260e6:
        runs.append(cur)
        cur = []
    cur.append(a)
runs.append(cur)
run = next(r for r in runs if len(r) >= warmup + steps)
lo, hi = run[warmup], run[warmup + steps - 1]
sel = [r for r in rows if r[0] >= lo and r[1] <= hi]
frames = steps - 1

def union(iv):
    out = []
    for a, b in sorted(iv):
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


L = lambda u: sum(b - a for a, b in u) / 1e6
span = (hi - lo) / 1e6
conv = [(a, b) for a, b, n, q in sel if "k_unet" in n]
oth = [(a, b) for a, b, n, q in sel if "k_unet" not in n]
cu, au = union(conv), union(conv + oth)
label = sys.argv[2] if len(sys.argv) > 2 else ""
print("%s window %.1f ms, %d forward passes started (%.2f ms per frame under the tracer)" % (label, span, frames, span / max(frames, 1)))
print("  network kernels: sum of durations %.1f ms, union %.1f ms (%.1f %% of the window; concurrency %.2f)" % (
    sum(b - a for a, b in conv) / 1e6, L(cu), 100 * L(cu) / span, sum(b - a for a, b in conv) / 1e6 / L(cu)))
print("  other kernels:   sum %.1f ms, union %.1f ms" % (sum(b - a for a, b in oth) / 1e6, L(union(oth))))
print("  no network kernel running: %.1f ms = %.1f %% of the window; no kernel at all: %.1f ms = %.1f %%" % (
    span - L(cu), 100 * (span - L(cu)) / span, span - L(au), 100 * (span - L(au)) / span))
starts = sorted(a for a in firsts if lo <= a <= hi)
gaps = [(starts[i + 1] - starts[i]) / 1e6 for i in range(len(starts) - 1)]
print("  starts of consecutive forward passes, ms apart:", " ".join("%.0f" % g for g in gaps[:24]))
