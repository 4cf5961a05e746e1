"""Timeline analysis of a rocprofv3 --kernel-trace CSV: how much of the small kernels' time overlaps the network's convolution kernels of other
frames.  usage: python tools/overlap.py <kernel_trace.csv>"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
# restrict to the last 60 % of the run (the timed region; the first part is warm-up / build)
lo = t0 + int(0.4 * (t1 - t0))
rows = [r for r in rows if r[0] >= lo]
span = (max(r[1] for r in rows) - rows[0][0]) / 1e6
conv = [r for r in rows if "k_unet" in r[2]]
other = [r for r in rows if "k_unet" not in r[2]]
def union(iv):
    iv = sorted(iv); out = []; 
    for a, b in iv:
        if out and a <= out[-1][1]: out[-1][1] = max(out[-1][1], b)
        else: out.append([a, b])
    return out
cu = union([(a, b) for a, b, *_ in conv]); ou = union([(a, b) for a, b, *_ in other]); au = union([(a, b) for a, b, *_ in rows])
L = lambda u: sum(b - a for a, b in u) / 1e6
def inter(u, v):
    i = j = 0; s = 0
    while i < len(u) and j < len(v):
        a = max(u[i][0], v[j][0]); b = min(u[i][1], v[j][1])
        if b > a: s += b - a
        if u[i][1] < v[j][1]: i += 1
        else: j += 1
    return s / 1e6
print("window %.1f ms: conv sum %.1f union %.1f | other sum %.1f union %.1f | any-kernel union %.1f (idle %.1f) | other∩conv %.1f ms" % (
    span, sum(b - a for a, b, *_ in conv) / 1e6, L(cu), sum(b - a for a, b, *_ in other) / 1e6, L(ou), L(au), span - L(au), inter(cu, ou)))
print("queues:", collections.Counter((r[3]) for r in rows).most_common(8))
by = collections.defaultdict(lambda: [0, 0.0])
for a, b, n, *_ in other:
    k = n.split("(")[0][:60]; by[k][0] += 1; by[k][1] += (b - a) / 1e6
for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]: print("  %-60s %5d  %.2f ms" % (k, c, t))
