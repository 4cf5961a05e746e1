"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs) into HBM traffic per kernel launch.
FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950; both counters are in KB.
usage: python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import csv, json, sys, collections


def per_kernel(path, counter):
    tot = collections.defaultdict(float)
    cnt = collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void tip::", "").replace("tip::", "").strip()
        tot[name] += float(r["Counter_Value"])
        cnt[name] += 1
    return tot, cnt


fetch, fc = per_kernel(sys.argv[1], "FETCH_SIZE")
write, wc = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for name in sorted(fetch, key=lambda n: -fetch[n]):
    if name.startswith("__amd") or name not in write:
        continue
    out[name] = {"calls": fc[name],
                 "fetch_MB_per_call_x2corrected": round(2.0 * fetch[name] / fc[name] / 1e3, 2),
                 "write_MB_per_call": round(write[name] / wc[name] / 1e3, 2)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("wrote", sys.argv[3], len(out), "kernels")
