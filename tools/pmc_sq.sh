#!/bin/bash
# dev tool (GPU box): SQ counters for the kernels of a workload, two passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/sq; mkdir -p $out
A="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"
B="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_ANY"
rocprofv3 --kernel-trace --pmc $A --output-format csv -d $out/a -o a -- python3 bench.py --workload ${1:-projection} --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline > $out/a.json 2> $out/a.err || { tail -5 $out/a.err; exit 1; }
rocprofv3 --kernel-trace --pmc $B --output-format csv -d $out/b -o b -- python3 bench.py --workload ${1:-projection} --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline > $out/b.json 2> $out/b.err || { tail -5 $out/b.err; exit 1; }
python3 - $A $B <<'PY'
import csv, glob, collections, sys
cols = sys.argv[1:]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int); dur = collections.defaultdict(float)
for d in ("a", "b"):
    f = glob.glob("gpurun_out/sq/%s/**/*counter_collection.csv" % d, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void tip::", "").replace("tip::", "")
        acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[n] += 1; dur[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
names = sorted(acc, key=lambda n: -dur[n])[:12]
print("kernel calls dur_us " + " ".join(c.replace("SQ_", "") for c in cols))
for n in names:
    c = max(cnt[n], 1)
    print(n[:34], c, "%.0f" % (dur[n] / c / 1e3), " ".join("%.3g" % (acc[n].get(k, 0) / c) for k in cols))
PY
rm -rf $out/a $out/b
