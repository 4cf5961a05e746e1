#!/bin/bash
# dev tool (GPU box): SQ counters for the projection kernels, two passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/sq; mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d $out/a -o a -- python3 bench.py --workload ${1:-projection} --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline > $out/a.json 2> $out/a.err || { tail -5 $out/a.err; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES --output-format csv -d $out/b -o b -- python3 bench.py --workload ${1:-projection} --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline > $out/b.json 2> $out/b.err || { tail -5 $out/b.err; exit 1; }
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for d in ("a", "b"):
    f = glob.glob("gpurun_out/sq/%s/**/*counter_collection.csv" % d, recursive=True)[0]
    seen = set()
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void tip::", "").replace("tip::", "")
        acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
        if d == "a" and r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[n] += 1
names = sorted(acc, key=lambda n: -acc[n].get("SQ_BUSY_CYCLES", 0))[:14]
cols = ["SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_SALU", "SQ_WAVES"]
print("kernel calls " + " ".join(c.replace("SQ_", "") for c in cols))
for n in names:
    c = max(cnt[n], 1)
    print(n[:34], c, " ".join("%.3g" % (acc[n].get(k, 0) / c) for k in cols))
PY
rm -rf $out/a $out/b
