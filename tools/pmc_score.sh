#!/bin/bash
# SQ counters of the projection's kernels (score passes first): where do the cycles of the MFMA kernels go?
set -o pipefail
out=gpurun_out/pmc_score
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" "SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o p$i -- python3 bench.py --workload projection --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline > $out/p$i.json 2> $out/p$i.err || { echo "pass $i failed"; tail -3 $out/p$i.err; }
done
python3 - $out <<'PY'
import sys, csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void tip::", "").replace("tip::", "")[:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
import json
out = {}
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_BUSY_CU_CYCLES", 0))[:8]:
    a = acc[k]
    per = {c: round(v / max(1, n[k][c]), 0) for c, v in sorted(a.items())}      # per launch
    busy = per.get("SQ_BUSY_CU_CYCLES", 0.0)
    if busy:
        per["mfma_pipe_busy_frac"] = round(per.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / busy / 4.0, 4)
        per["lds_active_frac"] = round(per.get("SQ_LDS_IDX_ACTIVE", 0.0) / busy, 4)
    out[k] = per
    print(k, per)
json.dump(out, open(sys.argv[1] + "/sq_projection.json", "w"), indent=1)
PY
