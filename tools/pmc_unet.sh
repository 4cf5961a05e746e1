#!/bin/bash
# dev tool (GPU box): SQ counters of the U-Net convolution kernel over one forward pass (tools/unet_layers.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/squ; mkdir -p $out
A="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES"
B="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA"
rocprofv3 --kernel-trace --pmc $A --output-format csv -d $out/a -o a -- python3 tools/unet_layers.py ${1:-2048} ${2:-bf16x3} > $out/a.log 2> $out/a.err || { tail -5 $out/a.err; exit 1; }
rocprofv3 --kernel-trace --pmc $B --output-format csv -d $out/b -o b -- python3 tools/unet_layers.py ${1:-2048} ${2:-bf16x3} > $out/b.log 2> $out/b.err || { tail -5 $out/b.err; exit 1; }
python3 - $A $B <<'PY'
import csv, glob, collections, sys
cols = sys.argv[1:]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int); dur = collections.defaultdict(float)
for d in ("a", "b"):
    f = glob.glob("gpurun_out/squ/%s/**/*counter_collection.csv" % d, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void tip::", "").replace("tip::", "")
        acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_INSTS_LDS"): cnt[n, r["Counter_Name"]] += 1
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": dur[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
names = sorted(dur, key=lambda n: -dur[n])[:6]
for n in names:
    print(n[:60], "total ms %.2f" % (dur[n] / 1e6))
    for k in cols:
        print("   %-28s %.4g" % (k, acc[n].get(k, 0)))
PY
rm -rf $out/a $out/b
