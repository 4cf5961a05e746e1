#!/bin/bash
# Run ON THE GPU BOX (through gpurun): collects the rocprofv3 evidence for profiles/ into gpurun_out/prof/.
#   classical leg: kernel stats with one frame in flight; the default command (headline = U-Net leg + classical leg),
#   FETCH_SIZE / WRITE_SIZE in separate passes; U-Net leg: kernel stats and an MFMA-busy counter pass.
set -o pipefail
tag=${1:-r03x}
out=gpurun_out/prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# heartbeat: a profiled run can be silent for minutes (the tracer buffers everything), and a silent command is
# taken for a hung one
( while true; do date +%T >> $out/heartbeat.txt; sleep 45; done ) &
hb=$!
trap "kill $hb 2>/dev/null" EXIT
# classical leg, one frame in flight: per-kernel durations that must agree with bench.py's HIP-event figures
rocprofv3 --kernel-trace --stats --output-format csv -d $out/s1 -o s1 -- python3 bench.py --workload classical --inflight 1 --no-cpu-baseline --no-secondary-leg --no-pcie-leg > $out/${tag}_bench_classical_inflight1_profiled.json 2> $out/s1.err || exit 1
echo "stats classical inflight 1 done"
# the default command (headline = U-Net leg, classical leg secondary), shortened
rocprofv3 --kernel-trace --stats --output-format csv -d $out/s3 -o s3 -- python3 bench.py --no-cpu-baseline --steps 6 --secondary-steps 32 > $out/${tag}_bench_default_profiled.json 2> $out/s3.err || exit 1
echo "stats default done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pf -o pf -- python3 bench.py --workload classical --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --no-secondary-leg --no-pcie-leg > $out/pf.json 2> $out/pf.err || exit 1
echo "pmc fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pw -o pw -- python3 bench.py --workload classical --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --no-secondary-leg --no-pcie-leg > $out/pw.json 2> $out/pw.err || exit 1
echo "pmc write done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/su -o su -- python3 bench.py --workload unet --steps 6 --warmup 3 --no-cpu-baseline --no-secondary-leg > $out/${tag}_bench_unet_profiled.json 2> $out/su.err || exit 1
echo "stats unet done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $out/pm -o pm -- python3 bench.py --workload unet --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline --no-secondary-leg --no-pcie-leg > $out/pm.json 2> $out/pm.err || exit 1
echo "pmc mfma done"
# HBM traffic of the network's kernels (the headline's dominant kernel group): one forward pass per launch of tools/unet_layers.py
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/uf -o uf -- python3 tools/unet_layers.py 2048 > $out/uf.log 2> $out/uf.err || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/uw -o uw -- python3 tools/unet_layers.py 2048 > $out/uw.log 2> $out/uw.err || exit 1
echo "pmc unet traffic done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $out/ps -o ps -- python3 bench.py --workload projection --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --no-pcie-leg > $out/ps.json 2> $out/ps.err || exit 1
echo "pmc mfma (score passes) done"
cp $(find $out/s1 -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats_inflight1.csv
cp $(find $out/s3 -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats_default.csv
cp $(find $out/su -name "*kernel_stats.csv" | head -1) $out/${tag}_unet_kernel_stats.csv
f=$(find $out/pf -name "*counter_collection.csv" | head -1); w=$(find $out/pw -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $f $w $out/${tag}_pmc_traffic.json
python3 tools/pmc_summary.py "$(find $out/uf -name '*counter_collection.csv' | head -1)" "$(find $out/uw -name '*counter_collection.csv' | head -1)" $out/${tag}_unet_pmc_traffic.json
for pair in "$f:$out/${tag}_pmc_fetch_size.csv" "$w:$out/${tag}_pmc_write_size.csv"; do
python3 - "${pair%%:*}" "${pair##*:}" <<'PY'
import sys, csv
# keep the per-dispatch counter rows of our kernels only (the raw file also lists every runtime fill/copy)
rows = [r for r in csv.reader(open(sys.argv[1]))]
hdr, body = rows[0], rows[1:]
ki = hdr.index("Kernel_Name")
keep = [r for r in body if not r[ki].startswith("__amd")]
csv.writer(open(sys.argv[2], "w")).writerows([hdr] + keep)
PY
done
# MFMA utilisation: busy cycles of the matrix pipe / busy CU cycles, per kernel (summed over launches) -- the U-Net's kernels
# and the projection's score passes
cat > $out/mfma_busy.py <<'PY'
import sys, csv, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); calls = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0][:100]
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_BUSY_CU_CYCLES":
        dur[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); calls[n] += 1
out = {}
for n in sorted(acc, key=lambda k: -dur[k])[:25]:
    a = acc[n]
    busy = a.get("SQ_BUSY_CU_CYCLES", 0.0)
    out[n] = {"calls": calls[n], "total_ms": round(dur[n] / 1e6, 3), "SQ_VALU_MFMA_BUSY_CYCLES": a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0),
              "SQ_BUSY_CU_CYCLES": busy, "mfma_busy_over_cu_busy": round(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / busy, 4) if busy else None,
              # the matrix pipe's busy cycles are counted per SIMD (4 per CU): fraction of the busy CUs' MFMA issue capacity
              "mfma_pipe_busy_frac": round(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / busy / 4.0, 4) if busy else None}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print("wrote", sys.argv[2])
PY
python3 $out/mfma_busy.py "$(find $out/pm -name '*counter_collection.csv' | head -1)" "$out/${tag}_unet_mfma_busy.json"
python3 $out/mfma_busy.py "$(find $out/ps -name '*counter_collection.csv' | head -1)" "$out/${tag}_score_mfma_busy.json"
rm -f $out/mfma_busy.py
rm -rf $out/s1 $out/s3 $out/pf $out/pw $out/su $out/pm $out/ps $out/uf $out/uw
ls -la $out
