#!/bin/bash
# Run ON THE GPU BOX (through gpurun): collects the rocprofv3 evidence for profiles/ into gpurun_out/prof/.
#   kernel stats with one frame in flight and with the default command, then FETCH_SIZE / WRITE_SIZE in separate passes
set -o pipefail
tag=${1:-r01x}
out=gpurun_out/prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/s1 -o s1 -- python3 bench.py --inflight 1 --no-cpu-baseline > $out/${tag}_bench_inflight1_profiled.json 2> $out/s1.err || exit 1
echo "stats inflight 1 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/s3 -o s3 -- python3 bench.py --no-cpu-baseline > $out/${tag}_bench_default_profiled.json 2> $out/s3.err || exit 1
echo "stats default done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pf -o pf -- python3 bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline > $out/pf.json 2> $out/pf.err || exit 1
echo "pmc fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pw -o pw -- python3 bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline > $out/pw.json 2> $out/pw.err || exit 1
echo "pmc write done"
cp $(find $out/s1 -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats_inflight1.csv
cp $(find $out/s3 -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats_default_inflight3.csv
f=$(find $out/pf -name "*counter_collection.csv" | head -1); w=$(find $out/pw -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $f $w $out/${tag}_pmc_traffic.json
python3 - "$f" "$out/${tag}_pmc_fetch_size.csv" <<'PY'
import sys, csv
# keep the per-dispatch counter rows of our kernels only (the raw file also lists every runtime fill/copy)
rows = [r for r in csv.reader(open(sys.argv[1]))]
hdr, body = rows[0], rows[1:]
ki = hdr.index("Kernel_Name")
keep = [r for r in body if not r[ki].startswith("__amd")]
csv.writer(open(sys.argv[2], "w")).writerows([hdr] + keep)
PY
python3 - "$w" "$out/${tag}_pmc_write_size.csv" <<'PY'
import sys, csv
rows = [r for r in csv.reader(open(sys.argv[1]))]
hdr, body = rows[0], rows[1:]
ki = hdr.index("Kernel_Name")
keep = [r for r in body if not r[ki].startswith("__amd")]
csv.writer(open(sys.argv[2], "w")).writerows([hdr] + keep)
PY
rm -rf $out/s1 $out/s3 $out/pf $out/pw
ls -la $out
