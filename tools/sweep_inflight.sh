#!/bin/bash
# dev tool: throughput of the default workload vs frames in flight (run on the GPU box)
for n in 1 2 3 4 5 6 8; do
  python bench.py --inflight $n --steps 36 --warmup 8 --no-cpu-baseline > gpurun_out/if.json 2>gpurun_out/if.err || { echo "inflight $n failed"; tail -3 gpurun_out/if.err; continue; }
  echo "inflight $n: $(python tools/kshow.py gpurun_out/if.json)"
done
