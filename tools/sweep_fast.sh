#!/bin/bash
# dev tool: A/B the sigma-30 fast pass variants in ONE run (box-to-box and run-to-run noise is ~7 %)
for rep in 1 2; do
for cfg in 1616,1616 11616,11616; do
  TIP_FAST_CFG=$cfg timeout -k 10 120 python bench.py --workload projection --inflight 1 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/sw.json 2>gpurun_out/sw.err || { echo "cfg $cfg failed"; tail -3 gpurun_out/sw.err; continue; }
  echo "cfg $cfg"; python tools/kshow.py gpurun_out/sw.json score_fast_y score_fast_x
done
done
