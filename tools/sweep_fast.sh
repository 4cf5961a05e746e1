#!/bin/bash
# dev tool: time the sigma-30 fast passes for every (waves, outputs-per-lane) configuration (run on the GPU box)
for cfg in 1608,1608 1616,1616 816,816 832,832 432,432; do
  TIP_FAST_CFG=$cfg python bench.py --workload projection --inflight 1 --steps 10 --warmup 2 > gpurun_out/sw.json 2>gpurun_out/sw.err || { echo "cfg $cfg failed"; tail -3 gpurun_out/sw.err; continue; }
  echo "cfg $cfg"; python tools/kshow.py gpurun_out/sw.json score_fast_y score_fast_x
done
