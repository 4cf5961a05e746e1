#!/bin/bash
# dev tool (GPU box): segment-stage time for several early-endgame settings (burst index, serial steps per component)
python -m pytest tests/test_gpu_segmentation.py tests/test_gpu_pipeline.py tests/test_gpu_edge_cases.py -m gpu -x -q > gpurun_out/t.log 2>&1; tail -2 gpurun_out/t.log
for t in 1,64 2,32 2,64 2,128 2,512 3,64 3,128 9,48; do
  echo "tune $t: $(TIP_WS_TUNE=$t python tools/stage_time.py 2>&1 | tail -1)"
done
