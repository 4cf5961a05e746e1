#!/bin/bash
# dev tool (GPU box): watershed tests, then a rocprofv3 kernel trace of tools/stage_time.py summarised per launch
set -o pipefail
python -m pytest tests/test_gpu_segmentation.py tests/test_gpu_pipeline.py tests/test_gpu_edge_cases.py -m gpu -x -q > gpurun_out/t.log 2>&1; tail -2 gpurun_out/t.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d gpurun_out/kt -o kt --output-format csv -- python tools/stage_time.py > gpurun_out/kt.log 2>&1
tail -2 gpurun_out/kt.log
python tools/ws_launches.py gpurun_out/kt > gpurun_out/ws_launches.txt 2>&1
grep -E "k_ws_tiles" gpurun_out/ws_launches.txt | head -7
sed -n '/--- totals/,$p' gpurun_out/ws_launches.txt | head -8
rm -rf gpurun_out/kt
