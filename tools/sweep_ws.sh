#!/bin/bash
# sweep of the watershed's everyday tile flavour and opening (TIP_WS_TILE, TIP_WS_OPEN): correctness first, then time
set -o pipefail
mkdir -p gpurun_out
for v in ${VARIANTS:-0 4 1 2 3}; do
  for o in ${OPENS:-10,8 6,6 4,4}; do
    export TIP_WS_TILE=$v TIP_WS_OPEN=$o
    if [ "$o" = "${FIRST_OPEN:-10,8}" ]; then
      timeout -k 10 300 python -m pytest tests/test_gpu_segmentation.py -m gpu -x -q -k "watershed" > gpurun_out/sw_ws_t_$v.log 2>&1 || { echo "variant $v: TESTS FAILED"; tail -5 gpurun_out/sw_ws_t_$v.log; continue 2; }
    fi
    timeout -k 10 150 python bench.py --inflight 1 --steps 20 --warmup 4 --no-cpu-baseline --no-unet-leg > gpurun_out/sw_ws.json 2> gpurun_out/sw_ws.err || { echo "variant $v open $o: bench failed"; tail -3 gpurun_out/sw_ws.err; break 2; }
    python - <<PY
import json
for l in open("gpurun_out/sw_ws.json"):
    if l.startswith("{"):
        d = json.loads(l); k = d["kernels"]
        ws = sum(v["ms_per_step"] for n, v in k.items() if n.startswith("ws_") or n.startswith("uf_") or n.startswith("scan"))
        print("variant $v open $o: %.1f fps  frame %.3f ms  ws_tiles n=%.1f %.3f ms  end_resolve %.3f  watershed kernels %.3f ms" % (
            d["value"], d["ms_per_step"], k["ws_tiles"]["n_per_step"], k["ws_tiles"]["ms_per_step"], k.get("ws_end_resolve", {}).get("ms_per_step", 0), ws))
PY
  done
done
