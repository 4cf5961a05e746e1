#!/bin/bash
# Clock-counter trace of the U-Net convolution kernel (csrc/tip_unet_conv.h built with -DUC_TRACE): every wave of workgroup (0, 0)
# sums, over the main loop, the shader clocks (s_memtime) it spends issuing copies, in the fragment reads + products, in the
# counted vmcnt wait and in the barrier, and reports its SIMD (HW_ID); the host prints one line per wave and launch on stderr.
# The five s_memtime round trips per step inflate a step by ~400 clocks (they show up in "barrier"); the shape is what counts.
#   here:        tools/unet_trace.sh build          -> tools/ubench/lib_trace.so (travels with the gpurun snapshot)
#   on the box:  tools/unet_trace.sh run [N]        -> gpurun_out/unet_trace.log (loads the trace build through TISSUE_HIP_LIB)
set -e
cd "$(dirname "$0")/.."
case "${1:-build}" in
build)
    python3 -c "from tissue_image_processing_amd import build as b; b.build(extra_flags=['-DUC_TRACE'], out='tools/ubench/lib_trace.so')"
    ls -la tools/ubench/lib_trace.so ;;
run)
    # the trace build is loaded through TISSUE_HIP_LIB: the product library is never replaced
    TISSUE_HIP_LIB="$PWD/tools/ubench/lib_trace.so" python3 tools/unet_layers.py "${2:-2048}" > gpurun_out/unet_trace.log 2>&1 || true
    grep -c UC_TRACE gpurun_out/unet_trace.log ;;
esac
