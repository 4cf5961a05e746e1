"""Per-layer time of the hand-written U-Net forward pass (GPU box): python tools/unet_layers.py [N] [mode].
UNET_LAYERS_BN_SHIFT=0.3 gives every BatchNorm a shift (dense activations instead of the random-init network's half-zero ones)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
if len(sys.argv) > 2:
    os.environ["TISSUE_HIP_UNET_ARITH"] = sys.argv[2]
import torch
from tissue_image_processing_amd import prediction_local as pl
net = pl._UNet(2, torch.device("cuda", 0), dtype=torch.float32, seed=1)
if os.environ.get("UNET_LAYERS_BN_SHIFT"):
    # the random-init network has identity BatchNorm statistics (shift 0): half of every stored activation tensor is exact zeros.
    # A shift makes them dense, as a trained network's are -- the matrix cores' power-limited clock depends on the operand bits.
    for k in net.p:
        if k.endswith(".t"):
            net.p[k] = torch.full_like(net.p[k], float(os.environ["UNET_LAYERS_BN_SHIFT"]))
x = torch.rand((1, 2, N, N), device="cuda")
for _ in range(2):
    net.forward(x)
torch.cuda.synchronize()
net.trace = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
net.forward(x)
e1.record()
torch.cuda.synchronize()
tot = 0.0
for name, flop, a, b in net.trace:
    ms = a.elapsed_time(b)
    tot += ms
    print("%-46s %8.3f ms  %7.1f TFLOP/s" % (name, ms, flop / ms / 1e9))
print("sum %.2f ms, wall %.2f ms, %.1f TFLOP/s useful" % (tot, e0.elapsed_time(e1), net.flops(N, N) / e0.elapsed_time(e1) / 1e9))
