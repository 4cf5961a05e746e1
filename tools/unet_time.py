import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tissue_image_processing_amd import prediction_local as pl
dev = torch.device("cuda", 0)
x = torch.rand((1, 2, 2048, 2048), device=dev)
for name, dtype, bench in [("fp32", torch.float32, False), ("bf16", torch.bfloat16, False), ("fp16", torch.float16, False), ("fp32+benchmark", torch.float32, True)]:
    torch.backends.cudnn.benchmark = bench
    net = pl._UNet(2, dev, dtype=dtype, seed=0)
    for _ in range(2):
        y = net.forward(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        y = net.forward(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("%-16s %.1f ms  %.1f TFLOP/s" % (name, dt * 1e3, net.flops(2048, 2048) / dt / 1e12), flush=True)
    del net, y
    torch.cuda.empty_cache()
