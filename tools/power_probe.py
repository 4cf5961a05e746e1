"""Board power and clocks while the network's forward pass runs back to back (GPU box): python tools/power_probe.py [mode]"""
import os, sys, subprocess, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    os.environ["TISSUE_HIP_UNET_ARITH"] = sys.argv[1]
import torch
from tissue_image_processing_amd import prediction_local as pl
net = pl._UNet(2, torch.device("cuda", 0), dtype=torch.float32, seed=1)
x = torch.rand((1, 2, 2048, 2048), device="cuda")
net.forward(x); torch.cuda.synchronize()
stop = False
samples = []
def probe():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "-d", "0"], capture_output=True, text=True, timeout=20).stdout
            samples.append(o)
        except Exception as e:
            samples.append("ERR %r" % (e,))
        time.sleep(0.5)
th = threading.Thread(target=probe); th.start()
t0 = time.time(); n = 0
while time.time() - t0 < 12:
    for _ in range(10):
        net.forward(x)
    torch.cuda.synchronize(); n += 10
dt = time.time() - t0
stop = True; th.join()
print("%d passes, %.2f ms each" % (n, 1e3 * dt / n))
import re
for s in samples[2:8]:
    keep = [l.strip() for l in s.splitlines() if re.search(r"Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)", l)]
    print(" | ".join(keep)[:400])
