"""Wall time of every stage of the U-Net leg, one frame in flight, stages separated by synchronisation (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tissue_image_processing_amd import _lib, synthetic
from tissue_image_processing_amd.pipeline import FramePipeline
from tissue_image_processing_amd.prediction_local import SegmentationPredictor
_lib.init(0)
Z, Y, X = 30, 2048, 2048
st = synthetic.make_stack(Z, Y, X, seed=100)
pipe = FramePipeline(2, Z, Y, X, reference_channel=0, airyscan=False, use_torch=True)
pred = SegmentationPredictor(None, (2, X, Y), device=0)
d = pipe.upload_stack(st)
pipe.project(d); pipe.sync()
pj = pipe._proj_t
padded, _ = pred.prepare_image(torch.stack([pj[1].T, pj[0].T]))
pred.model.calibrate_head(padded, 0.5)
def sync():
    pipe.sync(); torch.cuda.synchronize()
acc = {}
def timed(name, fn):
    sync(); t0 = time.perf_counter(); r = fn(); sync(); acc.setdefault(name, []).append(1e3 * (time.perf_counter() - t0)); return r
for it in range(6):
    timed("project", lambda: pipe.project(d))
    img = timed("stack planes", lambda: torch.stack([pj[1].T, pj[0].T]))
    padded, npad = timed("prepare_image", lambda: pred.prepare_image(img))
    prob = timed("forward", lambda: pred.model.forward(padded))
    p0 = prob[:, :, npad[1][0]:, npad[2][0]:][0, 0]
    lab, hc = timed("tail", lambda: pred.segment_probability(p0, return_device=True))
    timed("cell tables", lambda: pipe.cell_tables(labels_ptr=lab.data_ptr(), shape=(X, Y)))
for k, v in acc.items():
    print("%-16s %8.2f ms (median of %d, first %.2f)" % (k, float(np.median(v[1:])), len(v) - 1, v[0]))
print("sum %.2f ms" % sum(float(np.median(v[1:])) for v in acc.values()))
