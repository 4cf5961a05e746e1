import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from tissue_image_processing_amd import _lib, synthetic
from tissue_image_processing_amd.pipeline import FramePipeline
_lib.init(0)
Y=X=2048; Z=30
st = synthetic.make_stack(Z, Y, X, seed=100)
pipe = FramePipeline(2, Z, Y, X)
d = pipe.upload_stack(st)
for it in range(4):
    t0=time.perf_counter(); pipe.project(d); pipe.sync(); t1=time.perf_counter()
    pipe.segment(0); pipe.sync(); t2=time.perf_counter()
    pipe.cell_tables(); t3=time.perf_counter()
    print('project %.2f ms  segment %.2f ms  tables %.2f ms'%((t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3))
