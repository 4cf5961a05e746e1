import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tissue_image_processing_amd import synthetic, surface_projection as sp, basic_image_manipulations as bim
st = synthetic.make_stack(30, 2048, 2048, seed=100)
proj = sp.time_point_surface_projection(st[None], "TCZYX", 0, airyscan=False)
os.environ["TIP_WS_DEBUG"] = "1"
lab = bim.watershed_segmentation(proj[0], 0.03, 3, 3)
print(lab.max())
