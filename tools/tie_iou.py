import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from oracle import oracle as orc
from tissue_image_processing_amd import synthetic, surface_projection as sp, _segmentation as seg
from test_gpu_segmentation import label_iou
st = synthetic.make_stack(10, 512, 512, seed=44)
proj,_ = sp.time_point_surface_projection(st[None], "TCZYX", 0, airyscan=False, z_map=True)
zo = proj[0]
img16 = np.round(zo / zo.max() * 65535).astype(np.uint16)     # save_tiff's uint16 normalisation (bim.py:183-188)
lab, flags = seg.watershed_segmentation(img16, 0.03, 3, 3, return_flags=True)
# oracle on the same integer image: threshold, integer blur (trunc per axis), serial FIFO flood
s = img16.copy(); thr = orc.threshold_local_generic_max(s.astype(np.float64), 0.03, 3); s[s < thr] = 0
cur = s.astype(np.float64)
for ax in range(2):
    sg=[0,0]; sg[ax]=3
    cur = np.trunc(orc.blur_image(cur, tuple(sg)))
ref = orc.watershed(cur)
print("uint16 frame: labels gpu %d ref %d, mismatching pixels %.3f%%, IoU %.4f, flags %d" % (lab.max(), ref.max(), 100*float((lab!=ref).mean()), label_iou(lab, ref), flags))
