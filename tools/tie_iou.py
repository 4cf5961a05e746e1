"""Label agreement of the HIP watershed with skimage on inputs whose flood order depends on ties:
golden v (quantised landscape), vi / unet_tail (two-valued boundary images, mode B) and a uint16-normalised
synthetic frame through watershed_segmentation (the GUI's classical path, gui.py:1841-1845).  GPU box."""
import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle import oracle as orc
from tissue_image_processing_amd import synthetic, surface_projection as sp, _segmentation as seg
from test_gpu_segmentation import label_iou


def report(name, lab, ref, flags):
    print("%-14s labels gpu %d ref %d, mismatching pixels %d = %.4f%%, IoU %.4f, flags %d" % (
        name, lab.max(), ref.max(), int((lab != ref).sum()), 100 * float((lab != ref).mean()), label_iou(lab, ref), flags))


g = np.load('/root/repo/tests/golden/watershed.npz')
for k in ("v", "vi"):
    img = g[k + "_img"] if k == "v" else g["vi_boundary"]
    lab, flags = seg.watershed(img, return_flags=True)
    report("golden " + k, lab, g[k + "_labels"], flags)
t = np.load('/root/repo/tests/golden/unet_tail.npz')
lab, flags = seg.watershed(t["boundary"], return_flags=True)
report("golden tail", lab, t["labels"], flags)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
st = synthetic.make_stack(10, N, N, seed=44)
proj, _ = sp.time_point_surface_projection(st[None], "TCZYX", 0, airyscan=False, z_map=True)
zo = proj[0]
img16 = np.round(zo / zo.max() * 65535).astype(np.uint16)     # save_tiff's uint16 normalisation (bim.py:183-188)
lab, flags = seg.watershed_segmentation(img16, 0.03, 3, 3, return_flags=True)
# oracle on the same integer image: threshold, integer blur (trunc per axis), serial (value, age) heap flood
s = img16.copy(); thr = orc.threshold_local_generic_max(s.astype(np.float64), 0.03, 3); s[s < thr] = 0
cur = s.astype(np.float64)
for ax in range(2):
    sg = [0, 0]; sg[ax] = 3
    cur = np.trunc(orc.blur_image(cur, tuple(sg)))
report("uint16 %d^2" % N, lab, orc.watershed(cur), flags)
