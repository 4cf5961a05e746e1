#!/opt/conda/bin/python3.9
"""Golden for value ties between NON-ADJACENT non-marker pixels (diagonal / distance-2 pairs, no 4-adjacent tie anywhere):
skimage's (value, age) heap and a (value, raster index) order can disagree on such images through a "pulled" pixel (a lower
pixel enclosed by lines that pops right after a neighbour of exactly the tied value).  Searches small random landscapes for
cases where the two orders give different label maps and stores the images with skimage's labels (data only).

    /opt/conda/bin/python3.9 tools/make_goldens_ties.py     -> tests/golden/watershed_diag_ties.npz
"""
import heapq
import os

import numpy as np
import skimage
import skimage.segmentation
import scipy.ndimage as ndi
from skimage.morphology import local_minima

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def flood_raster_ties(img):
    """the serial flood with (value, raster index) keys instead of (value, age); markers = label(local_minima)"""
    Y, X = img.shape
    lab = ndi.label(local_minima(img, connectivity=1))[0].astype(np.int32)
    out = lab.copy()
    inq = lab > 0
    h = [(img[y, x], y * X + x) for y, x in zip(*np.nonzero(lab))]
    heapq.heapify(h)
    while h:
        v, i = heapq.heappop(h)
        y, x = divmod(i, X)
        if lab[y, x] == 0:
            labs = set()
            for dy, dx in ((-1, 0), (0, -1), (0, 1), (1, 0)):
                yy, xx = y + dy, x + dx
                if 0 <= yy < Y and 0 <= xx < X and out[yy, xx] > 0:
                    labs.add(out[yy, xx])
            if len(labs) != 1:
                continue                      # a line
            out[y, x] = labs.pop()
        for dy, dx in ((-1, 0), (0, -1), (0, 1), (1, 0)):
            yy, xx = y + dy, x + dx
            if 0 <= yy < Y and 0 <= xx < X and not inq[yy, xx]:
                inq[yy, xx] = True
                heapq.heappush(h, (img[yy, xx], yy * X + xx))
    return out


def adjacent_ties(img, lab0):
    nm = lab0 == 0
    t = (img[:, 1:] == img[:, :-1]) & nm[:, 1:] & nm[:, :-1]
    u = (img[1:, :] == img[:-1, :]) & nm[1:, :] & nm[:-1, :]
    return bool(t.any() or u.any())


def main():
    rng = np.random.default_rng(5)
    found = []
    tries = 0
    while len(found) < 6 and tries < 400000:
        tries += 1
        Y, X = int(rng.integers(5, 10)), int(rng.integers(5, 10))
        img = rng.permutation(Y * X).astype(np.float64).reshape(Y, X) * 0.37 + 0.11
        # tie a few diagonal / distance-2 pairs
        for _ in range(int(rng.integers(1, 5))):
            y, x = int(rng.integers(0, Y)), int(rng.integers(0, X))
            dy, dx = [(1, 1), (1, -1), (2, 0), (0, 2)][int(rng.integers(0, 4))]
            if 0 <= y + dy < Y and 0 <= x + dx < X:
                img[y + dy, x + dx] = img[y, x]
        lab0 = ndi.label(local_minima(img, connectivity=1))[0]
        if adjacent_ties(img, lab0):
            continue
        ref = skimage.segmentation.watershed(img, watershed_line=True)
        alt = flood_raster_ties(img)
        if not np.array_equal(ref, alt):
            found.append((img, ref.astype(np.int32)))
    out = {}
    for k, (img, ref) in enumerate(found):
        out["img%d" % k], out["labels%d" % k] = img, ref
    np.savez_compressed(os.path.join(OUT, "watershed_diag_ties.npz"), versions=np.array([np.__version__, skimage.__version__]), **out)
    print("tries", tries, "cases", len(found), [f[0].shape for f in found])


if __name__ == "__main__":
    main()
