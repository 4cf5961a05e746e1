"""Prototype (CPU, pure python) of the parallel-rounds exact watershed used to design the HIP kernel.
Not product code; kept for design traceability."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc

U, LINE = 0, -1

def ws_rounds(img, markers, cert_depth=1, verbose=False):
    ny, nx = img.shape
    n = ny * nx
    v = img.ravel().astype(np.float64)
    lab = markers.ravel().astype(np.int64).copy()
    Tv = np.where(lab > 0, v, np.inf)          # pop time (value part)
    Ti = np.where(lab > 0, np.arange(n), 0)    # pop time (index part)
    def nbrs(i):
        y, x = divmod(i, nx)
        out = []
        if y > 0: out.append(i - nx)
        if x > 0: out.append(i - 1)
        if x < nx - 1: out.append(i + 1)
        if y < ny - 1: out.append(i + nx)
        return out
    def key(i): return (v[i], i)
    def T(i): return (Tv[i], Ti[i])
    def cert(q, t, asker):
        for m in nbrs(q):
            if m == asker: continue
            if lab[m] == LINE: continue
            if lab[m] > 0:
                if T(m) < t: return False
            else:
                if key(m) < t: return False
        return True
    def decide(p, force):
        kp = key(p)
        S = set(); wait = False
        pull = None
        unb = []
        for q in nbrs(p):
            if lab[q] == LINE: continue
            if lab[q] > 0:
                if T(q) < kp: S.add(lab[q])
                else:
                    if pull is None or T(q) < pull[0]: pull = (T(q), lab[q])
            else:
                unb.append(q)
                if key(q) < kp and not force:
                    if not cert(q, kp, p): wait = True
        if wait: return None
        if S:
            if len(S) == 1: return (next(iter(S)), kp)
            return (LINE, kp)
        if pull is None: return None
        if not force:
            for q in unb:
                if key(q) > pull[0]: continue
                if not cert(q, pull[0], p): return None
        return (pull[1], pull[0])
    rounds = fallbacks = 0
    while True:
        und = np.nonzero(lab == U)[0]
        if und.size == 0: break
        dec = []
        for p in und:
            r = decide(p, False)
            if r is not None: dec.append((p, r))
        rounds += 1
        if not dec:
            # fallback: global min pop-time estimate among heap pixels
            best = None
            for p in und:
                ts = [T(q) for q in nbrs(p) if lab[q] > 0]
                if not ts: continue
                pt = max(key(p), min(ts))
                if best is None or pt < best[0]: best = (pt, p)
            if best is None: break   # unreachable pixels (enclosed by lines): stay 0
            p = best[1]
            r = decide(p, True)
            assert r is not None
            dec = [(p, r)]
            fallbacks += 1
        for p, (l, t) in dec:
            lab[p] = l; Tv[p], Ti[p] = t
    out = lab.reshape(ny, nx).copy()
    out[out < 0] = 0
    return out.astype(np.int32), rounds, fallbacks

if __name__ == "__main__":
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "watershed.npz"))
    for case, key in [("ii", "ii_img"), ("iii", "iii_img"), ("iv", "iv_img"), ("v", "v_img"), ("vi", "vi_boundary")]:
        img = g[key]
        markers, _ = orc.label4(orc.local_minima(img).astype(np.int32), 0)
        out, rounds, fb = ws_rounds(img, markers)
        ref = g[case + "_labels"]
        print(case, img.shape, "rounds", rounds, "fallbacks", fb, "mismatch", int((out != ref).sum()), "of", ref.size)


def ws_binary_bfs(img, markers):
    """Mode B: two-valued image, generation-synchronous BFS with (gen, idx) tie-break and tentative-label rule."""
    ny, nx = img.shape
    lab = markers.astype(np.int64).copy()
    gen = np.where(lab > 0, 0, -1)
    g = 0
    def nb(y, x):
        if y > 0: yield y - 1, x
        if x > 0: yield y, x - 1
        if x < nx - 1: yield y, x + 1
        if y < ny - 1: yield y + 1, x
    while True:
        g += 1
        tent = {}
        for y in range(ny):
            for x in range(nx):
                if lab[y, x] != 0: continue
                S = set(lab[a, b] for a, b in nb(y, x) if lab[a, b] > 0 and gen[a, b] < g)
                if not S: continue
                tent[(y, x)] = next(iter(S)) if len(S) == 1 else LINE
        if not tent: break
        fate = {k: (LINE if t == LINE else None) for k, t in tent.items()}
        changed = True
        while changed:
            changed = False
            for (y, x), t in tent.items():
                if fate[(y, x)] is not None: continue
                res = t
                pending = False
                for a, b in ((y - 1, x), (y, x - 1)):
                    if (a, b) in tent and tent[(a, b)] != t and tent[(a, b)] != LINE:
                        f = fate[(a, b)]
                        if f is None: pending = True
                        elif f != LINE: res = LINE
                if res == LINE: fate[(y, x)] = LINE; changed = True
                elif not pending: fate[(y, x)] = t; changed = True
        for (y, x), f in fate.items():
            lab[y, x] = f
            gen[y, x] = g
    out = lab.copy(); out[out < 0] = 0
    return out.astype(np.int32), g


def iou(a, b):
    """mean IoU of reference labels against best-overlapping test labels"""
    ious = []
    for l in np.unique(b):
        if l == 0: continue
        m = b == l
        cand = np.bincount(a[m])
        cand[0] = 0
        if cand.sum() == 0: ious.append(0.0); continue
        k = cand.argmax()
        ious.append((m & (a == k)).sum() / float((m | (a == k)).sum()))
    return float(np.mean(ious))


if __name__ == "__main__":
    for fn, key, lk in [("watershed", "vi_boundary", "vi_labels"), ("unet_tail", "boundary", "labels")]:
        g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", fn + ".npz"))
        img = g[key]
        markers, _ = orc.label4(orc.local_minima(img).astype(np.int32), 0)
        out, gens = ws_binary_bfs(img, markers)
        ref = g[lk]
        print(fn, "gens", gens, "mismatch", int((out != ref).sum()), "of", ref.size, "lines", int((out == 0).sum()), int((ref == 0).sum()), "IoU", iou(out, ref))
