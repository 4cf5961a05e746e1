"""Seeded synthetic confocal stacks.

Voronoi "membrane" tessellation on a curved surface + Poisson background,
two channels (ZO-1-like membranes, Atoh-like hair-cell fill), uint16.
Pure numpy so it runs identically under the oracle interpreter and the
product interpreter; used by the golden generator, the tests and bench.py.
"""
import numpy as np


def _two_nearest(sites, ny, nx, cell=32):
    """Distances to the nearest and 2nd-nearest site + index of the nearest, per pixel.

    Bucketed brute force: sites are binned on a `cell`-pixel grid and each
    pixel block only looks at sites of the 5x5 surrounding buckets (site
    density is 1/900 px^-2, so 160 px of context is ample).
    """
    d1 = np.empty((ny, nx), np.float64)
    d2 = np.empty((ny, nx), np.float64)
    i1 = np.empty((ny, nx), np.int64)
    by = (sites[:, 0] // cell).astype(np.int64)
    bx = (sites[:, 1] // cell).astype(np.int64)
    nby, nbx = -(-ny // cell), -(-nx // cell)
    order = np.lexsort((bx, by))
    sites_s, by_s, bx_s = sites[order], by[order], bx[order]
    key = by_s * nbx + bx_s
    starts = np.searchsorted(key, np.arange(nby * nbx + 1))
    R = 3
    for j in range(nby):
        y0, y1 = j * cell, min(ny, (j + 1) * cell)
        for i in range(nbx):
            x0, x1 = i * cell, min(nx, (i + 1) * cell)
            idx = []
            for jj in range(max(0, j - R), min(nby, j + R + 1)):
                a = starts[jj * nbx + max(0, i - R)]
                b = starts[jj * nbx + min(nbx, i + R + 1)]
                idx.append(np.arange(a, b))
            idx = np.concatenate(idx)
            if idx.size < 2:
                idx = np.arange(sites_s.shape[0])
            s = sites_s[idx]
            yy = np.arange(y0, y1, dtype=np.float64)[:, None, None]
            xx = np.arange(x0, x1, dtype=np.float64)[None, :, None]
            dist = np.sqrt((yy - s[None, None, :, 0]) ** 2 + (xx - s[None, None, :, 1]) ** 2)
            part = np.argpartition(dist, 1, axis=2)[:, :, :2]
            dd = np.take_along_axis(dist, part, axis=2)
            swap = dd[:, :, 0] > dd[:, :, 1]
            a0 = np.where(swap, dd[:, :, 1], dd[:, :, 0])
            a1 = np.where(swap, dd[:, :, 0], dd[:, :, 1])
            n0 = np.where(swap, part[:, :, 1], part[:, :, 0])
            d1[y0:y1, x0:x1] = a0
            d2[y0:y1, x0:x1] = a1
            i1[y0:y1, x0:x1] = order[idx[n0]]
    return d1, d2, i1


def make_sites(ny, nx, seed=0):
    rng = np.random.default_rng(seed)
    n = max(4, (ny * nx) // 900)
    sites = np.stack([rng.uniform(0, ny, n), rng.uniform(0, nx, n)], axis=1)
    is_hc = rng.uniform(size=n) < 0.3
    return sites, is_hc


def make_stack(nz, ny, nx, seed=0, sites=None, is_hc=None, offset=0, channels=2):
    """Returns uint16 (C, Z, Y, X). `offset` adds a constant (use 10000 for airyscan-style data)."""
    rng = np.random.default_rng(seed)
    if sites is None:
        sites, is_hc = make_sites(ny, nx, seed)
        rng = np.random.default_rng(seed + 7919)
    d1, d2, i1 = _two_nearest(sites, ny, nx)
    membrane = np.exp(-((d2 - d1) ** 2) / 4.0)
    hc = is_hc[i1].astype(np.float64)
    y = np.arange(ny, dtype=np.float64)[:, None]
    x = np.arange(nx, dtype=np.float64)[None, :]
    z0 = nz / 2.0 + (nz / 4.0) * np.sin(np.pi * y / ny) * np.cos(np.pi * x / nx)
    out = np.empty((channels, nz, ny, nx), np.uint16)
    for z in range(nz):
        profile = np.exp(-0.5 * ((z - z0) / 1.5) ** 2)
        c0 = 3000.0 * profile * membrane + rng.poisson(100, (ny, nx)) + offset
        out[0, z] = np.clip(c0, 0, 65535).astype(np.uint16)
        if channels > 1:
            c1 = 1500.0 * profile * hc + rng.poisson(100, (ny, nx)) + offset
            out[1, z] = np.clip(c1, 0, 65535).astype(np.uint16)
        for c in range(2, channels):
            out[c, z] = np.clip(rng.poisson(100, (ny, nx)) + offset, 0, 65535).astype(np.uint16)
    return out


def make_movie_sites(ny, nx, frames, seed=0):
    """Random-walking sites (sigma 1 px/frame) + global drift (0.5, -0.3) px/frame."""
    rng = np.random.default_rng(seed)
    sites, is_hc = make_sites(ny, nx, seed)
    out = []
    for t in range(frames):
        out.append(sites.copy())
        sites = sites + rng.normal(0, 1.0, sites.shape) + np.array([0.5, -0.3])
    return out, is_hc
