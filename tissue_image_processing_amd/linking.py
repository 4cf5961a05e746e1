"""Frame-to-frame feature linking for T1 (`Tissue.track_cells_iterator_with_trackpy`, ti.py:1881-1938).

PARITY UNPINNED.  Upstream delegates this step to the third-party package trackpy
(`trackpy.link_df_iter(search_range=100, adaptive_stop=10, memory=3, neighbor_strategy='BTree', dist_func=...)`, version
not pinned by the reference, not installed here, no reference test or golden covers it).  This module restates
trackpy's published linking model for exactly the parameters of that call site:

* distance `sqrt(dy^2 + dx^2 + 0.5 (sqrt(a1) - sqrt(a2))^2)` (`tracking_dist_func`, ti.py:1935-1938) -- which is the
  Euclidean distance between the points `(cy, cx, sqrt(area / 2))`, so a KD-tree over that embedding finds the candidates;
* a track may continue to any feature of the next frame within `search_range`; tracks and features that can reach each
  other through such candidate links form a *subnet*, and inside a subnet the set of links minimising
  `sum(d^2)` wins, an unlinked track or feature costing `search_range^2` (solved exactly here with the Hungarian method,
  `scipy.optimize.linear_sum_assignment`, on the subnet's augmented cost matrix);
* subnets larger than `MAX_SUBNET` (trackpy's default 30) are not solved at the full range: the range is multiplied by
  `adaptive_step` (0.95) for that subnet -- which splits it -- until every piece fits or the range falls below
  `adaptive_stop`, where linking fails (trackpy raises SubnetOversizeException; here `SubnetOversizeError`);
* `memory`: a track that finds no feature stays linkable at its last position for that many further frames;
* new tracks get consecutive particle numbers in feature order.

The tables involved are a few thousand rows per frame: this is host work (numpy / scipy), not a GPU kernel; the dense
array steps of tracking (drift, label lookup) are the HIP paths in `_registration.py` and `tip_lookup_max3_i32_dev`.
"""
import numpy as np

MAX_SUBNET = 30


class SubnetOversizeError(RuntimeError):
    pass


def embed(cy, cx, area):
    """(N, 3) points whose Euclidean distance equals the reference's tracking_dist_func."""
    return np.stack([np.atleast_1d(np.asarray(cy, np.float64)), np.atleast_1d(np.asarray(cx, np.float64)),
                     np.sqrt(np.atleast_1d(np.asarray(area, np.float64)) / 2.0)], axis=1)


def _solve_subnet(src, dst, es, ed, ev, null_range):
    """Minimise sum(d^2) over one subnet.  src/dst: the subnet's track / feature indices; (es, ed, ev): its candidate
    links (track, feature, distance).  An unlinked track or feature costs null_range^2.  Returns a list of (s, d)."""
    from scipy.optimize import linear_sum_assignment
    ns, nd = len(src), len(dst)
    big = 1e18
    null = float(null_range) ** 2
    # augmented square problem: rows = tracks + one "stays unborn" row per feature, cols = features + one "lost" col per track
    cost = np.full((ns + nd, nd + ns), big)
    cost[np.searchsorted(src, es), np.searchsorted(dst, ed)] = ev * ev
    cost[np.arange(ns), nd + np.arange(ns)] = null          # track i lost
    cost[ns + np.arange(nd), np.arange(nd)] = null          # feature j starts a new track
    cost[ns:, nd:] = 0.0
    rows, cols = linear_sum_assignment(cost)
    return [(src[r], dst[c]) for r, c in zip(rows, cols) if r < ns and c < nd and cost[r, c] < big]


def link_candidates(n_src, n_dst, es, ed, ev, search_range, adaptive_stop, adaptive_step):
    """All subnets of the candidate graph (es[k], ed[k]) with distances ev[k] <= search_range.  Subnets that fit
    (<= MAX_SUBNET tracks and features) are solved at the current range; the others get the range multiplied by
    adaptive_step -- links longer than that are dropped, which splits them -- and are looked at again."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    links = []
    rng = float(search_range)
    es, ed, ev = np.asarray(es, np.int64), np.asarray(ed, np.int64), np.asarray(ev, np.float64)
    while es.size:
        g = coo_matrix((np.ones(es.size, np.int8), (es, n_src + ed)), shape=(n_src + n_dst, n_src + n_dst))
        _, comp = connected_components(g, directed=False)
        ce = comp[es]                                    # component of every edge
        order = np.argsort(ce, kind="stable")
        es, ed, ev, ce = es[order], ed[order], ev[order], ce[order]
        starts = np.flatnonzero(np.r_[True, ce[1:] != ce[:-1]])
        ends = np.r_[starts[1:], ce.size]
        keep = np.zeros(es.size, bool)
        for a, b in zip(starts, ends):
            src = np.unique(es[a:b])
            dst = np.unique(ed[a:b])
            if src.size <= MAX_SUBNET and dst.size <= MAX_SUBNET:
                links.extend(_solve_subnet(src, dst, es[a:b], ed[a:b], ev[a:b], rng))
            else:
                keep[a:b] = True
        if not keep.any():
            break
        if adaptive_stop is None:
            raise SubnetOversizeError("a subnet exceeds %d particles at range %g" % (MAX_SUBNET, rng))
        rng *= adaptive_step
        if rng < adaptive_stop:
            raise SubnetOversizeError("a subnet still exceeds %d particles at the adaptive_stop range %g"
                                      % (MAX_SUBNET, adaptive_stop))
        keep &= ev <= rng
        es, ed, ev = es[keep], ed[keep], ev[keep]
    return links


class FrameLinker(object):
    """Sequential linker: feed the embedded features of consecutive frames to `link`, get particle numbers back."""

    def __init__(self, search_range=100.0, adaptive_stop=10.0, adaptive_step=0.95, memory=3):
        self.search_range = float(search_range)
        self.adaptive_stop = adaptive_stop
        self.adaptive_step = float(adaptive_step)
        self.memory = int(memory)
        self.next_particle = 0
        self.track_pos = np.zeros((0, 3))      # last known embedded position of every linkable track
        self.track_id = np.zeros((0,), np.int64)
        self.track_age = np.zeros((0,), np.int64)  # frames since the track was last seen (0 = seen in the previous frame)

    def link(self, points):
        """points: (N, 3) embedded features of the next frame.  Returns (N,) int64 particle numbers."""
        from scipy.spatial import cKDTree
        points = np.asarray(points, np.float64).reshape(-1, 3)
        n = points.shape[0]
        out_ids = np.full((n,), -1, np.int64)
        links = []
        if n and self.track_id.size:
            pairs = cKDTree(self.track_pos).sparse_distance_matrix(cKDTree(points), self.search_range, output_type="coo_matrix")
            links = link_candidates(self.track_id.size, n, pairs.row, pairs.col, pairs.data, self.search_range,
                                    self.adaptive_stop, self.adaptive_step)
        linked_src = np.zeros((self.track_id.size,), bool)
        for s, d in links:
            out_ids[d] = self.track_id[s]
            linked_src[s] = True
        fresh = np.flatnonzero(out_ids < 0)
        out_ids[fresh] = self.next_particle + np.arange(fresh.size)
        self.next_particle += fresh.size
        # tracks that found nothing stay linkable at their last position for `memory` more frames
        keep = ~linked_src & (self.track_age < self.memory)
        self.track_pos = np.concatenate([points, self.track_pos[keep]], axis=0)
        self.track_id = np.concatenate([out_ids, self.track_id[keep]])
        self.track_age = np.concatenate([np.zeros((n,), np.int64), self.track_age[keep] + 1])
        return out_ids
