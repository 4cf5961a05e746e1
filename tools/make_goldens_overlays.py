#!/opt/conda/bin/python3.9
"""Goldens for SURVEY 8(f) rank 4: the overlay images (`draw_cell_types`, `draw_neighbors_connections`, `draw_cell_tracking`,
`draw_all_cell_tracking`, `draw_marking_points`, `draw_events`, ti.py:584-607, 2585-2645), `detect_edge_cells` (ti.py:609-612) and the
event detection of `find_events_iterator` (ti.py:636-789), from the REFERENCE's own methods on a small synthetic movie.

    /opt/conda/bin/python3.9 tools/make_goldens_overlays.py     -> tests/golden/overlays.npz

The movie: 5 frames of a drifting Voronoi tessellation segmented with skimage's watershed and tracked with the reference's own
`track_cells_iterator`; then cells are made to vanish (delamination), to appear next to a neighbour inside the neighbour's old
footprint (division) and to change type (differentiation) by editing the label maps / tables the detection reads.  `add_event` is
replaced by a recorder (its bookkeeping -- `find_event_frame`, label fixing -- is per-cell table logic outside the array path), so the
golden pins WHICH events the detection finds, with which frames and ids.  Only data is written."""
import os
import sys
import tempfile
import types
import warnings

warnings.filterwarnings("ignore")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_stub("aicsimageio", AICSImage=object)
_stub("aicsimageio.readers", czi_reader=None, bioformats_reader=None)
_stub("aicsimageio.writers", ome_tiff_writer=None)
_stub("trackpy")
sys.path.insert(0, os.path.join(REF, "tissue_analyzing_tool"))

import numpy as np  # noqa: E402
import scipy.ndimage as ndi  # noqa: E402
import skimage  # noqa: E402
import skimage.segmentation  # noqa: E402
import tissue_info as ti  # noqa: E402  (reference)
from tissue_image_processing_amd import synthetic  # noqa: E402

if not hasattr(np, "bool"):          # ti.py:155 uses the alias numpy 1.24 removed
    np.bool = bool


def cells_table(t, frame):
    ci = t.get_cells_info(frame)
    cols = {}
    for k in ["area", "perimeter", "label", "cx", "cy", "n_neighbors", "valid", "type", "empty_cell"]:
        cols[k] = np.asarray(ci[k].to_numpy(), dtype=np.float64)
    maxn = max([len(s) for s in ci.neighbors] + [1])
    nb = np.zeros((ci.shape[0], maxn), np.int64)
    for i, s in enumerate(ci.neighbors):
        ss = sorted(int(v) for v in s)
        nb[i, :len(ss)] = ss
    cols["neighbors"] = nb
    return cols


def build(tmp, tag, ny, nx, frames, labs, vanish_pick, late_pick):
    """a tracked reference Tissue with one vanishing cell and one late hair cell (picked by position in the list of valid interior ids)"""
    t = ti.Tissue(frames, os.path.join(tmp, "movie_" + tag), ["zo", "atoh"], load_to_memory=True)
    for f in range(frames):
        t.labels_list[f] = labs[f].copy()
    t.drifts[:] = 0
    for f in range(frames):
        t.set_labels(f + 1, labs[f].copy(), reset_data=False)
        t.calculate_frame_cellinfo(f + 1)
        t.cell_info_list[f] = t.cells_info.copy()
    for _ in t.track_cells_iterator(1, frames):
        pass
    t.type_names = ["HC"]
    rng = np.random.default_rng(3)
    # types by track id: a third of the ids are HC (bit 0); one id turns HC at frame 4 only (a differentiation)
    ci1 = t.get_cells_info(1)
    ids = np.unique(ci1.label.to_numpy())
    hc_ids = set(int(i) for i in ids[rng.random(ids.size) < 0.33])
    edge1 = set(int(v) for v in ci1.label[ti.Tissue.detect_edge_cells(t.get_labels(1))])
    valid_ids = [int(r.label) for _, r in ci1.iterrows() if r.valid == 1 and int(r.label) not in edge1]
    late = [[i for i in valid_ids if i not in hc_ids][late_pick]]
    vanish = [[i for i in valid_ids if i not in late][vanish_pick]]
    out = {"labels": np.stack(labs), "vanish": np.asarray(vanish), "late": np.asarray(late)}
    for f in range(frames):
        ci = t.get_cells_info(f + 1)
        lab = t.get_labels(f + 1)
        if f >= 3:
            # the vanishing cell: its row loses validity and its pixels join the lines
            for v in vanish:
                rows = ci.index[ci.label == v]
                for r in rows:
                    lab[lab == r + 1] = 0
                    ci.loc[r, "valid"] = 0
        typ = np.zeros(ci.shape[0], np.uint8)
        for r, idv in enumerate(ci.label.to_numpy()):
            if int(idv) in hc_ids or (f >= 3 and int(idv) in late):
                typ[r] = 1
        ci["type"] = typ
        t.cell_info_list[f] = ci
        t.labels_list[f] = lab
        lut = np.concatenate([[ti.INVALID_TYPE_INDEX], np.where(ci.valid.to_numpy() == 1, typ, ti.INVALID_TYPE_INDEX)]).astype(np.uint8)
        cell_types = lut[np.clip(lab, 0, ci.shape[0])]
        t.set_cell_types(f + 1, cell_types)
        out["labels_final_%d" % f] = lab.copy()
        out["cell_types_%d" % f] = cell_types.copy()
        for k, v in cells_table(t, f + 1).items():
            out["ci%d_%s" % (f, k)] = v
    return t, out, valid_ids


REC = []


def recorder(self, event_type, start_frame, end_frame, start_pos=None, end_pos=None, second_end_pos=None, start_cell_id=None,
             daughter_cell_id=None, source="manual"):
    REC.append((event_type, int(start_frame), int(end_frame), -1 if start_cell_id is None else int(start_cell_id),
                -1 if daughter_cell_id is None else int(daughter_cell_id), source))
    return 0


def make(name, ny, nx, picks):
    tmp = tempfile.mkdtemp(prefix="tipgold_ov_")
    frames = 5
    sites_t, is_hc = synthetic.make_movie_sites(ny, nx, frames, seed=21)
    labs = []
    for f in range(frames):
        d1, d2, i1 = synthetic._two_nearest(sites_t[f], ny, nx)
        membrane = np.exp(-((d2 - d1) ** 2) / 4.0)
        labs.append(skimage.segmentation.watershed(ndi.gaussian_filter(membrane, 1.5), watershed_line=True).astype(np.int32))
    # which cell vanishes / differentiates: the first picks for which the reference's detection reports a delamination AND a
    # differentiation (its neighbour test is strict: every neighbour label, read as a row index, must be a valid cell seen in both frames)
    real_add_event = ti.Tissue.add_event
    chosen = picks
    nvalid = len(build(tmp, "probe", ny, nx, frames, labs, 0, 0)[2])
    print("valid interior ids:", nvalid)
    for lp in (range(0, min(60, nvalid - 2)) if chosen is None else ()):
        for vp in (10, 3):
            ti.Tissue.add_event = real_add_event
            t, out, valid_ids = build(tmp, "s%d_%d" % (vp, lp), ny, nx, frames, labs, vp, lp)
            ti.Tissue.add_event = recorder
            del REC[:]
            t.events = ti.make_df(0, ti.EVENTS_INFO_SPEC)
            for _ in t.find_events_iterator(1, frames, differentiation_type_name="HC"):
                pass
            kinds = set(r[0] for r in REC)
            if "delamination" in kinds and "differentiation" in kinds:
                chosen = (vp, lp)
                break
        if chosen:
            break
    print("picks:", chosen, "events with them:", REC)
    ti.Tissue.add_event = real_add_event
    t, out, valid_ids = build(tmp, "final", ny, nx, frames, labs, chosen[0], chosen[1])
    # ---- overlays ------------------------------------------------------------------------------------------------------------------
    for f in (1, 4):
        out["draw_cell_types_%d" % f] = np.asarray(t.draw_cell_types(f, "HC"), dtype=np.float64)
        out["draw_neighbors_%d" % f] = np.asarray(t.draw_neighbors_connections(f), dtype=np.float64)
        out["draw_all_tracking_%d" % f] = np.asarray(t.draw_all_cell_tracking(f), dtype=np.float64)
        out["tracking_labels_%d" % f] = np.asarray(t.get_trackking_labels(f))
        out["edge_cells_%d" % f] = np.asarray(ti.Tissue.detect_edge_cells(t.get_labels(f)))
    some = valid_ids[5]
    out["track_one_id"] = np.asarray(some)
    out["draw_cell_tracking_2"] = np.asarray(t.draw_cell_tracking(2, some, radius=6), dtype=np.float64)
    out["draw_cell_tracking_missing"] = np.asarray(t.draw_cell_tracking(2, 10 ** 6), dtype=np.float64)
    t.shape_fitting_points = [(20.5, 30.25), (100, 5), (nx - 7.1, ny - 13.8), (3, 2)]
    out["marking_points"] = np.asarray(t.shape_fitting_points, dtype=np.float64)
    out["draw_marking_points"] = np.asarray(t.draw_marking_points(1, radius=4), dtype=np.float64)
    # events drawn from a hand-made table (two overlapping disks: the later row wins; a division paints its daughter too)
    ev = []
    for k, (typ_, cid, did, sf, ef) in enumerate([("delamination", valid_ids[1], 0, 1, 3), ("division", valid_ids[2], valid_ids[6], 2, 4),
                                                  ("differentiation", valid_ids[1], 0, 2, 2), ("ablation", 10 ** 6, 0, 1, 5),
                                                  ("division", valid_ids[7], 10 ** 6, 1, 5)]):
        row = dict(ti.EVENTS_INFO_SPEC)
        row.update(type=typ_, start_frame=sf, end_frame=ef, cell_id=cid, daughter_id=did, source="manual")
        ev.append(row)
    import pandas as pd
    t.events = pd.DataFrame(ev)
    out["events_type"] = np.asarray([e["type"] for e in ev])
    for k in ("start_frame", "end_frame", "cell_id", "daughter_id"):
        out["events_" + k] = np.asarray([e[k] for e in ev], dtype=np.int64)
    for f in (2, 3):
        out["draw_events_%d" % f] = np.asarray(t.draw_events(f, radius=5), dtype=np.float64)
    # ---- event detection with add_event recorded --------------------------------------------------------------------------------
    ti.Tissue.add_event = recorder
    del REC[:]
    rec = REC
    t.events = ti.make_df(0, ti.EVENTS_INFO_SPEC)
    yielded = [int(f) for f in t.find_events_iterator(1, frames, differentiation_type_name="HC")]
    out["found_frames"] = np.asarray(yielded, dtype=np.int64)
    out["found_type"] = np.asarray([r[0] for r in rec]) if rec else np.zeros((0,), "U1")
    out["found_rows"] = np.asarray([[r[1], r[2], r[3], r[4]] for r in rec], dtype=np.int64).reshape(-1, 4)
    # ... and with the reference's own add_event (positions from the tables, significant_frame from find_event_frame): the table it builds
    ti.Tissue.add_event = real_add_event
    t.events = ti.make_df(0, ti.EVENTS_INFO_SPEC)
    for _ in t.find_events_iterator(1, frames, differentiation_type_name="HC"):
        pass
    ev = t.events
    out["table_type"] = np.asarray([str(v) for v in ev["type"]]) if ev.shape[0] else np.zeros((0,), "U1")
    out["table_source"] = np.asarray([str(v) for v in ev["source"]]) if ev.shape[0] else np.zeros((0,), "U1")
    for k in ("start_frame", "end_frame", "start_pos_x", "start_pos_y", "end_pos_x", "end_pos_y", "daughter_pos_x", "daughter_pos_y", "cell_id",
              "daughter_id", "significant_frame"):
        out["table_" + k] = np.asarray(ev[k].to_numpy(), dtype=np.float64) if ev.shape[0] else np.zeros((0,))
    print("events table rows:", ev.shape[0])
    print("events found:", rec, "frames yielded:", yielded)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), versions=np.array([np.__version__, skimage.__version__]), **out)
    print("wrote", name, {k: v.shape for k, v in out.items() if k.startswith("draw_")})


if __name__ == "__main__":
    make("overlays", 224, 288, None)            # delaminations and a differentiation (picks searched for)
    make("overlays_small", 112, 136, (3, 0))    # a division
