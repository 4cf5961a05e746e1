"""Drop-in for the array-heavy methods of the reference's tissue_info.Tissue (ti.py) on MI355X.

`TissueHipMixin` carries the hot methods with the reference's signatures; mixing it in FRONT of the reference's
own class (`class Tissue(TissueHipMixin, reference.Tissue)`) re-routes them to libtissue_hip.so while the
reference's state / cache / persistence code (ti.py:193-353, 3462-3823) stays the caller, unchanged.
`Tissue` below is a small self-contained in-memory host for the same methods (what the tests and bench use).

    calculate_frame_cellinfo(frame_number)                 ti.py:880-909   -> tip_regionprops_i32 + tip_neighbor_pairs_i32
    find_neighbors(frame_number, only_for_labels=None)     ti.py:1815-1842 -> tip_neighbor_pairs_i32
    update_labels(frame)                                   ti.py:2967-2975 -> tip_update_labels_i32
    calc_cell_types(...), update_cell_types_by_cells_info  ti.py:2338-2408 -> tip_regionprops_i32 + tip_label_order_stats_f64
    calc_neighbors_contact_matrix(frame)                   ti.py:4073-4094 -> tip_rankfilter2d (cross footprint)
    track_cells_iterator(...)                              ti.py:2037-2113 -> tip_rankfilter2d + host table logic
    get_trackking_labels(frame)                            ti.py:4021-4028 -> tip_lut_gather_i32
"""
import ctypes

import numpy as np
import pandas as pd

from . import _lib
from . import _segmentation as seg
from .basic_image_manipulations import blur_image

CELL_INFO_SPECS = {"area": 0, "perimeter": 0, "label": 0, "cx": 0, "cy": 0, "neighbors": set(), "n_neighbors": 0,
                   "valid": 0, "type": 0, "bounding_box_min_row": 0, "bounding_box_min_col": 0,
                   "bounding_box_max_row": 0, "bounding_box_max_col": 0, "empty_cell": 0}
INVALID_TYPE_INDEX = 255
# overlay colours (ti.py:68-77)
TRACK_COLOR = (0, 1, 0)
NEIGHBORS_COLOR = (1, 1, 1)
POS_COLOR = (1, 0, 1)
NEG_COLOR = (1, 1, 0)
MARKING_COLOR = (0.5, 0.5, 0.5)
EVENTS_COLOR = {"ablation": (1, 1, 0), "division": (0, 0, 1), "delamination": (1, 0, 0), "differentiation": (0, 1, 1),
                "promoted differentiation": (1, 1, 1)}
TRACKING_COLOR_CYCLE = [(1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1)]


def make_df(number_of_lines, specs):
    """ti.py:85-98."""
    if number_of_lines > 0:
        df = pd.DataFrame(index=np.arange(number_of_lines))
        for name, val in specs.items():
            df[name] = [set() for _ in range(number_of_lines)] if isinstance(val, set) else \
                [list() for _ in range(number_of_lines)] if isinstance(val, list) else val
    else:
        df = pd.DataFrame(columns=specs.keys())
    return df


def find_local_maxima(arr, window_size=7):
    """ti.py:141-144: blur sigma 7, maximum_filter(size=window) (default mode reflect), |blur - max| < 1e-6."""
    blurred = blur_image(np.asarray(arr, dtype=np.float64), 7)
    maxima = seg.maximum_filter(blurred, size=window_size)
    return np.abs(blurred - maxima) < 1e-6


def _type_bits(type):
    return np.asarray(type).astype(np.uint8)


def is_positive_for_type(type, type_index):
    """ti.py:146-176: is bit `type_index` of the cell-type byte set?  `type_index` may be a pair (types that must be set,
    types that must be clear).  Arrays in, boolean arrays out, with INVALID_TYPE_INDEX (255) cells never positive; a SCALAR
    type is only bit-tested, as upstream (ti.py:171-175 clears invalid cells for arrays only, so a scalar 255 reads as
    positive for every type -- kept, and pinned by tests/test_io_formats.py)."""
    if isinstance(type_index, tuple):
        must_have, must_lack = type_index
        verdict = np.ones(np.shape(type), dtype=bool)
        for t in must_have:
            verdict = verdict & np.asarray(is_positive_for_type(type, t), dtype=bool)
        for t in must_lack:
            verdict = verdict & ~np.asarray(is_positive_for_type(type, t), dtype=bool)
        return verdict if verdict.ndim else bool(verdict)
    if type_index < 0:
        return False
    bits = _type_bits(type)
    bit = np.uint8(1 << type_index)
    hit = (bits & bit) == bit
    if bits.ndim == 0:
        return np.bool_(hit)
    return hit & (bits != INVALID_TYPE_INDEX)


def change_type(current_type, type_index, is_positive):
    """ti.py:179-191: the type byte with bit `type_index` set or cleared; an invalid cell (255) starts from 0."""
    bits = _type_bits(current_type)
    bits = np.where(bits == INVALID_TYPE_INDEX, np.uint8(0), bits)
    bit = np.uint8(1 << type_index)
    out = (bits | bit) if is_positive else (bits & np.uint8(~bit & 0xFF))
    return out.astype(np.uint8)


def _link_two_frames(f, search_range, adaptive_stop, pos_columns, t_column, memory, neighbor_strategy, dist_func):
    """trackpy.link's call signature over linking.FrameLinker for a two-frame table (parity with trackpy unpinned, see
    linking.py): rows of the first frame get particles 0..n-1 in row order, rows of the second the particle of their
    partner or fresh numbers."""
    from .linking import FrameLinker, embed
    out = f.copy()
    frames = np.sort(f[t_column].unique())
    linker = FrameLinker(search_range=search_range, adaptive_stop=adaptive_stop, memory=memory)
    particle = np.zeros(len(f), np.int64)
    for t in frames:
        rows = np.flatnonzero(f[t_column].to_numpy() == t)
        feats = [f[c].to_numpy()[rows] for c in pos_columns]           # (cy, cx, area)
        particle[rows] = linker.link(embed(feats[0], feats[1], feats[2]))
    out["particle"] = particle
    return out


class TissueHipMixin(object):
    """Hot methods of Tissue; expects the host class to provide get_labels / get_cells_info / set_cells_info /
    get_cell_types / set_cell_types, max_cell_area, min_cell_area, type_names, drifts, valid_frames,
    number_of_frames, cells_number (all present in the reference's class, ti.py:218-253)."""

    # ---- C1 -------------------------------------------------------------------------------------------------
    def calculate_frame_cellinfo(self, frame_number):
        labels = self.get_labels(frame_number)
        if labels is None:
            return 0
        number_of_cells = int(np.max(labels))
        if number_of_cells == 0:
            return 0
        rp = seg.regionprops_arrays(labels, n=number_of_cells)      # SoA over labels 1..n, one device pass
        present = rp["area"] > 0                                      # labels that do not occur keep the table's zeros
        columns = {"label": rp["label"], "area": rp["area"], "perimeter": rp["perimeter"], "cx": rp["cx"], "cy": rp["cy"]}
        for j, edge in enumerate(("min_row", "min_col", "max_row", "max_col")):
            columns["bounding_box_" + edge] = rp["bbox"][:, j]
        cells_info = make_df(number_of_cells, CELL_INFO_SPECS)
        for name, values in columns.items():
            full = np.zeros(number_of_cells, dtype=np.float64 if name in ("perimeter", "cx", "cy") else np.int64)
            full[present] = np.asarray(values)[present]
            cells_info[name] = full
        # ti.py:903-907: a cell is valid when its area lies strictly between min/max_cell_area times the mean area
        areas = cells_info["area"].to_numpy()
        smallest, largest = self.min_cell_area * areas.mean(), self.max_cell_area * areas.mean()
        cells_info["valid"] = ((areas > smallest) & (areas < largest)).astype(int)
        self.set_cells_info(frame_number, cells_info)
        self.find_neighbors(frame_number, only_for_labels=np.flatnonzero(cells_info["valid"].to_numpy() == 1) + 1)
        return 0

    # ---- C2 -------------------------------------------------------------------------------------------------
    def find_neighbors(self, frame_number, only_for_labels=None):
        labels = self.get_labels(frame_number)
        if labels is None:
            return 0
        cells_info = self.get_cells_info(frame_number)
        pairs = seg.neighbor_pairs(labels)
        if only_for_labels is None:
            working_indices = cells_info.query("empty_cell == 0").index.to_numpy()
        else:
            working_indices = np.array(only_for_labels) - 1
        neighbors = cells_info["neighbors"].to_numpy()
        n_neighbors = cells_info["n_neighbors"].to_numpy().copy()
        by_hi = {}
        for hi, lo in pairs:
            by_hi.setdefault(int(hi), []).append(int(lo))
        for cell_index in working_indices:
            neighbors[cell_index] = set()
        for cell_index in working_indices:
            cell_label = int(cell_index) + 1
            nl = by_hi.get(cell_label)
            if nl:
                neighbors[cell_index] = neighbors[cell_index].union(nl)
                for nb in nl:
                    neighbors[nb - 1].add(cell_label)
                    n_neighbors[nb - 1] = len(neighbors[nb - 1])
        for cell_index in working_indices:
            n_neighbors[cell_index] = len(neighbors[cell_index])
        cells_info["neighbors"] = list(neighbors)
        cells_info["n_neighbors"] = n_neighbors
        return

    # ---- C4's caller: a drawn segmentation line splits a cell (ti.py:2878-2965) ------------------------------------
    def get_new_labels(self, frame, n_new_labels):
        """ti.py:2878-2897: label numbers for n new cells -- rows of deleted cells (empty_cell == 1) first, then fresh rows
        after the table's end (after the label maximum when there is no table)."""
        labels = self.get_labels(frame)
        if labels is None:
            return 0
        table = self.get_cells_info(frame)
        if table is None:
            return np.max(labels) + np.arange(1, n_new_labels + 1)
        free = table.index[table.empty_cell.to_numpy() == 1].to_numpy() + 1
        if free.size >= n_new_labels and free.size > 0:
            return free[:n_new_labels] if free.size > n_new_labels else free
        return np.hstack((free, table.shape[0] + np.arange(1, n_new_labels - free.size + 1))).astype(np.int64)

    def update_after_adding_segmentation_line(self, cell_label, frame):
        """ti.py:2900-2965: after the user has drawn a line of zeros through cell `cell_label`, re-label the connected pieces
        of the cell inside its bounding box (+2): the first piece keeps the label, the others get new ones; their table
        rows (regionprops of the box), neighbour sets and type map follow.  Connected components and per-piece reductions
        run on the device (label / regionprops on the box)."""
        labels = self.get_labels(frame)
        if labels is None:
            return None
        table = self.get_cells_info(frame)
        cell_types = self.get_cell_types(frame)
        if table is None:
            where = np.argwhere(labels == cell_label)
            r0, c0 = where.min(axis=0)
            r1, c1 = where.max(axis=0) + 1
        else:
            row = table.iloc[cell_label - 1]
            r0, c0, r1, c1 = (int(row["bounding_box_" + k]) for k in ("min_row", "min_col", "max_row", "max_col"))
        fr, fc = max(0, int(r0) - 2), max(0, int(c0) - 2)
        box = labels[fr:int(r1) + 2, fc:int(c1) + 2]                        # (a view: edits land in the frame's label map)
        pieces = seg.label((box != 0).astype(int), connectivity=1, background=0)
        piece_ids = np.unique(pieces[box == cell_label])
        if piece_ids.size == 1:
            print("New line did not split the cell")
            return 0
        new_labels = np.hstack((np.array([cell_label]), self.get_new_labels(frame, piece_ids.size - 1)))
        for piece, lab in zip(piece_ids, new_labels):
            box[pieces == piece] = lab
        if table is None:
            return None
        try:
            rp = seg.regionprops_arrays(np.ascontiguousarray(box))
            areas = table.area.to_numpy()
            smallest, largest = self.min_cell_area * np.mean(areas), self.max_cell_area * np.mean(areas)
            old_neighbors = list(table.neighbors[cell_label - 1].copy())
            old_type = table.type[cell_label - 1]
            for lab in sorted(int(v) for v in new_labels):
                if lab > rp["area"].size or rp["area"][lab - 1] == 0:
                    continue
                k = lab - 1
                table.loc[k] = pd.Series({
                    "area": rp["area"][k], "label": lab, "perimeter": rp["perimeter"][k],
                    "cx": rp["cx"][k] + fc, "cy": rp["cy"][k] + fr,
                    "bounding_box_min_row": rp["bbox"][k, 0] + fr, "bounding_box_min_col": rp["bbox"][k, 1] + fc,
                    "bounding_box_max_row": rp["bbox"][k, 2] + fr, "bounding_box_max_col": rp["bbox"][k, 3] + fc,
                    "valid": int(smallest < rp["area"][k] < largest), "empty_cell": 0, "neighbors": set(), "n_neighbors": 0,
                    "type": int(old_type)})
            self.find_neighbors(frame, only_for_labels=old_neighbors + list(new_labels))
            if cell_types is not None:
                for lab in new_labels:
                    cell_types[labels == lab] = old_type if table.valid[lab - 1] else INVALID_TYPE_INDEX
            return 0
        except IndexError:
            return 0

    # ---- C3 -------------------------------------------------------------------------------------------------
    def update_labels(self, frame):
        labels = self.get_labels(frame)
        lab32 = np.ascontiguousarray(labels, dtype=np.int32)
        _lib.check(_lib.lib().tip_update_labels_i32(_lib.ptr(lab32), lab32.shape[0], lab32.shape[1]))
        labels[...] = lab32
        self.last_action = []
        self._neighbors_labels = (0, 0)
        self.last_added_line = []
        self.update_cell_types_by_cells_info(frame)
        return 0

    # ---- whole-movie refreshes (ti.py:4230-4247) ---------------------------------------------------------------------------
    def update_bounding_box_for_all_cells(self):
        """ti.py:4230-4241: bounding boxes of every frame's table from its label map (one device pass per frame)."""
        for frame in range(1, self.number_of_frames + 1):
            labels, table = self.get_labels(frame), self.get_cells_info(frame)
            if table is None or labels is None:
                continue
            rp = seg.regionprops_arrays(labels)
            rows = np.flatnonzero(rp["area"] > 0)
            for j, edge in enumerate(("min_row", "min_col", "max_row", "max_col")):
                table.loc[rows, "bounding_box_" + edge] = rp["bbox"][rows, j]
            if hasattr(self, "save_cells_info"):
                self.save_cells_info()
        return 0

    def update_neighbors_for_all_cells(self):
        """ti.py:4243-4247."""
        for frame in range(1, self.number_of_frames + 1):
            self.find_neighbors(frame)
            if hasattr(self, "save_cells_info"):
                self.save_cells_info()
        return 0

    # ---- per-cell mean intensity (ti.py:1135-1150) ----------------------------------------------------------------------
    def calculate_mean_intensity(self, frame, valid_cells, intensity_img, type_name):
        """ti.py:1135-1150: mean of `intensity_img` over every cell (regionprops 'intensity_mean'), cached in the table's
        `mean_intensity_<type_name>` column; returns the means of the rows of `valid_cells`.  One device pass over the frame
        (tip_regionprops_i32 with an intensity plane); labels that do not occur have no mean upstream and none here."""
        labels = self.get_labels(frame)
        if labels is None:
            return 0
        column = "mean_intensity_" + type_name
        if column in valid_cells.columns:
            return valid_cells[column].to_numpy()
        rp = seg.regionprops_arrays(labels, intensity=np.asarray(intensity_img))
        present = np.flatnonzero(rp["area"] > 0)                    # regionprops_table lists the labels that occur
        present_labels = present + 1
        wanted = np.intersect1d(present_labels, valid_cells.index.to_numpy() + 1, return_indices=True)[1]
        table = self.get_cells_info(frame)
        if table is not None:
            table.loc[present_labels - 1, column] = rp["intensity_mean"][present]
        return rp["intensity_mean"][present][wanted]

    # ---- C5 -------------------------------------------------------------------------------------------------
    def calc_cell_types(self, type_marker_image, frame_number, type_name, threshold=0.1,
                        percentage_above_threshold=90, peak_window_size=0):
        cells_info = self.get_cells_info(frame_number)
        labels = self.get_labels(frame_number)
        if cells_info is None or labels is None:
            return 0
        new_type = type_name not in self.type_names
        if new_type:
            self.type_names.append(type_name)
            type_index = len(self.type_names) - 1
        else:
            type_index = self.type_names.index(type_name)
        img = np.asarray(type_marker_image, dtype=np.float64)
        rp = seg.regionprops_arrays(labels, intensity=img)
        present = rp["area"] > 0
        cell_indices = np.nonzero(present)[0]
        # per-label percentile(100 - p): two neighbouring order statistics of each region's pixels by radix select on the
        # device (tip_label_order_stats_f64), numpy's linear interpolation between them on the host
        marker_intensities = seg.percentile_per_label(labels, img, rp["area"].size, rp["area"],
                                                      100 - percentage_above_threshold)[cell_indices]
        if new_type:
            cells_info.loc[cell_indices, "mean_intensity_" + type_name] = rp["intensity_mean"][cell_indices]
        # validity follows the area rule of calculate_frame_cellinfo (ti.py:2360-2367); cells that become valid get neighbours
        areas = cells_info["area"].to_numpy()
        smallest, largest = self.min_cell_area * np.mean(areas), self.max_cell_area * np.mean(areas)
        now_valid = np.logical_and(areas < largest, areas > smallest)
        was_valid = cells_info["valid"].to_numpy() == 1
        self.find_neighbors(frame_number, only_for_labels=cells_info.index[np.logical_and(now_valid, ~was_valid)].to_numpy() + 1)
        cells_info = self.get_cells_info(frame_number)
        cells_info.loc[:, "valid"] = now_valid.astype(int)
        # a cell is positive when its percentile intensity exceeds `threshold` x the frame's 99th percentile (cells without
        # pixels have NaN there and land in neither set, as upstream) -- and, with a peak window, holds a local maximum
        cut = threshold * seg.percentile_frame(img, 99)
        pos_indices = cell_indices[marker_intensities > cut]
        neg_indices = cell_indices[marker_intensities <= cut]
        if peak_window_size > 0:
            peaks = find_local_maxima(img, window_size=peak_window_size)
            with_peak = np.unique(np.asarray(labels)[peaks]) - 1
            with_peak = with_peak[with_peak > 0]
            neg_indices = np.union1d(neg_indices, np.setdiff1d(pos_indices, with_peak))
            pos_indices = np.intersect1d(pos_indices, with_peak)
        for rows, flag in ((pos_indices, True), (neg_indices, False)):
            cells_info.loc[rows, "type"] = change_type(cells_info.loc[rows, "type"].to_numpy(), type_index, is_positive=flag)
        self.update_cell_types_by_cells_info(frame_number)
        return 0

    def update_cell_types_by_cells_info(self, frame):
        labels = self.get_labels(frame)
        cells_info = self.get_cells_info(frame)
        if labels is None or cells_info is None:
            return 0
        cell_types = self.get_cell_types(frame)
        if cell_types is None:
            cell_types = np.ones_like(labels) * INVALID_TYPE_INDEX
        n = cells_info.shape[0]
        # LUT per label: valid cells -> their type, invalid cells (by label column) -> 255, others keep old value
        lut = np.full(n + 1, -1, np.int64)
        valid = cells_info.valid.to_numpy() == 1
        types = cells_info.type.to_numpy().astype(np.int64)
        lut[1:][valid] = types[valid]
        lab = np.asarray(labels)
        sel = (lab > 0) & (lab <= n)
        mapped = np.where(sel, lut[np.where(sel, lab, 0)], -1)
        cell_types = np.where(mapped >= 0, mapped, cell_types)
        invalid_cells_labels = cells_info.label.to_numpy()[~valid]
        cell_types[np.isin(lab, invalid_cells_labels)] = INVALID_TYPE_INDEX
        self.set_cell_types(frame, cell_types)
        return 0

    # ---- C6 -------------------------------------------------------------------------------------------------
    def calc_neighbors_contact_matrix(self, frame):
        labels = self.get_labels(frame)
        cells_info = self.get_cells_info(frame)
        if labels is None or cells_info is None:
            return 0
        # the reference counts, inside every cell's bounding box, the pixels whose cross-footprint (max, min) of the labels
        # is the pair (larger, smaller): such a pixel touches both cells, hence lies in the box -- the count is a property
        # of the label map, taken in one device pass (tip_contact_pairs_i32)
        length = seg.contact_pairs(labels)
        max_index = cells_info.index.max()
        output = np.zeros((max_index + 1, max_index + 1))
        for index, neighbors in zip(cells_info.index.to_numpy(), cells_info["neighbors"].to_numpy()):
            me = int(index) + 1
            for nb in neighbors:
                output[index, nb - 1] = length.get((max(me, nb), min(me, nb)), 0)
        return output

    # ---- T3 -------------------------------------------------------------------------------------------------
    def track_cells_iterator(self, initial_frame=1, final_frame=-1, images=None, image_in_memory=False, use_piv=False):
        """ti.py:2037-2113, the label-lookup tracker: the previous frame's centroids (drift-corrected) are looked up in
        the current frame's 3x3-max-filtered label map, ids propagate one-to-one, unmatched cells get fresh ids.
        Generator yielding the frame numbers it finished, like the reference."""
        if use_piv:
            raise NotImplementedError("optical-flow (PIV) drift is out of scope (SURVEY.md 8f)")
        from .movie import assign_track_ids
        last = self.number_of_frames if final_frame == -1 else final_frame
        table = self.get_cells_info(initial_frame)
        if table is None:
            return 0
        ids = table.label.to_numpy().astype(np.int64)
        ids = assign_track_ids(None, None, ids.size, start_ids=ids)
        table.loc[:, "label"] = ids
        prev = dict(cx=table.cx.to_numpy().astype(np.float64).copy(), cy=table.cy.to_numpy().astype(np.float64).copy(),
                    ids=ids, empty=table.empty_cell.to_numpy(), frame=initial_frame)
        self.cells_number = max(self.cells_number, ids.max() if ids.size else 0)
        reuse_drifts = bool((self.drifts > 0).any())
        refresh_next = False
        for frame in range(initial_frame + 1, last + 1):
            if self.valid_frames[frame - 1] == 0:           # skipped frame: its drift is void, the next one is re-estimated
                if not np.isnan(self.drifts[frame - 1, 0]):
                    self.drifts[frame - 1, :] = np.nan
                    refresh_next = True
                continue
            if reuse_drifts and not refresh_next:
                dy, dx = self.drifts[frame - 1, 0], self.drifts[frame - 1, 1]
            else:
                dy, dx = self.update_drift(frame, prev["frame"], images=images, image_in_memory=image_in_memory)
            prev["cx"] -= dx
            prev["cy"] -= dy
            table = self.get_cells_info(frame)
            raw = self.get_labels(frame)
            if table is None or raw is None:
                continue
            lab = np.ascontiguousarray(raw, dtype=np.int32)
            qy = np.round(prev["cy"]).astype(np.int64)
            qx = np.round(prev["cx"]).astype(np.int64)
            hit = np.empty(qy.shape, np.int32)
            d_lab = _lib.DeviceBuffer(lab.nbytes).upload(lab)
            _lib.check(_lib.lib().tip_lookup_max3_i32_dev(_lib.dptr(d_lab.ptr), lab.shape[0], lab.shape[1], _lib.ptr(qy),
                                                         _lib.ptr(qx), ctypes.c_int64(qy.size), _lib.ptr(hit)))
            d_lab.free()
            hit = np.where(prev["empty"] == 0, hit, -1)
            ids = assign_track_ids(prev["ids"], hit, table.shape[0])
            table.loc[:, "label"] = ids
            self.cells_number = max(self.cells_number, ids.max() if ids.size else 0)
            prev = dict(cx=table.cx.to_numpy().astype(np.float64).copy(), cy=table.cy.to_numpy().astype(np.float64).copy(),
                        ids=ids, empty=table.empty_cell.to_numpy(), frame=frame)
            yield frame
        return 0

    # ---- T1 -------------------------------------------------------------------------------------------------
    @staticmethod
    def tracking_dist_func(first, second):
        """ti.py:1935-1938: the linking distance between two (cy, cx, area) features."""
        return np.sqrt((first[0] - second[0]) ** 2 + (first[1] - second[1]) ** 2 +
                       0.5 * (np.sqrt(first[2]) - np.sqrt(second[2])) ** 2)

    def track_cells_iterator_with_trackpy(self, initial_frame=1, final_frame=-1, images=None, image_in_memory=False):
        """ti.py:1881-1933.  Upstream hands the per-frame (cy, cx, area) tables -- valid, non-empty cells, centroids moved
        by the cumulative drift -- to trackpy.link_df_iter(search_range=100, adaptive_stop=10, memory=3) and writes
        `label = particle + 1`.  trackpy is a third-party package that is not part of the reference: the linking model
        is restated in linking.py for these parameters (PARITY UNPINNED, see there).  Generator yielding frame numbers."""
        from .linking import FrameLinker, embed
        last = self.number_of_frames if final_frame == -1 else final_frame
        reuse_drifts = bool((self.drifts > 0).any())
        linker = FrameLinker(search_range=100, adaptive_stop=10, memory=3)
        refresh_next = False
        total_dy = total_dx = 0.0
        previous_frame = 0
        for frame in range(initial_frame, last + 1):
            if np.isnan(self.drifts[frame - 1, :]).any():
                refresh_next = True
            if self.valid_frames[frame - 1] == 0:
                if not np.isnan(self.drifts[frame - 1, 0]):
                    self.drifts[frame - 1, :] = np.nan
                    refresh_next = True
                continue
            table = self.get_cells_info(frame)
            if table is None:
                continue
            rows = table.index[(table.valid.to_numpy() == 1) & (table.empty_cell.to_numpy() == 0)]
            if frame > initial_frame:
                if reuse_drifts and not refresh_next:
                    dy, dx = self.drifts[frame - 1, 0], self.drifts[frame - 1, 1]
                else:
                    dy, dx = self.update_drift(frame, previous_frame, images=images, image_in_memory=image_in_memory)
                total_dy += dy
                total_dx += dx
            previous_frame = frame
            if len(rows) == 0:      # (upstream would fail on an empty table; nothing to link in this frame)
                continue
            particles = linker.link(embed(table.loc[rows, "cy"].to_numpy() + total_dy, table.loc[rows, "cx"].to_numpy() + total_dx,
                                          table.loc[rows, "area"].to_numpy()))
            table.loc[rows, "label"] = particles + 1
            self.cells_number = max(self.cells_number, int(particles.max()) + 1)
            self.fix_duplicated_label_cells_in_frame(frame)
            yield frame
        return 0

    def track_cells_with_trackpy(self, initial_frame=1, final_frame=-1, images=None, image_in_memory=False):
        """ti.py:1874-1879: run the generator to its end, return the last frame it finished."""
        last_frame = initial_frame
        for frame in self.track_cells_iterator_with_trackpy(initial_frame, final_frame, images, image_in_memory):
            last_frame = frame
        return last_frame

    def fix_duplicated_label_cells_in_frame(self, frame):
        """ti.py:4288-4310: of the cells sharing a track id the first valid one (else the first) keeps it; the others get
        `row index + 1`, or ids above the frame's maximum where that is taken too."""
        table = self.get_cells_info(frame)
        if table is None:
            return 0
        ids = table.label.to_numpy()
        present = np.unique(ids)
        losers = []
        for dup in np.unique(ids[table.label.duplicated().to_numpy()]):
            group = table.index[ids == dup]
            ok = group[table.loc[group, "valid"].to_numpy() == 1]
            winner = ok[0] if len(ok) else group[0]
            losers.extend(i for i in group if i != winner)
        if not losers:
            return 0
        losers = np.asarray(losers)
        fresh = losers + 1
        clash = np.isin(fresh, present)
        fresh[clash] = present.max() + np.arange(1, int(clash.sum()) + 1)
        table.loc[losers, "label"] = fresh
        return 0

    @staticmethod
    def calculate_refine_drift(previous_image, current_image, course_shift_x, course_shift_y):
        """ti.py:1941-1980: crop the overlap given by the floored coarse shift, refine by phase cross-correlation."""
        from ._registration import phase_cross_correlation
        rx, ry = int(np.floor(course_shift_x)), int(np.floor(course_shift_y))
        sl = lambda r: (slice(r, None), slice(None, -r)) if r > 0 else ((slice(None, r), slice(-r, None)) if r < 0 else
                                                                        (slice(None), slice(None)))
        (p0, c0), (p1, c1) = sl(rx), sl(ry)
        previous_img = previous_image[p0, p1]
        current_img = current_image[c0, c1]
        refined_shift, _, _ = phase_cross_correlation(previous_img, current_img, upsample_factor=100)
        return rx + refined_shift[-2], ry + refined_shift[-1]

    def update_drift(self, frame, previous_frame, images=None, image_in_memory=False):
        """ti.py:1982-2035: stage shift (when a stage table is loaded) + phase-correlation refinement of the overlap.
        x/y are swapped between the stage table and the image, exactly as upstream: returns (shift_y, shift_x)."""
        stage = getattr(self, "stage_locations", None)
        if stage is not None:
            shift = (stage.loc[frame - 1, ["z", "y", "x"]].to_numpy() - stage.loc[previous_frame - 1, ["z", "y", "x"]].to_numpy()) / \
                stage.loc[frame - 1, ["physical_size_z", "physical_size_y", "physical_size_x"]].to_numpy()
        else:
            shift = (0, 0)
        shift_x = shift[-2]
        shift_y = shift[-1]
        if images is not None:
            previous_image = images[previous_frame - 1]
            current_image = images[frame - 1]
            if not image_in_memory:
                previous_image = previous_image.compute()
                current_image = current_image.compute()
            shift_x, shift_y = self.calculate_refine_drift(np.asarray(previous_image), np.asarray(current_image), shift_x, shift_y)
        self.drifts[frame - 1, 0] = shift_y
        self.drifts[frame - 1, 1] = shift_x
        return shift_y, shift_x

    # ---- re-linking one frame pair with local drifts (ti.py:2115-2246) ---------------------------------------------------
    def get_cell_id_by_position(self, frame, pos):
        """ti.py:475-488: the track id of the cell under pixel (x, y)."""
        labels, table = self.get_labels(frame), self.get_cells_info(frame)
        if labels is None or table is None:
            return 0
        x, y = pos
        row = labels[y, x] - 1
        if row < 0:
            return 0
        try:
            return table.label[row]
        except (IndexError, KeyError):
            return 0

    def get_cell_centroid_by_id(self, frame, id):
        """ti.py:491-498."""
        cell = self.get_cells_info(frame).query("label == %d and valid == 1 and empty_cell == 0" % id)
        if cell.shape[0] < 1:
            return None
        return cell.cx.values[0], cell.cy.values[0]

    def fix_one_frame_tracking_using_local_drifts(self, start_frame, end_frame, images, step_size=100, window_size=700,
                                                  image_in_memory=False, start_frame_pos=None, end_frame_pos=None, link=None):
        """ti.py:2115-2246: re-link end_frame (which must be the first valid frame after start_frame) to start_frame after
        moving start_frame's centroids by a LOCAL drift map -- the mean refined drift of the sliding windows over each
        centroid, every window through the device phase correlation (_registration.local_drifts) -- then carry the new
        track ids through the frames that follow with a running old -> new table.  `link` stands in for trackpy.link
        (same call signature; default: the linker of linking.py behind that signature, parity unpinned like T1)."""
        from ._registration import local_drifts, sample_local_drift
        next_frame = -1
        for frame in range(start_frame + 1, self.number_of_frames):          # (the last frame is never a candidate, as upstream)
            if self.valid_frames[frame - 1] == 1:
                next_frame = frame
                break
        if next_frame < 0 or next_frame != end_frame:
            return 0
        stage = getattr(self, "stage_locations", None)
        if start_frame_pos is not None and end_frame_pos is not None:
            a = self.get_cell_centroid_by_id(start_frame, self.get_cell_id_by_position(start_frame, start_frame_pos))
            b = self.get_cell_centroid_by_id(end_frame, self.get_cell_id_by_position(end_frame, end_frame_pos))
            shift = (b[1] - a[1], b[0] - a[0])
        elif stage is not None:
            shift = (stage.loc[next_frame - 1, ["z", "y", "x"]].to_numpy() - stage.loc[start_frame - 1, ["z", "y", "x"]].to_numpy()) / \
                stage.loc[frame - 1, ["physical_size_z", "physical_size_y", "physical_size_x"]].to_numpy()
        else:
            shift = (0, 0)
        first_image, second_image = images[start_frame - 1], images[next_frame - 1]
        if not image_in_memory:
            first_image, second_image = first_image.compute(), second_image.compute()
        keep = "valid == 1 and empty_cell == 0"
        first = self.get_cells_info(start_frame).query(keep)
        cx, cy = np.copy(first.cx.to_numpy()), np.copy(first.cy.to_numpy())
        # x / y are swapped between the stage table and the image, and the map is read at [cx, cy], as upstream
        drifts = local_drifts(first_image, second_image, shift[-2], shift[-1], step_size, window_size)
        rows, cols = np.round(cx).astype(int), np.round(cy).astype(int)
        H, W = np.shape(first_image)
        if rows.size and (rows.max() >= H or cols.max() >= W or rows.min() < -H or cols.min() < -W):
            raise IndexError("index %d is out of bounds for the %dx%d drift map" % (max(rows.max(), cols.max()), H, W))
        rows, cols = np.where(rows < 0, rows + H, rows), np.where(cols < 0, cols + W, cols)   # numpy's wrap, as upstream's map[cx, cy]
        dx, dy = sample_local_drift(drifts, rows, cols)
        cx -= dx
        cy -= dy
        first_tab = pd.DataFrame({"cx": cx, "cy": cy, "area": np.copy(first.area.to_numpy()), "frame_index": np.zeros(cx.shape),
                                  "label": np.copy(first.label.to_numpy())}, index=first.index)
        second_info = self.get_cells_info(next_frame)
        second = second_info.query(keep)
        second_tab = pd.DataFrame({"cx": second.cx.to_numpy(), "cy": second.cy.to_numpy(), "area": np.copy(second.area.to_numpy()),
                                   "frame_index": np.ones((second.shape[0],)), "label": np.copy(second.label.to_numpy())},
                                  index=second.index)
        if link is None:
            link = _link_two_frames
        linked = link(pd.concat([first_tab, second_tab]), search_range=100, adaptive_stop=10, pos_columns=["cy", "cx", "area"],
                      t_column="frame_index", memory=0, neighbor_strategy='BTree', dist_func=self.tracking_dist_func)
        first_ids = linked.query("frame_index == 0").label.to_numpy()
        second_linked = linked.query("frame_index == 1")
        particles = second_linked.particle.to_numpy()
        old_ids = second_linked.label.to_numpy()
        new_ids = np.copy(old_ids)
        # linked cells take the id of their partner in start_frame; unlinked cells whose id also lives in start_frame were
        # linked before and are not any more: fresh ids; the other unlinked cells keep theirs
        is_linked = particles < first_ids.size
        new_ids[is_linked] = first_ids[particles[is_linked]]
        stale = np.logical_and(~is_linked, np.isin(new_ids, first_ids))
        free = max(np.max(first_ids), np.max(new_ids)) + 1
        new_ids[stale] = np.arange(free, free + int(stale.sum()))
        second_info.loc[second_linked.index.to_numpy(), "label"] = new_ids
        # ids of start_frame that neither appear in end_frame nor were handed out keep their meaning (a cell may skip a frame)
        skipping = first_ids[np.logical_and(~np.isin(first_ids, old_ids, assume_unique=True),
                                            ~np.isin(first_ids, new_ids, assume_unique=True))]
        old_ids, new_ids = np.hstack([old_ids, skipping]), np.hstack([new_ids, skipping])
        for frame in range(next_frame + 1, self.number_of_frames):
            if self.valid_frames[frame - 1] != 1:
                continue
            info = self.get_cells_info(frame)
            cells = info.query(keep)
            ids = cells.label.to_numpy()
            known = np.isin(ids, old_ids, assume_unique=True)
            taken = np.isin(ids, new_ids, assume_unique=True)
            untouched = np.logical_and(~known, ~taken)          # neither renamed nor colliding: maps to itself
            old_ids, new_ids = np.hstack([old_ids, ids[untouched]]), np.hstack([new_ids, ids[untouched]])
            clash = np.logical_and(~known, taken)               # an id that now means another cell: fresh ids
            free = max(np.max(old_ids), np.max(new_ids)) + 1
            old_ids = np.hstack([old_ids, ids[clash]])
            new_ids = np.hstack([new_ids, np.arange(free, free + int(clash.sum()))])
            used = np.isin(old_ids, ids, assume_unique=True)
            by_key = np.argsort(old_ids[used])
            by_id = np.argsort(ids)
            info.loc[cells.index.to_numpy()[by_id], "label"] = new_ids[used][by_key]
        return 0

    # ---- T4 -------------------------------------------------------------------------------------------------
    def get_trackking_labels(self, frame):
        labels = self.get_labels(frame)
        cells_info = self.get_cells_info(frame)
        if labels is None or cells_info is None:
            return None
        cell_ids = np.ascontiguousarray(np.insert(cells_info.label.to_numpy(), 0, 0), dtype=np.int64)
        lab32 = np.ascontiguousarray(labels, dtype=np.int32)
        out = np.empty(lab32.shape, np.int64)
        _lib.check(_lib.lib().tip_lut_gather_i32(_lib.ptr(lab32), _lib.ptr(cell_ids), ctypes.c_int64(cell_ids.size),
                                                 _lib.ptr(out), ctypes.c_int64(lab32.size)))
        return out


    # ---- overlays (ti.py:584-607, 2585-2645) and event detection (ti.py:609-789) ---------------------------
    def type_name_to_index(self, type_name):
        """ti.py:366-372 (names only; the "pos / neg" list form goes through type_pos_neg_list_to_indices when the host class has it)."""
        names = list(getattr(self, "type_names", []))
        if type_name in names:
            return names.index(type_name)
        if ("pos" in type_name or "neg" in type_name) and hasattr(self, "type_pos_neg_list_to_indices"):
            return self.type_pos_neg_list_to_indices(type_name)
        return -1

    def get_cell_data_by_label(self, cell_id, frame):
        """ti.py:911-919: the valid, non-empty rows of the frame's table whose track id is cell_id (None when there is none)."""
        table = self.get_cells_info(frame)
        if table is None:
            return None
        rows = table.query("label == %d and valid == 1 and empty_cell == 0" % cell_id)
        return rows if rows.shape[0] > 0 else None

    @staticmethod
    def _disks_image(shape, centers, radius, colors):
        """(3, Y, X) float64: filled discs (skimage.draw.disk), later ones over earlier ones -- on the device"""
        Y, X = int(shape[0]), int(shape[1])
        cy = np.ascontiguousarray([c[0] for c in centers], dtype=np.float64)
        cx = np.ascontiguousarray([c[1] for c in centers], dtype=np.float64)
        rgb = np.ascontiguousarray(colors, dtype=np.float64).reshape(-1)
        out = np.empty((3, Y, X), np.float64)
        _lib.check(_lib.lib().tip_draw_disks_f64(Y, X, int(cy.size), _lib.ptr(cy), _lib.ptr(cx), ctypes.c_double(float(radius)),
                                                 _lib.ptr(rgb), _lib.ptr(out)))
        return out

    def draw_cell_types(self, frame_number, type_name=""):
        """ti.py:2585-2593: positive cells in POS_COLOR, the other valid cells in NEG_COLOR."""
        type_index = self.type_name_to_index(type_name)
        cell_types = self.get_cell_types(frame_number)
        if (not isinstance(type_index, tuple) and type_index < 0) or cell_types is None:
            return 0
        if isinstance(type_index, tuple):
            must = sum(1 << int(t) for t in set(type_index[0]))
            lack = sum(1 << int(t) for t in set(type_index[1]))
        else:
            must, lack = 1 << int(type_index), 0
        types = np.ascontiguousarray(np.asarray(cell_types).astype(np.uint8))
        out = np.empty((3,) + types.shape, np.float64)
        pos, neg = np.asarray(POS_COLOR, np.float64), np.asarray(NEG_COLOR, np.float64)
        _lib.check(_lib.lib().tip_draw_cell_types_u8(_lib.ptr(types), ctypes.c_long(types.size), must, lack, _lib.ptr(pos), _lib.ptr(neg),
                                                     _lib.ptr(out)))
        return out

    def draw_neighbors_connections(self, frame_number):
        """ti.py:2595-2606: a line between the (truncated) centroids of every cell and each of its neighbours."""
        labels = self.get_labels(frame_number)
        table = self.get_cells_info(frame_number)
        if labels is None or table is None:
            return np.zeros(np.shape(labels) if labels is not None else (0, 0))
        cy, cx = table.cy.to_numpy(), table.cx.to_numpy()
        ends = []
        for row, neigh in enumerate(table.neighbors):
            for nl in list(neigh):
                ends.append((int(cy[row]), int(cx[row]), int(cy[nl - 1]), int(cx[nl - 1])))
        ends = np.ascontiguousarray(ends, dtype=np.int32).reshape(-1, 4)
        Y, X = labels.shape
        out = np.empty((3, Y, X), np.float64)
        rgb = np.asarray(NEIGHBORS_COLOR, np.float64)
        _lib.check(_lib.lib().tip_draw_lines_f64(int(Y), int(X), int(ends.shape[0]), _lib.ptr(ends), _lib.ptr(rgb), _lib.ptr(out)))
        return out

    def draw_cell_tracking(self, frame_number, cell_label, radius=5):
        """ti.py:2608-2623: a disc on the tracked cell (all tracks in their cycle colours for label 0); a cell that is not in the
        frame gives upstream's plain 2-D zero image."""
        if cell_label == 0:
            return self.draw_all_cell_tracking(frame_number)
        labels = self.get_labels(frame_number)
        if labels is None:
            return 0
        cell = self.get_cell_data_by_label(cell_label, frame_number)
        if cell is None or cell.empty_cell.values[0] == 1:
            return np.zeros(labels.shape)
        return self._disks_image(labels.shape, [(cell.cy.values[0], cell.cx.values[0])], radius, [TRACK_COLOR])

    def draw_all_cell_tracking(self, frame):
        """ti.py:2625-2635: every pixel in the colour of its track id modulo six, background black."""
        track = np.ascontiguousarray(self.get_trackking_labels(frame), dtype=np.int32)
        out = np.empty((3,) + track.shape, np.float64)
        cyc = np.ascontiguousarray(TRACKING_COLOR_CYCLE, dtype=np.float64).reshape(-1)
        _lib.check(_lib.lib().tip_draw_tracking_i32(_lib.ptr(track), ctypes.c_long(track.size), _lib.ptr(cyc), _lib.ptr(out)))
        return out

    def draw_marking_points(self, frame_number, radius=5):
        """ti.py:2637-2645: discs on the shape-fitting points ((x, y) pairs)."""
        labels = self.get_labels(frame_number)
        pts = list(getattr(self, "shape_fitting_points", None) or [])
        return self._disks_image(labels.shape, [(p[1], p[0]) for p in pts], radius, [MARKING_COLOR] * len(pts))

    def draw_events(self, frame, radius=5):
        """ti.py:584-607: a disc in the event's colour on every cell with an event spanning the frame (and on a division's daughter)."""
        labels = self.get_labels(frame)
        if labels is None:
            return 0
        centers, colors = [], []
        for _, event in self.events.iterrows():
            if not (event.start_frame <= frame <= event.end_frame):
                continue
            cell = self.get_cell_data_by_label(event.cell_id, frame)
            if cell is None or cell.empty_cell.values[0] == 1:
                continue
            color = EVENTS_COLOR[event.type]
            centers.append((cell.cy.values[0], cell.cx.values[0]))
            colors.append(color)
            if event.type == "division":
                other = self.get_cell_data_by_label(event.daughter_id, frame)
                if other is not None and other.empty_cell.values[0] == 0:
                    centers.append((other.cy.values[0], other.cx.values[0]))
                    colors.append(color)
        return self._disks_image(labels.shape, centers, radius, colors)

    # -- the events table (ti.py:500-582, 998-1033): what find_events_iterator reports into ---------------------------------------
    def is_frame_valid(self, frame):
        return self.valid_frames[frame - 1] == 1

    def find_event_frame(self, event):
        """ti.py:998-1033, as written: the frame an event is pinned to.  (A delamination is pinned to the frame before the first
        valid one in which its cell is present and non-empty -- or whose area has fallen below min_cell_area --; a division to the
        frame before the daughter first shows; a differentiation's test compares the numeric type with the string "HC", so it ends
        at upstream's fall-through: the start frame.)"""
        first, last, kind = event["start_frame"], event["end_frame"], event["type"]
        if kind == "delamination":
            seen = first
            for frame in range(first, last + 1):
                if self.is_frame_valid(frame):
                    cell = self.get_cell_data_by_label(event["cell_id"], frame)
                    if cell is None or cell.empty_cell.values[0] == 0:
                        return seen
                    elif float(cell.area.values[0]) < self.min_cell_area:
                        return frame
                    seen = frame
        if kind == "division":
            seen = first
            for frame in range(first, last + 1):
                if self.is_frame_valid(frame):
                    cell = self.get_cell_data_by_label(event["daughter_id"], frame)
                    if cell is not None and cell.empty_cell.values[0] == 0:
                        return seen
                    seen = frame
        if kind == "differentiation":
            seen = first
            for frame in range(first, last + 1):
                if self.is_frame_valid(frame):
                    cell = self.get_cell_data_by_label(event["cell_id"], frame)
                    if cell is not None and cell.type.values[0] == "HC":
                        return seen
                    seen = frame
        print("Problem with finding event frame")
        return first

    def delete_event(self, start_frame, start_pos):
        cell_id = self.get_cell_id_by_position(start_frame, start_pos)
        hit = self.events.query("start_frame == %d and (cell_id == %d or daughter_id == %d)" % (start_frame, cell_id, cell_id))
        if hit.size > 0:
            self.events.drop(hit.index, inplace=True)
        return 0

    def add_event(self, event_type, start_frame, end_frame, start_pos=None, end_pos=None, second_end_pos=None, start_cell_id=None,
                  daughter_cell_id=None, source="manual"):
        """ti.py:500-560: a row of the events table -- cell ids from clicked positions or positions from the cells' centroids (the end
        position walks back from end_frame to the last frame that still has the cell; upstream's loop also gives up when that walk
        would pass start_frame on its NEXT step, found or not -- kept), the daughter's data for divisions, the pinned frame."""
        if start_frame is None:
            return 0
        if event_type == "delete event":
            self.delete_event(start_frame, start_pos)
            return 0
        if start_pos is not None:
            start_cell_id = self.get_cell_id_by_position(start_frame, start_pos)
        else:
            start_pos = self.get_cell_centroid_by_id(start_frame, start_cell_id)
            if start_pos is None:
                return 0
        if end_pos is not None:
            end_cell_id = self.get_cell_id_by_position(end_frame, end_pos)
        else:
            end_cell_id = start_cell_id
            back = 0
            while end_pos is None:
                end_pos = self.get_cell_centroid_by_id(end_frame - back, start_cell_id)
                back += 1
                if end_frame - back < start_frame:
                    return 0
        if start_cell_id != end_cell_id and event_type == "differentiation":
            self.fix_cell_label(end_frame, end_pos, start_cell_id)         # (manual-editing code of the host class, ti.py:3029-)
        row = {"type": event_type, "start_frame": start_frame, "end_frame": end_frame, "start_pos_x": start_pos[0],
               "start_pos_y": start_pos[1], "end_pos_x": end_pos[0], "end_pos_y": end_pos[1], "daughter_pos_x": 0, "daughter_pos_y": 0,
               "cell_id": start_cell_id, "daughter_id": 0, "source": source}
        if second_end_pos is not None or daughter_cell_id is not None:
            second_id = self.get_cell_id_by_position(end_frame, second_end_pos) if daughter_cell_id is None else daughter_cell_id
            if second_end_pos is None:
                second_end_pos = self.get_cell_centroid_by_id(end_frame, daughter_cell_id)
                if second_id is None:
                    return 0
            if start_cell_id != end_cell_id and start_cell_id == second_id:
                second_id = end_cell_id
            row["daughter_pos_x"], row["daughter_pos_y"], row["daughter_id"] = second_end_pos[0], second_end_pos[1], second_id
        row["significant_frame"] = int(self.find_event_frame(row))
        self.events = pd.concat([self.events, pd.DataFrame(row, index=[0])], ignore_index=True)
        return 0

    @staticmethod
    def detect_edge_cells(labels):
        """ti.py:609-612: row indices (label - 1) of the cells that touch the frame's border."""
        border = np.hstack([labels[0, :], labels[:, 0], labels[-1, :], labels[:, -1]])
        return np.unique(border[border > 0]) - 1

    def find_valid_frames(self, initial_frame, final_frame):
        """ti.py:621-626 (the upper end is exclusive, as upstream's arange)."""
        first, last = max(1, initial_frame), min(self.number_of_frames, final_frame)
        idx = np.arange(first, last) - 1
        return idx[np.asarray(self.valid_frames)[idx] == 1] + 1

    def find_events(self, initial_frame, final_frame):
        last = initial_frame
        for frame in self.find_events_iterator(initial_frame, final_frame):
            last = frame
        return last

    def find_events_iterator(self, initial_frame=1, final_frame=-1, differentiation_type_name="", differentiation_type_index=0):
        """ti.py:636-789: delaminations (a valid interior cell that is gone in the next valid frame while all its neighbours stay),
        differentiations (a cell that turns positive for the type while its neighbourhood stays) and divisions (a new interior cell
        whose centroid and a staying neighbour's centroid fall into ONE cell of the previous frame's label map), reported through
        self.add_event.  Upstream's quirks are part of the behaviour and kept: a neighbour LABEL is looked up as a ROW of the valid
        cells' table (no minus one), the border cells are those of the first frame throughout, drifts are truncated before they are
        added, and a division's end frame walks back to the daughter's last valid frame."""
        if differentiation_type_name:
            index = self.type_name_to_index(differentiation_type_name)
            if isinstance(index, tuple) or index >= 0:
                differentiation_type_index = index
        if final_frame == -1:
            final_frame = self.number_of_frames
        labels = table = None
        initial_frame -= 1
        while labels is None or table is None:
            initial_frame += 1
            labels, table = self.get_labels(initial_frame), self.get_cells_info(initial_frame)

        def valid_rows(tab):
            return tab.query("valid == 1 and empty_cell == 0")

        def positives(rows):
            return rows.loc[is_positive_for_type(rows.type.to_numpy(), differentiation_type_index)].label

        prev_rows = valid_rows(table)
        prev_ids, prev_pos = prev_rows.label, positives(prev_rows)
        prev_labels = np.copy(labels)
        first_edge_ids = table.label[self.detect_edge_cells(labels)]
        skipped = 0

        def neighbourhood_stays(rows, cell_id, ids, blocked, frame):
            """every neighbour label, read as a row of `ids`, is a cell that neither vanished nor sits on the border; None: no such cell"""
            neigh = rows.query("label == %d" % cell_id).neighbors
            if neigh.shape[0] < 1:
                return None
            if neigh.shape[0] > 1:
                print("Warning: more than one cell with the same id. frame: %d, cell id: %d" % (frame, cell_id))
            for n in list(neigh.values[0]):
                if n not in ids:                       # (membership in the Series' INDEX, as upstream)
                    return False
                if blocked(ids[n]):
                    return False
            return True

        for frame in range(initial_frame + 1, final_frame + 1):
            if not self.valid_frames[frame - 1]:
                skipped += 1
                continue
            around = self.find_valid_frames(frame - 5, frame + 5)
            start_frame, end_frame = np.min(around), np.max(around)
            labels, table = self.get_labels(frame), self.get_cells_info(frame)
            cur_rows = valid_rows(table)
            cur_ids, cur_pos = cur_rows.label, positives(cur_rows)
            cur_edge_ids = table.label[self.detect_edge_cells(labels)]
            if skipped < 3:
                gone = np.setdiff1d(prev_ids.values, cur_ids.values)
                blocked_prev = lambda i: i in gone or i in first_edge_ids.values      # noqa: E731
                for cid in gone:                                                    # delaminations
                    if cid in first_edge_ids.values:
                        continue
                    if neighbourhood_stays(prev_rows, cid, prev_ids, blocked_prev, frame):
                        self.add_event("delamination", start_frame, frame, start_cell_id=cid, source="automatic")
                both = np.intersect1d(cur_ids.values, prev_ids.values)
                for cid in np.intersect1d(np.setdiff1d(cur_pos.values, prev_pos.values), both):      # differentiations
                    if neighbourhood_stays(prev_rows, cid, prev_ids, blocked_prev, frame):
                        self.add_event("differentiation", start_frame, end_frame, start_cell_id=cid, source="automatic")
                for cid in np.setdiff1d(cur_ids.values, prev_ids.values):            # divisions
                    if cid in cur_edge_ids.values:
                        continue

                    def previous_label_under(cell):
                        """label of the previous frame's map under the cell's (drift-shifted) rounded centroid; None outside the frame"""
                        cen = self.get_cell_centroid_by_id(frame, cell)
                        px, py = int(np.round(cen[0])), int(np.round(cen[1]))
                        if self.drifts is not None:
                            drift = self.drifts[frame - 1, :]
                            if drift[0] != np.nan:
                                px += int(drift[1])
                                py += int(drift[0])
                        if px < 0 or px >= prev_labels.shape[1] or py < 0 or py >= prev_labels.shape[0]:
                            return None
                        return prev_labels[py, px]

                    mother_label = previous_label_under(cid)
                    if mother_label is None:
                        continue
                    neigh = cur_rows.query("label == %d" % cid).neighbors
                    if neigh.shape[0] < 1:
                        continue
                    if neigh.shape[0] > 1:
                        print("Warning: more than one cell with the same id. frame: %d, cell id: %d" % (frame, cid))
                    found, mother_id, daughter_pos, division_end = False, None, None, end_frame
                    for n in list(neigh.values[0]):
                        if n not in cur_ids:
                            found = False
                            break
                        nid = cur_ids[n]
                        if nid in both and nid not in cur_edge_ids.values:
                            under = previous_label_under(nid)
                            if under is None:
                                continue
                            if under == mother_label:
                                division_end, daughter_pos = end_frame + 1, None
                                while daughter_pos is None:
                                    division_end -= 1
                                    if self.valid_frames[division_end - 1] == 1:
                                        daughter_pos = self.get_cell_centroid_by_id(division_end, cid)
                                found, mother_id = True, nid
                    if found:
                        self.add_event("division", start_frame, division_end, start_cell_id=mother_id, daughter_cell_id=cid,
                                       second_end_pos=daughter_pos, source="automatic")
            prev_rows, prev_ids, prev_pos = cur_rows, cur_ids, cur_pos
            prev_labels = np.copy(labels)
            skipped = 0
            yield frame
        return 0


EVENTS_INFO_SPEC = {"type": "TBA", "start_frame": 0, "end_frame": 0, "start_pos_x": 0, "start_pos_y": 0, "end_pos_x": 0,
                    "end_pos_y": 0, "daughter_pos_x": 0, "daughter_pos_y": 0, "cell_id": 0, "daughter_id": 0,
                    "significant_frame": 0, "source": "manual"}      # ti.py:53-65


class Tissue(TissueHipMixin):
    """In-memory host of the hot methods (per-frame labels / cell tables / type maps kept in lists) that reads and writes
    the reference's `.seg` archives (ti.py:3474-3524, 3616-3757), so that the unmodified GUI opens what the GPU pipeline
    produced and vice versa.  The reference's interactive state (undo, line drawing, events editing, statistics) is out
    of scope."""

    def __init__(self, number_of_frames, data_path=None, channel_names=(), max_cell_area=10, min_cell_area=0.1,
                 load_to_memory=True):
        self.number_of_frames = number_of_frames
        self.data_path = data_path
        self.channel_names = list(channel_names)
        self.max_cell_area = max_cell_area
        self.min_cell_area = min_cell_area
        self.type_names = []
        self.drifts = np.zeros((number_of_frames, 2))
        self.valid_frames = np.ones((number_of_frames,)).astype(int)
        self.cells_number = 0
        self._labels = [None] * number_of_frames
        self._cells_info = [None] * number_of_frames
        self._cell_types = [None] * number_of_frames
        self.last_action = []
        self._neighbors_labels = (0, 0)
        self.last_added_line = []
        self.events = make_df(0, EVENTS_INFO_SPEC)
        self.shape_fitting_results = [dict() for _ in range(number_of_frames)]
        self.fake_channels = []

    # -- .seg archives: a flat zip (deflate) of per-frame files and movie-wide tables; member names as upstream ---------
    def _archive_members(self):
        """(member name, writer(file object)) for everything that is set."""
        import json
        import pickle
        members = []
        for k in range(self.number_of_frames):
            frame = k + 1
            if self._labels[k] is not None:
                members.append(("frame_%d_labels.npy" % frame, lambda fh, a=self._labels[k]: np.save(fh, a)))
            if self._cell_types[k] is not None:
                members.append(("frame_%d_types.npy" % frame, lambda fh, a=self._cell_types[k]: np.save(fh, a)))
            if self._cells_info[k] is not None:
                members.append(("frame_%d_data.pkl" % frame, lambda fh, df=self._cells_info[k]: df.to_pickle(fh, compression=None)))
        members.append(("events_data.pkl", lambda fh: self.events.to_pickle(fh, compression=None)))
        if self.drifts is not None:
            members.append(("drifts.npy", lambda fh: np.save(fh, self.drifts)))
        if self.valid_frames is not None:
            members.append(("valid_frames.npy", lambda fh: np.save(fh, self.valid_frames)))
        members.append(("shape_fitting_data.json", lambda fh: fh.write(json.dumps(self.shape_fitting_results).encode())))
        for name, value in (("cell_type_names.pkl", self.type_names), ("channel_names.pkl", self.channel_names),
                            ("fake_channels.pkl", self.fake_channels)):
            if value is not None:
                members.append((name, lambda fh, v=value: pickle.dump(v, fh)))
        return members

    def save(self, path):
        """ti.py:3716-3731: write `<path>.seg`; a generator of percent done, like upstream's (the GUI drives a progress
        bar with it)."""
        import io
        import zipfile
        members = self._archive_members()
        with zipfile.ZipFile(path.replace(".seg", "") + ".seg", "w", zipfile.ZIP_DEFLATED) as z:
            for i, (name, write) in enumerate(members):
                yield 100.0 * i / len(members)
                buf = io.BytesIO()
                write(buf)
                z.writestr(name, buf.getvalue())
        return 0

    def load(self, path, type_name=""):
        """ti.py:3733-3757: read a `.seg` archive (written here or by the reference); generator of percent done.
        Old single-type archives are upgraded as upstream does (ti.py:4211-4228)."""
        import io
        import json
        import pickle
        import re
        import zipfile
        with zipfile.ZipFile(path, "r") as z:
            names = z.namelist()
            for i, name in enumerate(names):
                yield 100.0 * i / len(names)
                raw = io.BytesIO(z.read(name))
                m = re.fullmatch(r"frame_(\d+)_(labels\.npy|types\.npy|data\.pkl)", name)
                if m:
                    k = int(m.group(1)) - 1
                    if not (0 <= k < self.number_of_frames):
                        continue
                    if m.group(2) == "labels.npy":
                        self._labels[k] = np.load(raw)
                    elif m.group(2) == "types.npy":
                        types = np.load(raw)
                        if types.max() <= 2 and types.min() >= 0:          # old single-type version (ti.py:4225-4228):
                            types[types == 0] = INVALID_TYPE_INDEX           # 0 meant invalid, 2 meant "SC" = negative for
                            types[types == 2] = 0                            # every type, in this order
                        self._cell_types[k] = types
                    else:
                        info = pd.read_pickle(raw, compression=None)
                        if len(info) and isinstance(info.at[info.index[0], "type"], str):
                            info.replace({"HC": 1, "SC": 0, "invalid": 0}, inplace=True)
                            if type_name and not self.type_names:
                                self.type_names = [type_name]
                        self._cells_info[k] = info
                elif name == "events_data.pkl":
                    self.events = pd.concat([self.events, pd.read_pickle(raw, compression=None)])
                    self.events["source"] = self.events["source"].fillna("manual")          # ti.py:3526-3536
                    self.events.drop_duplicates(inplace=True, ignore_index=True)
                elif name == "drifts.npy":
                    self.drifts = np.load(raw)
                elif name == "valid_frames.npy":
                    self.valid_frames = np.load(raw)
                elif name == "shape_fitting_data.json":
                    self.shape_fitting_results = json.loads(raw.getvalue().decode())
                elif name == "cell_type_names.pkl":
                    self.type_names = pickle.load(raw)
                elif name == "channel_names.pkl":
                    self.channel_names = pickle.load(raw)
                elif name == "fake_channels.pkl":
                    self.fake_channels = pickle.load(raw)
        return 0

    def set_labels(self, frame_number, labels, reset_data=False):
        if reset_data:
            self._cells_info[frame_number - 1] = None
            self._cell_types[frame_number - 1] = None
        self._labels[frame_number - 1] = labels

    def get_labels(self, frame_number):
        return self._labels[frame_number - 1]

    def set_cells_info(self, frame_number, cells_info):
        self._cells_info[frame_number - 1] = cells_info

    def get_cells_info(self, frame_number, type_name=""):
        return self._cells_info[frame_number - 1]

    def set_cell_types(self, frame_number, cell_types):
        self._cell_types[frame_number - 1] = cell_types

    def get_cell_types(self, frame_number):
        return self._cell_types[frame_number - 1]

    def load_labels_from_external_file_array(self, frame, image):
        """ti.py:3467-3472 without the TIFF read: labels = label(image, background=255, connectivity=1)."""
        labels = seg.label(image, background=255, connectivity=1)
        self.set_labels(frame, labels, reset_data=True)
        return labels
