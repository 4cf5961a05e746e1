"""GPU: update_after_adding_segmentation_line / get_new_labels (ti.py:2878-2965) against goldens from the reference's own
methods: a drawn line that splits a cell in two / four / not at all, re-use of a deleted cell's row, the no-table branch."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COLS = ["area", "perimeter", "label", "cx", "cy", "n_neighbors", "valid", "type", "bounding_box_min_row", "bounding_box_min_col",
        "bounding_box_max_row", "bounding_box_max_col", "empty_cell"]


@pytest.mark.parametrize("tag,empty_row,with_table", [("two", None, True), ("three", None, True), ("none", None, True),
                                                       ("reuse", 4, True), ("bare", None, False)])
def test_split_cell_golden(tag, empty_row, with_table):
    from tissue_image_processing_amd import tissue_info as ti
    g = np.load(os.path.join(ROOT, "tests", "golden", "split_cell.npz"))
    t = ti.Tissue(1, "movie", ["zo"])
    base = g["base"].copy()
    t.set_labels(1, base, reset_data=True)
    if with_table:
        t.calculate_frame_cellinfo(1)
        if empty_row is not None:
            info = t.get_cells_info(1)
            info.at[empty_row, "empty_cell"] = 1
            info.at[empty_row, "valid"] = 0
        types = np.full(base.shape, 3, dtype=np.uint8)
        lab_in = g[tag + "_in"]
        types[(base == 0) | ((empty_row is not None) & (base == (empty_row or 0) + 1))] = ti.INVALID_TYPE_INDEX
        t.set_cell_types(1, types)
    base[...] = g[tag + "_in"]                     # the drawn line (and the deleted cell) as the reference's run had them
    rc = t.update_after_adding_segmentation_line(int(g[tag + "_cell"]), 1)
    assert (-1 if rc is None else rc) == int(g[tag + "_rc"])
    np.testing.assert_array_equal(t.get_labels(1), g[tag + "_out"])
    if with_table:
        info = t.get_cells_info(1)
        assert info.shape[0] == g[tag + "_area"].shape[0]
        for col in COLS:
            got = np.asarray(info[col].to_numpy(), dtype=np.float64)
            if col == "perimeter":      # skimage adds the weighted border pixels one by one, the device multiplies counts: 1e-13
                np.testing.assert_allclose(got, g["%s_%s" % (tag, col)], rtol=1e-13, err_msg=col)
            else:
                np.testing.assert_array_equal(got, g["%s_%s" % (tag, col)], err_msg=col)
        nb = g[tag + "_neighbors"]
        for i, s in enumerate(info.neighbors):
            assert sorted(int(v) for v in s) == [int(v) for v in nb[i] if v > 0], i
        np.testing.assert_array_equal(t.get_cell_types(1), g[tag + "_types"])
