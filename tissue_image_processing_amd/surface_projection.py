"""Drop-in for the reference's surface_projection.py on MI355X: time_point_surface_projection (sp.py:17-85) and the
movie / large-image drivers around it (sp.py:168-316).

Signature, defaults, return types and error behaviour follow the reference; the arithmetic runs in
libtissue_hip.so (tip_project_u16 / tip_project_u16_binned).  Covered: bin_size == 1 (what every BASELINE config and
movie_surface_projection's default use) and bin_size > 1 with methods 'max_averages', 'max_std', 'multi_channel'
(sp.py:39-65), and build_manifold (the spiral of sp.py:87-165, as a scan of function tables on the device), also on the binned score.
"""

_METHODS = {"max_averages": 0, "max_std": 1, "multi_channel": 2}
_MANIFOLD = 16                 # TIP_PROJECT_MANIFOLD (include/tissue_hip.h)
import ctypes

import numpy as np

from . import _lib
from .basic_image_manipulations import put_channel_axis_first, gaussian_taps


def time_point_surface_projection(time_point, axes, reference_channel, min_z=0, max_z=0,
                                  method='max_averages', bin_size=1, airyscan=True, z_map=False, atoh_shift=0,
                                  build_manifold=False):
    if bin_size > 1 and method not in ("max_averages", "max_std", "multi_channel"):
        raise TypeError("exceptions must derive from BaseException")  # sp.py:53 raises a str
    if bin_size > 128:
        raise NotImplementedError("MI355X path covers bin_size <= 128")
    if axes.find("T") >= 0:
        time_point = time_point.reshape(time_point.shape[1:])
        image, _ = put_channel_axis_first(time_point, axes[1:])
    else:
        image, _ = put_channel_axis_first(time_point, axes)
    if axes.find("C") < 0 or image.ndim != 4:
        # sp.py:32 indexes image[reference_channel] on a (Z,Y,X) array and then blurs the 2-D slice with a
        # 3-tuple sigma -> scipy's RuntimeError (golden: tests/golden/projection.npz c_error)
        raise RuntimeError("sequence argument must have length equal to input rank")
    image = np.asarray(image)
    if image.dtype != np.uint16:
        # upstream casts anything to float32 (sp.py:26); the device path reads uint16 voxels, so other dtypes are taken when
        # they hold exactly such values (integer types in range, float stacks of whole numbers -- a uint16 movie that went
        # through a float conversion), which float32 represents exactly: same arithmetic from there on
        whole = image.size > 0 and (np.issubdtype(image.dtype, np.integer) or
                                    (np.issubdtype(image.dtype, np.floating) and bool(np.all(np.floor(image) == image))))
        if whole and image.min() >= 0 and image.max() <= 65535:
            image = image.astype(np.uint16)
        else:
            raise TypeError("MI355X projection takes stacks of uint16 values (microscope data); got %s with fractional, "
                            "negative or larger values" % image.dtype)
    image = np.ascontiguousarray(image)
    C, Z, Y, X = image.shape
    if not (-C <= reference_channel < C):
        raise IndexError("index %d is out of bounds for axis 0 with size %d" % (reference_channel, C))
    reference_channel %= C
    zlo, zhi = (min_z, min(max_z, Z)) if max_z > 0 else (0, Z)
    if zlo < 0:
        zlo = max(0, Z + zlo)
    if zhi <= zlo:
        raise ValueError("attempt to get argmax of an empty sequence")  # numpy's error for an empty z slice
    t05, t1, t2, t30 = gaussian_taps(0.5), gaussian_taps(1.0), gaussian_taps(2.0), gaussian_taps(30.0)
    proj = np.empty((C, Y, X), np.float64)
    zmap = np.empty((Y, X), np.int64)
    lib = _lib.lib()
    if bin_size > 1 or build_manifold:
        rc = lib.tip_project_u16_binned(_lib.ptr(image), C, Z, Y, X, int(zlo), int(zhi), int(min_z), int(reference_channel),
                                        (_METHODS[method] if bin_size > 1 else 0) | (_MANIFOLD if build_manifold else 0), int(bin_size), 1 if airyscan else 0, int(atoh_shift),
                                        _lib.ptr(t05), _lib.ptr(t1), _lib.ptr(t2), _lib.ptr(t30), _lib.ptr(proj),
                                        _lib.ptr(zmap))
    else:
        rc = lib.tip_project_u16(_lib.ptr(image), C, Z, Y, X, int(zlo), int(zhi), int(min_z), int(reference_channel),
                                 1 if airyscan else 0, int(atoh_shift), _lib.ptr(t05), _lib.ptr(t1), _lib.ptr(t2),
                                 _lib.ptr(t30), _lib.ptr(proj), _lib.ptr(zmap))
    _lib.check(rc)
    if z_map:
        return proj, zmap
    return proj


def build_continues_manifold(score):
    """sp.py:87-165: the z-map grown as a square spiral around the score's global maximum (int64, like upstream's
    astype(int)); on the device a scan of per-pixel function tables along every straight run of a ring (tip_manifold.h)."""
    s = np.ascontiguousarray(score, dtype=np.float32)
    if s.ndim != 3:
        raise ValueError("build_continues_manifold takes a (Z, Y, X) score")
    out = np.empty(s.shape[1:], np.int64)
    _lib.check(_lib.lib().tip_build_manifold_f32(_lib.ptr(s), s.shape[0], s.shape[1], s.shape[2], _lib.ptr(out)))
    return out


# ---- drivers (sp.py:168-316): whole movies / large images through the per-time-point projection -----------------------------
def get_image_metadata(path, series=0):
    """bim.py:85-88: the OME metadata object of an aicsimageio source; None for arrays / .npy / TIFF sources."""
    from .basic_image_manipulations import open_image
    src = open_image(path, series)
    img = getattr(src, "img", None)
    return getattr(img, "metadata", None)


def update_projection_metadata(metadata, frames_number, series=0):
    """sp.py:319-327: the projected movie's OME description (one scene, no z axis, uint16)."""
    if metadata is None:
        return None
    im = metadata.images[series]
    metadata.images = [im]
    im.name = 'position%d' % series
    px = im.pixels
    px.dimension_order = 'XYCTZ'
    px.size_z = 1
    px.size_t = frames_number
    px.type = 'uint16'
    px.planes = px.planes[:px.size_c]
    return metadata


def _barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def _project_scene(source, series, out_shape_t, rank, world, part_prefix, **params):
    """Projection (T, C, 1, Y, X) and z-map (T, 1, 1, Y, X) of one scene, one time point per read_image_in_chunks step.
    world > 1: time point t is computed by rank t % world (no data-path collective: every rank stores its time points
    as .npy parts next to the outputs, rank 0 assembles them after a barrier)."""
    import os
    from .basic_image_manipulations import open_image, read_image_in_chunks
    src = open_image(source, series)
    T, C, Z, Y, X = src.dims
    proj = np.zeros((T, C, 1, Y, X))
    zmap = np.zeros((T, 1, 1, Y, X))
    if world == 1:
        steps = read_image_in_chunks(src, series=series, apply_function=time_point_surface_projection,
                                     output=[proj, zmap], axes='TCZYX', z_map=True, **params)
        for _ in steps:
            pass
        return proj, zmap
    dx, dy = params.pop("dx", 0), params.pop("dy", 0)
    params.pop("dt", None)
    for t in range(rank, T, world):
        one = src.block(slice(t, t + 1), slice(0, C), slice(0, Z), slice(0, Y), slice(0, X))
        p1, z1 = np.zeros((1, C, 1, Y, X)), np.zeros((1, 1, 1, Y, X))
        for _ in read_image_in_chunks(one, dt=1, dx=dx, dy=dy, apply_function=time_point_surface_projection,
                                      output=[p1, z1], axes='TCZYX', z_map=True, **params):
            pass
        np.save("%s.t%06d.proj.npy" % (part_prefix, t), p1)
        np.save("%s.t%06d.zmap.npy" % (part_prefix, t), z1)
    _barrier(world)
    if rank == 0:
        for t in range(T):
            for arr, kind in ((proj, "proj"), (zmap, "zmap")):
                part = "%s.t%06d.%s.npy" % (part_prefix, t, kind)
                arr[t] = np.load(part)[0]
                os.remove(part)
    _barrier(world)
    return proj, zmap


def movie_surface_projection(files, reference_channel, position_final_movie, initial_positions_number, output_dir,
                             method, bin_size, build_manifold, only_position, zmin, zmax, airyscan, output_name="",
                             rank=0, world=1):
    """sp.py:168-236: project every time point of every position of a movie split over several files (image sources of
    basic_image_manipulations.open_image; scene = position) and write, per position, `positionN.tif` (uint16, TCYX),
    `zmap_positionN.npy` and -- when the sources carry OME stage metadata -- `stage_locations_positionN.pkl`.
    A position leaves the scene list after its final movie (position_final_movie, 1-based file numbers), which shifts the
    scene numbers of the ones behind it, as upstream.  rank / world: shard the time points over processes (one per GPU);
    rank 0 writes the outputs."""
    import os
    from .basic_image_manipulations import get_image_dimensions, concatenate_time_points, save_tiff
    positions = list(range(initial_positions_number))
    time_points_number = np.zeros((initial_positions_number, len(files)))
    projection_files = [[] for _ in range(initial_positions_number)]
    zmap_files = [[] for _ in range(initial_positions_number)]
    wanted = lambda position: only_position <= 0 or position == only_position - 1
    for file_num, file in enumerate(files):
        leaving = []
        dims = get_image_dimensions(file)
        for position_num, position in enumerate(positions):
            if position_final_movie[position] == file_num + 1:
                leaving.append(position)
            if not wanted(position):
                continue
            projection_path = os.path.join(output_dir, "position%d_movie%d_projection.npy" % (position, file_num))
            zmap_path = os.path.join(output_dir, "position%d_movie%d_zmap.npy" % (position, file_num))
            projection_files[position].append(projection_path)
            zmap_files[position].append(zmap_path)
            time_points_number[position, file_num] = dims.T
            if os.path.isfile(projection_path) and os.path.isfile(zmap_path):
                continue
            if reference_channel >= dims.C:
                reference_channel = dims.C - 1          # (sticks for the files that follow, as upstream)
            proj, zmap = _project_scene(file, position_num, dims.T, rank, world, projection_path,
                                        reference_channel=reference_channel, method=method, bin_size=bin_size, atoh_shift=0,
                                        build_manifold=build_manifold, min_z=zmin, max_z=zmax, airyscan=airyscan, dt=1)
            if rank == 0:
                np.save(projection_path, proj.reshape((dims.T, dims.C, dims.Y, dims.X)))
                np.save(zmap_path, zmap)
            _barrier(world)
        for position in leaving:
            positions.remove(position)
    if rank == 0:
        for position in range(initial_positions_number):
            if not wanted(position):
                continue
            metadata = update_projection_metadata(get_image_metadata(files[0], series=position),
                                                  np.sum(time_points_number[position, :]), series=position)
            movie = concatenate_time_points(projection_files[position])
            save_tiff(os.path.join(output_dir, output_name + "position%d.tif" % (position + 1)), movie, metadata=metadata,
                      axes="TCYX", data_type="uint16")
            zmaps = np.concatenate([np.load(f).astype("uint16") for f in zmap_files[position]], axis=0)
            np.save(os.path.join(output_dir, output_name + "zmap_position%d.npy" % (position + 1)), zmaps)
        save_stage_positions(files, position_final_movie, initial_positions_number, output_dir, only_position=only_position,
                             output_name=output_name)
        for f in [f for group in projection_files + zmap_files for f in group]:
            os.remove(f)
    _barrier(world)


def save_stage_positions(files, position_final_movie, initial_positions_number, output_dir, only_position=0, output_name=""):
    """sp.py:239-280: per position, the stage coordinates of every time point (one entry per frame of every file the
    position appears in) as `stage_locations_positionN.pkl`.  Needs the sources' OME metadata; sources without it
    (arrays, .npy, TIFF stacks) have no stage record and nothing is written."""
    import os
    import pickle
    meta = get_image_metadata(files[0])
    if meta is None:
        return
    stage_pos = []
    for i in range(initial_positions_number):
        im = meta.images[i]
        n = im.pixels.size_t
        stage_pos.append({"x": [im.stage_label.x] * n, "y": [im.stage_label.y] * n, "z": [im.stage_label.z] * n,
                          "x_unit": im.stage_label.x_unit, "y_unit": im.stage_label.y_unit, "z_unit": im.stage_label.z_unit,
                          "physical_size_x": im.pixels.physical_size_x, "physical_size_y": im.pixels.physical_size_y,
                          "physical_size_z": im.pixels.physical_size_z})
    positions = [p for p in range(initial_positions_number) if position_final_movie[p] != 1]
    for file_index in range(1, len(files)):
        meta = get_image_metadata(files[file_index])
        leaving = []
        for scene, position in enumerate(positions):
            if position_final_movie[position] == file_index + 1:
                leaving.append(position)
            if only_position > 0 and position != only_position - 1:
                continue
            im = meta.images[scene]
            for axis in "xyz":
                stage_pos[position][axis].extend([getattr(im.stage_label, axis)] * im.pixels.size_t)
        for position in leaving:
            positions.remove(position)
    for i in range(initial_positions_number):
        if only_position > 0 and i != only_position - 1:
            continue
        with open(os.path.join(output_dir, output_name + "stage_locations_position%d.pkl" % (i + 1)), 'wb') as f:
            pickle.dump(stage_pos[i], f)


def large_image_projection(input_dir, output_dir, input_file_name, position=1, reference_channel=0, chunk_size=0,
                           bin_size=1, channels_shift=0, min_z=0, max_z=0, method="", build_manifold=False,
                           airyscan=False, rank=0, world=1):
    """sp.py:283-316: project a large (tiled-scan) image in independent chunk_size x chunk_size blocks -- upstream's
    halo-less tiler: every block is projected on its own -- and write `<name>[_positionN]_projection.tif` (uint16) and
    `<name>[_positionN]_zmap.npy`.  Returns 0 when the input does not exist, as upstream."""
    import os
    from .basic_image_manipulations import get_image_dimensions, save_tiff
    several = hasattr(position, "__len__")
    path = os.path.join(input_dir, input_file_name)
    if not os.path.exists(path):
        return 0
    dims = get_image_dimensions(path)
    for pos in (position if several else [position]):
        proj, zmap = _project_scene(path, int(pos - 1), dims.T, rank, world, os.path.join(output_dir, input_file_name + ".part%d" % pos),
                                    dx=chunk_size, dy=chunk_size, dt=1, min_z=min_z, max_z=max_z,
                                    reference_channel=reference_channel, method=method, bin_size=bin_size,
                                    atoh_shift=channels_shift, build_manifold=build_manifold, airyscan=airyscan)
        if rank != 0:
            continue
        proj = proj.reshape((dims.T, dims.C, dims.Y, dims.X) if dims.T > 1 else (dims.C, dims.Y, dims.X))
        zmap = zmap.reshape((dims.T, dims.Y, dims.X))
        postfix = '.' + input_file_name.split('.')[-1]
        tag = "_position%d" % pos if several else ""
        save_tiff(os.path.join(output_dir, input_file_name.replace(postfix, tag + "_projection.tif")), proj,
                  axes="TCYX" if dims.T > 1 else "CYX", data_type="uint16")
        np.save(os.path.join(output_dir, input_file_name.replace(postfix, tag + "_zmap.npy")), zmap)
