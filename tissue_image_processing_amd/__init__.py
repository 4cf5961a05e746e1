"""tissue_image_processing_amd -- MI355X-native (gfx950) hot path of kasirershahartau/tissue_image_processing.

Drop-in modules that keep the reference's Python signatures and run hand-written HIP kernels through a
ctypes C-ABI (include/tissue_hip.h):

    basic_image_manipulations   blur_image, watershed_segmentation, put_channel_axis_first   (bim.py)
    surface_projection          time_point_surface_projection                                 (sp.py)
    tissue_info                 array-heavy Tissue methods                                     (ti.py)
    prediction_local            SegmentationPredictor                                          (pl.py)
"""
__version__ = "0.1.0"
