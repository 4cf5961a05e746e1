"""Drop-in module: put this directory ahead of the reference's on sys.path and `import prediction_local` (as the reference's
gui.py / surface_projection.py do by bare module name) resolves to the MI355X implementation."""
from tissue_image_processing_amd.prediction_local import *  # noqa: F401,F403
from tissue_image_processing_amd import prediction_local as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
