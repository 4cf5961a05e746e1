"""Drop-in module: put this directory ahead of the reference's on sys.path and `import surface_projection` (as the reference's
gui.py / surface_projection.py do by bare module name) resolves to the MI355X implementation."""
from tissue_image_processing_amd.surface_projection import *  # noqa: F401,F403
from tissue_image_processing_amd import surface_projection as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
