"""starst3r_amd -- MI355X-native hot path behind the Starst3r Python API.

Mirrors the public names of the reference package (starster/__init__.py:1-9, docs/api.rst):
Scene, reconstruct_scene, gs.*, load_image(s), process_image, prepare_images_for_mast3r, interp_se3(_path),
Mast3rModel.  Importing the package never needs a GPU; the HIP library is loaded on first use and the hot
path raises if it is missing (there is no CPU fallback).
"""
__version__ = "0.4.0+mi355x.1"

try:  # the reference aliases mast3r.model.AsymmetricMASt3R (starster/__init__.py:3); optional here
    from mast3r.model import AsymmetricMASt3R as Mast3rModel  # noqa: F401
except Exception:  # pragma: no cover - mast3r is not vendored by the reference either
    Mast3rModel = None

from . import gs  # noqa: E402,F401
from .image import *  # noqa: E402,F401,F403
from .reconstruct import *  # noqa: E402,F401,F403
from .scene import *  # noqa: E402,F401,F403
from .utils import *  # noqa: E402,F401,F403
