"""`import starster` drop-in: the reference package name (starster/__init__.py:1-9 of phuang1024/Starst3r) bound to the
MI355X-native implementation in `starst3r_amd`.  `starster.Scene`, `starster.reconstruct_scene`, `starster.gs.*`,
`starster.load_image(s)`, `starster.process_image`, `starster.interp_se3(_path)` and `starster.Mast3rModel` are the very
objects of `starst3r_amd`; the submodules `starster.gs / image / reconstruct / scene / utils` resolve to its modules, so
`from starster.gs import run_3dgs_optim` and `import starster.scene` work as with the reference."""
import sys as _sys

import starst3r_amd as _impl
from starst3r_amd import *  # noqa: F401,F403
from starst3r_amd import Mast3rModel, __version__, gs, image, reconstruct, scene, utils  # noqa: F401

# `reconstruct` is a submodule in the reference (its `from .reconstruct import *` leaves the MODULE bound under that
# name) and BASELINE.json's north_star also writes `reconstruct()` for `reconstruct_scene`: the module object is made
# callable so that both spellings work
import types as _types

_reconstruct_mod = _sys.modules["starst3r_amd.reconstruct"]


class _CallableModule(_types.ModuleType):
    def __call__(self, *args, **kwargs):
        return self.reconstruct_scene(*args, **kwargs)


_reconstruct_mod.__class__ = _CallableModule
reconstruct = _reconstruct_mod
for _name in ("gs", "image", "scene", "utils"):
    _sys.modules[__name__ + "." + _name] = getattr(_impl, _name)
_sys.modules[__name__ + ".reconstruct"] = _reconstruct_mod
del _name
