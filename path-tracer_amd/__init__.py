"""MI355X-native render core for the Sr3qt/Path-Tracer plugin's compute dispatch.

The compute path is csrc/libptk.so (HIP, gfx950) behind the C ABI of
include/ptk.h.  The Python modules here are the host-side mirror of the
reference's GDScript classes on either side of that boundary (scene records,
work dispatcher); they carry no renderer of their own.

The directory name contains a hyphen, so import it with
importlib.import_module("path-tracer_amd").
"""
from . import godot_math, scene, scenes, work_dispatcher  # noqa: F401
from .rendering_device import EXPORTS, LIB_PATH, PtkError, PtkRenderingDevice, load_library  # noqa: F401
from .work_dispatcher import PTRenderWindow, PTWorkDispatcher  # noqa: F401
