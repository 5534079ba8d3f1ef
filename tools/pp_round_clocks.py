"""Where the waves of the persistent path kernel spend their time, by kind of round (development aid).
Needs a measurement build (make -C path-tracer_amd/csrc variant NAME=clocks DEFS="-DPTK_WF_STATS -DPTK_WF_CLOCKS_ONLY")
selected with PTK_LIBRARY=variants/libptk_clocks.so.  usage: python tools/pp_round_clocks.py <fixture scene>"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pta = importlib.import_module("path-tracer_amd")
wd = pta.work_dispatcher
name = sys.argv[1] if len(sys.argv) > 1 else "benchmark_random_scene"
scene = pta.scenes.load_fixture_scene(name, os.path.join(ROOT, "tests", "golden"))
rd = pta.PtkRenderingDevice(0)
lib = rd.lib
lib.ptk_dev_pp_stats.restype = C.c_int
lib.ptk_dev_pp_stats.argtypes = [C.c_void_p, C.c_void_p]
rd.set_renderer(pta.rendering_device.PTK_RENDERER_PERSISTENT)
disp = wd.PTWorkDispatcher(rd)
disp.set_scene(scene)
disp.create_buffers(1920, 1080, order=8)
disp.load_shader(use_emission=(name == "fake_cornell_box"))
win = wd.PTRenderWindow(render_width=1920, render_height=1080, max_ray_depth=8, max_refraction_bounces=8 if name in ("fake_cornell_box", "sphere_scene3") else 4)
st = np.zeros(8, np.uint64)
win.frame = 1
disp.create_compute_list(win, scene.camera, wd.sample_time(1), sync=True)
lib.ptk_dev_pp_stats(rd.ctx, st.ctypes.data)  # clear
disp.create_compute_list(win, scene.camera, wd.sample_time(1), sync=True)
assert lib.ptk_dev_pp_stats(rd.ctx, st.ctypes.data) == 0
tk, tn = st[:4].astype(np.float64), np.maximum(st[4:].astype(np.float64), 1)
print(name, "kernel %.3f ms" % rd.last_dispatch_ms())
for k, label in enumerate(("inner", "leaf", "refill", "bounce")):
    print("%-7s rounds %9d  share of wave time %5.1f %%  ticks per round %7.0f" % (label, tn[k], 100 * tk[k] / tk.sum(), tk[k] / tn[k]))
