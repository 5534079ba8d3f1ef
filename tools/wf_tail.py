"""How much of a traversal launch is tail: per bounce, when each of the 4096 persistent waves ended relative
to the launch's span.  Needs the measurement variant (make -C path-tracer_amd/csrc ends) selected with
PTK_LIBRARY=variants/libptk_ends.so."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pta = importlib.import_module("path-tracer_amd")
wd = pta.work_dispatcher
scene = pta.scenes.atrium_scene(262144, seed=7)
rd = pta.PtkRenderingDevice(0)
lib = rd.lib
lib.ptk_dev_wf_ends.restype = C.c_int
lib.ptk_dev_wf_ends.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
rd.set_renderer(pta.rendering_device.PTK_RENDERER_WAVEFRONT)
disp = wd.PTWorkDispatcher(rd)
disp.set_scene(scene)
disp.create_buffers(1920, 1080, order=8)
disp.load_shader()
win = wd.PTRenderWindow(render_width=1920, render_height=1080, max_ray_depth=8, max_refraction_bounces=4)
for k in range(3):
    win.frame = k
    disp.create_compute_list(win, scene.camera, wd.sample_time(k), sync=True)
ITERS, WAVES = 16, 8192
buf = np.zeros(ITERS * WAVES * 2, np.uint64)
assert lib.ptk_dev_wf_ends(rd.ctx, buf.ctypes.data, buf.size) == 0
t = buf.reshape(ITERS, WAVES, 2).astype(np.int64)
print("| bounce | waves | span us | mean wave busy | idle share (tail + start skew) | last 5 % of the span: waves still running |")
print("|---|---|---|---|---|---|")
for it in range(ITERS):
    used = t[it][:, 1] > 0
    if used.sum() < 256:
        continue
    s, e = t[it][used, 0], t[it][used, 1]
    t0, t1 = s.min(), e.max()
    span = (t1 - t0) / 100.0
    if span < 50:
        continue
    busy = (e - s).sum() / (used.sum() * (t1 - t0))
    late = (e > t1 - 0.05 * (t1 - t0)).sum()
    # per workgroup (16 waves): the CU is released when its last wave ends
    wg_end = e.reshape(-1, 16).max(1)
    wg_idle = 1.0 - ((wg_end - t0).sum() / (len(wg_end) * (t1 - t0)))
    print("| %d | %d | %.0f | %.3f | %.1f %% (per workgroup end: %.1f %%) | %d |" % (it, used.sum(), span, busy, 100 * (1 - busy), 100 * wg_idle, late))
    alive = [int((e > t0 + f * (t1 - t0)).sum()) for f in (0.5, 0.6, 0.7, 0.75, 0.8, 0.85, 0.9, 0.95, 0.98)]
    print("    waves alive at 50/60/70/75/80/85/90/95/98 %% of the span: %s" % alive)
