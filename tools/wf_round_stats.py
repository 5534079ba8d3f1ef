"""Sharing of node blocks among the lanes of the traversal kernel's inner rounds, on the bench scene.

Needs the measurement variant of the library (make -C path-tracer_amd/csrc stats) selected with
PTK_LIBRARY=variants/libptk_stats.so: ptk_wf_traverse then counts, per inner round, the lanes on an
inner node, the distinct blocks they fetch, the size of the groups of lanes on one block, and the depth
of the blocks."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pta = importlib.import_module("path-tracer_amd")
wd = pta.work_dispatcher
w, h = (3840, 2160) if os.environ.get("PTK_4K") else (1920, 1080)
scene = pta.scenes.atrium_scene(262144, seed=7)
rd = pta.PtkRenderingDevice(0)
lib = rd.lib
lib.ptk_dev_wf_stats.restype = C.c_int
lib.ptk_dev_wf_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
rd.set_renderer(pta.rendering_device.PTK_RENDERER_WAVEFRONT)
disp = wd.PTWorkDispatcher(rd)
disp.set_scene(scene)
disp.create_buffers(w, h, order=8)
disp.load_shader()
win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=8, max_refraction_bounces=4)
st = np.zeros(280, np.uint64)
lib.ptk_dev_wf_stats(rd.ctx, st.ctypes.data, 280)  # clear
win.frame = 1
disp.create_compute_list(win, scene.camera, wd.sample_time(1), sync=True)
assert lib.ptk_dev_wf_stats(rd.ctx, st.ctypes.data, 280) == 0
st = st.astype(np.int64)
act, dist, grp = st[0:65], st[65:130], st[130:195]
dv, dd = st[200:232], st[232:264]
rounds = max(act.sum(), 1)
lanes = max((act * np.arange(65)).sum(), 1)
print("inner rounds", rounds, "lane visits", lanes, "mean lanes per round %.1f" % (lanes / rounds))
print("inner rounds with at most 16 / 24 / 32 / 40 / 48 lanes: " + " / ".join("%.1f %%" % (100.0 * act[:k + 1].sum() / rounds) for k in (16, 24, 32, 40, 48)))
print("distinct blocks per round: mean %.1f" % ((dist * np.arange(65)).sum() / rounds), " fetches saved by sharing: %.1f %%" % (100 * (1 - (dist * np.arange(65)).sum() / lanes)))
cum = np.cumsum(grp[::-1])[::-1]
for g in (2, 4, 8, 16, 32):
    print("lane visits in groups of >= %2d lanes on one block: %.1f %%" % (g, 100.0 * cum[g] / lanes))
print("socket lane visits", st[264], " leaf rounds", st[265], "lanes per leaf round %.1f" % (st[266] / max(st[265], 1)))
print("inner rounds: lanes waiting at a leaf %.1f, idle %.1f" % (st[267] / rounds, st[268] / rounds))
print("leaf rounds: lanes waiting at an inner node %.1f, idle %.1f" % (st[269] / max(st[265], 1), st[270] / max(st[265], 1)))
print("refill rounds", st[271])
tk, tn = st[272:275].astype(np.float64), np.maximum(st[276:279].astype(np.float64), 1)
print("wave time by kind of round (shader clock): inner %.1f %%, leaf %.1f %%, refill %.1f %%;  rounds %d / %d / %d;  ticks per round: inner %.0f, leaf %.0f, refill %.0f" % (
    100 * tk[0] / tk.sum(), 100 * tk[1] / tk.sum(), 100 * tk[2] / tk.sum(), tn[0], tn[1], tn[2], tk[0] / tn[0], tk[1] / tn[1], tk[2] / tn[2]))
print("| depth | lane visits | share | cumulative | distinct fetches | lanes per fetch |")
c = 0.0
for d in range(32):
    if dv[d] == 0:
        continue
    c += dv[d] / lanes
    print("| %d | %d | %.1f %% | %.1f %% | %d | %.2f |" % (d, dv[d], 100.0 * dv[d] / lanes, 100 * c, dd[d], dv[d] / dd[d]))
