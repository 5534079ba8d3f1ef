"""Lane utilisation of the traversal rounds on the bench scene.

Needs libptk built with -DPTK_ROUND_STATS: the traced kernel then reuses four counter slots that are
zero on the atrium scene (sphere_tests = inner rounds this lane was active in, hit_sphere = rounds the
lane was present for, sockets = leaf rounds present, hit_plane = leaf rounds active).
"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pta = importlib.import_module("path-tracer_amd")
wd = pta.work_dispatcher
w, h = 1920, 1080
scene = pta.scenes.atrium_scene(262144, seed=7)
rd = pta.PtkRenderingDevice(0)
disp = wd.PTWorkDispatcher(rd)
disp.set_scene(scene)
disp.create_buffers(w, h, order=8)
disp.load_shader(refraction_mode=2, use_emission=False, early_t_break=False)
win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=8, max_refraction_bounces=4)
rd.trace_enable(2)
win.frame = 1
disp.create_compute_list(win, scene.camera, wd.sample_time(1), sync=True)
c = rd.trace_read_counters(disp.image_buffer).reshape(h, w, 10).astype(np.int64)
names = pta.rendering_device.COUNTER_NAMES
tot = dict(zip(names, c.reshape(-1, 10).sum(0).tolist()))
print(tot)
tiles = c.reshape(h // 8, 8, w // 8, 8, 10).transpose(0, 2, 1, 3, 4).reshape(-1, 64, 10)
i = {n: k for k, n in enumerate(names)}
present = tiles[:, :, i["hit_sphere"]]
wave_rounds = present.max(1)
leaf_rounds = tiles[:, :, i["sockets"]].max(1)
inner_rounds = wave_rounds - leaf_rounds
inner_active = tiles[:, :, i["sphere_tests"]].sum()
leaf_active = tiles[:, :, i["hit_plane"]].sum()
rays = tot["rays"]
print("waves", len(tiles), "rounds/wave mean", wave_rounds.mean(), "max", wave_rounds.max())
print("inner rounds", inner_rounds.sum(), "lane utilisation", inner_active / (64.0 * inner_rounds.sum()))
print("leaf rounds", leaf_rounds.sum(), "lane utilisation", leaf_active / (64.0 * leaf_rounds.sum()))
print("per ray: inner steps", inner_active / rays, "leaf steps", leaf_active / rays,
      "box tests", tot["box_tests"] / rays, "tri tests", tot["tri_tests"] / rays)
print("children per inner step", tot["box_tests"] / max(inner_active, 1), "tris per leaf", tot["tri_tests"] / max(leaf_active, 1))
wave_ray_slots = tiles[:, :, i["rays"]].max(1).sum() * 64
print("path utilisation (rays / (64 * max rays in wave))", rays / wave_ray_slots)
