"""Run one fixture scene with one renderer pinned for a number of dispatches and print the per-dispatch kernel
time and the traversal launches' durations (development aid; meant to be run under rocprofv3 --kernel-trace
--stats as well: `rocprofv3 --kernel-trace --stats -d out -- python3 tools/scene_profile.py fake_cornell_box 2`)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pta = importlib.import_module("path-tracer_amd")
wd = pta.work_dispatcher
RD = pta.rendering_device
GOLD = os.path.join(ROOT, "tests", "golden")
name = sys.argv[1] if len(sys.argv) > 1 else "fake_cornell_box"
renderer = int(sys.argv[2]) if len(sys.argv) > 2 else RD.PTK_RENDERER_WAVEFRONT
w, h = (640, 360) if name == "sphere_scene1" else (1920, 1080)
depth, bounces = (4, 4) if name == "sphere_scene1" else (8, 8 if name in ("fake_cornell_box", "sphere_scene3") else 4)
scene = pta.scenes.load_fixture_scene(name, GOLD)
rd = pta.PtkRenderingDevice(0)
disp = wd.PTWorkDispatcher(rd)
disp.set_scene(scene)
disp.create_buffers(w, h, order=8)
disp.load_shader(refraction_mode=2, use_emission=(name == "fake_cornell_box"), early_t_break=False)
rd.set_renderer(renderer)
rd.set_option(pta.rendering_device.PTK_OPT_LAUNCH_TIMING, 1)
win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=depth, max_refraction_bounces=bounces)
n = 20
for k in range(n + 2):
    win.frame = k
    disp.create_compute_list(win, scene.camera, wd.sample_time(2), sync=True)
print(name, "renderer", renderer, "median ms per dispatch %.3f" % float(np.median(rd.dispatch_history_ms(n))))
if renderer == RD.PTK_RENDERER_WAVEFRONT:
    print("traverse launches ms:", " ".join("%.3f" % x for x in rd.dispatch_traverse_ms(0)))
