"""Host time spent inside one asynchronous dispatch call, per renderer (development aid)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pta = importlib.import_module("path-tracer_amd")
wd = pta.work_dispatcher
scene = pta.scenes.load_fixture_scene("sphere_scene1", os.path.join(ROOT, "tests", "golden"))
rd = pta.PtkRenderingDevice(0)
disp = wd.PTWorkDispatcher(rd); disp.set_scene(scene); disp.create_buffers(640, 360, order=8); disp.load_shader()
win = wd.PTRenderWindow(render_width=640, render_height=360, max_ray_depth=4, max_refraction_bounces=4)
push = wd.push_constant_bytes(scene.camera, win, 0.01)
for r in (1, 2, 3):
    rd.set_renderer(r)
    for _ in range(5):
        rd.compute_list_dispatch(push, 80, 45)
    rd.sync()
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        rd.compute_list_dispatch(push, 80, 45)
    t1 = time.perf_counter()
    rd.sync()
    t2 = time.perf_counter()
    print("renderer %d: %.1f us per dispatch call on the host, %.1f us per dispatch end to end" % (r, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
