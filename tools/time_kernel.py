"""Kernel time of the bench workload with whatever libptk PTK_LIBRARY names (development aid)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pta = importlib.import_module("path-tracer_amd")
wd = pta.work_dispatcher
w, h = (3840, 2160) if os.environ.get("PTK_4K") else (1920, 1080)
which = sys.argv[1] if len(sys.argv) > 1 else "atrium"
if which == "atrium":
    scene = pta.scenes.atrium_scene(262144, seed=7)
elif which == "spheres":
    scene = pta.scenes.random_spheres_scene(100, 3)
else:  # a fixture scene by name, e.g. benchmark_random_scene, dragon_test, fake_cornell_box
    scene = pta.scenes.load_fixture_scene(which, os.path.join(ROOT, "tests", "golden"))
rd = pta.PtkRenderingDevice(0)
rd.set_renderer(int(os.environ.get("PTK_RENDERER", "0")))
rd.set_option(pta.rendering_device.PTK_OPT_LAUNCH_TIMING, 1)
disp = wd.PTWorkDispatcher(rd)
disp.set_scene(scene)
bvh_type = int(os.environ["PTK_BVH_TYPE"]) if os.environ.get("PTK_BVH_TYPE") else None  # 5 = binned SAH
if bvh_type:
    for m in scene.meshes:
        m.bvh_type = bvh_type
import time as _t
_t0 = _t.perf_counter()
disp.create_buffers(w, h, order=8, bvh_type=bvh_type)
print("create_buffers (BVH build + bind + uploads) %.1f ms, %d nodes" % ((_t.perf_counter() - _t0) * 1e3, len(disp.packed.bvh) // 32))
disp.load_shader(refraction_mode=2, use_emission=False, early_t_break=False)
win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=8, max_refraction_bounces=4)
for k in range(8):
    win.frame = k
    disp.create_compute_list(win, scene.camera, wd.sample_time(k))
rd.sync()
ms = rd.dispatch_history_ms(6)
tr = rd.dispatch_traverse_ms(0)
print("traverse launches ms", ["%.3f" % x for x in tr], "sum %.3f" % sum(tr))
import time
t0 = time.perf_counter()
img = rd.texture_get_data(disp.image_buffer)
readback_ms = (time.perf_counter() - t0) * 1e3
import zlib
print(os.environ.get("PTK_LIBRARY", "default"), which, "kernel ms median %.3f" % float(np.median(ms)), "crc %08x" % zlib.crc32(bytes(img)), "image read-back %.2f ms" % readback_ms)
