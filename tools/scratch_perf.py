"""Scratch timing of the render kernel on the atrium scene (BVH from the oracle builder; not a bench)."""
import importlib, sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob
pta = importlib.import_module("path-tracer_amd")
wd = pta.work_dispatcher
which = sys.argv[1] if len(sys.argv) > 1 else "atrium"
w, h = 1920, 1080
if which == "atrium":
    scene = pta.scenes.atrium_scene(262144)
else:
    scene = pta.scenes.load_fixture_scene(which, os.path.join(ROOT, "tests", "golden"))
packed = wd.pack_scene(scene, ob.OracleBvhBuilder(), 8)
rd = pta.PtkRenderingDevice(0)
disp = wd.PTWorkDispatcher(rd)
disp.create_buffers(w, h, packed=packed)
disp.load_shader()
win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=8, max_refraction_bounces=4)
rd.trace_enable(2)
win.frame = 1
disp.create_compute_list(win, scene.camera, wd.sample_time(1), sync=True)
cnt = rd.trace_read_counters(disp.image_buffer).reshape(-1, 10).sum(0)
print("traced kernel ms", rd.last_dispatch_ms(), dict(zip(ob.STAT_NAMES, cnt.tolist())))
rd.trace_enable(0)
times = []
for k in range(6):
    win.frame = 1
    disp.create_compute_list(win, scene.camera, wd.sample_time(1), sync=True)
    times.append(rd.last_dispatch_ms())
ms = float(np.median(times))
bytes_ = ob.algorithmic_bytes(cnt, len(packed.planes) // 32, w * h)
print(which, "kernel ms", times, "Mrays/s %.1f" % (cnt[0] / ms / 1e3), "alg GB/s %.1f" % (bytes_ / ms / 1e6), "bytes/ray %.0f" % (bytes_ / cnt[0]))
