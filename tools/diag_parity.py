"""Diagnostic: print where GPU and oracle differ for one fixture scene (run on the GPU box)."""
import importlib, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob
pta = importlib.import_module("path-tracer_amd")
wd = pta.work_dispatcher
name = sys.argv[1] if len(sys.argv) > 1 else "benchmark_random_scene"
w, h, depth = 640, 360, 8
scene = pta.scenes.load_fixture_scene(name, os.path.join(ROOT, "tests", "golden"))
packed = wd.pack_scene(scene, ob.OracleBvhBuilder(), 8)
rd = pta.PtkRenderingDevice(0)
disp = wd.PTWorkDispatcher(rd)
disp.create_buffers(w, h, packed=packed)
disp.load_shader()
rd.trace_enable(3)
win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=depth, max_refraction_bounces=4)
cpu = np.zeros((h, w, 4), np.float32)
t = wd.sample_time(0)
out = ob.render(packed, wd.push_constant_bytes(scene.camera, win, t), cpu, w // 8, h // 8)
disp.create_compute_list(win, scene.camera, t, sync=True)
gpu = rd.texture_get_data(disp.image_buffer)
gc = rd.trace_read_counters(disp.image_buffer)
cc = out["pixel_stats"]
bad = np.argwhere((gc != cc).any(-1))
print("mismatching pixels:", len(bad))
for (y, x) in bad[:12]:
    print((x, y), "gpu", gc[y, x].tolist(), "cpu", cc[y, x].tolist(), "img gpu", gpu[y, x].tolist(), "cpu", cpu[y, x].tolist())
