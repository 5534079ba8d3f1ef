"""Queue lengths of the wavefront pipeline's launches on the bench scene: rays queued by the shading of the launch
before, searches parked by the traversal of the launch before (development aid)."""
import ctypes as C
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pta = importlib.import_module("path-tracer_amd")
wd = pta.work_dispatcher
scene = pta.scenes.atrium_scene(262144, seed=7)
rd = pta.PtkRenderingDevice(0)
disp = wd.PTWorkDispatcher(rd)
disp.set_scene(scene)
disp.create_buffers(1920, 1080, order=8)
disp.load_shader()
rd.set_renderer(pta.rendering_device.PTK_RENDERER_WAVEFRONT)
win = wd.PTRenderWindow(render_width=1920, render_height=1080, max_ray_depth=8, max_refraction_bounces=4)
for k in range(3):
    win.frame = k
    disp.create_compute_list(win, scene.camera, wd.sample_time(k), sync=True)
out = np.zeros(256, np.uint32)
rd.lib.ptk_dev_wf_counts.argtypes = [C.c_void_p, C.c_void_p]
assert rd.lib.ptk_dev_wf_counts(rd.ctx, out.ctypes.data) == 0
print("queued :", " ".join(str(x) for x in out[0:14]))
print("handed :", " ".join(str(x) for x in out[65:79]))
print("parked :", " ".join(str(x) for x in out[130:144]))
