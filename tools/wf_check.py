"""Renders a small frame with the one-kernel renderer and with the wavefront pipeline and compares the images
(a quick check after kernel changes; the parity tests do the same against the CPU oracle)."""
import importlib, os, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pta = importlib.import_module("path-tracer_amd")
wd = pta.work_dispatcher
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
w, h = 256, 144
scene = pta.scenes.atrium_scene(n, seed=7)
crc = {}
for mode in (1, 2):
    rd = pta.PtkRenderingDevice(0)
    rd.set_renderer(mode)
    disp = wd.PTWorkDispatcher(rd)
    disp.set_scene(scene)
    disp.create_buffers(w, h, order=8)
    disp.load_shader()
    win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=4, max_refraction_bounces=2)
    win.frame = 1
    disp.create_compute_list(win, scene.camera, wd.sample_time(1), sync=True)
    crc[mode] = zlib.crc32(bytes(rd.texture_get_data(disp.image_buffer)))
    rd.close()
print("crc single-kernel %08x wavefront %08x" % (crc[1], crc[2]), "SAME" if crc[1] == crc[2] else "DIFFERENT")
