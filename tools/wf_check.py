"""Runs a small wavefront dispatch on an index-checked libptk (built with -DPTK_CHECKED, named by PTK_LIBRARY)
and prints the error words the traversal kernel left, plus a comparison with the one-kernel renderer."""
import ctypes as C, importlib, os, sys, zlib
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
    if mode == 2 and hasattr(rd.lib, "ptk_debug_wf_check"):
        out = (C.c_uint * 4)()
        rd.lib.ptk_debug_wf_check.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
        rc = rd.lib.ptk_debug_wf_check(rd.ctx, out)
        print("check rc", rc, "error flags", out[0], "cur %08x" % out[1], "sp", out[2], "leaf cur %08x" % out[3])
    crc[mode] = zlib.crc32(bytes(rd.texture_get_data(disp.image_buffer)))
    rd.close()
print("crc single-kernel %08x wavefront %08x" % (crc[1], crc[2]), "SAME" if crc[1] == crc[2] else "DIFFERENT")
