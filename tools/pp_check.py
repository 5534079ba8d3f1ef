"""CRC of the image from the persistent renderer against the one-kernel renderer on fixture scenes, and kernel
times of all three renderers (development aid)."""
import importlib, os, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pta = importlib.import_module("path-tracer_amd")
wd = pta.work_dispatcher
GOLD = os.path.join(ROOT, "tests", "golden")
cases = [("sphere_scene1", 640, 360, 4, 4, False), ("benchmark_random_scene", 1920, 1080, 8, 4, False),
         ("fake_cornell_box", 1920, 1080, 8, 8, True), ("sphere_scene3", 1920, 1080, 8, 8, False),
         ("dragon_test", 1920, 1080, 8, 4, False), ("atrium", 1920, 1080, 8, 4, False)]
only = sys.argv[1:] 
for name, w, h, depth, bounces, emission in cases:
    if only and name not in only:
        continue
    scene = pta.scenes.atrium_scene(262144, seed=7) if name == "atrium" else pta.scenes.load_fixture_scene(name, GOLD)
    rd = pta.PtkRenderingDevice(0)
    disp = wd.PTWorkDispatcher(rd)
    disp.set_scene(scene)
    disp.create_buffers(w, h, order=8)
    disp.load_shader(refraction_mode=2, use_emission=emission, early_t_break=False)
    win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=depth, max_refraction_bounces=bounces)
    out = []
    for r in (1, 2, 3):
        rd.set_renderer(r)
        rd.texture_update(disp.image_buffer, np.zeros((h, w, 4), np.float32))
        for k in range(6):
            win.frame = k
            disp.create_compute_list(win, scene.camera, wd.sample_time(k))
        rd.sync()
        ms = float(np.median(rd.dispatch_history_ms(4)))
        out.append((ms, zlib.crc32(bytes(rd.texture_get_data(disp.image_buffer)))))
    print(name, " ".join("%.3f ms %08x" % o for o in out), "SAME" if len({o[1] for o in out}) == 1 else "DIFFERENT", flush=True)
    disp.free_rids()
    rd.close()
