"""Kernel time per sample of the BASELINE configurations' scenes with each renderer pinned, and the rays per
sample from a traced dispatch (development aid; prints a markdown table)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pta = importlib.import_module("path-tracer_amd")
wd = pta.work_dispatcher
RD = pta.rendering_device
GOLD = os.path.join(ROOT, "tests", "golden")
CASES = [  # label, scene, width, height, depth, refraction bounces, use_emission
    ("C1 sphere_scene1", "sphere_scene1", 640, 360, 4, 4, False),
    ("C2 benchmark_random_scene", "benchmark_random_scene", 1920, 1080, 8, 4, False),
    ("C3 fake_cornell_box (emission)", "fake_cornell_box", 1920, 1080, 8, 8, True),
    ("C3 sphere_scene3", "sphere_scene3", 1920, 1080, 8, 8, False),
    ("C4 dragon_test", "dragon_test", 1920, 1080, 8, 4, False),
    ("C5 atrium 262144 tris, 1080p", "atrium", 1920, 1080, 8, 4, False),
    ("C5 atrium 262144 tris, 4K", "atrium", 3840, 2160, 8, 4, False),
]
print("| configuration | rays per sample | tile kernel ms | pipeline ms | persistent kernel ms | Mrays/s (fastest) |")
print("|---|---|---|---|---|---|")
atrium = None
for label, name, w, h, depth, bounces, emission in CASES:
    if name == "atrium":
        atrium = atrium or pta.scenes.atrium_scene(262144, seed=7)
        scene = atrium
    else:
        scene = pta.scenes.load_fixture_scene(name, GOLD)
    rd = pta.PtkRenderingDevice(0)
    disp = wd.PTWorkDispatcher(rd)
    disp.set_scene(scene)
    disp.create_buffers(w, h, order=8)
    disp.load_shader(refraction_mode=2, use_emission=emission, early_t_break=False)
    win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=depth, max_refraction_bounces=bounces)
    rd.trace_enable(RD.PTK_TRACE_COUNTERS)
    win.frame = 2
    disp.create_compute_list(win, scene.camera, wd.sample_time(2), sync=True)
    rays = int(rd.trace_read_counters(disp.image_buffer).reshape(-1, RD.PTK_NCOUNTERS).sum(0)[RD.COUNTER_NAMES.index("rays")])
    rd.trace_enable(0)
    ms = {}
    for r in (RD.PTK_RENDERER_SINGLE_KERNEL, RD.PTK_RENDERER_WAVEFRONT, RD.PTK_RENDERER_PERSISTENT):
        rd.set_renderer(r)
        for k in range(8):
            win.frame = k
            disp.create_compute_list(win, scene.camera, wd.sample_time(2))
        rd.sync()
        ms[r] = float(np.median(rd.dispatch_history_ms(6)))
    best = min(ms.values())
    print("| %s, %dx%d, depth %d | %.2f M | %.3f | %.3f | %.3f | %.0f |" % (label, w, h, depth, rays / 1e6, ms[RD.PTK_RENDERER_SINGLE_KERNEL],
                                                                        ms[RD.PTK_RENDERER_WAVEFRONT], ms[RD.PTK_RENDERER_PERSISTENT], rays / best / 1e3))
    disp.free_rids()
    rd.close()
