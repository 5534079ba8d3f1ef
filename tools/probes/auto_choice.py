"""Which schedule AUTO settles on for one of the light configurations, and the kernel time of every dispatch on the way
(development aid).  usage (GPU box): python3 tools/probes/auto_choice.py [fixture name]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pta = importlib.import_module("path-tracer_amd")
wd = pta.work_dispatcher
RD = pta.rendering_device
name = sys.argv[1] if len(sys.argv) > 1 else "dragon_test"
names = {RD.PTK_RENDERER_SINGLE_KERNEL: "tile", RD.PTK_RENDERER_WAVEFRONT: "pipeline", RD.PTK_RENDERER_PERSISTENT: "persistent"}
rd = pta.PtkRenderingDevice(0)
for rep in range(3):
    scene = pta.scenes.load_fixture_scene(name, os.path.join(ROOT, "tests", "golden"))
    disp = wd.PTWorkDispatcher(rd)
    disp.set_scene(scene)
    disp.create_buffers(1920, 1080, order=8)
    disp.load_shader(refraction_mode=2, use_emission=False, early_t_break=False)
    win = wd.PTRenderWindow(render_width=1920, render_height=1080, max_ray_depth=8, max_refraction_bounces=4)
    rd.set_renderer(RD.PTK_RENDERER_AUTO)
    if rep > 0:
        rd.trace_enable(RD.PTK_TRACE_COUNTERS)
        disp.create_compute_list(win, scene.camera, wd.sample_time(2), sync=True)
        rd.trace_enable(0)
    out = []
    for k in range(14):
        win.frame = k
        disp.create_compute_list(win, scene.camera, wd.sample_time(2), sync=True)
        out.append("%s %.3f" % (names[rd.last_dispatch_renderer()], rd.last_dispatch_ms()))
    print(rep, " | ".join(out), flush=True)
    disp.free_rids()
