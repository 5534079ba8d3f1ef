"""Do two dispatches on two streams overlap their launches' tails?  Two contexts render the same workload; timed:
one context alone, then both with their dispatches enqueued alternately (development aid).
usage (GPU box): python3 tools/probes/two_streams.py [steps]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
pta = importlib.import_module("path-tracer_amd")
wd = pta.work_dispatcher
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
W, H = 1920, 1080
scene = pta.scenes.atrium_scene(262144, seed=7)

def make():
    rd = pta.PtkRenderingDevice(0)
    d = wd.PTWorkDispatcher(rd)
    d.set_scene(scene)
    d.create_buffers(W, H, order=8)
    d.load_shader(refraction_mode=2, use_emission=False, early_t_break=False)
    rd.set_renderer(pta.rendering_device.PTK_RENDERER_WAVEFRONT)
    win = wd.PTRenderWindow(render_width=W, render_height=H, max_ray_depth=8, max_refraction_bounces=4, max_samples=steps)
    return rd, d, win

a, b, c, e = make(), make(), make(), make()

def run(ctxs, n):
    for k in range(n):
        for rd, d, win in ctxs:
            win.frame = k
            d.create_compute_list(win, scene.camera, wd.sample_time(k))
    for rd, d, win in ctxs:
        rd.sync()

for ctxs, name in (([a], "one"), ([b], "other"), ([a, b], "both"), ([a, b, c], "three"), ([a, b, c, e], "four"), ([a], "one"), ([a, b], "both"), ([a, b, c], "three"), ([a, b, c, e], "four")):
    run(ctxs, 2)
    torch.cuda.synchronize()
    t = time.perf_counter()
    run(ctxs, steps)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) * 1e3 / (steps * len(ctxs))
    print("%-6s %.3f ms per dispatch" % (name, ms), flush=True)
