"""Wall time of scene set-up stages on the GPU box: BVH build, read-back, buffer creation, first dispatch
(validation + derived tables), for the fixture scenes and the bench scene (development aid)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pta = importlib.import_module("path-tracer_amd")
wd = pta.work_dispatcher
GOLD = os.path.join(ROOT, "tests", "golden")
cases = [("sphere_scene1", None), ("benchmark_random_scene", None), ("dragon_test", None), ("atrium", 262144)]
rd = pta.PtkRenderingDevice(0)
for name, n in cases:
    scene = pta.scenes.atrium_scene(n, seed=7) if name == "atrium" else pta.scenes.load_fixture_scene(name, GOLD)
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        tree, meshes = wd.build_scene_tree(scene, wd.DeviceBvhBuilder(rd), 8)
        rd.sync()
        t1 = time.perf_counter()
        bound = rd.bvh_bind(tree)
        tb = time.perf_counter()
        wd.DeviceBvhBuilder(rd).release(tree, meshes)
        rd.free_rid(bound[0]); rd.free_rid(bound[1])
        disp = wd.PTWorkDispatcher(rd)
        disp.set_scene(scene)
        t1b = time.perf_counter()
        disp.create_buffers(256, 144)
        disp.load_shader()
        t2 = time.perf_counter()
        win = wd.PTRenderWindow(render_width=256, render_height=144, max_ray_depth=2)
        disp.create_compute_list(win, scene.camera, 0.0, sync=True)
        t3 = time.perf_counter()
        disp.free_rids()
        cur = (t1 - t0, tb - t1, t2 - t1b, t3 - t2)
        best = cur if best is None else tuple(min(a, b) for a, b in zip(best, cur))
    print("%s: GPU BVH build (incl. host box packing) %.2f ms, bind %.2f ms | create_buffers (build + bind + read-back + all uploads) %.2f ms, first dispatch (validate + derive) %.2f ms" % ((name,) + tuple(x * 1e3 for x in best)))
