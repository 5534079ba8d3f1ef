"""Repeated device builds of one scene, for a kernel trace of the builder (rocprofv3 --kernel-trace --stats)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pta = importlib.import_module("path-tracer_amd")
wd = pta.work_dispatcher
which = sys.argv[1] if len(sys.argv) > 1 else "random"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
if which == "random":
    scene = pta.scenes.random_spheres_scene(grid=32, seed=5)
elif which == "atrium":
    scene = pta.scenes.atrium_scene(262144, seed=7)
else:
    scene = pta.scenes.load_fixture_scene(which, os.path.join(ROOT, "tests", "golden"))
rd = pta.PtkRenderingDevice(0)
b = wd.DeviceBvhBuilder(rd)
for k in range(reps):
    t0 = time.perf_counter()
    tree, meshes = wd.build_scene_tree(scene, b, 8)
    t1 = time.perf_counter()
    b.release(tree, meshes)
    print("build %d: %.3f ms" % (k, (t1 - t0) * 1e3), flush=True)
rd.close()
