"""Wall time of the device BVH build for the bench scene (scene-level tree + one 262 144-triangle mesh)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pta = importlib.import_module("path-tracer_amd")
wd = pta.work_dispatcher
scene = pta.scenes.atrium_scene(262144, seed=7)
rd = pta.PtkRenderingDevice(0)
builder = wd.DeviceBvhBuilder(rd)
for k in range(4):
    t0 = time.perf_counter()
    nodes, ids = wd.create_scene_bvh(scene, builder, 8)
    print("build %d: %.1f ms (%d nodes)" % (k, (time.perf_counter() - t0) * 1e3, len(nodes) // 32))
