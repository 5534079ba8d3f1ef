import ctypes as C, importlib, os, sys
import numpy as np
ROOT = "/root/repo" if os.path.exists("/root/repo/tools") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT)
pta = importlib.import_module("path-tracer_amd")
wd = pta.work_dispatcher
scene = pta.scenes.load_fixture_scene("dragon_test", os.path.join(ROOT, "tests", "golden"))
rd = pta.PtkRenderingDevice(0)
lib = rd.lib
lib.ptk_dev_tile_costs.restype = C.c_int
lib.ptk_dev_tile_costs.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
rd.set_renderer(pta.rendering_device.PTK_RENDERER_SINGLE_KERNEL)
disp = wd.PTWorkDispatcher(rd); disp.set_scene(scene); disp.create_buffers(1920, 1080, order=8); disp.load_shader()
win = wd.PTRenderWindow(render_width=1920, render_height=1080, max_ray_depth=8, max_refraction_bounces=4)
for f in range(4):
    win.frame = f
    disp.create_compute_list(win, scene.camera, wd.sample_time(f), sync=True)
n = C.c_size_t(0)
lib.ptk_dev_tile_costs(rd.ctx, None, 0, C.byref(n))
full = np.zeros(3 * n.value, np.uint32)
assert lib.ptk_dev_tile_costs(rd.ctx, full.ctypes.data, full.size, C.byref(n)) == 0
cost, hw, wg = full[:n.value].astype(np.int64), full[n.value:2 * n.value], full[2 * n.value:].astype(np.int64)
simd = ((hw >> 16) & 15).astype(np.int64) * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 15) * 4 + ((hw >> 4) & 3)
xcc = (hw >> 16) & 15
by_wg = np.argsort(wg)
print("first 40 workgroups: (wg, xcc, se, sh, cu, simd)")
for t in by_wg[:40]:
    h = hw[t]
    print(int(wg[t]), int((h >> 16) & 15), int((h >> 13) & 7), int((h >> 12) & 1), int((h >> 8) & 15), int((h >> 4) & 3))
first = by_wg[:4096]
ids, inv = np.unique(simd[first], return_inverse=True)
# do workgroups w and w + k share a SIMD for some fixed k?
pos = {}
for rank, t in enumerate(first):
    pos.setdefault(int(simd[t]), []).append(int(wg[t]))
diffs = {}
for v in pos.values():
    v.sort()
    for a, b in zip(v, v[1:]):
        diffs[b - a] = diffs.get(b - a, 0) + 1
print("gaps between the workgroup numbers that share a SIMD among the first 4096:", sorted(diffs.items(), key=lambda kv: -kv[1])[:8])
