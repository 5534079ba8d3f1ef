"""Writes tests/golden/c_harness_sphere_scene1.bin: the byte buffers PTWorkDispatcher.create_buffers would
upload for sphere_scene1 (BVH from the CPU oracle's builder), the pipeline switches and one push constant
block, in the container tests/c_abi/c_abi_harness.c reads.  Data only; run once, commit the result."""
import importlib
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

W, H, DEPTH = 320, 184, 4


def write_fixture(path):
    import oracle_binding as ob

    pta = importlib.import_module("path-tracer_amd")
    wd = pta.work_dispatcher
    scene = pta.scenes.load_fixture_scene("sphere_scene1", os.path.join(HERE, "golden"))
    packed = wd.pack_scene(scene, ob.OracleBvhBuilder(), 8)
    win = wd.PTRenderWindow(render_width=W, render_height=H, max_ray_depth=DEPTH, max_refraction_bounces=4)
    win.frame = 0
    push = wd.push_constant_bytes(scene.camera, win, wd.sample_time(5))
    buffers = [
        (wd.OBJECT_SET_INDEX, wd.MATERIALS_BIND, packed.materials), (wd.OBJECT_SET_INDEX, wd.SPHERES_BIND, packed.spheres),
        (wd.OBJECT_SET_INDEX, wd.PLANES_BIND, packed.planes), (wd.OBJECT_SET_INDEX, wd.TRIANGLES_BIND, packed.triangles),
        (wd.BVH_SET_INDEX, wd.OBJECT_ID_BIND, packed.object_ids), (wd.BVH_SET_INDEX, wd.BVH_BIND, packed.bvh),
        (wd.TRIANGLE_SET_INDEX, wd.TRIANGLE_VERTEX_BIND, bytes(12)), (wd.TRIANGLE_SET_INDEX, wd.TRIANGLE_UV_BIND, bytes(12)),
        (wd.TRIANGLE_SET_INDEX, wd.TRIANGLE_INDEX_BIND, bytes(12)), (wd.TRANSFORM_SET_INDEX, wd.TRANSFORM_BIND, packed.transforms),
    ]
    out = b"PTKH" + struct.pack("<I8i", 1, W, H, W // 8, H // 8, packed.bvh_order, 2, 0, 0) + push
    out += struct.pack("<I", len(packed.proc_tex_paths))
    for p in packed.proc_tex_paths:
        out += struct.pack("<I", len(p.encode())) + p.encode()
    out += struct.pack("<I", len(buffers))
    for s, b, data in buffers:
        out += struct.pack("<iiQ", s, b, len(data)) + data
    with open(path, "wb") as f:
        f.write(out)
    return len(out)


def read_fixture(path):
    """Parses the container back: (header dict, push bytes, texture paths, {(set, binding): bytes})."""
    with open(path, "rb") as f:
        d = f.read()
    assert d[:4] == b"PTKH"
    ver, w, h, gx, gy, order, rmode, emission, early_t = struct.unpack_from("<I8i", d, 4)
    off = 4 + 36
    push = d[off:off + 128]
    off += 128
    (n_paths,) = struct.unpack_from("<I", d, off)
    off += 4
    paths = []
    for _ in range(n_paths):
        (ln,) = struct.unpack_from("<I", d, off)
        paths.append(d[off + 4:off + 4 + ln].decode())
        off += 4 + ln
    (n_buf,) = struct.unpack_from("<I", d, off)
    off += 4
    bufs = {}
    for _ in range(n_buf):
        s, b, n = struct.unpack_from("<iiQ", d, off)
        bufs[(s, b)] = d[off + 16:off + 16 + n]
        off += 16 + n
    return dict(width=w, height=h, groups_x=gx, groups_y=gy, order=order, refraction_mode=rmode, use_emission=emission,
                early_t_break=early_t), push, paths, bufs


if __name__ == "__main__":
    n = write_fixture(os.path.join(HERE, "golden", "c_harness_sphere_scene1.bin"))
    print("wrote", n, "bytes")
