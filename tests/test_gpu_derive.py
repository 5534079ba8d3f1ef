"""The checks and derivations that precede a scene's first dispatch -- every reachable index, the deepest traversal
stack, the inner nodes compiled into child-slot blocks (what ray_tracer.comp:1115-1181, 1267-1317 walk through the
node buffer of pt_bvh_tree.gd:910-959) -- run on the device for large trees and on the host for small ones.  Both
must describe the same tree: these tests force either path with PTK_DEVICE_DERIVE and compare what they derive,
render through the device-derived structures against the oracle, patch them, and feed both the same broken trees."""
import copy
import ctypes as C
import struct

import numpy as np
import pytest

import oracle_binding as ob
from test_gpu_c5 import deep_chain_scene
from test_gpu_parity import _compare_image

pytestmark = pytest.mark.gpu


def _device(pta, monkeypatch, on_device):
    monkeypatch.setenv("PTK_DEVICE_DERIVE", "1" if on_device else "0")
    dev = pta.PtkRenderingDevice(0)
    dev.lib.ptk_dev_read_derived.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    return dev


def _derived(dev, kind):
    n = C.c_size_t(0)
    assert dev.lib.ptk_dev_read_derived(dev.ctx, kind, None, 0, C.byref(n)) == 0
    out = np.zeros(max(n.value, 1), np.uint32)
    assert dev.lib.ptk_dev_read_derived(dev.ctx, kind, out.ctypes.data, n.value, C.byref(n)) == 0
    return out[:n.value]


def _canonical(blocks, sockets, entry):
    """The tree behind an entry word, free of block and socket numbers."""
    entry = int(entry)
    if entry >= 0x80000000 or entry == 0:
        return ("leaf", entry)
    if (entry & 0xFF000000) == 0x7F000000:
        s = entry & 0x00FFFFFF
        return ("socket", int(sockets[2 * s + 1]), _canonical(blocks, sockets, sockets[2 * s]))
    b, size = entry & 0x00FFFFFF, entry >> 24
    blk = blocks[b * 64:(b + 1) * 64]
    kids = []
    for l in range(size):
        kids.append((tuple(int(blk[p * 8 + l]) for p in range(6)), _canonical(blocks, sockets, blk[48 + l])))
    assert not blk[48 + size:56].any() and not blk[57:64].any()
    return ("node", tuple(kids))


def _scenes(pta, golden_dir):
    wd = pta.work_dispatcher
    for name in ("sphere_scene1", "benchmark_random_scene", "sphere_scene3", "fake_cornell_box", "dragon_test", "test_scene_bvh_b"):
        scene = pta.scenes.load_fixture_scene(name, golden_dir)
        yield name, scene, wd.pack_scene(scene, ob.OracleBvhBuilder(), 8)
    scene = pta.scenes.atrium_scene(20000, seed=5)
    yield "atrium_20000", scene, wd.pack_scene(scene, ob.OracleBvhBuilder(), 8)
    scene, packed = deep_chain_scene(pta, depth=9)
    yield "deep_chain", scene, packed
    # nothing in the tree at all (an all-zero root, 8 bytes of id table), and a lone sphere (the root is a leaf)
    scene = pta.scene.PTScene()
    scene.add_object(pta.scene.PTPlane(normal=(0.0, 1.0, 0.0), distance=-1.0))
    scene.camera = pta.scene.PTCamera()
    yield "planes_only", scene, wd.pack_scene(scene, ob.OracleBvhBuilder(), 8)
    scene = pta.scene.PTScene()
    scene.add_object(pta.scene.PTSphere())
    scene.camera = pta.scene.PTCamera()
    yield "one_sphere", scene, wd.pack_scene(scene, ob.OracleBvhBuilder(), 8)


def test_device_and_host_derive_the_same_tree(pta, golden_dir, monkeypatch):
    wd = pta.work_dispatcher
    w, h = 160, 96
    for name, scene, packed in _scenes(pta, golden_dir):
        got = []
        for on_device in (False, True):
            dev = _device(pta, monkeypatch, on_device)
            try:
                d = wd.PTWorkDispatcher(dev)
                d.create_buffers(w, h, packed=packed)
                d.load_shader(use_emission=(name == "fake_cornell_box"))
                win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=4, max_refraction_bounces=4)
                d.create_compute_list(win, scene.camera, wd.sample_time(0), sync=True)
                info = _derived(dev, 5)
                assert (info[5] > 0) == on_device, f"{name}: the forced path did not run"
                blocks, sockets = _derived(dev, 0), _derived(dev, 1)
                node_slot, tri_slot, sph_slot = _derived(dev, 2).view(np.int32), _derived(dev, 3).view(np.int32), _derived(dev, 4).view(np.int32)
                tree = _canonical(blocks, sockets, info[6]) if info[4] else None
                # every node with a slot: the slot holds its box (and its id, if it is a leaf)
                nodes = np.frombuffer(packed.bvh, np.uint32).reshape(-1, 8)
                for i in np.nonzero(node_slot >= 0)[0]:
                    b, l = node_slot[i] >> 3, node_slot[i] & 7
                    blk = blocks[b * 64:(b + 1) * 64]
                    assert [blk[p * 8 + l] for p in range(6)] == [nodes[i][k] for k in (0, 1, 2, 4, 5, 6)], f"{name}: node {i}"
                got.append((tuple(int(x) for x in info[[0, 1, 2, 3, 4, 7, 8]]), tree, node_slot >= 0, tri_slot.copy(), sph_slot.copy()))
                d.free_rids()
            finally:
                dev.close()
        (hi, ht, hs, htri, hsph), (di, dt, ds, dtri, dsph) = got
        assert hi == di, f"{name}: stack need, top blocks, blocks, sockets, coverage, refraction, triangles: host {hi}, device {di}"
        assert ht == dt, f"{name}: the compiled trees differ"
        assert np.array_equal(hs, ds) and np.array_equal(htri, dtri) and np.array_equal(hsph, dsph), f"{name}: slot maps differ"


@pytest.mark.parametrize("renderer", ["wavefront", "persistent", "single_kernel"])
def test_render_and_patch_through_device_derived_structures(pta, golden_dir, monkeypatch, renderer):
    wd = pta.work_dispatcher
    R = pta.rendering_device
    w, h = 320, 184
    for name in ("dragon_test", "benchmark_random_scene"):
        scene = pta.scenes.load_fixture_scene(name, golden_dir)
        packed = copy.copy(wd.pack_scene(scene, ob.OracleBvhBuilder(), 8))
        dev = _device(pta, monkeypatch, True)
        try:
            dev.set_renderer({"wavefront": R.PTK_RENDERER_WAVEFRONT, "persistent": R.PTK_RENDERER_PERSISTENT, "single_kernel": R.PTK_RENDERER_SINGLE_KERNEL}[renderer])
            d = wd.PTWorkDispatcher(dev)
            d.create_buffers(w, h, packed=packed)
            d.load_shader()
            win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=5, max_refraction_bounces=4)
            cpu = np.zeros((h, w, 4), np.float32)

            def render(k):
                t = wd.sample_time(k)
                win.frame = k
                ob.render(packed, wd.push_constant_bytes(scene.camera, win, t), cpu, win.work_group_width, win.work_group_height)
                d.create_compute_list(win, scene.camera, t, sync=True)
                _compare_image(f"{name}/{renderer}[{k}]", dev.texture_get_data(d.image_buffer), cpu)

            render(0)
            assert _derived(dev, 5)[5] > 0
            validations = dev.update_stats()[0]
            # edits that keep the topology: the maps they need come back from the device at the first of them
            def apply(field, attr, offset, data):
                b = bytearray(getattr(packed, field))
                b[offset:offset + len(data)] = data
                setattr(packed, field, bytes(b))
                dev.buffer_update(getattr(d, attr), offset, len(data), data)

            n_nodes = len(packed.bvh) // 32
            for i in range(1, min(n_nodes, 300), 7):
                rec = np.frombuffer(packed.bvh, np.float32, 8, i * 32).copy()
                words = np.frombuffer(packed.bvh, np.int32, 8, i * 32)
                if words[3] == 0 and words[7] == 0:
                    continue
                rec[0:3] -= np.float32(0.0625)
                rec[4:7] += np.float32(0.0625)
                apply("bvh", "bvh_buffer", i * 32, rec[0:3].tobytes())
                apply("bvh", "bvh_buffer", i * 32 + 16, rec[4:7].tobytes())
            if name == "dragon_test":
                for i in range(0, 600, 11):
                    a = np.frombuffer(packed.triangles, np.float32, 12, i * 80).copy()
                    centre = (a[0:3] + a[4:7] + a[8:11]) / np.float32(3)
                    for v in (0, 4, 8):
                        a[v:v + 3] = a[v:v + 3] * np.float32(0.75) + centre * np.float32(0.25)
                    apply("triangles", "triangle_buffer", i * 80, a.tobytes())
            else:
                for i in range(0, 60, 3):
                    c = struct.unpack_from("<4f", packed.spheres, i * 32)
                    apply("spheres", "sphere_buffer", i * 32, struct.pack("<4f", c[0], c[1], c[2], c[3] * 0.5))
            render(1)
            assert dev.update_stats()[0] == validations, "the edits were to be patched, not re-validated"
            d.free_rids()
        finally:
            dev.close()


def _broken_trees(packed):
    """(what is wrong, node bytes) for trees the validation must refuse."""
    nodes = np.frombuffer(packed.bvh, np.int32).reshape(-1, 8)
    inner = [i for i in range(len(nodes)) if nodes[i][3] > 0]
    leaves = [i for i in range(len(nodes)) if nodes[i][3] < 0]

    def with_id(i, word):
        b = bytearray(packed.bvh)
        struct.pack_into("<I", b, i * 32 + 12, word & 0xFFFFFFFF)
        return bytes(b)

    def with_xf(i, x):
        b = bytearray(packed.bvh)
        struct.pack_into("<i", b, i * 32 + 28, x)
        return bytes(b)

    n = len(nodes)
    first, size = int(nodes[inner[-1]][3]) & 0x00FFFFFF, int(nodes[inner[-1]][3]) >> 24
    yield "children outside the buffer", with_id(inner[-1], (size << 24) | (n - 1))
    yield "a node reachable twice", with_id(inner[-1], (size << 24) | int(nodes[0][3]) & 0x00FFFFFF)
    yield "too many children", with_id(inner[-1], (9 << 24) | first)
    yield "leaf ids outside the table", with_id(leaves[-1], (-(1 << 22)) ^ (1 << 24))
    yield "an id word PTBVHTree never writes", with_id(leaves[-1], 0x01FFFFFF)
    yield "transform index out of range", with_xf(inner[-1], 1000)
    yield "a mesh socket that is a leaf", with_xf(leaves[-1], 0)


def test_broken_trees_are_refused_alike(pta, golden_dir, monkeypatch):
    wd = pta.work_dispatcher
    scene = pta.scenes.load_fixture_scene("dragon_test", golden_dir)
    packed = copy.copy(wd.pack_scene(scene, ob.OracleBvhBuilder(), 8))
    good = packed.bvh
    win = wd.PTRenderWindow(render_width=64, render_height=64, max_ray_depth=2)
    for what, bvh in _broken_trees(packed):
        messages = []
        for on_device in (False, True):
            dev = _device(pta, monkeypatch, on_device)
            try:
                packed.bvh = bvh
                d = wd.PTWorkDispatcher(dev)
                d.create_buffers(64, 64, packed=packed)
                d.load_shader()
                with pytest.raises(pta.PtkError) as err:
                    d.create_compute_list(win, scene.camera, 0.0, sync=True)
                messages.append(str(err.value))
                # the same context takes the sound tree afterwards
                dev.buffer_update(d.bvh_buffer, 0, len(good), good)
                d.create_compute_list(win, scene.camera, 0.0, sync=True)
                assert (_derived(dev, 5)[5] > 0) == on_device
                d.free_rids()
            finally:
                dev.close()
        assert messages[0] == messages[1], f"{what}: host says {messages[0]!r}, device path says {messages[1]!r}"
    packed.bvh = good
    # a primitive record whose material index is outside the material buffer
    bad = bytearray(packed.triangles)
    struct.pack_into("<i", bad, 5 * 80 + 64, 10000)
    packed.triangles = bytes(bad)
    messages = []
    for on_device in (False, True):
        dev = _device(pta, monkeypatch, on_device)
        try:
            d = wd.PTWorkDispatcher(dev)
            d.create_buffers(64, 64, packed=packed)
            d.load_shader()
            with pytest.raises(pta.PtkError) as err:
                d.create_compute_list(win, scene.camera, 0.0, sync=True)
            messages.append(str(err.value))
            d.free_rids()
        finally:
            dev.close()
    assert messages[0] == messages[1] and "material index" in messages[0]
