"""GPU parity of the incremental edit path (pt_renderer.gd:664-788): single records and node boxes are
rewritten with buffer_update between samples.  After every edit both renderers must still match the CPU
oracle fed with the same bytes, and edits that keep the tree's topology must be absorbed without a full
re-validation of the scene."""
import copy
import struct

import numpy as np
import pytest

import oracle_binding as ob
from test_gpu_parity import _compare_image, rd, rd_wf  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu


def _render_all(pta, devs, packed, scene, win, k, cpu):
    wd = pta.work_dispatcher
    t = wd.sample_time(k)
    win.frame = k
    ob.render(packed, wd.push_constant_bytes(scene.camera, win, t), cpu, win.work_group_width, win.work_group_height)
    for name, dev, disp in devs:
        disp.create_compute_list(win, scene.camera, t, sync=True)
        _compare_image(f"{name}[{k}]", dev.texture_get_data(disp.image_buffer), cpu)


def _edit(packed, field, offset, data):
    b = bytearray(getattr(packed, field))
    b[offset:offset + len(data)] = data
    setattr(packed, field, bytes(b))


@pytest.mark.parametrize("scene_name", ["benchmark_random_scene", "dragon_test"])
def test_edits_between_samples(pta, rd, rd_wf, golden_dir, scene_name):
    wd = pta.work_dispatcher
    scene = pta.scenes.load_fixture_scene(scene_name, golden_dir)
    packed = copy.copy(wd.pack_scene(scene, ob.OracleBvhBuilder(), 8))
    w, h = 320, 184
    devs = []
    for name, dev in (("one-kernel", rd), (rd_wf.pinned, rd_wf)):
        d = wd.PTWorkDispatcher(dev)
        d.create_buffers(w, h, packed=packed)
        d.load_shader()
        devs.append((name, dev, d))
    win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=5, max_refraction_bounces=4)
    cpu = np.zeros((h, w, 4), np.float32)
    rd.set_renderer(pta.rendering_device.PTK_RENDERER_SINGLE_KERNEL)
    try:
        _render_all(pta, devs, packed, scene, win, 0, cpu)
        base_stats = [dev.update_stats() for _, dev, _ in devs]

        def apply(field, attr, offset, data):
            _edit(packed, field, offset, data)
            for _, dev, d in devs:
                dev.buffer_update(getattr(d, attr), offset, len(data), data)

        # 1. a material changes colour (nothing derived depends on material contents)
        apply("materials", "material_buffer", 2 * 64, struct.pack("<3f", 0.9, 0.2, 0.1))
        # 2. primitives shrink / move inside their old boxes
        n_sph = len(packed.spheres) // 32
        for i in range(0, min(n_sph, 40), 3):
            c = struct.unpack_from("<4f", packed.spheres, i * 32)
            if c[3] > 0:
                apply("spheres", "sphere_buffer", i * 32, struct.pack("<4f", c[0], c[1], c[2], c[3] * 0.5))
        n_tri = len(packed.triangles) // 80
        for i in range(0, min(n_tri, 400), 7):
            a = np.frombuffer(packed.triangles, np.float32, 12, i * 80).copy()
            centre = (a[0:3] + a[4:7] + a[8:11]) / np.float32(3)
            for v in (0, 4, 8):  # pull the corners a quarter of the way towards the centre
                a[v:v + 3] = a[v:v + 3] * np.float32(0.75) + centre * np.float32(0.25)
            apply("triangles", "triangle_buffer", i * 80, a.tobytes())
        _render_all(pta, devs, packed, scene, win, 1, cpu)
        # 3. node boxes grow a little (a refit after a move); ids and transform indices stay
        n_nodes = len(packed.bvh) // 32
        for i in range(1, min(n_nodes, 200), 5):
            rec = np.frombuffer(packed.bvh, np.float32, 8, i * 32).copy()
            words = np.frombuffer(packed.bvh, np.int32, 8, i * 32)
            if words[3] == 0 and words[7] == 0:
                continue  # padding record
            rec[0:3] -= np.float32(0.125)
            rec[4:7] += np.float32(0.125)
            apply("bvh", "bvh_buffer", i * 32, rec[0:3].tobytes())         # min corner only
            apply("bvh", "bvh_buffer", i * 32 + 16, rec[4:7].tobytes())    # max corner only
        # 4. a plane is switched off (zero normal) if the scene has one
        if any(struct.unpack_from("<3f", packed.planes, 0)):
            apply("planes", "plane_buffer", 0, struct.pack("<3f", 0.0, 0.0, 0.0))
        _render_all(pta, devs, packed, scene, win, 2, cpu)
        for (_, dev, _), (full0, patched0) in zip(devs, base_stats):
            full, patched = dev.update_stats()
            assert full == full0, "topology-preserving edits must not trigger a full re-validation"
            assert patched > patched0
        # 5. an inner node loses its last child (its subtree can no longer be reached): the topology changed, this one does rebuild
        words = np.frombuffer(packed.bvh, np.int32).reshape(-1, 8)
        inner = next(i for i in range(n_nodes) if words[i, 3] > 0 and (words[i, 3] >> 24) >= 2 and words[i, 7] < 0)
        idx, count = int(words[inner, 3]) & 0x00FFFFFF, int(words[inner, 3]) >> 24
        apply("bvh", "bvh_buffer", inner * 32 + 12, struct.pack("<i", idx ^ ((count - 1) << 24)))
        _render_all(pta, devs, packed, scene, win, 3, cpu)
        for (_, dev, _), (full0, _) in zip(devs, base_stats):
            assert dev.update_stats()[0] == full0 + 1
        # 6. an edit that breaks the scene is reported at the next dispatch, not executed
        bad = struct.pack("<i", 10 ** 6)
        for _, dev, d in devs:
            dev.buffer_update(d.sphere_buffer, 16, 4, bad)  # material index of sphere 0
            with pytest.raises(pta.PtkError):
                d.create_compute_list(win, scene.camera, 0.0, sync=True)
            dev.buffer_update(d.sphere_buffer, 16, 4, packed.spheres[16:20])
            d.create_compute_list(win, scene.camera, 0.0, sync=True)
    finally:
        rd.set_renderer(pta.rendering_device.PTK_RENDERER_AUTO)
        for _, _, d in devs:
            d.free_rids()


@pytest.mark.parametrize("scene_name", ["benchmark_random_scene", "dragon_test"])
def test_objects_leave_and_join_leaves_between_samples(pta, rd, rd_wf, golden_dir, scene_name):
    """The plugin's remove_object / add_object on a leaf with room (pt_bvh_tree.gd:333-338, 375-385) as the buffers
    see them: the leaf's record comes back with another object count (same first id slot), and the ids of its slots
    are written again.  Both are absorbed by patching -- the parent's block entry, the slots' leaf records -- without
    a re-validation of the scene, and the image follows the oracle's on the edited bytes."""
    wd = pta.work_dispatcher
    scene = pta.scenes.load_fixture_scene(scene_name, golden_dir)
    packed = copy.copy(wd.pack_scene(scene, ob.OracleBvhBuilder(), 8))
    w, h = 320, 184
    devs = []
    for name, dev in (("one-kernel", rd), (rd_wf.pinned, rd_wf)):
        d = wd.PTWorkDispatcher(dev)
        d.create_buffers(w, h, packed=packed)
        d.load_shader()
        devs.append((name, dev, d))
    win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=5, max_refraction_bounces=4)
    cpu = np.zeros((h, w, 4), np.float32)
    rd.set_renderer(pta.rendering_device.PTK_RENDERER_SINGLE_KERNEL)
    try:
        _render_all(pta, devs, packed, scene, win, 0, cpu)
        base_stats = [dev.update_stats() for _, dev, _ in devs]

        def apply(field, attr, offset, data):
            _edit(packed, field, offset, data)
            for _, dev, d in devs:
                dev.buffer_update(getattr(d, attr), offset, len(data), data)

        words = np.frombuffer(packed.bvh, np.int32).reshape(-1, 8).copy()
        ids = np.frombuffer(packed.object_ids, np.int32).copy()
        leaves = []
        for i in range(1, words.shape[0]):
            node_id = int(words[i, 3])
            if node_id >= 0:
                continue
            idx = ((node_id & 0x00FFFFFF) ^ 0x00800000) - 0x00800000          # sign-extended low 24 bits
            count = abs(((node_id >> 24) & 0xFF ^ 0x80) - 0x80)                # |sign-extended top byte|
            if idx < 0 and count >= 2:
                leaves.append((i, -idx - 1, count))
        leaves = leaves[::max(1, len(leaves) // 24)][:24]
        assert len(leaves) >= 8

        def leaf_word(base, count):  # PTBVHTree.BVHNode.to_byte_array for a leaf, pt_bvh_tree.gd:931-944
            v = ((-base - 1) ^ ((count - 1) << 24)) & 0xFFFFFFFF
            return struct.pack("<I", v)

        # an object leaves each of these leaves: its id goes to the end of the leaf's slots, the others move up, the count drops
        for node, base, count in leaves:
            rolled = np.concatenate([ids[base + 1:base + count], ids[base:base + 1]]).astype(np.int32)
            apply("object_ids", "object_id_buffer", base * 4, rolled.tobytes())
            apply("bvh", "bvh_buffer", node * 32 + 12, leaf_word(base, count - 1))
        _render_all(pta, devs, packed, scene, win, 1, cpu)
        # ... and joins again
        for node, base, count in leaves:
            apply("bvh", "bvh_buffer", node * 32 + 12, leaf_word(base, count))
        _render_all(pta, devs, packed, scene, win, 2, cpu)
        # a slot is given to another primitive of the scene (an object swapped for another one of the same kind)
        for node, base, count in leaves[:8]:
            other = ids[leaves[-1][1]]
            apply("object_ids", "object_id_buffer", base * 4, struct.pack("<i", int(other)))
        _render_all(pta, devs, packed, scene, win, 3, cpu)
        for (_, dev, _), (full0, patched0) in zip(devs, base_stats):
            full, patched = dev.update_stats()
            assert full == full0, "objects leaving and joining leaves must be absorbed without re-validating the scene"
            assert patched >= patched0 + 3 * len(leaves)

        # a leaf that grows over a slot naming a primitive the scene does not have: reported at the next dispatch, not executed
        node, base, count = leaves[0]
        good = bytes(packed.object_ids[(base + count - 1) * 4:(base + count) * 4])
        for _, dev, d in devs:
            dev.buffer_update(d.bvh_buffer, node * 32 + 12, 4, leaf_word(base, count - 1))
            dev.buffer_update(d.object_id_buffer, (base + count - 1) * 4, 4, struct.pack("<I", (3 << 24) | 0x7FFFF0))
            d.create_compute_list(win, scene.camera, 0.0, sync=True)      # (the slot cannot be reached: a valid scene)
            dev.buffer_update(d.bvh_buffer, node * 32 + 12, 4, leaf_word(base, count))
            with pytest.raises(pta.PtkError):
                d.create_compute_list(win, scene.camera, 0.0, sync=True)
            dev.buffer_update(d.object_id_buffer, (base + count - 1) * 4, 4, good)
            d.create_compute_list(win, scene.camera, 0.0, sync=True)
    finally:
        rd.set_renderer(pta.rendering_device.PTK_RENDERER_AUTO)
        for _, _, d in devs:
            d.free_rids()


def test_buffers_grow_with_the_scene(pta, golden_dir):
    """PTWorkDispatcher.expand_object_buffer / expand_material_buffer / expand_bvh_buffer
    (pt_work_dispatcher.gd:511-637): objects and materials are added until every buffer's count step is crossed;
    the grown buffers take the old ones' bindings, the new tree replaces the old one, and the image equals the
    oracle's render of the freshly packed scene."""
    wd, sc, gm = pta.work_dispatcher, pta.scene, pta.godot_math
    scene = pta.scenes.load_fixture_scene("sphere_scene1", golden_dir)
    w, h = 160, 96
    dev = pta.PtkRenderingDevice(0)
    try:
        disp = wd.PTWorkDispatcher(dev)
        disp.set_scene(scene)
        disp.create_buffers(w, h)
        disp.load_shader()
        before = (disp.sphere_buffer_size, disp.plane_buffer_size, disp.triangle_buffer_size, disp.material_buffer_size, disp.bvh_buffer_size)
        assert not disp.check_object_buffer_size() and not disp.expand_material_buffer(), "a scene that fits grows nothing"
        rng = np.random.default_rng(23)
        mats = [sc.PTMaterial(albedo=tuple(rng.uniform(0.2, 0.9, 3).tolist()), metallic=float(rng.random() > 0.7)) for _ in range(20)]
        for i in range(70):  # past one sphere step (64)
            scene.add_object(sc.PTSphere(transform=gm.Transform3D(origin=tuple(rng.uniform(-4.0, 4.0, 3).tolist())), radius=0.25, material=mats[i % 20]))
        for i in range(18):  # past one triangle step (16)
            p = rng.uniform(-3.0, 3.0, 3)
            scene.add_object(sc.PTTriangle(vertex1=tuple(p.tolist()), vertex2=tuple((p + [0.8, 0.0, 0.1]).tolist()), vertex3=tuple((p + [0.1, 0.9, 0.0]).tolist()), material=mats[i % 20]))
        for i in range(17):  # past one plane step (16)
            scene.add_object(sc.PTPlane(normal=(0.0, 1.0, 0.0), distance=-6.0 - i, material=mats[i % 20]))
        assert disp.check_object_buffer_size()
        assert disp.expand_material_buffer()
        nodes, ids = wd.create_scene_bvh(scene, wd.DeviceBvhBuilder(dev), 8)
        assert not disp.expand_bvh_buffer(nodes, ids), "the new tree (about 20 nodes) still fits the 64-node buffer: rewritten in place"
        after = (disp.sphere_buffer_size, disp.plane_buffer_size, disp.triangle_buffer_size, disp.material_buffer_size, disp.bvh_buffer_size)
        assert all(a > b for a, b in zip(after[:4], before[:4])) and after[4] == before[4], (before, after)
        assert not disp.check_object_buffer_size(), "second call: everything fits"
        packed = wd.pack_scene(scene, ob.OracleBvhBuilder(), 8)
        cpu = np.zeros((h, w, 4), np.float32)
        for k in range(2):
            win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=4)
            win.frame = k
            t = wd.sample_time(k)
            ob.render(packed, wd.push_constant_bytes(scene.camera, win, t), cpu, w // 8, h // 8, trace=False)
            disp.create_compute_list(win, scene.camera, t, sync=True)
            _compare_image(f"grown[{k}]", dev.texture_get_data(disp.image_buffer), cpu)
        # growing by explicit steps keeps the contents
        assert disp.expand_object_buffer(sc.SPHERE, steps=2) and disp.sphere_buffer_size == after[0] + 128
        assert disp.expand_bvh_buffer(nodes, ids, steps=1) and disp.bvh_buffer_size == after[4] + 64
        win.frame = 2
        t = wd.sample_time(2)
        ob.render(packed, wd.push_constant_bytes(scene.camera, win, t), cpu, w // 8, h // 8, trace=False)
        disp.create_compute_list(win, scene.camera, t, sync=True)
        _compare_image("grown by steps", dev.texture_get_data(disp.image_buffer), cpu)
        disp.free_rids()
    finally:
        dev.close()


def test_launches_follow_the_materials(pta, golden_dir):
    """The pipeline launches once per bounce a path can take.  The bounce loop's bound grows with a path's refraction
    bounces (ray_tracer.comp:1476, 1519, 1573), which only a hit on a material with opacity < 1 (or on material 0)
    can raise: a scene without such a material among its primitives is rendered with max_default_depth launches,
    and a material edit that brings one back brings the launches back with it.  Same image as the oracle's each time."""
    wd = pta.work_dispatcher
    scene = pta.scenes.load_fixture_scene("sphere_scene1", golden_dir)
    packed = copy.copy(wd.pack_scene(scene, ob.OracleBvhBuilder(), 8))
    w, h = 320, 184
    dev = pta.PtkRenderingDevice(0)
    try:
        dev.set_renderer(pta.rendering_device.PTK_RENDERER_WAVEFRONT)
        d = wd.PTWorkDispatcher(dev)
        d.create_buffers(w, h, packed=packed)
        d.load_shader()
        win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=5, max_refraction_bounces=4)
        cpu = np.zeros((h, w, 4), np.float32)
        devs = [("pipeline", dev, d)]
        _render_all(pta, devs, packed, scene, win, 0, cpu)
        assert len(dev.dispatch_traverse_ms(0)) == 5 + 3  # the scene has glass
        n_mats = len(packed.materials) // 64
        glass = [m for m in range(1, n_mats) if struct.unpack_from("<f", packed.materials, m * 64 + 20)[0] < 1.0]
        assert glass
        for m in glass:  # every material opaque
            data = struct.pack("<f", 1.0)
            _edit(packed, "materials", m * 64 + 20, data)
            dev.buffer_update(d.material_buffer, m * 64 + 20, 4, data)
        _render_all(pta, devs, packed, scene, win, 1, cpu)
        assert len(dev.dispatch_traverse_ms(0)) == 5
        data = struct.pack("<f", 0.25)  # and one of them glass again
        _edit(packed, "materials", glass[0] * 64 + 20, data)
        dev.buffer_update(d.material_buffer, glass[0] * 64 + 20, 4, data)
        _render_all(pta, devs, packed, scene, win, 2, cpu)
        assert len(dev.dispatch_traverse_ms(0)) == 5 + 3
        d.free_rids()
    finally:
        dev.close()


def test_a_primitive_that_turns_to_glass_brings_the_launches_back(pta, golden_dir):
    """The same bound on the launches as above, reached from the primitives' side: a scene whose glass material no sphere
    uses is rendered with max_default_depth launches; one sphere record rewritten to use it (a patched edit, no
    re-validation) makes paths refract again, and the launches for their extra bounces return."""
    wd = pta.work_dispatcher
    scene = pta.scenes.load_fixture_scene("sphere_scene1", golden_dir)
    packed = copy.copy(wd.pack_scene(scene, ob.OracleBvhBuilder(), 8))
    n_mats, n_sph = len(packed.materials) // 64, len(packed.spheres) // 32
    glass = [m for m in range(1, n_mats) if struct.unpack_from("<f", packed.materials, m * 64 + 20)[0] < 1.0]
    opaque = [m for m in range(1, n_mats) if m not in glass][0]
    users = [i for i in range(n_sph) if struct.unpack_from("<i", packed.spheres, i * 32 + 16)[0] in glass
             and struct.unpack_from("<f", packed.spheres, i * 32 + 12)[0] > 0]
    assert glass and users
    for i in users:
        _edit(packed, "spheres", i * 32 + 16, struct.pack("<i", opaque))
    w, h = 320, 184
    dev = pta.PtkRenderingDevice(0)
    try:
        dev.set_renderer(pta.rendering_device.PTK_RENDERER_WAVEFRONT)
        d = wd.PTWorkDispatcher(dev)
        d.create_buffers(w, h, packed=packed)
        d.load_shader()
        win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=5, max_refraction_bounces=4)
        cpu = np.zeros((h, w, 4), np.float32)
        devs = [("pipeline", dev, d)]
        _render_all(pta, devs, packed, scene, win, 0, cpu)
        assert len(dev.dispatch_traverse_ms(0)) == 5
        validations = dev.update_stats()[0]
        data = struct.pack("<i", glass[0])
        _edit(packed, "spheres", users[0] * 32 + 16, data)
        dev.buffer_update(d.sphere_buffer, users[0] * 32 + 16, 4, data)
        _render_all(pta, devs, packed, scene, win, 1, cpu)
        assert len(dev.dispatch_traverse_ms(0)) == 5 + 3
        assert dev.update_stats()[0] == validations
        d.free_rids()
    finally:
        dev.close()


def _leaf_records(dev):
    import ctypes as C
    dev.lib.ptk_dev_read_derived.restype = C.c_int
    dev.lib.ptk_dev_read_derived.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    n = C.c_size_t(0)
    assert dev.lib.ptk_dev_read_derived(dev.ctx, 6, None, 0, C.byref(n)) == 0
    out = np.zeros(n.value, np.uint32)
    assert dev.lib.ptk_dev_read_derived(dev.ctx, 6, out.ctypes.data, n.value, C.byref(n)) == 0
    return out.reshape(-1, 12)


def test_triangle_patches_do_not_depend_on_the_callers_rounding_mode(pta, rd_wf, golden_dir):
    """A triangle rewritten with buffer_update gets the ray-independent part of its test (edges, normal components) from
    host arithmetic.  The calling thread rounds toward zero while it makes the edits: the patched leaf records must be,
    bit for bit, what the device kernel derives from the same bytes when the scene is bound anew (the image is a weak
    witness here: a last-bit change of a normal rarely changes which object a path hits)."""
    import ctypes
    import ctypes.util
    libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    FE_TONEAREST, FE_TOWARDZERO = 0x000, 0xC00  # <fenv.h>, x86-64
    wd = pta.work_dispatcher
    scene = pta.scenes.load_fixture_scene("dragon_test", golden_dir)
    packed = copy.copy(wd.pack_scene(scene, ob.OracleBvhBuilder(), 8))
    w, h = 320, 184
    d = wd.PTWorkDispatcher(rd_wf)
    d.create_buffers(w, h, packed=packed)
    d.load_shader()
    devs = [(rd_wf.pinned, rd_wf, d)]
    win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=5, max_refraction_bounces=4)
    cpu = np.zeros((h, w, 4), np.float32)
    try:
        _render_all(pta, devs, packed, scene, win, 0, cpu)
        validations = rd_wf.update_stats()[0]
        before = _leaf_records(rd_wf)
        edits = []
        for i in range(0, 1200, 3):
            a = np.frombuffer(packed.triangles, np.float32, 12, i * 80).copy()
            centre = (a[0:3] + a[4:7] + a[8:11]) / np.float32(3)
            for v in (0, 4, 8):
                a[v:v + 3] = a[v:v + 3] * np.float32(0.8125) + centre * np.float32(0.1875)
            edits.append((i * 80, a.tobytes()))
        assert libm.fesetround(FE_TOWARDZERO) == 0
        try:
            for offset, data in edits:
                rd_wf.buffer_update(d.triangle_buffer, offset, len(data), data)
        finally:
            libm.fesetround(FE_TONEAREST)
        for offset, data in edits:
            _edit(packed, "triangles", offset, data)
        _render_all(pta, devs, packed, scene, win, 1, cpu)
        assert rd_wf.update_stats()[0] == validations, "the edits were to be patched, not re-validated"
        patched = _leaf_records(rd_wf)
        assert (patched != before).any(axis=1).sum() >= len(edits) // 2, "the edits did not reach the leaf records"
        # the same bytes bound anew: the device kernel makes every record
        d.free_rids()
        d = wd.PTWorkDispatcher(rd_wf)
        d.create_buffers(w, h, packed=packed)
        d.load_shader()
        devs = [(rd_wf.pinned, rd_wf, d)]
        _render_all(pta, devs, packed, scene, win, 0, np.zeros((h, w, 4), np.float32))
        fresh = _leaf_records(rd_wf)
        assert patched.shape == fresh.shape
        differ = np.flatnonzero((patched != fresh).any(axis=1))
        assert differ.size == 0, f"{differ.size} patched leaf records differ from the device's, first at id slot {differ[0]}"
    finally:
        d.free_rids()
