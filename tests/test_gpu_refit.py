"""GPU parity of the incremental refit (ptk_bvh_bind + ptk_bvh_refit) against the CPU restatement of the
reference's update_aabb chains (oracle/pt_bvh_oracle.c:pbo_refit), through the C ABI: objects move over ten
frames; after every frame the bound node buffer equals the oracle's node bytes byte for byte, the set of
rewritten nodes equals the reference's update queue, and the rendered images equal the oracle's."""
import copy

import numpy as np
import pytest

import oracle_binding as ob
from test_gpu_parity import _compare_image

pytestmark = pytest.mark.gpu


def _sphere_box(center, r):
    c = np.asarray(center, np.float32)
    r = np.float32(r)
    return np.concatenate([c - r, np.array([r + r] * 3, np.float32)]).astype(np.float32)  # AABB(pos - r, 2r), pt_sphere.gd:58-65


def _setup(pta, scene, w, h):
    wd = pta.work_dispatcher
    oracle = ob.OracleBvhBuilder()
    ot, o_meshes = wd.build_scene_tree(scene, oracle, 8)
    cpu_nodes, cpu_ids = oracle.read(ot)
    packed = copy.copy(wd.pack_scene(scene, ob.OracleBvhBuilder(), 8))
    assert wd.create_bvh_byte_array(cpu_nodes) == packed.bvh
    dev = pta.PtkRenderingDevice(0)
    disp = wd.PTWorkDispatcher(dev)
    disp.create_buffers(w, h, packed=packed)
    disp.load_shader()
    gt, g_meshes = wd.build_scene_tree(scene, wd.DeviceBvhBuilder(dev), 8)
    hn, hi = dev.bvh_bind(gt)
    for handle in (disp.bvh_buffer, disp.object_id_buffer):  # the uploaded copies are replaced by the bound tree
        dev.free_rid(handle)
        disp.rids_to_free.remove(handle)
    disp.bvh_buffer, disp.object_id_buffer = hn, hi
    disp.rids_to_free += [hn, hi]
    assert dev.buffer_get_data(hn, 0, len(packed.bvh)) == packed.bvh, "bound node bytes differ from the oracle's tree"
    assert dev.buffer_get_data(hi, 0, len(packed.object_ids)) == packed.object_ids
    return wd, oracle, ot, bytearray(cpu_nodes), packed, dev, disp, (gt, g_meshes)


def _render_check(pta, wd, dev, disp, packed, scene, w, h, k, name):
    win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=4, max_refraction_bounces=3)
    win.frame = 0
    t = wd.sample_time(k)
    cpu = np.zeros((h, w, 4), np.float32)
    ob.render(packed, wd.push_constant_bytes(scene.camera, win, t), cpu, w // 8, h // 8, trace=False)
    for rname, r in (("one-kernel", 1), ("wavefront", 2), ("persistent", 3)):
        dev.set_renderer(r)
        dev.texture_update(disp.image_buffer, np.zeros((h, w, 4), np.float32))
        disp.create_compute_list(win, scene.camera, t, sync=True)
        _compare_image(f"{name}[{k}] {rname}", dev.texture_get_data(disp.image_buffer), cpu)


def test_moving_spheres_ten_frames(pta, golden_dir):
    scene = pta.scenes.load_fixture_scene("benchmark_random_scene", golden_dir)
    w, h = 256, 144
    wd, oracle, ot, cpu_nodes, packed, dev, disp, trees = _setup(pta, scene, w, h)
    rng = np.random.default_rng(17)
    sph = np.frombuffer(packed.spheres, np.float32).reshape(-1, 8).copy()
    n_sph = len(scene.spheres)
    full0 = dev.update_stats()[0]
    try:
        _render_check(pta, wd, dev, disp, packed, scene, w, h, 0, "refit")
        for frame in range(1, 11):
            # a mix of edits: real moves, a move smaller than the comparison tolerance, the same sphere twice, two
            # spheres that share leaves with others
            picks = list(rng.choice(n_sph, size=int(rng.integers(1, 40)), replace=False))
            if frame % 3 == 0:
                picks += picks[:2]  # edited twice in one call
            ids, vals = [], []
            for i in picks:
                step = rng.normal(0, 0.35, 3) if rng.random() > 0.25 else rng.normal(0, 3e-7, 3)
                sph[i, :3] = (sph[i, :3] + step.astype(np.float32)).astype(np.float32)
                if frame % 4 == 0:
                    sph[i, 3] = np.float32(sph[i, 3] * rng.uniform(0.8, 1.2))
                ids.append(i | (1 << 24))
                v = np.zeros(12, np.float32)
                v[:6] = _sphere_box(sph[i, :3], sph[i, 3])
                vals.append(v)
                dev.buffer_update(disp.sphere_buffer, i * 32, 16, sph[i, :4].tobytes())
            packed.spheres = sph.tobytes()
            queued = oracle.refit(ot, ids, np.stack(vals), cpu_nodes)
            n = dev.bvh_refit(ids, np.stack(vals))
            assert n == len(queued), f"frame {frame}: {n} nodes rewritten, the reference queues {len(queued)}"
            packed.bvh = wd.create_bvh_byte_array(bytes(cpu_nodes))
            got = dev.buffer_get_data(disp.bvh_buffer, 0, len(packed.bvh))
            if got != packed.bvh:
                a = np.frombuffer(got, np.uint32).reshape(-1, 8)
                b = np.frombuffer(packed.bvh, np.uint32).reshape(-1, 8)
                raise AssertionError(f"frame {frame}: node bytes differ at nodes {np.argwhere((a != b).any(1))[:8].ravel().tolist()}")
            _render_check(pta, wd, dev, disp, packed, scene, w, h, frame, "refit")
        assert dev.update_stats()[0] == full0 + 0 or dev.update_stats()[0] <= full0 + 1, "refits are patched in place, not re-validated"
    finally:
        disp.free_rids()
        dev.free_rid(trees[0])
        dev.close()
        oracle.release(ot)


def test_meshes_and_mesh_transforms(pta, golden_dir):
    """test_scene_bvh_b: objects inside meshes move (the chain climbs from the mesh's tree through its socket into
    the scene's tree) and a mesh's transform changes (update_mesh_socket_aabb)."""
    gm, sc = pta.godot_math, pta.scene
    scene = pta.scenes.load_fixture_scene("test_scene_bvh_b", golden_dir)
    scene.camera = sc.PTCamera(transform=gm.Transform3D.looking_at((4.0, 3.0, 6.0), (0.0, 0.0, 0.0)), fov=60.0)
    w, h = 192, 112
    wd, oracle, ot, cpu_nodes, packed, dev, disp, trees = _setup(pta, scene, w, h)
    rng = np.random.default_rng(5)
    try:
        ids_all = np.frombuffer(packed.object_ids, np.int32)
        sph = np.frombuffer(packed.spheres, np.float32).reshape(-1, 8).copy()
        sphere_ids = [int(i) for i in ids_all if (int(i) >> 24) == 1]
        for frame in range(1, 7):
            ids, vals = [], []
            for oid in rng.choice(sphere_ids, size=min(3, len(sphere_ids)), replace=False):
                i = int(oid) & 0xFFFFFF
                sph[i, :3] = (sph[i, :3] + rng.normal(0, 0.2, 3).astype(np.float32)).astype(np.float32)
                v = np.zeros(12, np.float32)
                v[:6] = _sphere_box(sph[i, :3], sph[i, 3])
                ids.append(int(oid)); vals.append(v)
                dev.buffer_update(disp.sphere_buffer, i * 32, 16, sph[i, :4].tobytes())
            if frame % 2 == 0:  # a mesh is moved: its socket's box is the transformed box of the mesh's root
                mi = frame // 2 % len(scene.meshes)
                mesh = scene.meshes[mi]
                mesh.transform = gm.Transform3D.looking_at(tuple(rng.uniform(-2, 2, 3)), tuple(rng.uniform(-0.5, 0.5, 3)))
                ids.append(mi | (pta.rendering_device.PTK_REFIT_MESH << 24))
                vals.append(np.array(mesh.transform.to_list(), np.float32))
                xf = wd.create_transform_byte_array(scene)
                dev.buffer_update(disp.transform_buffer, 0, len(xf), xf)
                packed.transforms = xf
            packed.spheres = sph.tobytes()
            queued = oracle.refit(ot, ids, np.stack(vals), cpu_nodes)
            n = dev.bvh_refit(ids, np.stack(vals))
            assert n == len(queued)
            packed.bvh = wd.create_bvh_byte_array(bytes(cpu_nodes))
            assert dev.buffer_get_data(disp.bvh_buffer, 0, len(packed.bvh)) == packed.bvh, f"frame {frame}: node bytes differ"
            _render_check(pta, wd, dev, disp, packed, scene, w, h, frame, "refit-mesh")
        with pytest.raises(pta.PtkError):
            dev.bvh_refit([123456 | (1 << 24)], np.zeros((1, 12), np.float32))  # not in the tree
        dev.buffer_update(disp.bvh_buffer, 0, 4, packed.bvh[:4])  # the caller edits the nodes itself: refit is off
        with pytest.raises(pta.PtkError):
            dev.bvh_refit(ids[:1], np.stack(vals)[:1])
    finally:
        disp.free_rids()
        dev.free_rid(trees[0])
        dev.close()
        oracle.release(ot)


def test_refit_4096_edits_byte_parity(pta):
    """10 000 spheres, three calls of 4096 edits each (a quarter of them below the comparison tolerance, many objects
    edited more than once per call): node bytes and queue size equal the reference's, edit order included."""
    scene = pta.scenes.random_spheres_scene(grid=50, seed=3)
    wd, oracle, ot, cpu_nodes, packed, dev, disp, trees = _setup(pta, scene, 64, 64)
    rng = np.random.default_rng(23)
    sph = np.frombuffer(packed.spheres, np.float32).reshape(-1, 8).copy()
    n_sph = len(scene.spheres)
    try:
        for call in range(3):
            picks = rng.integers(0, n_sph, 4096)
            ids = (picks | (1 << 24)).astype(np.int32)
            vals = np.zeros((4096, 12), np.float32)
            for j, i in enumerate(picks):
                step = rng.normal(0, 0.5, 3) if rng.random() > 0.25 else rng.normal(0, 2e-7, 3)
                sph[i, :3] = (sph[i, :3] + step.astype(np.float32)).astype(np.float32)
                vals[j, :6] = _sphere_box(sph[i, :3], sph[i, 3])
            queued = oracle.refit(ot, ids, vals, cpu_nodes)
            n = dev.bvh_refit(ids, vals)
            assert n == len(queued), f"call {call}: {n} vs {len(queued)} queued nodes"
            want = wd.create_bvh_byte_array(bytes(cpu_nodes))
            assert dev.buffer_get_data(disp.bvh_buffer, 0, len(want)) == want, f"call {call}: node bytes differ"
    finally:
        disp.free_rids()
        dev.free_rid(trees[0])
        dev.close()
        oracle.release(ot)
