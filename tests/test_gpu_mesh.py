"""GPU parity of mesh ingestion (ptk_mesh_to_triangles) through the C ABI, on the two meshes the reference
ships inline in its scene files: the dragon (compressed unorm16 positions, 16-bit indices, no UVs) and
grimchild (float32 positions, UVs in the attribute block, skinning channels that are skipped).  Bar:
records, boxes and UVs bit-exact against the numpy restatement (oracle/mesh_oracle.py) and the host packers;
the dragon's triangles also equal the committed tests/golden/scenes/dragon_test_tris.npy."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mesh_oracle as mo  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rd(pta):
    dev = pta.PtkRenderingDevice(0)
    yield dev
    dev.close()


def _surface(golden_dir, name):
    with np.load(os.path.join(golden_dir, "meshes", name + "_surface.npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name,n_tris", [("dragon", 8224), ("grimchild", 3168)])
def test_mesh_records_boxes_uvs(pta, rd, golden_dir, name, n_tris):
    s = _surface(golden_dir, name)
    rec, aabbs, uvs = rd.mesh_to_triangles(s, material_index=5, texture_id=(1 << 30) ^ 2)
    assert rec.shape == (n_tris, 20)
    tris = mo.triangle_vertices(s)
    ts = pta.scene.PTTriangleSet(vertices=tris)
    want = ts.records(np.full(n_tris, 5, np.int32), (1 << 30) ^ 2)
    assert np.array_equal(rec.view(np.uint32), want.view(np.uint32)), "80-byte records differ from the host packer's"
    assert np.array_equal(aabbs.view(np.uint32), ts.bvh_aabbs().view(np.uint32)), "triangle boxes differ"
    assert np.array_equal(uvs.view(np.uint32), mo.triangle_uvs(s).view(np.uint32)), "corner UVs differ"
    if name == "dragon":
        gold = np.load(os.path.join(golden_dir, "scenes", "dragon_test_tris.npy"))
        assert np.array_equal(rec[:, [0, 1, 2, 4, 5, 6, 8, 9, 10]].reshape(-1, 3, 3), gold)
        assert not uvs.any()
    else:
        assert uvs.min() >= -0.01 and uvs.max() <= 1.01 and uvs.std() > 0.1  # real texture coordinates


def test_mesh_with_transform_and_32bit_indices(pta, rd):
    """A synthetic surface above 65536 vertices (32-bit indices), with a rotation + translation; checked against
    Transform3D arithmetic on the host (basis rows dotted left to right, plus origin)."""
    rng = np.random.default_rng(4)
    nv = 70000
    verts = rng.uniform(-2, 2, (nv, 3)).astype(np.float32)
    idx = rng.integers(0, nv, 3 * 5000).astype("<u4")
    s = {"format": np.uint64(1 | (1 << 12)), "vertex_count": nv, "index_count": idx.size, "aabb": np.zeros(6, np.float32),
         "uv_scale": np.zeros(4, np.float32), "vertex_data": verts.view(np.uint8).ravel(), "attribute_data": np.zeros(0, np.uint8),
         "index_data": idx.view(np.uint8)}
    gm = pta.godot_math
    t = gm.Transform3D.looking_at((1.0, 2.0, 3.0), (0.0, 0.5, 0.0))
    t12 = np.array(t.to_list(), np.float32)
    rec, aabbs, _ = rd.mesh_to_triangles(s, transform12=t12, material_index=2)
    p = verts[idx.astype(np.int64)].reshape(-1, 3, 3)
    b, o = t12[:9].reshape(3, 3), t12[9:]
    want = np.stack([((b[i, 0] * p[..., 0] + b[i, 1] * p[..., 1]) + b[i, 2] * p[..., 2]) + o[i] for i in range(3)], axis=-1).astype(np.float32)
    got = rec[:, [0, 1, 2, 4, 5, 6, 8, 9, 10]].reshape(-1, 3, 3)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.all(aabbs[:, 3:] >= 0) and np.all(got.min(1) >= aabbs[:, :3] - 1e-5) and np.all(got.max(1) <= aabbs[:, :3] + aabbs[:, 3:] + 1e-5)


def test_mesh_errors(pta, rd, golden_dir):
    s = _surface(golden_dir, "dragon")
    bad = dict(s)
    bad["format"] = np.uint64(int(s["format"]) & ~(1 << 12))
    with pytest.raises(pta.PtkError, match="index"):
        rd.mesh_to_triangles(bad)
    bad = dict(s)
    bad["vertex_count"] = 100  # most indices now point past the vertex array
    with pytest.raises(pta.PtkError, match="index past"):
        rd.mesh_to_triangles(bad)
    bad = dict(s)
    bad["vertex_data"] = s["vertex_data"][:1000]
    with pytest.raises(pta.PtkError, match="vertex block"):
        rd.mesh_to_triangles(bad)
    bad = dict(s)
    bad["format"] = np.uint64(int(s["format"]) | (1 << 6))
    with pytest.raises(pta.PtkError, match="custom"):
        rd.mesh_to_triangles(bad)


def test_ingested_mesh_renders_like_the_fixture(pta, rd, golden_dir):
    """End to end: surface bytes -> records + boxes on the GPU -> GPU BVH build -> render equals the render of
    the dragon_test fixture scene (whose triangles came from the offline decoder)."""
    import oracle_binding as ob

    wd = pta.work_dispatcher
    scene = pta.scenes.load_fixture_scene("dragon_test", golden_dir)
    packed = wd.pack_scene(scene, ob.OracleBvhBuilder(), 8)
    s = _surface(golden_dir, "dragon")
    mesh = scene.meshes[0]
    mtl = scene.tri_set_material_indices(mesh, mesh.triangle_set)
    rec, aabbs, _ = rd.mesh_to_triangles(s, material_index=int(mtl[0]), texture_id=scene.get_texture_id(mesh.triangle_set.texture))
    assert np.all(mtl == mtl[0])
    tri_bytes = rec.tobytes()
    assert packed.triangles[: len(tri_bytes)] == tri_bytes
    ids = (np.arange(rec.shape[0], dtype=np.int32) + scene.tri_set_first_index(mesh.triangle_set)) + np.int32(3 << 24)
    mh = rd.bvh_build(mesh.bvh_order, mesh.bvh_type, aabbs, ids)
    th = rd.bvh_build(8, scene.default_bvh, np.zeros((0, 6), np.float32), np.zeros(0, np.int32), [mh], [mesh.transform.to_list()], [0])
    nodes, oid = rd.bvh_read(th)
    rd.free_rid(th)
    rd.free_rid(mh)
    assert wd.create_bvh_byte_array(nodes) == packed.bvh and np.array_equal(oid, np.frombuffer(packed.object_ids, np.int32))


def test_real_uvs_option(pta, golden_dir):
    """PTK_OPT_TRIANGLE_UVS: grimchild with a sampled texture; the corner UVs from ptk_mesh_to_triangles are bound at
    (4,1) and interpolated at the hit.  Off (the default) the image equals the reference's barycentric behaviour, on it
    equals the oracle run with the same table; the two differ."""
    import copy

    import oracle_binding as ob

    wd, sc, gm = pta.work_dispatcher, pta.scene, pta.godot_math
    s = _surface(golden_dir, "grimchild")
    rng = np.random.default_rng(21)
    scene = sc.PTScene(default_bvh=4)
    tex = sc.PTSampledTexture(image=rng.random((32, 32, 4), dtype=np.float32))
    mesh = sc.PTMesh(bvh_type=4, transform=gm.Transform3D(origin=(0.0, 0.0, 0.0)))
    mesh.scene = scene
    mesh.triangle_set = sc.PTTriangleSet(vertices=mo.triangle_vertices(s), texture=tex)
    scene.add_mesh(mesh)
    scene.camera = sc.PTCamera(transform=gm.Transform3D.looking_at((0.0, 0.0, 0.6), (0.0, -0.05, 0.0)), fov=40.0)
    packed = copy.copy(wd.pack_scene(scene, ob.OracleBvhBuilder(), 8))
    w, h = 320, 184
    imgs = {}
    for name, r in [("one-kernel", 1), ("wavefront", 2), ("persistent", 3)]:
        dev = pta.PtkRenderingDevice(0)
        try:
            dev.set_renderer(r)
            d = wd.PTWorkDispatcher(dev)
            d.create_buffers(w, h, packed=packed)
            d.load_shader()
            _, _, uvs = dev.mesh_to_triangles(s)
            dev.free_rid(d.triangle_uv_buffer)
            d.rids_to_free.remove(d.triangle_uv_buffer)
            d.triangle_uv_buffer = d._create_uniform(uvs.tobytes(), wd.TRIANGLE_SET_INDEX, wd.TRIANGLE_UV_BIND)
            win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=3, max_refraction_bounces=2)
            for on in (0, 1):
                dev.set_option(1, on)
                dev.texture_update(d.image_buffer, np.zeros((h, w, 4), np.float32))
                d.create_compute_list(win, scene.camera, wd.sample_time(2), sync=True)
                imgs[(name, on)] = dev.texture_get_data(d.image_buffer)
            d.free_rids()
        finally:
            dev.close()
    push = wd.push_constant_bytes(scene.camera, wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=3, max_refraction_bounces=2), wd.sample_time(2))
    for on in (0, 1):
        cpu = np.zeros((h, w, 4), np.float32)
        ob.render(packed, push, cpu, w // 8, h // 8, trace=False, triangle_uvs=mo.triangle_uvs(s) if on else None)
        for name in ("one-kernel", "wavefront", "persistent"):
            assert np.array_equal(imgs[(name, on)].view(np.uint32), cpu.view(np.uint32)), f"{name}, option {on}: differs from the oracle"
    assert not np.array_equal(imgs[("one-kernel", 0)], imgs[("one-kernel", 1)])
