"""Pins the CPU oracle against every vector the reference's own tests hold for this path
(tests/golden/reference_vectors.json, transcribed from tests/unit_tests/test_utils.gd and
tests/test_helpers/ptbvhtree_setup_helper.gd) and against hand-derived known answers."""
import json
import math
import os
import struct

import numpy as np
import pytest

import oracle_binding as ob


@pytest.fixture(scope="module")
def vectors(golden_dir):
    with open(os.path.join(golden_dir, "reference_vectors.json")) as f:
        return json.load(f)


def _a6(v):
    return np.array([float(x) for x in v], dtype=np.float32)


def test_is_aabb_valid_vectors(oracle, vectors):
    for case in vectors["is_aabb_valid"]:
        a = _a6(case["aabb"])
        assert bool(oracle.pbo_is_aabb_valid(a.ctypes.data)) == case["valid"], case["msg"]


def test_merge_aabb_vectors(oracle, vectors):
    for case in vectors["merge_aabb"]:
        a, b, out = _a6(case["a"]), _a6(case["b"]), np.zeros(6, dtype=np.float32)
        oracle.pbo_merge_aabb(a.ctypes.data, b.ctypes.data, out.ctypes.data)
        assert np.array_equal(out, _a6(case["out"])), case["msg"]


def missing_texture_scene(pta):
    """One white sphere whose sampled texture has no image (PTWorkDispatcher binds the 8x8 filler grid for it,
    pt_work_dispatcher.gd:350-370), seen in front of the sky."""
    sc, gm = pta.scene, pta.godot_math
    scene = sc.PTScene()
    scene.add_object(sc.PTSphere(radius=1.0, material=sc.PTMaterial(albedo=(1.0, 1.0, 1.0)), texture=sc.PTSampledTexture(image=None)))
    scene.camera = sc.PTCamera(transform=gm.Transform3D.looking_at((0.0, 0.0, 3.0), (0.0, 0.0, 0.0)), fov=50.0)
    return scene


def missing_texture_pixel_values(vectors):
    """The only three values a pixel of missing_texture_scene can take at depth 2, frame 0: sky, or sky times one
    of the two filler texels.  The texel bytes are the reference's (test_utils.gd:5-11: 56,16,56,56 and 0,0,0,56,
    DATA_FORMAT_R8G8B8A8_UNORM, i.e. byte / 255); the path multiplies 1 by (albedo 1 * texel), then by the sky
    colour (0.7, 0.7, 0.9, 1) of the second ray, which always leaves a convex object."""
    g = vectors["missing_texture_grid"]
    sky = np.array([0.7, 0.7, 0.9, 1.0], np.float32)
    unorm = lambda b: (np.array(b, np.float32) / np.float32(255.0)).astype(np.float32)
    return sky, (unorm(g["pink"]) * sky).astype(np.float32), (unorm(g["black"]) * sky).astype(np.float32)


def check_missing_texture_image(img, vectors):
    """The stored pixel is the path's colour with the shader's gamma (rgb ** camera.gamma, 1 / 2.2 by default, ray_tracer.comp:1682-1683;
    alpha as it is).  The power goes through the renderer's own pow, so rgb is compared within 2e-6 (and the
    image may hold three distinct values only); alpha, a plain byte / 255, bit for bit."""
    sky, pink, black = missing_texture_pixel_values(vectors)
    flat = img.reshape(-1, 4)
    assert np.unique(flat, axis=0).shape[0] == 3, "exactly three colours: sky and the two texels"

    def matches(v):
        rgb = np.power(v[:3].astype(np.float64), 1.0 / 2.2)
        return (np.abs(flat[:, :3] - rgb) <= 2e-6).all(1) & (flat[:, 3].view(np.uint32) == v[3:].view(np.uint32)[0])

    is_sky, is_pink, is_black = matches(sky), matches(pink), matches(black)
    assert (is_sky | is_pink | is_black).all(), f"{int((~(is_sky | is_pink | is_black)).sum())} pixels are none of sky / pink texel / black texel"
    # the sphere covers about a quarter of the frame; the 8x8 grid over (u, v) shows both texels in quantity
    assert is_pink.sum() > flat.shape[0] // 20 and is_black.sum() > flat.shape[0] // 20 and is_sky.sum() > flat.shape[0] // 4
    return int(is_pink.sum()), int(is_black.sum())


def test_missing_texture_texels_through_the_oracle(pta, vectors):
    """test_utils.gd:4-50 pins the filler grid's bytes and its parity rule (pink where (i%2 + j%2) % 2 == 0); here
    the oracle renders an object that samples it and every pixel must carry exactly those bytes."""
    g = vectors["missing_texture_grid"]
    assert g["pink"] == [56, 16, 56, 56] and g["black"] == [0, 0, 0, 56]
    wd = pta.work_dispatcher
    scene = missing_texture_scene(pta)
    packed = wd.pack_scene(scene, ob.OracleBvhBuilder(), 8)
    w, h = 96, 64
    img = np.zeros((h, w, 4), np.float32)
    win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=2, max_refraction_bounces=0)
    win.frame = 0
    ob.render(packed, wd.push_constant_bytes(scene.camera, win, wd.sample_time(0)), img, w // 8, h // 8)
    check_missing_texture_image(img, vectors)


# ---- the three hand-built trees -----------------------------------------------------------
def _objects_of(scene, sc):
    meshes = scene.meshes
    table = {
        "sphere": scene.scene_spheres, "triangle": scene.scene_triangles,
    }
    for mi, m in enumerate(meshes):
        table[f"mesh{mi}_sphere"] = [o for o in m.objects if o.get_type() == sc.SPHERE]
        table[f"mesh{mi}_triangle"] = [o for o in m.objects if o.get_type() == sc.TRIANGLE]
    return table


def _assemble(oracle, pta, golden_dir, tree):
    sc = pta.scene
    scene = pta.scenes.load_fixture_scene(tree["scene"], golden_dir)
    table = _objects_of(scene, sc)
    nodes = tree["nodes"]
    kinds = np.array([{"inner": 0, "leaf": 1, "socket": 2}[n["kind"]] for n in nodes], dtype=np.int32)
    parents = np.array([n["parent"] for n in nodes], dtype=np.int32)
    aabbs, ids, first, count = [], [], [], []
    for n in nodes:
        first.append(len(ids))
        objs = n.get("objects", [])
        count.append(len(objs))
        for kind, idx in objs:
            o = table[kind][idx]
            aabbs.append(o.get_bvh_aabb().to6())
            ids.append(sc.make_object_id(scene.get_object_index(o), o.get_type()))
    aabbs = np.array(aabbs, dtype=np.float32)
    ids = np.array(ids, dtype=np.int32)
    first, count = np.array(first, dtype=np.int32), np.array(count, dtype=np.int32)
    names = [n["name"] for n in nodes]
    leaf_order = np.array([names.index(x) for x in tree["leaf_nodes"]], dtype=np.int32)
    xf = np.zeros((len(nodes), 12), dtype=np.float32)
    mesh_index = np.full(len(nodes), -1, dtype=np.int32)
    for i, n in enumerate(nodes):
        if n["kind"] == "socket":
            xf[i] = scene.meshes[n["mesh"]].transform.to_list()
            mesh_index[i] = n["mesh"]
    t = oracle.pbo_assemble(tree["order"], len(nodes), kinds.ctypes.data, parents.ctypes.data, first.ctypes.data,
                            count.ctypes.data, len(ids), aabbs.ctypes.data, ids.ctypes.data, leaf_order.ctypes.data,
                            len(leaf_order), xf.ctypes.data, mesh_index.ctypes.data)
    assert t
    n = oracle.pbo_node_count(t)
    nb = np.zeros(n * 32, dtype=np.uint8)
    out_ids = np.zeros(max(1, len(ids)), dtype=np.int32)
    assert oracle.pbo_flatten(t, nb.ctypes.data, out_ids.ctypes.data) == 0
    oracle.pbo_free(t)
    return scene, names, ids, nb.tobytes(), out_ids[: len(ids)]


def _decode(nb):
    out = []
    for i in range(len(nb) // 32):
        mn = struct.unpack_from("<3f", nb, i * 32)
        nid, = struct.unpack_from("<i", nb, i * 32 + 12)
        mx = struct.unpack_from("<3f", nb, i * 32 + 16)
        xf, = struct.unpack_from("<i", nb, i * 32 + 28)
        idx = ((nid & 0xFFFFFF) ^ 0x800000) - 0x800000
        size = abs((((nid >> 24) & 0xFF) ^ 0x80) - 0x80)
        out.append(dict(min=mn, max=mx, node_id=nid, index=idx, size=size, xform=xf))
    return out


@pytest.mark.parametrize("which", ["a1", "a2", "b"])
def test_fixture_tree_layout(oracle, pta, golden_dir, vectors, which):
    tree = vectors["trees"][which]
    scene, names, ids, nb, out_ids = _assemble(oracle, pta, golden_dir, tree)
    nodes = _decode(nb)
    # node order = the fixture's bvh_list (ptbvhtree_setup_helper.gd)
    assert len(nodes) == len(tree["bvh_list"])
    order = [names.index(x) for x in tree["bvh_list"]]
    # every inner node points at its first child, children are consecutive (ptbvhtree_helper.gd:69-94)
    for pos, ni in enumerate(order):
        spec = tree["nodes"][ni]
        kids = [k for k, n in enumerate(tree["nodes"]) if n["parent"] == ni]
        if spec["kind"] != "leaf":
            assert nodes[pos]["size"] == len(kids)
            child_pos = [order.index(k) for k in kids]
            assert child_pos == list(range(nodes[pos]["index"], nodes[pos]["index"] + len(kids)))
            assert nodes[pos]["index"] > 0
            assert nodes[pos]["xform"] == (spec["mesh"] if spec["kind"] == "socket" else -1)
        else:
            assert nodes[pos]["size"] == len(spec["objects"])
            assert nodes[pos]["index"] <= -1
            assert nodes[pos]["xform"] == -1
    # object ids: leaf_nodes order, each leaf's objects consecutive and in order
    table = _objects_of(scene, pta.scene)
    expect = []
    leaf_start = {}
    for lname in tree["leaf_nodes"]:
        spec = tree["nodes"][names.index(lname)]
        leaf_start[lname] = len(expect)
        for kind, idx in spec["objects"]:
            o = table[kind][idx]
            expect.append(pta.scene.make_object_id(scene.get_object_index(o), o.get_type()))
    assert out_ids.tolist() == expect
    for pos, ni in enumerate(order):
        spec = tree["nodes"][ni]
        if spec["kind"] == "leaf":
            assert -nodes[pos]["index"] - 1 == leaf_start[spec["name"]]
    # counts the fixture states
    leaves = sum(1 for n in tree["nodes"] if n["kind"] == "leaf")
    assert leaves == tree["leaf_count"]
    assert len(expect) == tree["object_count"] + tree["mesh_object_count"]
    # boxes: a parent's padded box holds its children's boxes (same padding on both)
    for pos, ni in enumerate(order):
        spec = tree["nodes"][ni]
        if spec["kind"] == "inner":
            for k in range(nodes[pos]["size"]):
                c = nodes[nodes[pos]["index"] + k]
                assert all(c["min"][a] >= nodes[pos]["min"][a] - 1e-5 for a in range(3))
                assert all(c["max"][a] <= nodes[pos]["max"][a] + 1e-5 for a in range(3))


# ---- hand-derived known answers --------------------------------------------------------------
def _u32(x):
    return x & 0xFFFFFFFF


def _f32bits(f):
    return struct.unpack("<I", struct.pack("<f", f))[0]


def _py_hash(x, y, z):
    px, py, pz = _f32bits(x), _f32bits(y), _f32bits(z)
    qx = _u32(1103515245 * ((px >> 1) ^ py))
    qy = _u32(1103515245 * ((py >> 1) ^ pz))
    qz = _u32(1103515245 * ((pz >> 1) ^ px))
    h32 = _u32(1103515245 * ((qx ^ qz) ^ (qy >> 3)))
    n = h32 ^ (h32 >> 16)
    return float(np.float32(np.float32(n) * np.float32(1.0 / 2147483648.0)))


def _py_pcg(seed):
    state = _u32(_f32bits(seed) * 747796405 + 2891336453)
    word = _u32(((state >> ((state >> 28) + 4)) ^ state) * 277803737)
    r = (word >> 22) ^ word
    return float(np.float32(np.float32(r) / np.float32(4294967295.0)))


def test_hash_known_answers(oracle):
    """shaders/ray_tracer.comp:500-507 is pure u32 arithmetic; re-derived here with Python integers."""
    for x, y, z in [(0.0, 0.0, 0.0), (1.0, 2.0, 0.00016), (1919.0, 1079.0, 0.09984), (17.0, 300.0, 0.0)]:
        z32 = float(np.float32(z))
        assert oracle.pto_hash(x, y, z32) == _py_hash(x, y, z32)
    assert 0.0 <= oracle.pto_hash(5.0, 7.0, 0.0) <= 2.0


def test_pcg_known_answers(oracle):
    """shaders/ray_tracer.comp:476-481: PCG on the seed's bit pattern, seed post-incremented by 1.0."""
    import ctypes as C

    for s0 in [0.0, 0.5, 1.25, 1.9999, 42.0]:
        seed = C.c_float(s0)
        for k in range(6):
            expect = _py_pcg(float(np.float32(s0) + np.float32(k)))
            got = oracle.pto_pcg_next(C.byref(seed))
            assert got == expect
            assert seed.value == float(np.float32(s0) + np.float32(k + 1))
            assert 0.0 <= got <= 1.0


def test_rand_and_noise_range(oracle):
    import ctypes as C

    assert 0.0 <= oracle.pto_rand(0.03125) < 1.0
    seed = C.c_float(0.731)
    vals = [oracle.pto_noise_next(C.byref(seed)) for _ in range(64)]
    assert all(0.0 <= v < 1.0 for v in vals) and len(set(vals)) > 32


def test_math_accuracy(oracle):
    """The oracle's own sin/cos/acos/atan/pow stay within a few ulp of libm (they are definitions,
    not approximations under test, but a gross error would make the images meaningless)."""
    rng = np.random.default_rng(1)

    def probe(fn, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b if b is not None else np.zeros_like(a), dtype=np.float32)
        out = np.zeros_like(a)
        oracle.pto_math_probe(fn, a.size, a.ctypes.data, b.ctypes.data, out.ctypes.data)
        return out

    x = rng.uniform(-100, 100, 20000).astype(np.float32)
    assert np.max(np.abs(probe(0, x) - np.sin(x.astype(np.float64)))) < 2e-6
    assert np.max(np.abs(probe(1, x) - np.cos(x.astype(np.float64)))) < 2e-6
    u = rng.uniform(-1, 1, 20000).astype(np.float32)
    assert np.max(np.abs(probe(2, u) - np.arccos(u.astype(np.float64)))) < 2e-6
    y = rng.uniform(-5, 5, 20000).astype(np.float32)
    assert np.max(np.abs(probe(3, y, x) - np.arctan2(y.astype(np.float64), x.astype(np.float64)))) < 2e-6
    p = rng.uniform(1e-6, 20, 20000).astype(np.float32)
    g = np.full_like(p, 1 / 2.2)
    ref = np.power(p.astype(np.float64), 1 / 2.2)
    assert np.max(np.abs(probe(6, p, g) - ref) / ref) < 2e-6
    assert probe(6, np.array([0.0], np.float32), np.array([0.4545], np.float32))[0] == 0.0
    assert probe(6, np.array([1.0], np.float32), np.array([5.0], np.float32))[0] == 1.0


def test_godot_sort_matches_sorted_order(oracle):
    """SortArray restatement: any comparison sort gives the same order when keys are distinct
    (descending), and ties stay a permutation of the tied group."""
    rng = np.random.default_rng(3)
    for n in [0, 1, 2, 15, 16, 17, 33, 100, 1000, 5000]:
        keys = rng.permutation(n).astype(np.float32)
        perm = np.zeros(max(n, 1), dtype=np.int32)
        oracle.pbo_godot_sort_desc(n, keys.ctypes.data, perm.ctypes.data)
        assert keys[perm[:n]].tolist() == sorted(keys.tolist(), reverse=True)
    keys = rng.integers(0, 7, 3000).astype(np.float32)
    perm = np.zeros(3000, dtype=np.int32)
    oracle.pbo_godot_sort_desc(3000, keys.ctypes.data, perm.ctypes.data)
    assert sorted(perm.tolist()) == list(range(3000))
    assert np.all(np.diff(keys[perm]) <= 0)


# ---- analytic single-primitive scenes ----------------------------------------------------------
def _render(pta, scene, w, h, depth=1, **kw):
    wd = pta.work_dispatcher
    packed = wd.pack_scene(scene, ob.OracleBvhBuilder(), 8)
    win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=depth, **kw)
    img = np.zeros((h, w, 4), np.float32)
    out = ob.render(packed, wd.push_constant_bytes(scene.camera, win, 0.0), img, w // 8, h // 8)
    return img, out


def test_single_sphere_primary_hits(pta):
    """A unit sphere at the origin seen from z=+5: a pixel hits iff its ray passes within radius 1."""
    sc, gm = pta.scene, pta.godot_math
    scene = sc.PTScene()
    scene.add_object(sc.PTSphere(transform=gm.Transform3D(origin=(0, 0, 0)), radius=1.0, material=sc.PTMaterial(albedo=(0.5, 0.5, 0.5))))
    scene.camera = sc.PTCamera(transform=gm.Transform3D(origin=(0, 0, 5)), fov=40.0)
    w, h = 64, 64
    img, out = _render(pta, scene, w, h)
    vw, vh = scene.camera.viewport()
    ids = out["primary_id"]
    sphere_id = sc.make_object_id(0, sc.SPHERE)
    for py in range(0, h, 3):
        for px in range(0, w, 3):
            # pixel centre on the viewport plane at distance 1 (jitter is < half a pixel)
            x = (px + 0.5) / w * vw - vw / 2
            y = -((py + 0.5) / h * vh - vh / 2)
            d = np.array([x, y, -1.0])
            d /= np.linalg.norm(d)
            o = np.array([0, 0, 5.0])
            b = np.dot(o, d)
            dist2 = np.dot(o, o) - b * b
            if abs(math.sqrt(dist2) - 1.0) > 0.15:  # jitter moves a ray by up to ~0.7 pixel
                assert (ids[py, px] == sphere_id) == (dist2 < 1.0), (px, py)
    # depth 1: hit pixels keep albedo^gamma, misses the sky colour^gamma
    sky = np.array([0.7, 0.7, 0.9]) ** (1 / 2.2)
    assert np.allclose(img[0, 0, :3], sky, atol=1e-5)
    hit = np.argwhere(ids == sphere_id)[0]
    assert np.allclose(img[hit[0], hit[1], :3], 0.5 ** (1 / 2.2), atol=1e-5)


def test_plane_distance_and_depth_exhaustion(pta):
    """A ground plane y=-1: pixels below the horizon hit it; a path that runs out of depth keeps its
    throughput (no black), ray_tracer.comp:1573-1615."""
    sc, gm = pta.scene, pta.godot_math
    scene = sc.PTScene()
    scene.add_object(sc.PTPlane(normal=(0, 1, 0), distance=-1.0, material=sc.PTMaterial(albedo=(0.25, 0.5, 1.0))))
    # the BVH needs at least one object; park a tiny sphere far away
    scene.add_object(sc.PTSphere(transform=gm.Transform3D(origin=(0, 1000, 0)), radius=0.1))
    scene.camera = sc.PTCamera(transform=gm.Transform3D(origin=(0, 0, 0)), fov=60.0)
    img, out = _render(pta, scene, 32, 32)
    ids = out["primary_id"]
    plane_id = sc.make_object_id(0, sc.PLANE)
    assert np.all(ids[20:, :] == plane_id) and np.all(ids[:12, :] == -1)
    assert np.allclose(img[31, 5, :3], np.array([0.25, 0.5, 1.0]) ** (1 / 2.2), atol=1e-5)


def test_progressive_accumulation_rule(pta):
    """mix(prev, new, 1/max(1, frame)): frames 0 and 1 overwrite, frame 2 averages (ray_tracer.comp:1686-1693)."""
    sc, gm = pta.scene, pta.godot_math
    wd = pta.work_dispatcher
    scene = sc.PTScene()
    scene.add_object(sc.PTSphere(transform=gm.Transform3D(origin=(0, 0, 0)), radius=1.0))
    scene.camera = sc.PTCamera(transform=gm.Transform3D(origin=(0, 0, 5)), fov=40.0)
    packed = wd.pack_scene(scene, ob.OracleBvhBuilder(), 8)
    win = wd.PTRenderWindow(render_width=16, render_height=16, max_ray_depth=3)
    singles = []
    for k in range(3):
        img = np.zeros((16, 16, 4), np.float32)
        win.frame = 0
        ob.render(packed, wd.push_constant_bytes(scene.camera, win, wd.sample_time(k)), img, 2, 2)
        singles.append(img)
    acc = np.full((16, 16, 4), 7.0, np.float32)
    for k in range(3):
        win.frame = k
        ob.render(packed, wd.push_constant_bytes(scene.camera, win, wd.sample_time(k)), acc, 2, 2)
    expect = singles[1] * np.float32(0.5) + singles[2] * np.float32(0.5)
    assert np.array_equal(acc, expect)
