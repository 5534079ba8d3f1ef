"""GPU parity of the BVH builder: node bytes and object id table bit-exact against the CPU
restatement, through the C ABI (ptk_bvh_build / ptk_bvh_read)."""
import numpy as np
import pytest

import oracle_binding as ob

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rd(pta):
    dev = pta.PtkRenderingDevice(0)
    yield dev
    dev.close()


def _both(pta, rd, scene, order=8, bvh_type=None):
    wd = pta.work_dispatcher
    cpu_nodes, cpu_ids = wd.create_scene_bvh(scene, ob.OracleBvhBuilder(), order, bvh_type)
    gpu_nodes, gpu_ids = wd.create_scene_bvh(scene, wd.DeviceBvhBuilder(rd), order, bvh_type)
    return cpu_nodes, cpu_ids, gpu_nodes, gpu_ids


def _check(name, cpu_nodes, cpu_ids, gpu_nodes, gpu_ids):
    assert len(gpu_nodes) == len(cpu_nodes), f"{name}: node count {len(gpu_nodes) // 32} vs {len(cpu_nodes) // 32}"
    assert np.array_equal(np.asarray(gpu_ids), np.asarray(cpu_ids)), f"{name}: object id order differs"
    if gpu_nodes != cpu_nodes:
        a = np.frombuffer(gpu_nodes, dtype=np.uint32).reshape(-1, 8)
        b = np.frombuffer(cpu_nodes, dtype=np.uint32).reshape(-1, 8)
        bad = np.argwhere((a != b).any(1))[:5].ravel().tolist()
        raise AssertionError(f"{name}: node bytes differ, first nodes {bad}")
    print(f"{name}: {len(cpu_nodes) // 32} nodes, {len(cpu_ids)} ids bit-exact")


@pytest.mark.parametrize("name", ["sphere_scene1", "benchmark_random_scene", "fake_cornell_box", "sphere_scene3",
                                  "dragon_test", "test_scene_bvh_a", "test_scene_bvh_b"])
@pytest.mark.parametrize("bvh_type", [1, 2, 3, 4])
def test_fixture_scenes_all_types(pta, rd, golden_dir, name, bvh_type):
    """Every fixture scene with X / Y / Z / longest-axis trees.  The random scene's Y tree sorts 484
    equal keys (every small sphere sits at y = 0.2): the order then comes from the sort's tie handling."""
    scene = pta.scenes.load_fixture_scene(name, golden_dir)
    if bvh_type != 1:
        for m in scene.meshes:
            m.bvh_type = bvh_type
    _check(f"{name}/type{bvh_type}", *_both(pta, rd, scene, 8, bvh_type))


@pytest.mark.parametrize("order", [2, 3, 4, 8])
def test_orders(pta, rd, golden_dir, order):
    scene = pta.scenes.load_fixture_scene("benchmark_random_scene", golden_dir)
    _check(f"random/order{order}", *_both(pta, rd, scene, order, 4))


def _grid_triangles(n_side, seed):
    """Axis-aligned grid of triangles with heavy exact ties in every centre coordinate."""
    rng = np.random.default_rng(seed)
    i, j, k = np.meshgrid(np.arange(n_side), np.arange(n_side), np.arange(3), indexing="ij")
    base = np.stack([i, k * 0.5, j], axis=-1).reshape(-1, 3).astype(np.float32)
    e1 = np.array([0.75, 0.0, 0.0], np.float32)
    e2 = np.array([0.0, 0.25, 0.75], np.float32)
    tris = np.stack([base, base + e1, base + e2], axis=1)
    return tris[rng.permutation(tris.shape[0])]


@pytest.mark.parametrize("bvh_type", [1, 4])
def test_tie_heavy_grid(pta, rd, bvh_type):
    """3 x 40 x 40 grid: thousands of equal sort keys; exercises the emulated unstable sort on long
    runs of ties, and the single-occupied-bin halving."""
    sc = pta.scene
    scene = sc.PTScene(default_bvh=bvh_type)
    mesh = sc.PTMesh(bvh_type=bvh_type)
    mesh.scene = scene
    mesh.triangle_set = sc.PTTriangleSet(vertices=_grid_triangles(40, 5))
    scene.add_mesh(mesh)
    scene.add_object(sc.PTSphere(radius=0.5))
    _check(f"grid/type{bvh_type}", *_both(pta, rd, scene, 8, bvh_type))


def test_identical_objects_depth_limit(pta, rd):
    """200 identical spheres: every split lands in one bin, lists are halved until they fit."""
    sc, gm = pta.scene, pta.godot_math
    scene = sc.PTScene()
    for _ in range(200):
        scene.add_object(sc.PTSphere(transform=gm.Transform3D(origin=(1.0, 2.0, 3.0)), radius=0.25))
    _check("identical", *_both(pta, rd, scene, 8, 4))


@pytest.mark.parametrize("step", [0, 5])
def test_sort_depth_budget_runs_out(pta, rd, step):
    """The 4099-sphere random scene's X keys drive SortArray::introsort through its whole depth budget (median-of-3
    keeps picking near-extreme pivots), so a 1679-element segment ends in the heapsort fallback (partial_sort).
    step 0: all keys distinct (the device sorts such a segment by counting); step 5: every fifth sphere sits on
    its neighbour's centre, so the segment holds equal keys and the heapsort's own order among them shows."""
    sc, gm = pta.scene, pta.godot_math
    src = pta.scenes.random_spheres_scene(grid=32, seed=5)
    scene = sc.PTScene(default_bvh=sc.BVH_X_SORTED)
    spheres = list(src.scene_spheres)
    for i, s in enumerate(spheres):
        twin = spheres[i - 1] if step and i % step == 1 else s
        scene.add_object(sc.PTSphere(transform=gm.Transform3D(origin=tuple(float(v) for v in twin.position)), radius=float(twin.radius)))
    for t in (1, 4):
        _check(f"depth-budget/step{step}/type{t}", *_both(pta, rd, scene, 8, t))


def test_atrium_64k(pta, rd):
    """65 536-triangle stand-in: deep recursion, large sorts, one mesh socket."""
    scene = pta.scenes.atrium_scene(65536, seed=7)
    _check("atrium64k", *_both(pta, rd, scene, 8, 4))


def test_atrium_full_size(pta, rd):
    """The full 262 144-triangle benchmark scene."""
    scene = pta.scenes.atrium_scene(262144, seed=7)
    _check("atrium262k", *_both(pta, rd, scene, 8, 4))


def test_many_meshes(pta, rd):
    """Several instanced meshes next to scene-level objects: sockets inside terminal lists."""
    sc, gm = pta.scene, pta.godot_math
    rng = np.random.default_rng(11)
    scene = sc.PTScene(default_bvh=4)
    for i in range(5):
        scene.add_object(sc.PTSphere(transform=gm.Transform3D(origin=tuple(rng.uniform(-6, 6, 3))), radius=float(rng.uniform(0.2, 1.0))))
    for m in range(11):
        rot = gm.Transform3D.looking_at(tuple(rng.uniform(-8, 8, 3)), tuple(rng.uniform(-1, 1, 3)))
        mesh = sc.PTMesh(transform=rot, bvh_type=4)
        mesh.scene = scene
        base = rng.uniform(-1, 1, (300, 1, 3)).astype(np.float32)
        mesh.triangle_set = sc.PTTriangleSet(vertices=(base + rng.normal(0, 0.1, (300, 3, 3))).astype(np.float32))
        scene.add_mesh(mesh)
    _check("meshes", *_both(pta, rd, scene, 8, 4))
    _check("meshes/x", *_both(pta, rd, scene, 8, 1))


def test_builder_errors(pta, rd):
    with pytest.raises(pta.PtkError):
        rd.bvh_build(8, 1, np.zeros((0, 6), np.float32), np.zeros(0, np.int32))
    with pytest.raises(pta.PtkError):
        rd.bvh_build(1, 1, np.ones((2, 6), np.float32), np.zeros(2, np.int32))
    with pytest.raises(pta.PtkError):
        rd.bvh_build(8, 9, np.ones((2, 6), np.float32), np.zeros(2, np.int32))


def test_scene_tree_outlives_its_mesh_handles(pta, rd):
    """A scene tree shares ownership of the mesh trees merged into it: freeing a mesh's handle before the
    scene tree is read changes nothing, and a mesh tree with mesh sockets of its own is refused."""
    sc = pta.scene
    rng = np.random.default_rng(2)
    tris = (rng.uniform(-1, 1, (500, 1, 3)) + rng.normal(0, 0.1, (500, 3, 3))).astype(np.float32)
    ts = sc.PTTriangleSet(vertices=tris)
    aabbs = ts.bvh_aabbs()
    ids = (np.arange(500, dtype=np.int32)) + np.int32(3 << 24)
    sph_aabb = np.array([[2.0, 2.0, 2.0, 1.0, 1.0, 1.0]], np.float32)
    sph_id = np.array([1 << 24], np.int32)
    t12 = [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0, 0.5, 0.0, -0.5]

    oracle = ob.OracleBvhBuilder()
    om = oracle.build(8, 4, aabbs, ids)
    ot = oracle.build(8, 1, sph_aabb, sph_id, [om], [t12], [0])
    cpu_nodes, cpu_ids = oracle.read(ot)
    oracle.release(ot)

    mesh = rd.bvh_build(8, 4, aabbs, ids)
    tree = rd.bvh_build(8, 1, sph_aabb, sph_id, [mesh], [t12], [0])
    with pytest.raises(pta.PtkError):
        rd.bvh_build(8, 1, sph_aabb, sph_id, [tree], [t12], [0])  # a tree with sockets cannot be a mesh
    rd.free_rid(mesh)  # the scene tree keeps the mesh tree alive
    gpu_nodes, gpu_ids = rd.bvh_read(tree)
    _check("mesh handle freed first", cpu_nodes, cpu_ids, gpu_nodes, gpu_ids)
    gpu_nodes, gpu_ids = rd.bvh_read(tree)  # and again: reading is repeatable
    _check("second read", cpu_nodes, cpu_ids, gpu_nodes, gpu_ids)
    rd.free_rid(tree)


# ---- PTK_BVH_SAH: the builder type the reference leaves as an empty stub (pt_bvh_sweep_sah.gd) ------------------
def _sah_both(pta, rd, scene):
    for m in scene.meshes:
        m.bvh_type = 5
    return _both(pta, rd, scene, 8, 5)


@pytest.mark.parametrize("name", ["sphere_scene1", "benchmark_random_scene", "fake_cornell_box", "sphere_scene3", "dragon_test",
                                  "test_scene_bvh_a", "test_scene_bvh_b"])
def test_sah_fixture_scenes(pta, rd, golden_dir, name):
    """Binned-SAH trees: node bytes and id table bit-exact against the CPU restatement of the same rule."""
    _check(f"{name}/sah", *_sah_both(pta, rd, pta.scenes.load_fixture_scene(name, golden_dir)))


def test_sah_tie_heavy_grid_and_identical_objects(pta, rd):
    sc, gm = pta.scene, pta.godot_math
    scene = sc.PTScene(default_bvh=5)
    mesh = sc.PTMesh(bvh_type=5)
    mesh.scene = scene
    mesh.triangle_set = sc.PTTriangleSet(vertices=_grid_triangles(40, 5))
    scene.add_mesh(mesh)
    for _ in range(100):  # no spread at all: lists are cut in the middle
        scene.add_object(sc.PTSphere(transform=gm.Transform3D(origin=(1.0, 2.0, 3.0)), radius=0.25))
    _check("grid+identical/sah", *_sah_both(pta, rd, scene))


def test_sah_atrium_full_size_and_many_meshes(pta, rd):
    _check("atrium262k/sah", *_sah_both(pta, rd, pta.scenes.atrium_scene(262144, seed=7)))
    sc, gm = pta.scene, pta.godot_math
    rng = np.random.default_rng(11)
    scene = sc.PTScene(default_bvh=5)
    for i in range(5):
        scene.add_object(sc.PTSphere(transform=gm.Transform3D(origin=tuple(rng.uniform(-6, 6, 3))), radius=float(rng.uniform(0.2, 1.0))))
    for m in range(11):
        mesh = sc.PTMesh(transform=gm.Transform3D.looking_at(tuple(rng.uniform(-8, 8, 3)), tuple(rng.uniform(-1, 1, 3))), bvh_type=5)
        mesh.scene = scene
        base = rng.uniform(-1, 1, (300, 1, 3)).astype(np.float32)
        mesh.triangle_set = sc.PTTriangleSet(vertices=(base + rng.normal(0, 0.1, (300, 3, 3))).astype(np.float32))
        scene.add_mesh(mesh)
    _check("meshes/sah", *_sah_both(pta, rd, scene))
    with pytest.raises(pta.PtkError):
        rd.bvh_build(4, 5, np.ones((20, 6), np.float32), np.arange(20, dtype=np.int32))  # SAH is defined for order 8


def test_sah_tree_renders_the_same_image_with_less_work(pta, golden_dir):
    """The renderers run unchanged over a SAH tree: same image as the oracle over the same bytes (all schedules),
    and fewer box + object tests per ray than the reference's X-sorted tree on the random-spheres scene."""
    from test_gpu_parity import _compare_image

    wd = pta.work_dispatcher
    scene = pta.scenes.load_fixture_scene("benchmark_random_scene", golden_dir)
    w, h = 320, 184
    work = {}
    for typ in (1, 5):
        dev = pta.PtkRenderingDevice(0)
        try:
            disp = wd.PTWorkDispatcher(dev)
            disp.set_scene(scene)
            disp.create_buffers(w, h, order=8, bvh_type=typ)  # built and bound on the GPU
            disp.load_shader()
            win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=8, max_refraction_bounces=4)
            cpu = np.zeros((h, w, 4), np.float32)
            out = ob.render(disp.packed, wd.push_constant_bytes(scene.camera, win, wd.sample_time(1)), cpu, w // 8, h // 8)
            for r in (1, 2, 3):
                dev.set_renderer(r)
                dev.texture_update(disp.image_buffer, np.zeros((h, w, 4), np.float32))
                disp.create_compute_list(win, scene.camera, wd.sample_time(1), sync=True)
                _compare_image(f"type{typ} renderer{r}", dev.texture_get_data(disp.image_buffer), cpu)
            t = dict(zip(ob.STAT_NAMES, out["totals"].tolist()))
            work[typ] = (t["box_tests"] + t["obj_tests"]) / t["rays"]
            disp.free_rids()
        finally:
            dev.close()
    print("box + object tests per ray: X-sorted %.1f, SAH %.1f" % (work[1], work[5]))
    assert work[5] < 0.7 * work[1]
