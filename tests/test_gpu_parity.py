"""GPU parity: the HIP render path (through the C ABI) against the CPU oracle on the same bytes.

Bar: primary-hit object ids and per-pixel traversal counters bit-exact (integer work); radiance
within 1e-4 max-abs per channel before tonemapping -- the kernels are written so that the images
are in fact bit-identical, and the tests report how many texels differ at all."""
import numpy as np
import pytest

import oracle_binding as ob

pytestmark = pytest.mark.gpu

TOL = 1e-4  # stated tolerance: max abs per channel, linear values as stored in the rgba32f image


@pytest.fixture(scope="module")
def rd(pta):
    dev = pta.PtkRenderingDevice(0)
    yield dev
    dev.close()


@pytest.fixture(scope="module", params=["wavefront", "persistent"])
def rd_wf(pta, request):
    """A second context pinned to one of the two schedules that cover untraced plain dispatches: the
    wavefront pipeline or the persistent path kernel.  Every test that takes it runs once with each."""
    dev = pta.PtkRenderingDevice(0)
    dev.pinned = request.param
    dev.set_renderer(pta.rendering_device.PTK_RENDERER_WAVEFRONT if request.param == "wavefront"
                     else pta.rendering_device.PTK_RENDERER_PERSISTENT)
    yield dev
    dev.close()


def _compare_image(name, gpu_img, cpu_img):
    both_nan = np.isnan(gpu_img) & np.isnan(cpu_img)
    diff = np.where(both_nan, 0.0, np.abs(gpu_img.astype(np.float64) - cpu_img.astype(np.float64)))
    assert not np.isnan(diff).any(), f"{name}: NaN on one side only"
    nbits = np.count_nonzero((gpu_img.view(np.uint32) != cpu_img.view(np.uint32)) & ~both_nan)
    print(f"{name}: max abs diff {diff.max():.3e}, texel words differing bitwise: {nbits}")
    assert diff.max() <= TOL, f"{name}: radiance differs by {diff.max()}"
    assert nbits == 0, f"{name}: {nbits} words differ bitwise (within tolerance, but the paths are meant to be identical)"


def _compare(name, gpu_img, cpu_img, gpu_ids, cpu_ids, gpu_cnt, cpu_cnt):
    assert np.array_equal(gpu_ids, cpu_ids), f"{name}: {np.count_nonzero(gpu_ids != cpu_ids)} primary hit ids differ"
    assert np.array_equal(gpu_cnt, cpu_cnt), f"{name}: traversal counters differ in {np.count_nonzero((gpu_cnt != cpu_cnt).any(-1))} pixels"
    _compare_image(name, gpu_img, cpu_img)


def _run_case(pta, rd, rd_wf, golden_dir, name, w, h, depth, bounces, emission=False, dispatches=3, refraction_mode=2,
              early_t=False, scene=None, bvh_type=None, **win_kw):
    """Renders `dispatches` progressive samples three ways: CPU oracle, the one-kernel renderer (traced:
    primary ids + counters + image) and, where it applies and rd_wf is given, the wavefront renderer
    (image).  Everything must match the oracle."""
    wd = pta.work_dispatcher
    wf_applies = rd_wf is not None and not early_t and win_kw.get("render_mode", 0) not in (1, 2) and win_kw.get("use_bvh", True)
    if scene is None:
        scene = pta.scenes.load_fixture_scene(name, golden_dir)
    packed = wd.pack_scene(scene, ob.OracleBvhBuilder(), 8, bvh_type)
    disp = wd.PTWorkDispatcher(rd)
    disp.create_buffers(w, h, packed=packed)
    disp.load_shader(refraction_mode=refraction_mode, use_emission=emission, early_t_break=early_t)
    rd.trace_enable(3)
    disp_wf = None
    if wf_applies:
        disp_wf = wd.PTWorkDispatcher(rd_wf)
        disp_wf.create_buffers(w, h, packed=packed)
        disp_wf.load_shader(refraction_mode=refraction_mode, use_emission=emission, early_t_break=False)
    win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=depth, max_refraction_bounces=bounces, **win_kw)
    cpu_img = np.zeros((h, w, 4), np.float32)
    try:
        for k in range(dispatches):
            win.frame = k
            t = wd.sample_time(k)
            out = ob.render(packed, wd.push_constant_bytes(scene.camera, win, t), cpu_img, w // 8, h // 8,
                            refraction_mode=refraction_mode, use_emission=emission, early_t_break=early_t)
            disp.create_compute_list(win, scene.camera, t, sync=True)
            gpu_img = rd.texture_get_data(disp.image_buffer)
            _compare(f"{name}[{k}]", gpu_img, cpu_img, rd.trace_read_primary(disp.image_buffer), out["primary_id"],
                     rd.trace_read_counters(disp.image_buffer), out["pixel_stats"])
            if disp_wf is not None:
                disp_wf.create_compute_list(win, scene.camera, t, sync=True)
                assert (len(rd_wf.dispatch_traverse_ms(0)) > 0) == (rd_wf.pinned == "wavefront"), "the pinned renderer did not run"
                _compare_image(f"{name}[{k}] {rd_wf.pinned}", rd_wf.texture_get_data(disp_wf.image_buffer), cpu_img)
    finally:
        rd.trace_enable(0)
        disp.free_rids()
        if disp_wf is not None:
            disp_wf.free_rids()


def test_c1_sphere_scene1(pta, rd, rd_wf, golden_dir):
    """BASELINE config 1: sphere_scene1, 640x360, 1 spp, depth 4."""
    _run_case(pta, rd, rd_wf, golden_dir, "sphere_scene1", 640, 360, 4, 4, dispatches=2)


def test_c2_random_spheres(pta, rd, rd_wf, golden_dir):
    """BASELINE config 2 scene (487 spheres, diffuse/metal/glass, procedural textures), depth 8, reduced size."""
    _run_case(pta, rd, rd_wf, golden_dir, "benchmark_random_scene", 640, 360, 8, 4, dispatches=3)


def test_c3_cornell_emission(pta, rd, rd_wf, golden_dir):
    """BASELINE config 3a: fake_cornell_box with USE_EMISSION 1."""
    _run_case(pta, rd, rd_wf, golden_dir, "fake_cornell_box", 640, 360, 8, 8, emission=True, dispatches=3)


def test_c3_nested_dielectrics(pta, rd, rd_wf, golden_dir):
    """BASELINE config 3b: sphere_scene3 (priorities, air cut-outs)."""
    _run_case(pta, rd, rd_wf, golden_dir, "sphere_scene3", 640, 360, 8, 8, dispatches=3)


@pytest.mark.parametrize("mode", [0, 1])
def test_refraction_tracking_modes(pta, rd, rd_wf, golden_dir, mode):
    """use_refraction_tracking 0 and 1 (ray_tracer.comp:84)."""
    _run_case(pta, rd, rd_wf, golden_dir, "sphere_scene3", 320, 184, 6, 4, dispatches=2, refraction_mode=mode)


def test_c4_dragon_instanced_mesh(pta, rd, rd_wf, golden_dir):
    """BASELINE config 4: dragon (8224 triangles) as one instanced PTMesh, XYZ_SORTED order 8, depth 4."""
    _run_case(pta, rd, rd_wf, golden_dir, "dragon_test", 640, 360, 4, 4, dispatches=2)


def test_mixed_scene_with_meshes(pta, rd, rd_wf, golden_dir):
    """test_scene_bvh_b: two PTMeshes (spheres and triangles inside meshes) plus scene-level objects."""
    scene = pta.scenes.load_fixture_scene("test_scene_bvh_b", golden_dir)
    scene.camera = pta.scene.PTCamera(transform=pta.godot_math.Transform3D.looking_at((4.0, 3.0, 6.0), (0.0, 0.0, 0.0)), fov=60.0)
    _run_case(pta, rd, rd_wf, golden_dir, "test_scene_bvh_b", 320, 184, 5, 4, dispatches=2, scene=scene)


def test_early_t_break_variant(pta, rd, rd_wf, golden_dir):
    """use_early_t_break 1 (ray_tracer.comp:96,1128-1132,1280-1284)."""
    _run_case(pta, rd, rd_wf, golden_dir, "benchmark_random_scene", 320, 184, 6, 4, dispatches=2, early_t=True)
    _run_case(pta, rd, rd_wf, golden_dir, "dragon_test", 320, 184, 3, 4, dispatches=1, early_t=True)


def test_render_modes(pta, rd, rd_wf, golden_dir):
    """BVH heat view and normal view (ray_tracer.comp:1567-1570,1624-1632)."""
    _run_case(pta, rd, rd_wf, golden_dir, "benchmark_random_scene", 320, 184, 8, 4, dispatches=1, render_mode=1,
              display_node_count=True, display_object_count=True)
    _run_case(pta, rd, rd_wf, golden_dir, "sphere_scene3", 320, 184, 8, 4, dispatches=1, render_mode=2)


def test_brute_force_mode(pta, rd, rd_wf, golden_dir):
    """USE_BVH flag off: linear scans (ray_tracer.comp:1093-1101)."""
    _run_case(pta, rd, rd_wf, golden_dir, "sphere_scene1", 160, 96, 4, 4, dispatches=1, use_bvh=False)
    _run_case(pta, rd, rd_wf, golden_dir, "fake_cornell_box", 160, 96, 4, 4, dispatches=1, use_bvh=False)


def test_no_multisample_overwrites(pta, rd, rd_wf, golden_dir):
    _run_case(pta, rd, rd_wf, golden_dir, "sphere_scene1", 160, 96, 4, 4, dispatches=2, enable_multisampling=False)


def test_atrium_standin_small(pta, rd, rd_wf, golden_dir):
    """The Sponza stand-in generator at a size the oracle renders in seconds (16 384 triangles)."""
    scene = pta.scenes.atrium_scene(16384, seed=7)
    _run_case(pta, rd, rd_wf, golden_dir, "atrium16k", 320, 184, 8, 4, dispatches=2, scene=scene)


def test_random_spheres_10k(pta, rd, rd_wf, golden_dir):
    """Scaled variant of the random scene: ~10 000 spheres."""
    scene = pta.scenes.random_spheres_scene(grid=50, seed=3)
    _run_case(pta, rd, rd_wf, golden_dir, "spheres10k", 320, 184, 8, 4, dispatches=2, scene=scene)


def test_spheres_and_triangles_share_leaves(pta, rd, rd_wf, golden_dir):
    """Leaves of up to eight objects that mix spheres and triangles: with triangles in the scene, long leaves
    are shared between a lane and a helper lane, so sphere tests run on helpers too."""
    scene = pta.scenes.random_spheres_scene(grid=14, seed=11)
    rng = np.random.default_rng(5)
    mat = scene.spheres[0].material
    for _ in range(300):
        c = rng.uniform(-12.0, 12.0, 3).astype(np.float32)
        c[1] = abs(c[1]) * 0.15 + 0.2
        a, b = rng.uniform(-0.6, 0.6, (2, 3)).astype(np.float32)
        scene.add_object(pta.scene.PTTriangle(material=mat, vertex1=tuple(c), vertex2=tuple(c + a), vertex3=tuple(c + b)))
    _run_case(pta, rd, rd_wf, golden_dir, "spheres+triangles", 320, 184, 6, 4, dispatches=2, scene=scene)


def test_helped_leaves_handed_back(pta, golden_dir, monkeypatch):
    """The fallback of the shared leaves (a helper lane that cannot vouch for its half hands the whole leaf back to
    its owner) cannot be reached with real spheres; PTK_LEAF_REDO_ALL=1 takes it for every helped leaf.  All three
    renderers must still give the oracle's image."""
    monkeypatch.setenv("PTK_LEAF_REDO_ALL", "1")
    RD = pta.rendering_device
    scene = pta.scenes.atrium_scene(16384, seed=7)
    for renderer, pinned in ((RD.PTK_RENDERER_SINGLE_KERNEL, "tile"), (RD.PTK_RENDERER_WAVEFRONT, "wavefront"), (RD.PTK_RENDERER_PERSISTENT, "persistent")):
        a, b = pta.PtkRenderingDevice(0), pta.PtkRenderingDevice(0)
        try:
            b.pinned = pinned
            b.set_renderer(renderer)
            _run_case(pta, a, b, golden_dir, "atrium16k, leaves handed back", 320, 184, 6, 4, dispatches=1, scene=scene)
        finally:
            a.close()
            b.close()


@pytest.mark.parametrize("env", [
    {"PTK_WF_CARRY": "0"},
    {"PTK_WF_WG_CLAIM": "64", "PTK_WF_PARK_BELOW": "64"},
    {"PTK_WF_WG_CLAIM": "1024", "PTK_WF_MAX_LAG": "1", "PTK_WF_PARK_BELOW": "4"},
    {"PTK_WF_MAX_LAG": "6", "PTK_WF_PARK_BELOW": "40", "PTK_LEAF_SPLIT": "2"},
    {"PTK_LDS_WINDOW": "12", "PTK_LEAF_SPLIT": "7"},
    {"PTK_LDS_WINDOW": "64"},  # more than sixteen waves' windows fit a compute unit: held to what does
    {"PTK_LDS_WINDOW": "32"},
    {"PTK_WF_PARTS": "8", "PTK_WF_WG_CLAIM": "64"},  # a queue head per XCD, workgroups moving on to the next part
    {"PTK_WF_PARTS": "8", "PTK_WF_CARRY": "0"},
], ids=lambda e: ",".join(f"{k[4:].lower()}={v}" for k, v in e.items()))
def test_pipeline_tuning_variables_do_not_change_the_image(pta, golden_dir, monkeypatch, env):
    """The pipeline's scheduling knobs (chunk of the queue a workgroup takes, when and how far behind searches are
    parked, no carry-over at all, a short LDS stack window, where leaves are split) move work around; the image stays
    the oracle's."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    RD = pta.rendering_device
    scene = pta.scenes.atrium_scene(16384, seed=7)
    a, b = pta.PtkRenderingDevice(0), pta.PtkRenderingDevice(0)
    try:
        b.pinned = "wavefront"
        b.set_renderer(RD.PTK_RENDERER_WAVEFRONT)
        _run_case(pta, a, b, golden_dir, "atrium16k, tuned", 640, 360, 8, 4, dispatches=1, scene=scene)
    finally:
        a.close()
        b.close()


# ---- size-independent properties at full BASELINE sizes ------------------------------------------
@pytest.mark.parametrize("w,h", [(1920, 1080), (3840, 2160)])
def test_full_size_tiles_equal_whole(pta, rd, rd_wf, golden_dir, w, h):
    """Full BASELINE sizes (1080p and 4K): rendering the frame as four tiles (image_offset push
    constants) gives the same image as one dispatch, bit for bit (the seed depends only on absolute
    pixel coordinates and time)."""
    wd = pta.work_dispatcher
    scene = pta.scenes.load_fixture_scene("benchmark_random_scene", golden_dir)
    packed = wd.pack_scene(scene, ob.OracleBvhBuilder(), 8)
    disp = wd.PTWorkDispatcher(rd)
    disp.create_buffers(w, h, packed=packed)
    disp.load_shader()
    win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=8, max_refraction_bounces=4)
    imgs = []
    for _ in range(2):  # also: two runs are identical (determinism)
        win.frame = 0
        disp.create_compute_list(win, scene.camera, wd.sample_time(1), sync=True)
        imgs.append(rd.texture_get_data(disp.image_buffer))
    assert np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32))
    rd.texture_update(disp.image_buffer, np.zeros((h, w, 4), np.float32))
    gx, gy = w // 8, h // 8
    sx, sy = gx // 2, gy // 2 - 1  # an uneven split
    for (ox, oy, nx, ny) in [(0, 0, sx, sy), (sx, 0, gx - sx, sy), (0, sy, sx, gy - sy), (sx, sy, gx - sx, gy - sy)]:
        tile = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=8, max_refraction_bounces=4,
                                 x_offset=ox * 8, y_offset=oy * 8, work_group_width=nx, work_group_height=ny)
        tile.frame = 0
        disp.create_compute_list(tile, scene.camera, wd.sample_time(1), sync=True)
    tiled = rd.texture_get_data(disp.image_buffer)
    assert np.array_equal(tiled.view(np.uint32), imgs[0].view(np.uint32))
    # progressive accumulation: frames 0 and 1 overwrite, frame 2 is the mean of samples 1 and 2
    singles = []
    for k in (1, 2):
        win.frame = 0
        disp.create_compute_list(win, scene.camera, wd.sample_time(k), sync=True)
        singles.append(rd.texture_get_data(disp.image_buffer))
    for k in range(3):
        win.frame = k
        disp.create_compute_list(win, scene.camera, wd.sample_time(k), sync=True)
    acc = rd.texture_get_data(disp.image_buffer)
    expect = singles[0] * np.float32(0.5) + singles[1] * np.float32(0.5)
    assert np.array_equal(acc, expect)
    disp.free_rids()


def test_error_paths(pta, rd):
    """Bad arguments come back as errors, never as a crash."""
    with pytest.raises(pta.PtkError):
        rd.configure(bvh_order=1)
    with pytest.raises(pta.PtkError):
        rd.configure(proc_tex_paths=["res://user/custom.comp"])
    with pytest.raises(pta.PtkError):
        rd.buffer_update(987654, 0, 4, b"abcd")
    with pytest.raises(pta.PtkError):
        rd.compute_list_dispatch(bytes(128), 0, 1, 1)
    # a node buffer that points outside itself is rejected before any launch
    wd = pta.work_dispatcher
    import struct

    disp = wd.PTWorkDispatcher(rd)
    scene = pta.scene.PTScene()
    scene.add_object(pta.scene.PTSphere())
    scene.camera = pta.scene.PTCamera()
    packed = wd.pack_scene(scene, ob.OracleBvhBuilder(), 8)
    bad = bytearray(packed.bvh)
    struct.pack_into("<i", bad, 12, (3 << 24) | 5000)
    packed.bvh = bytes(bad)
    disp.create_buffers(64, 64, packed=packed)
    disp.load_shader()
    win = wd.PTRenderWindow(render_width=64, render_height=64)
    with pytest.raises(pta.PtkError):
        disp.create_compute_list(win, scene.camera, 0.0, sync=True)
    disp.free_rids()
    # a leaf id whose whole word is positive (index -1, count byte +1): the shader would read it as a leaf of
    # one, PTBVHTree never writes it; refused instead of being taken for an inner node
    bad = bytearray(pta.work_dispatcher.pack_scene(scene, ob.OracleBvhBuilder(), 8).bvh)
    struct.pack_into("<I", bad, 12, 0x01FFFFFF)
    packed.bvh = bytes(bad)
    disp = wd.PTWorkDispatcher(rd)
    disp.create_buffers(64, 64, packed=packed)
    disp.load_shader()
    with pytest.raises(pta.PtkError):
        disp.create_compute_list(win, scene.camera, 0.0, sync=True)
    disp.free_rids()
    rd.configure()
