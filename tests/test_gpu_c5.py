"""GPU parity at the size of the benchmarked configuration: the 262 144-triangle Sponza stand-in at
1920x1080 and 3840x2160, all three schedules, checked against the CPU oracle on windows of the frame;
a hand-made tree whose traversal stack outgrows the LDS window; and a long progressive run (257
dispatches, the length of BASELINE config 3).  Same bar as test_gpu_parity: ids and counters
bit-exact, images bit-identical (stated tolerance 1e-4)."""
import copy
import struct

import numpy as np
import pytest

import oracle_binding as ob
from test_gpu_parity import _compare, _compare_image

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def atrium_packed(pta):
    """The bench scene, packed once: BVH from the CPU oracle's builder (the HIP builder's output for this
    scene is compared with it byte for byte in test_gpu_bvh.test_atrium_full_size)."""
    wd = pta.work_dispatcher
    scene = pta.scenes.atrium_scene(262144, seed=7)
    return scene, wd.pack_scene(scene, ob.OracleBvhBuilder(), 8)


def _schedules(pta):
    r = pta.rendering_device
    return [("one-kernel", r.PTK_RENDERER_SINGLE_KERNEL), ("wavefront", r.PTK_RENDERER_WAVEFRONT),
            ("persistent", r.PTK_RENDERER_PERSISTENT)]


@pytest.mark.parametrize("w,h", [(1920, 1080), (3840, 2160)])
def test_c5_atrium_full_size(pta, atrium_packed, w, h):
    """Config C5's scene at its own sizes: one sample (depth 8) rendered by each schedule gives the same
    bits, and four 128x128 windows of it (image_offset push constants, corners and centre regions) equal
    the oracle's render of those windows; the traced tile kernel's primary ids and counters equal the
    oracle's there too."""
    wd = pta.work_dispatcher
    scene, packed = atrium_packed
    dev = pta.PtkRenderingDevice(0)
    try:
        disp = wd.PTWorkDispatcher(dev)
        disp.create_buffers(w, h, packed=packed)
        disp.load_shader()
        win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=8, max_refraction_bounces=4)
        win.frame = 0
        t = wd.sample_time(3)
        images = {}
        for name, r in _schedules(pta):
            dev.set_renderer(r)
            dev.texture_update(disp.image_buffer, np.zeros((h, w, 4), np.float32))
            disp.create_compute_list(win, scene.camera, t, sync=True)
            images[name] = dev.texture_get_data(disp.image_buffer)
        for name in ("wavefront", "persistent"):
            assert np.array_equal(images[name].view(np.uint32), images["one-kernel"].view(np.uint32)), \
                f"{name} differs from the one-kernel renderer at {w}x{h}"
        # windows: top-left corner, a region in the colonnade, the centre, bottom-right corner
        gx, gy = w // 8, h // 8
        spots = [(0, 0), (gx // 3, gy // 2), (gx // 2 - 8, gy // 2 - 8), (gx - 16, gy - 16)]
        dev.set_renderer(pta.rendering_device.PTK_RENDERER_AUTO)
        dev.trace_enable(3)
        for (ox, oy) in spots:
            tile = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=8, max_refraction_bounces=4,
                                     x_offset=ox * 8, y_offset=oy * 8, work_group_width=16, work_group_height=16)
            tile.frame = 0
            cpu = np.zeros((h, w, 4), np.float32)
            out = ob.render(packed, wd.push_constant_bytes(scene.camera, tile, t), cpu, 16, 16)
            ys, xs = slice(oy * 8, oy * 8 + 128), slice(ox * 8, ox * 8 + 128)
            _compare_image(f"c5 {w}x{h} window {ox},{oy}", np.ascontiguousarray(images["one-kernel"][ys, xs]),
                           np.ascontiguousarray(cpu[ys, xs]))
            dev.texture_update(disp.image_buffer, np.zeros((h, w, 4), np.float32))
            disp.create_compute_list(tile, scene.camera, t, sync=True)
            traced = dev.texture_get_data(disp.image_buffer)
            _compare(f"c5 {w}x{h} traced window {ox},{oy}", np.ascontiguousarray(traced[ys, xs]), np.ascontiguousarray(cpu[ys, xs]),
                     dev.trace_read_primary(disp.image_buffer)[ys, xs], out["primary_id"][ys, xs],
                     dev.trace_read_counters(disp.image_buffer)[ys, xs], out["pixel_stats"][ys, xs])
        dev.trace_enable(0)
        disp.free_rids()
    finally:
        dev.close()


def deep_chain_scene(pta, depth=6, seed=3):
    """A hand-written tree in which every inner node holds one inner child (slot 0) and seven one-sphere
    leaves, all with the same box: every child is accepted with equal entry distance, ties go to the
    lowest slot, so a ray walks down the chain with seven siblings left on the stack per level.  The
    deepest stack is 7 * (depth - 1) + 8 entries, past the 32-entry LDS window for depth >= 5."""
    wd, sc, gm = pta.work_dispatcher, pta.scene, pta.godot_math
    rng = np.random.default_rng(seed)
    n_sph = 7 * (depth - 1) + 8
    scene = sc.PTScene()
    glass = sc.PTMaterial(opacity=0.0, IOR=1.5, dielectric_priority=1)
    mats = [sc.PTMaterial(albedo=(0.8, 0.3, 0.2)), sc.PTMaterial(albedo=(0.3, 0.7, 0.9), metallic=0.8), glass]
    for i in range(n_sph):
        scene.add_object(sc.PTSphere(transform=gm.Transform3D(origin=tuple(rng.uniform(-3.0, 3.0, 3).tolist())),
                                     radius=float(rng.uniform(0.3, 0.9)), material=mats[i % 3]))
    scene.add_object(sc.PTPlane(normal=(0.0, 1.0, 0.0), distance=-4.0, material=mats[0]))
    scene.camera = sc.PTCamera(transform=gm.Transform3D.looking_at((0.0, 1.0, 11.0), (0.0, 0.0, 0.0)), fov=45.0)
    packed = copy.copy(wd.pack_scene(scene, ob.OracleBvhBuilder(), 8))

    def node(node_id):
        return struct.pack("<3fi3fi", -5.0, -5.0, -5.0, node_id, 5.0, 5.0, 5.0, -1)

    def leaf_id(slot):
        v = ((-slot - 1) ^ (0 << 24)) & 0xFFFFFFFF
        return v - (1 << 32) if v >= (1 << 31) else v

    recs, slot = [], 0
    recs.append(node(1 ^ (8 << 24)))  # root: children at 1..8
    for level in range(depth):
        first = 1 + 8 * level
        last_level = level == depth - 1
        for c in range(8):
            if c == 0 and not last_level:
                recs.append(node((first + 8) ^ (8 << 24)))
            else:
                recs.append(node(leaf_id(slot)))
                slot += 1
    assert slot == n_sph
    packed.bvh = wd.create_bvh_byte_array(b"".join(recs))
    packed.object_ids = np.array([i | (1 << 24) for i in range(n_sph)], np.int32).tobytes()
    return scene, packed


@pytest.mark.parametrize("window", [None, "64", "8"])
def test_stack_deeper_than_the_lds_window(pta, monkeypatch, window):
    """Stack need 43 > the LDS window (20 by default; the largest a compute unit holds when 64 is asked for; 8):
    entries past the window go to the global spill column in all three schedules and in the traced tile kernel;
    everything equals the oracle."""
    if window is not None:
        monkeypatch.setenv("PTK_LDS_WINDOW", window)
    wd = pta.work_dispatcher
    scene, packed = deep_chain_scene(pta, depth=6)
    w, h = 160, 96
    dev = pta.PtkRenderingDevice(0)
    try:
        disp = wd.PTWorkDispatcher(dev)
        disp.create_buffers(w, h, packed=packed)
        disp.load_shader()
        cpu = np.zeros((h, w, 4), np.float32)
        for k in range(2):
            win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=6, max_refraction_bounces=4)
            win.frame = k
            t = wd.sample_time(k)
            before = cpu.copy()
            out = ob.render(packed, wd.push_constant_bytes(scene.camera, win, t), cpu, w // 8, h // 8)
            assert out["pixel_stats"][..., 1].max() >= 8 * 6, "the rays do walk the whole chain"
            for name, r in _schedules(pta):
                dev.set_renderer(r)
                dev.texture_update(disp.image_buffer, before)
                disp.create_compute_list(win, scene.camera, t, sync=True)
                _compare_image(f"deep[{k}] {name}", dev.texture_get_data(disp.image_buffer), cpu)
            dev.set_renderer(pta.rendering_device.PTK_RENDERER_AUTO)
            dev.trace_enable(3)
            dev.texture_update(disp.image_buffer, before)
            disp.create_compute_list(win, scene.camera, t, sync=True)
            _compare(f"deep[{k}] traced", dev.texture_get_data(disp.image_buffer), cpu, dev.trace_read_primary(disp.image_buffer),
                     out["primary_id"], dev.trace_read_counters(disp.image_buffer), out["pixel_stats"])
            dev.trace_enable(0)
        disp.free_rids()
    finally:
        dev.close()


def test_progressive_257_dispatches(pta, golden_dir):
    """BASELINE config 3's length: 257 dispatches (frames 0..256 = 256 accumulated samples,
    ray_tracer.comp:1683-1693) of fake_cornell_box with USE_EMISSION 1 at 160x96; the accumulated image
    equals the oracle's after 2, 17, 65 and 257 dispatches, for each schedule."""
    wd = pta.work_dispatcher
    scene = pta.scenes.load_fixture_scene("fake_cornell_box", golden_dir)
    packed = wd.pack_scene(scene, ob.OracleBvhBuilder(), 8)
    w, h = 160, 96
    devs = []
    try:
        for name, r in _schedules(pta):
            dev = pta.PtkRenderingDevice(0)
            dev.set_renderer(r)
            d = wd.PTWorkDispatcher(dev)
            d.create_buffers(w, h, packed=packed)
            d.load_shader(use_emission=True)
            devs.append((name, dev, d))
        cpu = np.zeros((h, w, 4), np.float32)
        win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=8, max_refraction_bounces=8, max_samples=256)
        for k in range(257):
            win.frame = k
            t = wd.sample_time(k)
            ob.render(packed, wd.push_constant_bytes(scene.camera, win, t), cpu, w // 8, h // 8, use_emission=True, trace=False)
            for _, dev, d in devs:
                d.create_compute_list(win, scene.camera, t)
            if k in (1, 16, 64, 256):
                for name, dev, d in devs:
                    dev.sync()
                    _compare_image(f"progressive[{k}] {name}", dev.texture_get_data(d.image_buffer), cpu)
        assert float(cpu[..., :3].max()) > 0.0, "the emissive ceiling light reaches the image"
    finally:
        for _, dev, d in devs:
            d.free_rids()
            dev.close()
