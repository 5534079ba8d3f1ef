"""GPU parity on the edges of the dispatch contract: textures of every kind, image sizes that are not a
multiple of the 8x8 group, an empty scene, and the cases where the wavefront pipeline hands over to the
one-kernel renderer.  Same bar as test_gpu_parity: images bit-identical to the CPU oracle."""
import numpy as np
import pytest

import oracle_binding as ob
from test_gpu_parity import _compare_image, _run_case, rd, rd_wf  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu


def _textured_scene(pta):
    sc, gm = pta.scene, pta.godot_math
    rng = np.random.default_rng(11)
    scene = sc.PTScene()
    img = rng.random((8, 16, 4), dtype=np.float32)
    tex_a = sc.PTSampledTexture(image=img)
    tex_missing = sc.PTSampledTexture(image=None)  # filler grid
    tex_uv = sc.PTProceduralTexture(texture_path="res://shaders/procedural_textures/checker_board.comp")
    tex_sp = sc.PTProceduralTexture(texture_path="res://shaders/procedural_textures/checker_board_spatial.comp", is_spatial=True)
    grey = sc.PTMaterial(albedo=(0.8, 0.8, 0.8))
    metal = sc.PTMaterial(albedo=(0.9, 0.7, 0.5), metallic=0.9)
    glass = sc.PTMaterial(opacity=0.0, IOR=1.5)
    for i, (tex, mat) in enumerate([(tex_a, grey), (tex_missing, grey), (tex_uv, metal), (tex_sp, grey), (None, glass)]):
        scene.add_object(sc.PTSphere(radius=0.6, material=mat, texture=tex, transform=gm.Transform3D(origin=(-3.0 + 1.5 * i, 0.0, 0.0))))
    scene.add_object(sc.PTPlane(normal=(0.0, 1.0, 0.0), distance=-0.6, material=grey, texture=tex_sp))
    scene.add_object(sc.PTTriangle(vertex1=(-4.0, -0.6, -2.0), vertex2=(4.0, -0.6, -2.0), vertex3=(0.0, 3.0, -2.5), material=grey, texture=tex_a))
    scene.camera = sc.PTCamera(transform=gm.Transform3D.looking_at((0.0, 1.0, 6.0), (0.0, 0.0, 0.0)), fov=50.0)
    return scene


def test_sampled_filler_and_procedural_textures(pta, rd, rd_wf, golden_dir):
    _run_case(pta, rd, rd_wf, golden_dir, "textured", 320, 184, 6, 4, dispatches=2, scene=_textured_scene(pta))


def test_missing_texture_texels_are_the_reference_bytes(pta, rd, rd_wf, golden_dir):
    """ptk_texture_create(NULL) binds the filler grid: a white sphere sampling it shows exactly sky x (56,16,56,56)/255
    and sky x (0,0,0,56)/255, the bytes of the reference's own test (test_utils.gd:5-11), on every schedule."""
    import json
    import os
    from test_oracle_golden import check_missing_texture_image, missing_texture_scene
    with open(os.path.join(golden_dir, "reference_vectors.json")) as f:
        vectors = json.load(f)
    wd = pta.work_dispatcher
    scene = missing_texture_scene(pta)
    packed = wd.pack_scene(scene, ob.OracleBvhBuilder(), 8)
    w, h = 96, 64
    counts = []
    for dev in (rd, rd_wf):
        d = wd.PTWorkDispatcher(dev)
        d.create_buffers(w, h, packed=packed)
        d.load_shader()
        try:
            win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=2, max_refraction_bounces=0)
            win.frame = 0
            d.create_compute_list(win, scene.camera, wd.sample_time(0), sync=True)
            counts.append(check_missing_texture_image(dev.texture_get_data(d.image_buffer), vectors))
        finally:
            d.free_rids()
    assert counts[0] == counts[1]


def test_image_not_a_multiple_of_the_group(pta, rd, rd_wf, golden_dir):
    """100 x 60 pixels under 13 x 8 groups: invocations outside the image compute but never store."""
    wd = pta.work_dispatcher
    scene = pta.scenes.load_fixture_scene("sphere_scene1", golden_dir)
    packed = wd.pack_scene(scene, ob.OracleBvhBuilder(), 8)
    w, h, gx, gy = 100, 60, 13, 8
    cpu = np.zeros((h, w, 4), np.float32)
    disps = []
    for dev in (rd, rd_wf):
        d = wd.PTWorkDispatcher(dev)
        d.create_buffers(w, h, packed=packed)
        d.load_shader()
        disps.append((dev, d))
    try:
        for k in range(3):
            win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=4, work_group_width=gx, work_group_height=gy)
            win.frame = k
            t = wd.sample_time(k)
            ob.render(packed, wd.push_constant_bytes(scene.camera, win, t), cpu, gx, gy)
            for dev, d in disps:
                d.create_compute_list(win, scene.camera, t, sync=True)
                _compare_image(f"ragged[{k}]", dev.texture_get_data(d.image_buffer), cpu)
    finally:
        for _, d in disps:
            d.free_rids()


def test_empty_scene_is_all_sky(pta, rd, rd_wf, golden_dir):
    scene = pta.scene.PTScene()
    scene.camera = pta.scene.PTCamera()
    _run_case(pta, rd, rd_wf, golden_dir, "empty", 64, 64, 3, 2, dispatches=2, scene=scene)


def test_wavefront_hands_over_outside_its_range(pta, rd, golden_dir):
    """More than 64 bounces, or depth 0: AUTO does not use the pipeline (same image as the oracle), and a
    context pinned to the wavefront pipeline reports the dispatch as unsupported.  The persistent kernel has
    no bound on the bounce count; depth 0 it refuses too."""
    wd = pta.work_dispatcher
    scene = pta.scenes.load_fixture_scene("sphere_scene3", golden_dir)
    packed = wd.pack_scene(scene, ob.OracleBvhBuilder(), 8)
    w, h = 64, 64
    dev = pta.PtkRenderingDevice(0)
    try:
        d = wd.PTWorkDispatcher(dev)
        d.create_buffers(w, h, packed=packed)
        d.load_shader()
        for depth, bounces in [(70, 4), (0, 4), (8, 4)]:
            dev.texture_update(d.image_buffer, np.zeros((h, w, 4), np.float32))
            cpu = np.zeros((h, w, 4), np.float32)
            win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=depth, max_refraction_bounces=bounces)
            push = wd.push_constant_bytes(scene.camera, win, wd.sample_time(1))
            ob.render(packed, push, cpu, w // 8, h // 8)
            dev.set_renderer(pta.rendering_device.PTK_RENDERER_AUTO)
            d.create_compute_list(win, scene.camera, wd.sample_time(1), sync=True)
            if depth != 8:
                assert dev.dispatch_traverse_ms(0) == [], "outside its range the pipeline must not run"
            _compare_image(f"depth{depth}", dev.texture_get_data(d.image_buffer), cpu)
            if depth != 8:
                dev.set_renderer(pta.rendering_device.PTK_RENDERER_WAVEFRONT)
                with pytest.raises(pta.PtkError):
                    d.create_compute_list(win, scene.camera, wd.sample_time(1), sync=True)
                dev.set_renderer(pta.rendering_device.PTK_RENDERER_PERSISTENT)
                if depth == 0:
                    with pytest.raises(pta.PtkError):
                        d.create_compute_list(win, scene.camera, wd.sample_time(1), sync=True)
                else:
                    dev.texture_update(d.image_buffer, np.zeros((h, w, 4), np.float32))
                    d.create_compute_list(win, scene.camera, wd.sample_time(1), sync=True)
                    _compare_image(f"depth{depth} persistent", dev.texture_get_data(d.image_buffer), cpu)
        # 40 bounces are inside the pipeline's range: pinned to it, the image is the oracle's
        dev.set_renderer(pta.rendering_device.PTK_RENDERER_WAVEFRONT)
        dev.texture_update(d.image_buffer, np.zeros((h, w, 4), np.float32))
        cpu = np.zeros((h, w, 4), np.float32)
        win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=40, max_refraction_bounces=4)
        ob.render(packed, wd.push_constant_bytes(scene.camera, win, wd.sample_time(1)), cpu, w // 8, h // 8)
        d.create_compute_list(win, scene.camera, wd.sample_time(1), sync=True)
        assert len(dev.dispatch_traverse_ms(0)) > 40
        assert all(t == -1.0 for t in dev.dispatch_traverse_ms(0)), "launch timing is off unless asked for"
        _compare_image("depth40 pipeline", dev.texture_get_data(d.image_buffer), cpu)
        # the same dispatch with events around its traversal launches: same image, durations reported
        dev.set_option(pta.rendering_device.PTK_OPT_LAUNCH_TIMING, 1)
        dev.texture_update(d.image_buffer, np.zeros((h, w, 4), np.float32))
        d.create_compute_list(win, scene.camera, wd.sample_time(1), sync=True)
        timed = dev.dispatch_traverse_ms(0)
        assert len(timed) > 40 and all(t > 0.0 for t in timed)
        _compare_image("depth40 pipeline, timed", dev.texture_get_data(d.image_buffer), cpu)
        d.free_rids()
    finally:
        dev.close()


def test_auto_renderer_settles_on_the_faster_schedule(pta, golden_dir):
    """AUTO times both schedules on the first dispatches and keeps one; every dispatch on the way, whichever
    schedule rendered it, gives the oracle's image.  A light scene ends on the one-kernel renderer."""
    wd = pta.work_dispatcher
    scene = pta.scenes.load_fixture_scene("fake_cornell_box", golden_dir)
    packed = wd.pack_scene(scene, ob.OracleBvhBuilder(), 8)
    w, h = 256, 144
    dev = pta.PtkRenderingDevice(0)
    try:
        d = wd.PTWorkDispatcher(dev)
        d.create_buffers(w, h, packed=packed)
        d.load_shader()
        cpu = np.zeros((h, w, 4), np.float32)
        used = []
        for k in range(10):
            win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=6, max_refraction_bounces=4)
            win.frame = k
            t = wd.sample_time(k)
            ob.render(packed, wd.push_constant_bytes(scene.camera, win, t), cpu, w // 8, h // 8)
            d.create_compute_list(win, scene.camera, t, sync=True)
            used.append(len(dev.dispatch_traverse_ms(0)) > 0)
            _compare_image(f"auto[{k}]", dev.texture_get_data(d.image_buffer), cpu)
        assert True in used[:6] and False in used[:6], "every schedule is tried first (two dispatches each)"
        assert len(set(used[6:])) == 1, "then one of them is kept"
        d.free_rids()
    finally:
        dev.close()


def test_hand_made_tree_with_wide_leaves(pta, rd, rd_wf, golden_dir):
    """Node bytes a caller wrote by hand rather than PTBVHTree: a root with two leaves of 100 and 28 objects
    (the format allows up to 128 per leaf, ray_tracer.comp:600-606; the plugin's builder stops at `order`)."""
    import copy
    import struct

    wd, sc, gm = pta.work_dispatcher, pta.scene, pta.godot_math
    rng = np.random.default_rng(5)
    scene = sc.PTScene()
    glass = sc.PTMaterial(opacity=0.0, IOR=1.5, dielectric_priority=1)
    red = sc.PTMaterial(albedo=(0.8, 0.2, 0.2))
    for i in range(128):
        scene.add_object(sc.PTSphere(transform=gm.Transform3D(origin=tuple(rng.uniform(-3.0, 3.0, 3).tolist())),
                                     radius=float(rng.uniform(0.1, 0.6)), material=glass if i % 3 == 0 else red))
    scene.camera = sc.PTCamera(transform=gm.Transform3D.looking_at((0.0, 1.0, 9.0), (0.0, 0.0, 0.0)), fov=50.0)
    packed = copy.copy(wd.pack_scene(scene, ob.OracleBvhBuilder(), 8))

    def node(lo, hi, node_id, xf=-1):
        return struct.pack("<3fi3fi", lo, lo, lo, node_id, hi, hi, hi, xf)

    def leaf_id(first_slot, count):
        return ((-first_slot - 1) ^ ((count - 1) << 24)) & 0xFFFFFFFF

    to_i32 = lambda v: v - (1 << 32) if v >= (1 << 31) else v  # noqa: E731
    nodes = node(-4.0, 4.0, 1 ^ (2 << 24)) + node(-4.0, 4.0, to_i32(leaf_id(0, 100))) + node(-4.0, 4.0, to_i32(leaf_id(100, 28)))
    packed.bvh = wd.create_bvh_byte_array(nodes)
    packed.object_ids = np.array([i | (1 << 24) for i in range(128)], np.int32).tobytes()
    w, h = 160, 96
    cpu = np.zeros((h, w, 4), np.float32)
    disps = []
    for dev in (rd, rd_wf):
        d = wd.PTWorkDispatcher(dev)
        d.create_buffers(w, h, packed=packed)
        d.load_shader()
        disps.append((dev, d))
    try:
        for k in range(2):
            win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=6, max_refraction_bounces=6)
            win.frame = k
            t = wd.sample_time(k)
            ob.render(packed, wd.push_constant_bytes(scene.camera, win, t), cpu, w // 8, h // 8)
            for dev, d in disps:
                d.create_compute_list(win, scene.camera, t, sync=True)
                _compare_image(f"wide-leaves[{k}]", dev.texture_get_data(d.image_buffer), cpu)
    finally:
        for _, d in disps:
            d.free_rids()
