"""The C ABI used from plain C (tests/c_abi/c_abi_harness.c, gcc -std=c99) and the display conversion.

The harness renders sphere_scene1 from the committed byte fixture; its image must equal the CPU oracle's
render of the same bytes bit for bit, and its RGBA8 display output the numpy restatement of
shaders/canvas.gdshader:33 (pow 2.2) within one code value (the library's pow is its own polynomial)."""
import os
import subprocess

import numpy as np
import pytest

import oracle_binding as ob
from make_c_harness_fixture import read_fixture

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
HARNESS = os.path.join(HERE, "c_abi", "c_abi_harness")
FIXTURE = os.path.join(HERE, "golden", "c_harness_sphere_scene1.bin")


def test_rd_core_harness_matches_oracle(pta, tmp_path):
    """The Godot-free core of the GDExtension class (gdextension/src/ptk_rd_core.h) driven like a RenderingDevice from
    C++: shader from source text, unplaced resources, uniform sets with the 512 filler pairs, a compute list
    dispatched twice.  Its image equals the oracle's."""
    from godot_style_dispatcher import SHADER_TEMPLATE, load_shader_source
    harness = os.path.join(HERE, "c_abi", "rd_core_harness")
    assert os.path.exists(harness), "tests/c_abi/rd_core_harness is built by __graft_entry__.build()"
    hdr, push, paths, bufs = read_fixture(FIXTURE)
    assert not paths
    src = load_shader_source(SHADER_TEMPLATE.format(refraction=hdr["refraction_mode"], emission=hdr["use_emission"],
                                                    early_t=hdr["early_t_break"]), hdr["order"], [])
    src_path, img_path = tmp_path / "shader.glsl", tmp_path / "image.f32"
    src_path.write_text(src.source_compute)
    run = subprocess.run([harness, FIXTURE, str(src_path), str(img_path)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    w, h = hdr["width"], hdr["height"]
    gpu = np.fromfile(img_path, np.float32).reshape(h, w, 4)
    wd = pta.work_dispatcher
    packed = wd.PackedScene(materials=bufs[(2, 0)], spheres=bufs[(2, 1)], planes=bufs[(2, 2)], triangles=bufs[(2, 3)],
                            bvh=bufs[(3, 0)], object_ids=bufs[(3, 1)], transforms=bufs[(6, 0)], proc_tex_paths=paths,
                            sampled_textures=[], bvh_order=hdr["order"])
    cpu = np.zeros((h, w, 4), np.float32)
    ob.render(packed, push, cpu, hdr["groups_x"], hdr["groups_y"], trace=False)
    assert np.array_equal(gpu.view(np.uint32), cpu.view(np.uint32)), "the RD core harness's image differs from the oracle"


def display_restatement(img, transfer):
    """numpy form of ptk_image_read_display: clamp, canvas.gdshader's pow(rgb, 2.2), sRGB encode, round to 8 bits."""
    v = np.nan_to_num(img.astype(np.float64), nan=0.0).clip(0.0, 1.0)
    rgb = v[..., :3]
    if transfer >= 1:
        rgb = rgb ** 2.2
    if transfer >= 2:
        rgb = np.where(rgb <= 0.0031308, rgb * 12.92, 1.055 * rgb ** (1 / 2.4) - 0.055)
    return np.concatenate([rgb, v[..., 3:]], axis=-1) * 255.0  # unrounded code values


def check_display(out, img, transfer):
    """Codes equal round-half-up of the restatement, except where the restatement lies within 0.01 of a rounding
    boundary (the library's pow is its own float32 polynomial): there one code value either way is accepted."""
    x = display_restatement(img, transfer)
    ref = np.floor(x + 0.5).astype(np.int32)
    d = np.abs(out.astype(np.int32) - ref)
    near_boundary = np.abs((x + 0.5) - np.round(x + 0.5)) < 0.01
    assert d.max() <= 1, int(d.max())
    assert not np.any((d != 0) & ~near_boundary), int(np.count_nonzero((d != 0) & ~near_boundary))
    return d


def test_plain_c_harness_matches_oracle(pta, tmp_path):
    assert os.path.exists(HARNESS), "tests/c_abi/c_abi_harness is built by __graft_entry__.build()"
    img_path, disp_path = tmp_path / "image.f32", tmp_path / "display.rgba8"
    run = subprocess.run([HARNESS, FIXTURE, str(img_path), str(disp_path)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    hdr, push, paths, bufs = read_fixture(FIXTURE)
    w, h = hdr["width"], hdr["height"]
    gpu = np.fromfile(img_path, np.float32).reshape(h, w, 4)
    wd = pta.work_dispatcher
    packed = wd.PackedScene(materials=bufs[(2, 0)], spheres=bufs[(2, 1)], planes=bufs[(2, 2)], triangles=bufs[(2, 3)],
                            bvh=bufs[(3, 0)], object_ids=bufs[(3, 1)], transforms=bufs[(6, 0)], proc_tex_paths=paths,
                            sampled_textures=[], bvh_order=hdr["order"])
    cpu = np.zeros((h, w, 4), np.float32)
    ob.render(packed, push, cpu, hdr["groups_x"], hdr["groups_y"], trace=False)
    assert np.array_equal(gpu.view(np.uint32), cpu.view(np.uint32)), "the C harness's image differs from the oracle"
    check_display(np.fromfile(disp_path, np.uint8).reshape(h, w, 4), cpu, 1)


@pytest.mark.parametrize("transfer", [0, 1, 2])
def test_display_conversion(pta, transfer):
    """Random image with values below 0, above 1, NaN and an odd pixel count (tail of the 4-pixel lanes)."""
    rng = np.random.default_rng(9)
    w, h = 333, 7
    img = rng.uniform(-0.2, 1.3, (h, w, 4)).astype(np.float32)
    img[0, :16] = np.linspace(0.0, 1.0, 16, dtype=np.float32)[:, None]
    img[1, 0, 0] = np.nan
    img[1, 1] = [0.0, 1.0, 0.5, 1.0]
    dev = pta.PtkRenderingDevice(0)
    try:
        image = dev.image_create(w, h)
        dev.texture_update(image, img)
        d = check_display(dev.texture_get_display(image, transfer), img, transfer)
        if transfer == 0:
            assert d.max() == 0  # pure clamp and quantise: exact
        with pytest.raises(pta.PtkError):
            dev.texture_get_display(image, 7)
        dev.free_rid(image)
    finally:
        dev.close()


def test_rccl_entry_points_with_one_rank(pta, golden_dir):
    """ptk_comm_* / ptk_image_reduce_mean / ptk_image_gather_bands with a communicator of one rank (all a one-GPU box
    can hold: RCCL refuses two ranks on one device): librccl is found and bound at run time, the mean over one
    rank and the gather of one band leave the image bit for bit as it was, and misuse fails with a message."""
    import oracle_binding as ob
    wd = pta.work_dispatcher
    scene = pta.scenes.load_fixture_scene("sphere_scene1", golden_dir)
    packed = wd.pack_scene(scene, ob.OracleBvhBuilder(), 8)
    w, h = 128, 72
    dev = pta.PtkRenderingDevice(0)
    try:
        disp = wd.PTWorkDispatcher(dev)
        disp.create_buffers(w, h, packed=packed)
        disp.load_shader()
        win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=4)
        win.frame = 1
        disp.create_compute_list(win, scene.camera, wd.sample_time(0), sync=True)
        before = dev.texture_get_data(disp.image_buffer).copy()
        with pytest.raises(pta.rendering_device.PtkError, match="ptk_comm_init has not been called"):
            dev.image_reduce_mean(disp.image_buffer, 0)
        uid = pta.PtkRenderingDevice.comm_unique_id()
        assert len(uid) == 128 and any(uid)
        with pytest.raises(pta.rendering_device.PtkError):
            dev.comm_init(1, 3, uid)  # rank outside the world
        dev.comm_init(1, 0, uid)
        with pytest.raises(pta.rendering_device.PtkError, match="already belongs"):
            dev.comm_init(1, 0, uid)
        dev.image_reduce_mean(disp.image_buffer, 0)
        assert np.array_equal(dev.texture_get_data(disp.image_buffer).view(np.uint32), before.view(np.uint32))
        dev.image_gather_bands(disp.image_buffer, [0], [h], 0)
        assert np.array_equal(dev.texture_get_data(disp.image_buffer).view(np.uint32), before.view(np.uint32))
        with pytest.raises(pta.rendering_device.PtkError, match="leaves the image"):
            dev.image_gather_bands(disp.image_buffer, [8], [h], 0)
        with pytest.raises(pta.rendering_device.PtkError, match="outside the communicator"):
            dev.image_reduce_mean(disp.image_buffer, 1)
        dev.comm_destroy()
        dev.comm_destroy()  # idempotent
        disp.free_rids()
    finally:
        dev.close()
