"""The RenderingDevice-shaped facade (path-tracer_amd/godot_rd.py) under a dispatcher that speaks only Godot's API
(tests/godot_style_dispatcher.py: PTWorkDispatcher's calls, one for one): images bit-identical to the CPU oracle,
so pointing the plugin's `rd` at the facade is the whole integration."""
import importlib

import numpy as np
import pytest

import oracle_binding as ob
from godot_style_dispatcher import (CHECKER_XYZ, CHECKER_XZ, SHADER_TEMPLATE, STAND_IN_FILES, GodotStyleDispatcher,
                                    load_shader_source)
from test_gpu_edge import _textured_scene
from test_gpu_parity import _compare_image

pytestmark = pytest.mark.gpu
grd = importlib.import_module("path-tracer_amd.godot_rd")


@pytest.fixture(scope="module")
def grd_dev():
    dev = grd.RenderingDevice(0)
    for name, text in STAND_IN_FILES.items():  # the scene files' stand-ins compute the shipped functions
        dev.ptk.register_procedural_texture(text, name)
    yield dev
    dev.close()


def _source_for(packed, emission=False):
    texts = [STAND_IN_FILES[p.rsplit("/", 1)[-1]] for p in packed.proc_tex_paths]
    template = SHADER_TEMPLATE.format(refraction=2, emission=int(emission), early_t=0)
    return load_shader_source(template, packed.bvh_order, texts)


def _render_and_compare(pta, dev, name, scene, w, h, depth, bounces, emission=False, dispatches=2, bvh_type=None):
    wd = pta.work_dispatcher
    packed = wd.pack_scene(scene, ob.OracleBvhBuilder(), 8, bvh_type)
    disp = GodotStyleDispatcher(dev)
    disp.load_shader(_source_for(packed, emission))
    disp.create_buffers(w, h, packed)
    win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=depth, max_refraction_bounces=bounces)
    cpu = np.zeros((h, w, 4), np.float32)
    try:
        for k in range(dispatches):
            win.frame = k
            push = wd.push_constant_bytes(scene.camera, win, wd.sample_time(k))
            ob.render(packed, push, cpu, w // 8, h // 8, use_emission=emission)
            disp.create_compute_list(win, push)
            raw = dev.texture_get_data(disp.image_buffer, 0)  # pt_renderer.gd:327
            assert isinstance(raw, bytes) and len(raw) == w * h * 16
            _compare_image(f"{name}[{k}] via RenderingDevice facade", np.frombuffer(raw, np.float32).reshape(h, w, 4), cpu)
    finally:
        disp.free_rids()
    return packed


def test_fixture_scenes_through_the_facade(pta, grd_dev, golden_dir):
    for name, w, h, depth, bounces, emission in [("sphere_scene1", 640, 360, 4, 4, False),
                                                 ("benchmark_random_scene", 320, 184, 8, 4, False),
                                                 ("fake_cornell_box", 320, 184, 8, 8, True),
                                                 ("dragon_test", 320, 184, 4, 4, False)]:
        scene = pta.scenes.load_fixture_scene(name, golden_dir)
        _render_and_compare(pta, grd_dev, name, scene, w, h, depth, bounces, emission=emission)


def test_sampled_filler_and_procedural_textures_through_the_facade(pta, grd_dev):
    """Sampled RGBA32F textures, the 8x8 R8G8B8A8 filler made from the plugin's bytes (and placed in every unused slot
    of the 512-entry sampler array), both procedural functions."""
    _render_and_compare(pta, grd_dev, "textured", _textured_scene(pta), 320, 184, 6, 4)


def test_edit_and_growth_through_the_facade(pta, grd_dev, golden_dir):
    """PTRenderer's direct calls on wd.rd: buffer_update of one record (pt_renderer.gd:664-697), then a grown sphere
    buffer -- free_rid, storage_buffer_create, a new uniform set (pt_work_dispatcher.gd:511-567)."""
    wd = pta.work_dispatcher
    scene = pta.scenes.load_fixture_scene("sphere_scene1", golden_dir)
    packed = wd.pack_scene(scene, ob.OracleBvhBuilder(), 8)
    w, h = 256, 144
    disp = GodotStyleDispatcher(grd_dev)
    disp.load_shader(_source_for(packed))
    disp.create_buffers(w, h, packed)
    win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=4, max_refraction_bounces=4)
    try:
        push = wd.push_constant_bytes(scene.camera, win, wd.sample_time(0))
        disp.create_compute_list(win, push)
        # a material edit: the first user material turns into a mirror
        m = bytearray(packed.materials[2 * 64:3 * 64])
        m[16:20] = np.float32(1.0).tobytes()  # metallic
        grd_dev.buffer_update(disp.material_buffer, 2 * 64, 64, bytes(m))
        packed.materials = packed.materials[:2 * 64] + bytes(m) + packed.materials[3 * 64:]
        # growth by one step of 64 records
        packed.spheres = packed.spheres + bytes(64 * 32)
        disp.replace_buffer("sphere_buffer", packed.spheres, wd.OBJECT_SET_INDEX, wd.SPHERES_BIND)
        cpu = np.zeros((h, w, 4), np.float32)
        ob.render(packed, push, cpu, w // 8, h // 8)
        disp.create_compute_list(win, push)
        gpu = np.frombuffer(grd_dev.texture_get_data(disp.image_buffer, 0), np.float32).reshape(h, w, 4)
        _compare_image("edited + grown", gpu, cpu)
        assert grd_dev.buffer_get_data(disp.material_buffer, 2 * 64, 64) == bytes(m)
    finally:
        disp.free_rids()


def test_unknown_procedural_function_is_a_compile_error(pta, grd_dev):
    """A spliced function the library does not know: Godot would compile it; here shader_create_from_spirv returns
    an invalid RID with the reason in the SPIR-V object's error text, as a failed compile does."""
    other = "vec4 procedural_texture(vec3 pos) { return vec4(fract(pos), 1); }\n"
    src = load_shader_source(SHADER_TEMPLATE.format(refraction=2, emission=0, early_t=0), 8, [CHECKER_XZ, other, CHECKER_XYZ])
    spirv = grd_dev.shader_compile_spirv_from_source(src)
    info = grd_dev.ptk.shader_source_info(src.source_compute)
    assert info["order"] == 8 and info["builtins"] == [1, 0, 2]
    shader = grd_dev.shader_create_from_spirv(spirv)
    assert not shader.is_valid() and "procedural texture function 2" in spirv.compile_error_compute
    assert not grd_dev.compute_pipeline_create(shader).is_valid()
    no_order = grd.RDShaderSource()
    no_order.source_compute = "void main() {}"
    assert "order" in grd_dev.shader_compile_spirv_from_source(no_order).compile_error_compute
