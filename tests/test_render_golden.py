"""Render fixtures: CRC-32 of the images the CPU oracle produced for the fixture scenes when
tests/golden/render_crc.json was written (tests/make_render_golden.py).  The reference holds no render
vectors, so these do not pin the oracle to the reference; they pin it -- and the HIP renderers -- to a
recorded state, across compilers and machines (all arithmetic is IEEE binary32 with fixed operation order)."""
import importlib
import json
import os
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

with open(os.path.join(ROOT, "tests", "golden", "render_crc.json")) as f:
    GOLDEN = json.load(f)
IDS = [f"{c['scene']}-mode{c['refraction_mode']}" for c in GOLDEN["cases"]]


@pytest.mark.parametrize("case", GOLDEN["cases"], ids=IDS)
def test_oracle_reproduces_recorded_images(case, golden_dir):
    mk = importlib.import_module("make_render_golden")
    _, packed, img = mk.render_case(case["scene"], case["depth"], case["refraction_bounces"], case["use_emission"],
                                    case["refraction_mode"], golden_dir)
    assert "%08x" % zlib.crc32(packed.bvh) == case["bvh_crc32"]
    assert "%08x" % zlib.crc32(img.tobytes()) == case["image_crc32"]


@pytest.mark.gpu
@pytest.mark.parametrize("renderer", [1, 2, 3], ids=["one-kernel", "wavefront", "persistent"])
@pytest.mark.parametrize("case", GOLDEN["cases"], ids=IDS)
def test_gpu_reproduces_recorded_images(pta, case, renderer, golden_dir):
    """Same fixtures through the C ABI, BVH built on the GPU, no oracle involved at run time."""
    wd = pta.work_dispatcher
    w, h = GOLDEN["width"], GOLDEN["height"]
    scene = pta.scenes.load_fixture_scene(case["scene"], golden_dir)
    rd = pta.PtkRenderingDevice(0)
    try:
        rd.set_renderer(renderer)
        disp = wd.PTWorkDispatcher(rd)
        disp.set_scene(scene)
        disp.create_buffers(w, h, order=8)
        assert "%08x" % zlib.crc32(disp.packed.bvh) == case["bvh_crc32"]
        disp.load_shader(refraction_mode=case["refraction_mode"], use_emission=case["use_emission"])
        for k in range(GOLDEN["samples"]):
            win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=case["depth"],
                                    max_refraction_bounces=case["refraction_bounces"])
            win.frame = k
            disp.create_compute_list(win, scene.camera, wd.sample_time(k), sync=True)
        img = rd.texture_get_data(disp.image_buffer)
        assert "%08x" % zlib.crc32(np.ascontiguousarray(img).tobytes()) == case["image_crc32"]
        disp.free_rids()
    finally:
        rd.close()
