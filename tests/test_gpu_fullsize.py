"""The lighter BASELINE configurations at their own size and length: 1920x1080, the configuration's number of
progressive dispatches, all three schedules bit-equal over the whole frame, and four 128x128 windows of the
accumulated image equal to the CPU oracle's accumulation of the same windows (image_offset push constants).

  C2  benchmark_random_scene, 17 dispatches, depth 8
  C3  fake_cornell_box with USE_EMISSION 1 and sphere_scene3 (nested dielectrics), 257 dispatches, depth 8
  C4  dragon_test, 65 dispatches, depth 8; its trees come from the HIP builder and are bound with ptk_bvh_bind

Same bar as test_gpu_parity: images bit-identical (stated tolerance 1e-4)."""
import numpy as np
import pytest

import oracle_binding as ob
from test_gpu_parity import _compare_image

pytestmark = pytest.mark.gpu

W, H = 1920, 1080
CASES = [  # name, dispatches, depth, refraction bounces, USE_EMISSION, tree from the HIP builder
    ("benchmark_random_scene", 17, 8, 4, False, False),
    ("fake_cornell_box", 257, 8, 8, True, False),
    ("sphere_scene3", 257, 8, 8, False, False),
    ("dragon_test", 65, 8, 4, False, True),
]


@pytest.mark.parametrize("name,dispatches,depth,bounces,emission,device_tree", CASES)
def test_light_configurations_full_size(pta, golden_dir, name, dispatches, depth, bounces, emission, device_tree):
    wd, R = pta.work_dispatcher, pta.rendering_device
    scene = pta.scenes.load_fixture_scene(name, golden_dir)
    dev = pta.PtkRenderingDevice(0)
    try:
        disp = wd.PTWorkDispatcher(dev)
        if device_tree:
            disp.set_scene(scene)
            disp.create_buffers(W, H, order=8)  # ptk_bvh_build + ptk_bvh_bind; disp.packed mirrors the bound bytes
        else:
            disp.create_buffers(W, H, packed=wd.pack_scene(scene, ob.OracleBvhBuilder(), 8))
        packed = disp.packed
        disp.load_shader(use_emission=emission)
        win = wd.PTRenderWindow(render_width=W, render_height=H, max_ray_depth=depth, max_refraction_bounces=bounces,
                                max_samples=dispatches)
        images = {}
        for label, r in (("one-kernel", R.PTK_RENDERER_SINGLE_KERNEL), ("wavefront", R.PTK_RENDERER_WAVEFRONT),
                         ("persistent", R.PTK_RENDERER_PERSISTENT)):
            dev.set_renderer(r)
            dev.texture_update(disp.image_buffer, np.zeros((H, W, 4), np.float32))
            for k in range(dispatches):
                win.frame = k
                disp.create_compute_list(win, scene.camera, wd.sample_time(k))
            dev.sync()
            assert dev.last_dispatch_renderer() == r
            images[label] = dev.texture_get_data(disp.image_buffer)
        for label in ("wavefront", "persistent"):
            assert np.array_equal(images[label].view(np.uint32), images["one-kernel"].view(np.uint32)), \
                f"{name}: {label} differs from the one-kernel renderer after {dispatches} dispatches at {W}x{H}"
        gx, gy = W // 8, H // 8
        for (ox, oy) in [(0, 0), (gx // 3, gy // 2), (gx // 2 - 8, gy // 2 - 8), (gx - 16, gy - 16)]:
            tile = wd.PTRenderWindow(render_width=W, render_height=H, max_ray_depth=depth, max_refraction_bounces=bounces,
                                     max_samples=dispatches, x_offset=ox * 8, y_offset=oy * 8, work_group_width=16, work_group_height=16)
            cpu = np.zeros((H, W, 4), np.float32)
            for k in range(dispatches):
                tile.frame = k
                ob.render(packed, wd.push_constant_bytes(scene.camera, tile, wd.sample_time(k)), cpu, 16, 16,
                          use_emission=emission, trace=False)
            ys, xs = slice(oy * 8, oy * 8 + 128), slice(ox * 8, ox * 8 + 128)
            _compare_image(f"{name} {W}x{H} x{dispatches} window {ox},{oy}", np.ascontiguousarray(images["one-kernel"][ys, xs]),
                           np.ascontiguousarray(cpu[ys, xs]))
        disp.free_rids()
    finally:
        dev.close()
