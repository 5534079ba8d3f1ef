"""ptk_dispatch_noalloc (the entry point that allocates nothing: the path kernel's pixel counter and
stack-spill column live in a static device array of the library) against ptk_dispatch pinned to the same
persistent path kernel, through the C ABI.  Bar: images bit-identical.  Two calls in a row on the same
context, the second with a smaller input, so that whatever the first call left in the shared array is
seen to do no harm; a tree deep enough to use the spill part of the array; and the error paths."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
from test_gpu_c5 import deep_chain_scene

pytestmark = pytest.mark.gpu


def _hip_free_bytes():
    """Free device memory as the HIP runtime reports it (the test's own check that nothing was allocated)."""
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def _pair(pta, packed, w, h):
    """Two contexts with the same scene: `ref` renders with ptk_dispatch (persistent renderer pinned),
    `dut` with ptk_dispatch_noalloc after one preparing ptk_dispatch."""
    wd = pta.work_dispatcher
    out = []
    for _ in range(2):
        dev = pta.PtkRenderingDevice(0)
        dev.set_renderer(pta.rendering_device.PTK_RENDERER_PERSISTENT)
        d = wd.PTWorkDispatcher(dev)
        d.create_buffers(w, h, packed=packed)
        d.load_shader()
        out.append((dev, d))
    return out


def _push(wd, scene, w, h, k, gx=None, gy=None, depth=6):
    win = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=depth, max_refraction_bounces=4,
                            work_group_width=gx, work_group_height=gy)
    win.frame = k
    return win, wd.push_constant_bytes(scene.camera, win, wd.sample_time(k))


@pytest.mark.parametrize("case", ["random_spheres", "deep_chain"])
def test_noalloc_equals_dispatch_twice_second_smaller(pta, golden_dir, case):
    wd = pta.work_dispatcher
    if case == "deep_chain":
        scene, packed = deep_chain_scene(pta, depth=6)  # stack need 43: 23 entries per lane spill into the static array
    else:
        scene = pta.scenes.load_fixture_scene("benchmark_random_scene", golden_dir)
        packed = wd.pack_scene(scene, ob.OracleBvhBuilder(), 8)
    w, h = 320, 184
    (ref, dref), (dut, ddut) = _pair(pta, packed, w, h)
    try:
        # the preparing dispatch (validates the scene and allocates its derived tables), then clear both images
        win, push = _push(wd, scene, w, h, 0)
        for dev, d in ((ref, dref), (dut, ddut)):
            dev.compute_list_dispatch(push, win.work_group_width, win.work_group_height, 1)
            dev.sync()
            dev.texture_update(d.image_buffer, np.zeros((h, w, 4), np.float32))
        free_before = _hip_free_bytes()
        # call 1: the full frame; call 2, straight after it: a smaller dispatch (fewer groups) accumulating on top
        for k, (gx, gy) in enumerate([(w // 8, h // 8), (w // 16, h // 16 + 1)]):
            win, push = _push(wd, scene, w, h, k, gx, gy)
            ref.compute_list_dispatch(push, gx, gy, 1)
            dut.compute_list_dispatch_noalloc(push, gx, gy, 1)
            if k == 0:
                continue  # no sync between the two calls: the second follows the first on the stream
        ref.sync()
        dut.sync()
        assert _hip_free_bytes() == free_before, "device memory was allocated or freed"
        a, b = ref.texture_get_data(dref.image_buffer), dut.texture_get_data(ddut.image_buffer)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{case}: {np.count_nonzero(a != b)} words differ"
        assert float(np.abs(a).max()) > 0.0
        # and against the oracle (same two dispatches)
        cpu = np.zeros((h, w, 4), np.float32)
        for k, (gx, gy) in enumerate([(w // 8, h // 8), (w // 16, h // 16 + 1)]):
            win, push = _push(wd, scene, w, h, k, gx, gy)
            ob.render(packed, push, cpu, gx, gy, trace=False)
        assert np.array_equal(b.view(np.uint32), cpu.view(np.uint32)), f"{case}: differs from the oracle"
        # a third call after a sync, full size again
        win, push = _push(wd, scene, w, h, 2)
        ref.compute_list_dispatch(push, win.work_group_width, win.work_group_height, 1)
        dut.compute_list_dispatch_noalloc(push, win.work_group_width, win.work_group_height, 1)
        ref.sync()
        dut.sync()
        a, b = ref.texture_get_data(dref.image_buffer), dut.texture_get_data(ddut.image_buffer)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    finally:
        for dev, d in ((ref, dref), (dut, ddut)):
            d.free_rids()
            dev.close()


def test_two_contexts_share_the_static_array(pta, golden_dir):
    """Two contexts on one GPU, each on its own stream, both through ptk_dispatch_noalloc back to back without a sync
    in between, several rounds: the library orders their launches on the shared array, so each image equals what
    ptk_dispatch gives (and the oracle)."""
    wd = pta.work_dispatcher
    scene, packed = deep_chain_scene(pta, depth=6)  # uses the spill part of the array as well as the counter
    w, h = 320, 184
    (a, da), (b, db) = _pair(pta, packed, w, h)
    try:
        win, push = _push(wd, scene, w, h, 0)
        for dev, d in ((a, da), (b, db)):
            dev.compute_list_dispatch(push, win.work_group_width, win.work_group_height, 1)
            dev.sync()
            dev.texture_update(d.image_buffer, np.zeros((h, w, 4), np.float32))
        free_before = _hip_free_bytes()
        rounds = 6
        for k in range(rounds):
            win, push = _push(wd, scene, w, h, k)
            a.compute_list_dispatch_noalloc(push, win.work_group_width, win.work_group_height, 1)
            b.compute_list_dispatch_noalloc(push, win.work_group_width, win.work_group_height, 1)
        a.sync()
        b.sync()
        assert _hip_free_bytes() == free_before, "device memory was allocated or freed"
        cpu = np.zeros((h, w, 4), np.float32)
        for k in range(rounds):
            win, push = _push(wd, scene, w, h, k)
            ob.render(packed, push, cpu, win.work_group_width, win.work_group_height, trace=False)
        for name, (dev, d) in (("first", (a, da)), ("second", (b, db))):
            img = dev.texture_get_data(d.image_buffer)
            assert np.array_equal(img.view(np.uint32), cpu.view(np.uint32)), f"{name} context: {np.count_nonzero(img != cpu)} words differ from the oracle"
    finally:
        for dev, d in ((a, da), (b, db)):
            d.free_rids()
            dev.close()


def test_noalloc_error_paths(pta, golden_dir):
    wd = pta.work_dispatcher
    scene = pta.scenes.load_fixture_scene("sphere_scene1", golden_dir)
    packed = wd.pack_scene(scene, ob.OracleBvhBuilder(), 8)
    w, h = 64, 64
    dev = pta.PtkRenderingDevice(0)
    try:
        d = wd.PTWorkDispatcher(dev)
        d.create_buffers(w, h, packed=packed)
        d.load_shader()
        win, push = _push(wd, scene, w, h, 0)
        with pytest.raises(pta.PtkError, match="ptk_dispatch once"):  # scene never validated: that would allocate
            dev.compute_list_dispatch_noalloc(push, 8, 8, 1)
        dev.compute_list_dispatch(push, 8, 8, 1)
        dev.compute_list_dispatch_noalloc(push, 8, 8, 1)
        # an edit that needs a rebuild: a node's transform index changes (and is set back, so the scene stays what it was)
        node0 = bytes(packed.bvh[:32])
        dev.buffer_update(d.bvh_buffer, 0, 32, node0[:28] + (5).to_bytes(4, "little"))
        dev.buffer_update(d.bvh_buffer, 0, 32, node0)
        with pytest.raises(pta.PtkError, match="ptk_dispatch once"):
            dev.compute_list_dispatch_noalloc(push, 8, 8, 1)
        dev.compute_list_dispatch(push, 8, 8, 1)
        normal_view = wd.PTRenderWindow(render_width=w, render_height=h, max_ray_depth=4, render_mode=2)
        with pytest.raises(pta.PtkError, match="plain render mode"):
            dev.compute_list_dispatch_noalloc(wd.push_constant_bytes(scene.camera, normal_view, 0.0), 8, 8, 1)
        with pytest.raises(pta.PtkError):
            dev.compute_list_dispatch_noalloc(push, 0, 8, 1)
        d.free_rids()
    finally:
        dev.close()
    # a tree whose stack need is beyond what the static array holds at one wave per CU: refused with the limit named
    scene, packed = deep_chain_scene(pta, depth=30)  # need 7 * 29 + 8 = 211 > 20 + 127
    dev = pta.PtkRenderingDevice(0)
    try:
        d = wd.PTWorkDispatcher(dev)
        d.create_buffers(256, 256, packed=packed)
        d.load_shader()
        win, push = _push(wd, scene, 256, 256, 0, depth=2)
        dev.compute_list_dispatch(push, 32, 32, 1)  # ptk_dispatch allocates what the tree needs
        dev.sync()
        with pytest.raises(pta.PtkError, match="static scratch"):
            dev.compute_list_dispatch_noalloc(push, 32, 32, 1)
        d.free_rids()
    finally:
        dev.close()
