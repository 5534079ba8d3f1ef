"""Writes tests/golden/render_crc.json: CRC-32 of the rgba32f image the CPU oracle renders for each fixture
scene (160x96, three progressive samples, fixed sample times), with the settings used.  The reference holds
no render vectors, so these pin the build's own oracle against unintended change; the GPU tests compare the
HIP renderers with the same numbers.  Run from the repository root after a deliberate change of the oracle."""
import importlib, json, os, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob
pta = importlib.import_module("path-tracer_amd")
wd = pta.work_dispatcher

CASES = [  # name, depth, refraction bounces, use_emission, refraction_mode
    ("sphere_scene1", 4, 4, False, 2),
    ("benchmark_random_scene", 8, 4, False, 2),
    ("fake_cornell_box", 8, 8, True, 2),
    ("sphere_scene3", 8, 8, False, 2),
    ("sphere_scene3", 6, 4, False, 0),
    ("sphere_scene3", 6, 4, False, 1),
    ("dragon_test", 4, 4, False, 2),
]
W, H, SAMPLES = 160, 96, 3


def render_case(name, depth, bounces, emission, mode, golden_dir):
    scene = pta.scenes.load_fixture_scene(name, golden_dir)
    packed = wd.pack_scene(scene, ob.OracleBvhBuilder(), 8)
    img = np.zeros((H, W, 4), np.float32)
    for k in range(SAMPLES):
        win = wd.PTRenderWindow(render_width=W, render_height=H, max_ray_depth=depth, max_refraction_bounces=bounces)
        win.frame = k
        ob.render(packed, wd.push_constant_bytes(scene.camera, win, wd.sample_time(k)), img, W // 8, H // 8,
                  refraction_mode=mode, use_emission=emission, nthreads=4)
    return scene, packed, img


if __name__ == "__main__":
    golden = os.path.join(ROOT, "tests", "golden")
    out = {"width": W, "height": H, "samples": SAMPLES, "cases": []}
    for name, depth, bounces, emission, mode in CASES:
        _, packed, img = render_case(name, depth, bounces, emission, mode, golden)
        out["cases"].append({"scene": name, "depth": depth, "refraction_bounces": bounces, "use_emission": emission,
                             "refraction_mode": mode, "image_crc32": "%08x" % zlib.crc32(img.tobytes()),
                             "bvh_crc32": "%08x" % zlib.crc32(packed.bvh), "mean_rgba": [float(x) for x in img.mean(axis=(0, 1))]})
        print(out["cases"][-1])
    with open(os.path.join(golden, "render_crc.json"), "w") as f:
        json.dump(out, f, indent=1)
