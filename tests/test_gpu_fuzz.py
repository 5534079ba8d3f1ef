"""GPU parity on seeded random scenes: overlapping dielectric spheres with random priorities, metals,
emitters, free triangles, planes, and meshes under rotated / non-uniformly scaled transforms, built with a
random BVH type, at random depths and refraction-tracking modes.  Same bar as test_gpu_parity: primary
ids, per-pixel counters and images bit-identical to the CPU oracle, for the one-kernel renderer and the
wavefront pipeline."""
import math

import os

import numpy as np
import pytest

from test_gpu_parity import _run_case, rd, rd_wf  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu


def _rotation(rng):
    """A random rotation times a random non-uniform scale, as the nine row-major basis values."""
    a, b, c = rng.uniform(0.0, 2.0 * math.pi, 3)
    rx = np.array([[1, 0, 0], [0, math.cos(a), -math.sin(a)], [0, math.sin(a), math.cos(a)]])
    ry = np.array([[math.cos(b), 0, math.sin(b)], [0, 1, 0], [-math.sin(b), 0, math.cos(b)]])
    rz = np.array([[math.cos(c), -math.sin(c), 0], [math.sin(c), math.cos(c), 0], [0, 0, 1]])
    scale = np.diag(rng.uniform(0.5, 1.8, 3))
    return (rz @ ry @ rx @ scale).reshape(-1).tolist()


def _random_description(seed):
    rng = np.random.default_rng(seed)
    mats = {}
    for i in range(8):
        kind = rng.integers(0, 4)
        m = dict(albedo=rng.uniform(0.1, 1.0, 3).tolist(), roughness=0.0, metallic=0.0, opacity=1.0, reflectivity=1.0,
                 IOR=1.0, dielectric_priority=0, is_emissive=0)
        if kind == 1:
            m["metallic"] = float(rng.uniform(0.2, 1.0))
        elif kind == 2:  # dielectric, some of them "air" cut-outs (IOR 1)
            m["opacity"] = float(rng.choice([0.0, 0.0, 0.4]))
            m["IOR"] = float(rng.choice([1.0, 1.33, 1.5, 1.6, 2.4]))
            m["reflectivity"] = float(rng.uniform(0.0, 1.0))
            m["dielectric_priority"] = int(rng.integers(0, 4))
        elif kind == 3 and i % 2 == 0:
            m["albedo"] = rng.uniform(2.0, 15.0, 3).tolist()
            m["is_emissive"] = 1
        mats[f"m{i}"] = m
    names = list(mats) + [None]

    def pick():
        return names[rng.integers(0, len(names))]

    def sphere(spread):
        return dict(kind="sphere", radius=float(rng.uniform(0.2, 1.4)), material=pick(),
                    transform=[1, 0, 0, 0, 1, 0, 0, 0, 1] + rng.uniform(-spread, spread, 3).tolist())

    def triangle(spread):
        c = rng.uniform(-spread, spread, 3)
        v = [(c + rng.uniform(-1.2, 1.2, 3)).tolist() for _ in range(3)]
        return dict(kind="triangle", vertex1=v[0], vertex2=v[1], vertex3=v[2], material=pick())

    nodes = []
    for _ in range(int(rng.integers(1, 40))):
        nodes.append(sphere(3.5))
    for _ in range(int(rng.integers(0, 60))):
        nodes.append(triangle(3.5))
    for _ in range(int(rng.integers(0, 3))):
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        nodes.append(dict(kind="plane", normal=n.tolist(), distance=float(rng.uniform(-5.0, -2.0)), material=pick()))
    for _ in range(int(rng.integers(0, 3))):
        children = [sphere(1.5) for _ in range(int(rng.integers(0, 12)))] + [triangle(1.5) for _ in range(int(rng.integers(1, 30)))]
        nodes.append(dict(kind="mesh", transform=_rotation(rng) + rng.uniform(-3.0, 3.0, 3).tolist(),
                          bvh_type=int(rng.integers(1, 5)), bvh_order=int(rng.choice([2, 4, 8])), children=children,
                          override_material=pick() if rng.random() < 0.3 else None))
    return dict(bvh_order=8, default_bvh=int(rng.integers(1, 5)), materials=mats, nodes=nodes), rng


# (PTK_FUZZ_SEEDS=n runs more of them: a soak, not part of the regular suite)
@pytest.mark.parametrize("seed", range(int(os.environ.get("PTK_FUZZ_SEEDS", "40"))))
def test_random_scene(pta, rd, rd_wf, golden_dir, seed):
    desc, rng = _random_description(1000 + seed)
    scene = pta.scenes.scene_from_description(desc)
    eye = rng.uniform(-1.0, 1.0, 3)
    eye = eye / np.linalg.norm(eye) * rng.uniform(0.5, 9.0)  # sometimes inside the cloud of objects
    scene.camera = pta.scene.PTCamera(transform=pta.godot_math.Transform3D.looking_at(tuple(eye.tolist()), (0.0, 0.0, 0.0)),
                                      fov=float(rng.uniform(30.0, 90.0)))
    _run_case(pta, rd, rd_wf, golden_dir, f"fuzz{seed}", 160, 96, int(rng.integers(1, 9)), int(rng.integers(0, 9)),
              emission=bool(rng.random() < 0.3), dispatches=2, refraction_mode=int(rng.integers(0, 3)), scene=scene,
              bvh_type=int(desc["default_bvh"]))


@pytest.mark.parametrize("seed", range(6))
def test_singular_mesh_transform(pta, rd, rd_wf, golden_dir, seed):
    """A mesh squashed flat (one scale factor 0): its inverse matrix holds Inf / NaN, so rays that reach the
    mesh socket get NaN directions and ranges.  Whatever the shader's arithmetic makes of that, the HIP
    renderers make the same of it as the oracle (NaN texels must be NaN on both sides)."""
    desc, rng = _random_description(2000 + seed)
    children = [dict(kind="sphere", radius=0.7, material="m1", transform=[1, 0, 0, 0, 1, 0, 0, 0, 1, 0.2, 0.1, 0.0])]
    for _ in range(12):
        c = rng.uniform(-1.0, 1.0, 3)
        children.append(dict(kind="triangle", vertex1=(c + rng.uniform(-1, 1, 3)).tolist(), vertex2=(c + rng.uniform(-1, 1, 3)).tolist(),
                             vertex3=(c + rng.uniform(-1, 1, 3)).tolist(), material="m2"))
    basis = [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0]
    basis[4 * (seed % 3)] = 0.0  # zero scale on one axis
    desc["nodes"].append(dict(kind="mesh", transform=basis + [0.0, 0.0, 0.0], bvh_type=4, bvh_order=8, children=children))
    with np.errstate(all="ignore"):
        scene = pta.scenes.scene_from_description(desc)
        scene.camera = pta.scene.PTCamera(transform=pta.godot_math.Transform3D.looking_at((1.0, 2.0, 7.0), (0.0, 0.0, 0.0)), fov=60.0)
        _run_case(pta, rd, rd_wf, golden_dir, f"singular{seed}", 160, 96, 5, 4, dispatches=2, scene=scene, bvh_type=int(desc["default_bvh"]))


@pytest.mark.parametrize("seed", range(8))
def test_degenerate_and_coincident_triangles(pta, rd, rd_wf, golden_dir, seed):
    """Triangles a leaf's tests have to get right by the letter: zero area (two corners equal, three in a line), exact
    copies under other materials (equal distance: the later one in the leaf wins), nearly coincident ones, corners so
    close that the products of their edges are denormal, and corners so far out that they overflow -- in clusters of up
    to eight, so that they share leaves, long enough to be split between a lane and its helper."""
    desc, rng = _random_description(3000 + seed)
    names = [n for n in desc["materials"]]
    nodes = desc["nodes"]

    def tri(v1, v2, v3, m):
        return dict(kind="triangle", vertex1=[float(x) for x in v1], vertex2=[float(x) for x in v2], vertex3=[float(x) for x in v3], material=m)

    for _ in range(10):
        c = rng.uniform(-2.5, 2.5, 3)
        a, b, d = (c + rng.uniform(-1.0, 1.0, 3) for _ in range(3))
        kind = int(rng.integers(0, 6))
        for k in range(int(rng.integers(2, 9))):
            m = names[int(rng.integers(0, len(names)))]
            if kind == 0:    # exact copies
                nodes.append(tri(a, b, d, m))
            elif kind == 1:  # zero area among real ones
                nodes.append(tri(a, a, d, m) if k % 3 == 0 else tri(a, (a + d) * 0.5, d, m) if k % 3 == 1 else tri(a, b, d, m))
            elif kind == 2:  # one float apart
                nodes.append(tri(np.nextafter(a.astype(np.float32), np.float32(9), dtype=np.float32) if k & 1 else a, b, d, m))
            elif kind == 3:  # edges whose products are denormal
                e = np.float32(1e-20)
                nodes.append(tri(a, a + np.array([e, 0, 0]) * (k + 1), a + np.array([0, e, 0]) * (k + 1), m))
            elif kind == 4:  # a fan around one edge: equal distances along it
                nodes.append(tri(a, b, c + rng.uniform(-1.0, 1.0, 3), m))
            else:            # corners far out: edge products overflow
                nodes.append(tri(a, a + np.array([3e19, 0, 0]), a + np.array([0, 3e19 * (1 + k), 1.0]), m))
    with np.errstate(all="ignore"):
        scene = pta.scenes.scene_from_description(desc)
        eye = rng.uniform(-1.0, 1.0, 3)
        eye = eye / np.linalg.norm(eye) * rng.uniform(3.0, 9.0)
        scene.camera = pta.scene.PTCamera(transform=pta.godot_math.Transform3D.looking_at(tuple(eye.tolist()), (0.0, 0.0, 0.0)),
                                          fov=float(rng.uniform(40.0, 80.0)))
        _run_case(pta, rd, rd_wf, golden_dir, f"degenerate{seed}", 160, 96, int(rng.integers(2, 7)), 4, dispatches=2, scene=scene,
                  bvh_type=int(desc["default_bvh"]))
