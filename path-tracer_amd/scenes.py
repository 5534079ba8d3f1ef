"""Scene construction: from the extracted fixture descriptions (tests/golden/scenes/*.json)
and from seeded generators for the large synthetic workloads.

Registration order follows the Godot scene tree: every node enters the tree in
pre-order (PTObject._enter_tree, pt_object.gd:88-118), then PTMesh._ready adds
imported triangles and registers the mesh (pt_mesh.gd:72-106)."""
from __future__ import annotations

import json
import os
from typing import Dict, Optional

import numpy as np

from . import scene as sc
from .godot_math import Transform3D


def _transform(e) -> Transform3D:
    return Transform3D.from_tscn(e["transform"]) if "transform" in e else Transform3D()


def _material(d) -> sc.PTMaterial:
    return sc.PTMaterial(
        albedo=tuple(d["albedo"]), roughness=d["roughness"], metallic=d["metallic"], opacity=d["opacity"],
        reflectivity=d["reflectivity"], IOR=d["IOR"], dielectric_priority=d["dielectric_priority"],
        is_emissive=bool(d["is_emissive"]),
    )


def scene_from_description(desc: dict, base_dir: Optional[str] = None) -> sc.PTScene:
    scene = sc.PTScene(bvh_order=desc.get("bvh_order", 8), default_bvh=desc.get("default_bvh", sc.BVH_X_SORTED))
    mats: Dict[str, sc.PTMaterial] = {k: _material(v) for k, v in desc.get("materials", {}).items()}
    texs: Dict[str, sc.PTTextureAbstract] = {}
    for k, v in desc.get("textures", {}).items():
        texs[k] = sc.PTProceduralTexture(is_spatial=bool(v["is_spatial"]), texture_path=v["texture_path"])

    def make_object(e):
        common = dict(transform=_transform(e), material=mats.get(e.get("material")), texture=texs.get(e.get("texture")))
        if e["kind"] == "sphere":
            return sc.PTSphere(radius=e["radius"], **common)
        if e["kind"] == "plane":
            return sc.PTPlane(normal=tuple(e["normal"]), distance=e["distance"], **common)
        if e["kind"] == "triangle":
            return sc.PTTriangle(vertex1=tuple(e["vertex1"]), vertex2=tuple(e["vertex2"]), vertex3=tuple(e["vertex3"]), **common)
        raise ValueError(e["kind"])

    pending_meshes = []
    for e in desc["nodes"]:
        k = e["kind"]
        if k == "camera":
            if scene.camera is None:
                scene.camera = sc.PTCamera(transform=_transform(e), fov=e.get("fov", 75.0))
        elif k == "mesh":
            mesh = sc.PTMesh(transform=_transform(e), bvh_type=e.get("bvh_type", 1), bvh_order=e.get("bvh_order", 8),
                             default_material=mats.get(e.get("default_material")),
                             override_material=mats.get(e.get("override_material")))
            mesh.scene = scene
            for c in e.get("children", []):
                obj = make_object(c)
                mesh.add_object(obj)  # PTMesh.add_object -> scene.add_object (pt_mesh.gd:117-131)
            if "triangles_file" in e:
                tris = np.load(os.path.join(base_dir, e["triangles_file"]))
                mesh.triangle_set = sc.PTTriangleSet(vertices=tris.astype(np.float32))
            pending_meshes.append(mesh)
        else:
            scene.add_object(make_object(e))
    for mesh in pending_meshes:  # _ready runs after every _enter_tree
        scene.add_mesh(mesh)
    cam_names = sc.PTCamera.camera_setting_names
    start = desc.get("starting_camera", 0)
    if scene.camera is None:
        scene.camera = sc.PTCamera()
    if start:
        scene.camera.set_camera_setting(cam_names[start])
    return scene


def load_fixture_scene(name: str, golden_dir: str) -> sc.PTScene:
    base = os.path.join(golden_dir, "scenes")
    with open(os.path.join(base, name + ".json")) as f:
        desc = json.load(f)
    return scene_from_description(desc, base)


# ---- seeded synthetic scenes ------------------------------------------------------------
def _splitmix64(seed: int):
    state = seed & 0xFFFFFFFFFFFFFFFF

    def nxt() -> int:
        nonlocal state
        state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    return nxt


def random_spheres_scene(grid: int = 11, seed: int = 0) -> sc.PTScene:
    """Scaled variant of PTScene.create_random_scene (pt_scene.gd:668-726): a ground plane, three
    large spheres and (2*grid)^2 small ones, ~80 % diffuse / 15 % metal / 5 % glass.  Godot's
    RandomNumberGenerator is replaced by splitmix64, so the reference's own 22x22 scene is the
    benchmark_random_scene fixture, not this generator."""
    nxt = _splitmix64(seed)

    def randf():
        return (nxt() >> 11) * (1.0 / (1 << 53))

    def rr(a, b):
        return a + (b - a) * randf()

    scene = sc.PTScene()
    scene.add_object(sc.PTPlane(normal=(0.0, 1.0, 0.0), distance=-1.0, material=sc.PTMaterial(albedo=(0.5, 0.5, 0.5))))
    scene.add_object(sc.PTSphere(transform=Transform3D(origin=(0, 1, 0)), radius=1.0, material=sc.PTMaterial(IOR=1.5, opacity=0.0)))
    scene.add_object(sc.PTSphere(transform=Transform3D(origin=(-4, 1, 0)), radius=1.0, material=sc.PTMaterial(albedo=(0.4, 0.2, 0.1))))
    scene.add_object(sc.PTSphere(transform=Transform3D(origin=(4, 1, 0)), radius=1.0, material=sc.PTMaterial(albedo=(0.7, 0.6, 0.5), metallic=1.0)))
    for i in range(2 * grid):
        for j in range(2 * grid):
            center = ((i - grid) + 0.9 * randf(), 0.2, (j - grid) + 0.9 * randf())
            choose = randf()
            m = sc.PTMaterial()
            color = (randf(), randf(), randf())
            if choose < 0.8:
                pass
            elif choose < 0.95:
                color = (rr(0.5, 1.0), rr(0.5, 1.0), rr(0.5, 1.0))
                m.metallic = choose + 0.05
            else:
                color = (rr(0.88, 1.0), rr(0.88, 1.0), rr(0.88, 1.0))
                m.opacity = 0.0
                m.IOR = 1.6
            m.albedo = color
            scene.add_object(sc.PTSphere(transform=Transform3D(origin=center), radius=0.2, material=m))
    scene.camera = sc.PTCamera()
    scene.camera.set_camera_setting("book_ex")
    return scene


def _quad_grid(origin, du, dv, nu, nv):
    """Two triangles per cell of an nu x nv grid spanned by du, dv from origin: [nu*nv*2, 3, 3]."""
    o = np.asarray(origin, dtype=np.float64)
    du = np.asarray(du, dtype=np.float64)
    dv = np.asarray(dv, dtype=np.float64)
    iu, iv = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    p00 = o + iu[..., None] * du + iv[..., None] * dv
    p10, p01, p11 = p00 + du, p00 + dv, p00 + du + dv
    t1 = np.stack([p00, p10, p11], axis=-2)
    t2 = np.stack([p00, p11, p01], axis=-2)
    return np.concatenate([t1.reshape(-1, 3, 3), t2.reshape(-1, 3, 3)], axis=0)


def atrium_triangles(n_triangles: int = 262144, seed: int = 7) -> np.ndarray:
    """Seeded stand-in for the Crytek Sponza atrium (the reference's sponza-crytek.blend is not in the
    checkout): an enclosed hall -- floor, ceiling, four walls as fine quad grids -- two storeys of
    colonnades (prism columns), arches between columns, hanging drapes (wavy grids) and small clutter.
    Sizes mix uniform grids with clustered small triangles, like an architectural mesh.  Returns exactly
    n_triangles triangles, [N,3,3] float32, in a fixed order."""
    rng = np.random.default_rng(seed)
    L, W, H = 36.0, 16.0, 14.0  # hall extent: x in [-L/2, L/2], z in [-W/2, W/2], y in [0, H]
    parts = []
    shell = int(n_triangles * 0.30)
    # shell: 6 faces with cell counts proportional to area
    faces = [
        ((-L / 2, 0, -W / 2), (L, 0, 0), (0, 0, W)),      # floor
        ((-L / 2, H, -W / 2), (0, 0, W), (L, 0, 0)),      # ceiling
        ((-L / 2, 0, -W / 2), (0, H, 0), (L, 0, 0)),      # wall z-
        ((-L / 2, 0, W / 2), (L, 0, 0), (0, H, 0)),       # wall z+
        ((-L / 2, 0, -W / 2), (0, 0, W), (0, H, 0)),      # wall x-
        ((L / 2, 0, -W / 2), (0, H, 0), (0, 0, W)),       # wall x+
    ]
    areas = np.array([np.linalg.norm(np.cross(f[1], f[2])) for f in faces])
    for f, a in zip(faces, areas):
        cells = max(1, int(shell * a / areas.sum() / 2))
        lu, lv = np.linalg.norm(f[1]), np.linalg.norm(f[2])
        nu = max(1, int(round(np.sqrt(cells * lu / lv))))
        nv = max(1, cells // nu)
        g = _quad_grid(f[0], np.array(f[1]) / nu, np.array(f[2]) / nv, nu, nv)
        if f[0][1] == H:  # open roof over the courtyard, like the real atrium: the sky lights the hall
            c = g.mean(axis=1)
            g = g[~((np.abs(c[:, 0]) < L / 2 - 6.0) & (np.abs(c[:, 2]) < W / 2 - 4.0))]
        parts.append(g)
    # columns: two rows per storey, prisms subdivided vertically; detail scales with the budget
    n_cols = 14
    col_budget = int(n_triangles * 0.22)
    segs = 12 if n_triangles >= 65536 else 6
    rings = max(1, col_budget // (4 * n_cols * segs * 2))
    col = []
    for storey, (y0, y1) in enumerate([(0.0, 6.0), (6.5, 12.0)]):
        for side in (-1, 1):
            for c in range(n_cols):
                cx = -L / 2 + 2.0 + c * (L - 4.0) / (n_cols - 1)
                cz = side * (W / 2 - 3.0)
                r = 0.45 - 0.1 * storey
                ang = np.linspace(0, 2 * np.pi, segs + 1)
                ys = np.linspace(y0, y1, rings + 1)
                a0, ya = np.meshgrid(ang[:-1], ys[:-1], indexing="ij")
                a1, yb = np.meshgrid(ang[1:], ys[1:], indexing="ij")

                def p(a, y):
                    return np.stack([cx + r * np.cos(a), y, cz + r * np.sin(a)], axis=-1)

                col.append(np.stack([p(a0, ya), p(a1, ya), p(a1, yb)], axis=-2).reshape(-1, 3, 3))
                col.append(np.stack([p(a0, ya), p(a1, yb), p(a0, yb)], axis=-2).reshape(-1, 3, 3))
    parts.append(np.concatenate(col))
    # arches between neighbouring columns (half rings, two strips deep)
    arch_budget = int(n_triangles * 0.04)
    steps = max(2, arch_budget // (4 * (n_cols - 1) * 4))
    arch = []
    for y0 in (5.0, 11.0):
        for side in (-1, 1):
            cz = side * (W / 2 - 3.0)
            for c in range(n_cols - 1):
                x0 = -L / 2 + 2.0 + c * (L - 4.0) / (n_cols - 1)
                x1 = -L / 2 + 2.0 + (c + 1) * (L - 4.0) / (n_cols - 1)
                xm, rad = 0.5 * (x0 + x1), 0.5 * (x1 - x0) - 0.3
                th = np.linspace(0, np.pi, steps + 1)
                t0, t1 = th[:-1], th[1:]
                for zo in (-0.35, 0.0):
                    def q(t, z):
                        return np.stack([xm + rad * np.cos(t), y0 + rad * np.sin(t), np.full_like(t, cz + z)], axis=-1)

                    pa, pb, pc, pd = q(t0, zo), q(t1, zo), q(t1, zo + 0.35), q(t0, zo + 0.35)
                    arch.append(np.stack([pa, pb, pc], axis=-2))
                    arch.append(np.stack([pa, pc, pd], axis=-2))
    parts.append(np.concatenate(arch))
    used = sum(p.shape[0] for p in parts)
    # drapes: wavy vertical sheets across the hall, fine grids
    n_drapes = 6
    drape_budget = int((n_triangles - used) * 0.55)
    per = drape_budget // n_drapes
    for dgi in range(n_drapes):
        x = -L / 2 + 4.0 + dgi * (L - 8.0) / (n_drapes - 1)
        nu = max(2, int(np.sqrt(max(per, 8) / 2 * 2.0)))
        nv = max(2, (per // 2) // nu)
        u = np.linspace(-2.5, 2.5, nu + 1)
        v = np.linspace(7.0, 12.5, nv + 1)
        uu, vv = np.meshgrid(u, v, indexing="ij")
        xx = x + 0.25 * np.sin(uu * 3.0 + dgi) * (12.5 - vv) / 5.5
        P = np.stack([xx, vv, uu], axis=-1)
        p00, p10, p01, p11 = P[:-1, :-1], P[1:, :-1], P[:-1, 1:], P[1:, 1:]
        parts.append(np.concatenate([np.stack([p00, p10, p11], axis=-2).reshape(-1, 3, 3),
                                     np.stack([p00, p11, p01], axis=-2).reshape(-1, 3, 3)]))
    used = sum(p.shape[0] for p in parts)
    # clutter: clusters of small random triangles (vases, foliage) near the floor and ledges
    rest = n_triangles - used
    assert rest >= 0, (n_triangles, used)
    if rest > 0:
        n_clusters = 48
        centers = np.stack([rng.uniform(-L / 2 + 1, L / 2 - 1, n_clusters), rng.choice([0.6, 6.9], n_clusters),
                            rng.uniform(-W / 2 + 1, W / 2 - 1, n_clusters)], axis=1)
        which = rng.integers(0, n_clusters, rest)
        base = centers[which] + rng.normal(0, 0.35, (rest, 3)) * np.array([1.0, 0.6, 1.0])
        e1 = rng.normal(0, 0.06, (rest, 3))
        e2 = rng.normal(0, 0.06, (rest, 3))
        parts.append(np.stack([base, base + e1, base + e2], axis=1))
    tris = np.concatenate(parts)[:n_triangles]
    assert tris.shape[0] == n_triangles, tris.shape
    # No surface is exactly planar: every vertex moves by a few millimetres, by an amount that depends
    # only on its position (shared vertices stay shared, the hall stays closed).  The reference's
    # builder bins a node by its box centre inside the list's box grown by 1e-6; a triangle that is
    # exactly flat on the outermost face of the hall sits on the upper edge of the last bin and trips
    # the builder's `index < segment_count` assert (pt_bvh_axis_sort.gd:307-308).
    q = np.round(tris * 1024.0).astype(np.int64)
    hsh = (q[..., 0] * 73856093) ^ (q[..., 1] * 19349663) ^ (q[..., 2] * 83492791) ^ (seed * 2654435761)
    jit = np.stack([((hsh >> s) & 1023) for s in (3, 17, 31)], axis=-1).astype(np.float64)
    tris = tris + (jit / 1023.0 - 0.5) * 0.012
    return tris.astype(np.float32)


def atrium_scene(n_triangles: int = 262144, seed: int = 7, n_materials: int = 64) -> sc.PTScene:
    """One PTMesh (XYZ_SORTED, order 8, like scenes/sponza_crytek_import.tscn:20-26) holding the atrium;
    n_materials diffuse/metal materials with the spatial checker texture."""
    rng = np.random.default_rng(seed + 1)
    tris = atrium_triangles(n_triangles, seed)
    palette = []
    for i in range(n_materials):
        c = tuple(float(x) for x in rng.uniform(0.35, 0.95, 3))
        palette.append(sc.PTMaterial(albedo=c, metallic=float(rng.uniform(0.6, 1.0)) if i % 8 == 7 else 0.0))
    # material by spatial block so neighbouring triangles share one (first-use order = palette order)
    cen = tris.mean(axis=1)
    block = (np.floor(cen[:, 0] / 4.5).astype(np.int64) * 7 + np.floor(cen[:, 1] / 3.5).astype(np.int64) * 3
             + np.floor(cen[:, 2] / 4.0).astype(np.int64))
    raw = np.mod(block, n_materials).astype(np.int32)
    # renumber by first appearance so that registration order equals palette order
    _, first_idx = np.unique(raw, return_index=True)
    order = raw[np.sort(first_idx)]
    remap = np.full(n_materials, -1, dtype=np.int32)
    remap[order] = np.arange(order.shape[0], dtype=np.int32)
    pal_idx = remap[raw]
    palette = palette[: order.shape[0]]
    tex = sc.PTProceduralTexture(is_spatial=True, texture_path="res://shaders/procedural_textures/checker_board_spatial.comp")
    scene = sc.PTScene(bvh_order=8, default_bvh=sc.BVH_XYZ_SORTED)
    mesh = sc.PTMesh(bvh_type=sc.BVH_XYZ_SORTED, bvh_order=8)
    mesh.scene = scene
    mesh.triangle_set = sc.PTTriangleSet(vertices=tris, palette=palette, palette_index=pal_idx, texture=tex)
    scene.add_mesh(mesh)
    scene.camera = sc.PTCamera(transform=Transform3D.looking_at((-15.0, 4.0, 1.5), (10.0, 5.0, -1.0)), fov=60.0)
    return scene
