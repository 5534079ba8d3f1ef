#!/usr/bin/env python3
"""Turns the reference's scene files into small data fixtures under tests/golden/.

Run in the build container only (it reads /root/reference, which does not exist
on the GPU box):

    python tools/extract_fixtures.py

Output: tests/golden/scenes/<name>.json (scene description: nodes in tree order
with their transforms and exported properties, materials, textures) and, for
imported meshes, <name>_tris.npy ([N,3,3] float32 triangle vertices decoded
from the ArrayMesh surface the way Godot's surface_get_arrays does).  Only data
is copied -- numbers that the .tscn/.tres files hold -- never script text.
"""
from __future__ import annotations

import base64
import json
import os
import re
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "scenes")

SCRIPT_KIND = {
    "pt_scene.gd": "scene", "pt_sphere.gd": "sphere", "pt_plane.gd": "plane", "pt_triangle.gd": "triangle",
    "pt_mesh.gd": "mesh", "pt_camera.gd": "camera", "pt_camera_fps.gd": "camera",
    "pt_material.gd": "material", "pt_procedural_texture.gd": "procedural_texture",
    "pt_sampled_texture.gd": "sampled_texture",
}


def res_path(p: str) -> str:
    assert p.startswith("res://")
    return os.path.join(REF, p[len("res://"):])


def parse_value(v: str):
    v = v.strip()
    m = re.match(r"^(Transform3D|Vector3|Vector4|Color|AABB)\((.*)\)$", v, re.S)
    if m:
        return {"_t": m.group(1), "v": [float(x) for x in m.group(2).split(",")]}
    m = re.match(r'^(ExtResource|SubResource)\("([^"]+)"\)$', v)
    if m:
        return {"_t": m.group(1), "id": m.group(2)}
    if v in ("true", "false"):
        return v == "true"
    if v.startswith('"') and v.endswith('"'):
        return v[1:-1]
    try:
        return int(v)
    except ValueError:
        pass
    try:
        return float(v)
    except ValueError:
        return {"_raw": v}


def parse_tscn(path: str):
    text = open(path, encoding="utf-8").read()
    sections = []
    cur = None
    # split on section headers at line starts; property values may span lines (arrays/dicts)
    lines = text.split("\n")
    i = 0
    while i < len(lines):
        line = lines[i]
        hm = re.match(r"^\[(\w+)(.*)\]\s*$", line)
        if hm and not line.startswith("[{"):
            attrs = dict(re.findall(r'(\w+)=("[^"]*"|\w+\("[^"]*"\)|[^\s\]]+)', hm.group(2)))
            cur = {"kind": hm.group(1), "attrs": {k: parse_value(v) for k, v in attrs.items()}, "props": {}, "raw": {}}
            sections.append(cur)
            i += 1
            continue
        pm = re.match(r"^([\w/]+) = (.*)$", line)
        if pm and cur is not None:
            key, val = pm.group(1), pm.group(2)
            # multi-line values: balance brackets
            depth = val.count("[") + val.count("{") + val.count("(") - val.count("]") - val.count("}") - val.count(")")
            while depth > 0 and i + 1 < len(lines):
                i += 1
                val += "\n" + lines[i]
                depth += lines[i].count("[") + lines[i].count("{") + lines[i].count("(") - lines[i].count("]") - lines[i].count("}") - lines[i].count(")")
            cur["raw"][key] = val
            cur["props"][key] = parse_value(val) if "\n" not in val and len(val) < 4096 else {"_raw": None}
        i += 1
    return sections


def load_resource_file(path: str):
    secs = parse_tscn(path)
    ext = {s["attrs"]["id"]: s["attrs"] for s in secs if s["kind"] == "ext_resource"}
    res = next(s for s in secs if s["kind"] == "resource")
    script = ext[res["props"]["script"]["id"]]["path"]
    return SCRIPT_KIND[os.path.basename(script)], res["props"]


def material_dict(props) -> dict:
    c = props.get("albedo", {"v": [1, 1, 1, 1]})["v"]
    return {
        "albedo": c[:3],
        "roughness": props.get("roughness", 0.0),
        "metallic": props.get("metallic", 0.0),
        "opacity": props.get("opacity", 1.0),
        "reflectivity": props.get("reflectivity", 1.0),
        "IOR": props.get("IOR", 1.0),
        "dielectric_priority": props.get("dielectric_priority", 0),
        "is_emissive": props.get("is_emissive", False),
    }


def decode_array_mesh(raw_surfaces: str) -> np.ndarray:
    """Triangles of surface 0 the way RenderingServer.mesh_surface_get_arrays decodes them."""
    def grab(key):
        m = re.search(r'"%s": PackedByteArray\("([^"]*)"\)' % key, raw_surfaces)
        return base64.b64decode(m.group(1)) if m else None

    fmt = int(re.search(r'"format": (\d+)', raw_surfaces).group(1))
    vcount = int(re.search(r'"vertex_count": (\d+)', raw_surfaces).group(1))
    icount = int(re.search(r'"index_count": (\d+)', raw_surfaces).group(1))
    aabb = [float(x) for x in re.search(r'"aabb": AABB\(([^)]*)\)', raw_surfaces).group(1).split(",")]
    vdata, idata = grab("vertex_data"), grab("index_data")
    compressed = bool(fmt & (1 << 29))  # ARRAY_FLAG_COMPRESS_ATTRIBUTES
    if compressed:
        u = np.frombuffer(vdata, dtype="<u2", count=vcount * 4).reshape(vcount, 4)[:, :3]
        # Vector3(float(v)/65535.0 ...) -> real_t, then vec * aabb.size + aabb.position in float32
        vec = (u.astype(np.float32).astype(np.float64) / 65535.0).astype(np.float32)
        pos = np.array(aabb[:3], dtype=np.float32)
        size = np.array(aabb[3:], dtype=np.float32)
        verts = (vec * size + pos).astype(np.float32)
    else:
        verts = np.frombuffer(vdata, dtype="<f4", count=vcount * 3).reshape(vcount, 3).copy()
    idx = np.frombuffer(idata, dtype="<u2" if vcount <= 65536 else "<u4", count=icount)
    return verts[idx.astype(np.int64)].reshape(-1, 3, 3).astype(np.float32)


def raw_surface(raw_surfaces: str) -> dict:
    """Surface 0 of an ArrayMesh as the .tscn stores it (the dictionary RenderingServer.mesh_add_surface takes):
    format flags, counts, aabb, uv_scale and the vertex / attribute / index byte blocks, undecoded."""
    def grab(key):
        m = re.search(r'"%s": PackedByteArray\("([^"]*)"\)' % key, raw_surfaces)
        return np.frombuffer(base64.b64decode(m.group(1)), dtype=np.uint8) if m else np.zeros(0, np.uint8)

    def vec(key, ctor):
        m = re.search(r'"%s": %s\(([^)]*)\)' % (key, ctor), raw_surfaces)
        return np.array([float(x) for x in m.group(1).split(",")], dtype=np.float32) if m else None

    uv_scale = vec("uv_scale", "Vector4")
    return {
        "format": np.uint64(int(re.search(r'"format": (\d+)', raw_surfaces).group(1))),
        "vertex_count": np.int32(int(re.search(r'"vertex_count": (\d+)', raw_surfaces).group(1))),
        "index_count": np.int32(int(re.search(r'"index_count": (\d+)', raw_surfaces).group(1))),
        "aabb": vec("aabb", "AABB"),
        "uv_scale": uv_scale if uv_scale is not None else np.zeros(4, np.float32),
        "vertex_data": grab("vertex_data"), "attribute_data": grab("attribute_data"), "index_data": grab("index_data"),
    }


MESH_SURFACES = [  # (fixture name, scene file, name of the ArrayMesh sub-resource's owner node type)
    ("dragon", "test_models/dragon low poly/dragon_import.tscn"),
    ("grimchild", "test_models/grimchild/grimchild_test.tscn"),
]


def extract_mesh_surfaces():
    out_dir = os.path.join(OUT, "..", "meshes")
    os.makedirs(out_dir, exist_ok=True)
    for name, rel in MESH_SURFACES:
        secs = parse_tscn(os.path.join(REF, rel))
        meshes = [s for s in secs if s["kind"] == "sub_resource" and s["attrs"].get("type") == "ArrayMesh" and "_surfaces" in s["raw"]]
        # the render mesh is the sub-resource with the most vertices (a welded shadow mesh may precede it)
        surf = max((raw_surface(m["raw"]["_surfaces"]) for m in meshes), key=lambda d: int(d["vertex_count"]))
        np.savez_compressed(os.path.join(out_dir, name + "_surface.npz"), **surf)
        print(name, "surface: format", hex(int(surf["format"])), "vertices", int(surf["vertex_count"]), "indices", int(surf["index_count"]))


class SceneExtractor:
    def __init__(self, path: str):
        self.path = path
        self.secs = parse_tscn(path)
        self.ext = {s["attrs"]["id"]: s["attrs"] for s in self.secs if s["kind"] == "ext_resource"}
        self.sub = {s["attrs"]["id"]: s for s in self.secs if s["kind"] == "sub_resource"}
        self.materials = {}
        self.textures = {}

    def kind_of_script(self, ref):
        if not ref:
            return None
        path = self.ext[ref["id"]]["path"]
        return SCRIPT_KIND.get(os.path.basename(path))

    def resource_key(self, ref):
        """Registers a material/texture resource, returns its key (shared resources share a key)."""
        if ref is None:
            return None
        if ref["_t"] == "ExtResource":
            path = self.ext[ref["id"]]["path"]
            key = path
            kind, props = load_resource_file(res_path(path))
        else:
            s = self.sub[ref["id"]]
            key = os.path.basename(self.path) + "::" + ref["id"]
            kind, props = self.kind_of_script(s["props"].get("script")), s["props"]
        if kind == "material":
            self.materials.setdefault(key, material_dict(props))
        elif kind == "procedural_texture":
            self.textures.setdefault(key, {"kind": "procedural", "texture_path": props.get("texture_path", ""),
                                          "is_spatial": props.get("is_spatial", False)})
        else:
            raise ValueError(f"unsupported resource kind {kind} for {key}")
        return key

    def node_entry(self, s, kind):
        p = s["props"]
        e = {"kind": kind, "name": s["attrs"]["name"]}
        if "transform" in p:
            e["transform"] = p["transform"]["v"]
        if kind in ("sphere", "plane", "triangle"):
            e["material"] = self.resource_key(p.get("material"))
            e["texture"] = self.resource_key(p.get("texture"))
        if kind == "sphere":
            e["radius"] = p.get("radius", 1.0)
        elif kind == "plane":
            e["normal"] = p["normal"]["v"] if "normal" in p else [0.0, 1.0, 0.0]
            e["distance"] = p.get("distance", 0.0)
        elif kind == "triangle":
            e["vertex1"] = p["vertex1"]["v"] if "vertex1" in p else [-1.0, 0.0, 0.0]
            e["vertex2"] = p["vertex2"]["v"] if "vertex2" in p else [0.0, 0.0, -1.0]
            e["vertex3"] = p["vertex3"]["v"] if "vertex3" in p else [0.0, 0.0, 0.0]
        elif kind == "camera":
            e["fov"] = p.get("fov", 75.0)
        elif kind == "mesh":
            e["bvh_type"] = p.get("bvh_type", 1)
            e["bvh_order"] = p.get("bvh_order", 8)
            e["default_material"] = self.resource_key(p.get("default_material"))
            e["override_material"] = self.resource_key(p.get("override_material"))
            e["children"] = []
        return e

    def extract(self, name: str, tris_out: str | None = None):
        nodes = [s for s in self.secs if s["kind"] == "node"]
        root = nodes[0]
        out = {
            "name": name,
            "source": os.path.relpath(self.path, REF),
            "bvh_order": root["props"].get("bvh_order", 8),
            "default_bvh": root["props"].get("default_bvh", 1),
            "starting_camera": root["props"].get("starting_camera", 0),
            "nodes": [],
        }
        by_path = {".": out}
        for s in nodes[1:]:
            parent = s["attrs"].get("parent", ".")
            kind = self.kind_of_script(s["props"].get("script"))
            inst = s["attrs"].get("instance")
            if inst is not None:
                sub_path = res_path(self.ext[inst["id"]]["path"])
                sub = SceneExtractor(sub_path)
                sub_nodes = [x for x in sub.secs if x["kind"] == "node"]
                sub_root = sub_nodes[0]
                if sub.kind_of_script(sub_root["props"].get("script")) != "mesh":
                    continue
                merged = {"attrs": {"name": s["attrs"]["name"]}, "props": dict(sub_root["props"])}
                merged["props"].update(s["props"])
                merged["props"].pop("script", None)
                e = self.node_entry({"attrs": merged["attrs"], "props": {k: v for k, v in merged["props"].items() if k not in ("default_material", "override_material")}}, "mesh")
                e["default_material"] = self.resource_key(s["props"].get("default_material"))
                e["override_material"] = self.resource_key(s["props"].get("override_material"))
                # imported geometry: the MeshInstance3D child that carries an ArrayMesh
                for x in sub_nodes[1:]:
                    mref = x["props"].get("mesh")
                    if x["attrs"].get("type") == "MeshInstance3D" and mref and mref["_t"] == "SubResource":
                        tris = decode_array_mesh(sub.sub[mref["id"]]["raw"]["_surfaces"])
                        assert tris_out is not None
                        np.save(os.path.join(OUT, tris_out), tris)
                        e["triangles_file"] = tris_out
                        e["triangle_count"] = int(tris.shape[0])
                out["nodes"].append(e)
                continue
            if kind is None or kind == "scene":
                continue
            e = self.node_entry(s, kind)
            if parent == ".":
                out["nodes"].append(e)
                if kind == "mesh":
                    by_path[s["attrs"]["name"]] = e
            else:
                by_path[parent]["children"].append(e)
        out["materials"] = self.materials
        out["textures"] = self.textures
        return out


SCENES = [
    ("sphere_scene1", "scenes/sphere_scene1.tscn", None),
    ("benchmark_random_scene", "scenes/benchmark_random_scene.tscn", None),
    ("fake_cornell_box", "scenes/fake_cornell_box.tscn", None),
    ("sphere_scene3", "scenes/sphere_scene3.tscn", None),
    ("dragon_test", "test_models/dragon low poly/dragon_test.tscn", "dragon_test_tris.npy"),
    ("test_scene_bvh_a", "tests/test_scenes/test_scene_bvh_a.tscn", None),
    ("test_scene_bvh_b", "tests/test_scenes/test_scene_bvh_b.tscn", None),
]


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, rel, tris in SCENES:
        d = SceneExtractor(os.path.join(REF, rel)).extract(name, tris)
        with open(os.path.join(OUT, name + ".json"), "w") as f:
            json.dump(d, f, indent=1)
        print(name, "nodes:", len(d["nodes"]), "materials:", len(d["materials"]), "textures:", len(d["textures"]))
    extract_mesh_surfaces()


if __name__ == "__main__":
    sys.exit(main())
