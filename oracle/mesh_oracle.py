"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the plugin's mesh import for checking ptk_mesh_to_triangles.

Follows PTObjectContainer.mesh_to_pttriangles (proprietary_dataclasses/pt_object_container.gd:163-186),
PTTriangle.set_aabb / to_byte_array (public/primitives/pt_triangle.gd:107-112,125-150) and
PTObject.get_bvh_aabb (pt_object.gd:269-272) over a surface decoded the way Godot 4.3's
RenderingServer.mesh_surface_get_arrays decodes it.  Godot's source is not part of the reference checkout:
the decode rule (unorm16 positions relative to the surface aabb under ARRAY_FLAG_COMPRESS_ATTRIBUTES, 16-bit
indices up to 65536 vertices, attribute block = [colour][uv][uv2]) is restated from its published format and
is pinned only by plausibility of the decoded meshes (index counts that match the scene files, closed surfaces):
parity unpinned against Godot itself.  Imported by tests/ only."""
import numpy as np

F_COLOR, F_UV, F_UV2, F_INDEX, F_COMPRESS = 1 << 3, 1 << 4, 1 << 5, 1 << 12, 1 << 29


def surface_arrays(s):
    """(vertices [V,3] f32, uvs [V,2] f32 or None, indices [I] int64) of a raw surface dictionary."""
    fmt, vcount, icount = int(s["format"]), int(s["vertex_count"]), int(s["index_count"])
    compressed = bool(fmt & F_COMPRESS)
    vdata = np.asarray(s["vertex_data"], np.uint8).tobytes()
    if compressed:
        u = np.frombuffer(vdata, "<u2", vcount * 4).reshape(vcount, 4)[:, :3]
        vec = (u.astype(np.float32).astype(np.float64) / 65535.0).astype(np.float32)
        aabb = np.asarray(s["aabb"], np.float32)
        verts = (vec * aabb[3:] + aabb[:3]).astype(np.float32)
    else:
        verts = np.frombuffer(vdata, "<f4", vcount * 3).reshape(vcount, 3).copy()
    uvs = None
    if fmt & F_UV:
        stride = (4 if fmt & F_COLOR else 0) + (4 if compressed else 8) * (1 + (1 if fmt & F_UV2 else 0))
        off = 4 if fmt & F_COLOR else 0
        a = np.asarray(s["attribute_data"], np.uint8)[: vcount * stride].reshape(vcount, stride)
        if compressed:
            q = np.ascontiguousarray(a[:, off:off + 4]).view("<u2").reshape(vcount, 2)
            uvs = (q.astype(np.float32).astype(np.float64) / 65535.0).astype(np.float32)
            sc = np.asarray(s["uv_scale"], np.float32)
            if np.any(sc != 0):
                uvs = ((uvs - np.float32(0.5)) * sc[:2]).astype(np.float32)
        else:
            uvs = np.ascontiguousarray(a[:, off:off + 8]).view("<f4").reshape(vcount, 2).copy()
    idx = np.frombuffer(np.asarray(s["index_data"], np.uint8).tobytes(), "<u2" if vcount <= 65536 else "<u4", icount).astype(np.int64)
    return verts, uvs, idx


def triangle_uvs(s):
    """[N,6] float32: (u,v) of the three corners per triangle (zeros without a UV array)."""
    verts, uvs, idx = surface_arrays(s)
    n = idx.shape[0] // 3
    if uvs is None:
        return np.zeros((n, 6), np.float32)
    return uvs[idx].reshape(n, 6).astype(np.float32)


def triangle_vertices(s):
    verts, _, idx = surface_arrays(s)
    return verts[idx].reshape(-1, 3, 3).astype(np.float32)
