"""Godot 4.3 float32 (real_t) value-type arithmetic used on the host side.

The reference's GDScript computes object AABBs, transformed triangle points and
camera bases with Godot's single-precision Vector3 / AABB / Transform3D.  The
host mirror needs the same numbers, so the few operations it uses are restated
here with numpy.float32 scalars, one rounding per operation (Godot Engine 4.3:
core/math/vector3.h, aabb.h, basis.cpp, transform_3d.cpp).  Godot's sources are
not part of the reference checkout; results are pinned only by the AABB vectors
of the reference's tests/unit_tests/test_utils.gd.
"""
from __future__ import annotations

import math

import numpy as np

f32 = np.float32
_ZERO = f32(0.0)


def vec3(x=0.0, y=0.0, z=0.0):
    return np.array([x, y, z], dtype=np.float32)


def dot(a, b):
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def cross(a, b):
    return np.array(
        [
            f32(f32(a[1] * b[2]) - f32(a[2] * b[1])),
            f32(f32(a[2] * b[0]) - f32(a[0] * b[2])),
            f32(f32(a[0] * b[1]) - f32(a[1] * b[0])),
        ],
        dtype=np.float32,
    )


def normalized(v):
    """Vector3::normalized: divide by sqrt(length_squared), zero stays zero."""
    lsq = dot(v, v)
    if lsq == 0:
        return vec3()
    length = f32(math.sqrt(float(lsq)))  # sqrt of a float32 rounded back to float32 == sqrtf
    return np.array([f32(v[0] / length), f32(v[1] / length), f32(v[2] / length)], dtype=np.float32)


class Transform3D:
    """basis rows (3x3, row-major) + origin, as Godot stores and serialises it."""

    __slots__ = ("rows", "origin")

    def __init__(self, rows=None, origin=None):
        self.rows = np.eye(3, dtype=np.float32) if rows is None else np.asarray(rows, dtype=np.float32).reshape(3, 3)
        self.origin = vec3() if origin is None else np.asarray(origin, dtype=np.float32).reshape(3)

    @staticmethod
    def from_tscn(vals):
        """Transform3D(xx, xy, xz, yx, yy, yz, zx, zy, zz, ox, oy, oz) as written in .tscn files."""
        vals = [float(v) for v in vals]
        return Transform3D(np.array(vals[:9], dtype=np.float32).reshape(3, 3), np.array(vals[9:12], dtype=np.float32))

    def to_list(self):
        return [float(x) for x in self.rows.reshape(-1)] + [float(x) for x in self.origin]

    # basis columns, Godot's basis.x / basis.y / basis.z
    def column(self, i):
        return self.rows[:, i].copy()

    def xform(self, v):
        """Transform3D * Vector3: basis[row].dot(v) + origin[row]."""
        return np.array([f32(dot(self.rows[i], v) + self.origin[i]) for i in range(3)], dtype=np.float32)

    def basis_xform(self, v):
        return np.array([dot(self.rows[i], v) for i in range(3)], dtype=np.float32)

    def mul(self, other: "Transform3D") -> "Transform3D":
        """self * other (Transform3D::operator*): origin = xform(other.origin), basis = basis * other.basis."""
        origin = self.xform(other.origin)
        rows = np.zeros((3, 3), dtype=np.float32)
        for i in range(3):
            for j in range(3):
                rows[i, j] = dot(self.rows[i], other.rows[:, j])
        return Transform3D(rows, origin)

    def affine_inverse(self) -> "Transform3D":
        """Transform3D::affine_inverse: basis.invert() by cofactors, origin = basis.xform(-origin)."""
        m = self.rows

        def cofac(r1, c1, r2, c2):
            return f32(f32(m[r1, c1] * m[r2, c2]) - f32(m[r1, c2] * m[r2, c1]))

        co = [cofac(1, 1, 2, 2), cofac(1, 2, 2, 0), cofac(1, 0, 2, 1)]
        det = f32(f32(f32(m[0, 0] * co[0]) + f32(m[0, 1] * co[1])) + f32(m[0, 2] * co[2]))
        s = f32(f32(1.0) / det)
        inv = np.array(
            [
                [f32(co[0] * s), f32(cofac(0, 2, 2, 1) * s), f32(cofac(0, 1, 1, 2) * s)],
                [f32(co[1] * s), f32(cofac(0, 0, 2, 2) * s), f32(cofac(0, 2, 1, 0) * s)],
                [f32(co[2] * s), f32(cofac(0, 1, 2, 0) * s), f32(cofac(0, 0, 1, 1) * s)],
            ],
            dtype=np.float32,
        )
        t = Transform3D(inv, vec3())
        t.origin = t.basis_xform(np.array([-self.origin[0], -self.origin[1], -self.origin[2]], dtype=np.float32))
        return t

    @staticmethod
    def looking_at(position, target, up=(0.0, 1.0, 0.0)) -> "Transform3D":
        """Node3D.look_at from `position`: Basis::looking_at(target - position, up), -Z forward."""
        position = np.asarray(position, dtype=np.float32)
        d = np.asarray(target, dtype=np.float32) - position
        v_z = normalized(d)
        v_z = np.array([-v_z[0], -v_z[1], -v_z[2]], dtype=np.float32)
        v_x = normalized(cross(np.asarray(up, dtype=np.float32), v_z))
        v_y = cross(v_z, v_x)
        rows = np.stack([v_x, v_y, v_z], axis=1).astype(np.float32)  # columns are the axes
        return Transform3D(rows, position)


class AABB:
    """Godot AABB: position + size (float32)."""

    __slots__ = ("position", "size")

    def __init__(self, position=None, size=None):
        self.position = vec3() if position is None else np.asarray(position, dtype=np.float32).reshape(3).copy()
        self.size = vec3() if size is None else np.asarray(size, dtype=np.float32).reshape(3).copy()

    def end(self):
        return self.position + self.size

    def abs(self) -> "AABB":
        # MIN(size, 0) written as Godot's macro: (a < b) ? a : b
        mn = np.where(self.size < _ZERO, self.size, _ZERO).astype(np.float32)
        return AABB(self.position + mn, np.abs(self.size))

    def expand(self, p) -> "AABB":
        """AABB::expand_to on a copy."""
        p = np.asarray(p, dtype=np.float32)
        begin = self.position.copy()
        end = self.position + self.size
        begin = np.where(p < begin, p, begin).astype(np.float32)
        end = np.where(p > end, p, end).astype(np.float32)
        return AABB(begin, end - begin)

    def to6(self):
        return np.concatenate([self.position, self.size]).astype(np.float32)


def xform_aabb(t: Transform3D, a: AABB) -> AABB:
    """Transform3D::xform(AABB)."""
    mn = a.position
    mx = a.position + a.size
    tmin = np.zeros(3, dtype=np.float32)
    tmax = np.zeros(3, dtype=np.float32)
    for i in range(3):
        lo = hi = t.origin[i]
        for j in range(3):
            e = f32(t.rows[i, j] * mn[j])
            f = f32(t.rows[i, j] * mx[j])
            if e < f:
                lo = f32(lo + e)
                hi = f32(hi + f)
            else:
                lo = f32(lo + f)
                hi = f32(hi + e)
        tmin[i] = lo
        tmax[i] = hi
    return AABB(tmin, tmax - tmin)
