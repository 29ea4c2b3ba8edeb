"""Deterministic synthetic scene: a subdivided icosphere inside the Cornell box.

Closed-form, no random numbers: the standard 12-vertex icosahedron is subdivided `levels` times (each triangle
(a, b, c) -> (a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca), midpoints re-normalised in double precision),
scaled to radius 150 and centred at (278, 200, 279.6) inside the original Cornell box, and appended to the box's
34 triangles with the box's diffuse `floor` material.  levels = 8 gives 20 * 4**8 = 1,310,720 sphere triangles
(1,310,754 in total), the "1M-triangle subdivided mesh" workload.
"""
import numpy as np

from .scene import Scene

CENTER = np.array([278.0, 200.0, 279.6])
RADIUS = 150.0
FLOOR_MATERIAL = 2  # index of `floor` in the Cornell-Original material table


def icosphere(levels):
    """Returns (vertices float64 [V,3] on the unit sphere, faces int64 [F,3]) after `levels` subdivisions."""
    phi = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([(-1, phi, 0), (1, phi, 0), (-1, -phi, 0), (1, -phi, 0), (0, -1, phi), (0, 1, phi), (0, -1, -phi), (0, 1, -phi),
                  (phi, 0, -1), (phi, 0, 1), (-phi, 0, -1), (-phi, 0, 1)], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
                  (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)],
                 dtype=np.int64)
    for _ in range(levels):
        a, b, c = f[:, 0], f[:, 1], f[:, 2]
        edges = np.concatenate([np.stack([a, b], 1), np.stack([b, c], 1), np.stack([c, a], 1)])
        key = np.sort(edges, axis=1)
        uniq, inverse = np.unique(key[:, 0] * (len(v) + 1) + key[:, 1], return_inverse=True)
        lo, hi = uniq // (len(v) + 1), uniq % (len(v) + 1)
        mid = v[lo] + v[hi]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        base = len(v)
        v = np.concatenate([v, mid])
        n = len(f)
        ab, bc, ca = base + inverse[:n], base + inverse[n:2 * n], base + inverse[2 * n:]
        f = np.stack([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1), np.stack([c, ca, bc], 1), np.stack([ab, bc, ca], 1)],
                     axis=1).reshape(-1, 3)
    return v, f


def cornell_with_icosphere(box, levels=8):
    """box: the Cornell-Original Scene.  Returns a new Scene = box triangles + the icosphere (flat shaded)."""
    v, f = icosphere(levels)
    p = (CENTER + RADIUS * v).astype(np.float32)
    sphere = p[f]  # [F, 3, 3]
    pos = np.concatenate([box.pos, sphere])
    matid = np.concatenate([box.matid, np.full(len(sphere), FLOOR_MATERIAL, np.int32)])
    return Scene(pos, None, matid, box.materials, has_normals=False)
