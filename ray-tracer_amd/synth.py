"""Synthetic benchmark geometry.

The reference's headline scene, data/dragon.obj, is not part of its checkout, so the benchmark uses a
fixed-seed stand-in of the same class: an icosphere subdivided k times (20 * 4^k triangles) whose vertices
are pushed in and out along the normal by a sum of fixed pseudo-random sinusoids, then centred and scaled
to the unit sphere the way the reference normalises its meshes (reference framework/src/mesh.cpp:151-176).
k = 6 gives 81,920 triangles, k = 8 gives 1,310,720.
"""
import numpy as np


def _icosahedron():
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
                  [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    return v / np.linalg.norm(v, axis=1, keepdims=True), f


def _subdivide(v, f):
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=0)
    e.sort(axis=1)
    key = e[:, 0] * (len(v) + 1) + e[:, 1]
    uniq, inv = np.unique(key, return_inverse=True)
    a, b = uniq // (len(v) + 1), uniq % (len(v) + 1)
    mid = v[a] + v[b]
    mid /= np.linalg.norm(mid, axis=1, keepdims=True)
    nf = len(f)
    m01, m12, m20 = inv[:nf] + len(v), inv[nf:2 * nf] + len(v), inv[2 * nf:] + len(v)
    v2 = np.concatenate([v, mid], axis=0)
    f2 = np.concatenate([np.stack([f[:, 0], m01, m20], 1), np.stack([f[:, 1], m12, m01], 1),
                         np.stack([f[:, 2], m20, m12], 1), np.stack([m01, m12, m20], 1)], axis=0)
    return v2, f2


def dragon_class_mesh(level=6, seed=7):
    """Returns a mesh dict (vertices [n,8] f32: position, normal, uv; triangles [m,3] u32; material)."""
    v, f = _icosahedron()
    for _ in range(level):
        v, f = _subdivide(v, f)
    rng = np.random.RandomState(seed)
    disp = np.zeros(len(v))
    for octave in range(5):
        freq = 2.0 * (2.0 ** octave)
        for _ in range(6):
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            disp += (0.22 / (2.0 ** octave)) * np.sin(freq * (v @ d) + rng.uniform(0, 2 * np.pi))
    p = v * (1.0 + 0.35 * disp)[:, None]
    # smooth vertex normals from area-weighted face normals
    fn = np.cross(p[f[:, 1]] - p[f[:, 0]], p[f[:, 2]] - p[f[:, 0]])
    n = np.zeros_like(p)
    for c in range(3):
        np.add.at(n, f[:, c], fn)
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-30)
    p32 = p.astype(np.float32)
    center = p32.sum(axis=0, dtype=np.float32) / np.float32(len(p32))
    p32 = p32 - center
    p32 = p32 / np.float32(np.sqrt((p32.astype(np.float32) ** 2).sum(axis=1)).max())
    verts = np.zeros((len(p32), 8), np.float32)
    verts[:, 0:3] = p32
    verts[:, 3:6] = n.astype(np.float32)
    return dict(vertices=verts, triangles=f.astype(np.uint32), kd=(0.8, 0.8, 0.8), ks=(0.5, 0.5, 0.5), shininess=250.0,
                transparency=1.0, texture=-1)


def area_light():
    """The Cornell parallelogram light (reference src/scene.cpp:45-53) lifted above the unit sphere, so it lies
    outside the root box and shadows are not suppressed by the reference's root-box gate."""
    row = np.zeros((1, 22), np.float32)
    row[0, 0] = 2
    row[0, 1:] = [-0.2, 1.2, -0.2, 0.4, 0, 0, 0, 0, 0.4, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 1, 1]
    return row
