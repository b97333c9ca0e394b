"""The record-tail filter of k_trace rests on a per-leaf-entry proof made at scene upload (api.cpp: tailFilterFlag, exported as
mtsgpu_tail_filter_flag): "TriAccel::rayIntersect (triaccel.h:141-158), evaluated in binary32, rejects every projected point beyond
the leaf's box on the triangle's u or v axis".  Here the statement is attacked directly: for triangles whose flag is set, the
reference's expressions are evaluated in float32 (numpy, one rounding per operation, the reference's order) on points just beyond
each face, far beyond it, and at every magnitude in between -- none may pass the test.  CPU only."""
import ctypes as C

import numpy as np
import pytest

F = np.float32


def triaccel(A, B, Cc):
    """TriAccel::load (include/mitsuba/render/triaccel.h:63-96) in float32: returns the 12 dwords, or None (degenerate)"""
    A, B, Cc = (np.asarray(x, dtype=F) for x in (A, B, Cc))
    b, c = Cc - A, B - A
    N = np.cross(c.astype(F), b.astype(F)).astype(F)
    k = int(np.argmax(np.abs(N)))
    if N[k] == 0:
        return None
    u, v = (k + 1) % 3, (k + 2) % 3
    n_k = N[k]
    denom = F(b[u] * c[v] - b[v] * c[u])
    if denom == 0:
        return None
    n_u, n_v = F(N[u] / n_k), F(N[v] / n_k)
    n_d = F(F(F(A[u] * n_u) + F(A[v] * n_v)) + A[k])          # dot(A, N) / n_k evaluated like the reference's normalised form
    rec = np.zeros(12, dtype=np.uint32)
    f = rec.view(F)
    rec[0] = k
    f[1], f[2], f[3] = n_u, n_v, n_d
    f[4], f[5] = A[u], A[v]
    f[6], f[7] = F(b[u] / denom), F(-b[v] / denom)
    f[8], f[9] = F(c[v] / denom), F(-c[u] / denom)
    return rec


def passes(rec, pu, pv):
    """the barycentric part of TriAccel::rayIntersect on arrays of projected points, float32, -ffp-contract=off semantics"""
    f = rec.view(F)
    a_u, a_v, b_nu, b_nv, c_nu, c_nv = f[4], f[5], f[6], f[7], f[8], f[9]
    with np.errstate(all="ignore"):
        hu = (pu - a_u).astype(F); hv = (pv - a_v).astype(F)
        u = ((hv * b_nu).astype(F) + (hu * b_nv).astype(F)).astype(F)
        v = ((hu * c_nu).astype(F) + (hv * c_nv).astype(F)).astype(F)
        return (u >= 0) & (v >= 0) & ((u + v).astype(F) <= F(1.0))


def beyond_points(rng, lo, hi, n, margin):
    """float32 points beyond [lo, hi] by more than `margin` on the first coordinate, as the kernel decides it -- fl(p - hi) > margin
    above, fl(lo - p) > margin below --, the second coordinate anywhere; the closest such values are always among them"""
    out = []
    for bound, sign in ((hi, 1.0), (lo, -1.0)):
        b = F(bound)
        with np.errstate(all="ignore"):
            start = F(b + F(sign) * F(margin))
            near = [start]
            for _ in range(8):                               # the first few binary32 values around the threshold
                near.append(np.nextafter(near[-1], F(sign * np.inf), dtype=F))
            near = np.array(near, dtype=F)
            steps = (b + sign * (F(margin) + np.abs(b + F(1e-30)) * (F(2.0) ** rng.uniform(-23, 40, n)).astype(F))).astype(F)
            absd = (b + sign * (F(margin) + (F(10.0) ** rng.uniform(-12, 30, n)).astype(F))).astype(F)
            first = np.concatenate([near, steps, absd]).astype(F)
            dist = ((first - b) if sign > 0 else (b - first)).astype(F)
        first = first[np.isfinite(first) & (dist > F(margin))]
        second = np.concatenate([rng.uniform(-3, 3, len(first) // 2), (10.0 ** rng.uniform(-6, 30, len(first) - len(first) // 2)) * rng.choice([-1, 1], len(first) - len(first) // 2)]).astype(F)
        rng.shuffle(second)
        out.append((first, second))
    return out


def test_flagged_entries_reject_everything_beyond_their_box(mts):
    L = mts.lib()
    rng = np.random.RandomState(7)
    n_flag = n_total = 0
    for trial in range(4000):
        scale = 10.0 ** rng.uniform(-3, 3)
        centre = rng.uniform(-2, 2, 3) * scale
        shape = rng.choice(["fat", "sliver", "tiny"])
        T = rng.uniform(-1, 1, (3, 3)) * scale * (1e-3 if shape == "tiny" else 1.0)
        if shape == "sliver":
            T[2] = T[0] + (T[1] - T[0]) * rng.uniform(0, 1) + rng.uniform(-1, 1, 3) * scale * 1e-5
        A, B, Cc = (centre + T).astype(F)
        rec = triaccel(A, B, Cc)
        if rec is None:
            continue
        tri = np.stack([A, B, Cc])
        pad = (10.0 ** rng.uniform(-7, 0)) * scale
        lo = (tri.min(0) - F(pad) * rng.uniform(0, 1, 3)).astype(F)
        hi = (tri.max(0) + F(pad) * rng.uniform(0, 1, 3)).astype(F)
        if trial % 2:                                        # what the SAH builder produces: the box IS the triangle's bounds
            lo, hi = tri.min(0).astype(F), tri.max(0).astype(F)
            hi = np.where(hi > lo, hi, np.nextafter(hi, F(np.inf), dtype=F))
        if trial % 7 == 0:                                   # boxes that cut the triangle: the flag must not be set
            ax = rng.randint(3); hi[ax] = F(0.5) * (tri[:, ax].min() + tri[:, ax].max())
        margin = F(np.abs(np.concatenate([lo, hi])).max() * 2.0 ** -16) if trial % 3 else F(0.0)
        flag = L.mtsgpu_tail_filter_flag(rec.ctypes.data_as(C.POINTER(C.c_uint32)), lo.ctypes.data_as(C.POINTER(C.c_float)), hi.ctypes.data_as(C.POINTER(C.c_float)), margin)
        n_total += 1
        k = int(rec[0]); ku, kv = (k + 1) % 3, (k + 2) % 3
        if trial % 7 == 0 and ax != k:
            # a face through the triangle's projection: some point beyond it lies inside the triangle, so no proof can exist
            # (a cut along the triangle's own k axis says nothing about (u, v): the kernel does not filter on that axis)
            assert flag == 0 or hi[ax] >= tri[:, ax].max()
        if not flag:
            continue
        n_flag += 1
        for first, second in beyond_points(rng, lo[ku], hi[ku], 300, margin):
            assert not passes(rec, first, second).any(), (trial, "u axis")
        for first, second in beyond_points(rng, lo[kv], hi[kv], 300, margin):
            assert not passes(rec, second, first).any(), (trial, "v axis")
    # the proof is not vacuous: most triangles that fit their box get the flag
    assert n_flag > 0.5 * n_total, (n_flag, n_total)


def test_flag_refuses_what_it_cannot_prove(mts):
    L = mts.lib()
    def flag(rec, lo, hi, margin=0.0):
        lo, hi = np.asarray(lo, dtype=F), np.asarray(hi, dtype=F)
        return L.mtsgpu_tail_filter_flag(rec.ctypes.data_as(C.POINTER(C.c_uint32)), lo.ctypes.data_as(C.POINTER(C.c_float)), hi.ctypes.data_as(C.POINTER(C.c_float)), F(margin))
    rec = triaccel((0, 0, 0), (1, 0, 0), (0, 1, 0))
    assert flag(rec, (-1, -1, -1), (2, 2, 2)) == 1
    assert flag(rec, (-1, -1, -1), (0.5, 2, 2)) == 0          # the triangle reaches beyond x = 0.5
    assert flag(rec, (0, -1, -1), (2, 2, 2)) == 0             # vertex A on the face: hu can be 0 beyond it
    assert flag(rec, (-1, -1, -1), (1, 2, 2)) == 0            # vertex B on the face
    # ... which is what the builder's planes do to most triangles: with the kernel's margin the touched faces are provable
    assert flag(rec, (0, 0, -1), (1, 1, 2)) == 0 and flag(rec, (0, 0, -1), (1, 1, 2), 2.0 ** -16) == 1
    assert flag(rec, (-1, -1, -1), (0.5, 2, 2), 2.0 ** -16) == 0
    assert flag(rec, (-1, -1, -1), (2, 2, 2), np.nan) == 0 and flag(rec, (-1, -1, -1), (2, 2, 2), -1.0) == 0
    shape = rec.copy(); shape[0] = 3                          # a non-triangle primitive / degenerate triangle
    assert flag(shape, (-1, -1, -1), (2, 2, 2)) == 0
    nan = rec.copy(); nan.view(F)[6] = np.nan
    assert flag(nan, (-1, -1, -1), (2, 2, 2)) == 0
