"""Pins the oracle: golden vectors of the reference's own tests + published known answers.

  * Triangle::getClippedAABB      -- src/tests/test_kd.cpp:34-83 (all five cases)
  * radicalInverse / Halton / Hammersley / radicalInverseIncremental -- src/tests/test_samplers.cpp:33-87
  * Random == MT19937-64          -- 10000th output of the default seed (ISO C++ [rand.predef]),
                                     plus the values the survey measured from the reference
"""
import ctypes as C
import numpy as np
import pytest

f32p = C.POINTER(C.c_float)


def _p(a):
    return a.ctypes.data_as(f32p)


def _clip(orc, tri, bmin, bmax):
    tri = np.asarray(tri, dtype=np.float32)
    bmin = np.asarray(bmin, dtype=np.float32); bmax = np.asarray(bmax, dtype=np.float32)
    omin = np.zeros(3, dtype=np.float32); omax = np.zeros(3, dtype=np.float32)
    p0, p1, p2 = (np.ascontiguousarray(tri[i]) for i in range(3))
    valid = orc.lib().orc_clipped_aabb(_p(p0), _p(p1), _p(p2), _p(bmin), _p(bmax), _p(omin), _p(omax))
    return valid, omin, omax


UNIT_TRI = [(0, 0, 0), (1, 0, 0), (1, 1, 0)]


def test_sutherland_hodgman_kats(orc):
    # test_kd.cpp:46-53: split the triangle in half
    v, mn, mx = _clip(orc, UNIT_TRI, (0, .5, -1), (1, 1, 1))
    assert v and np.array_equal(mn, [.5, .5, 0]) and np.array_equal(mx, [1, 1, 0])
    # :55-60: completely clipped away
    v, mn, mx = _clip(orc, UNIT_TRI, (2, 2, 2), (3, 3, 3))
    assert not v
    # :62-69: no clipping when the box contains the triangle
    v, mn, mx = _clip(orc, UNIT_TRI, (-1, -1, -1), (1, 1, 1))
    assert v and np.array_equal(mn, [0, 0, 0]) and np.array_equal(mx, [1, 1, 0])
    # :71-77: a triangle within a flat cell is kept
    v, mn, mx = _clip(orc, UNIT_TRI, (-100, -100, 0), (100, 100, 0))
    assert v and np.array_equal(mn, [0, 0, 0]) and np.array_equal(mx, [1, 1, 0])
    # :79-83: touching the clip box gives a collapsed point box
    v, mn, mx = _clip(orc, UNIT_TRI, (0, 1, 0), (1, 2, 0))
    assert v and np.array_equal(mn, [1, 1, 0]) and np.array_equal(mx, [1, 1, 0])


HALTON = [
    0, 0, 0, 0, 0,
    0.500000000000000, 0.333333333333333, 0.200000000000000, 0.142857142857143, 0.090909090909091,
    0.250000000000000, 0.666666666666667, 0.400000000000000, 0.285714285714286, 0.181818181818182,
    0.750000000000000, 0.111111111111111, 0.600000000000000, 0.428571428571429, 0.272727272727273,
    0.125000000000000, 0.444444444444444, 0.800000000000000, 0.571428571428571, 0.363636363636364]
PRIMES = [2, 3, 5, 7, 11]


def test_halton_hammersley_kats(orc):
    """test_samplers.cpp:33-78 through the oracle's SAMPLERS (generate(), then next1D() per dimension, advance()):
    Halton sample i, dimension j == radicalInverse(prime_j, i); Hammersley: i / sampleCount first, then those (eps 1e-7)"""
    L = orc.lib()
    for i in range(5):
        for j in range(5):
            assert abs(L.orc_radical_inverse(PRIMES[j], i) - HALTON[i * 5 + j]) <= 1e-7
    table = np.array(HALTON).reshape(5, 5)
    halton = orc.render_params(4, sampler=2, spp=5)
    hammersley = orc.render_params(4, sampler=3, spp=5)
    for i in range(5):
        for key in (0, 12345):                      # the QMC samplers restart at every pixel (halton.cpp:58-61)
            got = orc.sampler_values(halton, key, i, 5)
            assert np.abs(got.astype(np.float64) - table[i]).max() <= 1e-7
            got = orc.sampler_values(hammersley, key, i, 6)
            assert np.abs(got.astype(np.float64) - np.concatenate([[i / 5.0], table[i]])).max() <= 1e-7
    # next2D() draws two consecutive dimensions, x first
    got = orc.sampler_values(halton, 0, 3, 2, two_d=True)
    assert np.abs(got.reshape(-1).astype(np.float64) - table[3, :4]).max() <= 1e-7


def test_radical_inverse_incremental(orc):
    """test_samplers.cpp:80-87"""
    L = orc.lib()
    x = np.float32(0.0)
    for i in range(20):
        assert x == L.orc_radical_inverse(2, i)
        x = np.float32(L.orc_radical_inverse_incremental(2, float(x)))


def test_random_is_mt19937_64(orc):
    import orc as O
    L = orc.lib()
    r = O.Random(); r.mti = 313
    L.orc_random_seed(C.byref(r), 5489)
    out = [L.orc_random_next_ulong(C.byref(r)) for _ in range(10000)]
    assert out[0] == 14514284786278117030           # SURVEY.md 8c.1, == std::mt19937_64()()
    assert out[9999] == 9981545732273789042         # ISO C++ [rand.predef]: 10000th invocation
    # default-constructed generator (mti == N+1) seeds itself with 5489 (random.cpp:149-150)
    r2 = O.Random(); r2.mti = 313
    assert L.orc_random_next_ulong(C.byref(r2)) == 14514284786278117030
    # first two nextFloat of a fresh generator
    r3 = O.Random(); r3.mti = 313
    assert float(L.orc_random_next_float(C.byref(r3))).hex() == "0x1.eded5c0000000p-1"
    assert float(L.orc_random_next_float(C.byref(r3))).hex() == "0x1.17901c0000000p-1"
    # Random(Random*) clone #0 of a fresh parent
    parent = O.Random(); parent.mti = 313
    child = O.Random(); child.mti = 313
    L.orc_random_seed_from(C.byref(child), C.byref(parent))
    assert L.orc_random_next_ulong(C.byref(child)) == 13719712115898985683
    # shuffle of [0..7] with a fresh generator
    r4 = O.Random(); r4.mti = 313
    a = np.arange(8, dtype=np.uint32)
    L.orc_random_shuffle_u32(C.byref(r4), a.ctypes.data_as(C.POINTER(C.c_uint32)), 8)
    assert a.tolist() == [7, 3, 1, 5, 2, 0, 4, 6]


def test_next_size_rejection(orc):
    import orc as O
    L = orc.lib()
    for n in (1, 2, 3, 1000, 2 ** 32 + 1):
        r = O.Random(); r.mti = 313
        vals = [L.orc_random_next_size(C.byref(r), n) for _ in range(200)]
        assert all(0 <= v < n for v in vals)
        if n > 2:
            assert len(set(vals)) > 1


def test_vdc_sobol_integer_code(orc):
    """ldsampler.cpp:104-118 against an independent bit-by-bit definition"""
    L = orc.lib()
    rng = np.random.RandomState(1)
    for n in list(range(64)) + rng.randint(0, 2 ** 31, 64).tolist():
        s = int(rng.randint(0, 2 ** 31)) * 2 + 1
        rev = int("{:032b}".format(n)[::-1], 2)
        assert L.orc_vdc_bits(n, s) == rev ^ s
        # Sobol' dimension 2: direction numbers v_k = v_{k-1} ^ (v_{k-1} >> 1), v_0 = 2^31
        v, acc, m = 1 << 31, s, n
        while m:
            if m & 1:
                acc ^= v
            m >>= 1
            v ^= v >> 1
        assert L.orc_sobol2_bits(n, s) == acc
    assert L.orc_u32_to_unit(0xFFFFFF80) == 1.0          # can be exactly 1
    assert L.orc_u32_to_unit(0x80000000) == 0.5


def test_ld_tables_are_a_02_sequence(orc):
    """generate2D: every elementary interval of a (0,2)-net holds exactly one point; shuffle keeps the set"""
    import orc as O
    L = orc.lib()
    spp, depth = 64, 3
    r = O.Random(); r.mti = 313
    t1 = np.zeros((depth, spp), dtype=np.float32); t2 = np.zeros((depth, spp, 2), dtype=np.float32)
    L.orc_ld_generate_mt(C.byref(r), spp, depth, _p(t1), _p(t2))
    for i in range(depth):
        for a, b in ((8, 8), (64, 1), (1, 64), (16, 4)):
            cells = (np.floor(t2[i, :, 0] * a).astype(int) * b + np.floor(t2[i, :, 1] * b).astype(int))
            assert len(np.unique(cells)) == spp
        assert len(np.unique(np.floor(t1[i] * spp).astype(int))) == spp
    k1 = np.zeros((depth, spp), dtype=np.float32); k2 = np.zeros((depth, spp, 2), dtype=np.float32)
    L.orc_ld_generate_keyed(7, 42, spp, depth, _p(k1), _p(k2))
    for i in range(depth):
        cells = (np.floor(k2[i, :, 0] * 8).astype(int) * 8 + np.floor(k2[i, :, 1] * 8).astype(int))
        assert len(np.unique(cells)) == spp


def test_deterministic_math_is_faithful(orc):
    """the binary64-evaluated elementary functions are within 1 ulp of libm (they replace std::sin etc.)"""
    L = orc.lib()
    rng = np.random.RandomState(5)

    def check(fn, ref, xs, ulps=1):
        for x in xs:
            x = np.float32(x)
            got = np.float32(fn(float(x)))
            exp = np.float32(ref(np.float64(x)))
            if np.isinf(exp) or exp == 0:
                assert got == exp or abs(got) < 1e-44
                continue
            assert abs(np.float64(got) - np.float64(exp)) <= ulps * np.spacing(np.abs(exp)), (fn, x, got, exp)

    ang = np.concatenate([rng.rand(2000) * 2 * np.pi, [0, np.pi / 2, np.pi, 2 * np.pi, 1e-8]]).astype(np.float32)
    check(L.orc_sinf, np.sin, [a for a in ang if abs(np.sin(np.float64(a))) > 1e-3])
    check(L.orc_cosf, np.cos, [a for a in ang if abs(np.cos(np.float64(a))) > 1e-3])
    check(L.orc_expf, np.exp, np.concatenate([-rng.rand(2000) * 100, rng.rand(100) * 80, [0, -104.5, -0.0]]))
    check(L.orc_logf, np.log, np.concatenate([rng.rand(2000), rng.rand(100) * 1e6 + 1e-3, [1.0]]))
    check(L.orc_atanf, np.arctan, np.concatenate([rng.rand(2000) * 3, rng.rand(200) * 1e4, [0, 1, 0.41421357]]))
    check(L.orc_pow4f, lambda x: x ** 4, rng.rand(2000))
    assert L.orc_logf(0.0) == -np.inf and np.isnan(L.orc_logf(-1.0))
    assert np.float32(L.orc_atanf(float("inf"))) == np.float32(np.pi / 2)
