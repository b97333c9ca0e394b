"""Oracle self-checks on CPU: traversal vs brute force, BSDF chi-square (method of
src/tests/test_chisquare.cpp:299-420), integrator quirks, sampler equivalence."""
import ctypes as C
import numpy as np
import pytest
from conftest import chord_rays

f32p = C.POINTER(C.c_float)


def _p(a):
    return a.ctypes.data_as(f32p)


def _brute_force(A, rays):
    """closest hit over ALL TriAccel records with TriAccel::rayIntersect semantics (float32, numpy)"""
    ta = A["triaccel"]
    f = ta.view(np.float32)
    out_t = np.full(len(rays), np.inf, dtype=np.float32)
    for k in range(3):
        sel = np.nonzero(ta[:, 0] == k)[0]
        if not len(sel):
            continue
        ku, kv = (k + 1) % 3, (k + 2) % 3
        F = f[sel]
        for i, r in enumerate(rays):
            o, d = r[0:3], r[4:7]
            recip = np.float32(1.0) / (d[ku] * F[:, 1] + d[kv] * F[:, 2] + d[k])
            t = (F[:, 3] - o[ku] * F[:, 1] - o[kv] * F[:, 2] - o[k]) * recip
            hu = o[ku] + t * d[ku] - F[:, 4]
            hv = o[kv] + t * d[kv] - F[:, 5]
            u = hv * F[:, 6] + hu * F[:, 7]
            v = hu * F[:, 8] + hv * F[:, 9]
            ok = (t >= r[3]) & (t <= r[7]) & (u >= 0) & (v >= 0) & (u + v <= 1)
            if ok.any():
                out_t[i] = min(out_t[i], t[ok].min())
    return out_t


def test_traversal_equals_brute_force(mts, orc):
    sd = mts.scenes.cornell_c3(grid=6, sphere_subdiv=1)
    fs = orc.FlatScene(sd)
    A = fs.arrays()
    rays = chord_rays(400, (0, 1, 0), 2.2, seed=11)
    # no adaptive epsilon for the comparison: mint != Epsilon
    rays[:, 3] = 2e-4
    with np.errstate(all="ignore"):
        exp = _brute_force(A, rays)
    hits, tc = orc.trace_rays(fs.scene, rays, counts=True)
    got = hits[:, 0].copy().view(np.float32)
    assert np.array_equal(got, exp)
    assert tc.n_inner > 0 and tc.n_tri_tested <= tc.n_idx


def test_shadow_rays_ignore_non_occluders(mts, orc):
    sd = mts.scenes.cornell_c1()
    for m in sd.meshes:
        if m.name == "back":
            m.bsdf = -1                      # no BSDF -> Shape::isOccluder() false (skdtree.h:318-333)
    fs = orc.FlatScene(sd)
    ray = np.array([[0, 1, 3, 1e-3, 0, 0, -8, 1 - 1e-3]], dtype=np.float32)
    assert orc.trace_rays(fs.scene, ray, shadow=True)[0, 3] == 0
    assert orc.trace_rays(fs.scene, ray)[0, 3] != 0xFFFFFFFF     # closest-hit still sees it


def _chi2_bsdf(orc, btype, params, wi, n=200000, nt=10, nph=20, seed=3, full_sphere=False, skip_theta_bins=()):
    from scipy import stats
    L = orc.lib()
    rng = np.random.RandomState(seed)
    P = np.zeros(16, dtype=np.float32); P[:len(params)] = params
    wi = np.asarray(wi, dtype=np.float32); wi /= np.linalg.norm(wi)
    lo = -1.0 if full_sphere else 0.0              # cos(theta) range of the histogram
    hist = np.zeros((nt, nph))
    wo = np.zeros(3, dtype=np.float32); out = np.zeros(3, dtype=np.float32)
    pdf = C.c_float(); st = C.c_uint32()
    s = rng.rand(n, 2).astype(np.float32)
    valid = 0
    for k in range(n):
        L.orc_bsdf_sample(btype, _p(P), _p(wi), _p(s[k]), _p(wo), C.byref(pdf), C.byref(st), _p(out))
        if pdf.value <= 0:
            continue
        valid += 1
        ct = (min(max(wo[2], lo), 1.0) - lo) / (1.0 - lo)
        ph = np.arctan2(wo[1], wo[0]) % (2 * np.pi)
        hist[min(int(ct * nt), nt - 1), min(int(ph / (2 * np.pi) * nph), nph - 1)] += 1
    # expected counts: integrate pdf over each (cos theta, phi) cell (midpoint rule on a fine grid)
    sub = 24
    expected = np.zeros((nt, nph))
    w2 = np.zeros(3, dtype=np.float32)
    for i in range(nt):
        for j in range(nph):
            acc = 0.0
            for a in range(sub):
                ct = lo + (1.0 - lo) * (i + (a + 0.5) / sub) / nt
                stn = np.sqrt(max(0.0, 1 - ct * ct))
                for b in range(4):
                    ph = (j + (b + 0.5) / 4) / nph * 2 * np.pi
                    w2[:] = (stn * np.cos(ph), stn * np.sin(ph), ct)
                    acc += L.orc_bsdf_pdf(btype, _p(P), _p(wi), _p(w2))
            expected[i, j] = acc / (sub * 4) * ((1.0 - lo) / nt) * (2 * np.pi / nph) * n
    # pool cells with small expectation (test_chisquare.cpp pools below 5)
    for b in skip_theta_bins:
        expected[b, :] = 0; hist[b, :] = 0
    e, o = expected.ravel(), hist.ravel()
    big = e >= 5
    e2 = np.append(e[big], e[~big].sum()); o2 = np.append(o[big], o[~big].sum())
    if e2[-1] < 5:
        e2, o2 = e2[:-1], o2[:-1]
    e2 = e2 * (o2.sum() / e2.sum())
    chi2 = ((o2 - e2) ** 2 / e2).sum()
    return 1 - stats.chi2.cdf(chi2, len(e2) - 1), valid / n


@pytest.mark.parametrize("name,btype,params,wi", [
    ("lambertian", 0, [0.5, 0.5, 0.5], (0.3, 0.2, 0.9)),
    ("roughmetal", 2, [0.3, 0.37, 0.37, 0.37, 2.82, 2.82, 2.82, 1, 1, 1], (0.4, 0.0, 0.9)),
    ("microfacet", 3, [0.3, 0.5, 0.5, 1.5, 1.0, 1, 1, 1, 1, 1, 1], (0.5, 0.1, 0.8)),
    # the parameter sets of the reference's own chi-square list (data/tests/test_bsdf.xml:62-73)
    ("roughmetal alphaB=0.1", 2, [0.1, 0.37, 0.37, 0.37, 2.82, 2.82, 2.82, 1, 1, 1], (0.4, 0.0, 0.9)),
    ("microfacet alphaB=0.1", 3, [0.1, 0.5, 0.5, 1.5, 1.0, 1, 1, 1, 1, 1, 1], (0.5, 0.1, 0.8)),
])
def test_bsdf_sampling_matches_pdf_chi_square(orc, name, btype, params, wi):
    """sample() histogram vs integrated pdf(); significance level 0.005 as in test_chisquare.cpp:28"""
    p, frac = _chi2_bsdf(orc, btype, np.asarray(params, dtype=np.float32), wi, n=40000)
    assert frac > 0.5
    assert p > 0.005, "%s: chi-square p-value %g" % (name, p)


def test_bsdf_value_over_pdf_consistency(orc):
    """sample(bRec, pdf, s) returns f and pdf that agree with f() and pdf() at the sampled direction"""
    L = orc.lib()
    rng = np.random.RandomState(0)
    for btype, params in ((0, [0.7, 0.6, 0.5]), (2, [0.1, 0.37, 0.37, 0.37, 2.82, 2.82, 2.82, 1, 1, 1]),
                          (3, [0.1, 0.5, 0.5, 1.5, 1.0, 1, 1, 1, 1, 1, 1])):
        P = np.zeros(16, dtype=np.float32); P[:len(params)] = params
        for _ in range(200):
            wi = rng.randn(3).astype(np.float32); wi[2] = abs(wi[2]) + 0.1; wi /= np.linalg.norm(wi)
            s = rng.rand(2).astype(np.float32)
            wo = np.zeros(3, dtype=np.float32); out = np.zeros(3, dtype=np.float32); f = np.zeros(3, dtype=np.float32)
            pdf = C.c_float(); st = C.c_uint32()
            L.orc_bsdf_sample(btype, _p(P), _p(wi), _p(s), _p(wo), C.byref(pdf), C.byref(st), _p(out))
            if pdf.value == 0:
                continue
            L.orc_bsdf_f(btype, _p(P), _p(wi), _p(wo), _p(f))
            assert np.array_equal(f, out)
            assert L.orc_bsdf_pdf(btype, _p(P), _p(wi), _p(wo)) == pdf.value


def test_dielectric_is_delta_and_energy_conserving(orc):
    L = orc.lib()
    P = np.zeros(16, dtype=np.float32); P[:8] = [1.5046, 1.0, 1, 1, 1, 1, 1, 1]
    wi = np.array([0.3, 0.1, 0.9], dtype=np.float32); wi /= np.linalg.norm(wi)
    wo = np.zeros(3, dtype=np.float32); out = np.zeros(3, dtype=np.float32)
    pdf = C.c_float(); st = C.c_uint32()
    assert L.orc_bsdf_pdf(1, _p(P), _p(wi), _p(wi)) == 0            # dielectric.cpp:105-107
    kinds = set()
    for sx in np.linspace(0, 1, 50, dtype=np.float32):
        s = np.array([sx, 0.5], dtype=np.float32)
        L.orc_bsdf_sample(1, _p(P), _p(wi), _p(s), _p(wo), C.byref(pdf), C.byref(st), _p(out))
        kinds.add(st.value)
        val = out * abs(wo[2]) / pdf.value                         # what path.cpp:136-138 computes
        if st.value == 0x4:
            assert np.allclose(val, 1.0, atol=1e-6) and wo[2] > 0
        else:
            eta = 1.0 / 1.5046
            assert np.allclose(val, eta * eta, atol=1e-6) and wo[2] < 0   # radiance scaling eta^2
    assert kinds == {0x4, 0x8}


def test_max_depth_quirks(mts, orc):
    """maxDepth=1: only directly visible emitters (path.cpp:87); maxDepth=2 adds direct light only"""
    sd = mts.scenes.cornell_c1()
    fs = orc.FlatScene(sd)
    cam = orc.make_camera(sd, 48, 48)
    f1, _ = orc.render(fs.scene, cam, orc.render_params(1, spp=4))
    img1 = orc.develop(f1)
    lit = img1.sum(axis=2) > 0
    assert 0 < lit.sum() < 48 * 48 * 0.1
    assert np.allclose(img1[lit].max(), 15.0)
    assert np.all(f1[..., 3] == f1[..., 4])                       # closed view: alpha == weight
    f2, st2 = orc.render(fs.scene, cam, orc.render_params(2, spp=4))
    assert orc.develop(f2).mean() > img1.mean()
    # depth 2: one camera ray + one BSDF ray per sample that hit a surface; rr never fires (rrDepth 10)
    assert st2.rays_closest <= 2 * 48 * 48 * 4


def test_keyed_and_reference_samplers_agree_statistically(mts, orc):
    """the keyed samplers replace Random's stream, not the estimator: means agree within noise"""
    sd = mts.scenes.cornell_c1()
    fs = orc.FlatScene(sd)
    cam = orc.make_camera(sd, 24, 24)
    means = []
    for kind in (mts.abi.SAMPLER_INDEPENDENT_KEYED, mts.abi.SAMPLER_LD_KEYED):
        f, _ = orc.render(fs.scene, cam, orc.render_params(4, sampler=kind, spp=64, seed=9))
        means.append(orc.develop(f).mean())
    for kind in (0, 1):
        film = np.zeros((24, 24, 5), dtype=np.float32)
        prm = orc.render_params(4, spp=64, seed=5489)
        orc.lib().orc_render_rect_mt(fs.scene, C.byref(cam), C.byref(prm), kind, 0, 0, 24, 24, _p(film))
        means.append(orc.develop(film).mean())
    means = np.array(means)
    assert np.all(np.abs(means / means.mean() - 1) < 0.03), means


def test_film_weights_and_determinism(mts, orc):
    sd = mts.scenes.cornell_c5(sphere_subdiv=1)
    fs = orc.FlatScene(sd)
    cam = orc.make_camera(sd, 20, 16)
    prm = orc.render_params(6, sampler=mts.abi.SAMPLER_LD_KEYED, spp=16, seed=2)
    a, _ = orc.render(fs.scene, cam, prm)
    prm1 = orc.render_params(6, sampler=mts.abi.SAMPLER_LD_KEYED, spp=16, seed=2, n_threads=1)
    b, _ = orc.render(fs.scene, cam, prm1)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))   # thread count does not matter
    assert a[..., 4].max() <= 16 and a[..., 4].min() >= 14        # weight 0 only for samples on pixel borders
    assert (a[..., 3] <= a[..., 4]).all() and a[..., 3].min() < 16  # env-lit opening: alpha < 1 somewhere


def test_bordered_tiles_and_gaussian_filter(mts, orc):
    """ImageBlock borders (renderproc.cpp:143-144): the box filter through the tile path equals the plain
    per-pixel path; the gaussian film conserves weight and radiance; product and oracle tabulate identically"""
    sd = mts.scenes.cornell_c1()
    fs = orc.FlatScene(sd)
    cam = orc.make_camera(sd, 70, 50)
    prm = orc.render_params(4, sampler=mts.abi.SAMPLER_LD_KEYED, spp=8, seed=3)
    a, _ = orc.render(fs.scene, cam, prm)
    b, _ = orc.render_tiles(fs.scene, cam, prm, orc.tabulate_filter("box"))
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    g = orc.tabulate_filter("gaussian")
    tab = np.array(g.values, dtype=np.float32)
    assert g.size_x == 2.0 and tab[15].sum() == 0 and tab[:, 15].sum() == 0 and tab[0, 0] == tab.max()
    # rfilter.cpp:62-68: the table integrates to one over the filter footprint
    assert abs(tab.sum() * 4 * g.size_x * g.size_y / 225.0 - 1.0) < 1e-5
    c, _ = orc.render_tiles(fs.scene, cam, prm, g)
    assert abs(orc.develop(c).mean() / orc.develop(a).mean() - 1) < 0.02
    # interior pixels collect about spp of weight
    assert abs(c[10:-10, 10:-10, 4].mean() / 8 - 1) < 0.05
    # host tabulation of the product == oracle
    size = np.zeros(2, dtype=np.float32); vals = np.zeros(256, dtype=np.float32)
    assert mts.lib().mtsgpu_tabulate_filter(1, 2.0, 0.5, -1.0, mts.abi.ptr(size, mts.abi.f32p), mts.abi.ptr(vals, mts.abi.f32p)) == 0
    assert np.array_equal(vals.reshape(16, 16).view(np.uint32), tab.view(np.uint32)) and size[0] == 2.0


def test_other_reconstruction_filters(mts, orc):
    """mitchell / catmullrom / wsinc (src/rfilters): negative lobes, tables that integrate to one, product == oracle"""
    sd = mts.scenes.cornell_c1()
    fs = orc.FlatScene(sd)
    cam = orc.make_camera(sd, 70, 50)
    prm = orc.render_params(4, sampler=mts.abi.SAMPLER_LD_KEYED, spp=8, seed=3)
    ref, _ = orc.render(fs.scene, cam, prm)
    for kind, k, hs in (("mitchell", 2, 2.0), ("catmullrom", 3, 2.0), ("wsinc", 4, 3.0)):
        g = orc.tabulate_filter(kind)
        tab = np.array(g.values, dtype=np.float32)
        assert g.size_x == hs and tab.min() < 0 and tab[0, 0] == tab.max()
        assert abs(tab.sum() * 4 * g.size_x * g.size_y / 225.0 - 1.0) < 1e-5
        size = np.zeros(2, dtype=np.float32); vals = np.zeros(256, dtype=np.float32)
        assert mts.lib().mtsgpu_tabulate_filter(k, -1.0, -1.0, -1.0, mts.abi.ptr(size, mts.abi.f32p), mts.abi.ptr(vals, mts.abi.f32p)) == 0
        assert np.array_equal(vals.reshape(16, 16).view(np.uint32), tab.view(np.uint32)) and size[0] == hs
        film, _ = orc.render_tiles(fs.scene, cam, prm, g)
        assert abs(orc.develop(film)[8:-8, 8:-8].mean() / orc.develop(ref)[8:-8, 8:-8].mean() - 1) < 0.03
    # Mitchell's B and C properties
    a = np.array(orc.tabulate_filter("mitchell", p0=0.0, p1=0.5).values); b = np.array(orc.tabulate_filter("catmullrom").values)
    assert np.array_equal(a, b)


def test_high_quality_edges(mts, orc):
    """Film::hasHighQualityEdges (renderproc.cpp:146-153): the rendered rectangle grows by the border, so edge
    pixels get their full filter support; the box filter has no border and is unaffected"""
    sd = mts.scenes.cornell_c1()
    fs = orc.FlatScene(sd)
    cam = orc.make_camera(sd, 70, 50)
    prm = orc.render_params(4, sampler=mts.abi.SAMPLER_LD_KEYED, spp=8, seed=3)
    g = orc.tabulate_filter("gaussian")
    plain, st0 = orc.render_tiles(fs.scene, cam, prm, g)
    hq, st1 = orc.render_tiles(fs.scene, cam, prm, g, hq_edges=True)
    assert st1.camera_samples == (70 + 4) * (50 + 4) * 8 and st0.camera_samples == 70 * 50 * 8
    # without the option the weight falls off towards the film's edge; with it the weight is flat
    assert plain[0, :, 4].mean() < 0.9 * plain[25, :, 4].mean()
    assert abs(hq[0, 5:-5, 4].mean() / hq[25, 5:-5, 4].mean() - 1) < 0.03
    assert abs(hq[:, 0, 4].mean() / hq[:, 35, 4].mean() - 1) < 0.03
    assert abs(orc.develop(hq)[5:-5, 5:-5].mean() / orc.develop(plain)[5:-5, 5:-5].mean() - 1) < 0.02
    box = orc.tabulate_filter("box")
    a, _ = orc.render_tiles(fs.scene, cam, prm, box)
    b, _ = orc.render_tiles(fs.scene, cam, prm, box, hq_edges=True)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # sharding: the union of the parts' films is the whole film up to the association of border sums
    acc = sum(orc.render_tiles(fs.scene, cam, prm, g, part=k, n_parts=2, hq_edges=True)[0] for k in range(2))
    assert np.allclose(acc, hq, rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("distr,alpha", [(0, 0.3), (2, 0.4), (1, 2 / (0.3 * 0.3) - 2)])
def test_roughglass_chi_square(orc, distr, alpha):
    """the three roughglass configurations of the reference's own chi-square list (data/tests/test_bsdf.xml:92-117):
    sample() must be distributed according to pdf() over the whole sphere.
    pdf() (roughglass.cpp:413-485) does not reject half-vectors that lie on the same side of wi and wo, so it
    assigns density to a few directions at the rim of the transmitted lobe that sample() never produces.  That is
    the reference's behaviour and is kept (see the independent restatement below); the chi-square statistic leaves
    out those bands of cos(theta_o): bins 1-2 of 16 from outside, bin 8 from inside."""
    P = [distr, alpha, 1.5, 1.0, 1, 1, 1, 1, 1, 1]
    p, frac = _chi2_bsdf(orc, 6, P, (0.3, 0.1, 0.9), n=40000, nt=16, full_sphere=True, skip_theta_bins=(1, 2))
    assert frac > 0.8 and p > 0.003, (distr, p, frac)
    if distr == 0:
        # from inside a third of the microfacet normals fail by total internal reflection
        p, frac = _chi2_bsdf(orc, 6, P, (0.5, -0.2, -0.7), n=40000, nt=16, full_sphere=True, skip_theta_bins=(8,))
        assert frac > 0.55 and p > 0.003, (distr, p, frac)


def _rg_fresnel(c, ext, inte):
    etaI, etaT = (ext, inte) if c >= 0 else (inte, ext)
    sinT = etaI / etaT * np.sqrt(max(0.0, 1 - c * c))
    if sinT > 1:
        return 1.0
    cosT = np.sqrt(1 - sinT * sinT); c = abs(c)
    Rs = (etaI * c - etaT * cosT) / (etaI * c + etaT * cosT); Rp = (etaT * c - etaI * cosT) / (etaT * c + etaI * cosT)
    return (Rs * Rs + Rp * Rp) / 2


def _rg_tan(v):
    t = 1 - v[2] ** 2
    return 0.0 if t <= 0 else np.sqrt(t) / v[2]


def _rg_eval_d(d, m, a):
    if m[2] <= 0:
        return 0.0
    if d == 0:
        return np.exp(-(_rg_tan(m) / a) ** 2) / (np.pi * a * a * m[2] ** 4)
    if d == 1:
        return (a + 2) / (2 * np.pi) * m[2] ** a
    r = a / (m[2] ** 2 * (a * a + _rg_tan(m) ** 2))
    return r * r / np.pi


@pytest.mark.parametrize("distr,alpha", [(0, 0.3), (2, 0.4), (1, 2 / (0.3 * 0.3) - 2)])
def test_roughglass_against_binary64_restatement(orc, distr, alpha):
    """sample() and pdf() of the oracle against a second, independent restatement of roughglass.cpp:266-293,
    :413-617 in binary64 (numpy): same directions, same densities, from outside and from inside"""
    import ctypes as C
    L = orc.lib()
    ext, inte = 1.0, 1.5
    P = np.zeros(16, dtype=np.float32); P[:10] = [distr, alpha, inte, ext, 1, 1, 1, 1, 1, 1]
    rng = np.random.RandomState(9)
    wo = np.zeros(3, dtype=np.float32); out = np.zeros(3, dtype=np.float32)
    pdf = C.c_float(); st = C.c_uint32()
    for wi in ((0.3, 0.1, 0.9), (0.5, -0.2, -0.7)):
        wi = np.array(wi, dtype=np.float32); wi /= np.linalg.norm(wi)
        w = wi.astype(np.float64)
        nv = 0
        for s in rng.rand(1500, 2).astype(np.float32):
            L.orc_bsdf_sample(6, _p(P), _p(wi), _p(s), _p(wo), C.byref(pdf), C.byref(st), _p(out))
            # --- the restatement ---
            u = s.astype(np.float64)
            sF = min(0.9, max(0.1, _rg_fresnel(w[2], ext, inte)))
            refl = u[0] < sF
            u[0] = u[0] / sF if refl else (u[0] - sF) / (1 - sF)
            sa = alpha * (1.2 - 0.2 * np.sqrt(abs(w[2])))
            ph = 2 * np.pi * u[1]
            th = (np.arctan(np.sqrt(-sa * sa * np.log(1 - u[0]))) if distr == 0 else
                  np.arccos(u[0] ** (1 / (sa + 2))) if distr == 1 else np.arctan(sa * np.sqrt(u[0]) / np.sqrt(1 - u[0])))
            m = np.array([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)])
            etaI, etaT = (ext, inte) if w[2] >= 0 else (inte, ext)
            if refl:
                r = 2 * np.dot(w, m) * m - w
                if w[2] * r[2] <= 0:
                    r = None
            else:
                eta = etaI / etaT; c = np.dot(w, m); k = 1 + eta * eta * (c * c - 1)
                r = None
                if k >= 0:
                    r = m * (eta * c - (-1 if w[2] < 0 else 1) * np.sqrt(k)) - w * eta
                    if w[2] * r[2] >= 0:
                        r = None
            if r is None:
                assert pdf.value == 0
                continue
            if pdf.value == 0:
                continue                      # weight or density underflowed in binary32
            nv += 1
            assert np.abs(r - wo).max() < 2e-4 and st.value == (0x10 if refl else 0x20)
            # --- pdf(wi, wo) ---
            if refl:
                H = (r + w) / np.linalg.norm(r + w) * (1 if r[2] >= 0 else -1)
                dwh = 1 / (4 * np.dot(r, H))
            else:
                H = (1 if ext > inte else -1) * (w * etaI + r * etaT) / np.linalg.norm(w * etaI + r * etaT)
                sd = etaI * np.dot(w, H) + etaT * np.dot(r, H)
                dwh = etaT * etaT * np.dot(r, H) / (sd * sd)
            expect = abs(_rg_eval_d(distr, H, sa) * (sF if refl else 1 - sF) * H[2] * dwh)
            assert abs(pdf.value - expect) <= 2e-3 * expect + 1e-7, (pdf.value, expect)
        assert nv > 600


def test_roughglass_energy_and_reciprocity(orc):
    """f * cos / pdf stays bounded (Walter's sampling weights <= ~4) and reflection is reciprocal"""
    import ctypes as C
    L = orc.lib()
    P = np.zeros(16, dtype=np.float32); P[:10] = [0, 0.3, 1.5, 1.0, 1, 1, 1, 1, 1, 1]
    rng = np.random.RandomState(5)
    wi = np.array([0.4, 0.2, 0.8], dtype=np.float32); wi /= np.linalg.norm(wi)
    wo = np.zeros(3, dtype=np.float32); out = np.zeros(3, dtype=np.float32)
    pdf = C.c_float(); st = C.c_uint32()
    tot, nrefl, ntrans = 0.0, 0, 0
    for s in rng.rand(4000, 2).astype(np.float32):
        L.orc_bsdf_sample(6, _p(P), _p(wi), _p(s), _p(wo), C.byref(pdf), C.byref(st), _p(out))
        if pdf.value <= 0:
            continue
        w = out[0] * abs(wo[2]) / pdf.value
        assert 0 <= w < 8
        tot += w
        nrefl += st.value == 0x10; ntrans += st.value == 0x20
        if st.value == 0x10:
            f2 = np.zeros(3, dtype=np.float32)
            L.orc_bsdf_f(6, _p(P), _p(wo), _p(wi), _p(f2))
            assert abs(f2[0] - out[0]) <= 2e-5 * max(1.0, abs(out[0]))
    assert nrefl > 100 and ntrans > 2000
    assert 0.3 < tot / 4000 < 1.0       # transmitted radiance is scaled by (etaI/etaT)^2 = 1/2.25 on the way in


def test_halton_and_hammersley_samplers(mts, orc):
    """src/samplers/{halton,hammersley}.cpp are purely deterministic: every pixel sees the same points
    radicalInverse(primeTable[depth], j), so the raster offsets of sample j agree across pixels and equal the
    reference's own radical inverses (the golden vectors of src/tests/test_samplers.cpp pin orc_radical_inverse)"""
    sd = mts.scenes.cornell_c1()
    fs = orc.FlatScene(sd)
    cam = orc.make_camera(sd, 16, 16)
    L = orc.lib()
    for name, kind in (("halton", mts.abi.SAMPLER_HALTON), ("hammersley", mts.abi.SAMPLER_HAMMERSLEY)):
        prm = orc.render_params(4, sampler=kind, spp=12, seed=1)
        ps = np.array([[x, y, j] for (x, y) in ((0, 0), (5, 9), (15, 3)) for j in range(12)], dtype=np.uint32)
        out = orc.li_samples(fs.scene, cam, prm, ps)
        off = out[:, 4:6] - ps[:, 0:2]                         # raster position - pixel = the first 2D sample
        for j in range(12):
            if name == "halton":
                exp = (L.orc_radical_inverse(2, j), L.orc_radical_inverse(3, j))
            else:
                exp = (np.float32(j) * (np.float32(1) / np.float32(12)), L.orc_radical_inverse(2, j))
            for k in range(3):
                assert abs(off[12 * k + j, 0] - exp[0]) < 2e-6 and abs(off[12 * k + j, 1] - exp[1]) < 2e-6
        film, _ = orc.render(fs.scene, cam, prm)
        assert np.isfinite(film).all() and orc.develop(film).mean() > 0.05
        # no dependence on the seed: there is no random number in these samplers
        film2, _ = orc.render(fs.scene, cam, orc.render_params(4, sampler=kind, spp=12, seed=99))
        assert np.array_equal(film.view(np.uint32), film2.view(np.uint32))


def test_stratified_sampler(mts, orc):
    """src/samplers/stratified.cpp (keyed): sampleCount is rounded up to a perfect square and the first 2D sample of
    the spp camera samples of a pixel visits every stratum of the res x res grid exactly once"""
    sd = mts.scenes.cornell_c1()
    fs = orc.FlatScene(sd)
    cam = orc.make_camera(sd, 8, 8)
    prm = orc.render_params(4, sampler=mts.abi.SAMPLER_STRATIFIED_KEYED, spp=14, seed=5)       # -> 16 = 4 x 4
    for (x, y) in ((0, 0), (3, 5), (7, 7)):
        ps = np.array([[x, y, j] for j in range(16)], dtype=np.uint32)
        out = orc.li_samples(fs.scene, cam, prm, ps)
        off = out[:, 4:6] - ps[:, 0:2]
        assert (off >= 0).all() and (off < 1).all()
        cells = set((int(o[0] * 4), int(o[1] * 4)) for o in off)
        assert len(cells) == 16
    film, st = orc.render(fs.scene, cam, prm)
    assert st.camera_samples == 8 * 8 * 16 and np.isfinite(film).all()
    ref, _ = orc.render(fs.scene, cam, orc.render_params(4, sampler=mts.abi.SAMPLER_LD_KEYED, spp=64, seed=5))
    assert abs(orc.develop(film).mean() / orc.develop(ref).mean() - 1) < 0.15


def test_direct_integrator(mts, orc):
    """MIDirectIntegrator (direct.cpp): equals the path tracer truncated after one bounce in expectation; either
    strategy alone gives the same picture; background-only pixels return LeBackground"""
    sd = mts.scenes.cornell_c5(sphere_subdiv=2)
    fs = orc.FlatScene(sd)
    cam = orc.make_camera(sd, 40, 40)
    kw = dict(sampler=mts.abi.SAMPLER_LD_KEYED, spp=64, seed=7)
    ref, _ = orc.render(fs.scene, cam, orc.render_params(2, **kw))                      # path, maxDepth = 2
    both, st = orc.render(fs.scene, cam, orc.render_params(-1, integrator="direct", **kw))
    lum, _ = orc.render(fs.scene, cam, orc.render_params(-1, integrator="direct", luminaire_samples=1, bsdf_samples=0, **kw))
    bs, _ = orc.render(fs.scene, cam, orc.render_params(-1, integrator="direct", luminaire_samples=0, bsdf_samples=1, **kw))
    m = [orc.develop(f).mean() for f in (ref, both, lum, bs)]
    assert abs(m[1] / m[0] - 1) < 0.03 and abs(m[2] / m[0] - 1) < 0.05 and abs(m[3] / m[0] - 1) < 0.15, m
    # at most two closest-hit rays and one shadow ray per camera sample
    n = 40 * 40 * 64
    assert n < st.rays_closest <= 2 * n and 0 < st.rays_shadow <= n
    # MIS with both strategies has less variance than BSDF sampling alone
    err = lambda f: float(np.mean((orc.develop(f) - orc.develop(ref)) ** 2))
    assert err(both) < err(bs)


def test_sample_arrays_of_the_three_samplers(orc):
    """Sampler::request2DArray / next2DArray (sampler.cpp:71-87) as generate() fills the arrays: plain draws
    (independent.cpp:63-66), one shuffled (0,2)-sequence (ldsampler.cpp:129-141,152-153), latinHypercube
    (util.cpp:529-540, stratified.cpp:136-138)"""
    import ctypes as C
    L = orc.lib()
    for f in (L.orc_independent_generate_array, L.orc_ld_generate_array, L.orc_latin_hypercube_array):
        f.argtypes = [C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_float)]; f.restype = None
    L.orc_ulong_to_float.argtypes = [C.c_uint64]; L.orc_ulong_to_float.restype = C.c_float
    def gen(f, n, key=(7, 11, 0)):
        st = C.c_uint64(L.orc_keyed_init(*key)); out = np.zeros((n, 2), dtype=np.float32)
        f(C.byref(st), n, out.ctypes.data_as(C.POINTER(C.c_float)))
        return out, st.value
    # independent: 2n consecutive draws of the stream, x first
    a, end = gen(L.orc_independent_generate_array, 50)
    st = C.c_uint64(L.orc_keyed_init(7, 11, 0))
    exp = np.array([L.orc_ulong_to_float(L.orc_keyed_next(C.byref(st))) for _ in range(100)], dtype=np.float32).reshape(50, 2)
    assert np.array_equal(a, exp) and end == st.value
    # ldsampler: 64 points of a scrambled (0,2)-sequence -- one point in every elementary interval of area 1/64
    a, _ = gen(L.orc_ld_generate_array, 64)
    for bx in range(7):
        cells = (np.floor(a[:, 0] * (1 << bx)).astype(int) << (6 - bx)) | np.floor(a[:, 1] * (1 << (6 - bx))).astype(int)
        assert sorted(cells) == list(range(64)), bx
    b, _ = gen(L.orc_ld_generate_array, 64, key=(7, 12, 0))
    assert not np.array_equal(a, b)
    # stratified: each coordinate visits each of the n strata exactly once, and the two are shuffled independently
    for n in (16, 60, 256):
        a, _ = gen(L.orc_latin_hypercube_array, n)
        sx, sy = np.floor(a[:, 0] * n).astype(int), np.floor(a[:, 1] * n).astype(int)
        assert sorted(sx) == list(range(n)) and sorted(sy) == list(range(n))
        assert not np.array_equal(sx, np.arange(n)) and not np.array_equal(sx, sy)
        assert (a >= 0).all() and (a < 1).all()


def test_direct_integrator_with_several_samples(mts, orc):
    """luminaireSamples / bsdfSamples > 1 (direct.cpp:129-150,163-195): the same expectation as one sample each, less
    variance per camera sample; the weights 1/N and the sample-count fractions of the MIS terms are in place"""
    sd = mts.scenes.cornell_c1()
    fs = orc.FlatScene(sd)
    cam = orc.make_camera(sd, 32, 32)
    ref, _ = orc.render(fs.scene, cam, orc.render_params(-1, integrator="direct", sampler=mts.abi.SAMPLER_LD_KEYED, spp=1024, seed=5))
    refimg = orc.develop(ref)
    # pixels inside one wall: elsewhere geometric aliasing at 16 camera samples hides the noise of the estimator
    lum = refimg.mean(axis=2); pad = np.pad(lum, 1, mode="edge")
    loc = np.stack([pad[i:i + 32, j:j + 32] for i in range(3) for j in range(3)])
    flat = (loc.max(0) - loc.min(0)) < 0.08 * np.maximum(lum, 1e-3)
    assert flat.sum() > 40
    err = {}
    for sampler in (mts.abi.SAMPLER_INDEPENDENT_KEYED, mts.abi.SAMPLER_LD_KEYED, mts.abi.SAMPLER_STRATIFIED_KEYED):
        for nl, nb in ((1, 1), (4, 4), (8, 0), (0, 8), (6, 2)):
            f, st = orc.render(fs.scene, cam, orc.render_params(-1, integrator="direct", sampler=sampler, spp=16, seed=9,
                                                                 luminaire_samples=nl, bsdf_samples=nb))
            img = orc.develop(f)
            assert abs(img.mean() / refimg.mean() - 1) < 0.05, (sampler, nl, nb, img.mean(), refimg.mean())
            err[(sampler, nl, nb)] = float(np.mean((img - refimg)[flat] ** 2))
            n = 32 * 32 * 16
            assert st.rays_shadow <= n * max(nl, 0) and st.rays_closest <= n * (1 + nb)
            if nl == 0: assert st.rays_shadow == 0
        assert err[(sampler, 4, 4)] < 0.5 * err[(sampler, 1, 1)], (sampler, err)


def test_atan2_is_faithful(orc):
    L = orc.lib()
    rng = np.random.RandomState(2)
    ys = np.concatenate([rng.randn(4000), [0.0, -0.0, 1.0, -1.0, 0.0, -0.0, 3e-30, -2e30]]).astype(np.float32)
    xs = np.concatenate([rng.randn(4000), [1.0, 1.0, 0.0, 0.0, -1.0, -1.0, -1e-30, 1e30]]).astype(np.float32)
    for y, x in zip(ys, xs):
        got = L.orc_atan2f(float(y), float(x))
        exp = np.arctan2(np.float64(y), np.float64(x))
        assert abs(got - exp) <= 1.2e-7 * max(1.0, abs(exp)), (y, x, got, exp)
    assert L.orc_atan2f(0.0, -1.0) == np.float32(np.pi) and L.orc_atan2f(-0.0, -1.0) == -np.float32(np.pi)


def test_envmap_luminaire(mts, orc):
    """`envmap` (src/luminaires/envmap.cpp): MIPMap level 0 + the sampling density; sample() and pdf() agree with each
    other, the sampled directions follow the density (chi-square over the density's own cells), and the film of an
    env-lit scene is finite and lit"""
    import ctypes as C
    from scipy import stats
    sd = mts.scenes.envlit()
    fs = orc.FlatScene(sd)
    arr = fs.arrays()
    W, H, pw, ph = arr["env_size"]
    assert (W, H) == (128, 64) and (pw, ph) == (16, 8)                 # 96x40 up-sampled; level 3 of the pyramid
    assert abs(arr["env_pdf"].sum() - 1) < 1e-5 and arr["env_cdf"][0] == 0 and arr["env_cdf"][-1] == 1
    assert (arr["env_pixels"] >= 0).all() and arr["env_pixels"].max() > 20     # the sun survives the Lanczos filter
    # the product's host code builds the same arrays bit for bit
    pscene = mts.Scene(sd)
    prod = mts.abi.scene_arrays(pscene.sc)
    for k in ("env_pixels", "env_pdf", "env_cdf", "lum_params"):
        assert np.asarray(arr[k]).tobytes() == np.asarray(prod[k]).tobytes(), k
    L = orc.lib()
    p = np.array([0.1, 0.5, 0.2], dtype=np.float32)
    out = np.zeros(13, dtype=np.float32)
    rng = np.random.RandomState(8)
    n = 20000
    counts = np.zeros(pw * ph)
    M = arr["lum_params"][0][7:16].reshape(3, 3).astype(np.float64)
    for s in rng.rand(n, 2).astype(np.float32):
        L.orc_luminaire_sample(fs.scene, 0, _p(p), _p(s), _p(out))
        assert out[12] > 0
        d = out[6:9].astype(np.float64)
        assert abs(np.linalg.norm(d) - 1) < 1e-5
        # Scene::pdfLuminaire of the sampled direction == the density sample() reported (same table cell)
        pdf = L.orc_luminaire_pdf(fs.scene, 0, _p(p), _p(out[0:3].copy()), _p(out[3:6].copy()), _p(out[6:9].copy()))
        dl = M @ (-d)
        u = 0.5 * (1 + np.arctan2(dl[0], -dl[2]) / np.pi) * pw; v = np.arccos(np.clip(dl[1], -1, 1)) / np.pi * ph
        if min(u % 1, 1 - u % 1, v % 1, 1 - v % 1) > 1e-3:             # away from the cell borders
            if abs(dl[1]) < 0.999:                                     # 1 - d.y^2 cancels next to the poles (envmap.cpp:190)
                assert abs(pdf - out[12]) <= 2e-3 * out[12], (pdf, out[12])
            counts[min(int(u), pw - 1) + pw * min(int(v), ph - 1)] += 1
        # the sampled point lies on the bounding sphere, the normal points inwards
        c, r = arr["lum_params"][0][3:6], arr["lum_params"][0][6]
        assert abs(np.linalg.norm(out[0:3] - c) - r) < 1e-3 * r
    e = arr["env_pdf"].astype(np.float64) * counts.sum()
    big = e >= 5
    chi2 = ((counts[big] - e[big]) ** 2 / e[big]).sum()
    assert 1 - stats.chi2.cdf(chi2, big.sum() - 1) > 0.003
    cam = orc.make_camera(sd, 48, 36)
    prm = orc.render_params(sd.max_depth, sampler=mts.abi.SAMPLER_LD_KEYED, spp=16, seed=4)
    film, st = orc.render(fs.scene, cam, prm)
    img = orc.develop(film)
    assert np.isfinite(film).all() and img.mean() > 0.1 and img[:10].mean() > img[-10:].mean() * 0.5


def test_sphere_shape(mts, orc):
    """`sphere` shapes (src/shapes/sphere.cpp): one kd-tree primitive each; hits agree with the analytic
    intersection in binary64; a sphere-shaped area luminaire lights the box; both hosts flatten identically"""
    sd = mts.scenes.spheres()
    fs = orc.FlatScene(sd)
    arr = mts.abi.scene_arrays(fs.scene.contents)
    assert list(arr["shape_type"][-8:]) == [1] * 8 and arr["tri_idx"][-1, 0] == 0xFFFFFFFF
    assert (arr["triaccel"][-8:, 0] == 0xFFFFFFFF).all() and (arr["triaccel"][:-8, 0] <= 3).all()
    pscene = mts.Scene(sd)                      # keep it alive: .sc points into it
    prod = mts.abi.scene_arrays(pscene.sc)
    for k in arr:
        assert np.asarray(arr[k]).tobytes() == np.asarray(prod[k]).tobytes(), k
    from conftest import chord_rays
    rays = chord_rays(20000, (0, 1, 0), 2.2, seed=11)
    hits = orc.trace_rays(fs.scene, rays)
    nprim = len(arr["tri_idx"])
    on_sphere = hits[:, 3] >= nprim - 8
    on_sphere &= hits[:, 3] != 0xFFFFFFFF
    assert on_sphere.sum() > 100
    t = hits[:, 0].view(np.float32)
    P = arr["shape_params"]
    o, d = rays[:, 0:3].astype(np.float64), rays[:, 4:7].astype(np.float64)
    for i in np.flatnonzero(on_sphere)[:2000]:
        shape = int(arr["triaccel"][hits[i, 3], 10])
        c, r = P[shape, 0:3].astype(np.float64), float(P[shape, 3])
        p = o[i] + float(t[i]) * d[i]
        assert abs(np.linalg.norm(p - c) - abs(r)) < 2e-5 * max(1.0, float(t[i]))
    cam = orc.make_camera(sd, 40, 40)
    prm = orc.render_params(sd.max_depth, sampler=mts.abi.SAMPLER_LD_KEYED, spp=16, seed=4)
    film, st = orc.render(fs.scene, cam, prm)
    img = orc.develop(film)
    assert np.isfinite(film).all() and img.mean() > 0.05
    # removing the sphere luminaire darkens the image: it is really sampled (Sphere::sampleSolidAngle)
    sd2 = mts.scenes.spheres(); sd2.meshes.pop(); sd2.lum_type.pop(); sd2.lum_params.pop()
    fs2 = orc.FlatScene(sd2)
    film2, _ = orc.render(fs2.scene, cam, prm)
    assert orc.develop(film2).mean() < 0.93 * img.mean()


def test_phong_chi_square_and_twosided(orc):
    """phong is in the reference's own chi-square list (data/tests/test_bsdf.xml); twosided mirrors the lobe"""
    import ctypes as C
    P = np.zeros(16, dtype=np.float32)
    # exponent 20, kd = ks = 1 (no renormalisation needed), rd .4, rs .3 -> sampling weights as Phong::configure
    ssw = np.float32(0.3) / np.float32(0.7)
    P[:11] = [20, 1, 1, ssw, np.float32(1) - ssw, .4, .4, .4, .3, .3, .3]
    p, frac = _chi2_bsdf(orc, 5, P[:11], (0.4, 0.1, 0.85), n=40000)
    assert frac > 0.9 and p > 0.005, p
    L = orc.lib()
    wi = np.array([0.3, -0.2, 0.8], dtype=np.float32); wi /= np.linalg.norm(wi)
    wo = np.array([-0.1, 0.4, 0.7], dtype=np.float32); wo /= np.linalg.norm(wo)
    f1 = np.zeros(3, dtype=np.float32); f2 = np.zeros(3, dtype=np.float32)
    L.orc_bsdf_f(5 | 0x100, _p(P), _p(wi), _p(wo), _p(f1))
    nwi, nwo = wi * np.float32([1, 1, -1]), wo * np.float32([1, 1, -1])
    L.orc_bsdf_f(5 | 0x100, _p(P), _p(nwi), _p(nwo), _p(f2))
    assert np.array_equal(f1, f2) and f1.max() > 0
    L.orc_bsdf_f(5, _p(P), _p(nwi), _p(nwo), _p(f2))
    assert f2.max() == 0                                            # one-sided without the adapter


def test_pow_acos_disk_are_faithful(orc):
    L = orc.lib()
    import ctypes as C
    L.orc_powf.argtypes = [C.c_float, C.c_float]; L.orc_powf.restype = C.c_float
    L.orc_acosf.argtypes = [C.c_float]; L.orc_acosf.restype = C.c_float
    rng = np.random.RandomState(2)
    for x, y in zip(rng.rand(3000).astype(np.float32), (rng.rand(3000) * 40).astype(np.float32)):
        got, exp = np.float32(L.orc_powf(float(x), float(y))), np.float32(np.float64(x) ** np.float64(y))
        assert abs(np.float64(got) - np.float64(exp)) <= np.spacing(np.abs(exp)) or exp < 1e-37
    assert L.orc_powf(0.0, 3.0) == 0 and L.orc_powf(5.0, 0.0) == 1
    for x in np.concatenate([rng.rand(2000) * 2 - 1, [1.0, -1.0, 0.0]]).astype(np.float32):
        got, exp = np.float32(L.orc_acosf(float(x))), np.float32(np.arccos(np.float64(x)))
        assert abs(np.float64(got) - np.float64(exp)) <= 2 * np.spacing(np.abs(exp)) + 1e-30
    out = np.zeros(2, dtype=np.float32)
    for s in rng.rand(500, 2).astype(np.float32):
        L.orc_square_to_disk_concentric(_p(np.ascontiguousarray(s)), _p(out))
        assert out[0] ** 2 + out[1] ** 2 <= 1 + 1e-6


def test_point_light_direct_illumination_is_analytic(mts, orc):
    """one delta luminaire, maxDepth 2: Li = rho/pi * I/d^2 * cos(theta) (point.cpp:55-63, weight 1 for delta lights)"""
    sd = mts.scenes.SceneDescription("floor")
    pos, tri = mts.scenes._quad((-5, 0, -5), (10, 0, 0), (0, 0, 10), (0, 1, 0))
    sd.add_mesh(pos, tri, bsdf=sd.lambertian(0.5), face_normals=True)
    sd.point_light((0.0, 2.0, 0.0), 10.0)
    sd.camera = dict(origin=(0.0, 3.0, 0.001), target=(0.0, 0.0, 0.0), up=(0.0, 0.0, -1.0), fov=30.0)
    fs = orc.FlatScene(sd)
    cam = orc.make_camera(sd, 33, 33)
    f, _ = orc.render(fs.scene, cam, orc.render_params(2, spp=4))
    img = orc.develop(f)
    centre = img[16, 16, 0]
    assert abs(centre - 0.5 / np.pi * 10.0 / 4.0) < 2e-3            # straight below the light: d = 2, cos = 1
    assert img[0, 0, 0] < centre


def test_furnace_analytic_radiance(mts, orc):
    """closed-form radiance in a constant environment (the whole Li estimator: MIS weights, throughput, Russian
    roulette, background pdf): a convex lambertian body shows rho * Le, a dielectric or mirror body returns
    (reflectance) * Le whatever the path length; and the direct light of a spherical area luminaire (solid-angle
    sampling of src/shapes/sphere.cpp + MIS) on a lambertian floor: Lo = rho * L * (R / d)^2 straight below it"""
    def sphere_scene(bsdf_fn, env=2.0):
        sd = mts.scenes.SceneDescription("furnace")
        sd.add_sphere((0.0, 0.0, 0.0), 1.0, bsdf=bsdf_fn(sd))
        sd.add_lum(mts.abi.LUM_CONSTANT, [env, env, env])
        sd.camera = dict(origin=(0.0, 0.0, 6.0), target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov=8.0)   # all pixels on the sphere
        return sd
    cases = [
        ("lambertian", lambda sd: sd.add_bsdf(mts.abi.BSDF_LAMBERTIAN, [0.5, 0.7, 0.2]), (1.0, 1.4, 0.4), 0.012),
        ("dielectric", lambda sd: sd.dielectric(1.5, 1.0), (2.0, 2.0, 2.0), 0.004),
        ("mirror", lambda sd: sd.mirror(0.9), (1.8, 1.8, 1.8), 2e-5),
        ("difftrans", lambda sd: sd.difftrans(0.6), None, None),
    ]
    for name, fn, expect, tol in cases:
        sd = sphere_scene(fn)
        fs = orc.FlatScene(sd)
        cam = orc.make_camera(sd, 24, 24)
        for sampler in (mts.abi.SAMPLER_INDEPENDENT_KEYED, mts.abi.SAMPLER_LD_KEYED):
            prm = orc.render_params(-1, rr_depth=10, sampler=sampler, spp=256, seed=3)
            film, _ = orc.render(fs.scene, cam, prm)
            img = orc.develop(film)
            assert (film[..., 3] == film[..., 4]).all()                         # every camera ray hits the sphere
            m = img.reshape(-1, 3).mean(axis=0)
            if expect is None:
                # light diffuses through the sphere's wall twice per crossing: strictly between 0 and Le, no NaN
                assert np.isfinite(img).all() and 0.05 < m[0] < 2.0
                continue
            for c in range(3):
                assert abs(m[c] / expect[c] - 1) < tol, (name, sampler, m, expect)

    # --- spherical area light above a lambertian floor, maxDepth 2 (direct light only) ---
    sd = mts.scenes.SceneDescription("spherelight")
    pos, tri = mts.scenes._quad((-6, 0, -6), (12, 0, 0), (0, 0, 12), (0, 1, 0))
    sd.add_mesh(pos, tri, bsdf=sd.lambertian(0.5), face_normals=True)
    lum = sd.add_lum(mts.abi.LUM_AREA, [10.0, 10.0, 10.0])
    sd.add_sphere((0.0, 2.0, 0.0), 0.5, bsdf=sd.lambertian(0.0), lum=lum)
    sd.camera = dict(origin=(2.5, 1.5, 0.0), target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov=1.0)
    fs = orc.FlatScene(sd)
    cam = orc.make_camera(sd, 9, 9)
    film, _ = orc.render(fs.scene, cam, orc.render_params(2, sampler=mts.abi.SAMPLER_LD_KEYED, spp=1024, seed=2))
    centre = orc.develop(film)[4, 4, 0]
    assert abs(centre / (0.5 * 10.0 * (0.5 / 2.0) ** 2) - 1) < 0.01, centre

    # --- a uniform environment MAP must act like the constant luminaire (checks the density normalisation of
    #     src/luminaires/envmap.cpp:123-193 against the closed form) ---
    sd = mts.scenes.SceneDescription("furnace_envmap")
    sd.add_sphere((0.0, 0.0, 0.0), 1.0, bsdf=sd.add_bsdf(mts.abi.BSDF_LAMBERTIAN, [0.5, 0.7, 0.2]))
    sd.envmap(np.full((16, 32, 3), 2.0, dtype=np.float32), 1.0)
    sd.camera = dict(origin=(0.0, 0.0, 6.0), target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov=8.0)
    fs = orc.FlatScene(sd)
    film, _ = orc.render(fs.scene, orc.make_camera(sd, 24, 24), orc.render_params(-1, sampler=mts.abi.SAMPLER_LD_KEYED, spp=256, seed=3))
    m = orc.develop(film).reshape(-1, 3).mean(axis=0)
    for c, e in enumerate((1.0, 1.4, 0.4)):
        assert abs(m[c] / e - 1) < 0.015, m
