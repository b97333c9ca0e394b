"""The chi-square procedure of the reference's own BSDF test, src/tests/test_chisquare.cpp:299-420 (test01_BSDF) with
src/libcore/chisquare.cpp, restated over a batch evaluator `evaluate(bsdf_type, params, op, wi, aux) -> [n][8]` that has
the layout of mtsgpu_bsdf_eval: the GPU suite hands it the device hook, the CPU suite the oracle's.  Test infrastructure."""
import numpy as np

# ---------------------------------------------------------------------------------------------------------------------
# src/tests/test_chisquare.cpp:299-420 (test01_BSDF) + src/libcore/chisquare.cpp on the BSDFs of data/tests/test_bsdf.xml
# ---------------------------------------------------------------------------------------------------------------------
SIGNIFICANCE_LEVEL = 0.005          # test_chisquare.cpp:28
THETA_BINS, WI_SAMPLES = 10, 20     # test_chisquare.cpp:305
PHI_BINS = 2 * THETA_BINS           # ChiSquare(thetaBins, 2*thetaBins, wiSamples)
SAMPLE_COUNT = THETA_BINS * PHI_BINS * 1000      # chisquare.cpp:31-32
MIN_EXP_FREQUENCY = 5               # include/mitsuba/core/chisquare.h:28


def square_to_sphere(s):           # util.cpp:552-558
    z = np.float32(1) - np.float32(2) * s[:, 1]
    r = np.sqrt(np.maximum(np.float32(0), np.float32(1) - z * z))
    phi = np.float32(2 * np.pi) * s[:, 0]
    return np.stack([r * np.cos(phi), r * np.sin(phi), z], axis=1).astype(np.float32)


def square_to_hemisphere_psa(s):   # util.cpp:572-588
    r = np.sqrt(s[:, 0]); phi = np.float32(2 * np.pi) * s[:, 1]
    x, y = r * np.cos(phi), r * np.sin(phi)
    z = np.sqrt(np.maximum(np.float32(0), np.float32(1) - np.minimum(np.float32(1), x * x + y * y)))
    return np.stack([x, y, z], axis=1).astype(np.float32)


def run_test(table, ref, num_tests, dist_params=1):
    """ChiSquare::runTest (chisquare.cpp:127-216): cells in order of their expected counts, pooling below 5,
    Sidak-corrected significance level.  Returns (accepted, p-value, chi-square, degrees of freedom)."""
    from scipy import stats
    tolerance = SAMPLE_COUNT * 1e-4
    pooled_counts = pooled_ref = chsq = 0.0
    pooled_cells = df = 0
    for idx in np.argsort(ref, kind="stable"):
        if ref[idx] == 0:
            if table[idx] > tolerance:
                return False, 0.0, np.inf, 0
        elif ref[idx] < MIN_EXP_FREQUENCY or (0 < pooled_ref < MIN_EXP_FREQUENCY):
            pooled_counts += table[idx]; pooled_ref += ref[idx]; pooled_cells += 1
        else:
            diff = table[idx] - ref[idx]
            chsq += diff * diff / ref[idx]; df += 1
    if pooled_cells > 0:
        diff = pooled_counts - pooled_ref
        chsq += diff * diff / pooled_ref; df += 1
    df -= dist_params + 1
    assert df > 0, "too few degrees of freedom"
    pval = 1.0 - stats.chi2.cdf(chsq, df)
    alpha = 1.0 - (1.0 - SIGNIFICANCE_LEVEL) ** (1.0 / num_tests)
    return pval >= alpha, pval, chsq, df


def chi_square(evaluate, btype, params, back_side, rng, wi_samples=None, G=48, wis=None):
    """one BSDF model: WI_SAMPLES incident directions, 2-D sampling (pass 0 of test01_BSDF; the path tracer never hands the
    BSDF a sampler, so pass 1 does not apply, and the device has no per-component queries: bRec.component = -1)"""
    # Reference table: the integral of pdf(wo) sin(theta) over every (theta, phi) cell (ChiSquare::fill uses an adaptive
    # cubature, relative error 1e-6).  Here: G x G Gauss-Legendre nodes per cell; 20 x 20 is too coarse for roughglass at
    # grazing incidence, where the edge of the lobe cuts through cells (a false rejection), 40 x 40 and up agree.
    gx, gw = np.polynomial.legendre.leggauss(G)
    th_edges = np.linspace(0, np.pi, THETA_BINS + 1); ph_edges = np.linspace(0, 2 * np.pi, PHI_BINS + 1)
    th = (0.5 * (th_edges[:-1] + th_edges[1:])[:, None] + 0.5 * (np.pi / THETA_BINS) * gx[None, :])        # [T][G]
    ph = (0.5 * (ph_edges[:-1] + ph_edges[1:])[:, None] + 0.5 * (2 * np.pi / PHI_BINS) * gx[None, :])      # [P][G]
    TH, PH = th[:, None, :, None], ph[None, :, None, :]                                                    # [T][P][G][G]
    wo_grid = np.stack(np.broadcast_arrays(np.sin(TH) * np.cos(PH), np.sin(TH) * np.sin(PH), np.cos(TH) + 0 * PH), axis=-1)
    wo_grid = wo_grid.reshape(-1, 3).astype(np.float32)                     # sphericalDirection (util.cpp:543-550)
    weight = (np.sin(TH) * gw[None, None, :, None] * gw[None, None, None, :]).repeat(PHI_BINS, axis=1)
    weight = weight * (0.5 * np.pi / THETA_BINS) * (0.5 * 2 * np.pi / PHI_BINS)
    failures = []
    n_wi = WI_SAMPLES if wi_samples is None else wi_samples      # the Sidak correction stays that of the reference's 20 tests
    if wis is None:
        wis = (square_to_sphere if back_side else square_to_hemisphere_psa)(rng.random_sample((n_wi, 2)).astype(np.float32))
    for wi in wis:
        # BSDFAdapter::generateSample: weight 1 for a valid sample, 0 when f is zero or pdf == 0
        s = rng.random_sample((SAMPLE_COUNT, 2)).astype(np.float32)
        out = evaluate(btype, params, 2, wi, s)
        wo, pdf, f = out[:, 0:3].astype(np.float64), out[:, 3], out[:, 4:7]
        ok = (pdf != 0) & (f != 0).any(axis=1)
        theta = np.arccos(np.clip(wo[:, 2], -1, 1)); phi = np.arctan2(wo[:, 1], wo[:, 0]); phi[phi < 0] += 2 * np.pi
        tb = np.clip(np.floor(theta * (THETA_BINS / np.pi)).astype(int), 0, THETA_BINS - 1)
        pb = np.clip(np.floor(phi * (PHI_BINS / (2 * np.pi))).astype(int), 0, PHI_BINS - 1)
        table = np.bincount((tb * PHI_BINS + pb)[ok], minlength=THETA_BINS * PHI_BINS).astype(np.float64)
        # BSDFAdapter::pdf: 0 where f() is zero, pdf() otherwise
        fg = evaluate(btype, params, 0, wi, wo_grid)[:, 0:3]
        pg = evaluate(btype, params, 1, wi, wo_grid)[:, 0].astype(np.float64)
        pg[(fg == 0).all(axis=1)] = 0.0
        ref = (pg.reshape(weight.shape) * weight).sum(axis=(2, 3)).ravel() * SAMPLE_COUNT
        accepted, pval, chsq, df = run_test(table, ref, WI_SAMPLES)
        if not accepted:
            failures.append((wi.tolist(), pval, chsq, df, ok.mean()))
    return failures


def bsdf_models(mts):
    """the BSDF instances of data/tests/test_bsdf.xml that lie on this path (ward and composite do not), parameter
    blocks as the plugins' constructors / configure() leave them"""
    sd = mts.scenes.SceneDescription("test_bsdf.xml")
    models = [
        ("lambertian", sd.lambertian(0.5), False),                                                   # :6
        ("roughglass ggx alpha .4", sd.roughglass(0.4, 1.5, 1.0, "ggx"), True),                      # :10-15
        ("difftrans", sd.difftrans(0.5), True),                                                      # :18
        ("phong exponent 20", sd.phong(20.0, rd=1.0, rs=1.0, kd=0.5, ks=0.5), False),                # :21-28
        ("twosided phong", sd.twosided(sd.phong(20.0, rd=1.0, rs=1.0, kd=0.5, ks=0.5)), True),       # :43-52
        ("microfacet alphaB .1", sd.microfacet(0.1, 0.5, 0.5, 1.5, 1.0, 1.0, 1.0), False),           # :80-86
        ("roughmetal alphaB .1", sd.roughmetal(0.1), False),                                         # :89-91
        ("roughglass beckmann alpha .3", sd.roughglass(0.3, 1.5, 1.0, "beckmann"), True),            # :95-100
        ("roughglass ggx alpha .4 (second instance)", sd.roughglass(0.4, 1.5, 1.0, "ggx"), True),    # :104-109
        ("roughglass phong alpha .3", sd.roughglass(0.3, 1.5, 1.0, "phong"), True),                  # :113-118
    ]
    return [(name, sd.bsdf_type[i], sd.bsdf_params[i], back) for name, i, back in models]


