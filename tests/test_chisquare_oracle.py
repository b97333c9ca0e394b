"""The oracle's BSDFs under the reference's own chi-square procedure (tests/chisquare_ref.py restates
src/tests/test_chisquare.cpp:299-420 + src/libcore/chisquare.cpp) with the models of data/tests/test_bsdf.xml, on CPU.
The GPU suite runs the same procedure on the device code (tests/test_gpu_round3.py)."""
import numpy as np
import pytest

from chisquare_ref import WI_SAMPLES, bsdf_models, chi_square


@pytest.mark.parametrize("index", range(10))
def test_oracle_bsdfs_pass_the_reference_chi_square(mts, orc, index):
    name, btype, params, back = bsdf_models(mts)[index]
    failures = chi_square(orc.bsdf_eval, btype, params, back, np.random.RandomState(2000 + index))
    assert not failures, "%s: rejected for %d of %d incident directions: %s" % (name, len(failures), WI_SAMPLES, failures[:3])


def test_the_procedure_rejects_a_wrong_density(mts, orc):
    """negative control: samples drawn with alphaB = 0.1 against the density of alphaB = 0.11 must be rejected"""
    name, btype, params, back = bsdf_models(mts)[6]          # roughmetal
    wrong = params.copy(); wrong[0] = 0.11

    def evaluate(t, P, op, wi, aux):
        return orc.bsdf_eval(t, P if op == 2 else wrong, op, wi, aux)
    failures = chi_square(evaluate, btype, params, back, np.random.RandomState(7), wi_samples=6)
    assert len(failures) >= 3, failures
