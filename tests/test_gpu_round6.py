"""GPU tests, sixth set: what round 6 added or left uncovered, each against the oracle."""
import numpy as np
import pytest

from test_gpu_parity import _setup

pytestmark = pytest.mark.gpu


def test_ld_apply_in_memory_with_many_slots_groups_and_chunked_launches(gpu_lib, mts, orc):
    """LowDiscrepancySampler::generate() above 16 384 samples per pixel (k_ld_scout + k_ld_apply: a lane per table, the 64
    slots of a wave interleaved in a scratch copy, the result transposed through LDS into the per-pixel rows) with everything
    the single-slot table read-out cannot reach: 384 pixels = six groups of 64 slots, 48 tables per pixel = 288 (group, table)
    pairs = two chunked launches (block0 > 0), and a second pass whose last group is ragged (200 + 184 slots).  The film
    equals the oracle's in crops from the first group, a middle group, the pass boundary and the ragged last group"""
    W, H, spp, depth = 24, 16, 32768, 24
    sd, scene, oscene, cam, ocam, it, op = _setup(mts, orc, "c1", W=W, H=H, sampler="ldsampler", spp=spp, max_depth=2)
    it.preprocess(scene, cam, sampler="ldsampler", sampleCount=spp, depth=depth, seed=0x5EED)
    op = orc.render_params(2, sampler=mts.abi.SAMPLER_LD_KEYED, spp=spp, ld_depth=depth, seed=0x5EED)
    it.set_options(max_paths=200 * spp)
    assert it.render()
    film = it.film()
    assert it.stats()["camera_samples"] == W * H * spp
    for x0, y0 in ((0, 0), (10, 5), (6, 8), (8, 8), (22, 15)):      # slot 200 = pixel (8, 8): the first of the second pass
        o, _ = orc.render(oscene.scene, ocam, op, rect=(x0, y0, x0 + 2, y0 + 1))
        assert np.array_equal(film[y0:y0 + 1, x0:x0 + 2].view(np.uint32), o[y0:y0 + 1, x0:x0 + 2].view(np.uint32)), (x0, y0)
        assert o[y0, x0, 4] > 0
