"""GPU tests, fifth set: the pipeline changes of round 5, each against the oracle or against the unchanged path."""
import numpy as np
import pytest

from test_gpu_parity import _setup

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("spp", [1024, 2048, 4096, 32768])
def test_ld_tables_above_512_samples(gpu_lib, mts, orc, spp):
    """LowDiscrepancySampler::generate() (ldsampler.cpp:125-158) above 512 samples per pixel: one wave per pixel finds the
    draws Random::nextSize accepts (random.cpp:196-215), one lane per table applies the swaps of Random::shuffle
    (random.h:145-148) -- in LDS, a wave per table, up to 16 384 samples; in memory, eight steps at a time, above -- scrambles
    and permutations equal the oracle's sequential loop"""
    sd, scene, oscene, cam, ocam, it, op = _setup(mts, orc, "c1", sampler="ldsampler", spp=spp)
    for key in (0, 7, 65535, 123456):
        t1, t2 = it.ld_tables(key, spp, 3)
        e1 = np.zeros((3, spp), dtype=np.float32)
        e2 = np.zeros((3, spp, 2), dtype=np.float32)
        orc.lib().orc_ld_generate_keyed(0x5EED, key, spp, 3, mts.abi.ptr(e1, mts.abi.f32p), mts.abi.ptr(e2, mts.abi.f32p))
        assert np.array_equal(t1.view(np.uint32), e1.view(np.uint32))
        assert np.array_equal(t2.view(np.uint32), e2.view(np.uint32))


@pytest.mark.parametrize("sampler", ["ldsampler", "stratified"])
def test_films_at_1024_samples_in_one_and_in_many_passes(gpu_lib, mts, orc, sampler):
    """the tables of 192 pixels (three waves of k_ld_apply per table, the last one ragged) and of ragged passes"""
    sd, scene, oscene, cam, ocam, it, op = _setup(mts, orc, "c5_small", W=16, H=12, sampler=sampler, spp=1024, max_depth=6)
    ofilm, _ = orc.render(oscene.scene, ocam, op)
    assert it.render()
    assert np.array_equal(it.film().view(np.uint32), ofilm.view(np.uint32))
    for max_paths in (1024 * 100, 1024 * 7):          # 2 and 28 passes (the last ones ragged)
        it.set_options(max_paths=max_paths)
        it.clear_film(); assert it.render()
        assert np.array_equal(it.film().view(np.uint32), ofilm.view(np.uint32)), max_paths
