"""CPU restatements (pure Python, small sizes) of the two parallel algorithms the sampler-table kernels use above 512 samples per
pixel (mitsuba-renderer_amd/csrc/sampler.hip: k_ld_scout, k_ld_apply_lds), against the sequential loop they replace:
Random::shuffle (include/mitsuba/core/random.h:145-148) driven by Random::nextSize (src/libcore/random.cpp:196-215) over the
keyed SplitMix64 stream.  The GPU tests compare the kernels themselves with the oracle; these tests pin the ARGUMENT -- that
the fixed point of the acceptance recurrence and the lane-ordered conflict phase reproduce the sequential states exactly."""
import random

M = (1 << 64) - 1
GAMMA = 0x9E3779B97F4A7C15


def mix(z):
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    return z ^ (z >> 31)


def sequential(st0, spp, n_tables):
    """the reference's order: one scramble draw, then for it = spp - 1 .. 1: swap(p[it], p[nextSize(it)])"""
    st, tables = st0, []
    for _ in range(n_tables):
        st = (st + GAMMA) & M
        scramble = mix(st)
        p = list(range(spp))
        for it in range(spp - 1, 0, -1):
            mask = (1 << it.bit_length()) - 1
            while True:
                st = (st + GAMMA) & M
                r = mix(st) & mask
                if r < it:
                    break
            p[it], p[r] = p[r], p[it]
        tables.append((scramble, p))
    return tables, st


def scout(st0, spp, n_tables, lanes=64):
    """k_ld_scout: `lanes` draws of the counter-based stream at a time; draw i is accepted by the step it stands at iff
    v_i < it - A_i, A_i = accepted draws before it -- solved as a fixed point of ballots"""
    drawn, out = 0, []
    for _ in range(n_tables):
        scramble = mix((st0 + (drawn + 1) * GAMMA) & M)
        drawn += 1
        other = [None] * spp
        it = spp - 1
        while it > 0:
            mask = (1 << it.bit_length()) - 1
            lo = (mask >> 1) + 1
            v = [mix((st0 + (drawn + l + 1) * GAMMA) & M) & mask for l in range(lanes)]
            acc = [v[l] + l < it for l in range(lanes)]
            while True:
                A = [sum(acc[:l]) for l in range(lanes)]
                nxt = [v[l] + A[l] < it for l in range(lanes)]
                if nxt == acc:
                    break
                acc = nxt
            total, avail = sum(acc), it - lo + 1
            consumed, steps, mine = lanes, total, list(acc)
            if total >= avail:          # the run of steps that share this mask ends inside the chunk
                last = [l for l in range(lanes) if acc[l] and A[l] == avail - 1][0]
                consumed, steps = last + 1, avail
                mine = [acc[l] and l <= last for l in range(lanes)]
            for l in range(lanes):
                if mine[l]:
                    other[it - A[l]] = v[l]
            it -= steps
            drawn += consumed
        out.append((scramble, other))
    return out, (st0 + drawn * GAMMA) & M


def apply_in_batches(other, spp, lanes=64, claim_slots=2048):
    """k_ld_apply_lds: lane l of a batch takes step it - l; it depends on an earlier lane j only if o_j == o_l or
    o_j == it - l, found through claim[x] = lowest lane whose partner is x (hashed: a false conflict only orders more
    lanes); independent lanes swap at once, the others in lane order"""
    p = list(range(spp))
    it0 = spp - 1
    while it0 >= 1:
        nb = min(lanes, it0)
        my_it = [it0 - l for l in range(nb)]
        o = [other[i] for i in my_it]
        claim = {}
        for l in range(nb):
            key = o[l] % claim_slots
            claim[key] = min(claim.get(key, 1 << 30), l)
        dep = [claim.get(o[l] % claim_slots, 1 << 30) < l or claim.get(my_it[l] % claim_slots, 1 << 30) < l for l in range(nb)]
        free = [l for l in range(nb) if not dep[l]]
        # the independent lanes at once: all reads, then all writes (what a SIMD instruction pair does)
        reads = {l: (p[my_it[l]], p[o[l]]) for l in free}
        for l in free:
            p[my_it[l]], p[o[l]] = reads[l][1], reads[l][0]
        for l in range(nb):
            if dep[l]:
                p[my_it[l]], p[o[l]] = p[o[l]], p[my_it[l]]
        it0 -= nb
    return p


def test_scout_finds_the_draws_the_sequential_loop_accepts():
    for spp in (2, 37, 64, 65, 520, 1024):
        for st0 in (12345, 0xDEADBEEFCAFEBABE, 0):
            want, st_want = sequential(st0, spp, 3)
            got, st_got = scout(st0, spp, 3)
            assert st_got == st_want
            for (sw, pw), (sg, other) in zip(want, got):
                assert sw == sg
                p = list(range(spp))
                for it in range(spp - 1, 0, -1):
                    p[it], p[other[it]] = p[other[it]], p[it]
                assert p == pw


def test_batched_swaps_with_ordered_conflicts_equal_the_sequential_swaps():
    rng = random.Random(5)
    for spp in (2, 3, 63, 64, 65, 200, 1024, 4096):
        for claim_slots in (2048, 16):          # 16: nearly every lane in false conflict -- still the same permutation
            other = [None] + [rng.randrange(it) for it in range(1, spp)]
            want = list(range(spp))
            for it in range(spp - 1, 0, -1):
                want[it], want[other[it]] = want[other[it]], want[it]
            assert apply_in_batches(other, spp, claim_slots=claim_slots) == want
    # adversarial partners: everybody points at entry 0, and a chain where o_j is the `it` position of lane j + 1
    for other in ([None] + [0] * 255, [None] + [max(0, it - 1) for it in range(1, 256)]):
        spp = len(other)
        want = list(range(spp))
        for it in range(spp - 1, 0, -1):
            want[it], want[other[it]] = want[other[it]], want[it]
        assert apply_in_batches(other, spp) == want
