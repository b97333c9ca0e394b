// sampler.hip -- K0: Sampler::generate() per pixel (the tables of ldsampler / stratified), the sample arrays, the
// reference's Random (MT19937-64) on the device, and the sampler read-out for tests.
#include "sampler.h"

namespace mg {

// ===========================================================================
// K0: LowDiscrepancySampler::generate() per pixel (src/samplers/ldsampler.cpp:125-158)
// with the keyed stream in place of Random.  The tables hold the permutation;
// values are f(perm[j]) at lookup time.
//
// Random::shuffle (random.h:145-148) is `for it = n - 1 .. 1: swap(p[it], p[nextSize(it)])`, and
// Random::nextSize (random.cpp:196-215) rejects: both the stream position of a step and the array
// it works on depend on all steps before it.  Above 512 samples per pixel (below, the tables of 64
// pixels fit the LDS of one wave: k_ld_tables_lds) the two chains are taken apart:
//   k_ld_scout   one WAVE per pixel walks the pixel's stream 64 draws at a time -- the stream is
//                counter based, lane i evaluates draw number base + i -- and finds out which draws
//                the steps accept; it leaves the partner index other[it] of every step of every
//                table (in the table's own row), the scrambles and the final stream position;
//   k_ld_apply_lds  (up to 16 384 samples) one wave per (pixel, table) applies the swaps in LDS, 64 steps at a time;
//   k_ld_apply   (above) one lane per (pixel, table) applies the swaps, 8 steps at a time with their 16
//                loads in flight together, in a scratch copy where the 64 pixels of a wave are
//                interleaved ([entry][lane]: p[it] is one line per access, not 64), and transposes
//                the result through LDS into the per-pixel row.
// C4 pass (18 k pixels x 4096 spp): scout 0.7 ms, apply 2.3 ms in LDS (11.5 ms in memory: one random 2-byte read
// and one random 2-byte write per step, each a whole line across the XCD's link); a lane per pixel doing everything
// took 15-21 ms in every memory layout tried (profiles/r05m_exp_sampler_tables_4096spp.txt).
// ===========================================================================
__global__ __launch_bounds__(256) void k_ld_scout(DConfig cfg, const uint32_t *pixel_keys, uint32_t n_slots,
                                                  uint32_t *scr, uint16_t *others, unsigned long long *state) {
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t slot = blockIdx.x * 4u + (threadIdx.x >> 6);      // one wave per sampler slot
	if (slot >= n_slots)
		return;
	const uint32_t spp = cfg.spp;
	const int depth = cfg.ld_depth;
	const bool ld = cfg.sampler_kind == 1;     // 4: StratifiedSampler::generate (stratified.cpp:121-141), permutations only
	constexpr uint64_t kGamma = 0x9E3779B97F4A7C15ULL;           // keyedNext: state += gamma; return sm64mix(state)
	const uint64_t st0 = keyedInit(cfg.seed, pixel_keys[slot], 0);
	uint64_t drawn = 0;                        // draws consumed so far (wave-uniform): draw number d is sm64mix(st0 + (d + 1) gamma)
	uint32_t *s = scr + (size_t) slot * 3 * depth;
	for (int arr = 0; arr < 2 * depth; ++arr) {
		if (ld) {
			// generate1D: the low half of one draw; generate2D: one 64-bit draw, dword[0] = low half, dword[1] = high half
			const uint64_t q = sm64mix(st0 + (drawn + 1) * kGamma);
			++drawn;
			if (lane == 0) {
				const int i = arr >> 1;
				if ((arr & 1) == 0) {
					s[i * 3 + 0] = (uint32_t) (q & 0xFFFFFFFFull);
				} else {
					s[i * 3 + 1] = (uint32_t) (q & 0xFFFFFFFFull);
					s[i * 3 + 2] = (uint32_t) (q >> 32);
				}
			}
		}
		uint16_t *row = others + ((size_t) slot * 2 * depth + arr) * spp;
		uint32_t it = spp - 1;
		while (it > 0) {
			// the steps it, it - 1, .. down to the highest bit of `it` share nextSize's bit mask
			const uint32_t mask = 0xFFFFFFFFu >> __builtin_clz(it), lo = (mask >> 1) + 1u;
			const uint32_t v = (uint32_t) sm64mix(st0 + (drawn + lane + 1) * kGamma) & mask;
			// draw i is accepted by the step it stands at, it - A_i, iff v_i < it - A_i, A_i = accepted draws before it.  A
			// fixed point of that recurrence is its (unique) sequential solution; every sweep settles at least one more lane
			uint64_t acc = __builtin_amdgcn_ballot_w64(v + lane < it);             // accepted whatever happened before
			uint32_t A;
			while (true) {
				A = __builtin_amdgcn_mbcnt_hi((uint32_t) (acc >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) acc, 0u));
				const uint64_t acc2 = __builtin_amdgcn_ballot_w64(v + A < it);
				if (acc2 == acc) break;
				acc = acc2;
			}
			const bool a = v + A < it;
			const uint32_t total = (uint32_t) __popcll(acc), avail = it - lo + 1u;
			uint32_t consumed = 64u, steps = total;
			bool mine = a;
			if (total >= avail) {
				// the run of steps with this mask ends inside the chunk: the draws after its last accepted one belong to the next mask
				const uint64_t last = __builtin_amdgcn_ballot_w64(a && A == avail - 1u);
				const uint32_t L = (uint32_t) __builtin_ctzll(last);
				consumed = L + 1u; steps = avail;
				mine = a && lane <= L;
			}
			if (mine) row[it - A] = (uint16_t) v;
			it -= steps;
			drawn += consumed;
		}
	}
	if (lane == 0) state[slot] = st0 + drawn * kGamma;
}

__global__ __launch_bounds__(64) void k_ld_apply(DConfig cfg, uint32_t n_slots, uint32_t block0, uint16_t *perm, uint16_t *scratch) {
	__shared__ uint16_t s_tile[64][66];        // [entry][lane], padded: the transposed reads spread over the banks
	const int nArr = 2 * cfg.ld_depth;
	const uint32_t blk = block0 + blockIdx.x;  // (group of 64 slots, table)
	const uint32_t group = blk / (uint32_t) nArr;
	const int arr = (int) (blk - group * (uint32_t) nArr);
	const uint32_t lane = threadIdx.x, slot0 = group * 64u, slot = slot0 + lane;
	const uint32_t spp = cfg.spp;
	uint16_t *S = scratch + (size_t) blockIdx.x * 64u * spp;
	if (slot < n_slots) {
		const uint16_t *row = perm + ((size_t) slot * nArr + arr) * spp;       // other[it], left by k_ld_scout
		for (uint32_t k = 0; k < spp; ++k) S[(size_t) k * 64u + lane] = (uint16_t) k;
		// Steps of a batch that touch the same entry -- other_j == other_k, or other_j == it_k for j < k; the it are distinct
		// and other_k < it_k -- are resolved in registers in step order, and the stores leave in step order, so the array
		// goes through exactly the states of the sequential loop.
		constexpr int kBatch = 8;
		uint32_t it = spp - 1;
		for (; it >= (uint32_t) kBatch; it -= (uint32_t) kBatch) {          // steps it, it - 1, .., it - kBatch + 1 (all >= 1)
			uint32_t oth[kBatch];
			uint16_t va[kBatch], vb[kBatch];
			#pragma unroll
			for (int j = 0; j < kBatch; ++j) oth[j] = row[it - (uint32_t) j];
			#pragma unroll
			for (int j = 0; j < kBatch; ++j) { va[j] = S[(size_t) (it - (uint32_t) j) * 64u + lane]; vb[j] = S[(size_t) oth[j] * 64u + lane]; }
			#pragma unroll
			for (int k = 0; k < kBatch; ++k) {
				uint16_t a = va[k], b = vb[k];
				#pragma unroll
				for (int j = 0; j < k; ++j) {          // vb[j] now holds what step j left at position oth[j]; the latest j wins
					if (oth[j] == it - (uint32_t) k) a = vb[j];
					if (oth[j] == oth[k]) b = vb[j];
				}
				va[k] = b;            // -> p[it - k]
				vb[k] = a;            // -> p[oth[k]]
			}
			#pragma unroll
			for (int k = 0; k < kBatch; ++k) { S[(size_t) (it - (uint32_t) k) * 64u + lane] = va[k]; S[(size_t) oth[k] * 64u + lane] = vb[k]; }
		}
		for (; it > 0; --it) {
			const uint32_t other = row[it];
			uint16_t *pa = S + (size_t) it * 64u + lane, *pb = S + (size_t) other * 64u + lane;
			const uint16_t a = *pa, b = *pb;
			*pa = b; *pb = a;
		}
	}
	__syncthreads();           // one wave: orders the lanes' scratch writes (and their reads of `row`) before the transposed pass below
	const uint32_t rows = (n_slots - slot0 < 64u) ? n_slots - slot0 : 64u;
	for (uint32_t k0 = 0; k0 < spp; k0 += 64u) {
		const uint32_t nk = (spp - k0 < 64u) ? spp - k0 : 64u;
		if (lane < rows)
			for (uint32_t k = 0; k < nk; ++k) s_tile[k][lane] = S[(size_t) (k0 + k) * 64u + lane];
		__syncthreads();
		if (lane < nk)
			for (uint32_t r = 0; r < rows; ++r)
				perm[((size_t) (slot0 + r) * nArr + arr) * spp + k0 + lane] = s_tile[lane][r];
		__syncthreads();
	}
}

// The same for tables that fit LDS next to a claim array (up to 16 384 samples per pixel): one WAVE per (pixel, table).
// Lane l of a batch takes step it - l: positions it - l and o_l = other[it - l].  It depends on an earlier lane j < l only
// if o_j == o_l or o_j == it - l (the `it` positions are distinct and o_l < it - l < it - j), which a claim array finds --
// claim[x] = lowest lane whose partner is x, by an LDS atomic minimum; 2048 slots, positions 2048 apart share one: a false
// conflict only moves a lane to the ordered phase.  Lanes without such a j swap at once (they share no position with any
// earlier lane, so their swaps commute with everything before them), the others follow in lane order: about 4096 / it lanes
// of 64.  Every swap is an LDS access: no line crosses the XCD's link (2.3 ms per C4 pass against 11.5 for k_ld_apply).
constexpr uint32_t kClaimSlots = 2048;
__global__ __launch_bounds__(64) void k_ld_apply_lds(DConfig cfg, uint32_t n_tables, uint16_t *perm) {
	extern __shared__ uint32_t s_apply[];
	uint32_t *claim = s_apply;                                                    // [kClaimSlots]
	uint16_t *p = reinterpret_cast<uint16_t *>(s_apply + kClaimSlots);            // [spp]
	const uint32_t spp = cfg.spp, lane = threadIdx.x;
	if (blockIdx.x >= n_tables) return;
	uint16_t *row = perm + (size_t) blockIdx.x * spp;      // table (slot, arr) = row slot * 2 depth + arr: other[] in, permutation out
	for (uint32_t k = lane; k < spp; k += 64u) p[k] = (uint16_t) k;
	uint32_t it0 = spp - 1;
	uint32_t oNext = (lane < it0) ? row[it0 - lane] : 0u;                       // steps it0 - lane >= 1
	__syncthreads();
	while (it0 >= 1u) {
		const uint32_t nb = it0 < 64u ? it0 : 64u;
		const bool active = lane < nb;
		const uint32_t myIt = it0 - lane, o = oNext;
		const uint32_t itN = it0 - nb;                                             // the batch after this one, requested now
		oNext = (itN >= 1u && lane < itN) ? row[itN - lane] : 0u;
		if (active) { claim[o & (kClaimSlots - 1u)] = 0xFFFFFFFFu; claim[myIt & (kClaimSlots - 1u)] = 0xFFFFFFFFu; }
		__syncthreads();
		if (active) atomicMin(&claim[o & (kClaimSlots - 1u)], lane);
		__syncthreads();
		const bool dep = active && (claim[o & (kClaimSlots - 1u)] < lane || claim[myIt & (kClaimSlots - 1u)] < lane);
		if (active && !dep) { const uint16_t a = p[myIt], b = p[o]; p[myIt] = b; p[o] = a; }
		__syncthreads();
		uint64_t todo = __builtin_amdgcn_ballot_w64(dep);
		while (todo) {
			const uint32_t k = (uint32_t) __builtin_ctzll(todo);
			todo &= todo - 1ull;
			if (lane == k) { const uint16_t a = p[myIt], b = p[o]; p[myIt] = b; p[o] = a; }
			__syncthreads();
		}
		it0 = itN;
	}
	for (uint32_t k = lane; k < spp; k += 64u) row[k] = p[k];
}

// Sampler::request2DArray arrays of one pixel (one lane per sampler slot, continuing its generate() stream):
// LowDiscrepancySampler::generate2D over all spp * size points (ldsampler.cpp:129-141,152-153) -- one 64-bit scramble
// and a shuffle of the point indices -- or latinHypercube(random, dest, spp * size, 2) (util.cpp:529-540,
// stratified.cpp:136-138)
__global__ void k_sample_arrays(DConfig cfg, uint32_t n_slots, const unsigned long long *state_in,
                                uint32_t *scr, uint16_t *perm, float2 *pts) {
	const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
	if (slot >= n_slots)
		return;
	uint64_t st = state_in[slot];
	for (int a = 0; a < cfg.arr_n; ++a) {
		const uint32_t n = cfg.spp * cfg.arr_size[a];
		if (cfg.sampler_kind == 1) {
			const uint64_t q = keyedNext(st);
			scr[((size_t) slot * cfg.arr_n + a) * 2 + 0] = (uint32_t) (q & 0xFFFFFFFFull);
			scr[((size_t) slot * cfg.arr_n + a) * 2 + 1] = (uint32_t) (q >> 32);
			uint16_t *p = perm + (size_t) slot * cfg.arr_total + cfg.arr_off[a];
			for (uint32_t k = 0; k < n; ++k) p[k] = (uint16_t) k;
			for (uint32_t it = n - 1; it > 0; --it) {
				const uint32_t other = (uint32_t) keyedNextSize(st, it);
				const uint16_t x = p[it], y = p[other];
				p[it] = y; p[other] = x;
			}
		} else {
			float *d = reinterpret_cast<float *>(pts + (size_t) slot * cfg.arr_total + cfg.arr_off[a]);
			const float delta = 1 / (float) n;
			for (uint32_t i = 0; i < n; ++i)
				for (uint32_t j = 0; j < 2; ++j)
					d[2 * i + j] = ((float) i + ulongToFloat(keyedNext(st))) * delta;
			for (uint32_t i = 0; i < 2; ++i)
				for (uint32_t j = 0; j < n; ++j) {
					const uint32_t other = (uint32_t) keyedNextSize(st, n);
					const float t = d[2 * j + i]; d[2 * j + i] = d[2 * other + i]; d[2 * other + i] = t;
				}
		}
	}
}

// The same tables with the permutations shuffled in LDS and written out in coalesced rows: the serial chain of a
// shuffle is ~2 dependent memory accesses per step, which LDS serves an order of magnitude faster than the L2.  One
// wave per workgroup; L = `lanes` of its lanes (a power of two, as many as fit: L * spp * 2 B <= 64 KB) shuffle one
// pixel each, all 64 lanes write the rows out.  Layout [entry][lane ^ entry mod L]: the XOR swizzle keeps both the
// per-lane shuffle accesses and the transposed write-out free of bank conflicts without padding.
__global__ __launch_bounds__(64) void k_ld_tables_lds(DConfig cfg, const uint32_t *pixel_keys, uint32_t n_slots, uint32_t lanes,
                                                      uint32_t *scr, uint16_t *perm, unsigned long long *state_out) {
	extern __shared__ uint16_t s_p[];
	const uint32_t L = lanes, lm = L - 1u;
	const uint32_t lane = threadIdx.x, slot0 = blockIdx.x * L, slot = slot0 + lane;
	const bool active = lane < L && slot < n_slots;
	const uint32_t spp = cfg.spp;
	const int depth = cfg.ld_depth;
	const bool ld = cfg.sampler_kind == 1;
	uint64_t st = active ? keyedInit(cfg.seed, pixel_keys[slot], 0) : 0ull;
	uint32_t *s = scr + (size_t) slot * 3 * depth;
	for (int arr = 0; arr < 2 * depth; ++arr) {
		if (active) {
			const int i = arr >> 1;
			if (ld) {
				if ((arr & 1) == 0) {
					s[i * 3 + 0] = (uint32_t) (keyedNext(st) & 0xFFFFFFFFull);
				} else {
					const uint64_t q = keyedNext(st);
					s[i * 3 + 1] = (uint32_t) (q & 0xFFFFFFFFull);
					s[i * 3 + 2] = (uint32_t) (q >> 32);
				}
			}
			for (uint32_t k = 0; k < spp; ++k) s_p[k * L + ((lane ^ k) & lm)] = (uint16_t) k;
			for (uint32_t it = spp - 1; it > 0; --it) {
				const uint32_t other = (uint32_t) keyedNextSize(st, it);
				const uint32_t ia = it * L + ((lane ^ it) & lm), ib = other * L + ((lane ^ other) & lm);
				const uint16_t a = s_p[ia], b = s_p[ib];
				s_p[ia] = b; s_p[ib] = a;
			}
		}
		__syncthreads();
		const uint32_t rows = (n_slots - slot0 < L) ? n_slots - slot0 : L;
		if (spp >= 64u) {
			for (uint32_t r = 0; r < rows; ++r) {
				uint16_t *dst = perm + ((size_t) (slot0 + r) * 2 * depth + arr) * spp;
				for (uint32_t k = lane; k < spp; k += 64u) dst[k] = s_p[k * L + ((r ^ k) & lm)];
			}
		} else {
			// short rows (a 1-spp frame: one entry per pixel): the lanes share out (row, entry) pairs instead of walking the
			// rows one by one with most of the wave idle
			for (uint32_t idx = lane; idx < rows * spp; idx += 64u) {
				const uint32_t r = idx / spp, k = idx - r * spp;
				perm[((size_t) (slot0 + r) * 2 * depth + arr) * spp + k] = s_p[k * L + ((r ^ k) & lm)];
			}
		}
		__syncthreads();
	}
	if (active && state_out) state_out[slot] = st;
}

// Sampler read-out for tests (mtsgpu_sampler_values): the sampler state k_generate creates for (pixel, sample j) in
// slot 0, then n draws
__global__ void k_sampler_values(DConfig cfg, uint32_t pixel_key, uint32_t j, uint32_t n, int two_d, float *out) {
	if (blockIdx.x != 0 || threadIdx.x != 0)
		return;
	PathSampler smp;
	smp.stream = keyedInit(cfg.seed, pixel_key, 1 + (uint64_t) j);
	smp.slot = 0; smp.j = j; smp.d1 = 0; smp.d2 = 0;
	for (uint32_t i = 0; i < n; ++i) {
		if (two_d) sampler_next2d(cfg, smp, out[2 * i], out[2 * i + 1]);
		else out[i] = sampler_next1d(cfg, smp);
	}
}

// ===========================================================================
// Random (src/libcore/random.cpp:99-227, include/mitsuba/core/random.h:82-148): the reference's MT19937-64 generator as
// it stands, on the device.  The samplers of this library draw from keyed streams instead (one sequential stream per
// worker cannot feed a wavefront); this generator is what a Mitsuba `Random` object is, one lane
// per object, and is pinned by the reference's known answers (mtsgpu_random_values, tests/test_gpu_round2.py).
// ===========================================================================
struct MtRandom { uint64_t mt[312]; int mti; };
__device__ inline void mt_seed(MtRandom &r, uint64_t s) {                     // random.cpp:99-103
	r.mt[0] = s;
	for (r.mti = 1; r.mti < 312; r.mti++)
		r.mt[r.mti] = 6364136223846793005ULL * (r.mt[r.mti - 1] ^ (r.mt[r.mti - 1] >> 62)) + (uint64_t) r.mti;
}
__device__ inline void mt_seed_array(MtRandom &r, const uint64_t *init_key, uint64_t key_length) {     // random.cpp:118-140
	uint64_t *mt = r.mt;
	mt_seed(r, 19650218ULL);
	uint64_t i = 1, j = 0, k = (312 > key_length ? 312 : key_length);
	for (; k; k--) {
		mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 62)) * 3935559000370003845ULL)) + init_key[j] + j;
		i++; j++;
		if (i >= 312) { mt[0] = mt[311]; i = 1; }
		if (j >= key_length) j = 0;
	}
	for (k = 311; k; k--) {
		mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 62)) * 2862933555777941757ULL)) - i;
		i++;
		if (i >= 312) { mt[0] = mt[311]; i = 1; }
	}
	mt[0] = 1ULL << 63;
}
__device__ inline uint64_t mt_next_ulong(MtRandom &r) {                       // random.cpp:143-178
	const uint64_t MATRIX_A = 0xB5026F5AA96619E9ULL, UM = 0xFFFFFFFF80000000ULL, LM = 0x7FFFFFFFULL;
	uint64_t *mt = r.mt;
	uint64_t x;
	if (r.mti >= 312) {
		if (r.mti == 313) mt_seed(r, 5489ULL);         // a default-constructed Random
		int i;
		for (i = 0; i < 312 - 156; i++) {
			x = (mt[i] & UM) | (mt[i + 1] & LM);
			mt[i] = mt[i + 156] ^ (x >> 1) ^ ((x & 1ULL) ? MATRIX_A : 0ULL);
		}
		for (; i < 311; i++) {
			x = (mt[i] & UM) | (mt[i + 1] & LM);
			mt[i] = mt[i + (156 - 312)] ^ (x >> 1) ^ ((x & 1ULL) ? MATRIX_A : 0ULL);
		}
		x = (mt[311] & UM) | (mt[0] & LM);
		mt[311] = mt[155] ^ (x >> 1) ^ ((x & 1ULL) ? MATRIX_A : 0ULL);
		r.mti = 0;
	}
	x = mt[r.mti++];
	x ^= (x >> 29) & 0x5555555555555555ULL;
	x ^= (x << 17) & 0x71D67FFFEDA60000ULL;
	x ^= (x << 37) & 0xFFF7EEE000000000ULL;
	x ^= (x >> 43);
	return x;
}
__device__ inline uint64_t mt_next_size(MtRandom &r, uint64_t n) {            // random.cpp:196-215: bit mask + rejection
	uint64_t bitmask = n;
	bitmask |= bitmask >> 1; bitmask |= bitmask >> 2; bitmask |= bitmask >> 4;
	bitmask |= bitmask >> 8; bitmask |= bitmask >> 16; bitmask |= bitmask >> 32;
	uint64_t result;
	while ((result = (mt_next_ulong(r) & bitmask)) >= n) { }
	return result;
}
// op 0: n x nextULong; 1: n x nextFloat (bit patterns); 2: n x nextSize(arg); 3: shuffle of 0 .. n-1 (random.h:145-148);
// seed == 0: a default-constructed Random, otherwise Random::seed(seed); clone > 0: the clone-th Random(Random *) copy
// of that generator (random.cpp:105-110: 312 draws from the parent, init_by_array), as the per-worker samplers are made
__global__ void k_random_values(MtRandom *state, int op, unsigned long long seed, unsigned long long arg, uint32_t clone, uint32_t n,
                                unsigned long long *out) {
	if (blockIdx.x != 0 || threadIdx.x != 0) return;
	MtRandom &r = state[0], &child = state[1];
	if (seed) mt_seed(r, seed); else r.mti = 313;
	MtRandom *g = &r;
	for (uint32_t c = 0; c < clone; ++c) {
		uint64_t *buf = reinterpret_cast<uint64_t *>(state + 2);
		for (int i = 0; i < 312; ++i) buf[i] = mt_next_ulong(r);
		mt_seed_array(child, buf, 312);
		g = &child;
	}
	if (op == 3) {
		for (uint32_t i = 0; i < n; ++i) out[i] = i;
		for (uint32_t it = n ? n - 1 : 0; it > 0; --it) {
			const uint64_t other = mt_next_size(*g, (uint64_t) it);
			const unsigned long long t = out[it]; out[it] = out[other]; out[other] = t;
		}
		return;
	}
	for (uint32_t i = 0; i < n; ++i) {
		if (op == 0) out[i] = mt_next_ulong(*g);
		else if (op == 1) out[i] = (unsigned long long) __float_as_uint(ulongToFloat(mt_next_ulong(*g)));
		else out[i] = mt_next_size(*g, arg);
	}
}
void launch_random_values(hipStream_t s, void *state, int op, unsigned long long seed, unsigned long long arg, uint32_t clone, uint32_t n,
                          unsigned long long *out) {
	hipLaunchKernelGGL(k_random_values, dim3(1), dim3(64), 0, s, reinterpret_cast<MtRandom *>(state), op, seed, arg, clone, n, out);
}
size_t random_state_bytes() { return 3 * sizeof(MtRandom); }

// The LDS kernel pays while all 64 lanes of a wave shuffle (up to 512 samples per pixel); above that the tables go through
// k_ld_scout + k_ld_apply and a scratch copy of ld_table_scratch_entries() entries
static bool ld_tables_sliced(uint32_t spp) { return (size_t) spp * 64 * sizeof(uint16_t) > 64 * 1024; }
constexpr uint32_t kApplyChunk = 256;
constexpr uint32_t kApplyLdsMaxSpp = 16384;          // k_ld_apply_lds: 8 KB of claims + 2 bytes per sample <= 40 KB per wave
size_t ld_table_scratch_entries(uint32_t n_slots, uint32_t spp, int depth) {
	if (!ld_tables_sliced(spp) || spp <= kApplyLdsMaxSpp) return 0;
	return (size_t) std::min<uint32_t>(kApplyChunk, blocks_for(n_slots, 64) * 2 * (uint32_t) depth) * 64 * spp;
}

// all tables of n_slots pixels; `state` (one word per slot) receives the pixels' streams after generate()
void launch_ld_tables(hipStream_t s, const DConfig &cfg, const uint32_t *pixel_keys, uint32_t n_slots,
                      uint32_t *scr, uint16_t *perm, unsigned long long *state, uint16_t *scratch) {
	if (!n_slots) return;
	if (!ld_tables_sliced(cfg.spp)) {
		const uint32_t lanes = 64;
		const size_t lds = (size_t) cfg.spp * lanes * sizeof(uint16_t);
		hipLaunchKernelGGL(k_ld_tables_lds, dim3(blocks_for(n_slots, lanes)), dim3(64), lds, s, cfg, pixel_keys, n_slots, lanes, scr, perm, state);
	} else {
		hipLaunchKernelGGL(k_ld_scout, dim3(blocks_for(n_slots, 4)), dim3(256), 0, s, cfg, pixel_keys, n_slots, scr, perm, state);
		if (cfg.spp <= kApplyLdsMaxSpp) {
			const uint32_t nTables = n_slots * 2 * (uint32_t) cfg.ld_depth;
			hipLaunchKernelGGL(k_ld_apply_lds, dim3(nTables), dim3(64), kClaimSlots * sizeof(uint32_t) + cfg.spp * sizeof(uint16_t), s, cfg, nTables, perm);
			return;
		}
		// kApplyChunk (group, table) pairs per launch: the scratch of a launch (128 MB at 4096 spp) stays in the Infinity Cache.
		// All 1 728 pairs of a C4 pass at once miss it on every swap (15.5 ms per pass); launches of 512 / 256 / 128 / 64
		// pairs take 13.0 / 11.5 / 14.0 / 20 ms -- below 256 the chip runs out of lanes (profiles/r05m_*)
		const uint32_t total = blocks_for(n_slots, 64) * 2 * cfg.ld_depth;
		for (uint32_t b0 = 0; b0 < total; b0 += kApplyChunk)
			hipLaunchKernelGGL(k_ld_apply, dim3(std::min(kApplyChunk, total - b0)), dim3(64), 0, s, cfg, n_slots, b0, perm, scratch);
	}
}

void launch_sample_arrays(hipStream_t s, const DConfig &cfg, uint32_t n_slots, const unsigned long long *state_in) {
	if (!n_slots || cfg.arr_n == 0) return;
	hipLaunchKernelGGL(k_sample_arrays, dim3(blocks_for(n_slots, 64)), dim3(64), 0, s, cfg, n_slots, state_in,
	                   const_cast<uint32_t *>(cfg.arr_scr), const_cast<uint16_t *>(cfg.arr_perm), const_cast<float2 *>(cfg.arr_pts));
}

void launch_sampler_values(hipStream_t s, const DConfig &cfg, uint32_t pixel_key, uint32_t j, uint32_t n, int two_d, float *out) {
	hipLaunchKernelGGL(k_sampler_values, dim3(1), dim3(64), 0, s, cfg, pixel_key, j, n, two_d, out);
}

} // namespace mg
