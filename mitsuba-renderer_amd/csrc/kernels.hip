// kernels.hip -- gfx950 kernels of the wavefront path tracer.
//
// One path per lane; paths never move in memory, per-bounce queues of path ids
// are compacted and material-sorted with wave ballots + prefix popcounts.
// Stage order per bounce (MIPathTracer::Li, src/integrators/path/path.cpp:47-216):
//   k_trace<closest>  ->  k_shade<bsdf type>  ->  k_trace<shadow>  ->  (next bounce)
#include "kernels.h"
#include "devmath.h"
#include <algorithm>

namespace mg {

// ===========================================================================
// small helpers
// ===========================================================================
__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }

// Wave-aggregated append: every lane with pred gets a unique slot; one atomic per wave.
__device__ __forceinline__ uint32_t wave_append(bool pred, uint32_t *counter) {
	const unsigned long long mask = __ballot(pred);
	if (mask == 0ull)
		return 0u;
	const uint32_t lane = lane_id();
	const uint32_t rank = (uint32_t) __popcll(mask & ((1ull << lane) - 1ull));
	const int leader = __ffsll((long long) mask) - 1;
	uint32_t base = 0;
	if ((int) lane == leader)
		base = atomicAdd(counter, (uint32_t) __popcll(mask));
	base = __shfl(base, leader);
	return base + rank;
}

__device__ __forceinline__ float sel3(float x, float y, float z, int axis) {
	return axis == 0 ? x : (axis == 1 ? y : z);
}

// MG_NT: non-temporal hints on data that is touched once per launch -- bit 0: path records, ray
// and id queues in k_trace; bit 1: leaf records in k_trace; bit 2: records and queues in k_shade -- so that they do not
// push the tree out of the L1 / L2
#ifndef MG_NT
#define MG_NT 4      // measured (64-spp C3 frame): bit 0 +12 ms, bit 1 +120 ms (the leaf records live in the L2), bit 2 -3.6 ms
#endif
typedef uint32_t nt_u4 __attribute__((ext_vector_type(4)));
template <int BIT, typename T> __device__ __forceinline__ T ld_stream(const T *p) {
	if (MG_NT & BIT) {
		static_assert(sizeof(T) == 16 || sizeof(T) == 4, "16-byte or 4-byte objects");
		T out;
		if (sizeof(T) == 16) { const nt_u4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u4 *>(p)); __builtin_memcpy(&out, &v, 16); }
		else { const uint32_t v = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(p)); __builtin_memcpy(&out, &v, 4); }
		return out;
	}
	return *p;
}
template <int BIT, typename T> __device__ __forceinline__ void st_stream(T *p, const T &v) {
	if (MG_NT & BIT) {
		static_assert(sizeof(T) == 16 || sizeof(T) == 4, "16-byte or 4-byte objects");
		if (sizeof(T) == 16) { nt_u4 x; __builtin_memcpy(&x, &v, 16); __builtin_nontemporal_store(x, reinterpret_cast<nt_u4 *>(p)); }
		else { uint32_t x; __builtin_memcpy(&x, &v, 4); __builtin_nontemporal_store(x, reinterpret_cast<uint32_t *>(p)); }
	} else {
		*p = v;
	}
}

__global__ void k_fill_u32(uint32_t *p, uint32_t v, size_t n) {
	size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) p[i] = v;
}
__global__ void k_iota(uint32_t *p, uint32_t n) {
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) p[i] = i;
}

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

// ===========================================================================
// Sampler::next1D / next2D (independent.cpp:72-81, ldsampler.cpp:172-186)
// ===========================================================================
struct PathSampler {
	uint64_t stream;
	uint32_t slot, j;
	uint32_t d1, d2;
};

// HaltonSequence::nextValue (halton.cpp:73-75) / HammersleySequence::nextValue (hammersley.cpp:75-82);
// m_sampleDepth is kept in d1 (low byte) and d2 (high byte)
__device__ __forceinline__ float qmc_next_value(const DConfig &cfg, PathSampler &s) {
	const uint32_t depth = s.d1 | (s.d2 << 8);
	const uint32_t next = depth + 1u;
	s.d1 = next & 0xFFu; s.d2 = (next >> 8) & 0xFFu;
	if (cfg.sampler_kind == 3) {
		if (depth == 0u)
			return s.j * (1.0f / cfg.spp);
		return radicalInverse((int) cfg.primes[min(depth - 1u, 999u)], (uint64_t) s.j);
	}
	return radicalInverse((int) cfg.primes[min(depth, 999u)], (uint64_t) s.j);
}

__device__ __forceinline__ float sampler_next1d(const DConfig &cfg, PathSampler &s) {
	if (cfg.sampler_kind == 4) {
		// StratifiedSampler::next1D (stratified.cpp:155-163)
		if ((int) s.d1 < cfg.ld_depth) {
			const int i = (int) s.d1++;
			const int k = (int) cfg.ld_perm[((size_t) s.slot * 2 * cfg.ld_depth + 2 * i) * cfg.spp + s.j];
			return (k + ulongToFloat(keyedNext(s.stream))) * (1 / (float) cfg.spp);
		}
		return ulongToFloat(keyedNext(s.stream));
	}
	if (cfg.sampler_kind >= 2)
		return qmc_next_value(cfg, s);
	if (cfg.sampler_kind == 1 && (int) s.d1 < cfg.ld_depth) {
		const int i = (int) s.d1++;
		const uint32_t k = cfg.ld_perm[((size_t) s.slot * 2 * cfg.ld_depth + 2 * i) * cfg.spp + s.j];
		return u32ToUnit(vdcBits(k, cfg.ld_scr[(size_t) s.slot * 3 * cfg.ld_depth + i * 3 + 0]));
	}
	return ulongToFloat(keyedNext(s.stream));
}

__device__ __forceinline__ void sampler_next2d(const DConfig &cfg, PathSampler &s, float &x, float &y) {
	if (cfg.sampler_kind == 4) {
		// StratifiedSampler::next2D (stratified.cpp:165-181); x is drawn first
		if ((int) s.d2 < cfg.ld_depth) {
			const int i = (int) s.d2++;
			const int k = (int) cfg.ld_perm[((size_t) s.slot * 2 * cfg.ld_depth + 2 * i + 1) * cfg.spp + s.j];
			const int kx = k % cfg.strat_res, ky = k / cfg.strat_res;
			const float invResolution = 1 / (float) cfg.strat_res;
			const float jx = ulongToFloat(keyedNext(s.stream)), jy = ulongToFloat(keyedNext(s.stream));
			x = (kx + jx) * invResolution; y = (ky + jy) * invResolution;
			return;
		}
		x = ulongToFloat(keyedNext(s.stream));
		y = ulongToFloat(keyedNext(s.stream));
		return;
	}
	if (cfg.sampler_kind >= 2) {
		x = qmc_next_value(cfg, s);
		y = qmc_next_value(cfg, s);
		return;
	}
	if (cfg.sampler_kind == 1 && (int) s.d2 < cfg.ld_depth) {
		const int i = (int) s.d2++;
		const uint32_t k = cfg.ld_perm[((size_t) s.slot * 2 * cfg.ld_depth + 2 * i + 1) * cfg.spp + s.j];
		const uint32_t *scr = cfg.ld_scr + (size_t) s.slot * 3 * cfg.ld_depth + i * 3;
		x = u32ToUnit(vdcBits(k, scr[1]));
		y = u32ToUnit(sobol2Bits(k, scr[2]));
		return;
	}
	// x first, then y (independent.cpp:76-81)
	x = ulongToFloat(keyedNext(s.stream));
	y = ulongToFloat(keyedNext(s.stream));
}

// Sampler::next2DArray (sampler.cpp:76-87): point k of the array `a` of camera sample s.j.  The keyed independent
// sampler fills its arrays from the pixel's generate() stream (independent.cpp:63-66), which is counter-based, so the
// point is computed in place; the other two read the tables of k_sample_arrays
__device__ __forceinline__ void sampler_array2d(const DConfig &cfg, const PathSampler &s, uint32_t pixelKey, int a, uint32_t k,
                                                float &x, float &y) {
	const size_t e = (size_t) cfg.arr_off[a] + (size_t) s.j * cfg.arr_size[a] + k;
	if (cfg.sampler_kind == 0) {
		const uint64_t st0 = keyedInit(cfg.seed, pixelKey, 0);
		x = ulongToFloat(sm64mix(st0 + 0x9E3779B97F4A7C15ULL * (uint64_t) (2 * e + 1)));
		y = ulongToFloat(sm64mix(st0 + 0x9E3779B97F4A7C15ULL * (uint64_t) (2 * e + 2)));
	} else if (cfg.sampler_kind == 1) {
		const uint32_t idx = cfg.arr_perm[(size_t) s.slot * cfg.arr_total + e];
		const uint32_t *scr = cfg.arr_scr + ((size_t) s.slot * cfg.arr_n + a) * 2;
		x = u32ToUnit(vdcBits(idx, scr[0]));
		y = u32ToUnit(sobol2Bits(idx, scr[1]));
	} else {
		const float2 v = cfg.arr_pts[(size_t) s.slot * cfg.arr_total + e];
		x = v.x; y = v.y;
	}
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

// ===========================================================================
// K1: camera samples (integrator.cpp:154-166, perspective.cpp:77-112)
// ===========================================================================
// The records leave through LDS: a lane that stores its own record slot by slot touches 64 different lines with every store
// instruction (eight instructions, 512 line requests per wave, every line written in eight pieces); instead eight lanes
// write one record together -- whole 128-byte lines, eight records per instruction -- as k_shade does.
constexpr int kGenBlock = 256;
__global__ __launch_bounds__(kGenBlock) void k_generate(DScene sc, DPaths ps, DConfig cfg, const uint32_t *pixel_list, uint32_t n_slots,
                                                        const uint32_t *explicit_samples, uint32_t n_paths, uint32_t *queue) {
	__shared__ float4 s_rec[kGenBlock / 64][64 * (kPathSlots + 1)];      // rows of 9 float4: conflict-free both ways
	const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
	float4 *rows = s_rec[threadIdx.x >> 6];
	float4 *row = rows + lane_id() * (kPathSlots + 1);
	if (id < n_paths) {
	uint32_t slot, j, pixel;
	if (explicit_samples) {
		// film pixel (x, y) of the crop window -> key in the full film's raster grid
		slot = id;
		pixel = (explicit_samples[3 * (size_t) id + 1] + (uint32_t) cfg.crop_y) * (uint32_t) cfg.pix_w
		      + explicit_samples[3 * (size_t) id] + (uint32_t) cfg.crop_x;
		j = explicit_samples[3 * (size_t) id + 2];
	} else {
		slot = id / cfg.spp;
		j = id - slot * cfg.spp;
		pixel = pixel_list[slot];
	}
	// raster pixel of the key; negative / beyond the film with highQualityEdges (renderproc.cpp:146-153)
	const int px = (int) (pixel % (uint32_t) cfg.pix_w) + cfg.pix_off, py = (int) (pixel / (uint32_t) cfg.pix_w) + cfg.pix_off;

	PathSampler smp;
	smp.stream = keyedInit(cfg.seed, pixel, 1 + (uint64_t) j);
	smp.slot = slot; smp.j = j; smp.d1 = 0; smp.d2 = 0;
	float sx, sy, lensX = 0, lensY = 0;
	if (cfg.aperture_radius > 0.0f && cfg.camera_kind == 0) sampler_next2d(cfg, smp, lensX, lensY);     // needsLensSample (integrator.cpp:156-157)
	sampler_next2d(cfg, smp, sx, sy);
	sx += (float) px; sy += (float) py;

	// m_rasterToCamera(Point(sx, sy, 0)) with the homogeneous divide (transform.h:133-149)
	const float *m = cfg.r2c;
	float ix = m[0] * sx + m[1] * sy + m[2] * 0.0f + m[3];
	float iy = m[4] * sx + m[5] * sy + m[6] * 0.0f + m[7];
	float iz = m[8] * sx + m[9] * sy + m[10] * 0.0f + m[11];
	float iw = m[12] * sx + m[13] * sy + m[14] * 0.0f + m[15];
	V3 ic(ix, iy, iz);
	if (iw != 1.0f)
		ic = divs(ic, iw);
	const bool ortho = cfg.camera_kind == 1;
	V3 lo(0.0f, 0.0f, 0.0f);
	if (ortho)
		lo = ic;                                   // OrthographicCamera::generateRay (orthographic.cpp:104-118)
	else if (cfg.aperture_radius > 0.0f) {
		// perspective.cpp:90-103: sample the aperture, aim at the focal plane
		float lpx, lpy;
		squareToDiskConcentric(lensX, lensY, lpx, lpy);
		lpx *= cfg.aperture_radius; lpy *= cfg.aperture_radius;
		const float tf = cfg.focus_depth / ic.z;
		const V3 itsFocal(0.0f + tf * ic.x, 0.0f + tf * ic.y, 0.0f + tf * ic.z);
		lo.x += lpx;
		lo.y += lpy;
		ic = itsFocal - lo;
	}
	V3 ld = ortho ? V3(0.0f, 0.0f, 1.0f) : normalize(ic);
	float invZ = 1.0f / ld.z;
	float mint = cfg.near_clip * invZ, maxt = cfg.far_clip * invZ;
	if (ortho) { mint = 0; maxt = cfg.far_clip - cfg.near_clip; }
	// m_cameraToWorld(localRay, ray) (transform.h:219-235)
	const float *w = cfg.c2w;
	V3 o(w[0] * lo.x + w[1] * lo.y + w[2] * lo.z + w[3],
	     w[4] * lo.x + w[5] * lo.y + w[6] * lo.z + w[7],
	     w[8] * lo.x + w[9] * lo.y + w[10] * lo.z + w[11]);
	float ow = w[12] * lo.x + w[13] * lo.y + w[14] * lo.z + w[15];
	if (ow != 1.0f)
		o = divs(o, ow);
	V3 d(w[0] * ld.x + w[1] * ld.y + w[2] * ld.z,
	     w[4] * ld.x + w[5] * ld.y + w[6] * ld.z,
	     w[8] * ld.x + w[9] * ld.y + w[10] * ld.z);

	row[0] = make_float4(o.x, o.y, o.z, mint);                         // ray_o
	row[1] = make_float4(d.x, d.y, d.z, maxt);                         // ray_d
	row[2] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(kNoPrim));  // hit: none yet
	row[3] = make_float4(1.0f, 1.0f, 1.0f, __int_as_float(1));        // thr; depth = 1 (integrator.h:186-191)
	const uint32_t flags = F_EMITTED | F_FIRST | (smp.d1 << F_D1_SHIFT) | (smp.d2 << F_D2_SHIFT);
	row[4] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(flags));    // Li
	row[5] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);                      // bsdf
	reinterpret_cast<uint4 &>(row[6]) = make_uint4((uint32_t) (smp.stream & 0xFFFFFFFFull), (uint32_t) (smp.stream >> 32), j, pixel);   // misc
	row[7] = make_float4(sx, sy, 0.0f, 0.0f);                          // spos
	queue[id] = id;
	if (ps.rqn_o) { st_stream<4>(&ps.rqn_o[id], row[0]); st_stream<4>(&ps.rqn_d[id], row[1]); }      // the camera rays in queue order
	}
	// program order suffices inside a wave (every row is written and read by the same wave)
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
	const uint32_t sub = lane_id() & 7u, grp = lane_id() >> 3;
	const uint32_t wave_first = id - lane_id();                         // path id of lane 0 of this wave
	#pragma unroll
	for (int r = 0; r < 8; ++r) {
		const uint32_t src = grp + 8u * r;
		if (wave_first + src < n_paths)
			st_stream<4>(&ps.base[(size_t) (wave_first + src) * kPathSlots + sub], rows[src * (kPathSlots + 1) + sub]);
	}
}


// ===========================================================================
// Sphere shape (src/shapes/sphere.cpp).  SP = shape parameter block:
// [0..2] centre [3] radius [4] inverted [5..13] objectToWorld 3x3 [14..22] worldToObject 3x3 [23] 1/area
// ===========================================================================
// solveQuadratic (src/libcore/util.cpp:450-488)
__device__ __forceinline__ bool solve_quadratic(float a, float b, float c, float &x0, float &x1) {
	if (a == 0) {
		if (b != 0) { x0 = x1 = -c / b; return true; }
		return false;
	}
	const float discrim = b * b - 4.0f * a * c;
	if (discrim < 0)
		return false;
	const float sqrtDiscrim = sqrtf(discrim);
	float temp;
	if (b < 0) temp = -0.5f * (b - sqrtDiscrim);
	else       temp = -0.5f * (b + sqrtDiscrim);
	x0 = temp / a;
	x1 = c / temp;
	if (x0 > x1) { const float t = x0; x0 = x1; x1 = t; }
	return true;
}
// the quadratic of Sphere::rayIntersect (sphere.cpp:94-101)
__device__ __forceinline__ bool sphere_roots(V3 center, float radius, V3 ro, V3 rd, float &nearT, float &farT) {
	const V3 o = ro - center;
	const float A = rd.x * rd.x + rd.y * rd.y + rd.z * rd.z;
	const float B = 2 * (rd.x * o.x + rd.y * o.y + rd.z * o.z);
	const float C = o.x * o.x + o.y * o.y + o.z * o.z - radius * radius;
	return solve_quadratic(A, B, C, nearT, farT);
}
// Sphere::rayIntersect(ray, mint, maxt, t, tmp) (sphere.cpp:94-116)
__device__ __forceinline__ bool sphere_intersect(V3 center, float radius, V3 ro, V3 rd, float mint, float maxt, float &t) {
	float nearT, farT;
	if (!sphere_roots(center, radius, ro, rd, nearT, farT))
		return false;
	if (nearT > maxt || farT < mint)
		return false;
	if (nearT < mint) {
		if (farT > maxt)
			return false;
		t = farT;
	} else {
		t = nearT;
	}
	return true;
}
// Sphere::rayIntersect(ray, mint, maxt) (sphere.cpp:118-134)
__device__ __forceinline__ bool sphere_occludes(V3 center, float radius, V3 ro, V3 rd, float mint, float maxt) {
	float nearT, farT;
	if (!sphere_roots(center, radius, ro, rd, nearT, farT))
		return false;
	if (nearT > maxt || farT < mint)
		return false;
	if (nearT < mint && farT > maxt)
		return false;
	return true;
}

// ===========================================================================
// K2/K4: kd-tree traversal.
// ShapeKDTree::rayIntersect (src/librender/skdtree.cpp:108-132, :180-199) +
// rayIntersectHavran<shadow> (include/mitsuba/render/sahkdtree3.h:170-300) +
// TriAccel::rayIntersect (include/mitsuba/render/triaccel.h:98-159).
//
// Havran's stack of exit points is a LIFO once the entry point is kept in
// registers.  An exit point is (far child, t, axis, split)
// and all four are functions of (parent node, ray), so the stack stores ONE dword
// per level: parent index * 2 + "far child is the right one".  The top exit
// point lives in registers; deeper levels sit in LDS ([level][lane], conflict
// free), levels beyond kStackLDS spill to HBM.  The 8-entry hashed mailbox
// (sahkdtree3.h:130-144) is kept (LDS) because it decides equal-t ties.
// ===========================================================================
#ifndef MG_STACK_LDS
#define MG_STACK_LDS 10
#endif
#ifndef MG_TOP_PAIRS
#define MG_TOP_PAIRS (MG_TRACE_BLOCK >= 512 ? 1024 : 128)
#endif
constexpr int kStackLDS = MG_STACK_LDS;       // stack levels kept in LDS (deeper ones spill: 1 push in 10^4 at 12 levels on C3)
// The first 2 * kTopPairs device nodes -- the root and the sibling pairs below it in breadth-first order, see
// mtsgpu_upload_scene -- are copied into LDS by every workgroup: each ray's descent from the root starts with 8-9
// levels that every other ray visits too, and a vector-memory request costs the CU ~0.5-1 ns per lane where an LDS
// read costs ~0.05 (profiles/r02_ta_gather_microbench.txt).  0 switches the cache off.
constexpr uint32_t kTopPairs = MG_TOP_PAIRS;
constexpr int kSpillLevels = 50 - kStackLDS;      // LDS + spill levels = MTS_KD_MAXDEPTH (48, gkdtree.h:35) + 2
static_assert(kStackLDS >= 1 && kStackLDS + kSpillLevels >= 48 + 2, "the traversal stack must hold every tree the reference can build");
constexpr uint32_t kSentinel = 0xFFFFFFFFu;
constexpr uint32_t kNullNode = 0xFFFFFFFFu;

size_t trace_spill_levels() { return kSpillLevels; }      // in dwords per thread
size_t trace_stack_levels() { return kStackLDS + kSpillLevels; }
uint32_t trace_top_nodes() { return 2u * kTopPairs; }

// Persistent waves: the grid is sized to fill the chip once and every wave walks its own 64-ray
// batches of the queue with a private cursor (no work-queue atomic: a single head word saturates at
// ~88 fetches/us, MI355X_MICROARCH.md "dequeue").  The kernel is bound by instruction issue under
// SIMD divergence (measured lane utilisation in DESIGN.md section 6), which three scheduling rules
// attack: idle lanes are refilled from the wave's next batch once q.refill_min of them are idle, and
// the descent / primitive loops are left as soon as fewer than q.desc_min / q.leaf_min lanes still
// need them (the others stop waiting; stragglers resume in the next round).  None of this changes
// what is computed for a ray.
// `first` is the queue index of the wave's first batch, `stride` the distance to its next one, `static_n` the
// statically dealt prefix of the queue.
// leaf record e (48 bytes at a stride of kLeafStride x 16): its head (k | flags | primitive, n_u, n_v, n_d) and the two halves of its tail
__device__ __forceinline__ const uint4 *leaf_head(const DTraceScene &sc, uint32_t e) { return &sc.leaf_ta[kLeafStride * (size_t) e]; }
__device__ __forceinline__ const uint4 *leaf_tail(const DTraceScene &sc, uint32_t e, uint32_t half) { return &sc.leaf_ta[kLeafStride * (size_t) e + 1 + half]; }

template <int MODE, bool COUNT, bool BIN>
__device__ __forceinline__ void trace_body(const DTraceScene &sc, const DPaths &ps, const DQueues &q, const TracePlan &plan,
                                           const uint32_t *queue, uint32_t n, const uint32_t first, const uint32_t stride,
                                           uint32_t (*s_stack)[kTraceBlock], uint32_t (*s_mbox)[kTraceBlock], const uint4 *s_top) {
	// node fetches: from the LDS copy of the top of the tree when the index lies inside it
	// COUNT: the requests this lane issued (global / served by the LDS copy) and, when q.rec is set, the list of them
	uint32_t g_pair = 0, l_pair = 0, g_node = 0, l_node = 0, g_tail = 0, g_spill = 0, g_head = 0;
	uint32_t rec_n = 0, rec_slot = 0;
	auto rec_add = [&](uint32_t kind, uint32_t idx) {
		if (COUNT && q.rec) {
			if (rec_n < q.rec_cap) q.rec[(size_t) rec_slot * q.rec_cap + rec_n] = (kind << 29) | idx;
			rec_n++;
		}
	};
	auto load_node = [&](uint32_t i) -> uint2 {
		if (kTopPairs && i < 2u * kTopPairs) { if (COUNT) l_node++; return reinterpret_cast<const uint2 *>(s_top)[i]; }
		if (COUNT) { g_node++; rec_add(kReqNode, i); }
		return sc.nodes[i];
	};
	auto load_pair = [&](uint32_t left) -> uint4 {
		if (kTopPairs && left < 2u * kTopPairs) { if (COUNT) l_pair++; return s_top[left >> 1]; }
		if (COUNT) { g_pair++; rec_add(kReqPair, left >> 1); }
		return reinterpret_cast<const uint4 *>(sc.nodes)[left >> 1];
	};
	// The hashed mailbox decides which of two primitives with equal t is reported (sahkdtree3.h:130-144, :278-283), so
	// closest-hit rays keep it.  For any-hit rays it only saves repeated tests of a primitive that spans several leaves --
	// the answer is a disjunction over the same primitives either way -- and its 8 dwords per lane are better spent on
	// the LDS copy of the tree: the shadow kernels run without it (counting builds keep it: the oracle counts with it).
	constexpr bool kMbox = MODE == 0 || COUNT;
	const uint32_t tid = threadIdx.x;
	const uint32_t gtid = blockIdx.x * kTraceBlock + tid;     // spill slot of this lane
	const uint32_t lane = lane_id();
	const uint32_t shard = blockIdx.x % kBinShards;     // contention on a bin counter is spread over kBinShards words

	uint32_t c_inner = 0, c_leaf = 0, c_idx = 0, c_tri = 0;
	// COUNT: candidates whose record tail was fetched although their plane distance lies outside the interval the ray spends
	// in the leaf being visited (the reference tests against the ray's whole interval, sahkdtree3.h:262-288); c_ten = stack[enPt].t
	uint32_t c_tail_out = 0; float c_ten = 0;
	uint32_t w_inner = 0, w_leaf = 0, w_outer = 0, w_batch = 0;   // COUNT: lane slots issued per loop (64 per wave iteration)
#define MG_WSLOT(w) do { if (COUNT && lane == (uint32_t) __builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) (w) += 64u; } while (0)

	// Ray supply of a wave.  The first plan.static_n queue entries are dealt out statically: the wave owns the batches
	// (64 consecutive entries) wave_id, wave_id + n_waves, ... and walks them without any atomic.  The rest of the
	// queue (about a quarter) is claimed batch by batch through ONE counter once the static share is used up, which
	// evens out the finishing times of the waves (measured: the mean wave used to live 0.90-0.94 of the kernel).
	// A lane whose ray has finished stays idle until at least q.refill_min lanes of the wave are idle; then the
	// finished rays are retired together (one hit store + one binning step) and the idle lanes take new rays.
	// A launch with fewer rays than the chip has lanes runs B < 64 rays per wave (lanes >= B stay idle): a wave lasts
	// as long as its slowest ray, and with the wave slots to spare narrower batches shorten that critical path.
	const uint32_t B = plan.batch, static_n = plan.static_n;
	const uint64_t limitMask = (B >= 64u) ? ~0ull : ((1ull << B) - 1ull);
	uint32_t next_static = first;           // queue index of this wave's next static batch (uniform)
	uint32_t sup_base = 0, sup_left = 0;    // the chunk being handed out: queue[sup_base .. sup_base + sup_left)
	bool dyn_done = static_n >= n;          // nothing (left) to claim dynamically
	const uint32_t refill_min = plan.refill_min, desc_min = plan.desc_min, leaf_min = plan.leaf_min;

	uint32_t id = 0;
	float ox = 0, oy = 0, oz = 0, dx = 1, dy = 1, dz = 1, rx = 1, ry = 1, rz = 1;
	float mint = 0, maxt = 0, tmax0 = 0;
	float enx = 0, eny = 0, enz = 0, exx = 0, exy = 0, exz = 0, ex_t = 0;   // stack[enPt].p, stack[exPt].p, stack[exPt].t
	int sp = 0;
	// the current exit point: ex_node = its far child, ex_ref = its stack word (parent index * 2 + "far child is the right one")
	uint32_t ex_node = kNullNode, ex_ref = kSentinel, cur = 0;
	float best_t = MG_INF, best_u = 0, best_v = 0;
	uint32_t best_prim = kNoPrim, best_shape = 0;
	// best_shape: shape index of the accepted hit (dword 10 of its record), read while the record is at hand
	uint32_t e_cont = kNoPrim;              // position inside an interrupted leaf
	uint2 nd = make_uint2(0u, 0u);          // sc.nodes[cur], fetched as soon as cur is known
	bool found = false;
	bool has = false;                       // this lane is traversing a ray
	bool done = false;                      // this lane holds a finished ray that has not been retired yet

	while (true) {
		const uint64_t liveMask = __builtin_amdgcn_ballot_w64(has);
		const uint32_t nlive = (uint32_t) __popcll(liveMask);
		const bool wantRays = nlive == 0u || B - nlive >= refill_min;
		if (wantRays && sup_left == 0u) {
			// next chunk: a batch of the static share, or one claimed from the shared tail of the queue
			if (next_static < static_n) {
				sup_base = next_static; sup_left = (static_n - next_static < B) ? static_n - next_static : B; next_static += stride;
			} else if (!dyn_done) {
				uint32_t b = 0;
				if (lane == 0) b = atomicAdd(&q.counters[(MODE == 0 ? kCntDynClosest : kCntDynShadow) * kCounterStride], 1u);
				b = (uint32_t) __builtin_amdgcn_readfirstlane((int) b);
				const unsigned long long base = (unsigned long long) static_n + (unsigned long long) B * b;
				if (base < n) { sup_base = (uint32_t) base; sup_left = (n - sup_base < B) ? n - sup_base : B; }
				else dyn_done = true;
			}
		}
		const uint32_t remaining = sup_left;
		if (nlive == 0u || (remaining != 0u && wantRays)) {
			// ---- retire the finished rays (all lanes take part in the ballots) ----
			if (MODE == 0) {
				int bin = -1;
				constexpr bool hitsToBins = BIN;      // a binned path's hit travels with its id (DQueues::bin_hits)
				if (done) {
					if (!hitsToBins) st_stream<1>(&ps.hit(id), make_uint4(__float_as_uint(best_t), __float_as_uint(best_u), __float_as_uint(best_v), best_prim));
					if (BIN) {
						bin = kNumBins - 1;
						if (found) {
							bin = (int) sc.shape_bin[best_shape];      // BSDF type of the hit shape, or the terminal bin (one lookup)
						}
					}
				}
				if (BIN) {
					// material sort: one ballot + prefix popcount per bin; lane b reserves the slots of bin b,
					// so the wave issues ONE returning atomic instruction for all bins
					uint32_t cnt = 0, rank = 0;
					#pragma unroll
					for (int b = 0; b < kNumBins; ++b) {
						const uint64_t m = __builtin_amdgcn_ballot_w64(bin == b);
						if (lane == (uint32_t) b) cnt = (uint32_t) __popcll(m);
						if (bin == b) rank = (uint32_t) __popcll(m & ((1ull << lane) - 1ull));
					}
					uint32_t base = 0;
					if (lane < (uint32_t) kNumBins && cnt != 0u)
						base = atomicAdd(&q.counters[(lane * kBinShards + shard) * kCounterStride], cnt);
					base = __shfl(base, bin < 0 ? 0 : bin);
					// a shard's segment holds its share of a statically dealt queue (api.cpp: ensurePaths); dynamic claims
					// can in principle exceed it: the entry is dropped then, the counter still counts it, and the host
					// repeats the launch with static dealing when it sees a count above the capacity
					const uint32_t pos = base + rank;
					if (bin >= 0 && pos < q.bin_seg_cap) {
						const size_t at = (size_t) bin * q.bin_stride + (size_t) shard * q.bin_seg_cap + pos;
						st_stream<1>(&q.bins_base[at], id);
						if (hitsToBins) st_stream<1>(&q.bin_hits[at], make_uint4(__float_as_uint(best_t), __float_as_uint(best_u), __float_as_uint(best_v), best_prim));
					}
				}
			} else if (MODE == 1) {
				// Scene::sampleLuminaire's visibility test passed: add the pending contribution (path.cpp:124)
				if (done && !found) {
					const float4 c = ps.shq_nee[id];            // id = position in the shadow queue; c.w = path id
					const uint32_t pid = __float_as_uint(c.w);
					if (q.nee_parked) {
						// parked in the record (DQueues::nee_parked): its next reader adds it
						ps.slot(pid, 2) = make_float4(c.x, c.y, c.z, __uint_as_float(kNeeTag));
					} else {
						float4 L = ps.Li(pid);
						L.x += c.x; L.y += c.y; L.z += c.z;
						ps.Li(pid) = L;
					}
				}
			} else {
				if (done)
					ps.hit(id) = make_uint4(0u, 0u, 0u, found ? 1u : 0u);
			}
			done = false;
			if (remaining == 0u)
				break;          // nlive == 0 and nothing left: the wave is finished

			// ---- refill: idle lane number r takes ray r of the current chunk ----
			const uint32_t r = (uint32_t) __popcll(~liveMask & limitMask & ((1ull << lane) - 1ull));
			const bool take = !has && lane < B && r < remaining;
			const uint32_t taken = (B - nlive < remaining) ? B - nlive : remaining;
			const uint32_t my = sup_base + r;
			sup_base += taken; sup_left -= taken;
			MG_WSLOT(w_batch);
			if (take) {
				id = (MODE == 1) ? my : ld_stream<1>(&queue[my]);       // shadow rays are addressed by their queue position
				// (rays that arrive in queue order are streamed, like shadow rays: not part of the recorded request list)
				if (COUNT && q.rec) { rec_slot = my; rec_n = 0; if (MODE != 1 && !(MODE == 0 && ps.rq_o)) { rec_add(kReqRay, id * kPathSlots); rec_add(kReqRay, id * kPathSlots + 1); } }
				float4 a, b;
				float rmint, rmaxt;
				if (MODE == 1) {
					a = ld_stream<1>(&ps.shq_o[my]); b = ld_stream<1>(&ps.shq_d[my]);
					rmint = kShadowEpsilon; rmaxt = 1 - kShadowEpsilon;      // Scene::isOccluded, scene.h:241-246
				} else {
					if (MODE == 0 && ps.rq_o) { a = ld_stream<1>(&ps.rq_o[my]); b = ld_stream<1>(&ps.rq_d[my]); }    // in queue order: no trip behind the id
					else { a = ld_stream<1>(&ps.ray_o(id)); b = ld_stream<1>(&ps.ray_d(id)); }
					rmint = a.w; rmaxt = b.w;
				}
				ox = a.x; oy = a.y; oz = a.z; dx = b.x; dy = b.y; dz = b.z;
				rx = 1.0f / dx; ry = 1.0f / dy; rz = 1.0f / dz;          // Ray::dRcp (ray.h:63-74)
				// AABB::rayIntersect (aabb.h:349-382) + adaptive epsilon (skdtree.cpp:114-122)
				bool go = true;
				mint = -MG_INF; maxt = MG_INF;
				#pragma unroll
				for (int i = 0; i < 3; ++i) {
					const float direction = sel3(dx, dy, dz, i), origin = sel3(ox, oy, oz, i);
					const float minVal = sc.aabb_min[i], maxVal = sc.aabb_max[i];
					if (direction == 0) {
						if (origin < minVal || origin > maxVal) go = false;
					} else {
						const float rc = sel3(rx, ry, rz, i);
						float t1 = (minVal - origin) * rc, t2 = (maxVal - origin) * rc;
						if (t1 > t2) { const float tmp = t1; t1 = t2; t2 = tmp; }
						mint = smax(mint, t1);
						maxt = smin(maxt, t2);
						if (mint > maxt) go = false;
					}
				}
				float rayMinT = rmint;
				if (rayMinT == kEpsilon) {
					float m = smax(smax(fabsf(ox), fabsf(oy)), fabsf(oz));
					if (MODE == 0) m = smax(m, kEpsilon);    // only the (ray, its) variant has the inner max
					rayMinT *= m;
				}
				if (rayMinT > mint) mint = rayMinT;
				if (rmaxt < maxt) maxt = rmaxt;
				if (!(maxt > mint)) go = false;
				best_t = MG_INF; best_u = 0; best_v = 0; best_prim = kNoPrim; best_shape = 0;
				found = false;
				done = !go;       // a ray that misses the scene's box is finished at once
				has = go;
				if (COUNT && q.rec && !go) { if (MODE != 1 && !(MODE == 0 && BIN)) rec_add(kReqHit, id * kPathSlots + 2); q.rec_len[rec_slot] = rec_n; }
				if (go) {
					if (kMbox) {
						#pragma unroll
						for (int i = 0; i < 8; ++i) s_mbox[i][tid] = 0xFFFFFFFFu;
					}
					// entry point (stack[enPt]) and current exit point (stack[exPt]) in registers
					enx = ox + mint * dx; eny = oy + mint * dy; enz = oz + mint * dz;      // stack[enPt].p = ray(mint)
					tmax0 = maxt;
					if (COUNT) c_ten = mint;
					ex_t = maxt; exx = ox + maxt * dx; exy = oy + maxt * dy; exz = oz + maxt * dz;
					ex_node = kNullNode; ex_ref = kSentinel;
					sp = 0; cur = 0; e_cont = kNoPrim;
					nd = load_node(0u);
				}
			}
		}

		// ---- one leaf visit of every live lane: descend, test the leaf, pop ----
		if (has) {
			{
				bool inner = !(nd.x & 0x80000000u);     // nd = sc.nodes[cur] is part of the lane's state
				// The descent stops as soon as fewer than q.desc_min lanes are still on inner nodes: the lanes
				// that wait in a leaf go on, the few stragglers resume their descent in the next round.
				do { if (inner) {
					// One step of rayIntersectHavran's inner loop (sahkdtree3.h:196-252), written without
					// branches: this loop is bound by instruction issue (exec-mask bookkeeping of a branchy
					// version costs more than the arithmetic), not by memory.  The entry / exit points are
					// kept as the 3-vectors the reference stores (ray(t) with the split axis overwritten).
					const float split = __uint_as_float(nd.y);
					const int axis = (int) (nd.x & 3u);
					const uint32_t left = nd.x >> 2;            // device nodes hold the absolute index of the left child
					// both children in one 16-byte load (sibling pairs are 16-byte aligned in the device order), issued
					// before the case logic below instead of after it: the step is a chain of dependent fetches
					const uint4 pair = load_pair(left);
					if (COUNT) c_inner++;
					MG_WSLOT(w_inner);
					const float pen = sel3(enx, eny, enz, axis), pex = sel3(exx, exy, exz, axis);
					const bool A = pen <= split, B = pex <= split, C = pen == split, D = split < pex;
					//   A &&  B        : left only            (N1-N3, P5, Z2, Z3)
					//   A && !B &&  C  : right only           (Z1)
					//   A && !B && !C  : near left, far right (N4)  -> push
					//  !A &&  D        : right only           (P1-P3, N5)
					//  !A && !D        : near right, far left (P4)  -> push
					// the case logic is done on wave masks (SALU) to keep it off the vector pipe
					const uint64_t mA = __builtin_amdgcn_ballot_w64(A), mB = __builtin_amdgcn_ballot_w64(B);
					const uint64_t mC = __builtin_amdgcn_ballot_w64(C), mD = __builtin_amdgcn_ballot_w64(D);
					const bool side1 = __builtin_amdgcn_inverse_ballot_w64(~mA | (~mB & mC));   // go to the right child now
					const bool push = __builtin_amdgcn_inverse_ballot_w64((mA & ~mB & ~mC) | (~mA & ~mD));
					const uint32_t side = side1 ? 1u : 0u;
					if (push) {
						// push the current exit point's reference; (cur, far child) becomes the exit point
						if (sp < kStackLDS) s_stack[sp][tid] = ex_ref;
						else { q.spill[(size_t) (sp - kStackLDS) * q.spill_stride + gtid] = ex_ref; if (COUNT) g_spill++; }
						++sp;
						const uint32_t farRight = A ? 1u : 0u;
						const float distToSplit = (split - sel3(ox, oy, oz, axis)) * sel3(rx, ry, rz, axis);
						ex_ref = (cur << 1) | farRight;
						ex_t = distToSplit;
						const float px = ox + distToSplit * dx, py = oy + distToSplit * dy, pz = oz + distToSplit * dz;
						exx = (axis == 0) ? split : px; exy = (axis == 1) ? split : py; exz = (axis == 2) ? split : pz;   // selects, not branches
						ex_node = left + farRight;
					}
					cur = left + side;
					nd = side1 ? make_uint2(pair.z, pair.w) : make_uint2(pair.x, pair.y);
				}
				// evaluated for all lanes after the step (a lane that did not step sits on a leaf): the flag then is one
				// compare on the merged register instead of a value carried through the branch
				inner = !(nd.x & 0x80000000u);
				} while ((uint32_t) __popcll(__builtin_amdgcn_ballot_w64(inner)) >= desc_min);

				if (!inner) {
				// --- leaf: test the primitives (sahkdtree3.h:262-288, skdtree.h:244-336) ---
				if (COUNT && e_cont == kNoPrim) c_leaf++;      // a resumed leaf was counted already
				MG_WSLOT(w_outer);
				bool hitShadow = false, more = false;
				{
					uint32_t e = (e_cont != kNoPrim) ? e_cont : (nd.x & 0x7FFFFFFFu);     // resume an interrupted leaf
					const uint32_t last = nd.y;
					// record = 3 x 16 B: A = (k<<30 | non-occluder<<29 | prim, n_u, n_v, n_d), B = (a_u, a_v, b_nu, b_nv),
					// C = (c_nu, c_nv, shape, -).  A alone decides the mailbox test and the plane distance t; B and C
					// are only fetched for primitives whose t lies inside [mint, maxt] (triaccel.h:141-149).
					uint4 A;
					more = e != last;
					if (more) { A = ld_stream<2>(leaf_head(sc, e)); if (COUNT) { g_head++; rec_add(kReqLeaf, kLeafStride * e); } }
					// like the descent, the primitive loop stops when fewer than q.leaf_min lanes have entries left;
					// those lanes keep their position (e_cont) and go on in the next round
					// the primitive loop runs at a raised wave priority: a wave in it holds the lanes of the others back the
					// shortest (1.55 entries per visit), and its record fetches go out ahead of the descent steps of the waves it
					// shares the SIMD with: 195.0 -> 192.5 ms of traversal per C3 frame (profiles/r04u_exp_trace_wave_priority.txt)
					__builtin_amdgcn_s_setprio(2);
					do { if (more) {
						uint4 An = A;
						if (e + 1 != last) { An = ld_stream<2>(leaf_head(sc, e + 1)); if (COUNT) { g_head++; rec_add(kReqLeaf, kLeafStride * (e + 1)); } }      // next record's head in flight
						const uint32_t prim = A.x & 0x1FFFFFFFu, k = A.x >> 30;
						if (COUNT) c_idx++;
						MG_WSLOT(w_leaf);
						// Flat form of the mailbox test + TriAccel::rayIntersect: the plane distance t is computed for
						// every entry (selects, no branches) and masked afterwards; only the barycentric part, which
						// needs the rest of the record, is conditional.
						uint32_t *mslot = &s_mbox[kMbox ? (prim & 7u) : 0u][tid];
						const bool fresh = !kMbox || *mslot != prim;              // not in the mailbox
						const bool occl = !(MODE != 0 && (A.x & 0x20000000u));   // shape->isOccluder() (skdtree.h:318-333)
						const bool ok = fresh && occl && (k != 3u);               // k == 3: degenerate triangle or another shape
						if (COUNT && fresh) c_tri++;
						if (sc.has_shapes && k == 3u && A.y != 0u && fresh && occl) {     // has_shapes is uniform: one scalar branch
							// a non-triangle shape (skdtree.h:287-296 / :328-332); A.y = shape type, B = centre + radius
							const uint4 B = *leaf_tail(sc, e, 0);
							if (COUNT) { g_tail++; rec_add(kReqLeaf, kLeafStride * e + 1u); }
							const V3 ctr(__uint_as_float(B.x), __uint_as_float(B.y), __uint_as_float(B.z));
							const float rad = __uint_as_float(B.w);
							if (MODE != 0) {
								if (sphere_occludes(ctr, rad, V3(ox, oy, oz), V3(dx, dy, dz), mint, maxt)) hitShadow = true;
							} else {
								float ts;
								if (sphere_intersect(ctr, rad, V3(ox, oy, oz), V3(dx, dy, dz), mint, maxt, ts)) {
									maxt = ts;
									best_t = ts; best_u = 0.0f; best_v = 0.0f; best_prim = prim;
									best_shape = leaf_tail(sc, e, 1)->z;
								}
							}
						}
						const bool k0 = k == 0u, k1 = k == 1u;
						const float n_u = __uint_as_float(A.y), n_v = __uint_as_float(A.z), n_d = __uint_as_float(A.w);
						const float o_u = k0 ? oy : (k1 ? oz : ox), o_v = k0 ? oz : (k1 ? ox : oy), o_k = k0 ? ox : (k1 ? oy : oz);
						const float d_u = k0 ? dy : (k1 ? dz : dx), d_v = k0 ? dz : (k1 ? dx : dy), d_k = k0 ? dx : (k1 ? dy : dz);
						const float recip = 1.0f / (d_u * n_u + d_v * n_v + d_k);
						const float t = (n_d - o_u * n_u - o_v * n_v - o_k) * recip;
						if (ok && !(t < mint || t > maxt)) {
							const uint4 B = ld_stream<2>(leaf_tail(sc, e, 0));
							const uint4 C = ld_stream<2>(leaf_tail(sc, e, 1));         // c_nu, c_nv, shape index, -
							if (COUNT) { g_tail += 2u; rec_add(kReqLeaf, kLeafStride * e + 1u); rec_add(kReqLeaf, kLeafStride * e + 2u); }
						if (COUNT && (t < c_ten - 1e-4f * fabsf(c_ten) || t > ex_t + 1e-4f * fabsf(ex_t))) c_tail_out++;
							const float a_u = __uint_as_float(B.x), a_v = __uint_as_float(B.y);
							const float b_nu = __uint_as_float(B.z), b_nv = __uint_as_float(B.w);
							const float c_nu = __uint_as_float(C.x), c_nv = __uint_as_float(C.y);
							const float hu = o_u + t * d_u - a_u;
							const float hv = o_v + t * d_v - a_v;
							const float u = hv * b_nu + hu * b_nv;
							const float v = hu * c_nu + hv * c_nv;
							if (u >= 0 && v >= 0 && u + v <= 1.0f) {
								if (MODE != 0) hitShadow = true;
								maxt = t;      // a later hit with equal t replaces this one (t > maxt rejects)
								best_t = t; best_u = u; best_v = v; best_prim = prim; best_shape = C.z;
							}
						}
						if (kMbox) *mslot = prim;         // (re)writing an entry that is already there changes nothing
						A = An;
						++e;
					}
					more = (e != last) && !hitShadow;      // for all lanes: those that did not step have e == last or hitShadow
					} while ((uint32_t) __popcll(__builtin_amdgcn_ballot_w64(more)) >= leaf_min);
					__builtin_amdgcn_s_setprio(0);
					e_cont = more ? e : kNoPrim;
				}
				bool finished = false;
				if (hitShadow) finished = true;
				else if (more) { /* leaf not finished yet */ }
				else if (ex_t > maxt) finished = true;
				else {
					// --- pop: the exit point becomes the entry point ---
					enx = exx; eny = exy; enz = exz;
					if (COUNT) c_ten = ex_t;
					cur = ex_node;
					if (cur == kNullNode) {
						finished = true;
					} else {
						--sp;
						nd = load_node(cur);         // in flight together with the parent's node below
						const uint32_t ref = (sp < kStackLDS) ? s_stack[sp][tid] : q.spill[(size_t) (sp - kStackLDS) * q.spill_stride + gtid];
						if (ref == kSentinel) {
							ex_t = tmax0; exx = ox + tmax0 * dx; exy = oy + tmax0 * dy; exz = oz + tmax0 * dz;
							ex_node = kNullNode; ex_ref = kSentinel;
						} else {
							// the exit point is a function of (parent node, ray): rebuilt with the reference's formulas (sahkdtree3.h:233,248-249)
							const uint2 pn = load_node(ref >> 1);
							const int axis = (int) (pn.x & 3u);
							const float split = __uint_as_float(pn.y);
							ex_node = (pn.x >> 2) + (ref & 1u);
							ex_t = (split - sel3(ox, oy, oz, axis)) * sel3(rx, ry, rz, axis);
							const float px = ox + ex_t * dx, py = oy + ex_t * dy, pz = oz + ex_t * dz;
							exx = (axis == 0) ? split : px; exy = (axis == 1) ? split : py; exz = (axis == 2) ? split : pz;
							ex_ref = ref;
						}
					}
				}
				if (finished) {
					has = false; done = true; found = (MODE == 0) ? (best_prim != kNoPrim) : hitShadow;
					if (COUNT && q.rec) { if (MODE != 1 && !(MODE == 0 && BIN)) rec_add(kReqHit, id * kPathSlots + 2); q.rec_len[rec_slot] = rec_n; }      // binned hits are streamed
				}
				}
			}
		}
	}

	if (COUNT) {
		// wave reduction, then one atomic per wave and counter
		unsigned long long v[16] = { c_inner, c_leaf, c_idx, c_tri, w_inner, w_leaf, w_outer, w_batch, g_pair, l_pair, g_node, l_node, g_tail, g_spill, g_head, c_tail_out };
		#pragma unroll
		for (int k = 0; k < 16; ++k) {
			unsigned long long x = v[k];
			for (int off = 32; off > 0; off >>= 1)
				x += __shfl_down(x, off);
			if (lane == 0)
				atomicAdd(&q.trace_counts[k], x);
		}
	}
}

template <int MODE, bool COUNT, bool BIN>
__global__ __launch_bounds__(kTraceBlock, trace_waves_per_simd(MODE)) void k_trace(DTraceScene sc, DPaths ps, DQueues q,
                                                          const uint32_t *queue, uint32_t n_host, const uint32_t *n_dev) {
	__shared__ uint32_t s_stack[kStackLDS][kTraceBlock];
	__shared__ uint32_t s_mbox[(MODE == 0 || COUNT) ? 8 : 1][kTraceBlock];
	__shared__ uint4 s_top[kTopPairs ? kTopPairs : 1];
	// the number of rays: known to the host, or left in device memory by the kernel that filled the queue
	const uint32_t n = n_dev ? (uint32_t) __builtin_amdgcn_readfirstlane((int) *n_dev) : n_host;
	const TracePlan plan = trace_plan(n, MODE, q);
	if (blockIdx.x >= plan.blocks)
		return;                            // a grid sized for the worst case: nothing left for this workgroup
	if (q.dev_stats && blockIdx.x == 0 && threadIdx.x == 0) {
		atomicAdd(&q.dev_stats[MODE == 0 ? kStatClosest : kStatShadow], (unsigned long long) n);
		atomicAdd(&q.dev_stats[kStatLaunches], 1ull);
	}
	const uint32_t first = (blockIdx.x * (kTraceBlock / 64u) + (threadIdx.x >> 6)) * plan.batch;
	const uint32_t stride = plan.blocks * (kTraceBlock / 64u) * plan.batch;      // queue entries per round of the grid
	if (kTopPairs) {
		// the device tree is padded to at least 2 * kTopPairs nodes (mtsgpu_upload_scene)
		for (uint32_t t = threadIdx.x; t < kTopPairs; t += kTraceBlock) s_top[t] = reinterpret_cast<const uint4 *>(sc.nodes)[t];
		__syncthreads();
	}
	trace_body<MODE, COUNT, BIN>(sc, ps, q, plan, queue, n, first, stride, s_stack, s_mbox, s_top);
}

// Device-driven bounces: the per-bin views k_shade needs, from the shard counters the closest-hit launch left in `cur`
// (what the host computes from a read-back otherwise), and the counter set of the NEXT bounce cleared.  One workgroup.
__global__ __launch_bounds__(256) void k_prep(const uint32_t *cur, uint32_t *next_set, BinView *views, uint32_t bin_seg_cap,
                                              unsigned long long *dev_stats) {
	__shared__ uint32_t s_cnt[kNumBins * kBinShards];
	const uint32_t t = threadIdx.x;
	if (t < (uint32_t) (kNumBins * kBinShards)) {
		const uint32_t c = cur[t * kCounterStride];
		s_cnt[t] = c;
		if (c > bin_seg_cap && dev_stats) atomicAdd(&dev_stats[kStatOverflow], 1ull);
	}
	if (t < (uint32_t) kNumCounters) next_set[t * kCounterStride] = 0u;
	__syncthreads();
	if (t < (uint32_t) kNumBins) {
		uint32_t acc = 0;
		for (int k = 0; k < kBinShards; ++k) {
			views[t].prefix[k] = acc;
			const uint32_t c = s_cnt[t * kBinShards + k];
			acc += c < bin_seg_cap ? c : bin_seg_cap;      // entries beyond the capacity were dropped (and flagged)
		}
		views[t].prefix[kBinShards] = acc;
	}
}

// ===========================================================================
// Intersection record, luminaires, BSDFs
// ===========================================================================
struct Its {
	V3 p, geoN, shS, shT, shN, wi;
	uint32_t shape;
};

// fillIntersectionRecord<true> (include/mitsuba/render/skdtree.h:352-432)
// t0, t1, t2: the first three chunks of the primitive's gather record (sc.tri_pos), fetched by the caller
__device__ __forceinline__ void fill_its(const DScene &sc, V3 rayO, V3 rayD, float t, uint32_t prim, float u, float v,
                                         const float4 t0, const float4 t1, const float4 t2, Its &its) {
	V3 sS, sT;
	if (__float_as_uint(t2.w) & 0x80000000u) {
		// Sphere::fillIntersectionRecord (src/shapes/sphere.cpp:136-178): its.p = ray(t), frame from dpdu / dpdv
		its.shape = __float_as_uint(t2.z);
		const float *SP = sc.shape_params + 24 * (size_t) its.shape;
		const float *O2W = SP + 5, *W2O = SP + 14;
		const V3 center(SP[0], SP[1], SP[2]);
		const float radius = SP[3];
		its.p = V3(rayO.x + t * rayD.x, rayO.y + t * rayD.y, rayO.z + t * rayD.z);
		const V3 pc = its.p - center;
		const V3 local(W2O[0] * pc.x + W2O[1] * pc.y + W2O[2] * pc.z, W2O[3] * pc.x + W2O[4] * pc.y + W2O[5] * pc.z,
		               W2O[6] * pc.x + W2O[7] * pc.y + W2O[8] * pc.z);
		const float theta = dacos(smin(smax(local.z / radius, -1.0f), 1.0f));
		const V3 du(-local.y * (2 * kPi), local.x * (2 * kPi), 0 * (2 * kPi));
		const V3 dpdu(O2W[0] * du.x + O2W[1] * du.y + O2W[2] * du.z, O2W[3] * du.x + O2W[4] * du.y + O2W[5] * du.z,
		              O2W[6] * du.x + O2W[7] * du.y + O2W[8] * du.z);
		V3 n = normalize(pc);
		const float zrad = sqrtf(local.x * local.x + local.y * local.y);
		if (zrad > 0) {
			const float invZRad = 1.0f / zrad, cosPhi = local.x * invZRad, sinPhi = local.y * invZRad;
			float st, ct;
			dsincos(theta, st, ct);
			const V3 dv((local.z * cosPhi) * kPi, (local.z * sinPhi) * kPi, (-st * radius) * kPi);
			const V3 dpdv(O2W[0] * dv.x + O2W[1] * dv.y + O2W[2] * dv.z, O2W[3] * dv.x + O2W[4] * dv.y + O2W[5] * dv.z,
			              O2W[6] * dv.x + O2W[7] * dv.y + O2W[8] * dv.z);
			sS = normalize(dpdu);
			sT = normalize(dpdv);
		} else {
			coordinateSystem(n, sS, sT);
		}
		if (SP[4] != 0.0f)
			n = V3(n.x * -1, n.y * -1, n.z * -1);
		its.geoN = n; its.shN = n;
	} else {
	const V3 p0(t0.x, t0.y, t0.z), p1(t0.w, t1.x, t1.y), p2(t1.z, t1.w, t2.x);
	const float bx = 1 - u - v, by = u, bz = v;
	its.p = V3(p0.x * bx + p1.x * by + p2.x * bz, p0.y * bx + p1.y * by + p2.y * bz, p0.z * bx + p1.z * by + p2.z * bz);
	V3 faceNormal = cross(p1 - p0, p2 - p0);
	const float len = length(faceNormal);
	if (!isZero(faceNormal))
		faceNormal = divs(faceNormal, len);
	its.geoN = faceNormal;
	its.shape = __float_as_uint(t2.z);
	if (__float_as_uint(t2.w) & 1u) {
		const float4 *TN = sc.tri_nrm + kTriStride * (size_t) prim;
		const float4 m0 = TN[0], m1 = TN[1], m2 = TN[2];
		const V3 n0(m0.x, m0.y, m0.z), n1(m0.w, m1.x, m1.y), n2(m1.z, m1.w, m2.x);
		its.shN = normalize(V3(n0.x * bx + n1.x * by + n2.x * bz, n0.y * bx + n1.y * by + n2.y * bz, n0.z * bx + n1.z * by + n2.z * bz));
	} else {
		its.shN = its.geoN;
	}
	coordinateSystem(its.shN, sS, sT);
	}
	its.shS = sS; its.shT = sT;
	const V3 md = -rayD;
	its.wi = V3(dot(md, its.shS), dot(md, its.shT), dot(md, its.shN));
}

struct LRec { V3 p, n, d, value; float pdf; int lum; };

// DiscretePDF::sample / sampleReuse (include/mitsuba/core/pdf.h:102-133)
__device__ __forceinline__ int dpdf_sample_reuse(const float *cdf, uint32_t n, float &sampleValue) {
	uint32_t lo = 0, count = n + 1;       // std::lower_bound over n + 1 knots
	while (count > 0) {
		const uint32_t step = count / 2, it = lo + step;
		if (cdf[it] < sampleValue) { lo = it + 1; count -= step + 1; }
		else count = step;
	}
	int index = (int) lo - 1;
	if (index < 0) index = 0;
	if (index > (int) n - 1) index = (int) n - 1;
	sampleValue = (sampleValue - cdf[index]) / (cdf[index + 1] - cdf[index]);
	return index;
}

// BSphere::rayIntersect (include/mitsuba/core/bsphere.h:85-118)
__device__ __forceinline__ bool bsphere_ray_intersect(V3 center, float radius, V3 o, V3 d, float &nearHit, float &farHit) {
	const V3 originToCenter = center - o;
	const float distToRayClosest = dot(originToCenter, d);
	const float tmp1 = dot(originToCenter, originToCenter) - radius * radius;
	if (tmp1 <= 0.0f) {
		nearHit = farHit = sqrtf(distToRayClosest * distToRayClosest - tmp1) + distToRayClosest;
		return true;
	}
	if (distToRayClosest < 0.0f)
		return false;
	const float sqrOriginToCenterLength = dot(originToCenter, originToCenter);
	const float sqrHalfChordDist = radius * radius - sqrOriginToCenterLength + distToRayClosest * distToRayClosest;
	if (sqrHalfChordDist < 0)
		return false;
	const float hitDistance = sqrtf(sqrHalfChordDist);
	nearHit = distToRayClosest - hitDistance;
	farHit = distToRayClosest + hitDistance;
	if (nearHit == 0)
		nearHit = farHit;
	return true;
}

// ---- EnvMapLuminaire (src/luminaires/envmap.cpp) ----
// MIPMap::triangle(0, x, y) with ERepeat (mipmap.cpp:226-243, getTexel :203-224)
__device__ __forceinline__ V3 env_triangle(const DScene &sc, float x, float y) {
	const int W = (int) sc.env_width, H = (int) sc.env_height;
	x = x * W - 0.5f;
	y = y * H - 0.5f;
	const int xPos = (int) floorf(x), yPos = (int) floorf(y);
	const float dx = x - xPos, dy = y - yPos;
	V3 acc(0, 0, 0);
	#pragma unroll
	for (int k = 0; k < 4; ++k) {
		int tx = xPos + (k >> 1), ty = yPos + (k & 1);
		if (tx <= 0 || ty < 0 || tx >= W || ty >= H) {
			int r = tx - (tx / W) * W; tx = (r < 0) ? r + W : r;               // modulo (util.cpp:424-427)
			r = ty - (ty / H) * H; ty = (r < 0) ? r + H : r;
		}
		const float *t = sc.env_pixels + 3 * ((size_t) tx + (size_t) W * ty);
		const float a = (k < 2) ? (1.0f - dx) : dx, b = (k & 1) ? dy : (1.0f - dy);
		const V3 term(t[0] * a * b, t[1] * a * b, t[2] * a * b);
		acc = (k == 0) ? term : V3(acc.x + term.x, acc.y + term.y, acc.z + term.z);
	}
	return acc;
}
// Le(direction) (envmap.cpp:147-153); LP = luminaire parameter block
__device__ __forceinline__ V3 env_le(const DScene &sc, const float *LP, V3 dir) {
	const float *M = LP + 7;
	const V3 d(M[0] * dir.x + M[1] * dir.y + M[2] * dir.z, M[3] * dir.x + M[4] * dir.y + M[5] * dir.z, M[6] * dir.x + M[7] * dir.y + M[8] * dir.z);
	const float u = .5f * (1 + datan2(d.x, -d.z) / kPi);
	const float v = dacos(smax(-1.0f, smin(1.0f, d.y))) / kPi;
	const V3 t = env_triangle(sc, u, v);
	return V3(t.x * LP[0], t.y * LP[0], t.z * LP[0]);
}
// pdf(p, lRec, delta) (envmap.cpp:176-193); ld = lRec.d
__device__ __forceinline__ float env_pdf(const DScene &sc, const float *LP, V3 ld) {
	const float *M = LP + 7;
	const V3 nd = -ld;
	const V3 d(M[0] * nd.x + M[1] * nd.y + M[2] * nd.z, M[3] * nd.x + M[4] * nd.y + M[5] * nd.z, M[6] * nd.x + M[7] * nd.y + M[8] * nd.z);
	const int rx = (int) sc.env_pdf_width, ry = (int) sc.env_pdf_height;
	const float x = .5f * (1 + datan2(d.x, -d.z) / kPi) * rx;
	const float y = dacos(smax(-1.0f, smin(1.0f, d.y))) / kPi * ry;
	int xPos = (int) floorf(x); xPos = xPos < 0 ? 0 : (xPos > rx - 1 ? rx - 1 : xPos);
	int yPos = (int) floorf(y); yPos = yPos < 0 ? 0 : (yPos > ry - 1 ? ry - 1 : yPos);
	const float pdf = sc.env_pdf[xPos + yPos * rx];
	const float sinTheta = sqrtf(smax(kEpsilon, 1 - d.y * d.y));
	const float psx = 2 * kPi / rx, psy = kPi / ry;
	return pdf / (psx * psy * sinTheta);
}

// Scene::sampleLuminaire without the visibility test (scene.cpp:396-415):
// returns true when a shadow ray has to be traced; value is already divided by pdf.
__device__ __forceinline__ bool sample_luminaire(const DScene &sc, V3 p, float s0, float s1, LRec &lRec) {
	float sx = s0, sy = s1;
	const int l = dpdf_sample_reuse(sc.lum_sel_cdf, sc.n_lums, sx);
	const float lumPdf = sc.lum_sel_pdf[l];
	const float *LP = sc.lum_params + kLumStride * (size_t) l;
	if (sc.lum_type[l] == 0u && sc.shape_type[sc.lum_shape[l]] == 1u) {
		// AreaLuminaire::sample (area.cpp:68-79) -> Sphere::sampleSolidAngle (src/shapes/sphere.cpp:196-237)
		const float *SP = sc.shape_params + 24 * (size_t) sc.lum_shape[l];
		const V3 center(SP[0], SP[1], SP[2]);
		const float radius = SP[3];
		const V3 w = center - p;
		const float invDistW = 1 / length(w);
		const float squareTerm = fabsf(radius * invDistW);
		if (squareTerm >= 1 - kEpsilon) {
			// inside the sphere: uniform sampling
			const V3 d = squareToSphere(sx, sy);
			lRec.p = V3(center.x + d.x * radius, center.y + d.y * radius, center.z + d.z * radius);
			lRec.n = d;
			const V3 lumToPoint = p - lRec.p;
			const float distSquared = dot(lumToPoint, lumToPoint), dp = dot(lumToPoint, lRec.n);
			lRec.pdf = (dp > 0) ? (SP[23] * distSquared * sqrtf(distSquared) / dp) : 0.0f;
		} else {
			const float cosThetaMax = sqrtf(smax(0.0f, 1 - squareTerm * squareTerm));
			// squareToCone (util.cpp:656-662)
			const float cosTheta = (1 - sx) + sx * cosThetaMax;
			const float sinTheta = sqrtf(1 - cosTheta * cosTheta);
			const float phi = sy * (2 * kPi);
			float sphi, cphi;
			dsincos(phi, sphi, cphi);
			const V3 cone(cphi * sinTheta, sphi * sinTheta, cosTheta);
			// Frame(w * invDistW).toWorld(cone)
			const V3 fn = w * invDistW;
			V3 fs, ft;
			coordinateSystem(fn, fs, ft);
			const V3 d(fs.x * cone.x + ft.x * cone.y + fn.x * cone.z, fs.y * cone.x + ft.y * cone.y + fn.y * cone.z,
			           fs.z * cone.x + ft.z * cone.y + fn.z * cone.z);
			float t;
			if (!sphere_intersect(center, radius, p, d, 0.0f, MG_INF, t)) {
				lRec.pdf = 0.0f;         // roundoff: no sample
			} else {
				lRec.p = V3(p.x + t * d.x, p.y + t * d.y, p.z + t * d.z);
				lRec.n = normalize(lRec.p - center);
				lRec.pdf = 1 / ((2 * kPi) * (1 - cosThetaMax));
			}
		}
		lRec.d = p - lRec.p;
		if (lRec.pdf > 0 && dot(lRec.d, lRec.n) > 0) {
			lRec.value = V3(LP[0], LP[1], LP[2]);
			lRec.d = normalize(lRec.d);
		} else {
			lRec.pdf = 0;
		}
	} else if (sc.lum_type[l] == 0u) {
		// AreaLuminaire::sample (area.cpp:68-79) -> Shape::sampleSolidAngle (shape.cpp:65-75)
		// -> TriMesh::sampleArea (trimesh.cpp:297-302) -> Triangle::sample (triangle.cpp:23-47)
		const uint32_t s = (uint32_t) sc.lum_shape[l];
		const uint32_t t0 = sc.shape_tri_offset[s], nT = sc.shape_tri_offset[s + 1] - t0;
		const int index = dpdf_sample_reuse(sc.lum_tri_cdf + sc.lum_cdf_offset[l], nT, sy);
		const size_t tri = (size_t) t0 + (uint32_t) index;
		const float4 *TP = sc.tri_pos + kTriStride * tri;
		const float4 q0 = TP[0], q1 = TP[1], q2 = TP[2];
		const V3 p0(q0.x, q0.y, q0.z), p1(q0.w, q1.x, q1.y), p2(q1.z, q1.w, q2.x);
		float bx, by;
		squareToTriangle(sx, sy, bx, by);
		const V3 sideA = p1 - p0, sideB = p2 - p0;
		lRec.p = V3(p0.x + (sideA.x * bx) + (sideB.x * by), p0.y + (sideA.y * bx) + (sideB.y * by), p0.z + (sideA.z * bx) + (sideB.z * by));
		if (__float_as_uint(q2.w) & 1u) {
			const float4 *TN = sc.tri_nrm + kTriStride * tri;
			const float4 m0 = TN[0], m1 = TN[1], m2 = TN[2];
			const V3 n0(m0.x, m0.y, m0.z), n1(m0.w, m1.x, m1.y), n2(m1.z, m1.w, m2.x);
			const float b0 = 1.0f - bx - by;
			lRec.n = normalize(V3(n0.x * b0 + n1.x * bx + n2.x * by, n0.y * b0 + n1.y * bx + n2.y * by, n0.z * b0 + n1.z * bx + n2.z * by));
		} else {
			lRec.n = normalize(cross(sideA, sideB));
		}
		const float pdfArea = sc.lum_inv_area[l];
		const V3 lumToPoint = p - lRec.p;
		const float distSquared = dot(lumToPoint, lumToPoint), dp = dot(lumToPoint, lRec.n);
		lRec.pdf = (dp > 0) ? (pdfArea * distSquared * sqrtf(distSquared) / dp) : 0.0f;
		lRec.d = p - lRec.p;
		if (lRec.pdf > 0 && dot(lRec.d, lRec.n) > 0) {
			lRec.value = V3(LP[0], LP[1], LP[2]);
			lRec.d = normalize(lRec.d);
		} else {
			lRec.pdf = 0;
		}
	} else if (sc.lum_type[l] == 2u || sc.lum_type[l] == 4u) {
		// PointLuminaire::sample (point.cpp:55-63) / SpotLuminaire::sample (spot.cpp:110-118)
		const V3 pos(LP[3], LP[4], LP[5]);
		const V3 lumToP = p - pos;
		const float invDist = 1.0f / length(lumToP);
		lRec.p = pos;
		lRec.d = lumToP * invDist;
		lRec.n = V3(0, 0, 0);
		lRec.pdf = 1.0f;
		V3 result(LP[0], LP[1], LP[2]);
		if (sc.lum_type[l] == 4u) {
			// falloffCurve (spot.cpp:84-103), constant texture; cosTheta = m_worldToLuminaire(d).z
			const float cosTheta = LP[16] * lRec.d.x + LP[17] * lRec.d.y + LP[18] * lRec.d.z;
			if (cosTheta <= LP[7]) result = V3(0, 0, 0);
			else if (!(cosTheta >= LP[6])) result = result * ((LP[8] - dacos(cosTheta)) * LP[9]);
		}
		lRec.value = result * (invDist * invDist);
	} else if (sc.lum_type[l] == 5u) {
		// EnvMapLuminaire::sampleDirection + sample (envmap.cpp:123-145, :159-172)
		const int rx = (int) sc.env_pdf_width, ry = (int) sc.env_pdf_height;
		const int idx = dpdf_sample_reuse(sc.env_cdf, (uint32_t) (rx * ry), sx);
		float pdf = sc.env_pdf[idx];
		const int row = idx / rx, col = idx - rx * row;
		const float x = col + sx, y = row + sy;
		const V3 tv = env_triangle(sc, x * (1.0f / rx), y * (1.0f / ry));
		const float psx = 2 * kPi / rx, psy = kPi / ry;
		const float theta = psy * y, phi = psx * x - kPi;
		float sinTheta, cosTheta, sinPhi, cosPhi;
		dsincos(theta, sinTheta, cosTheta); dsincos(phi, sinPhi, cosPhi);
		pdf = pdf / (psx * psy * sinTheta);
		const float *L2W = LP + 16;
		const V3 v(-sinTheta * sinPhi, -cosTheta, sinTheta * cosPhi);
		const V3 d(L2W[0] * v.x + L2W[1] * v.y + L2W[2] * v.z, L2W[3] * v.x + L2W[4] * v.y + L2W[5] * v.z, L2W[6] * v.x + L2W[7] * v.y + L2W[8] * v.z);
		lRec.pdf = pdf;
		lRec.value = V3(tv.x * LP[0], tv.y * LP[0], tv.z * LP[0]);
		const V3 center(LP[3], LP[4], LP[5]);
		const float radius = LP[6];
		float nearHit, farHit;
		if (length(p - center) <= radius && bsphere_ray_intersect(center, radius, p, -d, nearHit, farHit)) {
			lRec.p = V3(p.x - d.x * nearHit, p.y - d.y * nearHit, p.z - d.z * nearHit);
			lRec.n = normalize(center - lRec.p);
			lRec.d = d;
		} else {
			lRec.pdf = 0.0f;
		}
	} else if (sc.lum_type[l] == 6u) {
		// CollimatedBeamLuminaire::sample (src/luminaires/collimated.cpp:62-76)
		const float *Wm = LP + 4, *Lm = LP + 16;
		const V3 local(Wm[0] * p.x + Wm[1] * p.y + Wm[2] * p.z + Wm[3], Wm[4] * p.x + Wm[5] * p.y + Wm[6] * p.z + Wm[7],
		               Wm[8] * p.x + Wm[9] * p.y + Wm[10] * p.z + Wm[11]);
		if (sqrtf(local.x * local.x + local.y * local.y) > LP[3] || local.z < 0) {
			lRec.pdf = 0.0f;
		} else {
			lRec.p = V3(Lm[0] * local.x + Lm[1] * local.y + Lm[2] * 0.0f + Lm[3], Lm[4] * local.x + Lm[5] * local.y + Lm[6] * 0.0f + Lm[7],
			            Lm[8] * local.x + Lm[9] * local.y + Lm[10] * 0.0f + Lm[11]);
			lRec.d = V3(Lm[0] * 0.0f + Lm[1] * 0.0f + Lm[2] * 1.0f, Lm[4] * 0.0f + Lm[5] * 0.0f + Lm[6] * 1.0f, Lm[8] * 0.0f + Lm[9] * 0.0f + Lm[10] * 1.0f);
			lRec.n = V3(0, 0, 0);
			lRec.pdf = 1.0f;
			lRec.value = V3(LP[0], LP[1], LP[2]);
		}
	} else if (sc.lum_type[l] == 3u) {
		// DirectionalLuminaire::sample (directional.cpp:84-91)
		const V3 dir(LP[3], LP[4], LP[5]);
		const float k = 2 * LP[6];
		lRec.p = V3(p.x - dir.x * k, p.y - dir.y * k, p.z - dir.z * k);
		lRec.d = dir;
		lRec.n = V3(0, 0, 0);
		lRec.pdf = 1.0f;
		lRec.value = V3(LP[0], LP[1], LP[2]);
	} else {
		// ConstantLuminaire::sample (constant.cpp:73-87)
		const V3 d = squareToSphere(sx, sy);
		const V3 center(LP[3], LP[4], LP[5]);
		const float radius = LP[6];
		float nearHit, farHit;
		if (length(p - center) <= radius && bsphere_ray_intersect(center, radius, p, d, nearHit, farHit)) {
			lRec.p = V3(p.x + d.x * nearHit, p.y + d.y * nearHit, p.z + d.z * nearHit);
			lRec.pdf = 1.0f / (4 * kPi);
			lRec.n = normalize(center - lRec.p);
			lRec.d = -d;
			lRec.value = V3(LP[0], LP[1], LP[2]);
		} else {
			lRec.pdf = 0.0f;
		}
	}
	if (lRec.pdf != 0) {
		lRec.pdf *= lumPdf;
		const float recip = 1.0f / lRec.pdf;
		lRec.value = lRec.value * recip;
		lRec.lum = l;
		return true;
	}
	return false;
}

// Scene::pdfLuminaire (scene.cpp:381-394); Shape::pdfSolidAngle (shape.cpp:77-83); constant.cpp:89-91
__device__ __forceinline__ float pdf_luminaire(const DScene &sc, V3 p, int lum, V3 lp, V3 ln, V3 ld) {
	const float fraction = 1.0f / sc.lum_sel_sum;
	float pdf;
	if (sc.lum_type[lum] == 0u && sc.shape_type[sc.lum_shape[lum]] == 1u) {
		// Sphere::pdfSolidAngle (sphere.cpp:239-255)
		const float *SP = sc.shape_params + 24 * (size_t) sc.lum_shape[lum];
		const V3 w = p - V3(SP[0], SP[1], SP[2]);
		const float invDistW = 1 / length(w);
		const float squareTerm = fabsf(SP[3] * invDistW);
		if (squareTerm >= 1 - kEpsilon) {
			const V3 lumToPoint = p - lp;
			const float distSquared = dot(lumToPoint, lumToPoint), dp = dot(lumToPoint, ln);
			pdf = (dp > 0) ? (SP[23] * distSquared * sqrtf(distSquared) / dp) : 0.0f;
		} else {
			const float cosThetaMax = sqrtf(smax(0.0f, 1 - squareTerm * squareTerm));
			pdf = 1 / (2 * kPi * (1 - cosThetaMax));          // squareToConePdf (util.cpp:652-654)
		}
	} else if (sc.lum_type[lum] == 0u) {
		const V3 lumToPoint = p - lp;
		const float distSquared = dot(lumToPoint, lumToPoint);
		const float invDP = smax(0.0f, sqrtf(distSquared) / dot(lumToPoint, ln));
		pdf = sc.lum_inv_area[lum] * distSquared * invDP;
	} else if (sc.lum_type[lum] == 5u) {
		pdf = env_pdf(sc, sc.lum_params + kLumStride * (size_t) lum, ld);
	} else {
		pdf = 1.0f / (4 * kPi);
	}
	return pdf * fraction;
}

// --- BSDF building blocks (roughmetal.cpp:75-117 == microfacet.cpp:95-136) ---
enum : uint32_t { T_DIFFUSE_REFL = 0x1, T_DIFFUSE_TRANS = 0x2, T_DELTA_REFL = 0x4, T_DELTA_TRANS = 0x8, T_GLOSSY_REFL = 0x10, T_GLOSSY_TRANS = 0x20,
                  T_DELTA = 0xC, T_TRANSMISSION = 0x2A };

__device__ __forceinline__ float frame_tan_theta(V3 v) {      // frame.h:98-103
	const float temp = 1 - v.z * v.z;
	if (temp <= 0.0f) return 0.0f;
	return sqrtf(temp) / v.z;
}
__device__ __forceinline__ float beckmann_d(float alphaB, V3 m) {
	const float ex = frame_tan_theta(m) / alphaB;
	return dexp(-(ex * ex)) / (kPi * alphaB * alphaB * dpow4(m.z));
}
__device__ __forceinline__ V3 sample_beckmann_d(float alphaB, float sx, float sy) {
	const float thetaM = datan(sqrtf(-alphaB * alphaB * dlog(1.0f - sx)));
	const float phiM = (2.0f * kPi) * sy;
	float st, ct, sp, cp;
	dsincos(thetaM, st, ct); dsincos(phiM, sp, cp);
	return V3(st * cp, st * sp, ct);                           // sphericalDirection (util.cpp:543-550)
}
__device__ __forceinline__ float smith_g1(float alphaB, V3 v, V3 m) {
	if (dot(v, m) * v.z <= 0) return 0.0f;
	const float tanTheta = frame_tan_theta(v);
	if (tanTheta == 0.0f) return 1.0f;
	const float a = 1.0f / (alphaB * tanTheta);
	const float aSqr = a * a;
	if (a >= 1.6f) return 1.0f;
	return (3.535f * a + 2.181f * aSqr) / (1.0f + 2.276f * a + 2.577f * aSqr);
}
__device__ __forceinline__ V3 mf_reflect(V3 wi, V3 n) {
	const float s = 2.0f * dot(n, wi);
	return V3(n.x * s - wi.x, n.y * s - wi.y, n.z * s - wi.z);
}

template <int BT> struct Bsdf;

// Lambertian (src/bsdfs/lambertian.cpp:95-126)
template <> struct Bsdf<0> {
	static __device__ __forceinline__ V3 f(const float *P, V3 wi, V3 wo) {
		if (wi.z <= 0 || wo.z <= 0) return V3(0, 0, 0);
		return V3(P[0] * kInvPi, P[1] * kInvPi, P[2] * kInvPi);
	}
	static __device__ __forceinline__ float pdf(const float *P, V3 wi, V3 wo) {
		if (wi.z <= 0 || wo.z <= 0) return 0.0f;
		return wo.z * kInvPi;
	}
	static __device__ __forceinline__ V3 sample(const float *P, V3 wi, float sx, float sy, V3 &wo, float &pdf, uint32_t &st) {
		pdf = 0; st = 0; wo = V3(0, 0, 0);
		if (wi.z <= 0) return V3(0, 0, 0);
		wo = squareToHemispherePSA(sx, sy);
		st = T_DIFFUSE_REFL;
		pdf = wo.z * kInvPi;
		return V3(P[0] * kInvPi, P[1] * kInvPi, P[2] * kInvPi);
	}
};

// Dielectric (src/bsdfs/dielectric.cpp:101-107, :200-261): f = pdf = 0, delta sampling
template <> struct Bsdf<1> {
	static __device__ __forceinline__ V3 f(const float *, V3, V3) { return V3(0, 0, 0); }
	static __device__ __forceinline__ float pdf(const float *, V3, V3) { return 0.0f; }
	static __device__ __forceinline__ V3 sample(const float *P, V3 wi, float sx, float sy, V3 &wo, float &pdf, uint32_t &st) {
		const float cosThetaI = wi.z;
		float etaI = P[1], etaT = P[0];
		const bool entering = cosThetaI > 0.0f;
		if (!entering) { const float t = etaI; etaI = etaT; etaT = t; }
		const float eta = etaI / etaT, sinThetaTSqr = eta * eta * (1.0f - wi.z * wi.z);
		float Fr, cosThetaT = 0;
		if (sinThetaTSqr >= 1.0f) {
			Fr = 1.0f;
		} else {
			cosThetaT = sqrtf(1.0f - sinThetaTSqr);
			Fr = fresnelDielectric(fabsf(cosThetaI), cosThetaT, etaI, etaT);
			if (entering) cosThetaT = -cosThetaT;
		}
		if (sx <= Fr) {
			st = T_DELTA_REFL;
			wo = V3(-wi.x, -wi.y, wi.z);
			pdf = Fr * fabsf(wo.z);
			return V3(P[2] * Fr, P[3] * Fr, P[4] * Fr);
		} else {
			st = T_DELTA_TRANS;
			wo = V3(-eta * wi.x, -eta * wi.y, cosThetaT);
			pdf = (1 - Fr) * fabsf(wo.z);
			return V3(P[5] * (1 - Fr) * (eta * eta), P[6] * (1 - Fr) * (eta * eta), P[7] * (1 - Fr) * (eta * eta));
		}
	}
};

// RoughMetal (src/bsdfs/roughmetal.cpp:119-167) through BSDF::sample(bRec, pdf, s) (bsdf.cpp:37-48)
template <> struct Bsdf<2> {
	static __device__ __forceinline__ V3 f(const float *P, V3 wi, V3 wo) {
		if (wi.z <= 0 || wo.z <= 0) return V3(0, 0, 0);
		const V3 Hr = normalize(wi + wo);
		const float c = dot(wi, Hr);
		const V3 F(fresnelConductor1(c, P[1], P[4]), fresnelConductor1(c, P[2], P[5]), fresnelConductor1(c, P[3], P[6]));
		const float D = beckmann_d(P[0], Hr);
		const float G = smith_g1(P[0], wi, Hr) * smith_g1(P[0], wo, Hr);
		const float k = D * G / (4.0f * wi.z * wo.z);
		return V3(P[7] * (F.x * k), P[8] * (F.y * k), P[9] * (F.z * k));
	}
	static __device__ __forceinline__ float pdf(const float *P, V3 wi, V3 wo) {
		if (wi.z <= 0 || wo.z <= 0) return 0.0f;
		const V3 Hr = normalize(wi + wo);
		const float dwhr_dwo = 1.0f / (4.0f * fabsf(dot(wo, Hr)));
		return beckmann_d(P[0], Hr) * Hr.z * dwhr_dwo;
	}
	static __device__ __forceinline__ V3 sample(const float *P, V3 wi, float sx, float sy, V3 &wo, float &pdfv, uint32_t &st) {
		pdfv = 0; st = 0; wo = V3(0, 0, 0);
		if (wi.z <= 0) return V3(0, 0, 0);
		const V3 m = sample_beckmann_d(P[0], sx, sy);
		wo = mf_reflect(wi, m);
		st = T_GLOSSY_REFL;
		if (wo.z <= 0) return V3(0, 0, 0);
		const V3 fv = f(P, wi, wo);
		const float p = pdf(P, wi, wo);
		const V3 qv = fv * (1.0f / p);          // sample() = f / pdf; zero -> pdf 0, value 0
		if (isZero(qv)) return V3(0, 0, 0);
		pdfv = p;
		return fv;
	}
};

// Microfacet (src/bsdfs/microfacet.cpp:151-269) through BSDF::sample(bRec, pdf, s)
template <> struct Bsdf<3> {
	static __device__ __forceinline__ V3 f(const float *P, V3 wi, V3 wo) {
		if (wi.z <= 0 || wo.z <= 0) return V3(0, 0, 0);
		const float alphaB = P[0], kd = P[1], ks = P[2], intIOR = P[3], extIOR = P[4];
		const V3 Hr = normalize(wi + wo);
		const float F = fresnel(dot(wi, Hr), extIOR, intIOR);
		const float D = beckmann_d(alphaB, Hr);
		const float G = smith_g1(alphaB, wi, Hr) * smith_g1(alphaB, wo, Hr);
		const float specRef = D * G / (4.0f * wi.z * wo.z);
		const float fk = F * ks;
		V3 r(0.0f + (P[8] * specRef) * fk, 0.0f + (P[9] * specRef) * fk, 0.0f + (P[10] * specRef) * fk);
		const float dk = kInvPi * (1 - F) * kd;
		r.x += P[5] * dk; r.y += P[6] * dk; r.z += P[7] * dk;
		return r;
	}
	static __device__ __forceinline__ float pdf_spec(const float *P, V3 wi, V3 wo) {
		const V3 Hr = normalize(wi + wo);
		return beckmann_d(P[0], Hr) * Hr.z / (4.0f * fabsf(dot(wo, Hr)));
	}
	static __device__ __forceinline__ float pdf(const float *P, V3 wi, V3 wo) {
		if (wi.z <= 0 || wo.z <= 0) return 0.0f;
		const float kd = P[1], ks = P[2], intIOR = P[3], extIOR = P[4];
		float fr = fresnel(wi.z, extIOR, intIOR);
		fr = smin(smax(fr, 0.05f), 0.95f);
		const float diffuseSamplingWeight = (1 - fr) * kd;
		const float specularSamplingWeight = fr * ks;
		const float normalization = 1 / (diffuseSamplingWeight + specularSamplingWeight);
		return (specularSamplingWeight * pdf_spec(P, wi, wo) + diffuseSamplingWeight * (wo.z * kInvPi)) * normalization;
	}
	static __device__ __forceinline__ V3 sample(const float *P, V3 wi, float sx, float sy, V3 &wo, float &pdfv, uint32_t &st) {
		pdfv = 0; st = 0; wo = V3(0, 0, 0);
		if (wi.z <= 0) return V3(0, 0, 0);
		const float kd = P[1], ks = P[2], intIOR = P[3], extIOR = P[4];
		float fr = fresnel(wi.z, extIOR, intIOR);
		fr = smin(smax(fr, 0.05f), 0.95f);
		float diffuseSamplingWeight = (1 - fr) * kd;
		float specularSamplingWeight = fr * ks;
		const float normalization = 1 / (diffuseSamplingWeight + specularSamplingWeight);
		specularSamplingWeight *= normalization;
		diffuseSamplingWeight *= normalization;
		V3 qv(0, 0, 0);
		if (sx < specularSamplingWeight) {
			sx /= specularSamplingWeight;
			const V3 m = sample_beckmann_d(P[0], sx, sy);      // sampleSpecular (:203-218)
			wo = mf_reflect(wi, m);
			st = T_GLOSSY_REFL;
			if (wo.z <= 0) return V3(0, 0, 0);
			const float pdfValue = pdf(P, wi, wo);
			if (pdfValue == 0) return V3(0, 0, 0);
			qv = f(P, wi, wo) * (1.0f / pdfValue);
		} else {
			sx = (sx - specularSamplingWeight) / diffuseSamplingWeight;
			wo = squareToHemispherePSA(sx, sy);                // sampleLambertian (:224-229)
			st = T_DIFFUSE_REFL;
			qv = f(P, wi, wo) * (1.0f / pdf(P, wi, wo));
		}
		if (isZero(qv)) return V3(0, 0, 0);
		pdfv = pdf(P, wi, wo);
		return f(P, wi, wo);
	}
};

// Mirror (src/bsdfs/mirror.cpp:60-86): f = pdf = 0, delta reflection
template <> struct Bsdf<4> {
	static __device__ __forceinline__ V3 f(const float *, V3, V3) { return V3(0, 0, 0); }
	static __device__ __forceinline__ float pdf(const float *, V3, V3) { return 0.0f; }
	static __device__ __forceinline__ V3 sample(const float *P, V3 wi, float, float, V3 &wo, float &pdf, uint32_t &st) {
		wo = V3(-wi.x, -wi.y, wi.z);
		st = T_DELTA_REFL;
		pdf = fabsf(wo.z);
		return V3(P[0], P[1], P[2]);
	}
};

// Phong (src/bsdfs/phong.cpp:104-212), parameters after Phong::configure, through BSDF::sample(bRec, pdf, s)
template <> struct Bsdf<5> {
	static constexpr float kInvTwoPi = 0.15915494309189533577f;
	static __device__ __forceinline__ V3 f(const float *P, V3 wi, V3 wo) {
		if (wi.z <= 0 || wo.z <= 0) return V3(0, 0, 0);
		const V3 R(-wi.x, -wi.y, wi.z);
		const float alpha = dot(R, wo);
		float specRef;
		if (alpha <= 0.0f) specRef = 0.0f;
		else specRef = (P[0] + 2) * kInvTwoPi * dpow(alpha, P[0]) * P[2];
		V3 r(0.0f + P[8] * specRef, 0.0f + P[9] * specRef, 0.0f + P[10] * specRef);
		const float dk = kInvPi * P[1];
		r.x += P[5] * dk; r.y += P[6] * dk; r.z += P[7] * dk;
		return r;
	}
	static __device__ __forceinline__ float pdf_spec(const float *P, V3 wi, V3 wo) {
		const V3 R(-wi.x, -wi.y, wi.z);
		const float alpha = dot(R, wo);
		float specPdf = dpow(alpha, P[0]) * (P[0] + 1.0f) / (2.0f * kPi);
		if (alpha <= 0) specPdf = 0;
		return specPdf;
	}
	static __device__ __forceinline__ float pdf(const float *P, V3 wi, V3 wo) {
		if (wo.z <= 0 || wi.z <= 0) return 0.0f;
		return P[3] * pdf_spec(P, wi, wo) + P[4] * (wo.z * kInvPi);
	}
	static __device__ __forceinline__ V3 sample(const float *P, V3 wi, float sx, float sy, V3 &wo, float &pdfv, uint32_t &st) {
		pdfv = 0; st = 0; wo = V3(0, 0, 0);
		if (wi.z <= 0) return V3(0, 0, 0);
		V3 qv(0, 0, 0);
		if (sx <= P[3]) {
			sx /= P[3];
			const V3 R(-wi.x, -wi.y, wi.z);                     // sampleSpecular (:157-182)
			const float sinAlpha = sqrtf(1 - dpow(sy, 2 / (P[0] + 1)));
			const float cosAlpha = dpow(sy, 1 / (P[0] + 1));
			const float phi = (2.0f * kPi) * sx;
			float sp, cp; dsincos(phi, sp, cp);
			const V3 l(sinAlpha * cp, sinAlpha * sp, cosAlpha);
			V3 fs, ft;
			coordinateSystem(R, fs, ft);                         // Frame(R).toWorld(localDir)
			wo = V3(fs.x * l.x + ft.x * l.y + R.x * l.z, fs.y * l.x + ft.y * l.y + R.y * l.z, fs.z * l.x + ft.z * l.y + R.z * l.z);
			st = T_GLOSSY_REFL;
			if (wo.z <= 0) return V3(0, 0, 0);
			const float pdfVal = pdf(P, wi, wo);
			if (pdfVal == 0) return V3(0, 0, 0);
			qv = f(P, wi, wo) * (1.0f / pdfVal);
		} else {
			sx = (sx - P[3]) / P[4];
			wo = squareToHemispherePSA(sx, sy);                  // sampleDiffuse (:188-193)
			st = T_DIFFUSE_REFL;
			qv = f(P, wi, wo) * (1.0f / pdf(P, wi, wo));
		}
		if (isZero(qv)) return V3(0, 0, 0);
		pdfv = pdf(P, wi, wo);
		return f(P, wi, wo);
	}
};

// RoughGlass (src/bsdfs/roughglass.cpp) through BSDF::sample(bRec, pdf, s) (bsdf.cpp:37-48): the plugin's own
// 3-argument sample() takes its pdf by value (roughglass.cpp:619) and therefore does not override the virtual.
// path.cpp leaves bRec.sampler NULL (clamped Fresnel term of the surface normal), quantity = ERadiance.
// P: [0] distribution (0 beckmann, 1 phong, 2 ggx) [1] alpha [2] intIOR [3] extIOR [4..6] specRefl [7..9] specTrans
template <> struct Bsdf<6> {
	static constexpr float kInvTwoPi = 0.15915494309189533577f;
	static __device__ __forceinline__ float signum(float v) { return (v < 0) ? -1.0f : 1.0f; }
	// evalD (roughglass.cpp:213-257)
	static __device__ __forceinline__ float evalD(int distr, V3 m, float alpha) {
		if (m.z <= 0) return 0.0f;
		float result;
		if (distr == 0) {
			const float ex = frame_tan_theta(m) / alpha;
			result = dexp(-(ex * ex)) / (kPi * alpha * alpha * dpow4(m.z));
		} else if (distr == 1) {
			result = (alpha + 2) * kInvTwoPi * dpow(m.z, alpha);
		} else {
			const float tanTheta = frame_tan_theta(m), cosTheta = m.z;
			const float root = alpha / (cosTheta * cosTheta * (alpha * alpha + tanTheta * tanTheta));
			result = kInvPi * (root * root);
		}
		if ((double) result < 1e-40) result = 0;
		return result;
	}
	// sampleD (roughglass.cpp:266-293) + sphericalDirection (util.cpp:543-550)
	static __device__ __forceinline__ V3 sampleD(int distr, float sx, float sy, float alpha) {
		const float phiM = (2.0f * kPi) * sy;
		float thetaM;
		if (distr == 0) thetaM = datan(sqrtf(-alpha * alpha * dlog(1.0f - sx)));
		else if (distr == 1) thetaM = dacos(dpow(sx, (float) 1 / (alpha + 2)));
		else thetaM = datan(alpha * sqrtf(sx) / sqrtf(1.0f - sx));
		float st, ct, sp, cp;
		dsincos(thetaM, st, ct); dsincos(phiM, sp, cp);
		return V3(st * cp, st * sp, ct);
	}
	// smithG1 (roughglass.cpp:303-343)
	static __device__ __forceinline__ float smithG1(int distr, V3 v, V3 m, float alpha) {
		const float tanTheta = fabsf(frame_tan_theta(v));
		if (tanTheta == 0.0f) return 1.0f;
		if (dot(v, m) * v.z <= 0) return 0.0f;
		if (distr == 2) {
			const float root = alpha * tanTheta;
			return 2.0f / (1.0f + sqrtf(1.0f + root * root));
		}
		if (distr == 1) alpha = sqrtf(0.5f * alpha + 1) / tanTheta;     // falls through to the Beckmann case
		const float a = 1.0f / (alpha * tanTheta);
		const float aSqr = a * a;
		if (a >= 1.6f) return 1.0f;
		return (3.535f * a + 2.181f * aSqr) / (1.0f + 2.276f * a + 2.577f * aSqr);
	}
	// the half-vector of f() and pdf() (roughglass.cpp:355-377 == :417-446)
	static __device__ __forceinline__ V3 halfVector(const float *P, V3 wi, V3 wo, bool reflect, float etaI, float etaT) {
		if (reflect)
			return normalize(wo + wi) * signum(wo.z);
		const V3 n = normalize(V3(wi.x * etaI + wo.x * etaT, wi.y * etaI + wo.y * etaT, wi.z * etaI + wo.z * etaT));
		const float sgn = (P[3] > P[2]) ? 1.0f : -1.0f;
		return V3(sgn * n.x, sgn * n.y, sgn * n.z);
	}
	static __device__ __forceinline__ V3 f(const float *P, V3 wi, V3 wo) {
		const int distr = (int) P[0];
		const bool reflect = wi.z * wo.z > 0;
		float etaI = P[3], etaT = P[2];
		if (wi.z < 0) { const float t = etaI; etaI = etaT; etaT = t; }
		const V3 H = halfVector(P, wi, wo, reflect, etaI, etaT);
		const float alpha = P[1];
		const float D = evalD(distr, H, alpha);
		if (D == 0) return V3(0, 0, 0);
		const float F = fresnel(dot(wi, H), P[3], P[2]);
		const float G = smithG1(distr, wi, H, alpha) * smithG1(distr, wo, H, alpha);
		if (reflect) {
			const float value = F * D * G / (4.0f * wi.z * wo.z);
			return V3(P[4] * value, P[5] * value, P[6] * value);
		}
		const float sqrtDenom = etaI * dot(wi, H) + etaT * dot(wo, H);
		float value = ((1 - F) * D * G * etaT * etaT * dot(wi, H) * dot(wo, H)) / (wi.z * wo.z * sqrtDenom * sqrtDenom);
		value *= (etaI * etaI) / (etaT * etaT);                     // bRec.quantity == ERadiance
		const float av = fabsf(value);
		return V3(P[7] * av, P[8] * av, P[9] * av);
	}
	static __device__ __forceinline__ float pdf(const float *P, V3 wi, V3 wo) {
		const int distr = (int) P[0];
		const bool reflect = wi.z * wo.z > 0;
		float etaI = P[3], etaT = P[2];
		if (wi.z < 0) { const float t = etaI; etaI = etaT; etaT = t; }
		const V3 H = halfVector(P, wi, wo, reflect, etaI, etaT);
		float dwh_dwo;
		if (reflect) {
			dwh_dwo = 1.0f / (4.0f * dot(wo, H));
		} else {
			const float sqrtDenom = etaI * dot(wi, H) + etaT * dot(wo, H);
			dwh_dwo = (etaT * etaT * dot(wo, H)) / (sqrtDenom * sqrtDenom);
		}
		float alpha = P[1];
		alpha = alpha * (1.2f - 0.2f * sqrtf(fabsf(wi.z)));
		float prob = evalD(distr, H, alpha);
		const float F = smin(0.9f, smax(0.1f, fresnel(wi.z, P[3], P[2])));
		prob *= reflect ? F : (1 - F);
		return fabsf(prob * H.z * dwh_dwo);
	}
	// sample(bRec, sample) (roughglass.cpp:487-617), then pdf() and f() as BSDF::sample(bRec, pdf, s) does
	static __device__ __forceinline__ V3 sample(const float *P, V3 wi, float sx, float sy, V3 &wo, float &pdfv, uint32_t &st) {
		pdfv = 0; st = 0; wo = V3(0, 0, 0);
		const int distr = (int) P[0];
		bool choseReflection = true;
		float sampleF = smin(0.9f, smax(0.1f, fresnel(wi.z, P[3], P[2])));
		if (sx < sampleF) {
			sx /= sampleF;
		} else {
			sx = (sx - sampleF) / (1 - sampleF);
			choseReflection = false;
		}
		const float alpha = P[1];
		const float sampleAlpha = alpha * (1.2f - 0.2f * sqrtf(fabsf(wi.z)));
		const V3 m = sampleD(distr, sx, sy, sampleAlpha);
		V3 result;
		if (choseReflection) {
			const float k = 2 * dot(wi, m);                          // reflect (roughglass.cpp:180-182)
			wo = V3(k * m.x - wi.x, k * m.y - wi.y, k * m.z - wi.z);
			st = T_GLOSSY_REFL;
			if (wi.z * wo.z <= 0) return V3(0, 0, 0);
			result = V3(P[4], P[5], P[6]);
		} else {
			float etaI = P[3], etaT = P[2];
			if (wi.z < 0) { const float t = etaI; etaI = etaT; etaT = t; }
			const float eta = etaI / etaT, c = dot(wi, m);           // refract (roughglass.cpp:185-201)
			const float cosThetaTSqr = 1 + eta * eta * (c * c - 1);
			if (cosThetaTSqr < 0) return V3(0, 0, 0);
			const float k = eta * c - signum(wi.z) * sqrtf(cosThetaTSqr);
			wo = V3(m.x * k - wi.x * eta, m.y * k - wi.y * eta, m.z * k - wi.z * eta);
			st = T_GLOSSY_TRANS;
			if (wi.z * wo.z >= 0) return V3(0, 0, 0);
			const float scale = (etaI * etaI) / (etaT * etaT);
			result = V3(P[7] * scale, P[8] * scale, P[9] * scale);
		}
		float numerator = evalD(distr, m, alpha) * smithG1(distr, wi, m, alpha) * smithG1(distr, wo, m, alpha) * dot(wi, m);
		float denominator = evalD(distr, m, sampleAlpha) * m.z * wi.z * wo.z;
		float F = fresnel(dot(wi, m), P[3], P[2]);
		if (!choseReflection) {
			sampleF = 1 - sampleF;
			F = 1 - F;
		}
		numerator *= F;
		denominator *= sampleF;
		const float w = fabsf(numerator / denominator);
		const V3 qv(result.x * w, result.y * w, result.z * w);
		if (isZero(qv)) return V3(0, 0, 0);
		pdfv = pdf(P, wi, wo);
		return f(P, wi, wo);
	}
};

// DiffuseTransmitter (src/bsdfs/difftrans.cpp:92-131)
template <> struct Bsdf<7> {
	static __device__ __forceinline__ V3 f(const float *P, V3 wi, V3 wo) {
		if (wi.z * wo.z >= 0) return V3(0, 0, 0);
		return V3(P[0] * kInvPi, P[1] * kInvPi, P[2] * kInvPi);
	}
	static __device__ __forceinline__ float pdf(const float *, V3 wi, V3 wo) {
		if (wi.z * wo.z >= 0) return 0.0f;
		return fabsf(wo.z) * kInvPi;
	}
	static __device__ __forceinline__ V3 sample(const float *P, V3 wi, float sx, float sy, V3 &wo, float &pdfv, uint32_t &st) {
		wo = squareToHemispherePSA(sx, sy);
		if (wi.z > 0) wo.z *= -1;
		st = T_DIFFUSE_TRANS;
		pdfv = fabsf(wo.z) * kInvPi;
		if (wo.z == 0) return V3(0, 0, 0);
		return V3(P[0] * kInvPi, P[1] * kInvPi, P[2] * kInvPi);
	}
};

// "terminal" bin: never evaluated
template <> struct Bsdf<kNumBsdfTypes> {
	static __device__ __forceinline__ V3 f(const float *, V3, V3) { return V3(0, 0, 0); }
	static __device__ __forceinline__ float pdf(const float *, V3, V3) { return 0.0f; }
	static __device__ __forceinline__ V3 sample(const float *, V3, float, float, V3 &wo, float &pdf, uint32_t &st) {
		wo = V3(0, 0, 0); pdf = 0; st = 0; return V3(0, 0, 0);
	}
};

// TwoSidedBRDF adapter (src/bsdfs/twosided.cpp:80-130) around any BSDF whose type carries MTSGPU_BSDF_TWOSIDED
template <int BT> struct Bsdf2 {
	static __device__ __forceinline__ V3 f(bool two, const float *P, V3 wi, V3 wo) {
		if (two && wi.z < 0) { wi.z *= -1; wo.z *= -1; }
		return Bsdf<BT>::f(P, wi, wo);
	}
	static __device__ __forceinline__ float pdf(bool two, const float *P, V3 wi, V3 wo) {
		if (two && wi.z < 0) { wi.z *= -1; wo.z *= -1; }
		return Bsdf<BT>::pdf(P, wi, wo);
	}
	static __device__ __forceinline__ V3 sample(bool two, const float *P, V3 wi, float sx, float sy, V3 &wo, float &pdf, uint32_t &st) {
		bool flipped = false;
		if (two && wi.z < 0) { wi.z *= -1; flipped = true; }
		const V3 result = Bsdf<BT>::sample(P, wi, sx, sy, wo, pdf, st);
		if (flipped && !isZero(result) && pdf != 0) wo.z *= -1;
		return result;
	}
};

// BSDF::f / BSDF::pdf / BSDF::sample(bRec, pdf, sample) read out for tests (mtsgpu_bsdf_eval): the chi-square procedure of
// src/tests/test_chisquare.cpp:299-420 runs against exactly the code k_shade runs.  One query record per thread:
// wi = q[i][0..2]; op 0 / 1: wo = q[i][3..5]; op 2: sample = q[i][3..4].
template <int BT>
__device__ __forceinline__ void bsdf_eval_one(bool two, const float *P, int op, const float *q, float *o) {
	const V3 wi(q[0], q[1], q[2]);
	if (op == 0) {
		const V3 f = Bsdf2<BT>::f(two, P, wi, V3(q[3], q[4], q[5]));
		o[0] = f.x; o[1] = f.y; o[2] = f.z;
	} else if (op == 1) {
		o[0] = Bsdf2<BT>::pdf(two, P, wi, V3(q[3], q[4], q[5]));
	} else {
		V3 wo; float pdf; uint32_t st;
		const V3 f = Bsdf2<BT>::sample(two, P, wi, q[3], q[4], wo, pdf, st);
		o[0] = wo.x; o[1] = wo.y; o[2] = wo.z; o[3] = pdf; o[4] = f.x; o[5] = f.y; o[6] = f.z; o[7] = __uint_as_float(st);
	}
}
struct BsdfParams { float v[kBsdfNParams]; };
__global__ void k_bsdf_eval(uint32_t type, BsdfParams params, int op, uint32_t n, const float *queries, float *out) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const bool two = (type & 0x100u) != 0;
	const float *P = params.v, *q = queries + 6 * (size_t) i;
	float o[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
	switch (type & 0xFFu) {
		case 0: bsdf_eval_one<0>(two, P, op, q, o); break;
		case 1: bsdf_eval_one<1>(two, P, op, q, o); break;
		case 2: bsdf_eval_one<2>(two, P, op, q, o); break;
		case 3: bsdf_eval_one<3>(two, P, op, q, o); break;
		case 4: bsdf_eval_one<4>(two, P, op, q, o); break;
		case 5: bsdf_eval_one<5>(two, P, op, q, o); break;
		case 6: bsdf_eval_one<6>(two, P, op, q, o); break;
		default: bsdf_eval_one<7>(two, P, op, q, o); break;
	}
	#pragma unroll
	for (int k = 0; k < 8; ++k) out[8 * (size_t) i + k] = o[k];
}

__device__ __forceinline__ float mi_weight(float pdfA, float pdfB) {     // path.cpp:218-222
	pdfA *= pdfA;
	pdfB *= pdfB;
	return pdfA / (pdfA + pdfB);
}

// ===========================================================================
// K3+K5: one iteration of the loop of MIPathTracer::Li (path.cpp:61-209) for
// all paths whose current hit has BSDF type BT.  The tail of the previous
// iteration (emitter hit by the BSDF sample, Russian roulette, throughput
// update; path.cpp:171-208) runs first because it needs the new hit.
// ===========================================================================
// ROUNDS: the instantiation the rounds of MIDirectIntegrator use (DConfig::dr_mode != 0); the path tracer and the
// one-sample direct integrator run the one without that code
// The iteration for ONE path (id): what it leaves behind in registers is whether the path continues and its pending
// direct-light term with the shadow ray that guards it.
// The path's 128-byte record is staged in LDS by k_shade (`row`, slot k = record slot k): ro / rd / h / T4 / L4 were
// read from it already and, for a valid hit, slots 0, 1, 3 now hold the primitive's position chunks (see k_shade).
// What changes is written back to `row`: ray_o, ray_d, bsdf when the path continues; thr, Li, misc always.
// A staged path record in LDS: slot k of lane l lives at column k ^ (l & 7) of the lane's 8-slot row when the rows
// are packed (MG_SHADE_PACKED: 128 B per lane, what 5 waves per SIMD can afford; two lanes share a bank group), or at
// column k of a 9-slot row (144 B per lane, conflict-free)
#ifndef MG_SHADE_PACKED
#define MG_SHADE_PACKED 0
#endif
constexpr int kRowStride = MG_SHADE_PACKED ? kPathSlots : kPathSlots + 1;
struct ShadeRow {
	float4 *base; uint32_t x;
	__device__ __forceinline__ float4 &operator[](int k) const { return base[MG_SHADE_PACKED ? ((uint32_t) k ^ x) : (uint32_t) k]; }
};
__device__ __forceinline__ uint32_t shade_row_index(uint32_t lane, uint32_t k) { return lane * kRowStride + (MG_SHADE_PACKED ? (k ^ (lane & 7u)) : k); }

template <int BT, bool ROUNDS>
__device__ __forceinline__ void shade_path(const DScene &sc, const DPaths &ps, const DConfig &cfg, const uint32_t id,
                                           const float4 ro, const float4 rd, const uint4 h, const float4 T4, const float4 L4,
                                           const ShadeRow row, bool &continues, bool &wantShadow, V3 &neeV, V3 &shO, V3 &shD) {
	{
		// rounds of MIDirectIntegrator (DConfig::dr_mode): later BSDF samples start again from the camera hit
		const int mode = ROUNDS ? cfg.dr_mode : 0;
		const bool skipToNee = ROUNDS && mode == 1 && cfg.dr_index > 0, skipToBsdf = ROUNDS && mode == 2;
		const V3 rayO(ro.x, ro.y, ro.z), rayD(rd.x, rd.y, rd.z);
		const bool valid = h.w != kNoPrim;
		V3 thr(T4.x, T4.y, T4.z), Li(L4.x, L4.y, L4.z);
		int depth = __float_as_int(T4.w);
		uint32_t flags = __float_as_uint(L4.w);
		PathSampler smp;
		uint2 misc_zw;
		{
			const uint4 r = reinterpret_cast<const uint4 &>(row[6]);
			smp.stream = (uint64_t) r.x | ((uint64_t) r.y << 32);
			smp.slot = cfg.slot_per_path ? id : (id / cfg.spp);
			smp.j = r.z;
			misc_zw = make_uint2(r.z, r.w);
			smp.d1 = (flags >> F_D1_SHIFT) & 0xFFu; smp.d2 = (flags >> F_D2_SHIFT) & 0xFFu;
		}
		const bool direct = cfg.integrator == 1;
		Its its;
		if (valid)
			fill_its(sc, rayO, rayD, __uint_as_float(h.x), h.w, __uint_as_float(h.y), __uint_as_float(h.z), row[0], row[1], row[3], its);
		const int shapeLum = valid ? sc.shape_lum[its.shape] : -1;

		do {
			if (skipToNee || skipToBsdf) {
				// nothing before the sampling loops runs again
			} else if (flags & F_FIRST) {
				// rRec.rayIntersect (records.inl:89-105): alpha = 1 on a hit
				flags &= ~F_FIRST;
				if (valid) flags |= F_ALPHA;
				// while (rRec.depth <= m_maxDepth || m_maxDepth < 0) with depth == 1 (path.cpp:61): maxDepth == 0 never
				// enters the loop, the sample is black with the alpha of the camera ray
				if (!direct && !(depth <= cfg.max_depth || cfg.max_depth < 0))
					break;
			} else {
				// ---- tail of the previous iteration (path.cpp:147-208) ----
				const float4 B4 = row[5];
				const V3 bsdfVal(B4.x, B4.y, B4.z);
				const float bsdfPdf = B4.w;
				const uint32_t sampledType = flags >> F_ST_SHIFT;
				bool hitLuminaire = false;
				V3 lvalue(0, 0, 0), lp(0, 0, 0), ln(0, 0, 0);
				int llum = -1;
				if (valid) {
					if (shapeLum >= 0) {
						// LuminaireSamplingRecord(its, -ray.d); value = its.Le(-ray.d) (area.cpp:62-66)
						const float *LP = sc.lum_params + kLumStride * (size_t) shapeLum;
						lp = its.p; ln = its.geoN; llum = shapeLum;
						lvalue = (dot(-rayD, its.geoN) <= 0) ? V3(0, 0, 0) : V3(LP[0], LP[1], LP[2]);
						hitLuminaire = true;
					}
				} else {
					if (sc.background_lum >= 0) {
						const float *LP = sc.lum_params + kLumStride * (size_t) sc.background_lum;
						llum = sc.background_lum;
						lvalue = (sc.lum_type[llum] == 5u) ? env_le(sc, LP, normalize(rayD)) : V3(LP[0], LP[1], LP[2]);
						hitLuminaire = true;
					} else {
						if (!direct) depth++;
						break;
					}
				}
				if (hitLuminaire) {
					const float lumPdf = (!(sampledType & T_DELTA)) ? pdf_luminaire(sc, rayO, llum, lp, ln, -rayD) : 0.0f;
					// direct.cpp:189-191 weighs the two strategies by their sample counts
					const float weight = direct ? mi_weight(bsdfPdf * cfg.frac_bsdf, lumPdf * cfg.frac_lum) * cfg.weight_bsdf
					                            : mi_weight(bsdfPdf, lumPdf);
					Li.x += thr.x * lvalue.x * bsdfVal.x * weight;
					Li.y += thr.y * lvalue.y * bsdfVal.y * weight;
					Li.z += thr.z * lvalue.z * bsdfVal.z * weight;
				}
				if (!valid || direct)
					break;                                  // MIDirectIntegrator stops after its BSDF sample (direct.cpp:193)
				flags &= ~F_EMITTED;                       // rRec.type = ERadianceNoEmission
				if (depth >= cfg.rr_depth && !(sampledType & T_TRANSMISSION)) {
					const float approxAlbedo = smin(0.9f, smax(smax(bsdfVal.x, bsdfVal.y), bsdfVal.z));
					if (sampler_next1d(cfg, smp) > approxAlbedo)
						break;
					thr = thr * (1.0f / approxAlbedo);
				}
				thr = thr * bsdfVal;
				depth++;
				if (!(depth <= cfg.max_depth || cfg.max_depth < 0))
					break;
			}

			// ---- head of the iteration (path.cpp:62-98) ----
			if (!valid) {
				if (skipToNee || skipToBsdf) break;
				if ((flags & F_EMITTED) && sc.background_lum >= 0) {
					const float *LP = sc.lum_params + kLumStride * (size_t) sc.background_lum;
					const V3 le = (sc.lum_type[sc.background_lum] == 5u) ? env_le(sc, LP, normalize(rayD)) : V3(LP[0], LP[1], LP[2]);
					Li.x += thr.x * le.x; Li.y += thr.y * le.y; Li.z += thr.z * le.z;
				}
				break;
			}
			if (BT == kNumBsdfTypes)
				break;                                      // bsdf == NULL (path.cpp:72-77)
			const int bsdfIdx = sc.shape_bsdf[its.shape];
			const float *BP = sc.bsdf_params + 16 * (size_t) bsdfIdx;
			const bool twoSided = (sc.bsdf_type[bsdfIdx] & 0x100u) != 0;
			if (shapeLum >= 0 && (flags & F_EMITTED) && !(skipToNee || skipToBsdf)) {
				// Li += pathThroughput * its.Le(-ray.d) (path.cpp:80-81, area.cpp:62-66)
				const float *LP = sc.lum_params + kLumStride * (size_t) shapeLum;
				const V3 le = (dot(-rayD, its.geoN) <= 0) ? V3(0.0f, 0.0f, 0.0f) : V3(LP[0], LP[1], LP[2]);
				Li.x += thr.x * le.x; Li.y += thr.y * le.y; Li.z += thr.z * le.z;
			}
			if (!direct) {      // MonteCarloIntegrator properties; the direct integrator has neither (direct.cpp:33-41)
				if (cfg.max_depth > 0 && depth >= cfg.max_depth)
					break;
				const float wiDotGeoN = -dot(its.geoN, rayD), wiDotShN = its.wi.z;
				if (wiDotGeoN * wiDotShN < 0 && cfg.strict_normals)
					break;
			}
			const bool strict = cfg.strict_normals && !direct;

			// ---- luminaire sampling (path.cpp:100-126) ----
			if (!skipToBsdf) {
				float s0, s1;
				if (ROUNDS && direct && cfg.n_lum > 1) sampler_array2d(cfg, smp, misc_zw.y, 0, (uint32_t) cfg.dr_index, s0, s1);   // direct.cpp:122-123
				else sampler_next2d(cfg, smp, s0, s1);
				LRec lRec;
				if ((!direct || cfg.n_lum > 0) && sample_luminaire(sc, its.p, s0, s1, lRec)) {
					const V3 wo = -lRec.d;
					const V3 woL(dot(wo, its.shS), dot(wo, its.shT), dot(wo, its.shN));
					V3 bsdfVal = Bsdf2<BT>::f(twoSided, BP, its.wi, woL) * fabsf(woL.z);
					const float woDotGeoN = dot(its.geoN, wo);
					if (!isZero(bsdfVal) && (!strict || woDotGeoN * woL.z > 0)) {
						// isIntersectable() || isBackgroundLuminaire() (path.cpp:118-120): 0 for delta luminaires
						const uint32_t lt = sc.lum_type[lRec.lum];      // area, constant and envmap luminaires can be hit by BSDF samples
						const float bsdfPdf = (lt <= 1u || lt == 5u) ? Bsdf2<BT>::pdf(twoSided, BP, its.wi, woL) : 0.0f;
						const float weight = direct ? mi_weight(lRec.pdf * cfg.frac_lum, bsdfPdf * cfg.frac_bsdf) * cfg.weight_lum
						                            : mi_weight(lRec.pdf, bsdfPdf);          // direct.cpp:143-145
						// added to Li by k_trace<shadow> iff the segment is unoccluded
						// (kept in registers until the shadow-queue slot of this path is known, see the end of the kernel)
						neeV = V3(thr.x * lRec.value.x * bsdfVal.x * weight,
						          thr.y * lRec.value.y * bsdfVal.y * weight,
						          thr.z * lRec.value.z * bsdfVal.z * weight);
						shO = its.p;
						shD = lRec.p - its.p;               // Ray(p1, p2 - p1) (scene.h:241-246)
						wantShadow = true;
					}
				}
			}

			if (mode == 1)
				break;                                      // a luminaire round ends here

			// ---- BSDF sampling (path.cpp:128-146) ----
			float s0, s1;
			if (ROUNDS && direct && cfg.n_bsdf > 1) sampler_array2d(cfg, smp, misc_zw.y, cfg.n_lum > 1 ? 1 : 0, (uint32_t) cfg.dr_index, s0, s1);   // direct.cpp:156-157
			else sampler_next2d(cfg, smp, s0, s1);
			if (direct && cfg.n_bsdf <= 0)
				break;                                      // the sample is drawn even when it is not used (direct.cpp:156-161)
			V3 woL; float bsdfPdf; uint32_t sampledType;
			V3 bsdfVal = Bsdf2<BT>::sample(twoSided, BP, its.wi, s0, s1, woL, bsdfPdf, sampledType);
			if (!isZero(bsdfVal))
				bsdfVal = bsdfVal * fabsf(woL.z);          // sampleCos (bsdf.h:273-279)
			if (isZero(bsdfVal))
				break;
			bsdfVal = bsdfVal * (1.0f / bsdfPdf);
			const V3 wo(its.shS.x * woL.x + its.shT.x * woL.y + its.shN.x * woL.z,
			            its.shS.y * woL.x + its.shT.y * woL.y + its.shN.y * woL.z,
			            its.shS.z * woL.x + its.shT.z * woL.y + its.shN.z * woL.z);
			const float woDotGeoN = dot(its.geoN, wo);
			if (woDotGeoN * woL.z <= 0 && strict)
				break;
			// ray = Ray(its.p, wo, time): mint = Epsilon, maxt = inf
			row[0] = make_float4(its.p.x, its.p.y, its.p.z, kEpsilon);
			row[1] = make_float4(wo.x, wo.y, wo.z, MG_INF);
			row[5] = make_float4(bsdfVal.x, bsdfVal.y, bsdfVal.z, bsdfPdf);
			flags = (flags & 0x00FFFFFFu) | (sampledType << F_ST_SHIFT);
			continues = true;
		} while (false);
		if (!continues) { row[0] = ro; row[1] = rd; }      // a path that ends keeps its last ray (the slots held triangle data)

		flags = (flags & ~((0xFFu << F_D1_SHIFT) | (0xFFu << F_D2_SHIFT))) | (smp.d1 << F_D1_SHIFT) | (smp.d2 << F_D2_SHIFT);
		row[3] = make_float4(thr.x, thr.y, thr.z, __int_as_float(depth));
		row[4] = make_float4(Li.x, Li.y, Li.z, __uint_as_float(flags));
		reinterpret_cast<uint4 &>(row[6]) = make_uint4((uint32_t) (smp.stream & 0xFFFFFFFFull), (uint32_t) (smp.stream >> 32), misc_zw.x, misc_zw.y);
	}

}

#ifndef MG_SHADE_ALL_SLOTS
#define MG_SHADE_ALL_SLOTS 1      // whole 128-byte lines in both directions; 0 (only the slots needed: 7 read, 6 written, 3 of the
                                  // triangle) was measured at 66 ms instead of 44 ms per frame: partial lines cost a read-modify-write
#endif

#ifndef MG_SHADE_WAVES
#define MG_SHADE_WAVES 0
#endif
#if MG_SHADE_WAVES
#define MG_SHADE_BOUNDS __launch_bounds__(kShadeBlock, MG_SHADE_WAVES)
#else
#define MG_SHADE_BOUNDS __launch_bounds__(kShadeBlock)
#endif
// the workgroup's LDS: per-wave counts and the two queue offsets of the stream compaction, the staged path records
struct ShadeShared {
	uint32_t cnt[2][kShadeBlock / 64];
	uint32_t base[2];
	float4 rows[kShadeBlock / 64][64 * kRowStride];
};
// One workgroup of k_shade: the paths block * kShadeBlock .. of the material queue whose segment sizes are `prefix`
// (prefix[kBinShards] entries in kBinShards segments of bin_ids)
template <int BT, bool ROUNDS>
__device__ __forceinline__ void shade_block(const DScene &sc, const DPaths &ps, const DConfig &cfg, const DQueues &q, const uint32_t *prefix,
                                            const uint32_t *bin_ids, const uint32_t block, ShadeShared &sh) {
	uint32_t (&s_cnt)[2][kShadeBlock / 64] = sh.cnt;
	uint32_t (&s_base)[2] = sh.base;
	float4 (&s_rows)[kShadeBlock / 64][64 * kRowStride] = sh.rows;
	const uint32_t gtid = block * kShadeBlock + threadIdx.x;
	const uint32_t total = prefix[kBinShards];
	if (block * kShadeBlock >= total)
		return;                            // (uniform) a grid sized for the worst case
	const bool active = gtid < total;
	uint32_t id = 0u;
	uint4 binHit = make_uint4(0u, 0u, 0u, kNoPrim); bool haveBinHit = false;
	if (active) {
		int seg = 0;
		#pragma unroll
		for (int k = 1; k < kBinShards; ++k)
			if (gtid >= prefix[k]) seg = k;
		const size_t at = (size_t) seg * q.bin_seg_cap + (gtid - prefix[seg]);
		id = bin_ids[at];
		// the hit came with the id when the closest-hit kernel filled this bin (DQueues::bin_hits)
		if (q.bin_hits && bin_ids >= q.bins_base && bin_ids < q.bins_base + (size_t) kNumBins * q.bin_stride) {
			binHit = q.bin_hits[(size_t) (bin_ids - q.bins_base) + at]; haveBinHit = true;
		}
	}
	// ---- the path records of the wave, staged through LDS ----
	// The ids come from a material-sorted queue, so every lane owns a different 128-byte line.  Read field by field
	// that is seven 16-byte gathers per lane which each occupy the texture-address unit for 64 lines and -- the L1
	// holds 32 KB, the CU's waves hold far more lines -- mostly go to the L2 again.  Instead eight lanes fetch one
	// record together (one fully used line per request, eight records per instruction), rows of 9 float4 keep the
	// LDS accesses free of bank conflicts, and the rows are written back the same way: whole lines, coalesced.
	// All LDS traffic is private to the wave (program order suffices, no barrier).
	float4 *rows = s_rows[threadIdx.x >> 6];
	const ShadeRow row{ rows + lane_id() * kRowStride, lane_id() & 7u };
	const uint32_t sub = lane_id() & 7u, grp = lane_id() >> 3;
	const uint64_t actMask = __ballot(active);
	#pragma unroll
	for (int r = 0; r < 8; ++r) {
		const uint32_t src = grp + 8u * r;
		const uint32_t sid = (uint32_t) __shfl((int) id, (int) src);
		// slot 7 (the raster position) is only read by the film kernels
		if (((actMask >> src) & 1ull) && (MG_SHADE_ALL_SLOTS || sub != 7u)) rows[shade_row_index(src, sub)] = ld_stream<4>(&ps.base[(size_t) sid * kPathSlots + sub]);
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
	bool continues = false, wantShadow = false;
	V3 neeV(0, 0, 0), shO(0, 0, 0), shD(0, 0, 0);      // pending direct-light term and its shadow ray
	float4 ro = make_float4(0, 0, 0, 0), rd = ro, T4 = ro, L4 = ro;
	uint4 h = make_uint4(0u, 0u, 0u, kNoPrim);
	if (active) {
		// a direct-light term the any-hit kernel parked in slot 2 (DQueues::nee_parked) is added before anything else of this
		// Li iteration, where the sequential loop adds it (path.cpp:124)
		const float4 slot2 = row[2];
		L4 = settled_Li(row[4], slot2);
		h = haveBinHit ? binHit : reinterpret_cast<const uint4 &>(slot2);
		// what the write-back below leaves in slot 2: nothing while terms are parked there, otherwise the hit
		if (q.nee_parked) row[2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
		else reinterpret_cast<uint4 &>(row[2]) = h;
		ro = row[0]; rd = row[1]; T4 = row[3];
		if (ROUNDS && cfg.dr_mode == 2) {
			// rounds of MIDirectIntegrator: later BSDF samples start again from the camera hit (kept in ps.prim)
			if (cfg.dr_index > 0) {
				ro = ps.prim[3 * (size_t) id]; rd = ps.prim[3 * (size_t) id + 1];
				h = reinterpret_cast<const uint4 &>(ps.prim[3 * (size_t) id + 2]);
			} else if (cfg.n_bsdf > 1) {
				ps.prim[3 * (size_t) id] = ro; ps.prim[3 * (size_t) id + 1] = rd;
				ps.prim[3 * (size_t) id + 2] = reinterpret_cast<const float4 &>(h);
			}
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
	// the first 64 bytes of the hit primitives' gather records (three position chunks + one of the normals), four lanes
	// per record, into row slots 0, 1, 3, 4 -- whose contents sit in registers now
	{
		const uint32_t prim = h.w;
		const uint64_t validMask = __ballot(active && prim != kNoPrim);
		const uint32_t sub4 = lane_id() & 3u, grp4 = lane_id() >> 2;
		const uint32_t slotOf = sub4 < 2u ? sub4 : sub4 + 1u;      // chunks 0, 1, 2, 3 -> slots 0, 1, 3, 4
		#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const uint32_t src = grp4 + 16u * r;
			const uint32_t sprim = (uint32_t) __shfl((int) prim, (int) src);
			if (((validMask >> src) & 1ull) && (MG_SHADE_ALL_SLOTS || sub4 != 3u)) rows[shade_row_index(src, slotOf)] = sc.tri_pos[(size_t) sprim * kTriStride + sub4];
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
	if (active)
		shade_path<BT, ROUNDS>(sc, ps, cfg, id, ro, rd, h, T4, L4, row, continues, wantShadow, neeV, shO, shD);
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
	#pragma unroll
	for (int r = 0; r < 8; ++r) {
		const uint32_t src = grp + 8u * r;
		const uint32_t sid = (uint32_t) __shfl((int) id, (int) src);
		// the hit (slot 2) and the raster position (slot 7) do not change here
		if (((actMask >> src) & 1ull) && (MG_SHADE_ALL_SLOTS || (sub != 2u && sub != 7u))) st_stream<4>(&ps.base[(size_t) sid * kPathSlots + sub], rows[shade_row_index(src, sub)]);
	}

	// stream compaction: survivors -> next closest-hit queue, shadow rays -> shadow queue.
	// ballot + prefix popcount inside each wave, an LDS scan across the waves, ONE atomic per workgroup and queue
	// (a queue counter is a single word: every atomic on it serialises, which is why the workgroups are as large as
	// they can be: 1024 threads, 43.6 -> 41.9 ms per 64-spp frame against 512).  Measured and rejected: both queues
	// reserved with one 64-bit atomic on a shared word (43.4 ms); the reservation issued before the records are written
	// back so that its round trip hides under those stores (45 ms: the extra barrier delays the stores of every wave)
	const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
	const unsigned long long mN = __ballot(continues), mS = __ballot(wantShadow);
	if (lane == 0) { s_cnt[0][wave] = (uint32_t) __popcll(mN); s_cnt[1][wave] = (uint32_t) __popcll(mS); }
	__syncthreads();
	if (threadIdx.x < 2) {
		uint32_t total = 0;
		for (int w = 0; w < kShadeBlock / 64; ++w) total += s_cnt[threadIdx.x][w];
		s_base[threadIdx.x] = total ? atomicAdd(&q.counters[threadIdx.x == 0 ? kNextWord : kShadowWord], total) : 0u;
	}
	__syncthreads();
	uint32_t offN = s_base[0], offS = s_base[1];
	for (uint32_t w = 0; w < wave; ++w) { offN += s_cnt[0][w]; offS += s_cnt[1][w]; }
	const unsigned long long below = (1ull << lane) - 1ull;
	if (continues) {
		const uint32_t pos = offN + (uint32_t) __popcll(mN & below);
		q.next[pos] = id;
		if (ps.rqn_o) {       // the new ray once more, in the order of the queue it was appended to
			st_stream<4>(&ps.rqn_o[pos], rows[shade_row_index(lane, 0)]);
			st_stream<4>(&ps.rqn_d[pos], rows[shade_row_index(lane, 1)]);
		}
	}
	if (wantShadow) {
		// the shadow ray lives in queue order (coalesced for both kernels); the path id rides in nee.w
		const uint32_t pos = offS + (uint32_t) __popcll(mS & below);
		st_stream<4>(&ps.shq_o[pos], make_float4(shO.x, shO.y, shO.z, 0.0f));
		st_stream<4>(&ps.shq_d[pos], make_float4(shD.x, shD.y, shD.z, 0.0f));
		st_stream<4>(&ps.shq_nee[pos], make_float4(neeV.x, neeV.y, neeV.z, __uint_as_float(id)));
	}
}

template <int BT, bool ROUNDS>
__global__ MG_SHADE_BOUNDS void k_shade(DScene sc, DPaths ps, DConfig cfg, DQueues q, BinView view_host,
                                                       const BinView *views_dev, const uint32_t *bin_ids) {
	__shared__ ShadeShared sh;
	// the bin's segment sizes: a kernel argument when the host read the counters back, otherwise what k_prep wrote
	shade_block<BT, ROUNDS>(sc, ps, cfg, q, views_dev ? views_dev[BT].prefix : view_host.prefix, bin_ids, blockIdx.x, sh);
}

// All material queues of a bounce in ONE launch (device-driven bounces): the workgroups are dealt to the bins in bin order,
// ceil(size / kShadeBlock) each, the sizes read from what k_prep left in device memory.  A frame of few paths is a chain of
// short launches, and a launch of k_shade -- 1024 threads and 148 KB of LDS per workgroup -- costs 10-20 us even when
// nearly all of its worst-case grid exits at once: one launch per bounce instead of one per BSDF type present.
__global__ MG_SHADE_BOUNDS void k_shade_all(DScene sc, DPaths ps, DConfig cfg, DQueues q, const BinView *views_dev, uint32_t bin_mask) {
	__shared__ ShadeShared sh;
	uint32_t block = blockIdx.x;
	int bin = -1;
	for (int b = 0; b < kNumBins; ++b) {
		if (!((bin_mask >> b) & 1u)) continue;
		const uint32_t nb = (views_dev[b].prefix[kBinShards] + kShadeBlock - 1u) / kShadeBlock;
		if (block < nb) { bin = b; break; }
		block -= nb;
	}
	if (bin < 0) return;
	const uint32_t *prefix = views_dev[bin].prefix, *ids = q.bin(bin);
	switch (bin) {
		case 0: shade_block<0, false>(sc, ps, cfg, q, prefix, ids, block, sh); break;
		case 1: shade_block<1, false>(sc, ps, cfg, q, prefix, ids, block, sh); break;
		case 2: shade_block<2, false>(sc, ps, cfg, q, prefix, ids, block, sh); break;
		case 3: shade_block<3, false>(sc, ps, cfg, q, prefix, ids, block, sh); break;
		case 4: shade_block<4, false>(sc, ps, cfg, q, prefix, ids, block, sh); break;
		case 5: shade_block<5, false>(sc, ps, cfg, q, prefix, ids, block, sh); break;
		case 6: shade_block<6, false>(sc, ps, cfg, q, prefix, ids, block, sh); break;
		case 7: shade_block<7, false>(sc, ps, cfg, q, prefix, ids, block, sh); break;
		default: shade_block<kNumBsdfTypes, false>(sc, ps, cfg, q, prefix, ids, block, sh); break;
	}
}

// ===========================================================================
// K7: ImageBlock::putSample with the tabulated box filter
// (include/mitsuba/render/imageblock.h:80-138, src/librender/rfilter.cpp:40-69).
// One lane per pixel; its samples are added in sample-index order, so the film
// is bit-reproducible and independent of how the image was sharded.
// ===========================================================================
__global__ void k_accumulate(DPaths ps, DConfig cfg, uint32_t n_slots, uint32_t spp, float *film, unsigned long long *path_len) {
	const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
	unsigned long long depthSum = 0;
	if (slot < n_slots) {
	const int W = cfg.width, H = cfg.height;
	// TabulatedFilter of the box filter: size 0.5, factor = 15 / 0.5, table = 1 inside, 0 on the border row
	const float fsize = 0.5f, factor = 15 / fsize;
	// The film pixel being added to stays in registers while consecutive samples fall on it (with the box filter: all
	// samples of the lane's pixel): one load and one store per pixel instead of one of each per sample.  The sums are formed
	// in the same order as before, sample by sample, so the film keeps its bits.
	float *cur = nullptr;
	float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
	for (uint32_t j = 0; j < spp; ++j) {
		const size_t id = (size_t) slot * spp + j;
		const float4 L = settled_Li(ps.Li(id), ps.slot(id, 2));
		const float4 sp = ps.spos(id);
		if (path_len) depthSum += (unsigned long long) __float_as_int(ps.thr(id).w);      // same 128-byte line as Li / spos
		// Spectrum::isValid (spectrum.h:285-290)
		if (L.x != L.x || L.x < 0.0f || L.y != L.y || L.y < 0.0f || L.z != L.z || L.z < 0.0f)
			continue;
		const float alpha = (__float_as_uint(L.w) & F_ALPHA) ? 1.0f : 0.0f;
		const float sx = sp.x - 0.5f - 0, sy = sp.y - 0.5f - 0;
		int xStart = (int) ceilf(sx - fsize), xEnd = (int) floorf(sx + fsize);
		int yStart = (int) ceilf(sy - fsize), yEnd = (int) floorf(sy + fsize);
		// Film::putImageBlock keeps what falls inside the crop window (mfilm.cpp:118-143)
		xStart = max(cfg.crop_x, xStart); yStart = max(cfg.crop_y, yStart);
		xEnd = min(xEnd, cfg.crop_x + W - 1); yEnd = min(yEnd, cfg.crop_y + H - 1);
		for (int y = yStart; y <= yEnd; ++y) {
			const int iy = min((int) (factor * fabsf(y - sy)), 15);
			for (int x = xStart; x <= xEnd; ++x) {
				const int ix = min((int) (factor * fabsf(x - sx)), 15);
				const float weight = (ix == 15 || iy == 15) ? 0.0f : 1.0f;
				// zero-weight taps add spec*0 in the reference: a no-op for valid spectra
				if (weight == 0.0f)
					continue;
				float *px = film + 5 * ((size_t) (y - cfg.crop_y) * W + (x - cfg.crop_x));
				if (px != cur) {
					if (cur) { cur[0] = a0; cur[1] = a1; cur[2] = a2; cur[3] = a3; cur[4] = a4; }
					cur = px; a0 = px[0]; a1 = px[1]; a2 = px[2]; a3 = px[3]; a4 = px[4];
				}
				a0 += L.x * weight; a1 += L.y * weight; a2 += L.z * weight;
				a3 += alpha * weight;
				a4 += weight;
			}
		}
	}
	if (cur) { cur[0] = a0; cur[1] = a1; cur[2] = a2; cur[3] = a3; cur[4] = a4; }
	}
	if (path_len) {
		for (int off = 32; off > 0; off >>= 1)
			depthSum += __shfl_down(depthSum, off);
		if (lane_id() == 0 && depthSum)
			atomicAdd(path_len, depthSum);
	}
}

// The same sums with one WAVE per pixel, for passes of few pixels with many samples each (C4: 18 k pixels x 4096 spp, where
// a lane per pixel is a chain of 4096 dependent round trips on 288 waves: 4.2 ms per pass).  64 consecutive samples are
// loaded by the 64 lanes -- consecutive records: a stream -- and every lane works out its own sample's film pixel; when
// all of them fall, with weight one, on the pixel being summed (the box filter away from pixel borders) the sum is formed
// from the lanes' values in lane = sample order, one readlane + add per channel; any other chunk is added by lane 0 with
// the loop of k_accumulate.  Same additions in the same order: the film keeps its bits.
__device__ __forceinline__ float bcast(float v, uint32_t l) { return __uint_as_float((uint32_t) __builtin_amdgcn_readlane((int) __float_as_uint(v), (int) l)); }

__global__ __launch_bounds__(256) void k_accumulate_wave(DPaths ps, DConfig cfg, uint32_t n_slots, uint32_t spp, float *film, unsigned long long *path_len) {
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t slot = blockIdx.x * 4u + (threadIdx.x >> 6);
	if (slot >= n_slots) return;             // whole waves
	const int W = cfg.width, H = cfg.height;
	const float fsize = 0.5f, factor = 15 / fsize;
	unsigned long long depthSum = 0;
	float *cur = nullptr;                    // uniform: the film pixel being summed, its channels in a0 .. a4
	float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
	for (uint32_t j0 = 0; j0 < spp; j0 += 64u) {
		const uint32_t j = j0 + lane;
		const bool have = j < spp;
		const size_t id = (size_t) slot * spp + (have ? j : spp - 1u);
		const float4 L = settled_Li(ps.Li(id), ps.slot(id, 2));
		const float4 sp = ps.spos(id);
		if (path_len && have) depthSum += (unsigned long long) __float_as_int(ps.thr(id).w);
		// this lane's sample: valid (Spectrum::isValid, spectrum.h:285-290)?  which film pixels does it reach with weight one?
		const bool valid = have && !(L.x != L.x || L.x < 0.0f || L.y != L.y || L.y < 0.0f || L.z != L.z || L.z < 0.0f);
		const float alpha = (__float_as_uint(L.w) & F_ALPHA) ? 1.0f : 0.0f;
		const float sx = sp.x - 0.5f - 0, sy = sp.y - 0.5f - 0;
		int xStart = (int) ceilf(sx - fsize), xEnd = (int) floorf(sx + fsize);
		int yStart = (int) ceilf(sy - fsize), yEnd = (int) floorf(sy + fsize);
		xStart = max(cfg.crop_x, xStart); yStart = max(cfg.crop_y, yStart);
		xEnd = min(xEnd, cfg.crop_x + W - 1); yEnd = min(yEnd, cfg.crop_y + H - 1);
		int taps = 0; float *px = nullptr;
		for (int y = yStart; y <= yEnd; ++y) {
			const int iy = min((int) (factor * fabsf(y - sy)), 15);
			for (int x = xStart; x <= xEnd; ++x) {
				const int ix = min((int) (factor * fabsf(x - sx)), 15);
				if (ix == 15 || iy == 15) continue;           // weight 0: adds nothing
				++taps; px = film + 5 * ((size_t) (y - cfg.crop_y) * W + (x - cfg.crop_x));
			}
		}
		// the pixel of the first valid sample with a tap; the chunk is "plain" if every valid sample has exactly that one tap
		const uint64_t mValid = __builtin_amdgcn_ballot_w64(valid);
		const uint64_t mTap = __builtin_amdgcn_ballot_w64(valid && taps != 0);
		if (mValid == 0ull) continue;
		float *px0 = cur;
		if (mTap != 0ull) {
			const uint32_t f = (uint32_t) __builtin_ctzll(mTap);
			const unsigned long long p = (unsigned long long) px;
			px0 = (float *) (((unsigned long long) (uint32_t) __builtin_amdgcn_readlane((int) (p >> 32), (int) f) << 32)
			               | (unsigned long long) (uint32_t) __builtin_amdgcn_readlane((int) (p & 0xFFFFFFFFull), (int) f));
		}
		const bool plain = __builtin_amdgcn_ballot_w64(valid && taps != 0 && (taps != 1 || px != px0)) == 0ull;
		if (plain) {
			if (mTap == 0ull) continue;                        // valid samples that reach no film pixel
			if (px0 != cur) {
				if (cur && lane == 0) { cur[0] = a0; cur[1] = a1; cur[2] = a2; cur[3] = a3; cur[4] = a4; }
				cur = px0; a0 = px0[0]; a1 = px0[1]; a2 = px0[2]; a3 = px0[3]; a4 = px0[4];
			}
			for (uint64_t m = mTap; m; m &= m - 1ull) {          // sample order = lane order
				const uint32_t l = (uint32_t) __builtin_ctzll(m);
				a0 += bcast(L.x, l) * 1.0f; a1 += bcast(L.y, l) * 1.0f; a2 += bcast(L.z, l) * 1.0f;
				a3 += bcast(alpha, l) * 1.0f;
				a4 += 1.0f;
			}
		} else {
			// a sample on a pixel border, or samples of one slot on different pixels: lane 0 adds this chunk the slow way
			if (cur && lane == 0) { cur[0] = a0; cur[1] = a1; cur[2] = a2; cur[3] = a3; cur[4] = a4; }
			cur = nullptr;
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			if (lane == 0) {
				const uint32_t jEnd = min(j0 + 64u, spp);
				for (uint32_t jj = j0; jj < jEnd; ++jj) {
					const size_t i2 = (size_t) slot * spp + jj;
					const float4 L2 = settled_Li(ps.Li(i2), ps.slot(i2, 2));
					const float4 s2 = ps.spos(i2);
					if (L2.x != L2.x || L2.x < 0.0f || L2.y != L2.y || L2.y < 0.0f || L2.z != L2.z || L2.z < 0.0f) continue;
					const float al2 = (__float_as_uint(L2.w) & F_ALPHA) ? 1.0f : 0.0f;
					const float tx = s2.x - 0.5f - 0, ty = s2.y - 0.5f - 0;
					int x0 = max(cfg.crop_x, (int) ceilf(tx - fsize)), x1 = min((int) floorf(tx + fsize), cfg.crop_x + W - 1);
					int y0 = max(cfg.crop_y, (int) ceilf(ty - fsize)), y1 = min((int) floorf(ty + fsize), cfg.crop_y + H - 1);
					for (int y = y0; y <= y1; ++y) {
						const int iy = min((int) (factor * fabsf(y - ty)), 15);
						for (int x = x0; x <= x1; ++x) {
							const int ix = min((int) (factor * fabsf(x - tx)), 15);
							if (ix == 15 || iy == 15) continue;
							float *q = film + 5 * ((size_t) (y - cfg.crop_y) * W + (x - cfg.crop_x));
							q[0] += L2.x * 1.0f; q[1] += L2.y * 1.0f; q[2] += L2.z * 1.0f; q[3] += al2 * 1.0f; q[4] += 1.0f;
						}
					}
				}
			}
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		}
	}
	if (cur && lane == 0) { cur[0] = a0; cur[1] = a1; cur[2] = a2; cur[3] = a3; cur[4] = a4; }
	if (path_len) {
		for (int off = 32; off > 0; off >>= 1)
			depthSum += __shfl_down(depthSum, off);
		if (lane == 0 && depthSum)
			atomicAdd(path_len, depthSum);
	}
}

// the avgPathLength statistic for passes that do not run k_accumulate (filters wider than a pixel)
__global__ void k_path_lengths(DPaths ps, uint32_t n_paths, unsigned long long *path_len) {
	const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
	unsigned long long d = id < n_paths ? (unsigned long long) __float_as_int(ps.thr(id).w) : 0ull;
	for (int off = 32; off > 0; off >>= 1)
		d += __shfl_down(d, off);
	if (lane_id() == 0 && d)
		atomicAdd(path_len, d);
}

__global__ void k_triad(float4 *a, const float4 *b, const float4 *c, float s, size_t n) {
	const size_t stride = (size_t) gridDim.x * blockDim.x;
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const float4 x = b[i], y = c[i];
		a[i] = make_float4(x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w);
	}
}

// the vector-memory request roof (mtsgpu_gather_roof): every lane loads 16 bytes from its own random element of a
// footprint that fits the L2, the way k_trace walks the kd-tree; same grid shape as k_trace (7 workgroups of 256 per CU)
__global__ __launch_bounds__(256, 7) void k_gather_roof(const uint4 *data, uint32_t mask_elems, int iters, uint32_t *sink) {
	const uint32_t gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	uint32_t acc = 0, x = gw * 0x9E3779B9u + (threadIdx.x & 63u) * 0x85EBCA6Bu + 12345u;
	for (int i = 0; i < iters; ++i) {
		x ^= x << 13; x ^= x >> 17; x ^= x << 5;
		const uint4 v = data[x & mask_elems];
		acc += v.x ^ v.w;
	}
	if (acc == 0xDEADBEEFu) sink[0] = acc;
}

// ---- replay roof of the closest-hit traversal (mtsgpu_replay_roof) ----
// The requests a counting launch recorded for n rays (DQueues::rec), laid out for coalesced reading: 64 rays of similar
// length per batch, entries 4 g .. 4 g + 3 of lane l in the uint4 tr[(batch * cap / 4 + g) * 64 + l] (cap % 4 == 0).
__global__ void k_build_replay(const uint32_t *rec, const uint32_t *rec_len, const uint32_t *order, uint32_t n, uint32_t cap,
                               uint32_t *tr, uint32_t *batch_len) {
	const uint32_t b = blockIdx.x, l = threadIdx.x;       // one wave per batch
	const uint32_t r = b * 64u + l;
	const uint32_t ray = r < n ? order[r] : 0u;
	const uint32_t len = r < n ? min(rec_len[ray], cap) : 0u;
	uint32_t longest = len;
	for (int off = 32; off > 0; off >>= 1) longest = max(longest, (uint32_t) __shfl_xor((int) longest, off));
	longest = (longest + 3u) & ~3u;
	if (l == 0) batch_len[b] = longest;
	uint4 *out = reinterpret_cast<uint4 *>(tr) + (size_t) b * (cap / 4u) * 64u + l;
	for (uint32_t k = 0; k < longest; k += 4u) {
		uint32_t e[4];
		#pragma unroll
		for (uint32_t j = 0; j < 4u; ++j) e[j] = (k + j < len) ? rec[(size_t) ray * cap + k + j] : kReqNone;
		out[(size_t) (k / 4u) * 64u] = make_uint4(e[0], e[1], e[2], e[3]);
	}
}
// The same requests as a pure throughput test: every lane walks the list of its ray and issues one 16-byte load per entry
// from the line the traversal asked for (a single 8-byte node is read as the aligned pair that holds it, the store of the
// hit as a load of its slot), eight in flight per lane, NO dependence between them and no arithmetic -- what the memory
// system (TA, L1, L2, fabric, HBM) needs for this set of lines in this order from this grid.  The traversal itself cannot
// go faster than this however it is written; it goes slower by what its chains of dependent fetches (a descent step needs
// the node before it) and its arithmetic cost on top.  Same grid, same workgroup size and the LDS footprint of
// k_trace<closest>, so the same number of waves is resident.  One coalesced 16-byte read of the list per four requests
// comes on top.
__global__ __launch_bounds__(kTraceBlock, trace_waves_per_simd(0)) void k_replay(const uint2 *nodes, const uint4 *leaf_ta, float4 *paths,
                                                                                const uint32_t *tr, const uint32_t *batch_len,
                                                                                uint32_t n_batches, uint32_t cap, uint32_t zero, uint32_t *sink) {
	__shared__ uint32_t s_pad[kStackLDS + 8][kTraceBlock];
	__shared__ uint4 s_top[kTopPairs ? kTopPairs : 1];
	for (uint32_t t = threadIdx.x; t < kTopPairs; t += kTraceBlock) s_top[t] = reinterpret_cast<const uint4 *>(nodes)[t];
	s_pad[threadIdx.x & 7u][threadIdx.x] = zero;
	__syncthreads();
	const uint32_t lane = lane_id();
	const uint32_t wave = blockIdx.x * (kTraceBlock / 64u) + (threadIdx.x >> 6), n_waves = gridDim.x * (kTraceBlock / 64u);
	uint32_t acc = s_top[threadIdx.x % (kTopPairs ? kTopPairs : 1)].x & s_pad[threadIdx.x & 7u][threadIdx.x];
	const char *bNodes = reinterpret_cast<const char *>(nodes), *bLeaf = reinterpret_cast<const char *>(leaf_ta), *bPaths = reinterpret_cast<const char *>(paths);
	// one entry -> the address of its 16-byte chunk (selects, no branches: the wave issues ONE load instruction per entry)
	auto address = [&](uint32_t e) -> const uint4 * {
		const uint32_t kind = e >> 29, idx = e & 0x1FFFFFFFu;
		const char *base = (kind == kReqLeaf) ? bLeaf : ((kind == kReqRay || kind == kReqHit) ? bPaths : bNodes);
		const size_t off = (kind == kReqNode) ? (size_t) (idx >> 1) * 16u : (size_t) idx * 16u;
		return reinterpret_cast<const uint4 *>(base + off);
	};
	for (uint32_t b = wave; b < n_batches; b += n_waves) {
		const uint32_t groups = batch_len[b] / 4u;
		const uint4 *t = reinterpret_cast<const uint4 *>(tr) + (size_t) b * (cap / 4u) * 64u + lane;
		const uint4 none = make_uint4(kReqNone, kReqNone, kReqNone, kReqNone);
		for (uint32_t g = 0; g < groups; g += 2u) {
			const uint4 c0 = t[(size_t) g * 64u], c1 = (g + 1u < groups) ? t[(size_t) (g + 1u) * 64u] : none;
			const uint32_t e[8] = { c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w };
			uint4 p[8];
			#pragma unroll
			for (int j = 0; j < 8; ++j) {
				p[j] = make_uint4(0u, 0u, 0u, 0u);
				if (e[j] != kReqNone) p[j] = *address(e[j]);
			}
			#pragma unroll
			for (int j = 0; j < 8; ++j) acc ^= p[j].x ^ p[j].y ^ p[j].z ^ p[j].w;
		}
	}
	if (acc == 0xDEADBEEFu) sink[0] = acc;
}
__global__ void k_gather_strided(float4 *dst, const float4 *src, uint32_t n, uint32_t stride) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) dst[i] = src[(size_t) i * stride];
}
__global__ void k_iota_strided(uint32_t *p, uint32_t n, uint32_t stride) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) p[i] = i * stride;
}

__global__ void k_add_film(float *dst, const float *src, size_t n) {
	const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) dst[i] += src[i];
}

// ===========================================================================
// K7b: ImageBlock::putSample for reconstruction filters wider than a pixel (gaussian, ...).
// Like the reference, every tile owns a block with a border (renderproc.cpp:143-144,
// imageblock.h:80-138) that only its own samples splat into; the blocks are then added to the film
// (Film::putImageBlock, mfilm.cpp:118-143).  One lane per block pixel GATHERS the samples that
// reach it, in (tile pixel row-major, sample index) order, so the sums are reproducible.
// ===========================================================================
__global__ __launch_bounds__(256) void k_splat_blocks(DPaths ps, DConfig cfg, const TileMeta *tiles, uint32_t spp,
                                                     int block_size, float *blocks) {
	const TileMeta tm = tiles[blockIdx.x];
	const int border = cfg.filt_border, full = block_size + 2 * border;
	const int fullW = tm.w + 2 * border, fullH = tm.h + 2 * border;
	const float sizeX = cfg.filt_size_x, sizeY = cfg.filt_size_y;
	const float factorX = 15 / sizeX, factorY = 15 / sizeY;      // FILTER_RESOLUTION / size (rfilter.cpp:43-45)
	const int RX = (int) ceilf(sizeX + 0.5f), RY = (int) ceilf(sizeY + 0.5f);
	const float offX = (float) (tm.x0 - border), offY = (float) (tm.y0 - border);
	float *blk = blocks + (size_t) tm.block_index * full * full * 5;
	for (int p = threadIdx.x; p < fullW * fullH; p += blockDim.x) {
		const int yl = p / fullW, xl = p - yl * fullW;
		const int X = tm.x0 - border + xl, Y = tm.y0 - border + yl;
		float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
		if (X >= cfg.crop_x && X < cfg.crop_x + cfg.width && Y >= cfg.crop_y && Y < cfg.crop_y + cfg.height) {
			const int pyLo = max(Y - RY, tm.y0), pyHi = min(Y + RY, tm.y0 + tm.h - 1);
			const int pxLo = max(X - RX, tm.x0), pxHi = min(X + RX, tm.x0 + tm.w - 1);
			for (int py = pyLo; py <= pyHi; ++py)
				for (int px = pxLo; px <= pxHi; ++px) {
					const size_t first = ((size_t) tm.slot_base + (size_t) (py - tm.y0) * tm.w + (px - tm.x0)) * spp;
					for (uint32_t j = 0; j < spp; ++j) {
						const float4 L = settled_Li(ps.Li(first + j), ps.slot(first + j, 2));
						const float4 sp = ps.spos(first + j);
						if (L.x != L.x || L.x < 0.0f || L.y != L.y || L.y < 0.0f || L.z != L.z || L.z < 0.0f)
							continue;                                   // Spectrum::isValid
						const float slx = sp.x - 0.5f - offX, sly = sp.y - 0.5f - offY;
						int xStart = (int) ceilf(slx - sizeX), xEnd = (int) floorf(slx + sizeX);
						int yStart = (int) ceilf(sly - sizeY), yEnd = (int) floorf(sly + sizeY);
						xStart = max(0, xStart); yStart = max(0, yStart);
						xEnd = min(xEnd, fullW - 1); yEnd = min(yEnd, fullH - 1);
						if (xl < xStart || xl > xEnd || yl < yStart || yl > yEnd)
							continue;
						const int ix = min((int) (factorX * fabsf(xl - slx)), 15);
						const int iy = min((int) (factorY * fabsf(yl - sly)), 15);
						const float weight = cfg.filt_values[iy * 16 + ix];
						if (weight == 0.0f)
							continue;
						const float alpha = (__float_as_uint(L.w) & F_ALPHA) ? 1.0f : 0.0f;
						a0 += L.x * weight; a1 += L.y * weight; a2 += L.z * weight;
						a3 += alpha * weight; a4 += weight;
					}
				}
		}
		float *o = blk + 5 * ((size_t) yl * full + xl);
		o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3; o[4] = a4;
	}
}

// Film::putImageBlock for all tiles of one colour (tx%2 + 2*(ty%2)): their bordered blocks are
// disjoint, so plain adds are race-free and the film is bit-reproducible.
__global__ __launch_bounds__(256) void k_add_blocks(DConfig cfg, const TileMeta *tiles, uint32_t n_tiles, uint32_t colour,
                                                   int block_size, const float *blocks, float *film) {
	if (blockIdx.x >= n_tiles)
		return;
	const TileMeta tm = tiles[blockIdx.x];
	if (tm.colour != colour)
		return;
	const int border = cfg.filt_border, full = block_size + 2 * border;
	const int fullW = tm.w + 2 * border, fullH = tm.h + 2 * border;
	const float *blk = blocks + (size_t) tm.block_index * full * full * 5;
	for (int p = threadIdx.x; p < fullW * fullH; p += blockDim.x) {
		const int yl = p / fullW, xl = p - yl * fullW;
		const int X = tm.x0 - border + xl, Y = tm.y0 - border + yl;
		if (X < cfg.crop_x || X >= cfg.crop_x + cfg.width || Y < cfg.crop_y || Y >= cfg.crop_y + cfg.height)
			continue;                                                     // outside the crop region (mfilm.cpp:123-135)
		const float *b = blk + 5 * ((size_t) yl * full + xl);
		float *o = film + 5 * ((size_t) (Y - cfg.crop_y) * cfg.width + (X - cfg.crop_x));
		o[0] += b[0]; o[1] += b[1]; o[2] += b[2]; o[3] += b[3]; o[4] += b[4];
	}
}

// ===========================================================================
// launchers
// ===========================================================================
static inline unsigned blocks_for(size_t n, unsigned bs) { return (unsigned) ((n + bs - 1) / bs); }

void launch_fill_u32(hipStream_t s, uint32_t *p, uint32_t v, size_t n) {
	if (n) hipLaunchKernelGGL(k_fill_u32, dim3(blocks_for(n, 256)), dim3(256), 0, s, p, v, n);
}
void launch_iota(hipStream_t s, uint32_t *p, uint32_t n) {
	if (n) hipLaunchKernelGGL(k_iota, dim3(blocks_for(n, 256)), dim3(256), 0, s, p, n);
}

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

void launch_bsdf_eval(hipStream_t s, uint32_t type, const float *params, int op, uint32_t n, const float *queries, float *out) {
	BsdfParams p;
	for (int k = 0; k < kBsdfNParams; ++k) p.v[k] = params[k];
	if (n) hipLaunchKernelGGL(k_bsdf_eval, dim3(blocks_for(n, 256)), dim3(256), 0, s, type, p, op, n, queries, out);
}

void launch_generate(hipStream_t s, const DScene &sc, const DPaths &ps, const DConfig &cfg,
                     const uint32_t *pixel_list, uint32_t n_slots, const uint32_t *explicit_samples,
                     uint32_t n_paths, uint32_t *queue) {
	if (n_paths) hipLaunchKernelGGL(k_generate, dim3(blocks_for(n_paths, kGenBlock)), dim3(kGenBlock), 0, s, sc, ps, cfg,
	                                pixel_list, n_slots, explicit_samples, n_paths, queue);
}

template <int MODE, bool COUNT, bool BIN>
static void launch_trace_t(hipStream_t s, const DScene &sc, const DPaths &ps, const DQueues &q, const uint32_t *queue, uint32_t n,
                           const uint32_t *n_dev) {
	// persistent grid: enough workgroups to fill every CU, never more than there are rays (trace_plan); when only the
	// device knows the count, the grid is sized for the upper bound n and the surplus workgroups exit at once
	unsigned blocks = trace_plan(n, MODE, q).blocks;
	if (n_dev) {
		// any count up to n: the narrowest batches need the most workgroups
		const unsigned minBatch = (q.tune_batch >= 1 && q.tune_batch <= 64) ? q.tune_batch : (q.coherent ? 64u : 8u);
		unsigned perCu = trace_blocks_per_cu(MODE);
		if (q.tune_blocks_per_cu && q.tune_blocks_per_cu < perCu) perCu = q.tune_blocks_per_cu;
		blocks = std::min<unsigned>(blocks_for(n, minBatch * (kTraceBlock / 64)), q.n_cus * perCu);
	}
	if (!blocks) return;
	hipLaunchKernelGGL((k_trace<MODE, COUNT, BIN>), dim3(blocks), dim3(kTraceBlock), 0, s, trace_scene(sc), ps, q, queue, n, n_dev);
}

void launch_trace(hipStream_t s, int mode, bool count, bool bin, const DScene &sc, const DPaths &ps,
                  const DQueues &q, const uint32_t *queue, uint32_t n, bool coherent, const uint32_t *n_dev) {
	if (!n) return;
	DQueues qq = q;
	qq.coherent = coherent ? 1u : 0u;
	if (n_dev)
		qq.force_static = 1u;        // no dynamically claimed batches: the material-queue segments cannot overflow then
	if (mode == 0) {
		if (bin) { if (count) launch_trace_t<0, true, true>(s, sc, ps, qq, queue, n, n_dev); else launch_trace_t<0, false, true>(s, sc, ps, qq, queue, n, n_dev); }
		else     { if (count) launch_trace_t<0, true, false>(s, sc, ps, qq, queue, n, n_dev); else launch_trace_t<0, false, false>(s, sc, ps, qq, queue, n, n_dev); }
	} else if (mode == 1) {
		if (count) launch_trace_t<1, true, false>(s, sc, ps, qq, queue, n, n_dev); else launch_trace_t<1, false, false>(s, sc, ps, qq, queue, n, n_dev);
	} else {
		if (count) launch_trace_t<2, true, false>(s, sc, ps, qq, queue, n, n_dev); else launch_trace_t<2, false, false>(s, sc, ps, qq, queue, n, n_dev);
	}
}

void launch_prep(hipStream_t s, const uint32_t *cur, uint32_t *next_set, BinView *views_dev, uint32_t bin_seg_cap,
                 unsigned long long *dev_stats) {
	hipLaunchKernelGGL(k_prep, dim3(1), dim3(256), 0, s, cur, next_set, views_dev, bin_seg_cap, dev_stats);
}

void launch_shade(hipStream_t s, int bin, const DScene &sc, const DPaths &ps, const DConfig &cfg,
                  const DQueues &q, const BinView &view, const BinView *views_dev, uint32_t n_bound, const uint32_t *bin_ids) {
	const uint32_t n = views_dev ? n_bound : view.prefix[kBinShards];
	if (!n) return;
	if (!bin_ids) bin_ids = q.bin(bin);
	const dim3 g(blocks_for(n, kShadeBlock)), b(kShadeBlock);
	#define MG_SHADE(BT) do { if (cfg.dr_mode != 0) hipLaunchKernelGGL((k_shade<BT, true>), g, b, 0, s, sc, ps, cfg, q, view, views_dev, bin_ids); \
	                          else hipLaunchKernelGGL((k_shade<BT, false>), g, b, 0, s, sc, ps, cfg, q, view, views_dev, bin_ids); } while (0)
	switch (bin) {
		case 0: MG_SHADE(0); break;
		case 1: MG_SHADE(1); break;
		case 2: MG_SHADE(2); break;
		case 3: MG_SHADE(3); break;
		case 4: MG_SHADE(4); break;
		case 5: MG_SHADE(5); break;
		case 6: MG_SHADE(6); break;
		case 7: MG_SHADE(7); break;
		default: MG_SHADE(kNumBsdfTypes); break;
	}
	#undef MG_SHADE
}

void launch_shade_all(hipStream_t s, const DScene &sc, const DPaths &ps, const DConfig &cfg, const DQueues &q,
                      const BinView *views_dev, uint32_t bin_mask, uint32_t n_bound) {
	if (!n_bound || !bin_mask) return;
	// every bin rounds its size up to whole workgroups
	const unsigned blocks = blocks_for(n_bound, kShadeBlock) + (unsigned) __builtin_popcount(bin_mask);
	hipLaunchKernelGGL(k_shade_all, dim3(blocks), dim3(kShadeBlock), 0, s, sc, ps, cfg, q, views_dev, bin_mask);
}

void launch_accumulate(hipStream_t s, const DPaths &ps, const DConfig &cfg, uint32_t n_slots,
                       uint32_t spp_per_slot, float *film, unsigned long long *path_len) {
	if (!n_slots) return;
	// few pixels with many samples each: a wave per pixel (a lane per pixel leaves the chip empty and chains its loads)
	if (spp_per_slot >= 256u && n_slots <= (1u << 15))      // C4 pass (18 k pixels x 4096): 4.2 -> 1.8 ms; 65 k pixels x 1024: the lane form wins (1.7 against 2.2 ms)
		hipLaunchKernelGGL(k_accumulate_wave, dim3(blocks_for(n_slots, 4)), dim3(256), 0, s, ps, cfg, n_slots, spp_per_slot, film, path_len);
	else
		hipLaunchKernelGGL(k_accumulate, dim3(blocks_for(n_slots, 256)), dim3(256), 0, s, ps, cfg, n_slots, spp_per_slot, film, path_len);
}
void launch_path_lengths(hipStream_t s, const DPaths &ps, uint32_t n_paths, unsigned long long *path_len) {
	if (n_paths) hipLaunchKernelGGL(k_path_lengths, dim3(blocks_for(n_paths, 256)), dim3(256), 0, s, ps, n_paths, path_len);
}
void launch_triad(hipStream_t s, float4 *a, const float4 *b, const float4 *c, float scale, size_t n, unsigned blocks) {
	if (n) hipLaunchKernelGGL(k_triad, dim3(blocks ? blocks : 256u * 16u), dim3(256), 0, s, a, b, c, scale, n);
}
void launch_gather_roof(hipStream_t s, const uint4 *data, uint32_t mask_elems, int iters, unsigned blocks, uint32_t *sink) {
	hipLaunchKernelGGL(k_gather_roof, dim3(blocks), dim3(256), 0, s, data, mask_elems, iters, sink);
}
void launch_build_replay(hipStream_t s, const uint32_t *rec, const uint32_t *rec_len, const uint32_t *order, uint32_t n, uint32_t cap,
                         uint32_t *tr, uint32_t *batch_len) {
	if (n) hipLaunchKernelGGL(k_build_replay, dim3(blocks_for(n, 64)), dim3(64), 0, s, rec, rec_len, order, n, cap, tr, batch_len);
}
void launch_replay(hipStream_t s, const DScene &sc, const DPaths &ps, const DQueues &q, const uint32_t *tr, const uint32_t *batch_len,
                   uint32_t n_batches, uint32_t cap, uint32_t zero, uint32_t *sink) {
	const unsigned blocks = std::min<unsigned>(blocks_for(n_batches, kTraceBlock / 64), q.n_cus * trace_blocks_per_cu(0));
	if (blocks) hipLaunchKernelGGL(k_replay, dim3(blocks), dim3(kTraceBlock), 0, s, sc.nodes, sc.leaf_ta, ps.base, tr, batch_len, n_batches, cap, zero, sink);
}
void launch_gather_strided(hipStream_t s, float4 *dst, const float4 *src, uint32_t n, uint32_t stride) {
	if (n) hipLaunchKernelGGL(k_gather_strided, dim3(blocks_for(n, 256)), dim3(256), 0, s, dst, src, n, stride);
}
void launch_iota_strided(hipStream_t s, uint32_t *p, uint32_t n, uint32_t stride) {
	if (n) hipLaunchKernelGGL(k_iota_strided, dim3(blocks_for(n, 256)), dim3(256), 0, s, p, n, stride);
}
void launch_add_film(hipStream_t s, float *dst, const float *src, size_t n) {
	if (n) hipLaunchKernelGGL(k_add_film, dim3(blocks_for(n, 256)), dim3(256), 0, s, dst, src, n);
}

} // namespace mg

namespace mg {
void launch_splat_blocks(hipStream_t s, const DPaths &ps, const DConfig &cfg, const TileMeta *tiles, uint32_t n_tiles,
                         uint32_t spp, int block_size, float *blocks) {
	if (n_tiles) hipLaunchKernelGGL(k_splat_blocks, dim3(n_tiles), dim3(256), 0, s, ps, cfg, tiles, spp, block_size, blocks);
}
void launch_add_blocks(hipStream_t s, const DConfig &cfg, const TileMeta *tiles, uint32_t n_tiles, uint32_t colour,
                       int block_size, const float *blocks, float *film) {
	if (n_tiles) hipLaunchKernelGGL(k_add_blocks, dim3(n_tiles), dim3(256), 0, s, cfg, tiles, n_tiles, colour, block_size, blocks, film);
}
} // namespace mg
