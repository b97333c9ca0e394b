// sampler.h -- the per-path sampler state and Sampler::next1D / next2D / next2DArray as the kernels evaluate them
// (k_generate, k_shade, k_sampler_values).
#pragma once
#include "kdevice.h"

namespace mg {

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

} // namespace mg
