// film.hip -- the two ends of a pass: K1 camera samples (k_generate) and K7 ImageBlock::putSample (k_accumulate*,
// k_splat_blocks / k_add_blocks), plus the fill / iota / film-sum utilities.
#include "sampler.h"

namespace mg {

__global__ void k_fill_u32(uint32_t *p, uint32_t v, size_t n) {
	size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) p[i] = v;
}
__global__ void k_iota(uint32_t *p, uint32_t n) {
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) p[i] = i;
}

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

void launch_fill_u32(hipStream_t s, uint32_t *p, uint32_t v, size_t n) {
	if (n) hipLaunchKernelGGL(k_fill_u32, dim3(blocks_for(n, 256)), dim3(256), 0, s, p, v, n);
}
void launch_iota(hipStream_t s, uint32_t *p, uint32_t n) {
	if (n) hipLaunchKernelGGL(k_iota, dim3(blocks_for(n, 256)), dim3(256), 0, s, p, n);
}

void launch_generate(hipStream_t s, const DScene &sc, const DPaths &ps, const DConfig &cfg,
                     const uint32_t *pixel_list, uint32_t n_slots, const uint32_t *explicit_samples,
                     uint32_t n_paths, uint32_t *queue) {
	if (n_paths) hipLaunchKernelGGL(k_generate, dim3(blocks_for(n_paths, kGenBlock)), dim3(kGenBlock), 0, s, sc, ps, cfg,
	                                pixel_list, n_slots, explicit_samples, n_paths, queue);
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
void launch_add_film(hipStream_t s, float *dst, const float *src, size_t n) {
	if (n) hipLaunchKernelGGL(k_add_film, dim3(blocks_for(n, 256)), dim3(256), 0, s, dst, src, n);
}

void launch_splat_blocks(hipStream_t s, const DPaths &ps, const DConfig &cfg, const TileMeta *tiles, uint32_t n_tiles,
                         uint32_t spp, int block_size, float *blocks) {
	if (n_tiles) hipLaunchKernelGGL(k_splat_blocks, dim3(n_tiles), dim3(256), 0, s, ps, cfg, tiles, spp, block_size, blocks);
}
void launch_add_blocks(hipStream_t s, const DConfig &cfg, const TileMeta *tiles, uint32_t n_tiles, uint32_t colour,
                       int block_size, const float *blocks, float *film) {
	if (n_tiles) hipLaunchKernelGGL(k_add_blocks, dim3(n_tiles), dim3(256), 0, s, cfg, tiles, n_tiles, colour, block_size, blocks, film);
}

} // namespace mg
