// measure.hip -- measurement kernels, not part of a frame: streaming triad, random-gather roof, the replay of a recorded
// request stream (mtsgpu_replay_roof), strided helpers of the test hooks.
#include "kdevice.h"

namespace mg {

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

} // namespace mg
