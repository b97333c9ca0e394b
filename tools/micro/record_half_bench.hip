// record_half_bench -- does gfx950 move half lines?  Random 128-byte records of an 8 GiB buffer, eight lanes per record as
// in k_shade / record_gather_bench, but only the first `slots` 16-byte slots of every record are read and written back.
// hipcc --offload-arch=gfx950 -O3 tools/micro/record_half_bench.hip -o tools/micro/record_half_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include <random>
#include <algorithm>
typedef float nt_f4 __attribute__((ext_vector_type(4)));
// rmask / wmask: bit s set = slot s of every record is read / written
__global__ __launch_bounds__(512) void k_rw(nt_f4 *recs, const uint32_t *ids, uint32_t n, uint32_t rmask, uint32_t wmask) {
	const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t lane = threadIdx.x & 63u, sub = lane & 7u, grp = lane >> 3;
	uint32_t id = gtid < n ? ids[gtid] : 0u;
	nt_f4 acc[8];
	#pragma unroll
	for (int r = 0; r < 8; ++r) {
		const uint32_t sid = (uint32_t) __shfl((int) id, (int) (grp + 8u * r));
		acc[r] = nt_f4{0, 0, 0, 0};
		if ((rmask >> sub) & 1u) acc[r] = __builtin_nontemporal_load(&recs[(size_t) sid * 8 + sub]);
	}
	#pragma unroll
	for (int r = 0; r < 8; ++r) { acc[r].x += 1.0f; }
	#pragma unroll
	for (int r = 0; r < 8; ++r) {
		const uint32_t sid = (uint32_t) __shfl((int) id, (int) (grp + 8u * r));
		if ((wmask >> sub) & 1u) __builtin_nontemporal_store(acc[r], &recs[(size_t) sid * 8 + sub]);
	}
}
int main() {
	setvbuf(stdout, nullptr, _IONBF, 0);
	const uint32_t nRec = 64u << 20, n = 32u << 20;
	nt_f4 *recs; uint32_t *ids;
	if (hipMalloc(&recs, (size_t) nRec * 128) != hipSuccess) { printf("alloc failed\n"); return 1; }
	hipMalloc(&ids, (size_t) n * 4);
	hipMemset(recs, 0, (size_t) nRec * 128);
	std::vector<uint32_t> h(nRec);
	std::mt19937_64 rng(1);
	for (uint32_t i = 0; i < nRec; ++i) h[i] = i;
	for (uint32_t i = 0; i < n; ++i) { const uint32_t j = i + (uint32_t) (rng() % (nRec - i)); std::swap(h[i], h[j]); }
	hipMemcpy(ids, h.data(), (size_t) n * 4, hipMemcpyHostToDevice);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	struct Case { const char *what; uint32_t r, w; } cases[] = {
		{ "read 128 B, write 128 B (k_shade today)", 0xFF, 0xFF },
		{ "read 112 B (slots 0-6), write 112 B", 0x7F, 0x7F },
		{ "read 112 B (slots 0-6), write 128 B", 0x7F, 0xFF },
		{ "read 64 B (slots 0-3), write 64 B (slots 0-3)", 0x0F, 0x0F },
		{ "read 64 B (slots 4-7), write 64 B (slots 4-7)", 0xF0, 0xF0 },
		{ "read 128 B, write 64 B (slots 0-3)", 0xFF, 0x0F },
		{ "read 32 B (slots 0-1), write 16 B (slot 2)  (k_trace's record traffic)", 0x03, 0x04 },
		{ "read 128 B, no write", 0xFF, 0x00 },
		{ "read 64 B (slots 0-3), no write", 0x0F, 0x00 },
		{ "read 32 B (slots 0-1), no write", 0x03, 0x00 },
		{ "read 48 B (slots 3, 4, 7: k_accumulate), no write", 0x98, 0x00 },
	};
	for (const Case &c : cases) {
		float best = 1e30f;
		for (int rep = 0; rep < 3; ++rep) {
			hipEventRecord(e0);
			hipLaunchKernelGGL(k_rw, dim3(n / 512), dim3(512), 0, 0, recs, ids, n, c.r, c.w);
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
		}
		const double bytes = 16.0 * (__builtin_popcount(c.r) + __builtin_popcount(c.w)) * n;
		printf("%-75s %7.3f ms  %6.0f GB/s of bytes asked for  %.4f ns per record\n", c.what, best, bytes / (best * 1e-3) / 1e9, best * 1e6 / n);
	}
	return 0;
}
