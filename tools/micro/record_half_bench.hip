// record_half_bench -- does gfx950 move half lines?  Random 128-byte records of an 8 GiB buffer (32 M distinct ones per launch),
// but only slots [R0, R0+RK) of every record are read and slots [W0, W0+WK) written (16-byte slots, RK / WK lanes per record, all
// lanes of a wave busy in every request, every load independent of the others: a throughput test).
// The first version of this file (profiles/r05s_record_half_line_microbench.txt) masked the slots at run time; the compiler then
// waits for every load before it issues the next, and that kernel measured latency: its "a half costs what the line costs" was an
// artefact.
// hipcc --offload-arch=gfx950 -O3 tools/micro/record_half_bench.hip -o tools/micro/record_half_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include <random>
#include <algorithm>
typedef float nt_f4 __attribute__((ext_vector_type(4)));
// WSKIP: with WK == 8, slots >= 8 - WSKIP stay unwritten (a partial line written by eight lanes)
template <int R0, int RK, int W0, int WK, int WSKIP>
__global__ __launch_bounds__(512) void k_rw(nt_f4 *recs, const uint32_t *ids, uint32_t n, nt_f4 *sink) {
	const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t id = gtid < n ? ids[gtid] : 0u;
	nt_f4 sum = nt_f4{1, 1, 1, 1};
	if (RK > 0) {
		#pragma unroll
		for (int r = 0; r < RK; ++r) {
			const uint32_t sid = (uint32_t) __shfl((int) id, (int) (r * (64 / (RK ? RK : 1)) + lane / (RK ? RK : 1)));
			sum += __builtin_nontemporal_load(&recs[(size_t) sid * 8 + R0 + lane % (RK ? RK : 1)]);
		}
	}
	if (WK == 0) { if (sum.x + sum.y + sum.z + sum.w == 12345.678f) sink[0] = sum; return; }
	#pragma unroll
	for (int r = 0; r < WK; ++r) {
		const uint32_t sid = (uint32_t) __shfl((int) id, (int) (r * (64 / (WK ? WK : 1)) + lane / (WK ? WK : 1)));
		const uint32_t slot = lane % (WK ? WK : 1);
		if (WSKIP == 0 || slot < 8u - WSKIP) __builtin_nontemporal_store(sum, &recs[(size_t) sid * 8 + W0 + slot]);
	}
}
template <int R0, int RK, int W0, int WK, int WSKIP = 0>
static void run(const char *what, nt_f4 *recs, const uint32_t *ids, uint32_t n, nt_f4 *sink) {
	hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
	float best = 1e30f;
	for (int rep = 0; rep < 3; ++rep) {
		(void) hipEventRecord(e0);
		hipLaunchKernelGGL((k_rw<R0, RK, W0, WK, WSKIP>), dim3(n / 512), dim3(512), 0, 0, recs, ids, n, sink);
		(void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
		float ms; (void) hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
	}
	const double bytes = 16.0 * (RK + WK - WSKIP) * n;
	printf("%-72s %7.3f ms  %6.0f GB/s of bytes asked for  %.4f ns per record\n", what, best, bytes / (best * 1e-3) / 1e9, best * 1e6 / n);
}
int main() {
	setvbuf(stdout, nullptr, _IONBF, 0);
	const uint32_t nRec = 64u << 20, n = 32u << 20;
	nt_f4 *recs, *sink; uint32_t *ids;
	if (hipMalloc(&recs, (size_t) nRec * 128) != hipSuccess) { printf("alloc failed\n"); return 1; }
	(void) hipMalloc(&ids, (size_t) n * 4); (void) hipMalloc(&sink, 64);
	(void) hipMemset(recs, 0, (size_t) nRec * 128);
	std::vector<uint32_t> h(nRec);
	std::mt19937_64 rng(1);
	for (uint32_t i = 0; i < nRec; ++i) h[i] = i;
	for (uint32_t i = 0; i < n; ++i) { const uint32_t j = i + (uint32_t) (rng() % (nRec - i)); std::swap(h[i], h[j]); }
	(void) hipMemcpy(ids, h.data(), (size_t) n * 4, hipMemcpyHostToDevice);
	run<0, 8, 0, 0>("read 128 B", recs, ids, n, sink);
	run<0, 4, 0, 0>("read 64 B (slots 0-3)", recs, ids, n, sink);
	run<4, 4, 0, 0>("read 64 B (slots 4-7)", recs, ids, n, sink);
	run<2, 4, 0, 0>("read 64 B (slots 2-5: across the halves)", recs, ids, n, sink);
	run<0, 2, 0, 0>("read 32 B (slots 0-1)", recs, ids, n, sink);
	run<0, 1, 0, 0>("read 16 B (slot 0)", recs, ids, n, sink);
	run<0, 0, 0, 8>("write 128 B", recs, ids, n, sink);
	run<0, 0, 0, 8, 1>("write 112 B (slots 0-6)", recs, ids, n, sink);
	run<0, 0, 0, 4>("write 64 B (slots 0-3)", recs, ids, n, sink);
	run<0, 0, 2, 4>("write 64 B (slots 2-5: across the halves)", recs, ids, n, sink);
	run<0, 0, 0, 2>("write 32 B (slots 0-1)", recs, ids, n, sink);
	run<0, 0, 2, 1>("write 16 B (slot 2: a parked direct-light term)", recs, ids, n, sink);
	run<0, 8, 0, 8>("read 128 B, write 128 B (k_shade)", recs, ids, n, sink);
	run<0, 8, 0, 8, 1>("read 128 B, write 112 B", recs, ids, n, sink);
	run<0, 8, 0, 4>("read 128 B, write 64 B (slots 0-3)", recs, ids, n, sink);
	run<0, 4, 0, 4>("read 64 B, write 64 B (slots 0-3)", recs, ids, n, sink);
	run<4, 4, 4, 4>("read 64 B, write 64 B (slots 4-7)", recs, ids, n, sink);
	run<0, 4, 4, 4>("read 64 B (slots 0-3), write 64 B (slots 4-7)", recs, ids, n, sink);
	run<0, 2, 2, 1>("read 32 B (slots 0-1), write 16 B (slot 2)  (k_trace until round 4)", recs, ids, n, sink);
	return 0;
}
