// record_gather_bench -- read + write 128-byte records of a large buffer the way k_shade does (eight lanes per record,
// whole lines), with ids that are random, clustered in runs of consecutive ids, or sequential.
// hipcc --offload-arch=gfx950 -O3 tools/micro/record_gather_bench.hip -o tools/micro/record_gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include <random>
#include <algorithm>
typedef float nt_f4 __attribute__((ext_vector_type(4)));
template <int BS> __global__ __launch_bounds__(BS) void k_rw(nt_f4 *recs, const uint32_t *ids, uint32_t n, int write_mode, nt_f4 *dst) {
	const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t lane = threadIdx.x & 63u, sub = lane & 7u, grp = lane >> 3;
	uint32_t id = gtid < n ? ids[gtid] : 0u;
	nt_f4 acc[8];
	#pragma unroll
	for (int r = 0; r < 8; ++r) {
		const uint32_t sid = (uint32_t) __shfl((int) id, (int) (grp + 8u * r));
		acc[r] = __builtin_nontemporal_load(&recs[(size_t) sid * 8 + sub]);
	}
	#pragma unroll
	for (int r = 0; r < 8; ++r) { acc[r].x += 1.0f; }
	#pragma unroll
	for (int r = 0; r < 8; ++r) {
		const uint32_t src = grp + 8u * r;
		const uint32_t sid = (uint32_t) __shfl((int) id, (int) src);
		// write_mode 0: back to the record's home (random); 1: to the slot of the reading lane (sequential)
		nt_f4 *out = write_mode == 0 ? &recs[(size_t) sid * 8 + sub] : &dst[((size_t) (gtid - lane) + src) * 8 + sub];
		__builtin_nontemporal_store(acc[r], out);
	}
}
int main() {
	setvbuf(stdout, nullptr, _IONBF, 0);
	const uint32_t nRec = 64u << 20;            // 8 GiB of records
	const uint32_t n = 32u << 20;               // records touched per launch
	nt_f4 *recs, *dst; uint32_t *ids;
	if (hipMalloc(&recs, (size_t) nRec * 128) != hipSuccess || hipMalloc(&dst, (size_t) n * 128) != hipSuccess) { printf("alloc failed\n"); return 1; }
	hipMalloc(&ids, (size_t) n * 4);
	hipMemset(recs, 0, (size_t) nRec * 128);
	std::vector<uint32_t> h(n);
	std::mt19937_64 rng(1);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (int run : { 1, 8, 32, 64, 512, 0 }) {
		// run = 0: sequential; otherwise runs of `run` consecutive records at random bases (distinct bases)
		if (run == 0) for (uint32_t i = 0; i < n; ++i) h[i] = i;
		else {
			const uint32_t nRuns = n / run, slots = nRec / run;
			std::vector<uint32_t> bases(slots);
			for (uint32_t i = 0; i < slots; ++i) bases[i] = i;
			for (uint32_t i = 0; i < nRuns; ++i) { const uint32_t j = i + (uint32_t) (rng() % (slots - i)); std::swap(bases[i], bases[j]); }
			for (uint32_t i = 0; i < nRuns; ++i) for (int k = 0; k < run; ++k) h[(size_t) i * run + k] = bases[i] * run + k;
		}
		hipMemcpy(ids, h.data(), (size_t) n * 4, hipMemcpyHostToDevice);
		for (int wm = 0; wm < 2; ++wm) for (int bs : { 256, 512, 1024 }) {
			if (bs != 512 && run != 1) continue;          // the workgroup-size comparison for random ids only
			float best = 1e30f;
			for (int rep = 0; rep < 3; ++rep) {
				hipEventRecord(e0);
				if (bs == 256) hipLaunchKernelGGL(k_rw<256>, dim3(n / 256), dim3(256), 0, 0, recs, ids, n, wm, dst);
				else if (bs == 512) hipLaunchKernelGGL(k_rw<512>, dim3(n / 512), dim3(512), 0, 0, recs, ids, n, wm, dst);
				else hipLaunchKernelGGL(k_rw<1024>, dim3(n / 1024), dim3(1024), 0, 0, recs, ids, n, wm, dst);
				hipEventRecord(e1); hipEventSynchronize(e1);
				float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
			}
			printf("runs of %3d consecutive records, write %-10s, workgroups of %4d: %7.2f ms  %6.0f GB/s (read + write)  %.3f ns per record\n",
			       run, wm == 0 ? "home" : "sequential", bs, best, 2.0 * n * 128 / (best * 1e-3) / 1e9, best * 1e6 / n);
		}
	}
	return 0;
}
