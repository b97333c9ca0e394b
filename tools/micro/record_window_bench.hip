// record_window_bench -- what would k_shade's record reads cost if records moved with their path from queue to queue?
// Today a path's 128-byte record stays at its id, and the ids of a material bin are in the order the traversal kernel retired
// them: random lines of the pass's whole record array.  Records written in next-queue order by one k_shade launch and read by the
// next one in retirement order would be random only inside the window of rays in flight in the traversal kernel (6144 waves x 64
// lanes = 393 216 rays = 48 MiB of records).  Reads 32 M records of an 8 GiB buffer in orders of growing locality, and writes them
// back either in place or as a stream.
// hipcc --offload-arch=gfx950 -O3 tools/micro/record_window_bench.hip -o tools/micro/record_window_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include <random>
#include <algorithm>
typedef float nt_f4 __attribute__((ext_vector_type(4)));
// mode 0: read only; 1: write back in place; 2: write to out[] in launch order
__global__ __launch_bounds__(512) void k_rw(nt_f4 *recs, nt_f4 *out, const uint32_t *ids, uint32_t n, int mode) {
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
	if (mode == 0) {
		float s = 0; for (int r = 0; r < 8; ++r) s += acc[r].x + acc[r].y;
		if (s == 12345.678f) out[0] = acc[0];
		return;
	}
	const uint32_t wbase = gtid - lane;
	#pragma unroll
	for (int r = 0; r < 8; ++r) {
		const uint32_t sid = (uint32_t) __shfl((int) id, (int) (grp + 8u * r));
		if (mode == 1) __builtin_nontemporal_store(acc[r], &recs[(size_t) sid * 8 + sub]);
		else __builtin_nontemporal_store(acc[r], &out[(size_t) (wbase + grp + 8u * r) * 8 + sub]);
	}
}
int main() {
	setvbuf(stdout, nullptr, _IONBF, 0);
	const uint32_t nRec = 64u << 20, n = 32u << 20;
	nt_f4 *recs, *out; uint32_t *ids;
	if (hipMalloc(&recs, (size_t) nRec * 128) != hipSuccess || hipMalloc(&out, (size_t) n * 128) != hipSuccess) { printf("alloc failed\n"); return 1; }
	(void) hipMalloc(&ids, (size_t) n * 4);
	(void) hipMemset(recs, 0, (size_t) nRec * 128); (void) hipMemset(out, 0, (size_t) n * 128);
	std::vector<uint32_t> h(nRec);
	std::mt19937_64 rng(1);
	hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
	struct Order { const char *what; uint32_t window; } orders[] = {
		{ "random over the whole 8 GiB array (today)", 0 },
		{ "random inside windows of 393 216 records (48 MiB: the rays in flight)", 393216 },
		{ "random inside windows of 32 768 records (4 MiB)", 32768 },
		{ "random inside windows of 512 records (one workgroup)", 512 },
		{ "in order (a stream)", 1 },
	};
	const char *modes[] = { "no write", "written back in place", "written as a stream" };
	for (const Order &o : orders) {
		for (uint32_t i = 0; i < nRec; ++i) h[i] = i;
		if (o.window == 0) { for (uint32_t i = 0; i < n; ++i) { const uint32_t j = i + (uint32_t) (rng() % (nRec - i)); std::swap(h[i], h[j]); } }
		else if (o.window > 1) {
			for (uint32_t b = 0; b < n; b += o.window) { const uint32_t e = std::min(n, b + o.window); for (uint32_t i = b; i + 1 < e; ++i) { const uint32_t j = i + (uint32_t) (rng() % (e - i)); std::swap(h[i], h[j]); } }
		}
		(void) hipMemcpy(ids, h.data(), (size_t) n * 4, hipMemcpyHostToDevice);
		for (int mode = 0; mode < 3; ++mode) {
			float best = 1e30f;
			for (int rep = 0; rep < 3; ++rep) {
				(void) hipEventRecord(e0);
				hipLaunchKernelGGL(k_rw, dim3(n / 512), dim3(512), 0, 0, recs, out, ids, n, mode);
				(void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
				float ms; (void) hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
			}
			printf("%-75s %-24s %7.3f ms  %.4f ns per record\n", o.what, modes[mode], best, best * 1e6 / n);
		}
	}
	return 0;
}
