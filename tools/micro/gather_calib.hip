// gather_calib -- known-byte-count launches for calibrating rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ* on gfx950 in the access
// shapes of k_trace (MI355X_MICROARCH.md, "HBM": widths other than the 16-B-per-lane stream are uncalibrated).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/gather_calib.hip -o tools/micro/gather_calib
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -o p -- tools/micro/gather_calib
// Every launch touches every 128-byte line of its footprint exactly ONCE (line index = i * odd mod n_lines, a bijection), so
// the bytes a launch must move are known: n_lines x 128 if the memory system moves whole lines, n_lines x 64 / 32 if it moves
// the sectors that hold the bytes asked for.  Shapes (the kernel name carries the shape, the dispatch order the footprint):
//   k_g16     one 16-byte load per lane from its own line          (sibling pair, record head: k_trace's common request)
//   k_g8      one  8-byte load per lane from its own line          (single node at a pop)
//   k_g48     three 16-byte loads per lane: a 48-byte record at a random 48-byte-aligned index (head + two tail halves; three
//             of eight records straddle two lines) -- bytes by line count, computed on the host
//   k_stream  16 bytes per lane, consecutive lanes consecutive addresses (the guide's calibrated case: FETCH_SIZE = 1/2)
// Footprints 32 MB, 150 MB, 2 GB (inside the L2 + Infinity Cache / inside the Infinity Cache / beyond it); each (shape,
// footprint) is launched kReps times back to back so that residency shows as a difference between the first launch and the rest.
// Prints one line per launch group: shape, footprint, lines touched, the three candidate byte counts, time and lines per ns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kReps = 4;
constexpr unsigned long long kOdd = 1000003ull;      // prime, coprime to every footprint's line count below

__global__ __launch_bounds__(256) void k_g16(const uint4 *base, unsigned long long n_lines, uint32_t *sink) {
	const unsigned long long i = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_lines) return;
	const unsigned long long line = (i * kOdd) % n_lines;
	const uint4 v = base[line * 8ull + (i & 7ull)];        // 16 bytes at a varying offset inside the line
	if ((v.x ^ v.y ^ v.z ^ v.w) == 0xDEADBEEFu) sink[0] = v.x;
}
__global__ __launch_bounds__(256) void k_g8(const uint2 *base, unsigned long long n_lines, uint32_t *sink) {
	const unsigned long long i = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_lines) return;
	const unsigned long long line = (i * kOdd) % n_lines;
	const uint2 v = base[line * 16ull + (i & 15ull)];
	if ((v.x ^ v.y) == 0xDEADBEEFu) sink[0] = v.x;
}
// n_recs records of 48 bytes tile the footprint; record r = i * kOdd mod n_recs: every record once
__global__ __launch_bounds__(256) void k_g48(const uint4 *base, unsigned long long n_recs, uint32_t *sink) {
	const unsigned long long i = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_recs) return;
	const unsigned long long r = (i * kOdd) % n_recs;
	const uint4 a = base[r * 3ull], b = base[r * 3ull + 1ull], c = base[r * 3ull + 2ull];
	if ((a.x ^ b.y ^ c.z) == 0xDEADBEEFu) sink[0] = a.x;
}
__global__ __launch_bounds__(256) void k_stream(const uint4 *base, unsigned long long n16, uint32_t *sink) {
	const unsigned long long i = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n16) return;
	const uint4 v = base[i];
	if ((v.x ^ v.y ^ v.z ^ v.w) == 0xDEADBEEFu) sink[0] = v.x;
}
// The SUSTAINED line rate: the launches above are over in 6-350 us (one load per thread), their rates are ramp-up limited.  Here a
// chip-filling grid keeps eight independent 16-byte gathers per lane in flight for `iters` rounds over distinct random lines of the
// footprint (line = hash(i) mod n_lines; with 2^k lines and an odd multiplier every line is equally likely, repeats are rare and
// far apart): the rate the L2-miss path sustains for random lines.
__global__ __launch_bounds__(256) void k_sustained(const uint4 *base, unsigned long long n_lines, int iters, uint32_t *sink) {
	unsigned long long i = ((unsigned long long) blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
	uint32_t acc = 0;
	for (int it = 0; it < iters; ++it) {
		uint4 v[8];
		#pragma unroll
		for (int k = 0; k < 8; ++k) { i = i * 6364136223846793005ull + 1442695040888963407ull; v[k] = base[((i >> 20) % n_lines) * 8ull + (i & 7ull)]; }
		#pragma unroll
		for (int k = 0; k < 8; ++k) acc ^= v[k].x ^ v[k].w;
	}
	if (acc == 0xDEADBEEFu) sink[0] = acc;
}

// evicts the caches between launch groups: a 1 GiB stream through another buffer
__global__ __launch_bounds__(256) void k_flush(uint4 *p, unsigned long long n16) {
	const unsigned long long i = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n16) p[i] = make_uint4((uint32_t) i, 1u, 2u, 3u);
}

int main() {
	setvbuf(stdout, nullptr, _IONBF, 0);
	const unsigned long long footprints[3] = { 32ull << 20, 150ull << 20, 2048ull << 20 };
	const unsigned long long maxBytes = footprints[2];
	uint4 *buf = nullptr, *flush = nullptr; uint32_t *sink = nullptr;
	CHECK(hipMalloc((void **) &buf, maxBytes));
	CHECK(hipMalloc((void **) &flush, 1ull << 30));
	CHECK(hipMalloc((void **) &sink, 64));
	CHECK(hipMemset(buf, 1, maxBytes));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	printf("shape footprint_MB units lines bytes_if_128 bytes_if_64 bytes_if_32 ms_first ms_rest lines_per_ns_rest\n");
	for (int shape = 0; shape < 4; ++shape) {
		for (int f = 0; f < 3; ++f) {
			const unsigned long long bytes = footprints[f], nLines = bytes / 128ull;
			unsigned long long units = nLines, lines = nLines, sect64 = nLines, sect32 = nLines;
			if (shape == 2) {
				units = bytes / 48ull;
				// every record once: all lines of the records' span are touched; count 64-B and 32-B sectors exactly
				lines = (units * 48ull + 127ull) / 128ull; sect64 = (units * 48ull + 63ull) / 64ull; sect32 = (units * 48ull + 31ull) / 32ull;
			} else if (shape == 3) {
				units = bytes / 16ull; sect64 = 2ull * nLines; sect32 = 4ull * nLines;
			} else if (shape == 1) {
				sect64 = nLines; sect32 = nLines;      // 8 bytes lie in one 32-byte sector
			}
			const unsigned blocks = (unsigned) ((units + 255ull) / 256ull);
			hipLaunchKernelGGL(k_flush, dim3((unsigned) ((1ull << 26) / 256ull)), dim3(256), 0, 0, flush, 1ull << 26);
			float msFirst = 0, msRest = 0;
			for (int rep = 0; rep < kReps; ++rep) {
				CHECK(hipEventRecord(e0));
				if (shape == 0) hipLaunchKernelGGL(k_g16, dim3(blocks), dim3(256), 0, 0, buf, nLines, sink);
				else if (shape == 1) hipLaunchKernelGGL(k_g8, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const uint2 *>(buf), nLines, sink);
				else if (shape == 2) hipLaunchKernelGGL(k_g48, dim3(blocks), dim3(256), 0, 0, buf, units, sink);
				else hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, 0, buf, units, sink);
				CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
				float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
				if (rep == 0) msFirst = ms; else msRest += ms / (kReps - 1);
			}
			const char *names[4] = { "k_g16", "k_g8", "k_g48", "k_stream" };
			printf("%s %llu %llu %llu %llu %llu %llu %.4f %.4f %.3f\n", names[shape], bytes >> 20, units, lines, lines * 128ull,
			       shape == 3 ? lines * 128ull : sect64 * 64ull, shape == 3 ? lines * 128ull : sect32 * 32ull, msFirst, msRest, lines / (msRest * 1e6));
		}
	}
	// sustained random-line rates (see k_sustained): 256 CUs x 8 workgroups of 256 lanes, 8 x iters gathers per lane
	printf("sustained random 16-byte gathers, 8 in flight per lane, 2048 workgroups of 256:\n");
	for (int f = 0; f < 3; ++f) {
		const unsigned long long nLines = footprints[f] / 128ull;
		const int iters = 64;
		hipLaunchKernelGGL(k_flush, dim3((unsigned) ((1ull << 26) / 256ull)), dim3(256), 0, 0, flush, 1ull << 26);
		float best = 1e30f;
		for (int rep = 0; rep < 3; ++rep) {
			CHECK(hipEventRecord(e0));
			hipLaunchKernelGGL(k_sustained, dim3(2048), dim3(256), 0, 0, buf, nLines, iters, sink);
			CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
			float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
		}
		const double loads = 2048.0 * 256.0 * 8.0 * iters;
		printf("k_sustained %llu MB: %.3f ms for %.0f line requests = %.2f G lines/s = %.2f TB/s in 128-byte lines\n", footprints[f] >> 20, best, loads, loads / (best * 1e6), loads * 128.0 / (best * 1e-3) / 1e12);
	}
	return 0;
}
