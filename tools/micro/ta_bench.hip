// ta_bench -- what a gather costs on gfx950's vector-memory front end (TA / TCP): wave-instructions per microsecond
// and CU for 16-byte (and 8-byte) loads with different numbers of active lanes, address spreads and footprints.
// hipcc --offload-arch=gfx950 -O3 tools/micro/ta_bench.hip -o gpurun_out/ta_bench && gpurun_out/ta_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int BYTES>
__global__ __launch_bounds__(256, 7) void k_gather(const uint4 *data, uint32_t mask_elems, int iters, int active_lanes, int mode, uint32_t *sink) {
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	uint32_t acc = 0;
	uint32_t x = gw * 0x9E3779B9u + lane * 0x85EBCA6Bu + 12345u;
	if ((int) lane < active_lanes) {
		for (int i = 0; i < iters; ++i) {
			x ^= x << 13; x ^= x >> 17; x ^= x << 5;      // xorshift32
			uint32_t idx;
			if (mode == 0) idx = x & mask_elems;                                  // every lane its own random 16-byte element
			else if (mode == 1) idx = ((x & mask_elems) & ~63u) + lane;            // consecutive lanes, random 1-KiB block (coalesced)
			else if (mode == 2) idx = ((x >> 6) & (mask_elems >> 2)) * 4u + (lane & 3u);   // 4 lanes share a 64-byte sector
			else idx = (__builtin_amdgcn_readfirstlane((int) x) & mask_elems);     // all lanes the same element
			if (BYTES == 16) { const uint4 v = data[idx]; acc += v.x ^ v.w; }
			else { const uint2 v = reinterpret_cast<const uint2 *>(data)[idx * 2u]; acc += v.x ^ v.y; }
		}
	}
	if (acc == 0xDEADBEEFu) sink[0] = acc;
}

int main() {
	const size_t maxBytes = 256ull << 20;
	uint4 *d; uint32_t *sink;
	hipMalloc(&d, maxBytes); hipMalloc(&sink, 4);
	hipMemset(d, 1, maxBytes);
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	const int blocks = cus * 7, iters = 2000;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	struct Case { const char *name; int bytes, lanes, mode; size_t footprint; };
	const size_t KB = 1024, MB = 1024 * 1024;
	std::vector<Case> cases = {
		{ "b128 random, 64 lanes, 4 MB", 16, 64, 0, 4 * MB }, { "b128 random, 28 lanes, 4 MB", 16, 28, 0, 4 * MB },
		{ "b128 random, 16 lanes, 4 MB", 16, 16, 0, 4 * MB }, { "b128 random,  8 lanes, 4 MB", 16, 8, 0, 4 * MB },
		{ "b128 random,  1 lane,  4 MB", 16, 1, 0, 4 * MB },
		{ "b64  random, 64 lanes, 4 MB", 8, 64, 0, 4 * MB }, { "b64  random, 28 lanes, 4 MB", 8, 28, 0, 4 * MB },
		{ "b128 random, 64 lanes, 16 KB (L1)", 16, 64, 0, 16 * KB }, { "b128 random, 28 lanes, 16 KB (L1)", 16, 28, 0, 16 * KB },
		{ "b128 random, 64 lanes, 64 MB", 16, 64, 0, 64 * MB }, { "b128 random, 28 lanes, 64 MB", 16, 28, 0, 64 * MB },
		{ "b128 coalesced, 64 lanes, 4 MB", 16, 64, 1, 4 * MB }, { "b128 4-lane sectors, 64 lanes, 4 MB", 16, 64, 2, 4 * MB },
		{ "b128 one address, 64 lanes, 4 MB", 16, 64, 3, 4 * MB },
	};
	printf("%d CUs, %d workgroups of 256, %d loads per lane\n", cus, blocks, iters);
	for (const Case &c : cases) {
		const uint32_t mask = (uint32_t) (c.footprint / 16 - 1);
		for (int rep = 0; rep < 2; ++rep) {
			hipEventRecord(e0);
			if (c.bytes == 16) hipLaunchKernelGGL(k_gather<16>, dim3(blocks), dim3(256), 0, 0, d, mask, iters, c.lanes, c.mode, sink);
			else hipLaunchKernelGGL(k_gather<8>, dim3(blocks), dim3(256), 0, 0, d, mask, iters, c.lanes, c.mode, sink);
			hipEventRecord(e1); hipEventSynchronize(e1);
		}
		float ms; hipEventElapsedTime(&ms, e0, e1);
		const double waveInstr = (double) blocks * 4 * iters;
		printf("%-40s %8.3f ms  %7.1f wave-instr/us/CU  %7.1f G lane-loads/s  (%.0f ns per wave-instr per CU)\n", c.name, ms,
		       waveInstr / cus / (ms * 1e3), waveInstr * c.lanes / (ms * 1e-3) / 1e9, ms * 1e6 * cus / waveInstr);
	}
	return 0;
}
