// ta_bench -- what a gather costs on gfx950's vector-memory front end (TA / TCP): wave-instructions per microsecond
// and CU for 16-byte (and 8-byte) loads by number of active lanes, lanes per 128-byte line, cache policy and footprint.
// hipcc --offload-arch=gfx950 -O3 tools/micro/ta_bench.hip -o gpurun_out/ta_bench && gpurun_out/ta_bench
//
// Round 3: the "coalesced" and "4-lane sector" rows of round 2 took their block from the PER-LANE random number, so
// they were random too.  Here the block comes from a per-GROUP number (the first lane's, through ds_bpermute), the group
// being G consecutive lanes that share one 128-byte line (G = 8: the whole line, 16 bytes each) or one 64-byte half.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));
enum Policy { PLAIN = 0, NT = 1, SC1 = 2, SC0SC1 = 3, SC0 = 4 };

template <int POLICY> __device__ __forceinline__ u4 load16(const uint4 *p) {
	u4 v;
	if (POLICY == PLAIN) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
	else if (POLICY == NT) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
	else if (POLICY == SC1) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
	else if (POLICY == SC0) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
	else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
	return v;
}

// mode: lanes per group sharing a block.  share = 1: every lane its own random 16-byte element;
// share = G > 1: G consecutive lanes read consecutive 16-byte elements of one random, G*16-byte aligned block
// (G = 8: one 128-byte line per 8 lanes; G = 4: one 64-byte half; G = 64: 1 KiB per wave);  share = 0: one address.
// second = 1: every load is followed by a load of the NEXT 16-byte element of the same line (an L1 hit).
template <int POLICY>
__global__ __launch_bounds__(256, 7) void k_gather(const uint4 *data, uint32_t mask_elems, int iters, int active_lanes, int share, int second,
                                                   uint32_t *sink) {
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	uint32_t acc = 0;
	uint32_t x = gw * 0x9E3779B9u + lane * 0x85EBCA6Bu + 12345u;
	const uint32_t leader = share > 1 ? (lane / (uint32_t) share) * (uint32_t) share : lane;
	if ((int) lane < active_lanes) {
		for (int i = 0; i < iters; i += 4) {
			u4 v[4], w[4];
			#pragma unroll
			for (int k = 0; k < 4; ++k) {
				x ^= x << 13; x ^= x >> 17; x ^= x << 5;      // xorshift32
				uint32_t idx;
				if (share == 1) idx = x & mask_elems;
				else if (share == 0) idx = (uint32_t) __builtin_amdgcn_readfirstlane((int) x) & mask_elems;
				else {
					const uint32_t xl = (uint32_t) __builtin_amdgcn_ds_bpermute((int) (leader * 4u), (int) x);
					idx = ((xl & mask_elems) & ~((uint32_t) share - 1u)) + (lane - leader);
				}
				v[k] = load16<POLICY>(data + idx);
				if (second) w[k] = load16<POLICY>(data + (idx ^ 1u));
			}
			// the loaded registers are operands of the wait so that no use of them is scheduled in front of it
			asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) :: "memory");
			if (second) asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
			#pragma unroll
			for (int k = 0; k < 4; ++k) { acc += v[k].x ^ v[k].w; if (second) acc += w[k].y; }
		}
	}
	if (acc == 0xDEADBEEFu) sink[0] = acc;
}

template <int POLICY>
static float run(const uint4 *d, uint32_t mask, int blocks, int iters, int lanes, int share, int second, uint32_t *sink) {
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	float ms = 0;
	for (int rep = 0; rep < 2; ++rep) {
		hipEventRecord(e0);
		hipLaunchKernelGGL(k_gather<POLICY>, dim3(blocks), dim3(256), 0, 0, d, mask, iters, lanes, share, second, sink);
		hipEventRecord(e1); hipEventSynchronize(e1);
		hipEventElapsedTime(&ms, e0, e1);
	}
	hipEventDestroy(e0); hipEventDestroy(e1);
	return ms;
}

int main() {
	const size_t maxBytes = 256ull << 20;
	uint4 *d; uint32_t *sink;
	hipMalloc(&d, maxBytes); hipMalloc(&sink, 4);
	hipMemset(d, 1, maxBytes);
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	const int blocks = cus * 7, iters = 2000;
	struct Case { const char *name; int policy, lanes, share, second; size_t footprint; };
	const size_t KB = 1024, MB = 1024 * 1024;
	std::vector<Case> cases = {
		{ "random, 64 lanes, 4 MB", PLAIN, 64, 1, 0, 4 * MB }, { "random, 28 lanes, 4 MB", PLAIN, 28, 1, 0, 4 * MB },
		{ "random, 16 lanes, 4 MB", PLAIN, 16, 1, 0, 4 * MB }, { "random,  8 lanes, 4 MB", PLAIN, 8, 1, 0, 4 * MB },
		{ "random,  4 lanes, 4 MB", PLAIN, 4, 1, 0, 4 * MB }, { "random,  1 lane,  4 MB", PLAIN, 1, 1, 0, 4 * MB },
		{ "2 lanes / 32 B, 64 lanes, 4 MB", PLAIN, 64, 2, 0, 4 * MB }, { "4 lanes / 64-B half, 64 lanes, 4 MB", PLAIN, 64, 4, 0, 4 * MB },
		{ "8 lanes / 128-B line, 64 lanes, 4 MB", PLAIN, 64, 8, 0, 4 * MB }, { "16 lanes / 256 B, 64 lanes, 4 MB", PLAIN, 64, 16, 0, 4 * MB },
		{ "64 lanes / 1 KiB (coalesced), 4 MB", PLAIN, 64, 64, 0, 4 * MB }, { "one address, 64 lanes, 4 MB", PLAIN, 64, 0, 0, 4 * MB },
		{ "8 lanes / line, 32 lanes, 4 MB", PLAIN, 32, 8, 0, 4 * MB }, { "8 lanes / line, 8 lanes, 4 MB", PLAIN, 8, 8, 0, 4 * MB },
		{ "random + next 16 B of the line, 64 lanes, 4 MB", PLAIN, 64, 1, 1, 4 * MB },
		{ "random + next 16 B of the line, 28 lanes, 4 MB", PLAIN, 28, 1, 1, 4 * MB },
		{ "random nt, 64 lanes, 4 MB", NT, 64, 1, 0, 4 * MB }, { "random sc1, 64 lanes, 4 MB", SC1, 64, 1, 0, 4 * MB },
		{ "random sc0, 64 lanes, 4 MB", SC0, 64, 1, 0, 4 * MB }, { "random sc0 sc1, 64 lanes, 4 MB", SC0SC1, 64, 1, 0, 4 * MB },
		{ "random sc1, 28 lanes, 4 MB", SC1, 28, 1, 0, 4 * MB },
		{ "random, 64 lanes, 16 KB (L1)", PLAIN, 64, 1, 0, 16 * KB }, { "random, 28 lanes, 16 KB (L1)", PLAIN, 28, 1, 0, 16 * KB },
		{ "random, 64 lanes, 1 MB", PLAIN, 64, 1, 0, 1 * MB }, { "random, 64 lanes, 16 MB", PLAIN, 64, 1, 0, 16 * MB },
		{ "random, 64 lanes, 64 MB", PLAIN, 64, 1, 0, 64 * MB }, { "random, 28 lanes, 64 MB", PLAIN, 28, 1, 0, 64 * MB },
		{ "random sc1, 64 lanes, 64 MB", SC1, 64, 1, 0, 64 * MB }, { "random nt, 64 lanes, 64 MB", NT, 64, 1, 0, 64 * MB },
		{ "8 lanes / line, 64 lanes, 64 MB", PLAIN, 64, 8, 0, 64 * MB }, { "4 lanes / half, 64 lanes, 64 MB", PLAIN, 64, 4, 0, 64 * MB },
	};
	printf("%d CUs, %d workgroups of 256, %d loads per lane, 16 bytes per lane and load\n", cus, blocks, iters);
	for (const Case &c : cases) {
		const uint32_t mask = (uint32_t) (c.footprint / 16 - 1);
		float ms = 0;
		switch (c.policy) {
			case PLAIN: ms = run<PLAIN>(d, mask, blocks, iters, c.lanes, c.share, c.second, sink); break;
			case NT: ms = run<NT>(d, mask, blocks, iters, c.lanes, c.share, c.second, sink); break;
			case SC1: ms = run<SC1>(d, mask, blocks, iters, c.lanes, c.share, c.second, sink); break;
			case SC0: ms = run<SC0>(d, mask, blocks, iters, c.lanes, c.share, c.second, sink); break;
			default: ms = run<SC0SC1>(d, mask, blocks, iters, c.lanes, c.share, c.second, sink); break;
		}
		const double waveInstr = (double) blocks * 4 * iters * (c.second ? 2 : 1);
		const double lines = c.share > 1 ? (double) c.lanes / c.share * (c.share > 8 ? c.share / 8 : 1) : (c.share == 0 ? 1 : c.lanes);
		printf("%-50s %8.3f ms  %7.1f wave-instr/us/CU  %7.1f G lane-loads/s  %6.1f ns per wave-instr per CU  %5.2f ns per 128-B line\n", c.name, ms,
		       waveInstr / cus / (ms * 1e3), waveInstr * c.lanes / (ms * 1e-3) / 1e9, ms * 1e6 * cus / waveInstr,
		       ms * 1e6 * cus / waveInstr / lines);
	}
	return 0;
}
