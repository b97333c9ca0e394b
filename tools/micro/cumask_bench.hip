// cumask_bench -- what a CU-masked stream gets on gfx950: streaming bandwidth of a triad restricted to N CUs, for
// different placements of the N mask bits.  hipcc --offload-arch=gfx950 -O3 tools/micro/cumask_bench.hip -o gpurun_out/cumask_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
#include <cstdlib>
__global__ void k_triad(float4 *a, const float4 *b, const float4 *c, float s, size_t n) {
	const size_t stride = (size_t) gridDim.x * blockDim.x;
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const float4 x = b[i], y = c[i];
		a[i] = make_float4(x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w);
	}
}
__global__ void k_where(uint32_t *out) {
	// XCC_ID (hwreg 20) and HW_ID (hwreg 4): which XCD / SE / CU runs this workgroup
	if (threadIdx.x == 0) {
		uint32_t xcc, hw;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
		out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw;
	}
}
int main(int argc, char **argv) {
	const int only = argc > 1 ? atoi(argv[1]) : -1; int caseNo = -1;
	setvbuf(stdout, nullptr, _IONBF, 0);
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	const size_t n = (size_t) 64 << 20;      // float4 elements: 1 GiB per array
	float4 *a, *b, *c; hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&c, n * 16);
	hipMemset(b, 0, n * 16); hipMemset(c, 0, n * 16);
	uint32_t *where; hipMalloc(&where, 8 * 4096);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	struct Case { const char *name; int n; int mode; };
	const Case cases[] = { { "all CUs (no mask)", cus, -1 }, { "low bits", 16, 0 }, { "low bits", 32, 0 }, { "every 16th bit", 16, 1 }, { "every 8th bit", 32, 1 },
	                       { "low bits", 64, 0 }, { "every 4th bit", 64, 1 }, { "all but every 16th", cus - 16, 2 }, { "low bits", 8, 0 }, { "every 32nd bit", 8, 1 } };
	for (const Case &cs : cases) {
		++caseNo; if (only >= 0 && caseNo != only) continue;
		printf("case %d: %s n=%d ...\n", caseNo, cs.name, cs.n);
		hipStream_t s;
		std::vector<uint32_t> mask((cus + 31) / 32, 0u);
		if (cs.mode < 0) hipStreamCreate(&s);
		else {
			const int step = cs.mode == 0 ? 1 : cus / cs.n;
			if (cs.mode == 2) { for (int i = 0; i < cus; ++i) if (i % 16 != 0) mask[i / 32] |= 1u << (i % 32); }
			else for (int k = 0; k < cs.n; ++k) { const int i = k * step; mask[i / 32] |= 1u << (i % 32); }
			hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t) mask.size(), mask.data());
			if (e != hipSuccess) { printf("%-24s n=%3d: hipExtStreamCreateWithCUMask failed: %s\n", cs.name, cs.n, hipGetErrorString(e)); continue; }
		}
		// where do 64 workgroups land?
		hipMemsetAsync(where, 0xFF, 8 * 4096, s);
		hipLaunchKernelGGL(k_where, dim3(2048), dim3(64), 0, s, where);
		std::vector<uint32_t> h(4096);
		hipMemcpyAsync(h.data(), where, 4 * 4096, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
		uint32_t xccSeen = 0; std::vector<int> perXcc(16, 0); std::vector<uint8_t> seen(16 * 4096, 0); int distinct = 0;
		for (int i = 0; i < 2048; ++i) {
			const uint32_t x = h[2 * i] & 15u, hw = h[2 * i + 1];
			const uint32_t cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
			const uint32_t key = x * 256 + se * 32 + sh * 16 + cu;
			if (!seen[key]) { seen[key] = 1; ++distinct; perXcc[x]++; }
			xccSeen |= 1u << x;
		}
		float best = 1e30f;
		for (int rep = 0; rep < 3; ++rep) {
			hipEventRecord(e0, s);
			hipLaunchKernelGGL(k_triad, dim3(cs.n * 16), dim3(256), 0, s, a, b, c, 2.0f, n);
			hipEventRecord(e1, s); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
		}
		printf("%-24s n=%3d: triad %7.1f GB/s   distinct CUs seen %3d, per XCD:", cs.name, cs.n, 3.0 * n * 16 / (best * 1e-3) / 1e9, distinct);
		for (int x = 0; x < 8; ++x) printf(" %d", perXcc[x]);
		printf("\n");
		hipStreamDestroy(s);
	}
	return 0;
}
