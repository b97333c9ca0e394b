// valu_bench -- what one vector instruction costs a gfx950 SIMD, scalar against packed FP32, by waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/micro/valu_bench.hip -o tools/micro/valu_bench && tools/micro/valu_bench
// Question behind it: does v_pk_mul_f32 / v_pk_add_f32 (two IEEE binary32 operations per lane and
// instruction, no contraction) take the issue time of ONE v_mul_f32, i.e. can pairing halve the arithmetic of k_trace?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

// 8 independent chains per lane, 32 instructions per loop trip and kind
template <int KIND>
__global__ __launch_bounds__(256) void k_valu(float *out, int trips, float a, float b) {
	float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
	f2 p0 = { x0, x1 }, p1 = { x2, x3 }, p2 = { x4, x5 }, p3 = { x6, x7 }, p4 = { x1, x0 }, p5 = { x3, x2 }, p6 = { x5, x4 }, p7 = { x7, x6 };
	const f2 pa = { a, a }, pb = { b, b };
	for (int t = 0; t < trips; ++t) {
		#pragma unroll
		for (int r = 0; r < 4; ++r) {
			if (KIND == 0) {         // v_mul_f32
				asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
				             "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
				             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
			} else if (KIND == 1) {  // v_fma_f32
				asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
				             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
				             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
			} else if (KIND == 2) {  // v_pk_mul_f32
				asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
				             "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8"
				             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa));
			} else if (KIND == 3) {  // v_pk_add_f32
				asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
				             "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8"
				             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa));
			} else if (KIND == 4) {  // v_pk_fma_f32
				asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
				             "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9"
				             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa), "v"(pb));
			} else if (KIND == 5) {  // v_cndmask_b32 (vcc)
				asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
				             "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
				             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a) : "vcc");
			} else if (KIND == 6) {  // v_cmp_le_f32 into an SGPR pair (VOP3)
				asm volatile("v_cmp_le_f32 s[20:21], %0, %8\n v_cmp_le_f32 s[22:23], %1, %8\n v_cmp_le_f32 s[20:21], %2, %8\n v_cmp_le_f32 s[22:23], %3, %8\n"
				             "v_cmp_le_f32 s[20:21], %4, %8\n v_cmp_le_f32 s[22:23], %5, %8\n v_cmp_le_f32 s[20:21], %6, %8\n v_cmp_le_f32 s[22:23], %7, %8"
				             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a) : "s20", "s21", "s22", "s23");
			} else if (KIND == 7) {  // v_pk_mul_f32 with a broadcast second operand (op_sel: both halves take the low dword of src1)
				asm volatile("v_pk_mul_f32 %0, %0, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %1, %1, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %2, %8 op_sel_hi:[1,0]\n"
				             "v_pk_mul_f32 %3, %3, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %4, %4, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %5, %5, %8 op_sel_hi:[1,0]\n"
				             "v_pk_mul_f32 %6, %6, %8 op_sel_hi:[1,0]\n v_pk_mul_f32 %7, %7, %8 op_sel_hi:[1,0]"
				             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa));
			}
		}
	}
	float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
	s += p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
	if (s == 12345.678f) out[0] = s;
}

template <int KIND> static float run(int blocks, int threads, int trips, float *out) {
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	float ms = 0;
	for (int rep = 0; rep < 2; ++rep) {
		hipEventRecord(e0);
		hipLaunchKernelGGL(k_valu<KIND>, dim3(blocks), dim3(threads), 0, 0, out, trips, 1.0000001f, 1e-9f);
		hipEventRecord(e1); hipEventSynchronize(e1);
		hipEventElapsedTime(&ms, e0, e1);
	}
	return ms;
}

int main() {
	float *out; hipMalloc(&out, 4);
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	const double clk = prop.clockRate * 1e3;      // Hz
	const int trips = 20000;
	const char *names[] = { "v_mul_f32", "v_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_cndmask_b32", "v_cmp_le_f32 -> sgpr",
	                        "v_pk_mul_f32 op_sel_hi:[1,0]" };
	printf("%d CUs, clockRate %.0f MHz; cycles per wave-instruction and SIMD at the nominal clock (lower bound if the chip clocks down)\n", cus, clk / 1e6);
	for (int wps = 1; wps <= 8; wps *= 2) {
		const int threads = 256, blocks = cus * wps;    // wps workgroups of 4 waves per CU = wps waves per SIMD
		float ms[8];
		ms[0] = run<0>(blocks, threads, trips, out); ms[1] = run<1>(blocks, threads, trips, out); ms[2] = run<2>(blocks, threads, trips, out);
		ms[3] = run<3>(blocks, threads, trips, out); ms[4] = run<4>(blocks, threads, trips, out); ms[5] = run<5>(blocks, threads, trips, out);
		ms[6] = run<6>(blocks, threads, trips, out); ms[7] = run<7>(blocks, threads, trips, out);
		for (int k = 0; k < 8; ++k) {
			const double instrPerSimd = (double) wps * trips * 32;
			printf("%d waves/SIMD  %-30s %8.3f ms  %5.2f cycles per wave-instruction per SIMD\n", wps, names[k], ms[k], ms[k] * 1e-3 * clk / instrPerSimd);
		}
	}
	return 0;
}
