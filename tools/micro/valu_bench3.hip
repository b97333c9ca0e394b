// valu_bench3 -- v_cndmask_b32 by encoding and mask source on gfx950 (follow-up of valu_bench2: the VOP2 form reading VCC
// measured 23 cycles there).  hipcc --offload-arch=gfx950 -O3 tools/micro/valu_bench3.hip -o tools/micro/valu_bench3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(I0, I1, I2, I3, I4, I5, I6, I7) I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7
#define KIND_V(NAME, PRE, T0, T1, T2, T3, T4, T5, T6, T7, CLOB...) \
	__global__ __launch_bounds__(256) void NAME(float *out, int trips, float a, float b) { \
		float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
		asm volatile(PRE : "+v"(x0) : "v"(a) : CLOB); \
		for (int t = 0; t < trips; ++t) { \
			_Pragma("unroll") for (int r = 0; r < 4; ++r) \
				asm volatile(REP8(T0, T1, T2, T3, T4, T5, T6, T7) \
				             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : CLOB); \
		} \
		float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7; \
		if (s == 12345.678f) out[0] = s; \
	}
#define A(i) "v_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc"
KIND_V(k_e32_self, "v_cmp_lt_f32 vcc, 2.0, %0", A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), "vcc")
#define B(i) "v_cndmask_b32_e32 %" #i ", %8, %9, vcc"
KIND_V(k_e32_other, "v_cmp_lt_f32 vcc, 2.0, %0", B(0), B(1), B(2), B(3), B(4), B(5), B(6), B(7), "vcc")
#define Cc(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc"
KIND_V(k_e64_vcc_self, "v_cmp_lt_f32 vcc, 2.0, %0", Cc(0), Cc(1), Cc(2), Cc(3), Cc(4), Cc(5), Cc(6), Cc(7), "vcc")
#define D(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]"
KIND_V(k_e64_sgpr_self, "v_cmp_lt_f32 s[20:21], 2.0, %0", D(0), D(1), D(2), D(3), D(4), D(5), D(6), D(7), "s20", "s21")
// all-zero / all-one masks
KIND_V(k_e32_self_vcc0, "s_mov_b64 vcc, 0", A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), "vcc")
KIND_V(k_e32_self_vcc1, "s_mov_b64 vcc, -1", A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), "vcc")
// the pair as the compiler emits it in k_trace: compare into vcc, one wait state, two selects
#define P(i) "v_cmp_le_f32_e32 vcc, %" #i ", %8\n s_nop 0\n v_cndmask_b32_e32 %" #i ", %8, %9, vcc"
KIND_V(k_pair_vcc, "", P(0), P(1), P(2), P(3), P(4), P(5), P(6), P(7), "vcc")
#define Q(i) "v_cmp_le_f32_e64 s[20:21], %" #i ", %8\n v_cndmask_b32_e64 %" #i ", %8, %9, s[20:21]"
KIND_V(k_pair_sgpr, "", Q(0), Q(1), Q(2), Q(3), Q(4), Q(5), Q(6), Q(7), "s20", "s21")
#define M(i) "v_max_f32 %" #i ", %" #i ", %8"
KIND_V(k_max, "", M(0), M(1), M(2), M(3), M(4), M(5), M(6), M(7), "memory")
#define AD(i) "v_add_f32 %" #i ", %" #i ", %8"
KIND_V(k_add, "", AD(0), AD(1), AD(2), AD(3), AD(4), AD(5), AD(6), AD(7), "memory")
#define XO(i) "v_xor_b32 %" #i ", %" #i ", %8"
KIND_V(k_xor, "", XO(0), XO(1), XO(2), XO(3), XO(4), XO(5), XO(6), XO(7), "memory")
#define LS(i) "v_lshrrev_b32 %" #i ", 2, %" #i
KIND_V(k_lshr, "", LS(0), LS(1), LS(2), LS(3), LS(4), LS(5), LS(6), LS(7), "memory")
#define AD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9"
KIND_V(k_add3, "", AD3(0), AD3(1), AD3(2), AD3(3), AD3(4), AD3(5), AD3(6), AD3(7), "memory")
#define FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9"
KIND_V(k_fma, "", FMA(0), FMA(1), FMA(2), FMA(3), FMA(4), FMA(5), FMA(6), FMA(7), "memory")
#define MULE64(i) "v_mul_f32_e64 %" #i ", %" #i ", |%8|"
KIND_V(k_mul_e64, "", MULE64(0), MULE64(1), MULE64(2), MULE64(3), MULE64(4), MULE64(5), MULE64(6), MULE64(7), "memory")
#define LU64(i) "v_lshl_add_u64 v[30:31], v[30:31], 4, v[32:33]"
KIND_V(k_lshl_add_u64, "", LU64(0), LU64(1), LU64(2), LU64(3), LU64(4), LU64(5), LU64(6), LU64(7), "v30", "v31", "v32", "v33")
#define MAD64(i) "v_mad_u64_u32 v[30:31], s[20:21], %" #i ", 48, v[32:33]"
KIND_V(k_mad_u64_u32, "", MAD64(0), MAD64(1), MAD64(2), MAD64(3), MAD64(4), MAD64(5), MAD64(6), MAD64(7), "v30", "v31", "v32", "v33", "s20", "s21")
#define MBC(i) "v_mbcnt_lo_u32_b32 %" #i ", s20, %" #i
KIND_V(k_mbcnt, "", MBC(0), MBC(1), MBC(2), MBC(3), MBC(4), MBC(5), MBC(6), MBC(7), "memory")
#define RFL(i) "v_readfirstlane_b32 s20, %" #i
KIND_V(k_readfirstlane, "", RFL(0), RFL(1), RFL(2), RFL(3), RFL(4), RFL(5), RFL(6), RFL(7), "s20")

// spacing patterns of the VOP2 select: 8 instructions per group, chains as above
#define MU(i) "v_mul_f32 %" #i ", %" #i ", %8"
KIND_V(k_pat_2c6m, "v_cmp_lt_f32 vcc, 2.0, %0", A(0), A(1), MU(2), MU(3), MU(4), MU(5), MU(6), MU(7), "vcc")
KIND_V(k_pat_alt, "v_cmp_lt_f32 vcc, 2.0, %0", A(0), MU(1), A(2), MU(3), A(4), MU(5), A(6), MU(7), "vcc")
KIND_V(k_pat_c2m, "v_cmp_lt_f32 vcc, 2.0, %0", A(0), MU(1), MU(2), A(3), MU(4), MU(5), A(6), MU(7), "vcc")
KIND_V(k_pat_c3m, "v_cmp_lt_f32 vcc, 2.0, %0", A(0), MU(1), MU(2), MU(3), A(4), MU(5), MU(6), MU(7), "vcc")
KIND_V(k_pat_1c7m, "v_cmp_lt_f32 vcc, 2.0, %0", A(0), MU(1), MU(2), MU(3), MU(4), MU(5), MU(6), MU(7), "vcc")
KIND_V(k_pat_8m, "v_cmp_lt_f32 vcc, 2.0, %0", MU(0), MU(1), MU(2), MU(3), MU(4), MU(5), MU(6), MU(7), "vcc")
KIND_V(k_pat_nop, "v_cmp_lt_f32 vcc, 2.0, %0", A(0) "\n s_nop 0", A(1) "\n s_nop 0", A(2) "\n s_nop 0", A(3) "\n s_nop 0", A(4) "\n s_nop 0", A(5) "\n s_nop 0", A(6) "\n s_nop 0", A(7) "\n s_nop 0", "vcc")
KIND_V(k_pat_e32_e64, "v_cmp_lt_f32 vcc, 2.0, %0", A(0), Cc(1), A(2), Cc(3), A(4), Cc(5), A(6), Cc(7), "vcc")
KIND_V(k_pat_2e64_6m, "v_cmp_lt_f32 vcc, 2.0, %0", Cc(0), Cc(1), MU(2), MU(3), MU(4), MU(5), MU(6), MU(7), "vcc")
// VOP2 with the carry-in form of VCC: v_addc_co_u32_e32 reads VCC too
#define AC(i) "v_addc_co_u32_e32 %" #i ", vcc, %" #i ", %8, vcc"
KIND_V(k_addc_e32, "v_cmp_lt_f32 vcc, 2.0, %0", AC(0), AC(1), AC(2), AC(3), AC(4), AC(5), AC(6), AC(7), "vcc")
typedef void (*Kern)(float *, int, float, float);
int main() {
	float *out; hipMalloc(&out, 4);
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount; const double clk = prop.clockRate * 1e3; const int trips = 10000;
	struct K { const char *name; Kern k; int per; };
	std::vector<K> ks = {
		{ "v_cndmask_b32_e32 x, x, a, vcc (vcc mixed)", k_e32_self, 1 }, { "v_cndmask_b32_e32 x, a, b, vcc", k_e32_other, 1 },
		{ "v_cndmask_b32_e64 x, x, a, vcc", k_e64_vcc_self, 1 }, { "v_cndmask_b32_e64 x, x, a, s[20:21]", k_e64_sgpr_self, 1 },
		{ "v_cndmask_b32_e32 x, x, a, vcc (vcc = 0)", k_e32_self_vcc0, 1 }, { "v_cndmask_b32_e32 x, x, a, vcc (vcc = -1)", k_e32_self_vcc1, 1 },
		{ "pair v_cmp_e32 vcc; s_nop; v_cndmask_e32 vcc", k_pair_vcc, 1 }, { "pair v_cmp_e64 sgpr; v_cndmask_e64 sgpr", k_pair_sgpr, 1 },
		{ "v_max_f32", k_max, 1 }, { "v_add_f32", k_add, 1 }, { "v_xor_b32", k_xor, 1 }, { "v_lshrrev_b32", k_lshr, 1 }, { "v_add3_u32", k_add3, 1 },
		{ "v_fma_f32", k_fma, 1 }, { "v_mul_f32_e64 |abs|", k_mul_e64, 1 }, { "v_lshl_add_u64", k_lshl_add_u64, 1 }, { "v_mad_u64_u32", k_mad_u64_u32, 1 },
		{ "group: 2 cnd_e32 + 6 v_mul  (per instr)", k_pat_2c6m, 1 }, { "group: cnd_e32, mul alternating", k_pat_alt, 1 },
		{ "group: cnd_e32, mul, mul, ...", k_pat_c2m, 1 }, { "group: cnd_e32, 3 mul, ...", k_pat_c3m, 1 }, { "group: 1 cnd_e32 + 7 mul", k_pat_1c7m, 1 },
		{ "group: 8 mul", k_pat_8m, 1 }, { "cnd_e32; s_nop 0 (per cnd)", k_pat_nop, 1 }, { "group: cnd_e32, cnd_e64 alternating", k_pat_e32_e64, 1 },
		{ "group: 2 cnd_e64 + 6 mul", k_pat_2e64_6m, 1 }, { "v_addc_co_u32_e32 (reads vcc)", k_addc_e32, 1 },
		{ "v_mbcnt_lo_u32_b32", k_mbcnt, 1 }, { "v_readfirstlane_b32", k_readfirstlane, 1 },
	};
	printf("%d CUs; cycles per wave-instruction (or pair) and SIMD at the nominal clock %.0f MHz\n", cus, clk / 1e6);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (const K &k : ks) {
		printf("%-48s", k.name);
		for (int wps = 1; wps <= 8; wps *= 2) {
			float ms = 0;
			for (int rep = 0; rep < 2; ++rep) {
				hipEventRecord(e0);
				hipLaunchKernelGGL(k.k, dim3(cus * wps), dim3(256), 0, 0, out, trips, 1.0000001f, 0.5f);
				hipEventRecord(e1); hipEventSynchronize(e1);
				hipEventElapsedTime(&ms, e0, e1);
			}
			printf("  %d w/SIMD: %6.2f", wps, ms * 1e-3 * clk / ((double) wps * trips * 32));
		}
		printf("\n");
	}
	return 0;
}
