// valu_bench2 -- issue cost of the instruction kinds k_trace is made of, on gfx950, by waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/micro/valu_bench2.hip -o tools/micro/valu_bench2 && tools/micro/valu_bench2
// Every kind: 8 independent chains per lane, 32 instructions per loop trip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(I0, I1, I2, I3, I4, I5, I6, I7) I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7
// one instruction text with the chain register as %0..%7 and the constants %8 (vgpr), %9 (vgpr)
#define KIND_V(NAME, T0, T1, T2, T3, T4, T5, T6, T7, CLOB...) \
	__global__ __launch_bounds__(256) void NAME(float *out, int trips, float a, float b) { \
		__shared__ float lds[256]; if (trips < 0) lds[threadIdx.x] = a; \
		float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
		for (int t = 0; t < trips; ++t) { \
			_Pragma("unroll") for (int r = 0; r < 4; ++r) \
				asm volatile(REP8(T0, T1, T2, T3, T4, T5, T6, T7) \
				             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : CLOB); \
		} \
		float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7; \
		if (s == 12345.678f) out[0] = s + lds[0]; \
	}
#define SAME8(FMT) FMT(0), FMT(1), FMT(2), FMT(3), FMT(4), FMT(5), FMT(6), FMT(7)

#define F_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8"
KIND_V(k_mul, F_MUL(0), F_MUL(1), F_MUL(2), F_MUL(3), F_MUL(4), F_MUL(5), F_MUL(6), F_MUL(7), "memory")
#define F_CND_VCC(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc"
KIND_V(k_cnd_vcc, F_CND_VCC(0), F_CND_VCC(1), F_CND_VCC(2), F_CND_VCC(3), F_CND_VCC(4), F_CND_VCC(5), F_CND_VCC(6), F_CND_VCC(7), "memory")
#define F_CND_S(i) "v_cndmask_b32 %" #i ", %" #i ", %8, s[20:21]"
KIND_V(k_cnd_sgpr, F_CND_S(0), F_CND_S(1), F_CND_S(2), F_CND_S(3), F_CND_S(4), F_CND_S(5), F_CND_S(6), F_CND_S(7), "memory")
// select between two OTHER registers into the chain register (no read of the destination)
#define F_CND_2(i) "v_cndmask_b32 %" #i ", %8, %9, s[20:21]"
KIND_V(k_cnd_other, F_CND_2(0), F_CND_2(1), F_CND_2(2), F_CND_2(3), F_CND_2(4), F_CND_2(5), F_CND_2(6), F_CND_2(7), "memory")
#define F_CMP_VCC(i) "v_cmp_le_f32 vcc, %" #i ", %8"
KIND_V(k_cmp_vcc, F_CMP_VCC(0), F_CMP_VCC(1), F_CMP_VCC(2), F_CMP_VCC(3), F_CMP_VCC(4), F_CMP_VCC(5), F_CMP_VCC(6), F_CMP_VCC(7), "vcc")
#define F_CMP_S(i) "v_cmp_le_f32 s[20:21], %" #i ", %8"
KIND_V(k_cmp_sgpr, F_CMP_S(0), F_CMP_S(1), F_CMP_S(2), F_CMP_S(3), F_CMP_S(4), F_CMP_S(5), F_CMP_S(6), F_CMP_S(7), "s20", "s21")
#define F_CMPU_S(i) "v_cmp_eq_u32 s[20:21], %" #i ", %8"
KIND_V(k_cmpu_sgpr, F_CMPU_S(0), F_CMPU_S(1), F_CMPU_S(2), F_CMPU_S(3), F_CMPU_S(4), F_CMPU_S(5), F_CMPU_S(6), F_CMPU_S(7), "s20", "s21")
#define F_MOV(i) "v_mov_b32 %" #i ", %8"
KIND_V(k_mov, F_MOV(0), F_MOV(1), F_MOV(2), F_MOV(3), F_MOV(4), F_MOV(5), F_MOV(6), F_MOV(7), "memory")
#define F_ADDU(i) "v_add_u32 %" #i ", %" #i ", %8"
KIND_V(k_add_u32, F_ADDU(0), F_ADDU(1), F_ADDU(2), F_ADDU(3), F_ADDU(4), F_ADDU(5), F_ADDU(6), F_ADDU(7), "memory")
#define F_AND(i) "v_and_b32 %" #i ", %" #i ", %8"
KIND_V(k_and, F_AND(0), F_AND(1), F_AND(2), F_AND(3), F_AND(4), F_AND(5), F_AND(6), F_AND(7), "memory")
#define F_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i
KIND_V(k_lshl, F_LSHL(0), F_LSHL(1), F_LSHL(2), F_LSHL(3), F_LSHL(4), F_LSHL(5), F_LSHL(6), F_LSHL(7), "memory")
#define F_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %8"
KIND_V(k_lshl_add, F_LSHLADD(0), F_LSHLADD(1), F_LSHLADD(2), F_LSHLADD(3), F_LSHLADD(4), F_LSHLADD(5), F_LSHLADD(6), F_LSHLADD(7), "memory")
#define F_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9"
KIND_V(k_mad_u24, F_MAD24(0), F_MAD24(1), F_MAD24(2), F_MAD24(3), F_MAD24(4), F_MAD24(5), F_MAD24(6), F_MAD24(7), "memory")
#define F_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 1, 30"
KIND_V(k_bfe, F_BFE(0), F_BFE(1), F_BFE(2), F_BFE(3), F_BFE(4), F_BFE(5), F_BFE(6), F_BFE(7), "memory")
#define F_SUB(i) "v_sub_f32 %" #i ", %" #i ", %8"
KIND_V(k_sub, F_SUB(0), F_SUB(1), F_SUB(2), F_SUB(3), F_SUB(4), F_SUB(5), F_SUB(6), F_SUB(7), "memory")
#define F_MAX(i) "v_max_f32 %" #i ", %" #i ", %8"
KIND_V(k_max, F_MAX(0), F_MAX(1), F_MAX(2), F_MAX(3), F_MAX(4), F_MAX(5), F_MAX(6), F_MAX(7), "memory")
#define F_RCP(i) "v_rcp_f32 %" #i ", %" #i
KIND_V(k_rcp, F_RCP(0), F_RCP(1), F_RCP(2), F_RCP(3), F_RCP(4), F_RCP(5), F_RCP(6), F_RCP(7), "memory")
#define F_DIVFIX(i) "v_div_fixup_f32 %" #i ", %" #i ", %8, %9"
KIND_V(k_div_fixup, F_DIVFIX(0), F_DIVFIX(1), F_DIVFIX(2), F_DIVFIX(3), F_DIVFIX(4), F_DIVFIX(5), F_DIVFIX(6), F_DIVFIX(7), "memory")
#define F_DIVSCALE(i) "v_div_scale_f32 %" #i ", vcc, %" #i ", %8, %9"
KIND_V(k_div_scale, F_DIVSCALE(0), F_DIVSCALE(1), F_DIVSCALE(2), F_DIVSCALE(3), F_DIVSCALE(4), F_DIVSCALE(5), F_DIVSCALE(6), F_DIVSCALE(7), "vcc")
#define F_DIVFMAS(i) "v_div_fmas_f32 %" #i ", %" #i ", %8, %9"
KIND_V(k_div_fmas, F_DIVFMAS(0), F_DIVFMAS(1), F_DIVFMAS(2), F_DIVFMAS(3), F_DIVFMAS(4), F_DIVFMAS(5), F_DIVFMAS(6), F_DIVFMAS(7), "memory")
#define F_MOVDPP(i) "v_mov_b32_dpp %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
KIND_V(k_mov_dpp, F_MOVDPP(0), F_MOVDPP(1), F_MOVDPP(2), F_MOVDPP(3), F_MOVDPP(4), F_MOVDPP(5), F_MOVDPP(6), F_MOVDPP(7), "memory")
#define F_ADD64(i) "v_lshlrev_b64 v[30:31], 3, v[30:31]"
KIND_V(k_lshl_b64, F_ADD64(0), F_ADD64(1), F_ADD64(2), F_ADD64(3), F_ADD64(4), F_ADD64(5), F_ADD64(6), F_ADD64(7), "v30", "v31")
#define F_ADDCO(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %8"
KIND_V(k_add_co, F_ADDCO(0), F_ADDCO(1), F_ADDCO(2), F_ADDCO(3), F_ADDCO(4), F_ADDCO(5), F_ADDCO(6), F_ADDCO(7), "vcc")
#define F_ADDC(i) "v_addc_co_u32 %" #i ", vcc, %" #i ", %8, vcc"
KIND_V(k_addc_co, F_ADDC(0), F_ADDC(1), F_ADDC(2), F_ADDC(3), F_ADDC(4), F_ADDC(5), F_ADDC(6), F_ADDC(7), "vcc")
// SALU on masks, as the case logic of the descent step uses them
#define F_SAND(i) "s_and_b64 s[20:21], s[20:21], s[22:23]"
KIND_V(k_s_and_b64, F_SAND(0), F_SAND(1), F_SAND(2), F_SAND(3), F_SAND(4), F_SAND(5), F_SAND(6), F_SAND(7), "s20", "s21", "scc")
// LDS: conflict-free dword read / write at [lane] (address in the chain register is NOT used: v29 holds lane*4)
#define F_DSR(i) "ds_read_b32 %" #i ", %9"
KIND_V(k_ds_read_b32, F_DSR(0), F_DSR(1), F_DSR(2), F_DSR(3), F_DSR(4), F_DSR(5), F_DSR(6), F_DSR(7) "\n s_waitcnt lgkmcnt(0)", "memory")
#define F_DSW(i) "ds_write_b32 %9, %" #i
KIND_V(k_ds_write_b32, F_DSW(0), F_DSW(1), F_DSW(2), F_DSW(3), F_DSW(4), F_DSW(5), F_DSW(6), F_DSW(7) "\n s_waitcnt lgkmcnt(0)", "memory")
#define F_DSR128(i) "ds_read_b128 v[32:35], %9"
KIND_V(k_ds_read_b128, F_DSR128(0), F_DSR128(1), F_DSR128(2), F_DSR128(3), F_DSR128(4), F_DSR128(5), F_DSR128(6), F_DSR128(7) "\n s_waitcnt lgkmcnt(0)", "memory", "v32", "v33", "v34", "v35")
#define F_BPERM(i) "ds_bpermute_b32 %" #i ", %9, %" #i
KIND_V(k_ds_bpermute, F_BPERM(0), F_BPERM(1), F_BPERM(2), F_BPERM(3), F_BPERM(4), F_BPERM(5), F_BPERM(6), F_BPERM(7) "\n s_waitcnt lgkmcnt(0)", "memory")

typedef void (*Kern)(float *, int, float, float);

int main() {
	float *out; hipMalloc(&out, 4);
	hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	const double clk = prop.clockRate * 1e3;
	const int trips = 10000;
	struct K { const char *name; Kern k; bool lds; };
	std::vector<K> ks = {
		{ "v_mul_f32", k_mul }, { "v_sub_f32", k_sub }, { "v_max_f32", k_max }, { "v_mov_b32", k_mov }, { "v_add_u32", k_add_u32 }, { "v_and_b32", k_and },
		{ "v_lshlrev_b32", k_lshl }, { "v_lshl_add_u32", k_lshl_add }, { "v_mad_u32_u24", k_mad_u24 }, { "v_bfe_u32", k_bfe },
		{ "v_cndmask_b32 x, x, a, vcc", k_cnd_vcc }, { "v_cndmask_b32 x, x, a, s[20:21]", k_cnd_sgpr }, { "v_cndmask_b32 x, a, b, s[20:21]", k_cnd_other },
		{ "v_cmp_le_f32 vcc", k_cmp_vcc }, { "v_cmp_le_f32 s[20:21]", k_cmp_sgpr }, { "v_cmp_eq_u32 s[20:21]", k_cmpu_sgpr },
		{ "v_rcp_f32", k_rcp }, { "v_div_scale_f32", k_div_scale }, { "v_div_fmas_f32", k_div_fmas }, { "v_div_fixup_f32", k_div_fixup },
		{ "v_mov_b32_dpp quad_perm", k_mov_dpp }, { "v_lshlrev_b64", k_lshl_b64 }, { "v_add_co_u32", k_add_co }, { "v_addc_co_u32", k_addc_co },
		{ "s_and_b64", k_s_and_b64 }, { "ds_read_b32 (+wait per 8)", k_ds_read_b32 }, { "ds_write_b32 (+wait per 8)", k_ds_write_b32 },
		{ "ds_read_b128 (+wait per 8)", k_ds_read_b128 }, { "ds_bpermute_b32 (+wait per 8)", k_ds_bpermute },
	};
	printf("%d CUs, clockRate %.0f MHz; cycles per wave-instruction and SIMD at the nominal clock\n", cus, clk / 1e6);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (const K &k : ks) {
		printf("%-36s", k.name);
		for (int wps = 1; wps <= 8; wps *= 2) {
			float ms = 0;
			for (int rep = 0; rep < 2; ++rep) {
				hipEventRecord(e0);
				// b = lane * 4 as an LDS address for the DS kinds (bit pattern of a small denormal float otherwise)
				hipLaunchKernelGGL(k.k, dim3(cus * wps), dim3(256), 0, 0, out, trips, 1.0000001f, 0.0f);
				hipEventRecord(e1); hipEventSynchronize(e1);
				hipEventElapsedTime(&ms, e0, e1);
			}
			printf("  %d w/SIMD: %6.2f", wps, ms * 1e-3 * clk / ((double) wps * trips * 32));
		}
		printf("\n");
	}
	return 0;
}
