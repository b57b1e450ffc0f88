// Issue-rate microbenchmark of a few VALU instructions on one CU (development helper).
// Each wave runs `iters` x 32 independent instances of one instruction; 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REPEAT8(x) x x x x x x x x
#define BODY(asm_text)                                                                    \
    for (int i = 0; i < iters; i++) {                                                      \
        REPEAT8(asm volatile(asm_text : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc", "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29");) \
    }

template <int OP>
__global__ __launch_bounds__(1024) void bench(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b = seed | 1, c = seed * 11 + 3;
    long long t0 = clock64();
    if (OP == 0) { BODY("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4") }
    if (OP == 1) { BODY("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc") }
    if (OP == 2) { BODY("v_cndmask_b32_e64 %0, %0, %4, s[10:11]\n v_cndmask_b32_e64 %1, %1, %4, s[10:11]\n v_cndmask_b32_e64 %2, %2, %4, s[10:11]\n v_cndmask_b32_e64 %3, %3, %4, s[10:11]") }
    if (OP == 3) { BODY("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %4, %5, vcc") }
    if (OP == 4) { BODY("v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %4\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %4") }
    if (OP == 5) { BODY("v_cmp_lt_f32_e64 s[10:11], %0, %4\n v_cmp_lt_f32_e64 s[12:13], %1, %4\n v_cmp_lt_f32_e64 s[14:15], %2, %4\n v_cmp_lt_f32_e64 s[16:17], %3, %4") }
    if (OP == 6) { BODY("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_cmp_lt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %5, vcc") }
    if (OP == 7) { BODY("v_min_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_min_f32 %2, %2, %4\n v_min_f32 %3, %3, %4") }
    if (OP == 8) { BODY("v_pk_mul_f32 v[20:21], v[20:21], v[28:29]\n v_pk_mul_f32 v[22:23], v[22:23], v[28:29]\n v_pk_mul_f32 v[24:25], v[24:25], v[28:29]\n v_pk_mul_f32 v[26:27], v[26:27], v[28:29]") }
    if (OP == 9) { BODY("v_div_fmas_f32 %0, %0, %4, %5\n v_div_fmas_f32 %1, %1, %4, %5\n v_div_fmas_f32 %2, %2, %4, %5\n v_div_fmas_f32 %3, %3, %4, %5") }
    if (OP == 10) { BODY("v_bcnt_u32_b32 %0, %0, %4\n v_bcnt_u32_b32 %1, %1, %4\n v_bcnt_u32_b32 %2, %2, %4\n v_bcnt_u32_b32 %3, %3, %4") }
    if (OP == 11) { BODY("v_ffbh_u32 %0, %0\n v_ffbh_u32 %1, %1\n v_ffbh_u32 %2, %2\n v_ffbh_u32 %3, %3") }
    if (OP == 12) { BODY("v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %4, %5\n v_and_or_b32 %2, %2, %4, %5\n v_and_or_b32 %3, %3, %4, %5") }
    if (OP == 13) { BODY("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %5\n v_mov_b32 %3, %5") }
    if (OP == 14) { BODY("v_sub_f32 %0, %0, %4\n v_sub_f32 %1, %1, %4\n v_sub_f32 %2, %2, %4\n v_sub_f32 %3, %3, %4") }
    if (OP == 15) { BODY("v_mul_f32 %0, s20, %0\n v_mul_f32 %1, s20, %1\n v_mul_f32 %2, s20, %2\n v_mul_f32 %3, s20, %3") }
    if (OP == 16) { BODY("v_fma_f32 %0, s20, %0, %5\n v_fma_f32 %1, s20, %1, %5\n v_fma_f32 %2, s20, %2, %5\n v_fma_f32 %3, s20, %3, %5") }
    if (OP == 17) { BODY("v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_addc_co_u32 %1, vcc, %1, %1, vcc\n v_addc_co_u32 %2, vcc, %2, %2, vcc\n v_addc_co_u32 %3, vcc, %3, %3, vcc") }
    if (OP == 18) { BODY("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5") }
    if (OP == 19) { BODY("v_max_f32_e64 %0, abs(%0), neg(%4)\n v_max_f32_e64 %1, abs(%1), neg(%4)\n v_max_f32_e64 %2, abs(%2), neg(%4)\n v_max_f32_e64 %3, abs(%3), neg(%4)") }
    if (OP == 20) { BODY("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4") }
    if (OP == 21) { BODY("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4") }
    if (OP == 22) { BODY("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4") }
    if (OP == 23) { BODY("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3") }
    if (OP == 24) { BODY("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3") }
    if (OP == 25) { BODY("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3") }
    if (OP == 26) { BODY("v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %1, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %2, %2, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_xor_b32_sdwa %3, %3, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD") }
    if (OP == 27) { BODY("v_lshl_add_u32 %0, %0, 3, %4\n v_lshl_add_u32 %1, %1, 3, %4\n v_lshl_add_u32 %2, %2, 3, %4\n v_lshl_add_u32 %3, %3, 3, %4") }
    if (OP == 28) { BODY("v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5") }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
    if (threadIdx.x == 0) reinterpret_cast<long long*>(out + 4096)[OP] = t1 - t0;
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 1 << 20);
    const char* names[] = {"v_add_u32", "v_cndmask_e32 vcc chain", "v_cndmask_e64 s[10:11]", "v_cndmask_e32 src vgprs only", "v_cmp_lt_f32 vcc", "v_cmp_lt_f32_e64 sgpr", "v_cmp+v_cndmask pairs", "v_min_f32", "v_pk_mul_f32 (2 lanes of work)", "v_div_fmas_f32", "v_bcnt_u32_b32", "v_ffbh_u32", "v_and_or_b32", "v_mov_b32", "v_sub_f32", "v_mul_f32 with sgpr", "v_fma_f32 with sgpr", "v_addc_co_u32", "v_mad_f32/v_fmac", "v_max_f32 abs/neg mods (VOP3)", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_rcp_f32", "v_sqrt_f32", "v_cvt_f32_u32", "v_xor_b32 sdwa", "v_lshl_add_u32", "v_mad_u32_u24"};
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(OP)                                                                              \
    {                                                                                        \
        bench<OP><<<1, 1024>>>(d, 10, 1);                                                    \
        hipDeviceSynchronize();                                                              \
        hipEventRecord(e0);                                                                  \
        bench<OP><<<1, 1024>>>(d, iters, 1);                                                 \
        hipEventRecord(e1);                                                                  \
        hipDeviceSynchronize();                                                              \
        float ms;                                                                            \
        hipEventElapsedTime(&ms, e0, e1);                                                    \
        long long clk;                                                                       \
        hipMemcpy(&clk, reinterpret_cast<long long*>(d + 4096) + OP, 8, hipMemcpyDeviceToHost); \
        double per_simd_instr = (double)iters * 32 * 4;  /* 4 waves per SIMD */              \
        printf("%-28s %8.3f ms  %6.2f ns per wave-instruction per SIMD   (clock64 delta %lld)\n", names[OP], ms, ms * 1e6 / per_simd_instr, clk); \
    }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17) RUN(18) RUN(19) RUN(20) RUN(21) RUN(22) RUN(23) RUN(24) RUN(25) RUN(26) RUN(27) RUN(28)
    return 0;
}
