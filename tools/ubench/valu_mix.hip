// Issue cost of VALU instruction kinds on the whole chip (development helper): 2048 workgroups of 256 threads,
// every wave runs chains of ONE instruction kind; time per wave-instruction per SIMD, clocks held up by a long run.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REPEAT8(x) x x x x x x x x
#define BODY(asm_text)                                                                    \
    for (int i = 0; i < iters; i++) {                                                      \
        REPEAT8(asm volatile(asm_text : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c), "s"(sb) : "vcc", "s10", "s11");) \
    }

template <int OP>
__global__ __launch_bounds__(256) void bench(float* out, int iters, float seed, unsigned long long mask) {
    float a0 = threadIdx.x + seed, a1 = a0 * 3.0f, a2 = a0 * 5.0f, a3 = a0 * 7.0f;
    const float b = 1.0001f + seed, c = 0.5f + seed;
    const float sb = seed + 1.5f;
    if ((mask >> (threadIdx.x & 63)) & 1ull) {
    if (OP == 0) { BODY("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5") }
    if (OP == 1) { BODY("v_fma_f32 %0, %0, %6, %5\n v_fma_f32 %1, %1, %6, %5\n v_fma_f32 %2, %2, %6, %5\n v_fma_f32 %3, %3, %6, %5") }
    if (OP == 2) { BODY("v_sub_f32 %0, %0, %4\n v_sub_f32 %1, %1, %4\n v_sub_f32 %2, %2, %4\n v_sub_f32 %3, %3, %4") }
    if (OP == 3) { BODY("v_sub_f32 %0, %6, %0\n v_sub_f32 %1, %6, %1\n v_sub_f32 %2, %6, %2\n v_sub_f32 %3, %6, %3") }
    if (OP == 4) { BODY("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4") }
    if (OP == 5) { BODY("v_mul_f32 %0, %6, %0\n v_mul_f32 %1, %6, %1\n v_mul_f32 %2, %6, %2\n v_mul_f32 %3, %6, %3") }
    if (OP == 6) { BODY("v_min_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_min_f32 %2, %2, %4\n v_min_f32 %3, %3, %4") }
    if (OP == 7) { BODY("v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %1, %1, %4, %5\n v_max3_f32 %2, %2, %4, %5\n v_max3_f32 %3, %3, %4, %5") }
    if (OP == 8) { BODY("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4") }
    if (OP == 9) { BODY("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4") }
    if (OP == 10) { BODY("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4") }
    if (OP == 11) { BODY("v_cndmask_b32_e64 %0, %0, %4, s[10:11]\n v_cndmask_b32_e64 %1, %1, %4, s[10:11]\n v_cndmask_b32_e64 %2, %2, %4, s[10:11]\n v_cndmask_b32_e64 %3, %3, %4, s[10:11]") }
    if (OP == 12) { BODY("v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %4\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %4") }
    if (OP == 13) { BODY("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %5\n v_mov_b32 %3, %5") }
    if (OP == 14) { BODY("v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3") }
    if (OP == 15) { BODY("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4") }
    if (OP == 16) { BODY("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3") }
    if (OP == 17) { BODY("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5") }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
template <>
__global__ __launch_bounds__(256) void bench<17>(float* out, int iters, float seed, unsigned long long mask) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {threadIdx.x + seed, 1.0f}, a1 = a0 * 3.0f, a2 = a0 * 5.0f, a3 = a0 * 7.0f;
    const f2 b = {1.0001f + seed, 1.0f}, c = {0.5f + seed, 0.25f};
    if ((mask >> (threadIdx.x & 63)) & 1ull) {
        for (int i = 0; i < iters; i++) {
            REPEAT8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a1.x + a2.y + a3.y;
}

int main() {
    float* d;
    hipMalloc(&d, 64 << 20);
    const char* names[] = {"v_fma_f32 vgprs", "v_fma_f32 one sgpr", "v_sub_f32 vgprs", "v_sub_f32 sgpr", "v_mul_f32 vgprs", "v_mul_f32 sgpr", "v_min_f32", "v_max3_f32", "v_add_u32", "v_xor_b32", "v_mul_lo_u32", "v_cndmask_e64 sgpr mask", "v_cmp_lt_f32 vcc", "v_mov_b32", "v_lshrrev_b32", "v_add_f32 vgprs", "v_rcp_f32", "v_pk_fma_f32"};
    const int iters = 3000, blocks = 4096;       // 256 CUs x 4 blocks (4 waves per SIMD) x 4 rounds
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // warm the clocks up
    for (int k = 0; k < 20; k++) bench<0><<<blocks, 256>>>(d, iters, 0.0f, ~0ull);
    hipDeviceSynchronize();
#define RUN(OP, MASK, LABEL)                                                                  \
    {                                                                                        \
        bench<OP><<<blocks, 256>>>(d, iters, 0.0f, MASK);                                    \
        hipEventRecord(e0);                                                                  \
        for (int k = 0; k < 3; k++) bench<OP><<<blocks, 256>>>(d, iters, 0.0f, MASK);        \
        hipEventRecord(e1);                                                                  \
        hipDeviceSynchronize();                                                              \
        float ms;                                                                            \
        hipEventElapsedTime(&ms, e0, e1);                                                    \
        double per_simd = 3.0 * (double)iters * 32 * (blocks * 4.0 / 1024.0);                \
        printf("%-26s %-34s %8.3f ms  %6.3f ns per wave-instruction per SIMD\n", names[OP], LABEL, ms, ms * 1e6 / per_simd); \
    }
    RUN(0, ~0ull, "all lanes") RUN(1, ~0ull, "all lanes") RUN(2, ~0ull, "all lanes") RUN(3, ~0ull, "all lanes") RUN(4, ~0ull, "all lanes") RUN(5, ~0ull, "all lanes")
    RUN(6, ~0ull, "all lanes") RUN(7, ~0ull, "all lanes") RUN(8, ~0ull, "all lanes") RUN(9, ~0ull, "all lanes") RUN(10, ~0ull, "all lanes") RUN(11, ~0ull, "all lanes")
    RUN(12, ~0ull, "all lanes") RUN(13, ~0ull, "all lanes") RUN(14, ~0ull, "all lanes") RUN(15, ~0ull, "all lanes") RUN(16, ~0ull, "all lanes") RUN(17, ~0ull, "all lanes")
    RUN(0, 0xffffffffull, "lanes 0-31") RUN(0, 0xffffull, "lanes 0-15") RUN(0, 0x1111111111111111ull, "every 4th lane") RUN(0, 1ull, "one lane") RUN(0, 0x0001000100010001ull, "one lane per quarter")
    RUN(0, 0x3ull, "two lanes") RUN(0, 0xfull, "lanes 0-3") RUN(0, 0xffull, "lanes 0-7") RUN(0, 0x0101010101010101ull, "every 8th lane") RUN(0, 0x00010001ull, "lanes 0 and 16") RUN(0, 0x100000001ull, "lanes 0 and 32")
    RUN(0, 0x1ffull, "9 lanes") RUN(0, 0x3ffull, "10 lanes") RUN(0, 0xfffull, "12 lanes") RUN(0, 0x3fffull, "14 lanes") RUN(0, 0x7fffull, "15 lanes") RUN(0, 0x1ffffull, "17 lanes") RUN(0, 0x00ff00ffull, "8 + 8 lanes in two quarters") RUN(0, 0x000f000f000f000full, "4 lanes in each quarter")
    RUN(6, 0xffffffffull, "lanes 0-31") RUN(6, 0xffffull, "lanes 0-15") RUN(6, 0x1111111111111111ull, "every 4th lane") RUN(6, 1ull, "one lane") RUN(6, 0x0001000100010001ull, "one lane per quarter")
    return 0;
}
