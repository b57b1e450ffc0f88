// Exhaustive check (development helper): for every float x with 2^-100 <= |x| <= 2^100, does a
// Newton-refined v_rcp_f32 equal the correctly rounded 1.0f / x?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ float rcp_nr1(float x) {
    float r0 = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}
__device__ __forceinline__ float rcp_nr2(float x) {
    float r = rcp_nr1(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}

__global__ void check(unsigned long long* bad, uint32_t* example) {
    const uint32_t lo = (127u - 100u) << 23, hi = (127u + 100u) << 23;
    unsigned long long b0 = 0, b1 = 0, b2 = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t bits = (uint32_t)i, mag = bits & 0x7fffffffu;
        if (mag < lo || mag > hi) continue;
        const float x = __uint_as_float(bits);
        const float want = 1.0f / x;
        if (__float_as_uint(__builtin_amdgcn_rcpf(x)) != __float_as_uint(want)) b0++;
        if (__float_as_uint(rcp_nr1(x)) != __float_as_uint(want)) { b1++; example[0] = bits; }
        if (__float_as_uint(rcp_nr2(x)) != __float_as_uint(want)) { b2++; example[1] = bits; }
    }
    atomicAdd(&bad[4], b0);
    atomicAdd(&bad[0], b1);
    atomicAdd(&bad[1], b2);
}

// Division a / s through the refined reciprocal (Markstein): q0 = a*r, rem = a - s*q0 (one fma,
// exact), q = q0 + rem*r.  Sampled pairs, since 2^64 cannot be enumerated.
__global__ void check_div(unsigned long long* bad, uint32_t* example, int use_nr2) {
    unsigned long long b = 0;
    uint32_t state = 0x9e3779b9u * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    for (int k = 0; k < 4096; k++) {
        state ^= state << 13; state ^= state >> 17; state ^= state << 5;
        uint32_t abits = state;
        state ^= state << 13; state ^= state >> 17; state ^= state << 5;
        uint32_t sbits = state;
        // exponents within +-40 of 1.0
        abits = (abits & 0x807fffffu) | (((abits >> 23) % 81u + 87u) << 23);
        sbits = (sbits & 0x807fffffu) | (((sbits >> 23) % 81u + 87u) << 23);
        const float a = __uint_as_float(abits), s = __uint_as_float(sbits);
        const float r = use_nr2 ? rcp_nr2(s) : rcp_nr1(s);
        const float q0 = a * r;
        const float rem = __builtin_fmaf(-s, q0, a);
        const float q = __builtin_fmaf(rem, r, q0);
        if (__float_as_uint(q) != __float_as_uint(a / s)) { b++; example[2] = abits; example[3] = sbits; }
    }
    atomicAdd(&bad[2 + use_nr2], b);
}

int main() {
    unsigned long long* bad;
    uint32_t* ex;
    hipMalloc(&bad, 64);
    hipMalloc(&ex, 64);
    hipMemset(bad, 0, 64);
    hipMemset(ex, 0, 64);
    check<<<4096, 256>>>(bad, ex);
    check_div<<<4096, 256>>>(bad, ex, 0);
    check_div<<<4096, 256>>>(bad, ex, 1);
    hipDeviceSynchronize();
    unsigned long long h[5];
    uint32_t e[4];
    hipMemcpy(h, bad, 40, hipMemcpyDeviceToHost);
    hipMemcpy(e, ex, 16, hipMemcpyDeviceToHost);
    printf("raw v_rcp_f32 (control): %llu mismatches\n", h[4]);
    printf("reciprocal, one Newton step:  %llu mismatches (last example bits 0x%08x)\n", h[0], e[0]);
    printf("reciprocal, two Newton steps: %llu mismatches (last example bits 0x%08x)\n", h[1], e[1]);
    printf("division via reciprocal (1 step): %llu mismatches of %llu sampled pairs (a 0x%08x s 0x%08x)\n", h[2], 4096ull * 4096 * 256, e[2], e[3]);
    printf("division via reciprocal (2 steps): %llu mismatches\n", h[3]);
    return 0;
}
