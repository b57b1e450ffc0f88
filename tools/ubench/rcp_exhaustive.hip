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

// The same division with numerators from the subnormals up to 2^-40 (exponent field 0..87) over divisors within
// 2^+-48: where does the correction step stop giving the IEEE quotient (its remainder or the quotient itself
// underflows)?  Reports the largest |a| and the largest |a / s| at which a mismatch was seen.
__global__ void check_div_tiny(unsigned long long* bad, uint32_t* worst) {
    unsigned long long b = 0, guarded = 0;
    uint32_t state = 0x85ebca6bu * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    for (int k = 0; k < 4096; k++) {
        state ^= state << 13; state ^= state >> 17; state ^= state << 5;
        uint32_t abits = state;
        state ^= state << 13; state ^= state >> 17; state ^= state << 5;
        uint32_t sbits = state;
        abits = (abits & 0x807fffffu) | (((abits >> 23) % 88u) << 23);
        sbits = (sbits & 0x807fffffu) | (((sbits >> 23) % 97u + 79u) << 23);
        const float a = __uint_as_float(abits), s = __uint_as_float(sbits);
        const float r = rcp_nr1(s);
        const float q0 = a * r;
        const float rem = __builtin_fmaf(-s, q0, a);
        const float q = a == 0.0f ? q0 : __builtin_fmaf(rem, r, q0);
        const float want = a / s;
        if (__float_as_uint(q) != __float_as_uint(want)) {
            b++;
            atomicMax(&worst[0], abits & 0x7fffffffu);
            atomicMax(&worst[1], __float_as_uint(want) & 0x7fffffffu);
            // the kernels' quotient() sends numerators below 2^-80 through the ordinary division
            if (fabsf(a) >= 0x1p-80f) guarded++;
        }
    }
    atomicAdd(&bad[5], b);
    atomicAdd(&bad[6], guarded);
}

int main() {
    unsigned long long* bad;
    uint32_t* ex;
    (void)hipMalloc(&bad, 64);
    (void)hipMalloc(&ex, 64);
    (void)hipMemset(bad, 0, 64);
    (void)hipMemset(ex, 0, 64);
    check<<<4096, 256>>>(bad, ex);
    check_div<<<4096, 256>>>(bad, ex, 0);
    check_div<<<4096, 256>>>(bad, ex, 1);
    check_div_tiny<<<4096, 256>>>(bad, ex + 4);
    (void)hipDeviceSynchronize();
    unsigned long long h[7];
    uint32_t e[6];
    (void)hipMemcpy(h, bad, 56, hipMemcpyDeviceToHost);
    (void)hipMemcpy(e, ex, 24, hipMemcpyDeviceToHost);
    printf("raw v_rcp_f32 (control): %llu mismatches\n", h[4]);
    printf("reciprocal, one Newton step:  %llu mismatches (last example bits 0x%08x)\n", h[0], e[0]);
    printf("reciprocal, two Newton steps: %llu mismatches (last example bits 0x%08x)\n", h[1], e[1]);
    printf("division via reciprocal (1 step): %llu mismatches of %llu sampled pairs (a 0x%08x s 0x%08x)\n", h[2], 4096ull * 4096 * 256, e[2], e[3]);
    printf("division via reciprocal (2 steps): %llu mismatches\n", h[3]);
    float worst_a, worst_q;
    __builtin_memcpy(&worst_a, &e[4], 4);
    __builtin_memcpy(&worst_q, &e[5], 4);
    printf("division via reciprocal, numerators below 2^-39: %llu mismatches of %llu sampled pairs; largest |a| with one %g (bits 0x%08x), largest |a/s| with one %g\n",
           h[5], 4096ull * 4096 * 256, (double)worst_a, e[4], (double)worst_q);
    printf("of those with a numerator of 2^-80 or more (which the kernels do not send through the ordinary division): %llu mismatches\n", h[6]);
    return 0;
}
