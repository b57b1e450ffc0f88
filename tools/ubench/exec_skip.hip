// Does a VALU instruction cost less when whole 16-lane quarters of the wave are inactive?  (development helper)
// One workgroup of 16 waves (4 per SIMD); every wave runs the same chain of independent v_fma_f32 under an
// EXEC mask given on the command line of the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(1024) void bench(float* out, int iters, unsigned long long mask) {
    float a0 = threadIdx.x, a1 = a0 * 3.0f, a2 = a0 * 5.0f, a3 = a0 * 7.0f;
    const float b = 1.0001f, c = 0.5f;
    const bool on = (mask >> (threadIdx.x & 63)) & 1ull;
    if (on) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int k = 0; k < 8; k++)
                asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

int main() {
    float* d;
    hipMalloc(&d, 1 << 20);
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    struct { const char* name; unsigned long long mask; } cases[] = {
        {"all 64 lanes", ~0ull},
        {"lanes 0-31", 0xffffffffull},
        {"lanes 0-15", 0xffffull},
        {"lanes 16-31", 0xffff0000ull},
        {"lanes 48-63", 0xffff000000000000ull},
        {"every 4th lane (16 lanes over all quarters)", 0x1111111111111111ull},
        {"every 2nd lane (32 lanes over all quarters)", 0x5555555555555555ull},
        {"lanes 0-15 and 32-47", 0x0000ffff0000ffffull},
        {"one lane", 1ull},
        {"one lane per quarter", 0x0001000100010001ull},
    };
    for (int rep = 0; rep < 2; rep++)
        for (auto& cs : cases) {
            bench<<<1, 1024>>>(d, 10, cs.mask);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            bench<<<1, 1024>>>(d, iters, cs.mask);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-48s %8.3f ms  %6.2f ns per wave-instruction per SIMD\n", cs.name, ms, ms * 1e6 / ((double)iters * 32 * 4));
        }
    return 0;
}
