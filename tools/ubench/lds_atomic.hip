// What does an LDS float addition cost?  Every wave adds to its own 64 floats of LDS; the lanes'
// addresses are all different, pairwise equal (lane and lane + 32, what two samples of one pixel
// in a wave look like), four or 64 lanes on one address, or only half of the lanes add.  Compared with a plain
// read-add-write of the same words (which is only right when no two lanes share an address).
//   ./lds_atomic   -> pattern, ns per wave-level instruction per CU, lane-additions per ns on the chip
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int PATTERN>
__global__ __launch_bounds__(256, 4) void add_loop(int steps, float* sink) {
    __shared__ float sums[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    sums[wave][lane] = 0.0f;
    __syncthreads();
    int slot = lane;
    if (PATTERN == 1) slot = lane & 31;
    if (PATTERN == 2) slot = lane & 15;
    if (PATTERN == 3) slot = 0;
    float v = 1.0f + lane;
    for (int s = 0; s < steps; s++) {
        if (PATTERN == 4) { if (lane < 32) atomicAdd(&sums[wave][slot], v); }
        else if (PATTERN == 5) { float o = sums[wave][slot]; sums[wave][slot] = o + v; }
        else atomicAdd(&sums[wave][slot], v);
        v += 1.0f;
    }
    __syncthreads();
    if (sums[wave][lane] == 12345.678f) *sink = v;
}

int main() {
    float* sink = nullptr;
    CHECK(hipMalloc(&sink, 4));
    hipEvent_t t0, t1;
    CHECK(hipEventCreate(&t0));
    CHECK(hipEventCreate(&t1));
    int cus = 256;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const char* names[6] = {"64 addresses", "32 addresses (pairs)", "16 addresses (fours)", "1 address", "32 lanes, 32 addresses", "read-add-write, 64 addresses"};
    const int steps = 20000, blocks = cus * 4;
    std::printf("pattern, ns per wave-level addition per CU, lane-additions per ns (chip)\n");
    for (int p = 0; p < 6; p++) {
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(t0));
            switch (p) {
            case 0: hipLaunchKernelGGL(add_loop<0>, dim3(blocks), dim3(256), 0, 0, steps, sink); break;
            case 1: hipLaunchKernelGGL(add_loop<1>, dim3(blocks), dim3(256), 0, 0, steps, sink); break;
            case 2: hipLaunchKernelGGL(add_loop<2>, dim3(blocks), dim3(256), 0, 0, steps, sink); break;
            case 3: hipLaunchKernelGGL(add_loop<3>, dim3(blocks), dim3(256), 0, 0, steps, sink); break;
            case 4: hipLaunchKernelGGL(add_loop<4>, dim3(blocks), dim3(256), 0, 0, steps, sink); break;
            default: hipLaunchKernelGGL(add_loop<5>, dim3(blocks), dim3(256), 0, 0, steps, sink); break;
            }
            CHECK(hipEventRecord(t1));
            CHECK(hipEventSynchronize(t1));
            float ms = 0.0f;
            CHECK(hipEventElapsedTime(&ms, t0, t1));
            if (rep == 1) {
                const double wave_ops_per_cu = 16.0 * steps;           // 4 blocks x 4 waves per CU
                const double lanes = (p == 4 ? 32.0 : 64.0) * blocks * 4 * steps;
                std::printf("%-30s %8.2f %10.1f\n", names[p], ms * 1e6 / wave_ops_per_cu, lanes / (ms * 1e6));
            }
        }
    }
    return 0;
}
