// How many random cache lines per second can the chip bring in?  Every lane reads whole 128-byte
// records (eight 16-byte loads, like a BVH node) at hashed positions of a region of a given size;
// the next position depends on the data read (like a tree walk: the child to visit is known only
// after the node has arrived), with `chains` independent walks per lane in flight.
//   ./random_fetch            -> table: region size x waves per SIMD -> records/s, GB/s
//   ./random_fetch shared     -> the same with the loads of a step shared out inside groups of eight lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void walk(const float4* records, uint32_t mask, int steps, float* sink) {
    uint32_t at = hash32(blockIdx.x * 256u + threadIdx.x) & mask;
    float acc = 0.0f;
    for (int s = 0; s < steps; s++) {
        const float4* r = records + (size_t)at * 8;
        float4 a = r[0], b = r[1], c = r[2], d = r[3], e = r[4], f = r[5], g = r[6], h = r[7];
        acc += a.x + b.y + c.z + d.w + e.x + f.y + g.z + h.w;
        at = hash32(at + (uint32_t)__float_as_uint(a.x) + (uint32_t)s) & mask;     // depends on the data
    }
    if (acc == 12345.678f) *sink = acc;
}

// The same walk with the loads shared out inside groups of eight lanes: instruction k of a step
// fetches the record that lane k of the group wants, each of the eight lanes one 16-byte quad of it,
// so one load instruction touches 8 cache lines instead of 64 (every line whole).  The quads reach
// their owner through LDS (144-byte rows: a lane's eight 16-byte reads are conflict-free).
__device__ __forceinline__ void wave_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void walk_shared(const float4* records, uint32_t mask, int steps, float* sink) {
    __shared__ float4 stage_all[4][64 * 9];
    float4* stage = stage_all[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63, q = lane & 7, group = lane & ~7;
    uint32_t at = hash32(blockIdx.x * 256u + threadIdx.x) & mask;
    float acc = 0.0f;
    for (int s = 0; s < steps; s++) {
        float4 part[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t other = (uint32_t)__shfl((int)at, group | k);
            part[k] = records[(size_t)other * 8 + q];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) stage[(group | k) * 9 + q] = part[k];
        wave_lds_order();
        const float4* r = stage + lane * 9;
        float4 a = r[0], b = r[1], c = r[2], d = r[3], e = r[4], f = r[5], g = r[6], h = r[7];
        wave_lds_order();
        acc += a.x + b.y + c.z + d.w + e.x + f.y + g.z + h.w;
        at = hash32(at + (uint32_t)__float_as_uint(a.x) + (uint32_t)s) & mask;
    }
    if (acc == 12345.678f) *sink = acc;
}

int main(int argc, char** argv) {
    const bool shared_loads = argc > 1 && argv[1][0] == 's';
    const size_t max_records = (size_t)1 << 24;                 // 2 GiB
    float4* mem = nullptr;
    float* sink = nullptr;
    CHECK(hipMalloc(&mem, max_records * 128));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(mem, 0, max_records * 128));
    hipEvent_t t0, t1;
    CHECK(hipEventCreate(&t0));
    CHECK(hipEventCreate(&t1));
    int cus = 256;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    std::printf("region MiB, waves/SIMD, records/s (G), GB/s of 128-byte records, ns per dependent step\n");
    for (int log2n = 14; log2n <= 24; log2n += 2) {
        const uint32_t mask = (1u << log2n) - 1u;
        for (int waves : {4, 8}) {
            if (shared_loads && waves == 8) continue;          // 36 KB of LDS per block: four blocks per CU
            const int blocks = cus * waves, steps = 2000;
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(t0));
                if (shared_loads) {
                    if (waves == 4) hipLaunchKernelGGL(walk_shared<4>, dim3(blocks), dim3(256), 0, 0, mem, mask, steps, sink);
                    else hipLaunchKernelGGL(walk_shared<8>, dim3(blocks), dim3(256), 0, 0, mem, mask, steps, sink);
                } else if (waves == 4) hipLaunchKernelGGL(walk<4>, dim3(blocks), dim3(256), 0, 0, mem, mask, steps, sink);
                else hipLaunchKernelGGL(walk<8>, dim3(blocks), dim3(256), 0, 0, mem, mask, steps, sink);
                CHECK(hipEventRecord(t1));
                CHECK(hipEventSynchronize(t1));
                float ms = 0.0f;
                CHECK(hipEventElapsedTime(&ms, t0, t1));
                if (rep == 1) {
                    const double recs = (double)blocks * 256 * steps;
                    std::printf("%8.1f  %d  %8.2f  %8.1f  %8.1f\n", (double)(mask + 1) * 128 / 1048576.0, waves, recs / ms / 1e6, recs * 128 / ms / 1e6, ms * 1e6 / steps);
                }
            }
        }
    }
    return 0;
}
