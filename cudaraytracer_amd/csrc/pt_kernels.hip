// Kernels and the device half of the C ABI (crt_scene_*, crt_buffers_*, crt_render*).
//
// Replaces render() / render_kernel (src/render.cuh:108-181) and the per-element scene
// upload of scene::scene (src/scene.cuh:61-89).
//
// Kernel structure
//   render_fused   persistent waves.  A wave takes work items (an 8x4 pixel tile x a chunk of the
//                  samples) from a global counter; every lane carries one path.  A lane whose
//                  path ended starts the next path of the item in the same iteration in which
//                  the other lanes trace their next bounce, so divergence over path LENGTH
//                  (1..max_bounces) never idles lanes, and two items are open at a time so the
//                  last long paths of one item overlap the first paths of the next.  Finished
//                  paths add their results to the item's tile sums in LDS; one store of
//                  14 floats per (pixel, chunk) is all that reaches HBM.
//   resolve_passes adds the chunk sums in chunk order, folds them into the running mean of
//                  the pass buffers ((old*done + sum) / (done + n), which for n = 1 is the
//                  reference's update at render.cuh:147-152) and writes the reference layout.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "crt_internal.h"
#include "pt_device.h"

namespace crt {

// Waves per SIMD a kernel variant is compiled for (its register budget is 512 / waves VGPRs) and the
// persistent grid is sized for.  The flat kernel is bound by instruction issue with many short
// dependent chains: it gains from every resident wave until spills set in (Cornell: 9.5 ms at four
// waves, 8.4 at five, 8.2 at six, 9.1 at seven); its variant for basic scenes needs 82-92 registers,
// the general one about 110.  The tree walks need all 128.
#ifndef CRT_FLAT_BASIC_WAVES_PER_SIMD
#define CRT_FLAT_BASIC_WAVES_PER_SIMD 6
#endif
#ifndef CRT_FLAT_WAVES_PER_SIMD
#define CRT_FLAT_WAVES_PER_SIMD 6            // (five were best while the tile sums were LDS atomics; materials 13.6 -> 13.1 ms with six since)
#endif
#ifndef CRT_TREE_BASIC_WAVES_PER_SIMD
#define CRT_TREE_BASIC_WAVES_PER_SIMD 4      // tree walk over untransformed triangles only
#endif
#ifndef CRT_DEEP_STACK_ROWS
#define CRT_DEEP_STACK_ROWS 12
#endif
constexpr int kDeepStackRows = CRT_DEEP_STACK_ROWS;   // LDS rows of the traversal stack for trees that need more than 7 entries
#ifndef CRT_TREE_TILE_ROWS
#define CRT_TREE_TILE_ROWS 4
#endif
#ifndef CRT_TREE_WAVES_PER_SIMD
#define CRT_TREE_WAVES_PER_SIMD 4
#endif
constexpr int kBlock = 256;          // 4 waves
constexpr int kWaves = kBlock / 64;
constexpr int kPasses = 14;          // color3 albedo3 emission3 normal3 depth opacity
// Pixels of one work item: 8 wide, 4 rows (3.5 KB of pass sums per wave and open item: six
// workgroups of the flat kernel, four of the tree walks with their traversal stacks, fit a CU's LDS).
constexpr int kTileW = 8;
#ifndef CRT_FLAT_TILE_ROWS
#define CRT_FLAT_TILE_ROWS 4
#endif
constexpr int tile_rows(int mode) { return mode == 2 ? CRT_FLAT_TILE_ROWS : CRT_TREE_TILE_ROWS; }
constexpr int waves_per_simd(int mode, int feat) {
    if (mode != 2) return feat == kFeatBasic ? CRT_TREE_BASIC_WAVES_PER_SIMD : CRT_TREE_WAVES_PER_SIMD;
    return feat == kFeatBasicXform ? CRT_FLAT_BASIC_WAVES_PER_SIMD : CRT_FLAT_WAVES_PER_SIMD;
}

// How a kernel reaches the scene geometry
enum : int {
    kModeTreeGlobal = 0,   // BVH4 walk, node/primitive records read from HBM through L1/L2
    kModeTreeLds = 1,      // BVH4 walk over a field-major copy of the scene in LDS
    kModeFlat = 2          // <= 64 box slots: wave-uniform box loop + per-lane candidate mask
};

// Work-item counters of a launch: the ray count in the first words, then kQueues counters kQueueStride words apart
// (each in a memory channel of its own).
constexpr int kQueues = 8;
constexpr int kQueueStride = 64;
// Behind them: how many items fell into each class of measured cost, and how many of each class have been placed in
// the order of the next launch (order_by_cost).
constexpr int kCostClasses = 128;
constexpr int kCostWordsAt = (kQueues + 1) * kQueueStride;
constexpr size_t kCounterBytes = (size_t)(kCostWordsAt + 2 * kCostClasses) * sizeof(unsigned int);
static_assert(kCounterBytes % 256 == 0, "the chunk sums behind the counters stay 256-byte aligned");
constexpr int kMaxOrderedItems = 1 << 19;     // frames cut into more items than this run in the natural order
// A scene's own counters are followed by the cost of every item of the last launch and the order made of them
// (kMaxOrderedItems words each); these word offsets are relative to RenderArgs::queue.
constexpr int kItemCostFromQueue = (int)(kCounterBytes / sizeof(unsigned int)) - kQueueStride;
constexpr int kItemOrderFromQueue = kItemCostFromQueue + kMaxOrderedItems;

// Class of an item's cost (ticks of the 100 MHz clock it was open for): 2^CRT_COST_CLASS_BITS classes per power of two.
#ifndef CRT_COST_CLASS_BITS
#define CRT_COST_CLASS_BITS 2
#endif
__host__ __device__ __forceinline__ int cost_class(unsigned int ticks) {
    const float f = (float)(ticks > 0u ? ticks : 1u);
    unsigned int bits;
    __builtin_memcpy(&bits, &f, 4);
    const int k = (int)(bits >> (23 - CRT_COST_CLASS_BITS)) - (127 << CRT_COST_CLASS_BITS);
    return k < 0 ? 0 : (k >= kCostClasses ? kCostClasses - 1 : k);
}

struct RenderArgs {
    DevScene sc;
    FlatView flat;                   // small-scene box slots (flat mode)
    float* partial;                  // [chunks][14][pix_padded]  chunk sums, tile-major pixels
    unsigned long long* ray_counter;
    int* stack_spill;                // tree modes: global part of the traversal stacks, [entries][spill_threads]
    int spill_threads;
    unsigned int* queue;             // next work item (tile, chunk) to hand out; zero at launch
    // (kQueues counters, kQueueStride words apart, each handing out the items i with i % kQueues == its number)
    // Launch order of the items (kept behind the counters, see kCostWordsAt ...: no further pointers to hold in
    // registers).  order_mode 0: the n-th item handed out is item n.  1: the same, and every finished item leaves
    // the time it was open in the cost array; the host turns them into the
    // order of the next launch of the same frame, most expensive items first.  2: as 1, and the n-th item handed
    // out is order[n].
    int order_mode;
    int tiles_x, tiles_y, ntiles, pix_padded;
    int first_sample, samples, chunk_size, chunks;
    int prim_batch;                  // tree modes: lanes with leaf candidates that trigger a round of primitive tests
    int ready_percent;               // tree modes: shade once this share of the wave's rays has finished its walk
    // Row bands (multi-GPU tile sharding): the launch covers the tile rows g with
    // (g / band_tiles) % band_count == band_index; tiles_y counts only those.  A launch of the whole
    // image has band_tiles = tiles_y, band_count = 1, band_index = 0.
    int band_tiles, band_index, band_count;
};

// Tile row of the image for the l-th tile row of a band launch.
__device__ __forceinline__ int band_global_tile_row(int l, int band_tiles, int band_index, int band_count) {
    const int b = l / band_tiles;
    return (b * band_count + band_index) * band_tiles + (l - b * band_tiles);
}

// Cooperative copy of the traversal data HBM -> LDS (field-major, see SceneLds).
__device__ __forceinline__ SceneLds stage_scene_in_lds(const float4* nodes, int nn, const float4* prims, int no,
                                                       const float* xforms, int nx, float4* lds) {
    float4* ln = lds;
    float4* lp = ln + kNodeFields * nn;
    float* lx = reinterpret_cast<float*>(lp + kPrimQuads * no);
    for (int i = threadIdx.x; i < nn * kNodeFields; i += kBlock) {
        int node = i / kNodeFields, f = i - node * kNodeFields;
        ln[f * nn + node] = nodes[node * kNodeStrideGlobal + f];
    }
    for (int i = threadIdx.x; i < no * kPrimQuads; i += kBlock) {
        int obj = i / kPrimQuads, q = i - obj * kPrimQuads;
        lp[q * no + obj] = prims[i];
    }
    for (int i = threadIdx.x; i < nx * kXformFloats; i += kBlock) lx[i] = xforms[i];
    SceneLds m;
    m.nodes = ln; m.prims = lp; m.xforms = lx; m.node_count = nn; m.object_count = no; m.root_entry = 0; m.order_flip = 0;
    return m;
}

// Flat mode: slot records (slot-major, last slot first) and the transforms of world-group primitives.
__device__ __forceinline__ FlatLds stage_flat_in_lds(const FlatView& F, const float* xforms, int nx, float4* lds) {
    const int n = F.slot_count;
    float* lx = reinterpret_cast<float*>(lds + kSlotQuads * n);
    for (int i = threadIdx.x; i < n * kSlotQuads; i += kBlock) {
        int slot = i / kSlotQuads, q = i - slot * kSlotQuads;
        lds[(n - 1 - slot) * kSlotQuads + q] = F.slots[i];     // candidate bit b of the box phase = slot n-1-b
    }
    for (int i = threadIdx.x; i < nx * kXformFloats; i += kBlock) lx[i] = xforms[i];
    FlatLds L;
    L.slots = lds; L.xforms = lx; L.slot_count = n;
    return L;
}

// Adds a finished path's contribution to the tile sums of its wave: a plain read-add-write, valid only
// while no other lane of the wave adds to the same pixel in the same instruction (render_fused's
// `accumulate` sees to that).  An LDS float atomic (ds_add_f32) takes about three cycles PER LANE on
// gfx950 whatever the addresses are (tools/ubench/lds_atomic.hip: 80 ns for a 64-lane instruction, the
// LDS pipe of the CU is busy all that time); the plain form is three full-rate instructions.
// Sums of a wave: [plane][2 open items][tile pixel]; `slot` = item buffer * TP + pixel.
// ATOMIC: the atomic form after all, zeros skipped (adding an exact zero cannot change a sum that starts at
// +0, so either form gives the same bits).  The flat kernel for basic scenes is bound by instruction issue
// at six waves per SIMD, not by the LDS pipe, and is 1 % faster with it (Cornell 7.87 against 7.96 ms); every
// other variant is faster with the plain form (orthographic 1.65 -> 1.49 ms, the 976-triangle scene 9.55 ->
// 9.10, materials 13.9 -> 13.4).
template <int TP, bool ATOMIC>
__device__ __forceinline__ void acc_add(float* acc, int plane, int slot, float v) {
    if (ATOMIC) { if (v != 0.0f) atomicAdd(&acc[plane * 2 * TP + slot], v); }
    else acc[plane * 2 * TP + slot] += v;
}
template <int TP, bool ATOMIC>
__device__ __forceinline__ void acc_add3(float* acc, int plane, int slot, V3 v) {
    acc_add<TP, ATOMIC>(acc, plane, slot, v.x); acc_add<TP, ATOMIC>(acc, plane + 1, slot, v.y); acc_add<TP, ATOMIC>(acc, plane + 2, slot, v.z);
}

// LDS carve-up of a block: [pass sums: waves x 2 x 14 x 32 floats][sampler scratch][stacks: STACK x 256 ints][scene copy]
constexpr int kCoopWordsPerWave = 192;          // coop_sphere_sample: published streams (128 words), then candidate points (192) over them
constexpr int acc_floats_per_wave(int mode) { return 2 * kPasses * kTileW * tile_rows(mode); }   // two work items in flight per wave
constexpr int kCamWords = 32;                    // DevCamera copy of the block
constexpr int kItemWords = 16;                   // per wave: the cold part of its work items (tile position, first sample, ...)
constexpr int kTagWords = 64;                    // per wave: which lane adds to a pixel of the two open items this round
constexpr int kParkWords = 4;                    // flat mode: per-lane path state kept in LDS (albedo to show, place in the tile sums)
static_assert(sizeof(DevCamera) <= kCamWords * 4, "camera copy does not fit");
constexpr int wave_lds_bytes(int mode) { return (acc_floats_per_wave(mode) + kCoopWordsPerWave + kItemWords + kTagWords) * 4 + kCamWords * 4 / kWaves; }

// One work item of a wave: the 8x4 pixel tile `tile`, samples [s0, s0 + items/32).
struct WorkItem {
    int next, total;
    bool valid;
    int tile, chunk, tx, ty, s0;      // (tree modes; the flat mode keeps these in LDS)
};

// The kernel's arguments, read from the kernel-argument segment where they are used.  A kernel that reads its
// parameter `A` directly has every word it will ever need loaded at its start; there are some 90 of them, the flat
// variants have 100 scalar registers, and the compiler parks the overflow in lanes of a vector register and
// fetches each word back with a v_readlane - a vector instruction, in kernels bound by vector issue.  Read through
// this pointer (opaque to the compiler, so that the loads stay where the use is and are not moved in front of the
// path loop) a value costs a scalar load from the constant cache and its register only while it is in use:
// Cornell 49 -> 19 spilled scalar registers, 7.28 -> 7.09 ms.  (Host pass: the parameter itself.)
#if defined(__HIP_DEVICE_COMPILE__)
typedef const RenderArgs __attribute__((address_space(4)))* KernelArgsPtr;
__device__ __forceinline__ KernelArgsPtr kernel_args_now() {
    const char __attribute__((address_space(4)))* base = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(base));
    return (KernelArgsPtr)base;
}
#define ARGS_NOW() (*kernel_args_now())
#else
#define ARGS_NOW() (A)
#endif

// The path tracer.  Persistent waves: the grid is sized to fill the chip once, and every wave
// takes work items (an 8x4 pixel tile x a chunk of samples) from a global counter until none are
// left.  A wave keeps TWO items open: lanes whose path ended take new paths from the `fill`
// item, while the last long paths of the previous item (`drain`) finish, so the wave stays
// full across item boundaries; an item's tile sums go to HBM once its last path has ended.
template <int MODE, int STACK, int GROUPS, int FAST, int FEAT = kFeatAll>
__global__ __launch_bounds__(kBlock, waves_per_simd(MODE, FEAT)) void render_fused(RenderArgs A) {
    constexpr int kTileH = tile_rows(MODE), kTilePixels = kTileW * kTileH;
    constexpr int kAccFloatsPerTile = kPasses * kTilePixels, kAccFloatsPerWave = 2 * kAccFloatsPerTile;
    extern __shared__ float4 smem[];
    float* acc_all = reinterpret_cast<float*>(smem);
    uint32_t* coop_all = reinterpret_cast<uint32_t*>(acc_all + kWaves * kAccFloatsPerWave);
    // The camera is read from LDS where a new path starts: as kernel arguments its 25 values would
    // sit in scalar registers for the whole kernel, and the compiler, out of scalar registers,
    // parks them in lanes of a vector register and fetches each one back with a v_readlane.
    DevCamera* lds_cam = reinterpret_cast<DevCamera*>(coop_all + kWaves * kCoopWordsPerWave);
    // Flat mode: work-item values that are the same for every lane and only read where a path starts or an
    // item is flushed live here, one copy per wave, instead of in a vector register per value for the whole
    // kernel (the compiler does not keep them in scalar registers, and at 80 registers they end up in scratch
    // memory): {tx, ty, first sample, -, tile, chunk, tile and chunk of the draining item}.  The tree modes
    // have the registers (cap 128) and are faster with them.
    constexpr bool kItemInLds = MODE == kModeFlat;
    typedef volatile int __attribute__((address_space(3))) * LdsVolatileIntPtr;
    // (the wave's number, said to be the same in every lane: what is derived from it - the wave's parts of the LDS - then sits in
    // scalar registers instead of vector registers that are live, or spilled to scratch memory, through the whole kernel)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    LdsVolatileIntPtr item_mem = (LdsVolatileIntPtr)(reinterpret_cast<int*>(coop_all + kWaves * kCoopWordsPerWave + kCamWords) + wave * kItemWords);
    static_assert(2 * kTilePixels <= kTagWords, "one tag word per pixel of the two open items");
    int* tag = reinterpret_cast<int*>(coop_all + kWaves * kCoopWordsPerWave + kCamWords + kWaves * kItemWords) + wave * kTagWords;
    int* stack_mem = reinterpret_cast<int*>(coop_all + kWaves * kCoopWordsPerWave + kCamWords + kWaves * (kItemWords + kTagWords));
    float* entry_mem = reinterpret_cast<float*>(stack_mem + STACK * kBlock);      // tree modes: 4 slot codes + 4 box entry distances per lane
    // Flat mode: the albedo a path shows in the albedo pass is found at its first hit and only added when
    // the path ends (an exhausted path shows black, render.cuh:96-97); it waits in LDS, three words per
    // lane, instead of three registers that would be live through every traversal in between.
    // (the lane's address is formed anew at each use, one instruction: computed once, it is a register that is live
    // through the whole kernel - or a spill slot in scratch memory that is loaded back at every use)
    auto park_of_lane = [&]() {
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        return entry_mem + t;
    };
    float4* lds_scene = reinterpret_cast<float4*>(entry_mem + (MODE == kModeFlat ? kParkWords * kBlock : 8 * kBlock));
    const DevScene& sc = A.sc;
    if (threadIdx.x == 0) *lds_cam = sc.cam;

    const int lane = threadIdx.x & 63;
    float* acc = acc_all + wave * kAccFloatsPerWave;              // [2][14][tile pixels] sums of the wave's two open tiles
    uint32_t* coop_pub = coop_all + wave * kCoopWordsPerWave;
    // (every lane has read the published streams before any lane stores a candidate point: one wave, LDS in order)
    float* coop_res = reinterpret_cast<float*>(coop_pub);
#pragma unroll
    for (int k = 0; k < kAccFloatsPerWave / 64; k++) acc[k * 64 + lane] = 0.0f;

    SceneGlobal SG;
    SG.nodes = sc.nodes; SG.prims = sc.prims; SG.xforms = sc.xforms; SG.root_entry = sc.root_entry; SG.order_flip = sc.order_flip;
    SceneLds SL = {};
    if constexpr (MODE == kModeTreeLds) { SL = stage_scene_in_lds(sc.nodes, sc.node_count, sc.prims, sc.object_count, sc.xforms, sc.xform_count, lds_scene); SL.root_entry = sc.root_entry; SL.order_flip = sc.order_flip; }
    FlatLds FL = {};
    if constexpr (MODE == kModeFlat) FL = stage_flat_in_lds(A.flat, sc.xforms, sc.xform_count, lds_scene);
    __syncthreads();

    LaneStack stk;
    stk.attach(stack_mem, STACK, kBlock, threadIdx.x);
    stk.entry = (LdsFloatPtr)(entry_mem + threadIdx.x);
    stk.code = (LdsIntPtr)(reinterpret_cast<int*>(entry_mem) + 4 * kBlock + threadIdx.x);
    stk.spill = (GlobalIntPtr)(A.stack_spill + (size_t)blockIdx.x * kBlock + threadIdx.x);
    stk.spill_stride = (size_t)A.spill_threads;

    // next work item of this wave (wave-uniform); invalid once the counter has run past the end
    // The work items come from kQueues counters, counter k handing out the items i with i % kQueues == k: additions to
    // ONE address from the whole chip go through one after the other, about 85 million a second, and a frame of
    // many short items (a small scene in front of a black background) asks for more than that.  A wave starts at
    // the counter of its block's number (blocks go round the chip's eight dies) and, once a counter has run out,
    // goes on with the next one for good; when all have run out the queue is empty.
    // (which counter, and how many have run out: flat mode keeps the two in LDS with the rest of the item)
    unsigned int queue_at = blockIdx.x % kQueues, queues_out = 0;
    unsigned int fill_opened = 0, drain_opened = 0;               // (tree modes) when the two open items were taken from the queue
    if constexpr (kItemInLds) { if (lane == 0) { item_mem[8] = (int)queue_at; item_mem[9] = 0; } }
    auto fetch = [&]() {
        const auto& P = ARGS_NOW();
        const int item_count = P.ntiles * P.chunks;
        WorkItem w = {};
        unsigned int at = kItemInLds ? (unsigned int)item_mem[8] : queue_at;
        unsigned int out = kItemInLds ? (unsigned int)item_mem[9] : queues_out;
        at = __builtin_amdgcn_readfirstlane(at);
        out = __builtin_amdgcn_readfirstlane(out);
        unsigned int q = 0xffffffffu;
        while (out < (unsigned int)kQueues) {
            unsigned int n = 0;
            if (lane == 0) n = atomicAdd(P.queue + at * kQueueStride, 1u);
            n = __builtin_amdgcn_readfirstlane(n);
            // (n never comes near 2^32 / kQueues: a counter stops rising once every wave has seen it run out)
            if (n * kQueues + at < (unsigned int)item_count) { q = n * kQueues + at; break; }
            out++;
            at = (at + 1) % kQueues;
        }
        if constexpr (kItemInLds) { if (lane == 0) { item_mem[8] = (int)at; item_mem[9] = (int)out; } }
        else { queue_at = at; queues_out = out; }
        w.valid = q < (unsigned int)item_count;
        if (P.order_mode == 2 && w.valid) {
            unsigned int v = 0;
            if (lane == 0) v = P.queue[kItemOrderFromQueue + q];
            q = __builtin_amdgcn_readfirstlane(v);
        }
        const unsigned int now = (unsigned int)wall_clock64();
        if constexpr (kItemInLds) { if (lane == 0) item_mem[10] = (int)now; }
        else fill_opened = now;
        const int chunk = (int)(q / (unsigned int)P.ntiles);
        const int tile = (int)q - chunk * P.ntiles;
        const int s0 = P.first_sample + chunk * P.chunk_size;
        const int s1 = min(s0 + P.chunk_size, P.first_sample + P.samples);
        const int tx = tile % P.tiles_x, ty = band_global_tile_row(tile / P.tiles_x, P.band_tiles, P.band_index, P.band_count);
        if constexpr (kItemInLds) {
            if (lane == 0) {
                item_mem[0] = tx;
                item_mem[1] = ty;
                item_mem[2] = s0;
                item_mem[4] = tile;
                item_mem[5] = chunk;
            }
        } else {
            w.tx = tx; w.ty = ty; w.s0 = s0; w.tile = tile; w.chunk = chunk;
        }
        w.total = w.valid ? (s1 - s0) * kTilePixels : 0;
        w.next = 0;
        return w;
    };
    // tile sums of one finished item -> HBM (lane = pixel of the tile), buffer cleared for the next item
    auto flush = [&](int buffer, int tile, int chunk, unsigned int opened) {
        const auto& P = ARGS_NOW();
        if (P.order_mode != 0 && lane == 0) {
            const unsigned int ticks = (unsigned int)wall_clock64() - opened;
            P.queue[kItemCostFromQueue + chunk * P.ntiles + tile] = ticks;
        }
        float* sums = acc + buffer * kTilePixels;
        float* out = P.partial + (size_t)chunk * kPasses * P.pix_padded + (size_t)tile * kTilePixels + lane;
        const size_t ps = P.pix_padded;
        if (lane < kTilePixels) {
#pragma unroll
            for (int k = 0; k < kPasses; k++) {
                out[k * ps] = sums[k * 2 * kTilePixels + lane];
                sums[k * 2 * kTilePixels + lane] = 0.0f;
            }
        }
    };

    WorkItem fill = fetch();                                      // wave-uniform state
    int fill_buffer = 0;
    bool draining = false;                                        // the other buffer holds an item whose paths are still running
    int drain_tile = 0, drain_chunk = 0;                          // (tree modes)

    unsigned int rays = 0;
    bool alive = false;
    // item buffer * tile pixels + pixel of the tile: where this lane's path adds its results.  Set where a path starts and read where
    // it adds; the flat variants that are short of registers keep it in LDS beside the pending albedo (the compiler
    // otherwise spills it to scratch memory: a store per path and a load per addition).
    constexpr bool kPixoffInLds = MODE == kModeFlat && FEAT != kFeatBasicXform;
    int pixoff_reg = 0;
    auto set_pixoff = [&](int v) {
        if constexpr (kPixoffInLds) reinterpret_cast<int*>(park_of_lane())[3 * kBlock] = v;
        else pixoff_reg = v;
    };
    auto get_pixoff = [&]() {
        if constexpr (kPixoffInLds) return reinterpret_cast<const int*>(park_of_lane())[3 * kBlock];
        else return pixoff_reg;
    };
    Ray ray;
    ray.o = v3(0, 0, 0); ray.d = v3(0, 0, 1); ray.time = 0.0f;
    V3 throughput = v3(1, 1, 1), pend_albedo = v3(0, 0, 0);       // (pend_albedo: tree modes; flat mode keeps it in `park`)
    auto set_pending_albedo = [&](V3 a) {
        if constexpr (MODE == kModeFlat) { float* park = park_of_lane(); park[0] = a.x; park[kBlock] = a.y; park[2 * kBlock] = a.z; }
        else pend_albedo = a;
    };
    auto pending_albedo = [&]() {
        if constexpr (MODE == kModeFlat) { const float* park = park_of_lane(); return v3(park[0], park[kBlock], park[2 * kBlock]); }
        else return pend_albedo;
    };
    int bounces = 0;
    Rng rng;
    rng.pos = 0;
    TreeState ts;                                                 // tree modes: the lane's walk in progress
    ts.begin(INFINITY, sc.root_entry);
    bool walking = false;

    // Runs `adds` (plain read-add-writes to this lane's pixel) for every lane that calls, lanes that share a
    // pixel one after the other: each lane writes its number to the pixel's tag word, the lane whose number
    // stands there adds, the others try again.  One round unless two samples of a pixel finish in the same
    // instruction.  (LDS executes a wave's instructions in order; the fences only keep the compiler from moving
    // or caching the accesses.)
    constexpr bool kAtomicSums = MODE == kModeFlat && FEAT == kFeatBasicXform;
    SectionClock clk;
    // (profiling build, flat mode: the time in the additions to the tile sums is booked under the tree walk's section, which the flat mode does not use)
    auto accumulate = [&](auto&& adds) {
        [[maybe_unused]] float dep = 0.0f;
        if constexpr (MODE == kModeFlat) CRT_CLOCK(clk, kSecNode, dep);
        if constexpr (kAtomicSums) {
            adds();
        } else {
            bool todo = true;
            const int pixoff = get_pixoff();
            while (true) {
                if (todo) tag[pixoff] = lane;
                wave_lds_sync();
                const bool win = todo && tag[pixoff] == lane;
                if (win) { adds(); todo = false; }
                wave_lds_sync();
                if (!lanes_where(todo)) break;
            }
        }
        if constexpr (MODE == kModeFlat) CRT_CLOCK(clk, kSecScatter, dep);
    };

    // Flat mode: the lanes at their first hit are exactly the ones that took a new path at the top of this
    // iteration, consecutive samples of one item; no two of them share a pixel unless more than a tile's
    // worth of paths started at once.  Then they add without asking.
    bool first_hits_unique = false;
    auto accumulate_first = [&](auto&& adds) {
        if (kAtomicSums || (MODE == kModeFlat && first_hits_unique)) {
            [[maybe_unused]] float dep = 0.0f;
            if constexpr (MODE == kModeFlat) CRT_CLOCK(clk, kSecNode, dep);
            adds();
            if constexpr (MODE == kModeFlat) CRT_CLOCK(clk, kSecScatter, dep);
        } else {
            accumulate(adds);
        }
    };

    clk.start();
    while (true) {
        CRT_PROF(kProfOuter);
        CRT_CLOCK(clk, kSecRegen, ray.d.x);
        if constexpr (MODE == kModeFlat) {
            // the same in every lane: said so, the tests below are scalar compares and branches
            // (Cornell -1.6 %; the tree modes are 1.3 % slower with it)
            fill.next = __builtin_amdgcn_readfirstlane(fill.next);
            fill.total = __builtin_amdgcn_readfirstlane(fill.total);
            fill.valid = __builtin_amdgcn_readfirstlane((int)fill.valid) != 0;
            fill_buffer = __builtin_amdgcn_readfirstlane(fill_buffer);
            draining = __builtin_amdgcn_readfirstlane((int)draining) != 0;
        }
        // the drained item is complete when none of its paths is alive any more
        if (draining) {
            const bool mine = alive && ((get_pixoff() >= kTilePixels) == ((fill_buffer ^ 1) != 0));
            if (!lanes_where(mine)) {
                if constexpr (kItemInLds) flush(fill_buffer ^ 1, item_mem[6], item_mem[7], (unsigned int)item_mem[11]);
                else flush(fill_buffer ^ 1, drain_tile, drain_chunk, drain_opened);
                draining = false;
            }
        }
        // the fill item has handed out all its paths: let it drain and open the next one
        if (fill.valid && fill.next >= fill.total && !draining) {
            draining = true;
            if constexpr (kItemInLds) {
                if (lane == 0) { item_mem[6] = item_mem[4]; item_mem[7] = item_mem[5]; item_mem[11] = item_mem[10]; }
            } else {
                drain_tile = fill.tile; drain_chunk = fill.chunk; drain_opened = fill_opened;
            }
            fill_buffer ^= 1;
            fill = fetch();
        }
        if constexpr (MODE == kModeFlat && !kAtomicSums) first_hits_unique = true;    // (no new paths this iteration: nobody is at a first hit)
        if (fill.next < fill.total) {                             // wave-uniform
            const unsigned long long free_mask = lanes_where(!alive);
            if constexpr (MODE == kModeFlat && !kAtomicSums) first_hits_unique = __builtin_popcountll(free_mask) <= kTilePixels;
            if (!alive) {
                CRT_PROF(kProfRegen);
                const int item = fill.next + __builtin_popcountll(free_mask & ((1ull << lane) - 1ull));
                if (item < fill.total) {
                    const int pix = item & (kTilePixels - 1);
                    const int s0 = kItemInLds ? item_mem[2] : fill.s0, tx = kItemInLds ? item_mem[0] : fill.tx, ty = kItemInLds ? item_mem[1] : fill.ty;
                    const int sample = s0 + item / kTilePixels + 1;       // sample numbers are 1-based (render.cuh:128)
                    const int pi = tx * kTileW + (pix & 7), pj = ty * kTileH + (pix >> 3);
                    const auto& sc = ARGS_NOW().sc;
                    if (pi < sc.width && pj < sc.height && sc.max_bounces > 0) {
                        set_pixoff(fill_buffer * kTilePixels + pix);
                        rng.init(pj * sc.width + pi, sample);
                        ray = camera_ray<FEAT>(*lds_cam, pi, pj, rng);
                        throughput = v3(1, 1, 1);
                        if constexpr (MODE != kModeFlat) pend_albedo = v3(0, 0, 0);   // (first read after a bounce-0 hit has set it)
                        bounces = 0;
                        alive = true;
                        if constexpr (MODE != kModeFlat) { ts.begin(INFINITY, sc.root_entry); walking = true; }
                    }
                }
            }
            fill.next += __builtin_popcountll(free_mask);
        }
        CRT_CLOCK(clk, kSecOther, ray.o.x);
        // (The rest of the round under an `if`, not skipped with a `continue`: with two ways back to the top of the loop the
        // compiler kept the path state in one set of registers up to here and in another from here on and copied some twenty
        // values from one to the other and back in every round - Cornell 7.10 -> 6.91 ms, the normal-map scene -5 %.)
        const bool any_alive = lanes_where(alive) != 0ull;
        if (!any_alive && !fill.valid && !draining) break;        // queue empty and everything flushed
        if (any_alive) {
        Hit h;
        h.t = 0.0f; h.obj = -1; h.bu = 0.0f; h.bv = 0.0f;
        Surface sf = {};
        float4 m0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        int mat_type = -1;
        // closest hit (render.cuh:47).  Flat mode: every living lane, in one go.  Tree modes: the
        // walks run until enough of them have finished; only those lanes are shaded this time.
        bool ready = alive;
        if constexpr (MODE != kModeFlat) {
            CRT_CLOCK(clk, kSecNode, ray.o.y);
            const int living = __builtin_popcountll(lanes_where(alive));
            const int ready_target = max(1, living * A.ready_percent / 100);
            if constexpr (MODE == kModeTreeLds) tree_run<FAST, FEAT>(SL, ray, 0.001f, ts, walking, stk, living - ready_target + 1, A.prim_batch);
            else tree_run<FAST, FEAT>(SG, ray, 0.001f, ts, walking, stk, living - ready_target + 1, A.prim_batch);
            ready = alive && !walking;
        }
        rays += (unsigned int)__builtin_popcountll(lanes_where(ready));   // one count per wave, in a scalar register
        if (ready) {
            CRT_PROF(kProfTrace);
            if constexpr (MODE == kModeFlat) {
                CRT_CLOCK(clk, kSecBox, ray.o.y);
                {
                    const FlatView view = ARGS_NOW().flat;
                    h = traverse_flat<GROUPS, FAST, FEAT>(FL, view, ray, 0.001f, INFINITY, clk);
                    // (rank in the reference's visiting order -> object index)
                    if (h.obj >= 0) h.obj = *at32(view.ent_obj, h.obj);
                }
            } else {
                h = ts.result();
            }
            CRT_CLOCK(clk, kSecSurface, h.t);
            if (h.obj >= 0) {
                CRT_PROF(kProfShade);
                const DevScene scene = ARGS_NOW().sc;
                sf = make_surface<false, FEAT, FAST>(scene, ray, h);
                m0 = *at32(scene.mats, sf.mat * 2);
                mat_type = material_kind(m0);
            }
        }
        // the sphere sample of a diffuse or metal bounce, drawn by the whole wave together
        CRT_CLOCK(clk, kSecSample, sf.p.x);
        const V3 rv = coop_sphere_sample<FAST, (FEAT & kFeatMaterials) != 0>(mat_type == 0 || ((FEAT & kFeatMaterials) && mat_type == 1), (FEAT & kFeatMaterials) && mat_type == 1, rng, coop_pub, coop_res, lane);
        V3 rv_dep = rv;
        CRT_CLOCK(clk, kSecScatter, rv_dep.x);
        if (ready) {
            const DevScene sc = ARGS_NOW().sc;
            if (h.obj >= 0) {
                V3 attenuation;
                Ray next;
                // what the normal and depth passes show of a first hit, whichever way the path goes on (render.cuh:55-56, 69-70)
                V3 n = v3(0.0f, 0.0f, 0.0f);
                float depth = 0.0f;
                if (bounces == 0) {
                    n = sf.unit_normal;
                    if (!sf.has_unit) n = unit_vector<FAST>(sf.normal);
                    depth = h.t * length(ray.d);
                }
                if (scatter_with<FEAT, FAST>(sc, ray, sf, m0, rv_dep, rng, attenuation, next)) {  // render.cuh:52-61
                    if (bounces == 0) {
                        set_pending_albedo(attenuation);
                        accumulate_first([&]() {
                            const int pixoff = get_pixoff();
                            acc_add3<kTilePixels, kAtomicSums>(acc, 9, pixoff, n);
                            acc_add<kTilePixels, kAtomicSums>(acc, 12, pixoff, depth);
                            acc_add<kTilePixels, kAtomicSums>(acc, 13, pixoff, 1.0f);
                        });
                    }
                    throughput = throughput * attenuation;
                    ray = next;
                    bounces++;
                    if (bounces >= sc.max_bounces) alive = false;   // exhausted: col = albedo = 0 (render.cuh:96-97)
                    else if constexpr (MODE != kModeFlat) { ts.begin(INFINITY, sc.root_entry); walking = true; }
                } else {                                             // render.cuh:64-74
                    V3 emitted = emit<FEAT>(sc, sf, m0);
                    V3 c = throughput * emitted;
                    V3 alb = emitted;
                    if (bounces != 0) alb = pending_albedo();
                    accumulate([&]() {
                        const int pixoff = get_pixoff();
                        acc_add3<kTilePixels, kAtomicSums>(acc, 0, pixoff, c);
                        if (bounces == 0) {
                            acc_add3<kTilePixels, kAtomicSums>(acc, 6, pixoff, emitted);
                            acc_add3<kTilePixels, kAtomicSums>(acc, 9, pixoff, n);
                            acc_add<kTilePixels, kAtomicSums>(acc, 12, pixoff, depth);
                            acc_add<kTilePixels, kAtomicSums>(acc, 13, pixoff, 1.0f);
                        }
                        acc_add3<kTilePixels, kAtomicSums>(acc, 3, pixoff, alb);
                    });
                    alive = false;
                }
            } else {                                                 // render.cuh:76-91
                V3 c = throughput * environment<FEAT, FAST>(sc, ray.d);
                V3 alb = c;
                if (bounces != 0) alb = pending_albedo();
                accumulate([&]() {
                    const int pixoff = get_pixoff();
                    acc_add3<kTilePixels, kAtomicSums>(acc, 0, pixoff, c);
                    if (bounces == 0) acc_add<kTilePixels, kAtomicSums>(acc, 12, pixoff, __builtin_nanf(""));   // depth of a miss is NaN (render.cuh:88)
                    acc_add3<kTilePixels, kAtomicSums>(acc, 3, pixoff, alb);
                });
                alive = false;
            }
        }
        CRT_CLOCK(clk, kSecOther, throughput.x);
        }
    }
    clk.flush();

    // one ray-count atomic per wave
    if (lane == 0) atomicAdd(A.ray_counter, (unsigned long long)rays);
}

// The launch order of the next frame: items by the class of their measured cost, most expensive class first
// (within a class in no particular order).  The last items a launch hands out are then its cheapest, and the
// waves run dry within the time one cheap item takes instead of one expensive one.  `classes` = items per class
// (count_cost_classes), `placed` = zero on entry.  One thread per item, 4 items per thread.
constexpr int kOrderItemsPerThread = 4;
// (The classes are counted here, one addition per class and block, not by the render kernel item by item: the
// items of a plain scene fall into a handful of classes, and a hundred thousand additions to a handful of
// addresses take longer than the frame - the orthographic cube went from 1.3 to 1.7 ms.)
__global__ __launch_bounds__(kBlock) void count_cost_classes(const unsigned int* cost, int n, unsigned int* classes) {
    __shared__ unsigned int here[kCostClasses];
    for (int k = threadIdx.x; k < kCostClasses; k += kBlock) here[k] = 0;
    __syncthreads();
    const int base = (blockIdx.x * kBlock + threadIdx.x) * kOrderItemsPerThread;
#pragma unroll
    for (int k = 0; k < kOrderItemsPerThread; k++)
        if (base + k < n) atomicAdd(&here[cost_class(cost[base + k])], 1u);
    __syncthreads();
    for (int k = threadIdx.x; k < kCostClasses; k += kBlock)
        if (here[k]) atomicAdd(&classes[k], here[k]);
}
__global__ __launch_bounds__(kBlock) void order_by_cost(const unsigned int* cost, int n, const unsigned int* classes, unsigned int* placed, unsigned int* order) {
    __shared__ unsigned int here[kCostClasses], first[kCostClasses];
    for (int k = threadIdx.x; k < kCostClasses; k += kBlock) here[k] = 0;
    __syncthreads();
    int cls[kOrderItemsPerThread];
    unsigned int rank[kOrderItemsPerThread];
    const int base = (blockIdx.x * kBlock + threadIdx.x) * kOrderItemsPerThread;
#pragma unroll
    for (int k = 0; k < kOrderItemsPerThread; k++) {
        cls[k] = -1; rank[k] = 0;
        if (base + k < n) {
            cls[k] = cost_class(cost[base + k]);
            rank[k] = atomicAdd(&here[cls[k]], 1u);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < kCostClasses; k += kBlock) {
        unsigned int before = 0;                       // items of all more expensive classes
        for (int j = k + 1; j < kCostClasses; j++) before += classes[j];
        first[k] = before + (here[k] ? atomicAdd(&placed[k], here[k]) : 0u);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kOrderItemsPerThread; k++) {
        if (cls[k] < 0) continue;
        const unsigned int at = first[cls[k]] + rank[k];
        if (at < (unsigned int)n) order[at] = (unsigned int)(base + k);
    }
}

struct ResolveArgs {
    const float* partial;
    float *color, *albedo, *emission, *normal, *depth, *opacity;
    int width, height, tiles_x, tile_rows, ntiles, pix_padded, chunks;
    float done, count;     // samples already in the buffers / samples summed now
    int fold;              // 0: overwrite with sum/count, 1: (old*done + sum)/(done+count)
    int band_tiles, band_index, band_count;   // as in RenderArgs; the pass arrays hold the band's rows only
};

__global__ __launch_bounds__(kBlock) void resolve_passes(ResolveArgs R) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= R.pix_padded) return;
    const int tile_pixels = kTileW * R.tile_rows;
    const int tile = t / tile_pixels, lane = t - tile * tile_pixels;
    const int local_row = tile / R.tiles_x;
    const int pi = (tile % R.tiles_x) * kTileW + (lane & 7);
    const int pj = band_global_tile_row(local_row, R.band_tiles, R.band_index, R.band_count) * R.tile_rows + (lane >> 3);
    if (pi >= R.width || pj >= R.height) return;
    const int pixel = (local_row * R.tile_rows + (lane >> 3)) * R.width + pi;    // row within the band's own buffer
    float v[kPasses];
#pragma unroll
    for (int k = 0; k < kPasses; k++) {
        float acc = R.partial[(size_t)k * R.pix_padded + t];
        for (int c = 1; c < R.chunks; c++) acc += R.partial[((size_t)c * kPasses + k) * R.pix_padded + t];
        v[k] = acc;
    }
    float* dst[kPasses] = {R.color + 3 * pixel, R.color + 3 * pixel + 1, R.color + 3 * pixel + 2,
                           R.albedo + 3 * pixel, R.albedo + 3 * pixel + 1, R.albedo + 3 * pixel + 2,
                           R.emission + 3 * pixel, R.emission + 3 * pixel + 1, R.emission + 3 * pixel + 2,
                           R.normal + 3 * pixel, R.normal + 3 * pixel + 1, R.normal + 3 * pixel + 2,
                           R.depth + pixel, R.opacity + pixel};
    const float total = R.done + R.count;
#pragma unroll
    for (int k = 0; k < kPasses; k++) {
        float x = R.fold ? (*dst[k] * R.done + v[k]) / total : v[k] / R.count;
        *dst[k] = x;
    }
}

// ===================================================================== wavefront path ====
// The path loop cut into kernels per bounce with the path state in HBM (crt_scene_options.wavefront):
//   wf_generate    camera rays of a batch of samples -> queue of bounce 0
//   wf_trace       closest hits of the rays of the current queue.  Persistent waves; a lane whose
//                  walk has ended writes its hit and takes the NEXT RAY of the queue at once
//                  (in the fused kernel it waits until enough other lanes have finished, because
//                  its next ray only exists after shading); carrying walk state only, the kernel
//                  needs 75 registers and fits six waves per SIMD
//   wf_shade       surface, scatter / emit / environment for every hit of the queue (every lane
//                  busy); surviving paths move to the next queue; results go to per-path records
//                  keyed by (sample, pixel)
//   wf_accumulate  adds the batch's per-path records, in sample order, to the pixel sums that
//                  resolve_passes turns into the pass buffers.
// A queue is cut into SEGMENTS of kWfSegment slots.  Segment k of bounce b+1 receives the survivors
// of segment k of bounce b, packed at its front, and a count per segment says how many there are:
// compaction needs no global atomic at all (one workgroup owns a segment and counts in LDS; a
// single device-wide queue tail would take one same-address atomic per wave, which serialises at
// 12-50 ns each - 12 ms for the first bounce of a 4K frame), and every access of trace and shade
// is a dense float4 per lane (the state moves with the path from queue to queue).  Same device
// functions, same bits as the fused kernel; sums are added in sample order, so a bulk render is
// reproducible bit for bit.
//
// Measured (MI355X): the walk's loops do run fuller (1 M triangles: node steps at 79 % of the lanes
// against 43 % in the fused kernel, 18 instead of 34 wave-level node steps per 64 rays), but the
// frame is not faster: 44.6 ms against 37.2 ms for the million-triangle scene at 8 spp (trace
// 34.7, shade 8.1, generate + accumulate 1.8), 26 ms against 12.4 ms for the 976-triangle mesh
// scene.  Both pipelines fetch the same 28-30 GB per frame into L2 (about 210 bytes per ray: nodes
// and triangles, 190 MB, do not stay in the 4 MB of an XCD's L2), and with a tenth of all node and
// triangle reads missing L2 nearly every wave-level step waits for one of its lanes' misses; four
// or six waves per SIMD made no difference to the trace kernel.  The fused kernel stays the
// default for every scene; this pipeline is kept for what it adds: bulk renders that are
// reproducible bit for bit with per-path records instead of per-sample chunk sums.
struct WfQueue {
    float4 *ray0, *ray1;     // (o.xyz, d.x)  (d.y, d.z, time, -)
    float4 *st0, *st1;       // (throughput.xyz, rng.pos)  (pending albedo.xyz, -)
    uint2* meta;             // (path id within the batch, bounces so far)
    unsigned int* count;     // [segments] live slots at the front of each segment
};
struct WfArgs {
    DevScene sc;
    WfQueue cur, next;
    float4* hit;                     // [slot of cur] (t, object, bu, bv)
    float4* res;                     // [4][batch_capacity] by path id: (color|depth) (albedo|opacity) (emission|0) (normal|0)
    unsigned int* cursors;           // [max_bounces + 1][kWfCursors] (padded) next entry of each segment list of a bounce
    unsigned long long* ray_counter;
    unsigned long long* block_rays;  // [trace grid] rays taken per workgroup of wf_trace, summed by wf_sum_rays
    float* partial;                  // [14][pix_padded] pixel sums of the launch
    int bounce;
    int segments;                    // segments of this batch's queues
    int batch_samples, batch_capacity;   // samples in this batch; stride of the res planes
    int first_sample;                // sample number of the batch's first sample is first_sample + 1
    int pix_padded, tiles_x, band_tiles, band_index, band_count;
    int* stack_spill;
    int spill_threads;
    int prim_batch;
    int refill_lanes;                // wf_trace: hand out new rays once this many lanes of a wave are idle
    int accumulate_first;            // wf_accumulate: 1 = overwrite the sums (first batch of a launch)
};

constexpr int kWfTileH = 4;          // 8x4-pixel tiles like the tree modes of the fused kernel
constexpr int kWfTilePixels = kTileW * kWfTileH;
constexpr int kWfSegment = 1024;     // slots per queue segment (a multiple of the workgroup size)
// Trace waves take segments in ascending order, so that at any moment the whole chip works on one
// narrow band of the image (path ids run tile by tile with all samples of a tile together) and the
// part of the hierarchy in use stays in L2.  One cursor for all waves would be one same-address
// atomic per segment, about 50 ns each; instead the segments are dealt round-robin to
// kWfCursors lists with a cursor each (one cache line apart), a wave starts on the list of its own
// number and moves on to the next list when one runs out.
constexpr int kWfCursors = 64;
constexpr int kWfCursorStride = 32;  // unsigned ints between cursors: 128 bytes
#ifndef CRT_WF_TRACE_WAVES_PER_SIMD
#define CRT_WF_TRACE_WAVES_PER_SIMD 6
#endif

// Position in the segment's output for the lanes of this wave that keep a path (`keep`): the
// workgroup's running count lives in LDS.
__device__ __forceinline__ unsigned int wf_segment_slot(bool keep, unsigned int* lds_count, int lane) {
    const unsigned long long mask = lanes_where(keep);
    unsigned int off = 0;
    if (lane == 0 && mask) off = atomicAdd(lds_count, (unsigned int)__builtin_popcountll(mask));
    return __builtin_amdgcn_readfirstlane(off) + (unsigned int)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
}

__global__ __launch_bounds__(kBlock) void wf_generate(WfArgs A) {
    __shared__ unsigned int out_count;
    const DevScene& sc = A.sc;
    const int lane = threadIdx.x & 63;
    const long long total = (long long)A.batch_samples * A.pix_padded;
    for (int seg = blockIdx.x; seg < A.segments; seg += gridDim.x) {
        __syncthreads();
        if (threadIdx.x == 0) out_count = 0;
        __syncthreads();
        const size_t base = (size_t)seg * kWfSegment;
        for (int i = threadIdx.x; i < kWfSegment; i += kBlock) {
            const long long p = (long long)base + i;
            bool live = false;
            Ray ray;
            ray.o = v3(0, 0, 0); ray.d = v3(0, 0, 1); ray.time = 0.0f;
            Rng rng;
            rng.pos = 0;
            if (p < total && sc.max_bounces > 0) {
                // path id = (tile * samples + sample) * 32 + pixel of the tile
                const int pix = (int)(p % kWfTilePixels);
                const long long ts = p / kWfTilePixels;
                const int s = (int)(ts % A.batch_samples), tile = (int)(ts / A.batch_samples);
                const int pi = (tile % A.tiles_x) * kTileW + (pix & 7);
                const int pj = band_global_tile_row(tile / A.tiles_x, A.band_tiles, A.band_index, A.band_count) * kWfTileH + (pix >> 3);
                if (pi < sc.width && pj < sc.height) {
                    rng.init(pj * sc.width + pi, A.first_sample + s + 1);      // sample numbers are 1-based (render.cuh:128)
                    ray = camera_ray(sc.cam, pi, pj, rng);
                    live = true;
                }
            }
            const size_t slot = base + wf_segment_slot(live, &out_count, lane);
            if (live) {
                A.cur.ray0[slot] = make_float4(ray.o.x, ray.o.y, ray.o.z, ray.d.x);
                A.cur.ray1[slot] = make_float4(ray.d.y, ray.d.z, ray.time, 0.0f);
                A.cur.st0[slot] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(rng.pos));
                A.cur.st1[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                A.cur.meta[slot] = make_uint2((unsigned int)p, 0u);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) A.cur.count[seg] = out_count;
    }
}

template <int STACK, int FAST>
__global__ __launch_bounds__(kBlock, CRT_WF_TRACE_WAVES_PER_SIMD) void wf_trace(WfArgs A) {
    extern __shared__ float4 smem[];
    int* stack_mem = reinterpret_cast<int*>(smem);
    const DevScene& sc = A.sc;
    const int lane = threadIdx.x & 63;
    SceneGlobal S;
    S.nodes = sc.nodes; S.prims = sc.prims; S.xforms = sc.xforms; S.root_entry = sc.root_entry; S.order_flip = sc.order_flip;
    LaneStack stk;
    stk.attach(stack_mem, STACK, kBlock, threadIdx.x);
    stk.entry = (LdsFloatPtr)(reinterpret_cast<float*>(stack_mem + STACK * kBlock) + threadIdx.x);
    stk.code = (LdsIntPtr)(stack_mem + (STACK + 4) * kBlock + threadIdx.x);
    stk.spill = (GlobalIntPtr)(A.stack_spill + (size_t)blockIdx.x * kBlock + threadIdx.x);
    stk.spill_stride = (size_t)A.spill_threads;

    // A wave takes one segment of the queue at a time (one atomic per segment) and hands its rays
    // to the lanes as their walks end.
    size_t have_lo = 0, have_hi = 0;                 // wave-uniform: slots [have_lo, have_hi) still to hand out
    bool dry = false;                                // no segment left
    int list = (int)((blockIdx.x * kWaves + (threadIdx.x >> 6)) % kWfCursors);   // wave-uniform: the segment list in use
    unsigned int* const cursors = A.cursors + (size_t)A.bounce * kWfCursors * kWfCursorStride;
    unsigned long long rays = 0;                     // wave-uniform: rays this wave took
    Ray ray;
    ray.o = v3(0, 0, 0); ray.d = v3(0, 0, 1); ray.time = 0.0f;
    TreeState ts;
    ts.begin(INFINITY, sc.root_entry);
    bool walking = false;
    size_t slot = 0;
    while (true) {
        // hand rays to the lanes without one
        while (!dry) {
            const unsigned long long idle = lanes_where(!walking);
            if (!idle) break;
            if (have_lo == have_hi) {
                unsigned int k = 0;
                if (lane == 0) k = atomicAdd(&cursors[list * kWfCursorStride], 1u);
                k = __builtin_amdgcn_readfirstlane(k) * kWfCursors + (unsigned int)list;
                if (k >= (unsigned int)A.segments) {
                    // this list has run out: look at all cursors at once (lane l reads list l's) and
                    // move to the next list that still has segments, if any
                    const unsigned int seen = __hip_atomic_load(&cursors[lane * kWfCursorStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long open = lanes_where((unsigned long long)seen * kWfCursors + (unsigned int)lane < (unsigned long long)A.segments);
                    if (!open) { dry = true; break; }
                    const unsigned long long after = open & ~((2ull << list) - 1ull);
                    list = after ? __builtin_ctzll(after) : __builtin_ctzll(open);
                    continue;
                }
                const unsigned int n = A.cur.count[k];
                have_lo = (size_t)k * kWfSegment;
                have_hi = have_lo + n;
                rays += n;
                continue;
            }
            const unsigned int rank = (unsigned int)__builtin_popcountll(idle & ((1ull << lane) - 1ull));
            const unsigned int take = (unsigned int)min((size_t)__builtin_popcountll(idle), have_hi - have_lo);
            if (!walking && rank < take) {
                slot = have_lo + rank;
                const float4 r0 = A.cur.ray0[slot], r1 = A.cur.ray1[slot];
                ray.o = v3(r0.x, r0.y, r0.z); ray.d = v3(r0.w, r1.x, r1.y); ray.time = r1.z;
                ts.begin(INFINITY, sc.root_entry);
                walking = true;
            }
            have_lo += take;
        }
        const int active = __builtin_popcountll(lanes_where(walking));
        if (active == 0) break;
        // walk until 8 lanes have finished (or, with the queue dry, until every walk has ended)
        const bool was = walking;
        tree_run<FAST>(S, ray, 0.001f, ts, walking, stk, dry ? 1 : max(1, active - A.refill_lanes + 1), A.prim_batch);
        if (was && !walking) {
            const Hit h = ts.result();
            A.hit[slot] = make_float4(h.t, __int_as_float(h.obj), h.bu, h.bv);
        }
    }
    // a ray = one closest-hit query.  Counted per workgroup in a slot of its own (thousands of
    // atomics on one address would cost more than a late bounce's whole walk)
    __shared__ unsigned long long block_sum;
    if (threadIdx.x == 0) block_sum = 0;
    __syncthreads();
    if (lane == 0 && rays) atomicAdd(&block_sum, rays);
    __syncthreads();
    if (threadIdx.x == 0 && block_sum) A.block_rays[blockIdx.x] += block_sum;
}

// Adds up the per-workgroup ray counts of the trace kernels of a launch and clears them.
__global__ __launch_bounds__(kBlock) void wf_sum_rays(unsigned long long* block_rays, int blocks, unsigned long long* ray_counter) {
    __shared__ unsigned long long total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    unsigned long long mine = 0;
    for (int b = threadIdx.x; b < blocks; b += kBlock) { mine += block_rays[b]; block_rays[b] = 0; }
    if (mine) atomicAdd(&total, mine);
    __syncthreads();
    if (threadIdx.x == 0) *ray_counter += total;
}

__global__ __launch_bounds__(kBlock) void wf_shade(WfArgs A) {
    __shared__ uint32_t coop_all[kWaves * kCoopWordsPerWave];
    __shared__ unsigned int out_count;
    const DevScene& sc = A.sc;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* coop_pub = coop_all + wave * kCoopWordsPerWave;
    // (every lane has read the published streams before any lane stores a candidate point: one wave, LDS in order)
    float* coop_res = reinterpret_cast<float*>(coop_pub);
    const size_t cap = (size_t)A.batch_capacity;
    for (int seg = blockIdx.x; seg < A.segments; seg += gridDim.x) {
        __syncthreads();
        if (threadIdx.x == 0) out_count = 0;
        __syncthreads();
        const unsigned int count = A.cur.count[seg];
        const size_t seg_base = (size_t)seg * kWfSegment;
        for (unsigned int first = 0; first < count; first += kBlock) {
        const unsigned int i = first + threadIdx.x;
        const bool active = i < count;
        const size_t in = seg_base + i;
        Ray ray;
        ray.o = v3(0, 0, 0); ray.d = v3(0, 0, 1); ray.time = 0.0f;
        V3 throughput = v3(1, 1, 1), pend_albedo = v3(0, 0, 0);
        Rng rng;
        rng.pos = 0;
        unsigned int path = 0, bounces = 0;
        Hit h;
        h.t = 0.0f; h.obj = -1; h.bu = 0.0f; h.bv = 0.0f;
        if (active) {
            const float4 r0 = A.cur.ray0[in], r1 = A.cur.ray1[in], s0 = A.cur.st0[in], s1 = A.cur.st1[in], hh = A.hit[in];
            const uint2 m = A.cur.meta[in];
            ray.o = v3(r0.x, r0.y, r0.z); ray.d = v3(r0.w, r1.x, r1.y); ray.time = r1.z;
            throughput = v3(s0.x, s0.y, s0.z); rng.pos = __float_as_uint(s0.w);
            pend_albedo = v3(s1.x, s1.y, s1.z);
            path = m.x; bounces = m.y;
            h.t = hh.x; h.obj = __float_as_int(hh.y); h.bu = hh.z; h.bv = hh.w;
        }
        Surface sf = {};
        float4 m0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        int mat_type = -1;
        if (active && h.obj >= 0) {
            sf = make_surface<false>(sc, ray, h);
            m0 = *at32(sc.mats, sf.mat * 2);
            mat_type = material_kind(m0);
        }
        const V3 rv = coop_sphere_sample(mat_type == 0 || mat_type == 1, mat_type == 1, rng, coop_pub, coop_res, lane);
        bool alive = false;                                   // the path goes on to the next bounce
        if (active) {
            // what this segment of the path contributes; the aov values exist at bounce 0 only (render.cuh:53-58,67-73,84-90)
            V3 color = v3(0, 0, 0), albedo = v3(0, 0, 0), emission = v3(0, 0, 0), normal = v3(0, 0, 0);
            float depth = 0.0f, opacity = 0.0f;
            bool ended = false;
            if (h.obj >= 0) {
                V3 attenuation;
                Ray next;
                if (scatter_with(sc, ray, sf, m0, rv, rng, attenuation, next)) {             // render.cuh:52-61
                    if (bounces == 0) {
                        pend_albedo = attenuation;
                        normal = sf.has_unit ? sf.unit_normal : unit(sf.normal);
                        depth = h.t * length(ray.d);
                        opacity = 1.0f;
                    }
                    throughput = throughput * attenuation;
                    ray = next;
                    alive = bounces + 1 < (unsigned int)sc.max_bounces;     // exhausted: col = albedo = 0 (render.cuh:96-97)
                } else {                                                                      // render.cuh:64-74
                    const V3 emitted = emit(sc, sf, m0);
                    color = throughput * emitted;
                    albedo = pend_albedo;
                    if (bounces == 0) {
                        albedo = emitted;
                        emission = emitted;
                        normal = sf.has_unit ? sf.unit_normal : unit(sf.normal);
                        depth = h.t * length(ray.d);
                        opacity = 1.0f;
                    }
                    ended = true;
                }
            } else {                                                                          // render.cuh:76-91
                color = throughput * environment(sc, ray.d);
                albedo = pend_albedo;
                if (bounces == 0) {
                    albedo = color;
                    depth = __builtin_nanf("");                // depth of a miss is NaN (render.cuh:88)
                }
                ended = true;
            }
            float4* R = A.res + path;
            if (bounces == 0) {                                // the record is written whole at bounce 0
                R[0] = make_float4(color.x, color.y, color.z, depth);
                R[cap] = make_float4(albedo.x, albedo.y, albedo.z, opacity);
                R[2 * cap] = make_float4(emission.x, emission.y, emission.z, 0.0f);
                R[3 * cap] = make_float4(normal.x, normal.y, normal.z, 0.0f);
            } else if (ended) {                                // later: colour and albedo of the path's end
                float* r0 = reinterpret_cast<float*>(R);
                float* r1 = reinterpret_cast<float*>(R + cap);
                r0[0] = color.x; r0[1] = color.y; r0[2] = color.z;
                r1[0] = albedo.x; r1[1] = albedo.y; r1[2] = albedo.z;
            }
        }
        const size_t out = seg_base + wf_segment_slot(alive, &out_count, lane);
        if (alive) {
            A.next.ray0[out] = make_float4(ray.o.x, ray.o.y, ray.o.z, ray.d.x);
            A.next.ray1[out] = make_float4(ray.d.y, ray.d.z, ray.time, 0.0f);
            A.next.st0[out] = make_float4(throughput.x, throughput.y, throughput.z, __uint_as_float(rng.pos));
            A.next.st1[out] = make_float4(pend_albedo.x, pend_albedo.y, pend_albedo.z, 0.0f);
            A.next.meta[out] = make_uint2(path, bounces + 1u);
        }
        }
        __syncthreads();
        if (threadIdx.x == 0) A.next.count[seg] = out_count;
    }
}

// Adds the per-path records of a batch to the pixel sums, samples in ascending order.
__global__ __launch_bounds__(kBlock) void wf_accumulate(WfArgs A) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= A.pix_padded) return;
    const size_t cap = (size_t)A.batch_capacity;
    float v[kPasses];
#pragma unroll
    for (int k = 0; k < kPasses; k++) v[k] = A.accumulate_first ? 0.0f : A.partial[(size_t)k * A.pix_padded + t];
    const int tile = t / kWfTilePixels, pix = t - tile * kWfTilePixels;
    for (int s = 0; s < A.batch_samples; s++) {
        const float4* R = A.res + ((size_t)tile * A.batch_samples + s) * kWfTilePixels + pix;
        const float4 a = R[0], b = R[cap], c = R[2 * cap], d = R[3 * cap];
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[12] += a.w;
        v[3] += b.x; v[4] += b.y; v[5] += b.z; v[13] += b.w;
        v[6] += c.x; v[7] += c.y; v[8] += c.z;
        v[9] += d.x; v[10] += d.y; v[11] += d.z;
    }
#pragma unroll
    for (int k = 0; k < kPasses; k++) A.partial[(size_t)k * A.pix_padded + t] = v[k];
}

// Multi-GPU tile split: builds the full pass arrays from the ranks' band buffers after an
// all-gather.  gathered + r * stride holds rank r's six arrays back to back (color, albedo,
// emission, normal as [rows_r * W][3], depth and opacity as [rows_r * W]) for the rows it owns.
struct AssembleArgs {
    const float* gathered;
    size_t stride;
    float *color, *albedo, *emission, *normal, *depth, *opacity;
    int width, height, band_rows, band_count;
};
__global__ __launch_bounds__(kBlock) void assemble_bands(AssembleArgs A) {
    const size_t pixel = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (pixel >= (size_t)A.width * A.height) return;
    const int pj = (int)(pixel / A.width), pi = (int)(pixel - (size_t)pj * A.width);
    const int band = pj / A.band_rows, rank = band % A.band_count;
    const int local_row = (band / A.band_count) * A.band_rows + (pj - band * A.band_rows);
    // rows owned by `rank`
    const int period = A.band_rows * A.band_count;
    const int whole = A.height / period, rest = A.height - whole * period;
    const int rows = whole * A.band_rows + min(A.band_rows, max(0, rest - rank * A.band_rows));
    const size_t n = (size_t)rows * A.width, src = (size_t)local_row * A.width + pi;
    const float* base = A.gathered + (size_t)rank * A.stride;
    float* const dst3[4] = {A.color, A.albedo, A.emission, A.normal};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        dst3[k][3 * pixel] = base[3 * (k * n + src)];
        dst3[k][3 * pixel + 1] = base[3 * (k * n + src) + 1];
        dst3[k][3 * pixel + 2] = base[3 * (k * n + src) + 2];
    }
    A.depth[pixel] = base[12 * n + src];
    A.opacity[pixel] = base[13 * n + src];
}

// Bounce-0 closest hit per pixel for the parity tables.
struct PrimaryArgs {
    DevScene sc;
    FlatView flat;
    int sample;
    int32_t* obj;
    float *t, *u, *v;
    int32_t* front;
    int* stack_spill;
    int spill_threads;
};
template <int MODE, int STACK, int GROUPS>
__global__ __launch_bounds__(kBlock) void trace_primary_kernel(PrimaryArgs A) {
    extern __shared__ float4 smem[];
    int* stack_mem = reinterpret_cast<int*>(smem);
    float4* lds_scene = reinterpret_cast<float4*>(stack_mem + (STACK + (MODE == kModeFlat ? 0 : 8)) * kBlock);
    const DevScene& sc = A.sc;
    FlatLds FL = {};
    if constexpr (MODE == kModeFlat) FL = stage_flat_in_lds(A.flat, sc.xforms, sc.xform_count, lds_scene);
    __syncthreads();
    // grid-stride over the pixels: the grid is bounded so that the stacks' global part is too
    for (int idx = blockIdx.x * kBlock + threadIdx.x; idx < sc.width * sc.height; idx += gridDim.x * kBlock) {
        const int pi = idx % sc.width, pj = idx / sc.width;
        Rng rng;
        rng.init(idx, A.sample);
        Ray ray = camera_ray(sc.cam, pi, pj, rng);
        Hit h;
        if constexpr (MODE == kModeFlat) {
            SectionClock clk;
            clk.start();
            h = traverse_flat<GROUPS, 0>(FL, A.flat, ray, 0.001f, INFINITY, clk);
            if (h.obj >= 0) h.obj = *at32(A.flat.ent_obj, h.obj);
        } else {
            SceneGlobal S;
            S.nodes = sc.nodes; S.prims = sc.prims; S.xforms = sc.xforms; S.root_entry = sc.root_entry; S.order_flip = sc.order_flip;
            LaneStack stk;
            stk.attach(stack_mem, STACK, kBlock, threadIdx.x);
            stk.entry = (LdsFloatPtr)(reinterpret_cast<float*>(stack_mem + STACK * kBlock) + threadIdx.x);
            stk.code = (LdsIntPtr)(stack_mem + (STACK + 4) * kBlock + threadIdx.x);
            stk.spill = (GlobalIntPtr)(A.stack_spill + (size_t)blockIdx.x * kBlock + threadIdx.x);
            stk.spill_stride = (size_t)A.spill_threads;
            h = traverse(S, ray, 0.001f, INFINITY, stk);
        }
        A.obj[idx] = h.obj;
        if (h.obj >= 0) {
            Surface sf = make_surface<true>(sc, ray, h);
            A.t[idx] = h.t; A.u[idx] = sf.u; A.v[idx] = sf.v; A.front[idx] = sf.front ? 1 : 0;
        } else {
            A.t[idx] = INFINITY; A.u[idx] = 0.0f; A.v[idx] = 0.0f; A.front[idx] = 0;
        }
    }
}

// Single-function probes for known-answer tests: one thread per case evaluates ONE building block
// of the path with the same device functions the render kernels call.
//   kind 0 box      in[14] = min[3] max[3] origin[3] dir[3] tmin tmax      out[1]  = bit 0: slab test as the tree walk
//                                                                                     writes it, bit 1: as the flat mode does
//   kind 1 object   in[10] = object origin[3] dir[3] time tmin tmax         out[11] = hit t u v front normal[3] p[3]
//   kind 2 texture  in[6]  = texture u v p[3]                               out[3]  = colour
//   kind 3 camera   in[3]  = i j sample                                     out[7]  = origin[3] dir[3] time
// Integers travel as float bit patterns.
struct ProbeArgs {
    DevScene sc;
    int kind, count;
    const float* in;
    float* out;
};
__global__ __launch_bounds__(kBlock) void probe_kernel(ProbeArgs A) {
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= A.count) return;
    const DevScene& sc = A.sc;
    if (A.kind == 0) {
        const float* in = A.in + (size_t)idx * 14;
        const V3 mn = v3(in[0], in[1], in[2]), mx = v3(in[3], in[4], in[5]), o = v3(in[6], in[7], in[8]), d = v3(in[9], in[10], in[11]);
        const float tmin = in[12], tmax = in[13];
        const V3 inv = slab_reciprocal(d, 0);
        // tree walk: near / far plane picked by the sign of the reciprocal (tree_run)
        float t0 = ((inv.x < 0.0f ? mx.x : mn.x) - o.x) * inv.x, t1 = ((inv.x < 0.0f ? mn.x : mx.x) - o.x) * inv.x;
        float lo = fmaxf(tmin, t0), hi = fminf(tmax, t1);
        t0 = ((inv.y < 0.0f ? mx.y : mn.y) - o.y) * inv.y; t1 = ((inv.y < 0.0f ? mn.y : mx.y) - o.y) * inv.y;
        lo = fmaxf(lo, t0); hi = fminf(hi, t1);
        t0 = ((inv.z < 0.0f ? mx.z : mn.z) - o.z) * inv.z; t1 = ((inv.z < 0.0f ? mn.z : mx.z) - o.z) * inv.z;
        lo = fmaxf(lo, t0); hi = fminf(hi, t1);
        int result = !(hi < lo) ? 1 : 0;
        // flat mode: fmin / fmax of the two plane distances, open upper end (flat_box_hit); only
        // comparable when tmax is +inf
        NativeQuad b0 = {mn.x, mx.x, mn.y, mx.y}, b1 = {mn.z, mx.z, 0.0f, 0.0f};
        const unsigned long long m = flat_box_hit(b0, b1, o, inv, tmin);
        if ((m >> (threadIdx.x & 63)) & 1ull) result |= 2;
        A.out[idx] = __int_as_float(result);
    } else if (A.kind == 1) {
        const float* in = A.in + (size_t)idx * 10;
        float* out = A.out + (size_t)idx * 11;
        const int obj = __float_as_int(in[0]);
        Ray r;
        r.o = v3(in[1], in[2], in[3]); r.d = v3(in[4], in[5], in[6]); r.time = in[7];
        const float tmin = in[8], tmax = in[9];
        for (int k = 0; k < 11; k++) out[k] = 0.0f;
        if (obj < 0 || obj >= sc.object_count) return;
        SceneGlobal S;
        S.nodes = sc.nodes; S.prims = sc.prims; S.xforms = sc.xforms; S.root_entry = sc.root_entry; S.order_flip = sc.order_flip;
        const float4 q0 = S.prim(obj, 0), q1 = S.prim(obj, 1), q2 = S.prim(obj, 2);
        const int xf = __float_as_int(q2.y), type = __float_as_int(q2.z);
        V3 o = r.o, d = r.d;
        if (xf >= 0) to_object_space(xform_record(S.xforms, xf), (xf & kXfIdentityRotation) != 0, r.o, r.d, o, d);
        Hit h;
        h.t = 0.0f; h.obj = obj; h.bu = 0.0f; h.bv = 0.0f;
        bool ok;
        if (type == 2) ok = tri_test(o, d, v3(q0.x, q0.y, q0.z), v3(q0.w, q1.x, q1.y), v3(q1.z, q1.w, q2.x), tmin, tmax, 0, h.t, h.bu, h.bv);
        else ok = sphere_test<0>(o, d, r.time, v3(q0.x, q0.y, q0.z), v3(q1.x, q1.y, q1.z), q0.w, tmin, tmax, h.t);
        if (!ok) return;
        const Surface sf = make_surface<true>(sc, r, h);
        out[0] = 1.0f; out[1] = h.t; out[2] = sf.u; out[3] = sf.v; out[4] = sf.front ? 1.0f : 0.0f;
        out[5] = sf.normal.x; out[6] = sf.normal.y; out[7] = sf.normal.z;
        out[8] = sf.p.x; out[9] = sf.p.y; out[10] = sf.p.z;
    } else if (A.kind == 2) {
        const float* in = A.in + (size_t)idx * 6;
        const int tex = __float_as_int(in[0]);
        V3 c = v3(0.0f, 0.0f, 0.0f);
        if (tex >= 0 && tex < sc.texture_count) c = texture_value(sc, tex, in[1], in[2], v3(in[3], in[4], in[5]));
        A.out[(size_t)idx * 3] = c.x; A.out[(size_t)idx * 3 + 1] = c.y; A.out[(size_t)idx * 3 + 2] = c.z;
    } else {
        const float* in = A.in + (size_t)idx * 3;
        const int i = __float_as_int(in[0]), j = __float_as_int(in[1]), sample = __float_as_int(in[2]);
        Rng rng;
        rng.init(j * sc.width + i, sample);
        DevCamera cam = sc.cam;
        cam.needs_time = 1;                                  // the probe always reports the drawn time
        const Ray r = camera_ray(cam, i, j, rng);
        float* out = A.out + (size_t)idx * 7;
        out[0] = r.o.x; out[1] = r.o.y; out[2] = r.o.z; out[3] = r.d.x; out[4] = r.d.y; out[5] = r.d.z; out[6] = r.time;
    }
}

}  // namespace crt

// ====================================================================== host side =======
using namespace crt;

// Scratch of crt_render_static(), the entry point for callers that must not allocate: counters
// (the first kCounterBytes) and chunk sums.  One copy per GPU, held until the process ends.  In host code
// the array's name is only a handle for hipGetSymbolAddress(); the device address it yields (one
// per device, taken with that device current) is what every launch and memset uses.
constexpr size_t kStaticCounterBytes = kCounterBytes;
constexpr size_t kStaticScratchBytes = (size_t)64 << 20;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

constexpr int kMaxDevices = 64;
constexpr int kWfMaxTraceBlocks = 8192;
static unsigned char* g_static_scratch[kMaxDevices];     // device address of `scratch` on each GPU
static std::atomic<int> g_static_busy[kMaxDevices];      // 1 while a crt_render_static call uses that GPU's array
static std::mutex g_static_scratch_mutex;

// Device address of `scratch` on `device`, which must be the calling thread's current device.
static hipError_t static_scratch_address(int device, unsigned char** out) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_static_scratch_mutex);
    if (!g_static_scratch[device]) {
        void* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        g_static_scratch[device] = static_cast<unsigned char*>(ptr);
    }
    *out = g_static_scratch[device];
    return hipSuccess;
}

struct crt_scene {
    int device = 0, cu_count = 256;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_mid = nullptr, ev_stop = nullptr;
    float last_trace_ms = 0.0f, last_resolve_ms = 0.0f;
    void* blob = nullptr;            // nodes | prims | shade | xforms | mats | texs | images
    std::vector<void*> image_mem;
    DevScene dev = {};
    crt_render_info render = {};
    float far_margin = 0.0f;         // flat mode: largest margin of a cuboid whose far faces are tested last (0: none)
    int stack_entries = 8;           // LDS rows of the traversal stack (8 or 12): one row of -1 and 7 or 11 entries
    int spill_entries = 0;           // entries beyond them, in global memory
    int spill_threads = 0;
    int* stack_spill = nullptr;
    int lds_scene_bytes = 0;         // field-major copy of nodes + prims + xforms
    int lds_flat_bytes = 0;          // flat mode: slot records + xforms
    bool can_lds = false, can_flat = false;
    int features = kFeatAll;         // kFeat* bits of what the scene contains (selects the kernel variant)
    FlatView flat = {};
    int mode = 0;                    // 0 auto, 1 tree in HBM, 2 tree in LDS, 3 flat
    bool ordered = false;            // walked in the reference's order (a primitive sticks out of its box)
    float* partial = nullptr;        // chunk sums, sized at creation for the largest launch of this scene
    size_t partial_bytes = 0;
    unsigned long long* ray_counter = nullptr;   // kCounterBytes (rays traced, work-item counters, cost classes), then the item costs and the item order
    // Launch order by measured cost: what every item of the last launch cost, the order made from it, and the cut
    // of the frame (tiles, chunks, bands) that order belongs to.
    bool order_ready = false;
    int order_key[6] = {0, 0, 0, 0, 0, 0};
    // Whether that order pays is measured, not assumed (it gains up to a sixth where item costs differ widely and
    // loses a few percent where they do not: neighbouring tiles no longer run at the same time and share fewer
    // texels).  Launches of one cut of the frame: 0 natural order (nothing measured yet), 1 by cost (neither is
    // timed: the first launches of a scene pay for cold caches and rising clocks), 2 and 4 natural order, 3 and 5 by
    // cost, all four timed, 6... the order with the fastest timed launch.
    int order_stage = 0, launched_stage = -1;
    float by_cost_ms = 0.0f, natural_ms = 0.0f;
    static constexpr int kOrderDecided = 6;
    unsigned long long* host_counter = nullptr;  // pinned: the ray count of the last launch lands here
    unsigned char* static_scratch = nullptr;     // this GPU's copy of `scratch` (crt_render_static)
    crt_scene_options opt = {};
    bool launched = false;
    // wavefront path (large scenes): queues, hits and per-path records for one batch of samples
    bool wavefront = false;
    size_t wf_capacity = 0;          // paths per batch
    void* wf_mem = nullptr;          // everything below, one allocation
    WfQueue wf_queue[2] = {};
    float4* wf_hit = nullptr;
    float4* wf_res = nullptr;
    unsigned int* wf_cursors = nullptr;  // [max_bounces + 1] segment cursors of the trace kernels
    unsigned long long* wf_block_rays = nullptr;   // [kWfMaxTraceBlocks], zero between launches
};

struct crt_buffers {
    int width = 0, height = 0;
    float* base = nullptr;           // 14*W*H floats
    bool owned = false;
    float* color() const { return base; }
    float* albedo() const { return base + 3 * (size_t)width * height; }
    float* emission() const { return base + 6 * (size_t)width * height; }
    float* normal() const { return base + 9 * (size_t)width * height; }
    float* depth() const { return base + 12 * (size_t)width * height; }
    float* opacity() const { return base + 13 * (size_t)width * height; }
};

namespace {

#define CRT_HIP(call)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return crt::set_error(CRT_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_));       \
    } while (0)

bool identity9(const float* m) {
    const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < 9; i++)
        if (m[i] != I[i]) return false;
    return true;
}
bool zero3(const float* v) { return v[0] == 0.0f && v[1] == 0.0f && v[2] == 0.0f; }

float as_float(int32_t i) { float f; std::memcpy(&f, &i, 4); return f; }

// Checks every index the kernels will follow, so that no launch can read out of bounds.
int validate(const crt_scene_desc* d, int& bvh_depth, std::string& why) {
    auto bad = [&](const std::string& s) { why = s; return CRT_ERR_INVALID; };
    if (d->object_count < 0 || d->material_count < 0 || d->texture_count <= 0) return bad("counts must be non-negative and there must be a background texture");
    if (d->object_count && !d->objects) return bad("objects is null");
    if (d->material_count && !d->materials) return bad("materials is null");
    if (!d->textures) return bad("textures is null");
    if (d->background_index < 0 || d->background_index >= d->texture_count) return bad("background_index out of range");
    if (d->render.image_width <= 0 || d->render.image_height <= 0) return bad("image size must be positive");
    if ((long long)d->render.image_width * d->render.image_height > (1ll << 30)) return bad("image too large");
    if (d->render.max_bounces < 0) return bad("max_bounces must be >= 0");
    if (d->camera.type != CRT_CAM_PERSPECTIVE && d->camera.type != CRT_CAM_ORTHOGRAPHIC) return bad("unknown camera type");
    for (int i = 0; i < d->texture_count; i++) {
        const crt_texture& t = d->textures[i];
        if (t.type == CRT_TEX_CHECKER) {
            // children must come earlier in the list: this is what the loader produces and it
            // makes the texture graph acyclic (the device walks it with a bounded loop)
            if (t.odd_idx < 0 || t.odd_idx >= i || t.even_idx < 0 || t.even_idx >= i) return bad("checker texture " + std::to_string(i) + " must reference earlier textures");
        } else if (t.type == CRT_TEX_IMAGE) {
            if (t.width <= 0 || t.height <= 0 || (t.channels != 1 && t.channels != 4) || !t.pixels) return bad("image texture " + std::to_string(i) + " is invalid (channels must be 1 or 4)");
        } else if (t.type != CRT_TEX_SOLID_COLOR) return bad("unknown texture type");
    }
    for (int i = 0; i < d->material_count; i++) {
        const crt_material& m = d->materials[i];
        if (m.type < CRT_MAT_LAMBERTIAN || m.type > CRT_MAT_DIFFUSE_LIGHT) return bad("unknown material type");
        if ((m.type == CRT_MAT_LAMBERTIAN || m.type == CRT_MAT_DIFFUSE_LIGHT) && (m.tex_idx < 0 || m.tex_idx >= d->texture_count))
            return bad("material " + std::to_string(i) + " texture index out of range");
    }
    for (int i = 0; i < d->object_count; i++) {
        const crt_object& o = d->objects[i];
        if (o.type != CRT_OBJ_SPHERE && o.type != CRT_OBJ_TRI) return bad("object " + std::to_string(i) + " has unknown type");
        if (o.mat_idx < 0 || o.mat_idx >= d->material_count) return bad("object " + std::to_string(i) + " material index out of range");
        if (o.normal_idx >= d->texture_count) return bad("object " + std::to_string(i) + " normal texture index out of range");
        if (o.normal_idx >= 0 && d->textures[o.normal_idx].type != CRT_TEX_IMAGE) return bad("object " + std::to_string(i) + " normal map must be an image texture");
    }
    const crt_bvh4& b = d->bvh;
    if (b.slot_count < 4 || b.slot_count % 4 != 0 || !b.min_bounds || !b.max_bounds || !b.children || !b.object_index) return bad("BVH arrays missing");
    const int nodes = b.slot_count / 4;
    // every node except the root must be referenced exactly once and only by a lower-numbered
    // node (pre-order numbering), which also rules out cycles
    std::vector<int> depth(nodes, 0);
    std::vector<uint8_t> seen(nodes, 0);
    depth[0] = 1;
    seen[0] = 1;
    bvh_depth = 1;
    for (int n = 0; n < nodes; n++) {
        if (!seen[n]) return bad("BVH node " + std::to_string(n) + " is unreachable or referenced before its parent");
        for (int i = 0; i < 4; i++) {
            int s = n * 4 + i;
            int c = b.children[s], o = b.object_index[s];
            if (o >= d->object_count || o < -1) return bad("BVH object index out of range");
            if (c < -1 || c >= nodes) return bad("BVH child index out of range");
            if (c >= 0) {
                if (c <= n || seen[c]) return bad("BVH child links must form a pre-order tree");
                seen[c] = 1;
                depth[c] = depth[n] + 1;
                bvh_depth = std::max(bvh_depth, depth[c]);
            }
        }
    }
    return CRT_OK;
}

struct Blob {
    std::vector<uint8_t> bytes;
    size_t add(const void* p, size_t n) {
        size_t off = (bytes.size() + 255) & ~size_t(255);
        bytes.resize(off + n);
        if (n) std::memcpy(bytes.data() + off, p, n);
        return off;
    }
};

void free_scene(crt_scene* sc) {
    if (!sc) return;
    (void)hipSetDevice(sc->device);
    if (sc->stream) (void)hipStreamSynchronize(sc->stream);
    for (void* p : sc->image_mem) (void)hipFree(p);
    if (sc->blob) (void)hipFree(sc->blob);
    if (sc->partial) (void)hipFree(sc->partial);
    if (sc->wf_mem) (void)hipFree(sc->wf_mem);
    if (sc->ray_counter) (void)hipFree(sc->ray_counter);
    if (sc->host_counter) (void)hipHostFree(sc->host_counter);
    if (sc->stack_spill) (void)hipFree(sc->stack_spill);
    if (sc->ev_start) (void)hipEventDestroy(sc->ev_start);
    if (sc->ev_mid) (void)hipEventDestroy(sc->ev_mid);
    if (sc->ev_stop) (void)hipEventDestroy(sc->ev_stop);
    if (sc->stream) (void)hipStreamDestroy(sc->stream);
    delete sc;
}

int kernel_mode(const crt_scene* sc) {
    switch (sc->mode) {
    case 1: return kModeTreeGlobal;
    case 2: return sc->can_lds ? kModeTreeLds : kModeTreeGlobal;
    case 3: return sc->can_flat ? kModeFlat : (sc->can_lds ? kModeTreeLds : kModeTreeGlobal);
    default: return sc->can_flat ? kModeFlat : (sc->can_lds ? kModeTreeLds : kModeTreeGlobal);
    }
}

int variant_features(const crt_scene* sc);

template <int STACK, int FAST, int FEAT>
int launch_tree_feat(crt_scene* sc, const RenderArgs& A, dim3 grid, int mode) {
    size_t lds = (size_t)kWaves * wave_lds_bytes(kModeTreeGlobal) + (size_t)(STACK + 8) * kBlock * sizeof(int);
    if (mode == kModeTreeLds) {
        lds += (size_t)sc->lds_scene_bytes;
        hipLaunchKernelGGL((render_fused<kModeTreeLds, STACK, 0, FAST, FEAT>), grid, dim3(kBlock), lds, sc->stream, A);
    } else {
        hipLaunchKernelGGL((render_fused<kModeTreeGlobal, STACK, 0, FAST, FEAT>), grid, dim3(kBlock), lds, sc->stream, A);
    }
    CRT_HIP(hipGetLastError());
    return CRT_OK;
}
template <int STACK, int FAST>
int launch_tree(crt_scene* sc, const RenderArgs& A, dim3 grid, int mode) {
    if (variant_features(sc) == kFeatBasic) return launch_tree_feat<STACK, FAST, kFeatBasic>(sc, A, grid, mode);
    if (variant_features(sc) & kFeatExactTies) return launch_tree_feat<STACK, FAST, kFeatAll | kFeatExactTies>(sc, A, grid, mode);
    return launch_tree_feat<STACK, FAST, kFeatAll>(sc, A, grid, mode);
}

template <int GROUPS, int FEAT>
void launch_flat_feat(crt_scene* sc, const RenderArgs& A, dim3 grid, size_t lds) {
    if (sc->dev.fast_rcp) hipLaunchKernelGGL((render_fused<kModeFlat, 0, GROUPS, 1, FEAT>), grid, dim3(kBlock), lds, sc->stream, A);
    else hipLaunchKernelGGL((render_fused<kModeFlat, 0, GROUPS, 0, FEAT>), grid, dim3(kBlock), lds, sc->stream, A);
}
// The feature set of the kernel variant a scene runs: the smallest compiled set that covers it.
int variant_features(const crt_scene* sc) {
    if (kernel_mode(sc) == kModeFlat) {
        if ((sc->features & ~kFeatBasicXform) == 0) return kFeatBasicXform;
        if ((sc->features & ~kFeatPlain) == 0) return kFeatPlain;
        if ((sc->features & ~kFeatTextured) == 0) return kFeatTextured;
        return (sc->features & kFeatWide) ? kFeatAll : (kFeatAll & ~kFeatWide);
    }
    // the flat mode's feature bits say nothing about a tree walk
    if (sc->opt.exact_ties) return kFeatAll | kFeatExactTies;
    return (sc->features & ~(kFeatSlotXf | kFeatWide)) == 0 ? kFeatBasic : kFeatAll;
}
template <int GROUPS>
void launch_flat(crt_scene* sc, const RenderArgs& A, dim3 grid, size_t lds) {
    switch (variant_features(sc)) {
    case kFeatBasicXform: launch_flat_feat<GROUPS, kFeatBasicXform>(sc, A, grid, lds); break;
    case kFeatPlain: launch_flat_feat<GROUPS, kFeatPlain>(sc, A, grid, lds); break;
    case kFeatTextured: launch_flat_feat<GROUPS, kFeatTextured>(sc, A, grid, lds); break;
    case kFeatAll & ~kFeatWide: launch_flat_feat<GROUPS, kFeatAll & ~kFeatWide>(sc, A, grid, lds); break;
    default: launch_flat_feat<GROUPS, kFeatAll>(sc, A, grid, lds); break;
    }
}

int launch_fused(crt_scene* sc, const RenderArgs& A, dim3 grid) {
    const int mode = kernel_mode(sc);
    if (mode == kModeFlat) {
        size_t lds = (size_t)kWaves * wave_lds_bytes(kModeFlat) + (size_t)kParkWords * kBlock * sizeof(float) + (size_t)sc->lds_flat_bytes;
        switch (sc->flat.groups) {
        case 0: launch_flat<0>(sc, A, grid, lds); break;
        case 1: launch_flat<1>(sc, A, grid, lds); break;
        default: launch_flat<2>(sc, A, grid, lds); break;
        }
        CRT_HIP(hipGetLastError());
        return CRT_OK;
    }
    if (sc->dev.fast_rcp) return sc->stack_entries == 8 ? launch_tree<8, 1>(sc, A, grid, mode) : launch_tree<kDeepStackRows, 1>(sc, A, grid, mode);
    return sc->stack_entries == 8 ? launch_tree<8, 0>(sc, A, grid, mode) : launch_tree<kDeepStackRows, 0>(sc, A, grid, mode);
}

// True when every divisor the kernels invert on the path (triangle determinants that pass the
// 1e-8 test, ray direction components beyond 1e-12, and the divisors of quotient(): squared and
// plain lengths of directions and normals, sphere radii) is guaranteed to lie in [2^-100, 2^100],
// where the Newton-refined hardware reciprocal equals the IEEE quotient bit for bit.  The lower
// ends are the path's own thresholds; the upper ends follow from the scene: a direction is a
// camera ray (bounded by the view plane), normal + unit vector, a reflection + fuzz * unit-ball
// vector or a refraction, possibly taken through an object's inverse rotation, and
// |det| = |e1 . (d x e2)| <= |e1| |e2| |d|.
bool fast_reciprocal_is_exact(const crt_scene_desc* d) {
    auto len = [](const float* v) { return std::sqrt((double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2]); };
    const crt_camera& c = d->camera;
    double dir = 2.5;
    if (c.type == CRT_CAM_PERSPECTIVE) {
        const float to_corner[3] = {c.lower_left_corner[0] - c.center[0], c.lower_left_corner[1] - c.center[1], c.lower_left_corner[2] - c.center[2]};
        dir = std::max(dir, len(to_corner) + (d->render.image_width + 1.0) * len(c.pixel_delta_u) + (d->render.image_height + 1.0) * len(c.pixel_delta_v) +
                                2.0 * (len(c.defocus_disk_u) + len(c.defocus_disk_v)));
    } else {
        dir = std::max(dir, len(c.w));
    }
    // Longest surface normal a scatter can see.  Normal-mapped and smooth-shaded normals are
    // normalised on the device; a flat triangle's normal is whatever the description holds; a
    // sphere's is (p - centre) / radius, about 1 but with the rounding error of p, which is
    // relative to the size of the coordinates, divided by the radius.
    double normal = 1.0, rot = 1.0, edges = 0.0;
    for (int i = 0; i < d->object_count; i++) {
        const crt_object& o = d->objects[i];
        double f = 0.0;
        for (int k = 0; k < 9; k++) f += (double)o.rotation_inv[k] * o.rotation_inv[k];
        rot = std::max(rot, std::sqrt(f));
        double g = 0.0;
        for (int k = 0; k < 9; k++) g += (double)o.rotation[k] * o.rotation[k];
        const double rot_fwd = std::max(1.0, std::sqrt(g));         // normals are rotated to world space
        if (o.type == CRT_OBJ_TRI) {
            edges = std::max(edges, len(o.p1) * len(o.p2));
            if (o.normal_idx < 0 && !o.shade_smooth) normal = std::max(normal, len(o.n0) * rot_fwd);
        } else {
            // the radius divides (p - centre), |d|^2 divides the roots: both through quotient()
            const double radius = std::fabs((double)o.p2[0]);
            if (!(radius > 0x1p-80 && radius < 0x1p80)) return false;
            if (o.normal_idx >= 0) continue;
            const double reach = len(o.p0) + len(o.p1) + radius + len(o.anchor) + len(o.translation);
            normal = std::max(normal, (1.0 + 4e-6 * reach / radius) * rot_fwd);
        }
    }
    // |v - 2 (v.n) n| for a unit v
    const double reflected = 1.0 + 2.0 * normal * normal;
    dir = std::max(dir, normal + 1.0);                               // lambertian: normal + unit vector
    for (int i = 0; i < d->material_count; i++) {
        const crt_material& m = d->materials[i];
        if (m.type == CRT_MAT_METAL) dir = std::max(dir, reflected + std::fabs((double)m.fuzz));
        if (m.type == CRT_MAT_DIELECTRIC) {
            const double ior = std::fabs((double)m.ior);
            if (!(ior > 0.0) || !std::isfinite(ior)) return false;
            const double ratio = std::max(ior, 1.0 / ior);
            const double perp = ratio * (1.0 + normal);              // ratio * (uv + cos * n), |cos| <= 1
            dir = std::max(dir, std::max(reflected, perp + (1.0 + perp) * normal));   // + sqrt(|1 - perp^2|) * n
        }
    }
    const double dir_object = dir * rot;
    // |d|^2 (sphere roots) and |d| (unit vectors) are divisors too: 3 dir^2 must stay below 2^100.  (Their lower ends
    // are not proven but practically certain: a scattered direction is normal + unit vector with a component
    // beyond 1e-8, a reflection plus a bounded fuzz vector, a refraction, or a camera ray.)
    return std::isfinite(dir_object) && std::isfinite(edges) && dir_object < 0x1p48 && edges * dir_object < 0x1p99;
}

// The largest distance between a point a ray can start from (the camera's ray origins, any surface) and a point
// of the scene: the diagonal of the box around every object's box and the camera.
double scene_reach(const crt_scene_desc* d) {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    auto add = [&](double x, double y, double z) {
        const double p[3] = {x, y, z};
        for (int a = 0; a < 3; a++) { lo[a] = std::fmin(lo[a], p[a]); hi[a] = std::fmax(hi[a], p[a]); }
    };
    for (int i = 0; i < d->object_count; i++) {
        const crt_object& o = d->objects[i];
        if (!o.visible) continue;
        add(o.bbox_min[0], o.bbox_min[1], o.bbox_min[2]);
        add(o.bbox_max[0], o.bbox_max[1], o.bbox_max[2]);
    }
    const crt_camera& c = d->camera;
    const double w = d->render.image_width + 1.0, h = d->render.image_height + 1.0;
    for (int corner = 0; corner < 4; corner++) {
        const double i = (corner & 1) ? w : -1.0, j = (corner & 2) ? h : -1.0;
        double p[3];
        for (int a = 0; a < 3; a++) p[a] = c.lower_left_corner[a] + i * c.pixel_delta_u[a] + j * c.pixel_delta_v[a];
        if (c.type != CRT_CAM_PERSPECTIVE) add(p[0], p[1], p[2]);              // orthographic rays start on the view plane
    }
    for (int su = -1; su <= 1; su += 2)
        for (int sv = -1; sv <= 1; sv += 2)
            add(c.center[0] + su * c.defocus_disk_u[0] + sv * c.defocus_disk_v[0], c.center[1] + su * c.defocus_disk_u[1] + sv * c.defocus_disk_v[1],
                c.center[2] + su * c.defocus_disk_u[2] + sv * c.defocus_disk_v[2]);
    double sq = 0.0;
    for (int a = 0; a < 3; a++) sq += (hi[a] - lo[a]) * (hi[a] - lo[a]);
    return std::sqrt(sq);
}

// Leaf slots in the order the reference's traversal visits them when every box is hit
// (bvh.cuh:387-419): slots 0..3 of a node in order, pushed children popped last-in first-out.
void reference_visit_order(const crt_bvh4& b, std::vector<int>& leaf_slots) {
    std::vector<int> stack;
    stack.push_back(0);
    while (!stack.empty()) {
        int node = stack.back();
        stack.pop_back();
        for (int i = 0; i < 4; i++) {
            int s = node * 4 + i;
            if (b.object_index[s] >= 0) leaf_slots.push_back(s);
            if (b.children[s] >= 0) stack.push_back(b.children[s]);
        }
    }
}

// Does anything look at the surface coordinates (u, v) of this object: a normal map, or an image
// texture somewhere under its material's texture (checkers pass (u, v) on to their children)?
bool texture_reads_uv(const crt_scene_desc* d, int tex, int depth) {
    if (tex < 0) return false;
    if (tex >= d->texture_count || depth > 64) return true;          // out of the ordinary: assume it does
    const crt_texture& t = d->textures[tex];
    if (t.type == CRT_TEX_IMAGE) return true;
    if (t.type == CRT_TEX_CHECKER) return texture_reads_uv(d, t.odd_idx, depth + 1) || texture_reads_uv(d, t.even_idx, depth + 1);
    return false;
}
bool reads_uv(const crt_scene_desc* d, const crt_object& o) {
    if (o.normal_idx >= 0) return true;
    if (o.mat_idx < 0 || o.mat_idx >= d->material_count) return true;
    return texture_reads_uv(d, d->materials[o.mat_idx].tex_idx, 0);
}

// Does every primitive lie inside the box of its own leaf slot?  The reference shrinks the ray
// interval to the closest hit so far while it walks (bvh.cuh:398,407-411), so a primitive is
// only tested if its leaf box is still hit at that moment.  When every box contains its
// primitive, the closest hit comes out the same whatever the order of the walk.  When one does
// not, it depends on the reference's own order.  That happens: hittable::translate also moves
// the rotation anchor, and hittable::rotate replaces an earlier rotation and re-adds the
// translation to an already translated box (hittable.cuh:77-127 against :429), so a translate
// wrapped around a rotate, or nested rotates, leave the box and the tested geometry apart.
// Such a scene is walked in the reference's order over the reference's tree.
bool primitives_inside_their_boxes(const crt_scene_desc* d) {
    for (int s = 0; s < d->bvh.slot_count; s++) {
        const int oi = d->bvh.object_index[s];
        if (oi < 0) continue;
        const crt_object& o = d->objects[oi];
        const float* lo = &d->bvh.min_bounds[3 * s];
        const float* hi = &d->bvh.max_bounds[3 * s];
        float scale = 1.0f;
        for (int a = 0; a < 3; a++) scale = std::fmax(scale, std::fmax(std::fabs(lo[a]), std::fabs(hi[a])));
        const float tol = 1e-5f * scale;
        const bool moved = !(identity9(o.rotation) && identity9(o.rotation_inv) && zero3(o.anchor) && zero3(o.translation));
        auto inside = [&](const float p[3], float radius) {
            float w[3] = {p[0], p[1], p[2]};
            if (moved) {                                            // inverse of the ray transform at hittable.cuh:429
                const float q[3] = {p[0] - o.anchor[0], p[1] - o.anchor[1], p[2] - o.anchor[2]};
                for (int r = 0; r < 3; r++)
                    w[r] = o.rotation[3 * r] * q[0] + o.rotation[3 * r + 1] * q[1] + o.rotation[3 * r + 2] * q[2] + o.anchor[r] + o.translation[r];
            }
            for (int a = 0; a < 3; a++)
                if (!(w[a] - radius >= lo[a] - tol && w[a] + radius <= hi[a] + tol)) return false;
            return true;
        };
        const float second[3] = {o.p0[0] + o.p1[0], o.p0[1] + o.p1[1], o.p0[2] + o.p1[2]};   // vertex 1 / centre at time 1
        if (o.type == CRT_OBJ_TRI) {
            const float third[3] = {o.p0[0] + o.p2[0], o.p0[1] + o.p2[1], o.p0[2] + o.p2[2]};
            if (!inside(o.p0, 0.0f) || !inside(second, 0.0f) || !inside(third, 0.0f)) return false;
        } else {
            const float radius = std::fabs(o.p2[0]);
            if (!inside(o.p0, radius) || !inside(second, radius)) return false;
        }
    }
    return true;
}

// How a launch of `samples` samples is cut into work items: every pixel tile x a chunk of the
// samples.  About 16 to 24 items per resident wave keep the end of the launch (waves running dry one by
// one) to a few percent of its length.  `capacity` = bytes of chunk-sum scratch available.
struct ChunkPlan {
    int tile_h, tiles_x, tiles_y, ntiles, pix_padded, chunk_size, chunks;
    int band_tiles;        // tile rows per band (whole image: all of them)
    size_t bytes;          // chunk sums of this plan
    bool fits;
};
// Row bands of a multi-GPU tile split: band `index` of `count` owns the pixel rows j with
// (j / rows) % count == index.  {0, 0, 1} = the whole image.
struct Bands {
    int rows, index, count;
};
// Number of pixel rows of an image of `height` rows that the band owns.
int band_row_count(int height, Bands b) {
    if (b.count <= 1) return height;
    const int period = b.rows * b.count;
    const int full = height / period, rest = height - full * period;
    return full * b.rows + std::min(b.rows, std::max(0, rest - b.index * b.rows));
}
ChunkPlan plan_chunks(const crt_scene* sc, int mode, int samples, size_t capacity, Bands bands = {0, 0, 1}) {
    ChunkPlan p;
    p.tile_h = tile_rows(mode);
    p.tiles_x = (sc->dev.width + kTileW - 1) / kTileW;
    p.tiles_y = (band_row_count(sc->dev.height, bands) + p.tile_h - 1) / p.tile_h;
    p.band_tiles = bands.count <= 1 ? std::max(1, p.tiles_y) : bands.rows / p.tile_h;
    p.ntiles = p.tiles_x * p.tiles_y;
    p.pix_padded = p.ntiles * kTileW * p.tile_h;
    const size_t chunk_bytes = (size_t)kPasses * p.pix_padded * sizeof(float);
    // (98304: the best compromise of a sweep over 32768 ... 262144 on the scenes of this repository; scenes with long
    // paths of very different lengths - glass spheres, 30 bounces - still gain 3 % at 131072, scenes with short paths lose 3 %)
    const int target_items = sc->opt.work_items > 0 ? sc->opt.work_items : 98304;
    long long chunks = sc->opt.deterministic ? samples : std::min<long long>(samples, std::max<long long>(1, ((long long)target_items + p.ntiles - 1) / p.ntiles));
    // A chunk holds at least eight samples where the launch has them (256 paths per item): a small cut of a frame - a
    // reduced image, one rank's bands of the tile split - would otherwise fall into items of one or two wave-fulls, and
    // taking and flushing an item (14 x 32 sums) then costs as much as tracing it.  Measured, one frame at a time: an eighth
    // of the Cornell frame 1.33 -> 1.10 ms, Cornell at 256 x 256 0.80 -> 0.69, materials at 500 x 100 1.29 -> 0.92, the
    // orthographic cube at 256 x 256 and 8 samples 0.18 -> 0.09 (tools/items_sweep.py); whole frames of the usual sizes
    // are cut as before.
    if (!sc->opt.deterministic && sc->opt.work_items == 0) chunks = std::min<long long>(chunks, std::max<long long>(1, samples / 8));
    p.fits = chunk_bytes <= capacity;
    if (!sc->opt.deterministic) chunks = std::max<long long>(1, std::min<long long>(chunks, (long long)(capacity / chunk_bytes)));
    else if ((size_t)chunks * chunk_bytes > capacity) p.fits = false;
    p.chunk_size = (int)((samples + chunks - 1) / chunks);
    // An even number of samples, so that an item's paths fill whole waves (a tile has 32 pixels).  Where paths are short
    // and alike - a sphere in front of an environment map: two rays each - the lanes of a wave stay in step, all at
    // their first hit in one round and all leaving the scene in the next, and each round runs one branch of the
    // shading code.  Half a wave of paths left over at the end of an item splits the wave into two groups a round
    // apart for the rest of the launch, and every round runs both branches: measured 20 % slower (that scene at 297
    // against 300 samples in three chunks), 10 % with an environment map alone, 6 % on the orthographic cube.
    if (!sc->opt.deterministic && p.chunk_size > 1) p.chunk_size += p.chunk_size & 1;
    p.chunks = (samples + p.chunk_size - 1) / p.chunk_size;
    p.bytes = (size_t)p.chunks * chunk_bytes;
    return p;
}

template <int STACK>
void launch_wf_trace(crt_scene* sc, const WfArgs& W, dim3 grid) {
    const size_t lds = (size_t)(STACK + 8) * kBlock * sizeof(int);
    if (sc->dev.fast_rcp) hipLaunchKernelGGL((wf_trace<STACK, 1>), grid, dim3(kBlock), lds, sc->stream, W);
    else hipLaunchKernelGGL((wf_trace<STACK, 0>), grid, dim3(kBlock), lds, sc->stream, W);
}

// The wavefront pipeline for one launch: the samples in batches that fit the path buffers, per
// batch one generate, max_bounces x (trace, shade) and one accumulate, all on the scene's stream
// with the queue lengths kept on the device (no host round trip between bounces).
int launch_wavefront(crt_scene* sc, const RenderArgs& A, const ChunkPlan& plan, Bands bands) {
    WfArgs W;
    W.sc = sc->dev;
    W.hit = sc->wf_hit;
    W.res = sc->wf_res;
    W.cursors = sc->wf_cursors;
    W.ray_counter = A.ray_counter;
    W.block_rays = sc->wf_block_rays;
    W.partial = A.partial;
    W.batch_capacity = (int)std::min<size_t>(sc->wf_capacity, 0x7fffffff);
    W.pix_padded = plan.pix_padded; W.tiles_x = plan.tiles_x;
    W.band_tiles = plan.band_tiles; W.band_index = bands.index; W.band_count = bands.count;
    W.stack_spill = sc->stack_spill; W.spill_threads = sc->spill_threads;
    W.prim_batch = A.prim_batch;
    W.refill_lanes = sc->opt.ready_percent > 0 ? std::max(1, 64 * sc->opt.ready_percent / 100) : 8;
    const int per_batch = (int)std::max<size_t>(1, sc->wf_capacity / (size_t)plan.pix_padded);
    if ((size_t)plan.pix_padded > sc->wf_capacity) return crt::set_error(CRT_ERR_UNSUPPORTED, "wavefront buffers hold less than one sample of this image");
    const int resident = sc->cu_count * 4;                                  // generate / shade / accumulate: grid-stride
    const dim3 trace_grid((unsigned)std::min(std::min(sc->cu_count * CRT_WF_TRACE_WAVES_PER_SIMD, sc->spill_threads / kBlock), kWfMaxTraceBlocks));
    const size_t counter_bytes = (size_t)(sc->dev.max_bounces + 1) * kWfCursors * kWfCursorStride * sizeof(unsigned int);
    for (int done = 0; done < A.samples; done += per_batch) {
        W.batch_samples = std::min(per_batch, A.samples - done);
        W.first_sample = A.first_sample + done;
        W.accumulate_first = done == 0 ? 1 : 0;
        const long long paths = (long long)W.batch_samples * plan.pix_padded;
        W.segments = (int)((paths + kWfSegment - 1) / kWfSegment);
        CRT_HIP(hipMemsetAsync(sc->wf_cursors, 0, counter_bytes, sc->stream));
        if (sc->dev.max_bounces == 0) CRT_HIP(hipMemsetAsync(sc->wf_res, 0, sc->wf_capacity * 64, sc->stream));   // no path starts: all records zero
        W.cur = sc->wf_queue[0]; W.next = sc->wf_queue[1];
        W.bounce = 0;
        const dim3 wide((unsigned)std::min<long long>(W.segments, (long long)resident * 2));
        hipLaunchKernelGGL(wf_generate, wide, dim3(kBlock), 0, sc->stream, W);
        for (int b = 0; b < sc->dev.max_bounces; b++) {
            W.bounce = b;
            if (sc->stack_entries == 8) launch_wf_trace<8>(sc, W, trace_grid);
            else launch_wf_trace<kDeepStackRows>(sc, W, trace_grid);
            hipLaunchKernelGGL(wf_shade, wide, dim3(kBlock), 0, sc->stream, W);
            std::swap(W.cur, W.next);
        }
        hipLaunchKernelGGL(wf_accumulate, dim3((unsigned)((plan.pix_padded + kBlock - 1) / kBlock)), dim3(kBlock), 0, sc->stream, W);
        CRT_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(wf_sum_rays, dim3(1), dim3(kBlock), 0, sc->stream, sc->wf_block_rays, (int)trace_grid.x, A.ray_counter);
    CRT_HIP(hipGetLastError());
    return CRT_OK;
}

}  // namespace

extern "C" {

int crt_device_count(int* count) {
    crt::clear_error();
    if (!count) return crt::set_error(CRT_ERR_INVALID, "crt_device_count: null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return crt::set_error(CRT_ERR_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *count = n;
    return CRT_OK;
}

int crt_set_device(int device) {
    crt::clear_error();
    CRT_HIP(hipSetDevice(device));
    return CRT_OK;
}

int crt_scene_options_init(crt_scene_options* options) {
    crt::clear_error();
    if (!options) return crt::set_error(CRT_ERR_INVALID, "crt_scene_options_init: null argument");
    std::memset(options, 0, sizeof(*options));
    options->struct_size = (int32_t)sizeof(*options);
    return CRT_OK;
}

int crt_scene_create(const crt_scene_desc* d, crt_scene** out) { return crt_scene_create_ex(d, nullptr, out); }

static int scene_create(const crt_scene_desc* d, const crt_scene_options* options, crt_scene** out);

int crt_scene_create_ex(const crt_scene_desc* d, const crt_scene_options* options, crt_scene** out) {
    try {                                                    // no exception may cross the C boundary
        return scene_create(d, options, out);
    } catch (const std::exception& e) {
        return crt::set_error(CRT_ERR_INVALID, std::string("crt_scene_create: ") + e.what());
    }
}

static int scene_create(const crt_scene_desc* d, const crt_scene_options* options, crt_scene** out) {
    crt::clear_error();
    if (!d || !out) return crt::set_error(CRT_ERR_INVALID, "crt_scene_create: null argument");
    crt_scene_options opt;
    std::memset(&opt, 0, sizeof(opt));
    if (options) {
        if (options->struct_size <= 0 || options->struct_size > (int32_t)sizeof(opt))
            return crt::set_error(CRT_ERR_INVALID, "crt_scene_create: options->struct_size is not set (use crt_scene_options_init)");
        std::memcpy(&opt, options, (size_t)options->struct_size);
    }
    opt.struct_size = (int32_t)sizeof(opt);
    if (opt.accel < CRT_ACCEL_AUTO || opt.accel > CRT_ACCEL_ORDERED) return crt::set_error(CRT_ERR_INVALID, "crt_scene_create: options->accel must be CRT_ACCEL_*");
    if (opt.mode < 0 || opt.mode > 3) return crt::set_error(CRT_ERR_INVALID, "crt_scene_create: options->mode must be 0..3");
    if (opt.general_kernel < 0 || opt.general_kernel > 1) return crt::set_error(CRT_ERR_INVALID, "crt_scene_create: options->general_kernel must be 0 or 1");
    if (opt.flat_cuboids < 0 || opt.flat_cuboids > 3) return crt::set_error(CRT_ERR_INVALID, "crt_scene_create: options->flat_cuboids must be 0..3");
    if (opt.wavefront < 0 || opt.wavefront > 2 || opt.wavefront_paths < 0) return crt::set_error(CRT_ERR_INVALID, "crt_scene_create: options->wavefront must be 0..2 and wavefront_paths >= 0");
    if (opt.item_order < 0 || opt.item_order > 2) return crt::set_error(CRT_ERR_INVALID, "crt_scene_create: options->item_order must be 0, 1 or 2");
    if (opt.chip_parts < 0 || opt.chip_parts > 16) return crt::set_error(CRT_ERR_INVALID, "crt_scene_create: options->chip_parts must be 0..16");
    if (opt.exact_ties < 0 || opt.exact_ties > 1) return crt::set_error(CRT_ERR_INVALID, "crt_scene_create: options->exact_ties must be 0 or 1");
    if (opt.work_items > (1 << 26)) return crt::set_error(CRT_ERR_INVALID, "crt_scene_create: options->work_items must not exceed 67 108 864");
    if (opt.work_items < 0 || opt.blocks < 0 || opt.ready_percent < 0 || opt.ready_percent > 100 || opt.prim_batch < 0 || opt.prim_batch > 64)
        return crt::set_error(CRT_ERR_INVALID, "crt_scene_create: option out of range (work_items, blocks >= 0; ready_percent 0..100; prim_batch 0..64)");
    int bvh_depth = 0;
    std::string why;
    if (validate(d, bvh_depth, why) != CRT_OK) return crt::set_error(CRT_ERR_INVALID, "crt_scene_create: " + why);
    // The hierarchy the tree modes walk: built here over the reference tree's leaf slots (see
    // accel_build.cpp); CRT_ACCEL_REFERENCE walks the reference tree itself.
    std::vector<AccelNode> accel;
    int accel_depth = bvh_depth;
    // CRT_ACCEL_ORDERED: the reference's tree in the reference's order, whatever the scene.
    const bool ordered = opt.accel == CRT_ACCEL_ORDERED || !primitives_inside_their_boxes(d);
    if (ordered || opt.accel == CRT_ACCEL_REFERENCE) accel_from_reference(d->bvh, accel);
    else build_accel(d->bvh, accel, accel_depth);
    // A node pushes at most 4 children and pops one before descending: <= 3*(depth-1)+1 live entries.
    int need = 3 * (accel_depth - 1) + 1;
    if (need > 1024) {
        // a pathologically unbalanced surface-area split (sizes in a geometric series): the
        // reference's median split is balanced by construction, walk that instead
        accel_from_reference(d->bvh, accel);
        accel_depth = bvh_depth;
        need = 3 * (accel_depth - 1) + 1;
    }
    if (need > 1024) return crt::set_error(CRT_ERR_UNSUPPORTED, "crt_scene_create: hierarchy deeper than 342 levels");
    // LDS part of the traversal stack: 8 rows for shallow trees, else 12, one of them the row of -1 below the
    // stacks (deeper entries go to the global part; with the 8 words of pending candidates beside it a
    // workgroup stays under 40 KB)
    const int stack_entries = need <= 7 ? 8 : kDeepStackRows;

    const int nodes = (int)accel.size();
    if (accel.size() >= (size_t(1) << 31) / kNodeEntryScale) return crt::set_error(CRT_ERR_UNSUPPORTED, "crt_scene_create: more than 16 million hierarchy nodes are not supported");
    auto accel_leafy = [&](int node) {
        for (int i = 0; i < 4; i++)
            if (accel[(size_t)node].object[i] >= 0) return true;
        return false;
    };
    // ---- repack: nodes (node-major, 8 float4), prims, shade, xforms (deduplicated), mats, texs
    std::vector<float> node_rec((size_t)nodes * kNodeStrideGlobal * 4, 0.0f);
    for (int n = 0; n < nodes; n++) {
        float* rec = &node_rec[(size_t)n * kNodeStrideGlobal * 4];
        const AccelNode& an = accel[(size_t)n];
        for (int i = 0; i < 4; i++) {
            for (int a = 0; a < 3; a++) {
                rec[(a)*4 + i] = an.lo[i][a];
                rec[(3 + a) * 4 + i] = an.hi[i][a];
            }
            const int o = an.object[i], c = an.child[i];
            // slot code: object index of a leaf slot, the child's stack entry (child * kNodeEntryScale + leafy;
            // leafy = the child node has leaf slots of its own) of an internal slot, -1 for an empty one;
            // which of the three a slot is stands in the `kinds` word below
            int32_t code = o >= 0 ? o : (c >= 0 ? c * kNodeEntryScale + (accel_leafy(c) ? 1 : 0) : -1);
            rec[6 * 4 + i] = as_float(code);
        }
        // Visiting order of the four slots for the direction octants 0-3 (bit 0/1 set: the ray runs
        // towards -x/-y; +z): slots sorted by their box centre along the octant's diagonal, nearest
        // first.  Stored per octant as 16 bits, for slot i in bits 4i..4i+3 the slots that come AFTER
        // it.  Octant 7-o is the mirror image of octant o and reads o's entry backwards
        // (DevScene::order_flip = 15).  A scene that has to be walked in the reference's order
        // (bvh.cuh:391-418: children pushed in slot order, popped last-in first-out) gets "3,2,1,0"
        // for every octant and order_flip = 0.  Third word: leaf slots (bits 0-3), internal slots (bits 4-7).
        uint32_t words[2] = {0, 0};
        for (int oct = 0; oct < 4; oct++) {
            float key[4];
            int idx[4] = {3, 2, 1, 0};
            if (!ordered) {
                for (int i = 0; i < 4; i++) {
                    idx[i] = i;
                    float k = 0.0f;
                    for (int a = 0; a < 3; a++) {
                        const float c = 0.5f * (an.lo[i][a] + an.hi[i][a]);
                        k += ((oct >> a) & 1) ? -c : c;
                    }
                    key[i] = std::isfinite(k) ? k : INFINITY;            // empty slots last
                }
                std::stable_sort(idx, idx + 4, [&](int a, int b) { return key[a] < key[b]; });
            }
            uint32_t entry = 0, after = 0;
            for (int r = 3; r >= 0; r--) {
                entry |= after << (4 * idx[r]);
                after |= 1u << idx[r];
            }
            words[oct >> 1] |= entry << ((oct & 1) * 16);
        }
        uint32_t kinds = 0;
        for (int i = 0; i < 4; i++) {
            if (an.object[i] >= 0) kinds |= 1u << i;
            else if (an.child[i] >= 0) kinds |= 16u << i;
        }
        rec[7 * 4 + 0] = as_float((int32_t)words[0]);
        rec[7 * 4 + 1] = as_float((int32_t)words[1]);
        rec[7 * 4 + 2] = as_float((int32_t)kinds);
    }
    // position of every object in the reference's leaf visiting order (tie rule of the closest hit)
    std::vector<int> leaf_slots;
    reference_visit_order(d->bvh, leaf_slots);
    std::vector<int32_t> visit_rank((size_t)d->object_count, 0);
    for (size_t rank = 0; rank < leaf_slots.size(); rank++) {
        const int oi = d->bvh.object_index[leaf_slots[rank]];
        if (oi >= 0 && oi < d->object_count) visit_rank[(size_t)oi] = (int32_t)rank;
    }
    std::vector<float> xforms;
    bool has_plain_xf = false;
    std::vector<float> prim_rec((size_t)d->object_count * kPrimQuads * 4, 0.0f), shade_rec((size_t)d->object_count * kShadeQuads * 4, 0.0f),
                       cold_rec((size_t)d->object_count * kShadeColdQuads * 4, 0.0f);
    for (int i = 0; i < d->object_count; i++) {
        const crt_object& o = d->objects[i];
        int xf = -1;
        if (!(identity9(o.rotation) && identity9(o.rotation_inv) && zero3(o.anchor) && zero3(o.translation))) {
            float rec[kXformFloats];
            std::memcpy(rec, o.rotation_inv, 36);
            std::memcpy(rec + 9, o.anchor, 12);
            std::memcpy(rec + 12, o.translation, 12);
            std::memcpy(rec + 15, o.rotation, 36);
            const int count = (int)(xforms.size() / kXformFloats);
            for (int k = count - 1; k >= 0 && k >= count - 64; k--)     // objects of one group sit next to each other
                if (!std::memcmp(&xforms[(size_t)k * kXformFloats], rec, sizeof(rec))) { xf = k; break; }
            if (xf < 0) { xf = count; xforms.insert(xforms.end(), rec, rec + kXformFloats); }
        }
        float* p = &prim_rec[(size_t)i * kPrimQuads * 4];
        if (o.type == CRT_OBJ_TRI) {
            p[0] = o.p0[0]; p[1] = o.p0[1]; p[2] = o.p0[2]; p[3] = o.p1[0];
            p[4] = o.p1[1]; p[5] = o.p1[2]; p[6] = o.p2[0]; p[7] = o.p2[1];
            p[8] = o.p2[2];
        } else {
            p[0] = o.p0[0]; p[1] = o.p0[1]; p[2] = o.p0[2]; p[3] = o.p2[0];
            p[4] = o.p1[0]; p[5] = o.p1[1]; p[6] = o.p1[2];
        }
        // transforms without a rotation (translated objects, every sphere) are marked: the kernels then skip the matrix products
        const int xf_coded = (xf >= 0 && identity9(o.rotation) && identity9(o.rotation_inv)) ? (xf | kXfIdentityRotation) : xf;
        if (xf_coded != xf) has_plain_xf = true;
        p[9] = as_float(xf_coded); p[10] = as_float(o.type); p[11] = as_float(visit_rank[(size_t)i]);
        // Shading record in two parts.  Hot, 32 bytes, read for every hit: flat normal (or first vertex
        // normal), material, second vertex normal or the precomputed unit world normal, and normal map +
        // flags in one word ((normal_idx + 1) << 3 | bit 0 smooth, bit 1 something reads (u, v), bit 2 unit
        // normal precomputed).  Cold, 64 bytes, only read for smooth, textured or normal-mapped objects:
        // third vertex normal, vertex uvs, tangent, bitangent.  A flat untextured triangle (every triangle of
        // a large mesh) costs a quarter of a cache line instead of a whole one per hit.
        float* s = &shade_rec[(size_t)i * kShadeQuads * 4];
        int32_t flags = (o.shade_smooth ? 1 : 0) | (reads_uv(d, o) ? 2 : 0);
        s[0] = o.n0[0]; s[1] = o.n0[1]; s[2] = o.n0[2]; s[3] = as_float(o.mat_idx);
        s[4] = o.n1[0]; s[5] = o.n1[1]; s[6] = o.n1[2];
        if (o.type == CRT_OBJ_TRI && !o.shade_smooth && o.normal_idx < 0) {
            // unit(normal) of the normal pass (render.cuh:55), as make_surface() would compute it for a
            // front-facing hit: the flat normal, rotated if the object is transformed, divided by its length
            V3 n = v3(o.n0);
            if (xf >= 0) n = matvec(o.rotation, n);
            n = unit(n);
            s[4] = n.x; s[5] = n.y; s[6] = n.z;
            flags |= 4;
        }
        s[7] = as_float(((o.normal_idx + 1) << 3) | flags);
        float* c = &cold_rec[(size_t)i * kShadeColdQuads * 4];
        c[0] = o.n2[0]; c[1] = o.n2[1]; c[2] = o.n2[2];
        c[4] = o.uv0[0]; c[5] = o.uv0[1]; c[6] = o.uv1[0]; c[7] = o.uv1[1];
        c[8] = o.uv2[0]; c[9] = o.uv2[1]; c[10] = o.tangent[0]; c[11] = o.tangent[1];
        c[12] = o.tangent[2]; c[13] = o.bitangent[0]; c[14] = o.bitangent[1]; c[15] = o.bitangent[2];
    }
    const bool xforms_used = !xforms.empty();
    if (xforms.empty()) xforms.assign(kXformFloats, 0.0f);
    std::vector<float> mat_rec((size_t)std::max(1, d->material_count) * 8, 0.0f);
    for (int i = 0; i < d->material_count; i++) {
        const crt_material& m = d->materials[i];
        float* r = &mat_rec[(size_t)i * 8];
        r[0] = as_float(m.type); r[1] = as_float(m.tex_idx); r[2] = m.fuzz; r[3] = m.ior;
        r[4] = m.albedo[0]; r[5] = m.albedo[1]; r[6] = m.albedo[2];
        // (layout and the two shortcuts: "Material record" in pt_device.h)
        if ((m.type == CRT_MAT_LAMBERTIAN || m.type == CRT_MAT_DIFFUSE_LIGHT) && m.tex_idx >= 0 && m.tex_idx < d->texture_count &&
            d->textures[m.tex_idx].type == CRT_TEX_SOLID_COLOR) {
            const crt_texture& t = d->textures[m.tex_idx];
            r[0] = as_float(m.type | kMatOneColour); r[1] = t.color[0]; r[2] = t.color[1]; r[3] = t.color[2];
        }
        if (m.type == CRT_MAT_DIELECTRIC) {
            auto schlick_r0 = [](float ratio) { float r0 = (1.0f - ratio) / (1.0f + ratio); return r0 * r0; };
            const float inv_ior = 1.0f / m.ior;
            r[4] = inv_ior; r[5] = schlick_r0(inv_ior); r[6] = schlick_r0(m.ior);
        }
    }
    std::vector<float> tex_rec((size_t)d->texture_count * 8, 0.0f);
    std::vector<DevImage> images;
    std::vector<const crt_texture*> image_src;
    for (int i = 0; i < d->texture_count; i++) {
        const crt_texture& t = d->textures[i];
        float* r = &tex_rec[(size_t)i * 8];
        r[0] = as_float(t.type); r[1] = t.color[0]; r[2] = t.color[1]; r[3] = t.color[2];
        r[4] = as_float(t.odd_idx); r[5] = as_float(t.even_idx); r[6] = t.inv_scale; r[7] = as_float(-1);
        if (t.type == CRT_TEX_IMAGE) {
            r[7] = as_float((int)images.size());
            DevImage im = {nullptr, t.width, t.height, t.channels, 0};
            images.push_back(im);
            image_src.push_back(&t);
        }
    }

    // ---- flat mode (small scenes): box slots, see pt_device.h "flat traversal"
    std::vector<float> flat_box_rec, flat_slot_rec, flat_group_xf;
    std::vector<int32_t> flat_ent_obj;
    int flat_groups = 0, flat_box_end[kMaxFlatGroups + 1] = {0, 0, 0}, flat_cuboid_end[kMaxFlatGroups + 1] = {0, 0, 0};
    int flat_cuboid_quad0 = 0, flat_slot_count = 0, flat_far_shift[kMaxFlatGroups + 1] = {-1, -1, -1};
    float flat_far_margin = 0.0f;    // largest margin in use (crt_scene_describe)
    uint32_t flat_sphere_bits = 0, flat_world_face_a[3] = {0, 0, 0}, flat_world_face_b[3] = {0, 0, 0};
    bool flat_slot_xf = false;       // some flat slot's primitives carry their own transform
    // the flat mode tests in no particular order and keeps no box entry distances: not for scenes that have to be walked
    // in the reference's order, nor where near-ties are to be decided like the reference's ordered walk decides them
    bool can_flat = !ordered && !opt.exact_ties && !leaf_slots.empty() && leaf_slots.size() <= 64;
    if (can_flat) {
        struct Slot {
            int group, xf, type, rank_a, rank_b;
            bool has_b;
            float box[6];     // min xyz, max xyz
            float raw[6];     // rotated groups: the same before padding
        };
        std::vector<Slot> slots;
        std::vector<int> group_of_xf;      // rotation transform index -> group number (1..)
        for (size_t rank = 0; rank < leaf_slots.size() && can_flat; rank++) {
            const int ls = leaf_slots[rank];
            const int oi = d->bvh.object_index[ls];
            const crt_object& o = d->objects[oi];
            flat_ent_obj.push_back(oi);
            const int xf = __builtin_bit_cast(int32_t, prim_rec[(size_t)oi * kPrimQuads * 4 + 9]);
            Slot sl;
            sl.group = 0; sl.xf = xf; sl.type = o.type; sl.rank_a = (int)rank; sl.rank_b = 0; sl.has_b = false;
            for (int a = 0; a < 3; a++) { sl.box[a] = d->bvh.min_bounds[3 * ls + a]; sl.box[3 + a] = d->bvh.max_bounds[3 * ls + a]; }
            std::memcpy(sl.raw, sl.box, sizeof(sl.raw));
            const bool rotated = !(identity9(o.rotation) && identity9(o.rotation_inv));
            if (o.type == CRT_OBJ_TRI && xf >= 0 && rotated) {
                // own group per rotation; object-space box of the three vertices, padded
                int g = 0;
                for (size_t k = 0; k < group_of_xf.size(); k++)
                    if (group_of_xf[k] == xf) g = (int)k + 1;
                if (!g) {
                    if ((int)group_of_xf.size() == kMaxFlatGroups) { can_flat = false; break; }
                    group_of_xf.push_back(xf);
                    g = (int)group_of_xf.size();
                }
                sl.group = g;
                sl.xf = -1;
                for (int a = 0; a < 3; a++) {
                    const float v0 = o.p0[a], v1 = o.p0[a] + o.p1[a], v2 = o.p0[a] + o.p2[a];
                    const float lo = std::fmin(v0, std::fmin(v1, v2)), hi = std::fmax(v0, std::fmax(v1, v2));
                    const float pad = 0.001f + 1e-5f * (std::fabs(lo) + std::fabs(hi));
                    sl.box[a] = lo - pad;
                    sl.box[3 + a] = hi + pad;
                    sl.raw[a] = lo;
                    sl.raw[3 + a] = hi;
                }
            }
            bool merged = false;
            if (o.type == CRT_OBJ_TRI) {
                for (Slot& other : slots) {
                    if (!other.has_b && other.type == CRT_OBJ_TRI && other.group == sl.group && other.xf == sl.xf &&
                        !std::memcmp(other.box, sl.box, sizeof(sl.box))) {
                        other.has_b = true;
                        other.rank_b = (int)rank;
                        merged = true;
                        break;
                    }
                }
            }
            if (!merged) slots.push_back(sl);
        }
        can_flat = can_flat && (int)slots.size() <= kMaxFlatSlots;
        if (can_flat) {
            flat_groups = (int)group_of_xf.size();
            std::stable_sort(slots.begin(), slots.end(), [](const Slot& a, const Slot& b) { return a.group < b.group; });
            for (int g : group_of_xf) flat_group_xf.insert(flat_group_xf.end(), &xforms[(size_t)g * kXformFloats], &xforms[(size_t)(g + 1) * kXformFloats]);
            // Cuboids (see flat_cuboid() in pt_device.h): sets of four to six slots of a group whose boxes are
            // [thin interval of one axis] x [the same outer intervals on the other two axes].  A seed slot
            // taken as a face of axis a fixes the outer intervals of the other two axes; every other
            // slot that fits them with some interval of its own on a is the opposite face, and the outer
            // interval of a is whichever value lets the most slots fit as faces of the other axes.
            struct Cuboid {
                int group, faces;
                int face_slot[6];      // xA xB yA yB zA zB, -1 = absent
                float outer[6];        // xlo xhi ylo yhi zlo zhi
            };
            std::vector<Cuboid> cuboids;
            std::vector<char> in_cuboid(slots.size(), 0);
            if (opt.flat_cuboids != 2) {
                auto interval_eq = [](const float* box, int axis, float lo, float hi) { return box[axis] == lo && box[3 + axis] == hi; };
                for (size_t seed = 0; seed < slots.size(); seed++) {
                    if (in_cuboid[seed]) continue;
                    Cuboid best = {};
                    for (int a = 0; a < 3; a++) {
                        const int b = (a + 1) % 3, c = (a + 2) % 3;
                        const float* sb = slots[seed].box;
                        // candidate outer intervals of axis a: what the possible faces of axes b and c have there
                        std::vector<size_t> givers;
                        for (size_t t = 0; t < slots.size(); t++) {
                            if (t == seed || in_cuboid[t] || slots[t].group != slots[seed].group) continue;
                            const float* tb = slots[t].box;
                            const bool fits_b = interval_eq(tb, b, sb[b], sb[3 + b]), fits_c = interval_eq(tb, c, sb[c], sb[3 + c]);
                            if (fits_b != fits_c) givers.push_back(t);
                        }
                        givers.push_back(seed);                                   // stands for "no face on the other axes"
                        for (size_t giver : givers) {
                            Cuboid cu = {};
                            cu.group = slots[seed].group;
                            for (int k = 0; k < 6; k++) cu.face_slot[k] = -1;
                            cu.outer[2 * b] = sb[b]; cu.outer[2 * b + 1] = sb[3 + b];
                            cu.outer[2 * c] = sb[c]; cu.outer[2 * c + 1] = sb[3 + c];
                            cu.outer[2 * a] = slots[giver].box[a]; cu.outer[2 * a + 1] = slots[giver].box[3 + a];
                            for (size_t t = 0; t < slots.size(); t++) {
                                if (in_cuboid[t] || slots[t].group != cu.group) continue;
                                const float* tb = slots[t].box;
                                bool fits[3];
                                for (int x = 0; x < 3; x++) fits[x] = interval_eq(tb, x, cu.outer[2 * x], cu.outer[2 * x + 1]);
                                int thin_axis = -1;
                                if (giver == seed) { if (fits[b] && fits[c]) thin_axis = a; }       // outer interval of a unknown: faces of a only
                                else if (fits[b] && fits[c] && !fits[a]) thin_axis = a;
                                else if (fits[a] && fits[c] && !fits[b]) thin_axis = b;
                                else if (fits[a] && fits[b] && !fits[c]) thin_axis = c;
                                if (thin_axis < 0) continue;
                                int* pair = &cu.face_slot[2 * thin_axis];
                                if (pair[0] < 0) pair[0] = (int)t;
                                else if (pair[1] < 0) pair[1] = (int)t;
                                else continue;                                    // a third parallel face stays an ordinary slot
                                cu.faces++;
                            }
                            if (cu.faces > best.faces) best = cu;
                        }
                    }
                    // A cuboid of a rotated group is laid out on four planes per axis, lo-p < lo+p < hi-p < hi+p
                    // (flat_cuboid_shared()): p at least the padding of every box involved, and the cuboid
                    // thicker than 2p on every axis, else its slots stay as they are.
                    if (best.faces >= 4 && best.group > 0) {
                        float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
                        bool known[3] = {false, false, false}, fits = true;
                        for (int k = 0; k < 6 && fits; k++) {
                            if (best.face_slot[k] < 0) continue;
                            const float* raw = slots[(size_t)best.face_slot[k]].raw;
                            for (int x = 0; x < 3; x++) {
                                if (x == k / 2) continue;                          // its own axis: the face plane, below
                                if (!known[x]) { lo[x] = raw[x]; hi[x] = raw[3 + x]; known[x] = true; }
                                else if (lo[x] != raw[x] || hi[x] != raw[3 + x]) fits = false;
                            }
                        }
                        int placed[6] = {-1, -1, -1, -1, -1, -1};                  // a face lies in the lower (A) or upper (B) plane of its axis
                        for (int k = 0; k < 6 && fits; k++) {
                            if (best.face_slot[k] < 0) continue;
                            const float* raw = slots[(size_t)best.face_slot[k]].raw;
                            const int x = k / 2;
                            int where = -1;
                            if (known[x] && raw[x] == raw[3 + x] && raw[x] == lo[x]) where = 2 * x;
                            else if (known[x] && raw[x] == raw[3 + x] && raw[x] == hi[x]) where = 2 * x + 1;
                            if (where < 0 || placed[where] >= 0) fits = false;
                            else placed[where] = best.face_slot[k];
                        }
                        for (int k = 0; k < 6; k++) best.face_slot[k] = placed[k];
                        for (int x = 0; x < 3 && fits; x++) {
                            const float pad = 0.001f + 2e-5f * std::fmax(std::fabs(lo[x]), std::fabs(hi[x]));
                            if (!(known[x] && hi[x] - lo[x] > 4.0f * pad)) fits = false;
                            best.outer[2 * x] = lo[x]; best.outer[2 * x + 1] = hi[x];   // unpadded; padded when the record is written
                        }
                        if (!fits) best.faces = 0;
                    }
                    // world group: face A of an axis is the lower one, face B the upper one (a lone face: by the side of
                    // the outer interval it lies on), which is what tells a ray's near faces from its far ones
                    if (best.faces >= 4 && best.group == 0) {
                        for (int x = 0; x < 3; x++) {
                            int& fa = best.face_slot[2 * x];
                            int& fb = best.face_slot[2 * x + 1];
                            auto mid = [&](int slot) { return 0.5 * ((double)slots[(size_t)slot].box[x] + slots[(size_t)slot].box[3 + x]); };
                            const double outer_mid = 0.5 * ((double)best.outer[2 * x] + best.outer[2 * x + 1]);
                            if (fa >= 0 && fb >= 0) { if (mid(fa) > mid(fb)) std::swap(fa, fb); }
                            else if (fa >= 0 && mid(fa) > outer_mid) std::swap(fa, fb);
                            else if (fb >= 0 && mid(fb) < outer_mid) std::swap(fa, fb);
                        }
                    }
                    if (best.faces >= 4) {                                        // below that the shared part costs more than it saves
                        for (int k = 0; k < 6; k++) if (best.face_slot[k] >= 0) in_cuboid[(size_t)best.face_slot[k]] = 1;
                        cuboids.push_back(best);
                    }
                }
            }
            // box-test order = candidate-bit order: per group its cuboid faces, then its other slots
            std::vector<Slot> ordered_slots;
            std::vector<float> cuboid_rec;
            int box_end[kMaxFlatGroups + 1] = {0, 0, 0}, cuboid_end[kMaxFlatGroups + 1] = {0, 0, 0};
            int far_cuboid_last_slot[kMaxFlatGroups + 1] = {-1, -1, -1};
            int world_face_slot[6] = {-1, -1, -1, -1, -1, -1};           // faces of the world group's last cuboid, if its far faces wait
            // the world group's largest cuboid (a room) last: the last one is the one whose far faces wait
            std::stable_sort(cuboids.begin(), cuboids.end(), [](const Cuboid& a, const Cuboid& b) {
                auto volume = [](const Cuboid& c) { return (double)(c.outer[1] - c.outer[0]) * (c.outer[3] - c.outer[2]) * (c.outer[5] - c.outer[4]); };
                if (a.group != b.group) return a.group < b.group;
                return a.group == 0 && volume(a) < volume(b);
            });
            // hit_tri's rounding bound per cuboid (see the rotated groups below): 1024 x 20 x 2^-24 x reach / sine of the
            // sharpest angle between two triangle edges, at most an eighth of the cuboid's thinnest side; +inf = not in use
            auto far_margin = [&](const Cuboid& cu, float thinnest) {
                double sine = 1.0;
                for (int k = 0; k < 6; k++) {
                    if (cu.face_slot[k] < 0) continue;
                    const Slot& sl = slots[(size_t)cu.face_slot[k]];
                    if (sl.type != CRT_OBJ_TRI) return INFINITY;
                    for (int which = 0; which < (sl.has_b ? 2 : 1); which++) {
                        const crt_object& t = d->objects[flat_ent_obj[which ? sl.rank_b : sl.rank_a]];
                        const V3 e1 = v3(t.p1), e2 = v3(t.p2);
                        const double area = length(cross(e1, e2)), sides = (double)length(e1) * length(e2);
                        sine = std::fmin(sine, sides > 0.0 ? area / sides : 0.0);
                    }
                }
                const double m = 0x1p-14 * scene_reach(d) / sine;
                if (!(opt.flat_cuboids != 3 && sine > 0.0 && std::isfinite(m) && m <= 0.125 * thinnest)) return INFINITY;
                flat_far_margin = std::fmax(flat_far_margin, (float)m);
                return (float)m;
            };
            int world_cuboids_left = 0;
            for (const Cuboid& cu : cuboids) world_cuboids_left += cu.group == 0;
            for (int g = 0; g <= kMaxFlatGroups; g++) {
                for (const Cuboid& cu : cuboids) {
                    if (cu.group != g) continue;
                    float rec[20] = {0};
                    int present = 0;
                    for (int k = 0; k < 6; k++) {
                        if (cu.face_slot[k] < 0) continue;
                        present |= 1 << k;
                        ordered_slots.push_back(slots[(size_t)cu.face_slot[k]]);
                    }
                    rec[6] = as_float(present);
                    if (g == 0) {                                                 // flat_cuboid(): every face keeps its own interval
                        for (int k = 0; k < 6; k++) rec[k] = cu.outer[k];
                        for (int k = 0; k < 6; k++) {
                            // (an absent face: the plane that would face the ray's origin at infinity, see flat_cuboid())
                            if (cu.face_slot[k] < 0) { rec[8 + 2 * k] = rec[8 + 2 * k + 1] = (k & 1) ? INFINITY : -INFINITY; continue; }
                            const Slot& sl = slots[(size_t)cu.face_slot[k]];
                            rec[8 + 2 * k] = sl.box[k / 2]; rec[8 + 2 * k + 1] = sl.box[3 + k / 2];
                        }
                        // the far faces of the last (largest) cuboid wait for the other candidates, like those of the rotated groups' cuboids
                        rec[7] = INFINITY;
                        if (--world_cuboids_left == 0) {
                            const float thinnest = std::fmin(cu.outer[1] - cu.outer[0], std::fmin(cu.outer[3] - cu.outer[2], cu.outer[5] - cu.outer[4]));
                            rec[7] = far_margin(cu, thinnest);
                            if (std::isfinite(rec[7])) {
                                int at = (int)ordered_slots.size() - cu.faces;
                                for (int k = 0; k < 6; k++) if (cu.face_slot[k] >= 0) world_face_slot[k] = at++;
                            }
                        }
                        cuboid_rec.insert(cuboid_rec.end(), rec, rec + 20);
                    } else {                                                      // flat_cuboid_shared(): four planes per axis
                        float thinnest = INFINITY;
                        for (int x = 0; x < 3; x++) {
                            const float lo = cu.outer[2 * x], hi = cu.outer[2 * x + 1];
                            const float pad = 0.001f + 2e-5f * std::fmax(std::fabs(lo), std::fabs(hi));
                            rec[2 * x] = lo - pad; rec[2 * x + 1] = hi + pad;
                            rec[8 + 2 * x] = lo + pad; rec[8 + 2 * x + 1] = hi - pad;
                            thinnest = std::fmin(thinnest, hi - lo);
                        }
                        // Far faces wait for the near ones (flat_cuboid_shared()) in a group's last cuboid when it has all
                        // six faces.  The margin: hit_tri's t, measured along a face's axis, is off by at most about
                        // 20 x 2^-24 x reach / sin(angle between the triangle's edges), reach = the largest distance
                        // between a ray origin and a point of the scene (far_margin() above).
                        far_cuboid_last_slot[g] = -1;                             // (the group's LAST cuboid is the one: its distance is what the box phase hands on)
                        const float margin = present == 63 ? far_margin(cu, thinnest) : INFINITY;
                        rec[14] = margin;
                        if (std::isfinite(margin)) far_cuboid_last_slot[g] = (int)ordered_slots.size() - 1;   // its face zB
                        cuboid_rec.insert(cuboid_rec.end(), rec, rec + 16);
                    }
                    cuboid_end[g]++;
                }
                for (size_t k = 0; k < slots.size(); k++) {
                    if (slots[k].group != g || in_cuboid[k]) continue;
                    const Slot& sl = slots[k];
                    const float brec[8] = {sl.box[0], sl.box[3], sl.box[1], sl.box[4], sl.box[2], sl.box[5], 0.0f, 0.0f};
                    flat_box_rec.insert(flat_box_rec.end(), brec, brec + 8);
                    ordered_slots.push_back(sl);
                    box_end[g]++;
                }
            }
            for (int g = 1; g <= kMaxFlatGroups; g++) { box_end[g] += box_end[g - 1]; cuboid_end[g] += cuboid_end[g - 1]; }
            for (int g = 0; g <= kMaxFlatGroups; g++) { flat_box_end[g] = box_end[g]; flat_cuboid_end[g] = cuboid_end[g]; }
            flat_cuboid_quad0 = (int)(flat_box_rec.size() / 4);
            flat_box_rec.insert(flat_box_rec.end(), cuboid_rec.begin(), cuboid_rec.end());
            flat_slot_count = (int)ordered_slots.size();
            // candidate bit of slot s = slot_count - 1 - s (the first slot tested ends up highest)
            for (int g = 1; g <= kMaxFlatGroups; g++) {
                const int bit = far_cuboid_last_slot[g] < 0 ? -1 : flat_slot_count - 1 - far_cuboid_last_slot[g];
                flat_far_shift[g] = (bit >= 0 && bit + 6 <= 32 && opt.flat_cuboids != 3) ? bit : -1;
            }
            for (int k = 0; k < 6; k++) {
                const int bit = world_face_slot[k] < 0 ? -1 : flat_slot_count - 1 - world_face_slot[k];
                if (bit >= 0 && bit < 32) (k & 1 ? flat_world_face_b : flat_world_face_a)[k / 2] = 1u << bit;
            }
            for (int s = 0; s < flat_slot_count; s++)
                if (ordered_slots[(size_t)s].type != CRT_OBJ_TRI && flat_slot_count - 1 - s < 32) flat_sphere_bits |= 1u << (flat_slot_count - 1 - s);
            for (const Slot& sl : ordered_slots) {
                float rec[kSlotQuads * 4] = {0};
                const crt_object& A = d->objects[flat_ent_obj[sl.rank_a]];
                const crt_object& B = d->objects[flat_ent_obj[sl.has_b ? sl.rank_b : sl.rank_a]];
                if (sl.type == CRT_OBJ_TRI) {
                    const float* fa[9] = {&A.p0[0], &A.p0[1], &A.p0[2], &A.p1[0], &A.p1[1], &A.p1[2], &A.p2[0], &A.p2[1], &A.p2[2]};
                    const float* fb[9] = {&B.p0[0], &B.p0[1], &B.p0[2], &B.p1[0], &B.p1[1], &B.p1[2], &B.p2[0], &B.p2[1], &B.p2[2]};
                    for (int k = 0; k < 9; k++) { rec[2 + k] = *fa[k]; rec[11 + k] = *fb[k]; }
                } else {
                    rec[2] = A.p0[0]; rec[3] = A.p0[1]; rec[4] = A.p0[2]; rec[5] = A.p2[0];
                    rec[6] = A.p1[0]; rec[7] = A.p1[1]; rec[8] = A.p1[2];
                    // anchor and translation of the sphere's own transform (read when it has no rotation)
                    for (int k = 0; k < 3; k++) { rec[12 + k] = A.anchor[k]; rec[15 + k] = A.translation[k]; }
                }
                const int32_t meta = sl.rank_a | (sl.rank_b << 6) | ((sl.has_b ? 1 : 0) << 12) | (sl.type << 13) | (sl.group << 15);
                rec[0] = as_float(meta);
                rec[1] = as_float(sl.xf);
                if (sl.xf >= 0) flat_slot_xf = true;
                flat_slot_rec.insert(flat_slot_rec.end(), rec, rec + kSlotQuads * 4);
            }
        }
    }

    // the kernels form byte offsets into each of these tables in 32 bits (at32), and texel numbers in an int
    for (size_t words : {node_rec.size(), prim_rec.size(), shade_rec.size(), cold_rec.size(), xforms.size(), mat_rec.size(), tex_rec.size(), flat_ent_obj.size()})
        if (words >= (size_t(1) << 30)) return crt::set_error(CRT_ERR_UNSUPPORTED, "crt_scene_create: a scene table of 4 GiB or more is not supported");
    for (const DevImage& im : images)
        if ((size_t)im.width * im.height >= (size_t(1) << 31)) return crt::set_error(CRT_ERR_UNSUPPORTED, "crt_scene_create: an image of 2^31 pixels or more is not supported");
    crt_scene* sc = new crt_scene();
    struct Guard {                                           // frees the half-built scene on every early exit, thrown or returned
        crt_scene* sc;
        ~Guard() { if (sc) free_scene(sc); }
    } guard{sc};
    if (hipGetDevice(&sc->device) != hipSuccess) return crt::set_error(CRT_ERR_DEVICE, "crt_scene_create: no HIP device");
    if (hipDeviceGetAttribute(&sc->cu_count, hipDeviceAttributeMultiprocessorCount, sc->device) != hipSuccess || sc->cu_count <= 0) sc->cu_count = 256;
#define CRT_HIP_OR_FREE(call)                                                                                          \
    do {                                                                                                               \
        hipError_t e_ = (call);                                                                                        \
        if (e_ != hipSuccess) return crt::set_error(CRT_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_));                     \
    } while (0)
    CRT_HIP_OR_FREE(hipStreamCreateWithFlags(&sc->stream, hipStreamNonBlocking));
    CRT_HIP_OR_FREE(static_scratch_address(sc->device, &sc->static_scratch));
    CRT_HIP_OR_FREE(hipEventCreate(&sc->ev_start));
    CRT_HIP_OR_FREE(hipEventCreate(&sc->ev_mid));
    CRT_HIP_OR_FREE(hipEventCreate(&sc->ev_stop));
    for (size_t k = 0; k < images.size(); k++) {
        void* mem = nullptr;
        size_t bytes = (size_t)images[k].width * images[k].height * images[k].channels;
        CRT_HIP_OR_FREE(hipMalloc(&mem, bytes));
        sc->image_mem.push_back(mem);
        CRT_HIP_OR_FREE(hipMemcpy(mem, image_src[k]->pixels, bytes, hipMemcpyHostToDevice));
        images[k].pixels = static_cast<const uint8_t*>(mem);
    }
    if (images.empty()) images.push_back(DevImage{nullptr, 0, 0, 0, 0});
    Blob blob;
    size_t off_nodes = blob.add(node_rec.data(), node_rec.size() * 4);
    size_t off_prims = blob.add(prim_rec.data(), prim_rec.size() * 4);
    size_t off_shade = blob.add(shade_rec.data(), shade_rec.size() * 4);
    size_t off_cold = blob.add(cold_rec.data(), cold_rec.size() * 4);
    size_t off_xf = blob.add(xforms.data(), xforms.size() * 4);
    size_t off_mats = blob.add(mat_rec.data(), mat_rec.size() * 4);
    size_t off_texs = blob.add(tex_rec.data(), tex_rec.size() * 4);
    size_t off_img = blob.add(images.data(), images.size() * sizeof(DevImage));
    size_t off_fbox = blob.add(flat_box_rec.data(), flat_box_rec.size() * 4);
    size_t off_fslot = blob.add(flat_slot_rec.data(), flat_slot_rec.size() * 4);
    size_t off_fobj = blob.add(flat_ent_obj.data(), flat_ent_obj.size() * 4);
    size_t off_fgx = blob.add(flat_group_xf.data(), flat_group_xf.size() * 4);
    blob.bytes.resize((blob.bytes.size() + 255) & ~size_t(255));
    CRT_HIP_OR_FREE(hipMalloc(&sc->blob, blob.bytes.size()));
    CRT_HIP_OR_FREE(hipMemcpy(sc->blob, blob.bytes.data(), blob.bytes.size(), hipMemcpyHostToDevice));
    // (the copies above went through the null stream, which the scene's own stream does not wait for: everything has landed before the first render)
    CRT_HIP_OR_FREE(hipStreamSynchronize(nullptr));
    CRT_HIP_OR_FREE(hipMalloc((void**)&sc->ray_counter, kCounterBytes + (size_t)2 * kMaxOrderedItems * sizeof(unsigned int)));
    sc->spill_entries = std::max(0, need - (stack_entries - 1));       // one LDS row is the row of -1 below the stacks
    sc->spill_threads = std::max(sc->cu_count * (4 * std::max(std::max(std::max(CRT_FLAT_BASIC_WAVES_PER_SIMD, CRT_FLAT_WAVES_PER_SIMD), std::max(CRT_TREE_WAVES_PER_SIMD, CRT_TREE_BASIC_WAVES_PER_SIMD)), CRT_WF_TRACE_WAVES_PER_SIMD) / kWaves), 1024) * kBlock;
    if (sc->spill_entries > 0) {
        CRT_HIP_OR_FREE(hipMalloc((void**)&sc->stack_spill, (size_t)sc->spill_entries * sc->spill_threads * sizeof(int)));
    }
#undef CRT_HIP_OR_FREE
    uint8_t* base = static_cast<uint8_t*>(sc->blob);
    DevScene& D = sc->dev;
    D.nodes = reinterpret_cast<const float4*>(base + off_nodes);
    D.prims = reinterpret_cast<const float4*>(base + off_prims);
    D.shade = reinterpret_cast<const float4*>(base + off_shade);
    D.shade_cold = reinterpret_cast<const float4*>(base + off_cold);
    D.xforms = reinterpret_cast<const float*>(base + off_xf);
    D.mats = reinterpret_cast<const float4*>(base + off_mats);
    D.texs = reinterpret_cast<const float4*>(base + off_texs);
    D.images = reinterpret_cast<const DevImage*>(base + off_img);
    D.node_count = nodes;
    D.object_count = d->object_count;
    D.xform_count = (int)(xforms.size() / kXformFloats);
    D.material_count = d->material_count;
    D.texture_count = d->texture_count;
    D.background_index = d->background_index;
    D.width = d->render.image_width;
    D.height = d->render.image_height;
    D.max_bounces = d->render.max_bounces;
    D.root_entry = accel_leafy(0) ? 1 : 0;
    D.order_flip = ordered ? 0 : 15;
    D.fast_rcp = fast_reciprocal_is_exact(d) ? 1 : 0;
    if (opt.ieee_division) D.fast_rcp = 0;
    const crt_camera& c = d->camera;
    D.cam.type = c.type;
    D.cam.defocus_angle = c.defocus_angle;
    D.cam.needs_time = 0;
    for (int i = 0; i < d->object_count; i++) {
        const crt_object& o = d->objects[i];
        if (o.type != CRT_OBJ_TRI && !(o.p1[0] == 0.0f && o.p1[1] == 0.0f && o.p1[2] == 0.0f)) D.cam.needs_time = 1;
    }
    D.cam.center = v3(c.center); D.cam.llc = v3(c.lower_left_corner);
    D.cam.du = v3(c.pixel_delta_u); D.cam.dv = v3(c.pixel_delta_v); D.cam.w = v3(c.w);
    D.cam.defocus_u = v3(c.defocus_disk_u); D.cam.defocus_v = v3(c.defocus_disk_v);
    sc->render = d->render;
    sc->stack_entries = stack_entries;
    sc->lds_scene_bytes = (int)(((size_t)nodes * kNodeFields + (size_t)d->object_count * kPrimQuads) * 16 + xforms.size() * 4);
    // keep a block (pass sums + stacks + scene copy) under 40 KB so that four blocks still fit a CU
    sc->can_lds = (size_t)sc->lds_scene_bytes + (size_t)(stack_entries + 8) * kBlock * 4 + (size_t)kWaves * wave_lds_bytes(kModeTreeLds) <= 40 * 1024;
    sc->can_flat = can_flat;
    sc->ordered = ordered;
    {
        int feat = 0;
        if (has_plain_xf) feat |= kFeatPlainXf;
        if (xforms_used) feat |= kFeatXform;
        if (flat_slot_xf || !can_flat) feat |= kFeatSlotXf;
        if (flat_slot_count > 32) feat |= kFeatWide;
        for (int i = 0; i < d->object_count; i++) {
            const crt_object& o = d->objects[i];
            if (o.type != CRT_OBJ_TRI) feat |= kFeatSphere;
            if (o.shade_smooth || o.normal_idx >= 0 || reads_uv(d, o)) feat |= kFeatShading;
        }
        for (int i = 0; i < d->texture_count; i++)
            if (d->textures[i].type != CRT_TEX_SOLID_COLOR) feat |= kFeatShading;
        for (int i = 0; i < d->material_count; i++)
            if (d->materials[i].type == CRT_MAT_METAL || d->materials[i].type == CRT_MAT_DIELECTRIC) feat |= kFeatMaterials;
        if (d->textures[d->background_index].type != CRT_TEX_SOLID_COLOR) feat |= kFeatEnvTexture;
        if (c.type != 0 || c.defocus_angle > 0.0f || D.cam.needs_time) feat |= kFeatCamera;
        if (opt.general_kernel == 1) feat = kFeatAll;
        sc->features = feat;
    }
    sc->flat.boxes = reinterpret_cast<const float4*>(base + off_fbox);
    sc->flat.slots = reinterpret_cast<const float4*>(base + off_fslot);
    sc->flat.ent_obj = reinterpret_cast<const int32_t*>(base + off_fobj);
    sc->flat.group_xf = reinterpret_cast<const float*>(base + off_fgx);
    sc->flat.slot_count = can_flat ? flat_slot_count : 0;
    for (int g = 0; g <= kMaxFlatGroups; g++) { sc->flat.box_end[g] = flat_box_end[g]; sc->flat.cuboid_end[g] = flat_cuboid_end[g]; }
    sc->flat.cuboid_quad0 = flat_cuboid_quad0;
    sc->flat.groups = flat_groups;
    sc->flat.sphere_bits = flat_sphere_bits;
    sc->far_margin = flat_far_margin;
    for (int a = 0; a < 3; a++) { sc->flat.world_face_a[a] = flat_world_face_a[a]; sc->flat.world_face_b[a] = flat_world_face_b[a]; }
    for (int g = 0; g <= kMaxFlatGroups; g++) sc->flat.far_shift[g] = flat_far_shift[g];
    sc->lds_flat_bytes = (int)((size_t)sc->flat.slot_count * kSlotQuads * 16 + xforms.size() * 4);
    sc->mode = opt.mode;
    sc->opt = opt;
    // Chunk sums: sized here for the largest launch this scene can ask for, so that a render
    // allocates nothing.  A launch cuts the frame into about `work_items` items (tile x chunk of
    // the samples), so the number of chunks is bounded by work_items / tiles whatever the sample
    // count; the 8x4-pixel tiling of the tree modes has the larger padded pixel count.
    {
        const ChunkPlan worst = plan_chunks(sc, kModeTreeGlobal, std::max(1, d->render.samples_per_pixel), (size_t)-1);
        const ChunkPlan worst_flat = plan_chunks(sc, kModeFlat, std::max(1, d->render.samples_per_pixel), (size_t)-1);
        sc->partial_bytes = std::max(worst.bytes, worst_flat.bytes);
        hipError_t e = hipMalloc((void**)&sc->partial, sc->partial_bytes);
        if (e == hipSuccess) e = hipHostMalloc((void**)&sc->host_counter, sizeof(unsigned long long), hipHostMallocDefault);
        if (e != hipSuccess) return crt::set_error(CRT_ERR_DEVICE, std::string("crt_scene_create: scratch allocation: ") + hipGetErrorString(e));
        *sc->host_counter = 0;
    }
    // Wavefront pipeline: on request only (measured slower than the fused kernel on every scene of
    // this repository, see the kernel comments).  One batch holds as many whole samples of the
    // image as fit `wavefront_paths` path slots of 224 bytes.
    {
        sc->wavefront = opt.wavefront == 1;
        if (sc->wavefront) {
            const ChunkPlan tiles = plan_chunks(sc, kModeTreeGlobal, 1, (size_t)-1);
            const size_t budget = opt.wavefront_paths > 0 ? (size_t)opt.wavefront_paths : (size_t)64 << 20;
            const size_t samples = std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, d->render.samples_per_pixel), budget / (size_t)tiles.pix_padded));
            const size_t cap = samples * (size_t)tiles.pix_padded;
            const size_t segments = (cap + kWfSegment - 1) / kWfSegment;
            const size_t counters = (size_t)(d->render.max_bounces + 1) * kWfCursors * kWfCursorStride * sizeof(unsigned int);
            const size_t seg_counts = (segments * sizeof(unsigned int) + 255) & ~(size_t)255;
            const size_t slots = segments * kWfSegment;            // queues are whole segments
            const size_t block_rays = (size_t)kWfMaxTraceBlocks * sizeof(unsigned long long);
            const size_t bytes = slots * (2 * (4 * 16 + 8) + 16) + cap * 4 * 16 + counters + 2 * seg_counts + block_rays;
            hipError_t e = hipMalloc(&sc->wf_mem, bytes);
            if (e != hipSuccess) return crt::set_error(CRT_ERR_DEVICE, std::string("crt_scene_create: wavefront buffers (") + std::to_string(bytes >> 20) + " MiB): " + hipGetErrorString(e));
            unsigned char* m = static_cast<unsigned char*>(sc->wf_mem);
            auto take = [&](size_t n) { unsigned char* r = m; m += n; return r; };
            sc->wf_cursors = reinterpret_cast<unsigned int*>(take(counters));
            sc->wf_block_rays = reinterpret_cast<unsigned long long*>(take(block_rays));
            e = hipMemset(sc->wf_block_rays, 0, block_rays);
            if (e == hipSuccess) e = hipStreamSynchronize(nullptr);     // (runs on the null stream; the scene's stream does not wait for it)
            if (e != hipSuccess) return crt::set_error(CRT_ERR_DEVICE, std::string("crt_scene_create: ") + hipGetErrorString(e));
            for (int q = 0; q < 2; q++) {
                sc->wf_queue[q].count = reinterpret_cast<unsigned int*>(take(seg_counts));
                sc->wf_queue[q].ray0 = reinterpret_cast<float4*>(take(slots * 16));
                sc->wf_queue[q].ray1 = reinterpret_cast<float4*>(take(slots * 16));
                sc->wf_queue[q].st0 = reinterpret_cast<float4*>(take(slots * 16));
                sc->wf_queue[q].st1 = reinterpret_cast<float4*>(take(slots * 16));
            }
            sc->wf_hit = reinterpret_cast<float4*>(take(slots * 16));
            sc->wf_res = reinterpret_cast<float4*>(take(cap * 64));
            for (int q = 0; q < 2; q++) sc->wf_queue[q].meta = reinterpret_cast<uint2*>(take(slots * 8));
            sc->wf_capacity = cap;
        }
    }
    guard.sc = nullptr;
    *out = sc;
    return CRT_OK;
}

void crt_scene_destroy(crt_scene* sc) { free_scene(sc); }

int crt_scene_set_mode(crt_scene* sc, int mode) {
    crt::clear_error();
    if (!sc) return crt::set_error(CRT_ERR_INVALID, "crt_scene_set_mode: null scene");
    if (mode < 0 || mode > 3) return crt::set_error(CRT_ERR_INVALID, "crt_scene_set_mode: mode must be 0..3");
    sc->mode = mode;
    return CRT_OK;
}

int crt_scene_describe(crt_scene* sc, char* text, int size) {
    crt::clear_error();
    if (!sc) return crt::set_error(CRT_ERR_INVALID, "crt_scene_describe: null scene");
    static const char* const names[3] = {"tree_hbm", "tree_lds", "flat"};
    char line[320];
    // items=: how the work items of the last launch were ordered, and (measured choice) what the timed launches took
    char items[96];
    if (sc->opt.item_order == 1 || sc->opt.deterministic) std::snprintf(items, sizeof(items), "natural");
    else if (sc->opt.item_order == 2) std::snprintf(items, sizeof(items), "by-cost");
    else if (sc->order_stage < crt_scene::kOrderDecided) std::snprintf(items, sizeof(items), "measuring(%d/%d)", sc->order_stage, crt_scene::kOrderDecided);
    else std::snprintf(items, sizeof(items), "%s(natural %.3f ms, by cost %.3f ms)", sc->by_cost_ms < 0.995f * sc->natural_ms ? "by-cost" : "natural", (double)sc->natural_ms, (double)sc->by_cost_ms);
    // far=: candidate bits of the cuboid faces that are tested last (world group, rotated groups 1 and 2)
    unsigned world_far = 0;
    for (int a = 0; a < 3; a++) world_far |= sc->flat.world_face_a[a] | sc->flat.world_face_b[a];
    const int n = std::snprintf(line, sizeof(line), "mode=%s features=0x%03x slots=%d cuboids=%d groups=%d far=%x/%x/%x margin=%g reciprocal=%s stack=%d+%d nodes=%d objects=%d order=%s items=%s%s", names[kernel_mode(sc)], sc->features,
                                sc->flat.slot_count, sc->flat.cuboid_end[kMaxFlatGroups], sc->flat.groups, world_far, sc->flat.far_shift[1] >= 0 ? 63u << sc->flat.far_shift[1] : 0u,
                                sc->flat.far_shift[2] >= 0 ? 63u << sc->flat.far_shift[2] : 0u, (double)sc->far_margin, sc->dev.fast_rcp ? "newton" : "ieee", sc->stack_entries, sc->spill_entries,
                                sc->dev.node_count, sc->dev.object_count, sc->ordered ? "reference" : "free", items,
                                sc->wavefront && kernel_mode(sc) == kModeTreeGlobal ? " pipeline=wavefront" : "");
    if (text && size > 0) std::snprintf(text, (size_t)size, "%s", line);
    return n;
}

void* crt_scene_stream(crt_scene* sc) { return sc ? (void*)sc->stream : nullptr; }

int crt_buffers_create(int width, int height, crt_buffers** out) {
    crt::clear_error();
    if (!out || width <= 0 || height <= 0) return crt::set_error(CRT_ERR_INVALID, "crt_buffers_create: bad arguments");
    void* mem = nullptr;
    size_t bytes = (size_t)kPasses * width * height * sizeof(float);
    CRT_HIP(hipMalloc(&mem, bytes));
    // (a memset of device memory returns before it has run, and it runs on the null stream, which the scenes' own
    // non-blocking streams do not wait for: without the wait a render into the new buffers can be overtaken by
    // their zeros - seen with eight processes on one GPU, whose queues take turns on the hardware)
    hipError_t e = hipMemset(mem, 0, bytes);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) { (void)hipFree(mem); return crt::set_error(CRT_ERR_DEVICE, std::string("hipMemset: ") + hipGetErrorString(e)); }
    crt_buffers* b = new crt_buffers();
    b->width = width; b->height = height; b->base = static_cast<float*>(mem); b->owned = true;
    *out = b;
    return CRT_OK;
}

int crt_buffers_wrap(int width, int height, void* device_ptr, crt_buffers** out) {
    crt::clear_error();
    if (!out || !device_ptr || width <= 0 || height <= 0) return crt::set_error(CRT_ERR_INVALID, "crt_buffers_wrap: bad arguments");
    if (reinterpret_cast<uintptr_t>(device_ptr) % 16) return crt::set_error(CRT_ERR_INVALID, "crt_buffers_wrap: pointer must be 16-byte aligned");
    crt_buffers* b = new crt_buffers();
    b->width = width; b->height = height; b->base = static_cast<float*>(device_ptr); b->owned = false;
    *out = b;
    return CRT_OK;
}

void crt_buffers_destroy(crt_buffers* b) {
    if (!b) return;
    if (b->owned && b->base) (void)hipFree(b->base);
    delete b;
}

int crt_buffers_download(const crt_buffers* b, float* color, float* albedo, float* emission, float* normal, float* depth, float* opacity) {
    crt::clear_error();
    if (!b) return crt::set_error(CRT_ERR_INVALID, "crt_buffers_download: null buffers");
    const size_t n = (size_t)b->width * b->height;
    CRT_HIP(hipDeviceSynchronize());
    if (color) CRT_HIP(hipMemcpy(color, b->color(), n * 12, hipMemcpyDeviceToHost));
    if (albedo) CRT_HIP(hipMemcpy(albedo, b->albedo(), n * 12, hipMemcpyDeviceToHost));
    if (emission) CRT_HIP(hipMemcpy(emission, b->emission(), n * 12, hipMemcpyDeviceToHost));
    if (normal) CRT_HIP(hipMemcpy(normal, b->normal(), n * 12, hipMemcpyDeviceToHost));
    if (depth) CRT_HIP(hipMemcpy(depth, b->depth(), n * 4, hipMemcpyDeviceToHost));
    if (opacity) CRT_HIP(hipMemcpy(opacity, b->opacity(), n * 4, hipMemcpyDeviceToHost));
    return CRT_OK;
}

int crt_buffers_upload(crt_buffers* b, const float* color, const float* albedo, const float* emission, const float* normal, const float* depth, const float* opacity) {
    crt::clear_error();
    if (!b) return crt::set_error(CRT_ERR_INVALID, "crt_buffers_upload: null buffers");
    const size_t n = (size_t)b->width * b->height;
    CRT_HIP(hipDeviceSynchronize());
    if (color) CRT_HIP(hipMemcpy(b->color(), color, n * 12, hipMemcpyHostToDevice));
    if (albedo) CRT_HIP(hipMemcpy(b->albedo(), albedo, n * 12, hipMemcpyHostToDevice));
    if (emission) CRT_HIP(hipMemcpy(b->emission(), emission, n * 12, hipMemcpyHostToDevice));
    if (normal) CRT_HIP(hipMemcpy(b->normal(), normal, n * 12, hipMemcpyHostToDevice));
    if (depth) CRT_HIP(hipMemcpy(b->depth(), depth, n * 4, hipMemcpyHostToDevice));
    if (opacity) CRT_HIP(hipMemcpy(b->opacity(), opacity, n * 4, hipMemcpyHostToDevice));
    CRT_HIP(hipStreamSynchronize(nullptr));      // (a progressive render folds into these on a stream that does not wait for the null stream)
    return CRT_OK;
}

// Where a launch keeps its chunk sums and counters.
struct ScratchView {
    unsigned long long* counters;    // kCounterBytes: rays traced, then the work-item counters
    float* partial;
    size_t partial_bytes;
    bool keeps_order;                // the scene's own scratch: item costs are recorded and order the next launch
};

static int launch_render(crt_scene* sc, crt_buffers* buf, int first_sample, int samples, int fold_done, const ScratchView& S, const char* who,
                         Bands bands = {0, 0, 1}) {
    if (!sc || !buf) return crt::set_error(CRT_ERR_INVALID, std::string(who) + ": null argument");
    if (bands.count < 1 || bands.index < 0 || bands.index >= bands.count || (bands.count > 1 && (bands.rows <= 0 || bands.rows % 8 != 0)))
        return crt::set_error(CRT_ERR_INVALID, std::string(who) + ": band_rows must be a positive multiple of 8 and 0 <= band_index < band_count");
    const int own_rows = band_row_count(sc->dev.height, bands);
    if (buf->width != sc->dev.width || buf->height != own_rows)
        return crt::set_error(CRT_ERR_INVALID, std::string(who) + ": buffer size does not match the scene" + (bands.count > 1 ? " (a band's buffers hold its own " + std::to_string(own_rows) + " rows)" : ""));
    if (own_rows == 0) return crt::set_error(CRT_ERR_INVALID, std::string(who) + ": this band owns no rows of the image");
    if (samples <= 0 || first_sample < 0) return crt::set_error(CRT_ERR_INVALID, std::string(who) + ": samples must be positive");
    if (sc->launched) return crt::set_error(CRT_ERR_INVALID, std::string(who) + ": a render is already in flight on this scene");
    CRT_HIP(hipSetDevice(sc->device));
    const ChunkPlan plan = plan_chunks(sc, kernel_mode(sc), samples, S.partial_bytes, bands);
    if (!plan.fits)
        return crt::set_error(CRT_ERR_UNSUPPORTED, std::string(who) + ": the chunk sums of this launch (" + std::to_string(sc->opt.deterministic ? (size_t)samples * kPasses * plan.pix_padded * sizeof(float) : (size_t)kPasses * plan.pix_padded * sizeof(float)) +
                                                       " bytes) exceed the scratch available (" + std::to_string(S.partial_bytes) + " bytes)");
    RenderArgs A;
    A.sc = sc->dev;
    A.tiles_x = plan.tiles_x;
    A.tiles_y = plan.tiles_y;
    A.ntiles = plan.ntiles;
    A.pix_padded = plan.pix_padded;
    A.chunk_size = plan.chunk_size;
    A.chunks = plan.chunks;
    A.first_sample = first_sample;
    A.samples = samples;
    A.flat = sc->flat;
    A.partial = S.partial;
    A.stack_spill = sc->stack_spill;
    A.spill_threads = sc->spill_threads;
    // Scheduling of the tree walk (measured sweeps, profiles/r2_e_sweep*).  A hierarchy that fits an XCD's 4 MB L2
    // runs best with primitive tests in batches of 8 lanes and shading once 65 % of the wave's walks have ended (the
    // 976-triangle mesh: 10.3 ms; 10.5-11.7 elsewhere in the sweep).  One that does not fit waits on memory in every
    // step and shades earlier, at 45-50 %; its best batch depends on how long a wave stays on one pixel tile: with many
    // samples per work item the lanes' rays are close to each other, reach leaves together and fill batches of 16
    // (1 M triangles at 128 spp: 390 ms against 403 with the small-scene setting), with few samples per item
    // primitives are best tested as soon as two lanes hold some (the same scene at 8 spp: 30.7 ms against 32.5).
    const bool beyond_l2 = (size_t)sc->dev.node_count * kNodeStrideGlobal * 16 > ((size_t)4 << 20);
    const bool long_items = plan.chunk_size >= 32;
    A.prim_batch = sc->opt.prim_batch > 0 ? sc->opt.prim_batch : (beyond_l2 ? (long_items ? 16 : 2) : 8);
    A.ready_percent = sc->opt.ready_percent > 0 ? sc->opt.ready_percent : (beyond_l2 ? 45 : 65);
    A.band_tiles = plan.band_tiles;
    A.band_index = bands.index;
    A.band_count = bands.count;
    A.ray_counter = S.counters;
    A.queue = reinterpret_cast<unsigned int*>(S.counters) + kQueueStride;
    CRT_HIP(hipMemsetAsync(S.counters, 0, kCounterBytes, sc->stream));
    // Launch order: the items of this cut of the frame, most expensive first, as the last launch of the same cut
    // measured them (the first launch of a cut runs in the natural order, tile after tile and chunk after chunk).
    const long long items_ll = (long long)plan.ntiles * plan.chunks;
    const int order_key[6] = {plan.ntiles, plan.chunks, plan.chunk_size, bands.rows, bands.index, bands.count};
    const bool fused = !(sc->wavefront && kernel_mode(sc) == kModeTreeGlobal);
    const bool may_order = S.keeps_order && fused && sc->opt.item_order != 1 && !sc->opt.deterministic && items_ll <= kMaxOrderedItems;
    const bool same_cut = sc->order_ready && std::equal(order_key, order_key + 6, sc->order_key);
    if (!same_cut) sc->order_stage = 0;
    // (by cost only where that is at least 0.5 % faster: sorting the costs takes 10 to 15 us of every launch)
    const bool decided = sc->opt.item_order == 0 && sc->order_stage >= crt_scene::kOrderDecided;
    const bool by_cost = sc->opt.item_order == 2 || (!decided && (sc->order_stage & 1)) || (decided && sc->by_cost_ms < 0.995f * sc->natural_ms);
    // (once the natural order has won, the item costs are no longer recorded or sorted)
    const bool keeps_order = may_order && !(decided && !by_cost);
    A.order_mode = !keeps_order ? 0 : (same_cut && by_cost ? 2 : 1);
    sc->launched_stage = keeps_order ? sc->order_stage : -1;
    // persistent waves: as many blocks as the chip holds at once (or fewer when there is less work)
    const long long items = (long long)plan.ntiles * plan.chunks;
    int resident = sc->cu_count * (4 * waves_per_simd(kernel_mode(sc), variant_features(sc)) / kWaves);   // 4 SIMDs per CU
    if (sc->opt.blocks > 0) resident = std::min(sc->spill_threads / kBlock, sc->opt.blocks);
    else if (sc->opt.chip_parts > 1) resident = std::max(1, resident / sc->opt.chip_parts);
    dim3 grid((unsigned)std::min<long long>((items + kWaves - 1) / kWaves, resident));
    CRT_HIP(hipEventRecord(sc->ev_start, sc->stream));
    const bool wavefront = sc->wavefront && kernel_mode(sc) == kModeTreeGlobal;
    int rc = wavefront ? launch_wavefront(sc, A, plan, bands) : launch_fused(sc, A, grid);
    if (rc != CRT_OK) return rc;
    CRT_HIP(hipEventRecord(sc->ev_mid, sc->stream));
    if (keeps_order) {
        const int n = (int)items_ll, per_block = kBlock * kOrderItemsPerThread;
        unsigned int* words = reinterpret_cast<unsigned int*>(S.counters);
        hipLaunchKernelGGL(count_cost_classes, dim3((n + per_block - 1) / per_block), dim3(kBlock), 0, sc->stream, (const unsigned int*)(A.queue + kItemCostFromQueue), n, words + kCostWordsAt);
        hipLaunchKernelGGL(order_by_cost, dim3((n + per_block - 1) / per_block), dim3(kBlock), 0, sc->stream, (const unsigned int*)(A.queue + kItemCostFromQueue), n,
                           (const unsigned int*)(words + kCostWordsAt), words + kCostWordsAt + kCostClasses, A.queue + kItemOrderFromQueue);
        CRT_HIP(hipGetLastError());
        std::copy(order_key, order_key + 6, sc->order_key);
        sc->order_ready = true;
        if (sc->order_stage < crt_scene::kOrderDecided) sc->order_stage++;
    } else if (S.keeps_order && !may_order) {
        sc->order_ready = false;
        sc->order_stage = 0;
    }
    ResolveArgs R;
    R.partial = S.partial;
    R.color = buf->color(); R.albedo = buf->albedo(); R.emission = buf->emission(); R.normal = buf->normal();
    R.depth = buf->depth(); R.opacity = buf->opacity();
    R.width = sc->dev.width; R.height = sc->dev.height; R.tiles_x = plan.tiles_x; R.tile_rows = plan.tile_h; R.ntiles = plan.ntiles; R.pix_padded = plan.pix_padded;
    R.chunks = wavefront ? 1 : plan.chunks;
    R.done = (float)fold_done;
    R.count = (float)samples;
    R.fold = fold_done > 0 ? 1 : 0;
    R.band_tiles = plan.band_tiles; R.band_index = bands.index; R.band_count = bands.count;
    hipLaunchKernelGGL(resolve_passes, dim3((plan.pix_padded + kBlock - 1) / kBlock), dim3(kBlock), 0, sc->stream, R);
    CRT_HIP(hipGetLastError());
    CRT_HIP(hipEventRecord(sc->ev_stop, sc->stream));
    // the ray count travels to pinned host memory behind the kernels; crt_render_finish only waits
    CRT_HIP(hipMemcpyAsync(sc->host_counter, S.counters, sizeof(unsigned long long), hipMemcpyDeviceToHost, sc->stream));
    sc->launched = true;
    return CRT_OK;
}

int crt_render_launch(crt_scene* sc, crt_buffers* buf, int first_sample, int samples, int fold_done) {
    crt::clear_error();
    if (!sc) return crt::set_error(CRT_ERR_INVALID, "crt_render_launch: null argument");
    const ScratchView S = {sc->ray_counter, sc->partial, sc->partial_bytes, true};
    return launch_render(sc, buf, first_sample, samples, fold_done, S, "crt_render_launch");
}

int crt_band_row_count(int height, int band_rows, int band_index, int band_count) {
    crt::clear_error();
    if (height < 0 || band_count < 1 || band_index < 0 || band_index >= band_count || (band_count > 1 && (band_rows <= 0 || band_rows % 8 != 0)))
        return crt::set_error(CRT_ERR_INVALID, "crt_band_row_count: band_rows must be a positive multiple of 8 and 0 <= band_index < band_count");
    return band_row_count(height, Bands{band_rows, band_index, band_count});
}

int crt_render_bands_launch(crt_scene* sc, crt_buffers* own, int first_sample, int samples, int band_rows, int band_index, int band_count) {
    crt::clear_error();
    if (!sc) return crt::set_error(CRT_ERR_INVALID, "crt_render_bands_launch: null argument");
    const ScratchView S = {sc->ray_counter, sc->partial, sc->partial_bytes, true};
    return launch_render(sc, own, first_sample, samples, 0, S, "crt_render_bands_launch", Bands{band_rows, band_index, band_count});
}

int crt_render_bands(crt_scene* sc, crt_buffers* own, int first_sample, int samples, int band_rows, int band_index, int band_count,
                     float* seconds_out, uint64_t* rays_out) {
    int rc = crt_render_bands_launch(sc, own, first_sample, samples, band_rows, band_index, band_count);
    if (rc != CRT_OK) return rc;
    float ms = 0.0f;
    rc = crt_render_finish(sc, &ms, rays_out);
    if (seconds_out) *seconds_out = ms * 1e-3f;
    return rc;
}

int crt_buffers_assemble_bands(crt_buffers* full, const float* gathered, size_t rank_stride_floats, int band_rows, int band_count, void* hip_stream) {
    crt::clear_error();
    if (!full || !gathered) return crt::set_error(CRT_ERR_INVALID, "crt_buffers_assemble_bands: null argument");
    if (band_count < 1 || band_rows <= 0 || band_rows % 8 != 0) return crt::set_error(CRT_ERR_INVALID, "crt_buffers_assemble_bands: band_rows must be a positive multiple of 8");
    size_t largest = 0;
    for (int r = 0; r < band_count; r++) largest = std::max(largest, (size_t)band_row_count(full->height, Bands{band_rows, r, band_count}));
    if (rank_stride_floats < largest * full->width * kPasses) return crt::set_error(CRT_ERR_INVALID, "crt_buffers_assemble_bands: rank stride smaller than the largest band");
    AssembleArgs A;
    A.gathered = gathered; A.stride = rank_stride_floats;
    A.color = full->color(); A.albedo = full->albedo(); A.emission = full->emission(); A.normal = full->normal();
    A.depth = full->depth(); A.opacity = full->opacity();
    A.width = full->width; A.height = full->height; A.band_rows = band_rows; A.band_count = band_count;
    const size_t n = (size_t)full->width * full->height;
    hipLaunchKernelGGL(assemble_bands, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, static_cast<hipStream_t>(hip_stream), A);
    CRT_HIP(hipGetLastError());
    return CRT_OK;
}

int crt_render_finish(crt_scene* sc, float* kernel_ms_out, uint64_t* rays_out) {
    crt::clear_error();
    if (!sc) return crt::set_error(CRT_ERR_INVALID, "crt_render_finish: null scene");
    if (!sc->launched) return crt::set_error(CRT_ERR_INVALID, "crt_render_finish: nothing was launched");
    sc->launched = false;
    CRT_HIP(hipStreamSynchronize(sc->stream));
    if (kernel_ms_out) {
        float ms = 0.0f;
        CRT_HIP(hipEventElapsedTime(&ms, sc->ev_start, sc->ev_stop));
        *kernel_ms_out = ms;
    }
    CRT_HIP(hipEventElapsedTime(&sc->last_trace_ms, sc->ev_start, sc->ev_mid));
    CRT_HIP(hipEventElapsedTime(&sc->last_resolve_ms, sc->ev_mid, sc->ev_stop));
    if (sc->launched_stage == 2) sc->natural_ms = sc->last_trace_ms;
    if (sc->launched_stage == 3) sc->by_cost_ms = sc->last_trace_ms;
    if (sc->launched_stage == 4) sc->natural_ms = std::min(sc->natural_ms, sc->last_trace_ms);
    if (sc->launched_stage == 5) sc->by_cost_ms = std::min(sc->by_cost_ms, sc->last_trace_ms);
    if (rays_out) *rays_out = *sc->host_counter;
    return CRT_OK;
}

int crt_render_timings(crt_scene* sc, float* trace_ms, float* resolve_ms) {
    crt::clear_error();
    if (!sc) return crt::set_error(CRT_ERR_INVALID, "crt_render_timings: null scene");
    if (trace_ms) *trace_ms = sc->last_trace_ms;
    if (resolve_ms) *resolve_ms = sc->last_resolve_ms;
    return CRT_OK;
}

int crt_profile_counters(uint64_t* out64, int reset) {
    crt::clear_error();
#ifdef CRT_PROFILE
    unsigned long long host[128];
    if (out64) {
        CRT_HIP(hipDeviceSynchronize());
        CRT_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(crt::g_prof), sizeof(host)));
        for (int i = 0; i < 128; i++) out64[i] = host[i];
    }
    if (reset) {
        std::memset(host, 0, sizeof(host));
        CRT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(crt::g_prof), host, sizeof(host)));
    }
    return CRT_OK;
#else
    (void)out64; (void)reset;
    return crt::set_error(CRT_ERR_UNSUPPORTED, "crt_profile_counters: this library was built without -DCRT_PROFILE");
#endif
}

int crt_render_range(crt_scene* sc, crt_buffers* buf, int first_sample, int samples, float* seconds_out, uint64_t* rays_out) {
    int rc = crt_render_launch(sc, buf, first_sample, samples, 0);
    if (rc != CRT_OK) return rc;
    float ms = 0.0f;
    rc = crt_render_finish(sc, &ms, rays_out);
    if (seconds_out) *seconds_out = ms * 1e-3f;
    return rc;
}

int crt_render(crt_scene* sc, crt_buffers* buf, int samples, int samples_done, float* seconds_out, uint64_t* rays_out) {
    crt::clear_error();
    if (!sc || !buf) return crt::set_error(CRT_ERR_INVALID, "crt_render: null argument");
    if (samples_done < 0) return crt::set_error(CRT_ERR_INVALID, "crt_render: samples_done must be >= 0");
    if (samples < 0) samples = sc->render.samples_per_pixel;                       // render.cuh:120-124
    else samples = std::min(samples, sc->render.samples_per_pixel);
    if (samples == 0) {
        if (seconds_out) *seconds_out = 0.0f;
        if (rays_out) *rays_out = 0;
        return CRT_OK;
    }
    int rc = crt_render_launch(sc, buf, samples_done, samples, samples_done);
    if (rc != CRT_OK) return rc;
    float ms = 0.0f;
    rc = crt_render_finish(sc, &ms, rays_out);
    if (seconds_out) *seconds_out = ms * 1e-3f;
    return rc;
}

int crt_render_static(crt_scene* sc, crt_buffers* buf, int samples, int samples_done, float* seconds_out, uint64_t* rays_out) {
    crt::clear_error();
    if (!sc || !buf) return crt::set_error(CRT_ERR_INVALID, "crt_render_static: null argument");
    if (samples_done < 0) return crt::set_error(CRT_ERR_INVALID, "crt_render_static: samples_done must be >= 0");
    if (samples < 0) samples = sc->render.samples_per_pixel;                       // render.cuh:120-124
    else samples = std::min(samples, sc->render.samples_per_pixel);
    if (samples == 0) {
        if (seconds_out) *seconds_out = 0.0f;
        if (rays_out) *rays_out = 0;
        return CRT_OK;
    }
    if (!sc->static_scratch) {                              // (set at scene creation; kept for safety)
        CRT_HIP(hipSetDevice(sc->device));
        CRT_HIP(static_scratch_address(sc->device, &sc->static_scratch));
    }
    // Every call on a GPU shares that GPU's array: a second call while one is running (another thread, another
    // scene) is refused instead of mixing its sums into the first one's.
    if (sc->device < 0 || sc->device >= kMaxDevices) return crt::set_error(CRT_ERR_UNSUPPORTED, "crt_render_static: device number out of range");
    struct Busy {
        std::atomic<int>& flag;
        bool mine;
        explicit Busy(std::atomic<int>& f) : flag(f), mine(f.exchange(1) == 0) {}
        ~Busy() { if (mine) flag.store(0); }
    } busy(g_static_busy[sc->device]);
    if (!busy.mine) return crt::set_error(CRT_ERR_INVALID, "crt_render_static: another crt_render_static call is running on this GPU (they share one scratch array)");
    // launch_render() cuts the frame into as many sample chunks as fit the array and refuses an
    // image whose sums of a single chunk (56 bytes per padded pixel) do not
    const ScratchView S = {reinterpret_cast<unsigned long long*>(sc->static_scratch),
                           reinterpret_cast<float*>(sc->static_scratch + kStaticCounterBytes), kStaticScratchBytes - kStaticCounterBytes, false};
    int rc = launch_render(sc, buf, samples_done, samples, samples_done, S, "crt_render_static");
    if (rc != CRT_OK) return rc;
    float ms = 0.0f;
    rc = crt_render_finish(sc, &ms, rays_out);
    if (seconds_out) *seconds_out = ms * 1e-3f;
    return rc;
}

int crt_probe(crt_scene* sc, int kind, int count, const float* in, float* out) {
    crt::clear_error();
    static const int in_floats[4] = {14, 10, 6, 3}, out_floats[4] = {1, 11, 3, 7};
    if (!sc || !in || !out) return crt::set_error(CRT_ERR_INVALID, "crt_probe: null argument");
    if (kind < 0 || kind > 3 || count < 0) return crt::set_error(CRT_ERR_INVALID, "crt_probe: kind must be 0..3 and count >= 0");
    if (count == 0) return CRT_OK;
    CRT_HIP(hipSetDevice(sc->device));
    const size_t in_bytes = (size_t)count * in_floats[kind] * sizeof(float), out_bytes = (size_t)count * out_floats[kind] * sizeof(float);
    void* mem = nullptr;
    CRT_HIP(hipMalloc(&mem, in_bytes + out_bytes));
    ProbeArgs A;
    A.sc = sc->dev; A.kind = kind; A.count = count;
    A.in = static_cast<const float*>(mem);
    A.out = reinterpret_cast<float*>(static_cast<unsigned char*>(mem) + in_bytes);
    hipError_t e = hipMemcpy(mem, in, in_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(probe_kernel, dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, sc->stream, A);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(sc->stream);
    if (e == hipSuccess) e = hipMemcpy(out, A.out, out_bytes, hipMemcpyDeviceToHost);
    (void)hipFree(mem);
    if (e != hipSuccess) return crt::set_error(CRT_ERR_DEVICE, std::string("crt_probe: ") + hipGetErrorString(e));
    return CRT_OK;
}

int crt_trace_primary(crt_scene* sc, int sample, int32_t* object_index, float* t, float* u, float* v, int32_t* front_face) {
    crt::clear_error();
    if (!sc) return crt::set_error(CRT_ERR_INVALID, "crt_trace_primary: null scene");
    if (sample < 1) return crt::set_error(CRT_ERR_INVALID, "crt_trace_primary: sample numbers start at 1");
    CRT_HIP(hipSetDevice(sc->device));
    const size_t n = (size_t)sc->dev.width * sc->dev.height;
    void* mem = nullptr;
    CRT_HIP(hipMalloc(&mem, n * 20));
    PrimaryArgs A;
    A.sc = sc->dev;
    A.sample = sample;
    A.obj = static_cast<int32_t*>(mem);
    A.t = reinterpret_cast<float*>(A.obj + n);
    A.u = A.t + n;
    A.v = A.u + n;
    A.front = reinterpret_cast<int32_t*>(A.v + n);
    dim3 grid((unsigned)std::min<size_t>((n + kBlock - 1) / kBlock, (size_t)sc->spill_threads / kBlock));
    A.flat = sc->flat;
    A.stack_spill = sc->stack_spill;
    A.spill_threads = sc->spill_threads;
    if (kernel_mode(sc) == kModeFlat) {
        const size_t lds = (size_t)sc->lds_flat_bytes;
        switch (sc->flat.groups) {
        case 0: hipLaunchKernelGGL((trace_primary_kernel<kModeFlat, 0, 0>), grid, dim3(kBlock), lds, sc->stream, A); break;
        case 1: hipLaunchKernelGGL((trace_primary_kernel<kModeFlat, 0, 1>), grid, dim3(kBlock), lds, sc->stream, A); break;
        default: hipLaunchKernelGGL((trace_primary_kernel<kModeFlat, 0, 2>), grid, dim3(kBlock), lds, sc->stream, A); break;
        }
    } else {
        if (sc->stack_entries == 8) hipLaunchKernelGGL((trace_primary_kernel<kModeTreeGlobal, 8, 0>), grid, dim3(kBlock), (8 + 8) * kBlock * 4, sc->stream, A);
        else hipLaunchKernelGGL((trace_primary_kernel<kModeTreeGlobal, kDeepStackRows, 0>), grid, dim3(kBlock), (kDeepStackRows + 8) * kBlock * 4, sc->stream, A);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(sc->stream);
    if (e == hipSuccess && object_index) e = hipMemcpy(object_index, A.obj, n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && t) e = hipMemcpy(t, A.t, n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && u) e = hipMemcpy(u, A.u, n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && v) e = hipMemcpy(v, A.v, n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && front_face) e = hipMemcpy(front_face, A.front, n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(mem);
    if (e != hipSuccess) return crt::set_error(CRT_ERR_DEVICE, std::string("crt_trace_primary: ") + hipGetErrorString(e));
    return CRT_OK;
}

}  // extern "C"
