// Device-side path-tracing building blocks for gfx950 (wave64).
//
// These functions restate, for a flat SoA scene, what the reference does on the device:
//   camera rays        src/camera.cuh:240-278
//   BVH4 traversal     src/bvh.cuh:380-421 with the slab test of src/aabb.cuh:132-156
//   primitive tests    src/hittable.cuh:272-359 under the object transform of :426-446
//   hit record fill    src/hittable.cuh:292-299, :338-356, :367-386, :438-444
//   materials          src/material.cuh:290-381, textures src/texture.cuh:252-324
//   path loop          src/render.cuh:36-98
//
// Design notes
//   * The reference fills a complete hit record at every accepted candidate.  Only the last
//     accepted one survives, so traversal here carries just (t, object, barycentrics) and
//     the record is derived once per ray afterwards from the same expressions.
//   * The slab test's reciprocal depends only on the ray, so it is hoisted out of the box
//     loop; the near/far plane of every axis is picked by the per-lane ADDRESS of the load
//     (min or max field) instead of a swap, which is value-identical to aabb.cuh:144-148.
//   * Objects whose transform is the identity with zero anchor and translation skip the
//     two 3x3 products: (o - 0 - 0) + 0 and I*v are exact, so results do not change.
//   * All float expressions keep the reference's association order and the file is built
//     with -ffp-contract=off; acos/atan2/sin/cos are the plain-float evaluations of det_math.h
//     and (1-cos)^5 is a multiplication chain: with the shared counter RNG every path is
//     reproducible bit for bit on a CPU that evaluates the same expressions.
#ifndef CRT_PT_DEVICE_H
#define CRT_PT_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "det_math.h"
#include "vecmath.h"

namespace crt {

// Profiling builds (never the product build):
//   -DCRT_PROFILE                      wave-level iteration counts and active-lane sums of the kernel's
//                                      loops in a device array (slot 2k = times a wave executed site k,
//                                      slot 2k+1 = lanes active there);
//   -DCRT_PROFILE -DCRT_PROFILE_CLOCK  instead, the shader-clock time of the waves per section of the
//                                      path loop (slots 32..39); the counters' atomics would dominate
//                                      those times, so the two are separate builds.
// The lanes of the wave on which `p` holds, as a 64-bit mask in scalar registers.  (HIP's __ballot takes an
// int: the predicate is first turned into 0 / 1 in a vector register and compared with zero again, two
// instructions more per use.)
__device__ __forceinline__ unsigned long long lanes_where(bool p) { return __builtin_amdgcn_ballot_w64(p); }

#ifdef CRT_PROFILE
__device__ unsigned long long g_prof[128];
#endif
#if defined(CRT_PROFILE) && !defined(CRT_PROFILE_CLOCK)
__device__ __forceinline__ void prof_tick(int site) {
    unsigned long long m = lanes_where(true);
    if ((int)(threadIdx.x & 63) == __builtin_ctzll(m)) {
        atomicAdd(&g_prof[2 * site], 1ull);
        atomicAdd(&g_prof[2 * site + 1], (unsigned long long)__builtin_popcountll(m));
    }
}
#define CRT_PROF(site) prof_tick(site)
#else
#define CRT_PROF(site) ((void)0)
#endif
#ifdef CRT_PROFILE_CLOCK
struct SectionClock {
    unsigned long long last, acc[8], began;
    int cur;
    __device__ __forceinline__ void start() {
        began = wall_clock64();        // (100 MHz, one counter for the whole chip)
        for (int k = 0; k < 8; k++) acc[k] = 0;
        cur = 0;
        float dep = 0.0f;
        last = 0;
        mark(0, dep);
        acc[0] = 0;
    }
    // `dep` is a value the section that ends here produced: the clock read is ordered after it.
    template <typename T>
    __device__ __forceinline__ void mark(int section, T& dep) {
        unsigned long long now;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now), "+v"(dep) : : "memory");
#pragma unroll
        for (int k = 0; k < 8; k++) if (k == cur) acc[k] += now - last;
        last = now;
        cur = section;
    }
    __device__ __forceinline__ void flush() {
        float dep = 0.0f;
        mark(0, dep);
        if ((threadIdx.x & 63) == 0) {
            for (int k = 0; k < 8; k++) atomicAdd(&g_prof[32 + k], acc[k]);
            // when the waves began and ended: earliest start (kept as the largest complement), latest end, sum of ends, waves
            const unsigned long long ended = wall_clock64();
            atomicMax(&g_prof[40], ~began);
            atomicMax(&g_prof[41], ended);
            atomicAdd(&g_prof[42], ended - began);
            atomicAdd(&g_prof[43], 1ull);
            atomicMax(&g_prof[44], began);
        }
    }
};
#define CRT_CLOCK(clk, section, dep) (clk).mark(section, dep)
#else
struct SectionClock {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void flush() {}
};
#define CRT_CLOCK(clk, section, dep) ((void)0)
#endif
enum { kSecOther = 0, kSecRegen = 1, kSecBox = 2, kSecPrim = 3, kSecSurface = 4, kSecSample = 5, kSecScatter = 6, kSecNode = 7 };
enum { kProfOuter = 0, kProfRegen = 1, kProfTrace = 2, kProfNode = 3, kProfPrim = 4, kProfShade = 5, kProfReject = 6,
       kProfXform = 7 };

constexpr float kPi = 3.1415926535897932385f;
constexpr int kNodeFields = 8;        // minx miny minz maxx maxy maxz code order   (float4 / int4 each)
constexpr int kNodeStrideGlobal = 8;  // float4 per node in HBM (one 128-byte line)
constexpr int kPrimQuads = 3;         // float4 per primitive record
constexpr int kShadeQuads = 2;        // float4 per shading record, the part every hit reads
constexpr int kShadeColdQuads = 4;    // float4 of its other part (smooth, textured or normal-mapped objects only)
constexpr int kXformFloats = 24;      // rinv[9] anchor[3] trans[3] rot[9]

// What a scene contains, found at upload.  The kernels are compiled for sets of these features (a
// template argument FEAT): code for a feature outside the set is left out, and a scene runs the
// variant with the smallest set that covers it.  The results are the same in every variant that
// covers the scene.
enum : int {
    kFeatPlainXf = 1 << 0,      // transforms without a rotation are flagged (kXfIdentityRotation)
    kFeatSphere = 1 << 1,       // sphere objects
    kFeatSlotXf = 1 << 2,       // flat mode: a slot whose primitives carry their own transform
    kFeatWide = 1 << 3,         // flat mode: more than 32 slots
    kFeatShading = 1 << 4,      // smooth normals, objects whose (u, v) is read, normal maps, checker or image textures
    kFeatMaterials = 1 << 5,    // metal or dielectric materials
    kFeatEnvTexture = 1 << 6,   // the environment is not one colour
    kFeatCamera = 1 << 7,       // orthographic camera, defocus blur, moving spheres (ray time)
    kFeatXform = 1 << 8,        // objects under a transform (rotate / translate wrappers)
    kFeatAll = 0x1ff,
    kFeatExactTies = 1 << 9,    // tree walks only, on request: near-ties decided the way the reference's ordered walk decides them (consider_exact())
    kFeatBasic = 0,             // untransformed flat triangles under diffuse materials and lights of one colour, pinhole camera
    kFeatBasicXform = kFeatXform,  // the same with transformed objects (rotated boxes)
    // any shapes, transforms and camera kinds under diffuse materials and lights of one colour, environment of one colour
    // (no textures or normal maps, no metal or glass: none of the code that looks things up per hit)
    kFeatPlain = kFeatPlainXf | kFeatSphere | kFeatSlotXf | kFeatCamera | kFeatXform,
    // every shape, material and texture kind seen through the pinhole camera without defocus or moving spheres, under an
    // environment of one colour, in at most 32 slots
    kFeatTextured = kFeatAll & ~(kFeatWide | kFeatCamera | kFeatEnvTexture)
};

struct DevImage {
    const uint8_t* pixels;
    int32_t width, height, channels, pad;
};

struct DevCamera {
    int32_t type;
    int32_t needs_time;       // some sphere moves (center direction != 0)
    float defocus_angle;
    V3 center, llc, du, dv, w, defocus_u, defocus_v;
};

// Everything a kernel needs, passed by value (lands in SGPRs / scalar loads).
struct DevScene {
    const float4* nodes;      // [node_count][8]     node-major
    const float4* prims;      // [object_count][3]
    const float4* shade;      // [object_count][2]
    const float4* shade_cold; // [object_count][4]
    const float* xforms;      // [xform_count][24]
    const float4* mats;       // [material_count][2]
    const float4* texs;       // [texture_count][2]
    const DevImage* images;
    int32_t node_count, object_count, xform_count, material_count, texture_count, background_index;
    int32_t width, height, max_bounces;
    int32_t root_entry;       // stack entry of BVH node 0: leafy flag in bit 0
    int32_t order_flip;       // see SceneGlobal::order_flip
    int32_t fast_rcp;         // the host proved reciprocal()'s fast form exact for this scene (selects the kernel variant)
    DevCamera cam;
};

// ------------------------------------------------------------------ RNG -----------------
// Keyed counter RNG: state = H(H(1984 + pixel) + n*PHI) for sample number n (1-based);
// draw k (0-based) = H(state + (k+1)*PHI); H = lowbias32; uniform = ((x>>8)+1) * 2^-24,
// in (0,1] like curand_uniform.  Replaces curand_init(1984+pixel, ...) (render.cuh:117).
__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// (kept as ONE word, pos = state + draws * PHI modulo 2^32: a draw is an addition and a hash, and a path carries
// one register instead of two)
constexpr uint32_t kRngStep = 0x9E3779B9u;
struct Rng {
    uint32_t pos;
    __device__ __forceinline__ void init(int pixel_index, int sample_number) {
        uint32_t s = lowbias32(1984u + (uint32_t)pixel_index);
        pos = lowbias32(s + (uint32_t)sample_number * kRngStep);
    }
    __device__ __forceinline__ float uniform() {
        pos += kRngStep;
        uint32_t x = lowbias32(pos);
        return (float)((x >> 8) + 1u) * (1.0f / 16777216.0f);
    }
    // A draw whose value is never read only advances the counter.
    __device__ __forceinline__ void skip(uint32_t n) { pos += n * kRngStep; }
    __device__ __forceinline__ V3 range3(float lo, float hi) {       // vec3::random(min,max) vec3.cuh:191
        float x = lo + (hi - lo) * uniform();
        float y = lo + (hi - lo) * uniform();
        float z = lo + (hi - lo) * uniform();
        return v3(x, y, z);
    }
};

__device__ __forceinline__ void random_in_unit_disk(Rng& r, float& px, float& py) {   // vec3.cuh:249-255
    do {
        float a = r.uniform(), b = r.uniform();
        px = 2.0f * a - 1.0f;
        py = 2.0f * b - 1.0f;
    } while (px * px + py * py + 0.0f * 0.0f >= 1.0f);
}

// ------------------------------------------------------------------ camera --------------
struct Ray {
    V3 o, d;
    float time;
};
template <int FEAT = kFeatAll>
__device__ __forceinline__ Ray camera_ray(const DevCamera& cam, int i, int j, Rng& rng) {
    // sample_square draws three uniforms and uses two (camera.cuh:93-96)
    const float offx = -0.5f + (0.5f - -0.5f) * rng.uniform();
    const float offy = -0.5f + (0.5f - -0.5f) * rng.uniform();
    rng.skip(1);
    V3 on_plane = cam.llc + ((float)i + offx) * cam.du + ((float)j + offy) * cam.dv;
    Ray r;
    if (!(FEAT & kFeatCamera) || cam.type == 0) {   // perspective_ray (camera.cuh:240-248)
        if (!(FEAT & kFeatCamera) || cam.defocus_angle <= 0.0f) {
            r.o = cam.center;
        } else {
            float px, py;
            random_in_unit_disk(rng, px, py);
            r.o = cam.center + (px * cam.defocus_u) + (py * cam.defocus_v);
        }
        r.d = on_plane - r.o;
    } else {                                        // orthographic_ray (camera.cuh:259-266)
        r.o = on_plane;
        r.d = cam.w;
    }
    // ray time (camera.cuh:246,264) is only read by moving spheres
    if ((FEAT & kFeatCamera) && cam.needs_time) r.time = rng.uniform();
    else { r.time = 0.0f; rng.skip(1); }
    return r;
}

// base[index] with the byte offset formed in 32 bits.  An `int` index is sign-extended and the address
// computed in 64-bit vector arithmetic (two to four instructions per load); with an unsigned 32-bit byte
// offset the load takes the table's address from scalar registers and the offset from one vector
// register.  The host refuses a scene with a table of 4 GiB or more.
template <class T>
__device__ __forceinline__ const T* at32(const T* base, int index) {
    return (const T*)((const char*)base + (uint32_t)index * (uint32_t)sizeof(T));
}

// ------------------------------------------------------------------ scene memory views --
// Two views with the same interface: HBM (node-major records, served by L1/L2) and an LDS
// copy (field-major, so that lanes on different nodes fall on different banks).
// A reference to a node, in a slot code of its parent and on the traversal stack: node index * kNodeEntryScale +
// leafy (bit 0: the node has leaf slots of its own), >= 0; -1 = none.
constexpr int kNodeEntryScale = 128;
struct SceneGlobal {
    const float4* nodes;
    const float4* prims;
    const float* xforms;
    int root_entry;        // stack entry of node 0
    int order_flip;        // 15: rays towards -z read a node's slot order backwards; 0: one order for every ray (reference-order walk)
    // entry: a stack entry, node * 128 + leafy = the byte offset of the node's record (128 bytes) + its flag
    __device__ __forceinline__ float4 node_field(int entry, int field) const {
        static_assert(kNodeStrideGlobal * 16 == kNodeEntryScale, "a stack entry is a byte offset into the node records");
        return *(const float4*)((const char*)nodes + (((uint32_t)entry & ~(uint32_t)(kNodeEntryScale - 1)) + (uint32_t)field * 16u));
    }
    __device__ __forceinline__ float4 prim(int obj, int q) const {
        return *at32(prims, obj * kPrimQuads + q);
    }
    __device__ __forceinline__ const float* xform(int x) const { return at32(xforms, x * kXformFloats); }
};
struct SceneLds {
    const float4* nodes;   // [7][node_count]
    const float4* prims;   // [3][object_count]
    const float* xforms;
    int node_count, object_count;
    int root_entry;
    int order_flip;
    __device__ __forceinline__ float4 node_field(int entry, int field) const { return nodes[field * node_count + entry / kNodeEntryScale]; }
    __device__ __forceinline__ float4 prim(int obj, int q) const { return prims[q * object_count + obj]; }
    __device__ __forceinline__ const float* xform(int x) const { return xforms + x * kXformFloats; }
};

// Per-lane traversal stack.  The first `lds_entries` entries (7 or 11) of lane l live in LDS at
// base[e * stride + l] (conflict-free); a walk that goes deeper continues in a per-thread slice
// of global memory (spill[(e - lds_entries) * spill_stride]), so that LDS use - and with it the
// number of resident waves - does not grow with the depth of the tree.
// The two parts are addressed through pointers of their own address space: with generic pointers
// the compiler folds the two branches of get() into one address select and ONE flat load, and a
// flat load that ends up in LDS takes the long way round through the vector-memory path - on
// every pop of the walk.
typedef int __attribute__((address_space(3))) * LdsIntPtr;
typedef float __attribute__((address_space(3))) * LdsFloatPtr;
typedef int __attribute__((address_space(1))) * GlobalIntPtr;
struct LaneStack {
    LdsIntPtr base;
    int stride;
    int lds_entries;
    // Slot codes and box entry distances of the node whose leaf candidates are pending: lane l's
    // slot i at code[i * stride] / entry[i * stride].  Written once per node with leaf hits and read
    // once per candidate, they sit in LDS (8 words per lane) instead of eight registers that would
    // be live across the whole path loop.
    LdsIntPtr code;
    LdsFloatPtr entry;
    GlobalIntPtr spill;
    size_t spill_stride;
    // The LDS part is `rows` rows of `threads` words.  Row 0 holds -1 for every lane and is what a pop from an
    // empty stack reads ("no node left": the pop needs no test for the empty stack); entry e of this lane is at
    // base[e * stride] with base in row 1.
    __device__ __forceinline__ void attach(int* lds_rows, int rows, int threads, int thread) {
        if (rows > 0) lds_rows[thread] = -1;                       // (no rows: a kernel variant that never walks a tree)
        base = (LdsIntPtr)(lds_rows + threads + thread);
        stride = threads;
        lds_entries = rows - 1;
    }
    __device__ __forceinline__ void put(int e, int v) {
        if (e < lds_entries) base[e * stride] = v;
        else spill[(size_t)(e - lds_entries) * spill_stride] = v;
    }
    __device__ __forceinline__ int get(int e) const {
        return e < lds_entries ? base[e * stride] : spill[(size_t)(e - lds_entries) * spill_stride];
    }
};

struct Hit {
    float t;     // closest accepted ray parameter
    int obj;     // object index, -1 = miss
    float bu, bv;  // triangle barycentrics
};

// word = (word << 1) | (this lane's bit of `in`), and the bit shifted out goes to this lane's
// bit of the result: one add-with-carry, the compare result never leaves the scalar mask.
__device__ __forceinline__ unsigned long long shift_in(unsigned& word, unsigned long long in) {
    unsigned long long out;
    asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(word), "=s"(out) : "v"(word), "s"(in));
    return out;
}
// 1.0f / x.  With `fast` set (a compile-time constant at every call site): v_rcp_f32 plus one Newton step (two FMAs) instead of the general
// division sequence.  For every float with 2^-100 <= |x| <= 2^100 that is bit-identical to the
// IEEE quotient (checked over all 2^32 inputs on gfx950, tools/ubench/rcp_exhaustive.hip).  The
// host sets `fast` only for scenes whose geometry bounds every divisor used on the path into
// that range (see fast_reciprocal_is_exact()); otherwise the ordinary division runs.
__device__ __forceinline__ float reciprocal(float x, int fast) {
    if (fast) {
        const float r0 = __builtin_amdgcn_rcpf(x);
        const float e = __builtin_fmaf(-x, r0, 1.0f);
        return __builtin_fmaf(e, r0, r0);
    }
    return 1.0f / x;
}
// a / s.  With `fast`: through the exact reciprocal r of s and one correction step (q0 = a*r, the
// remainder a - s*q0 comes out of one fma exactly, q = q0 + remainder*r): the correctly rounded
// quotient whenever r is the correctly rounded reciprocal and nothing underflows (Markstein), five
// instructions after the shared reciprocal instead of the eleven of the general division sequence;
// sampled against the IEEE quotient in tools/ubench/rcp_exhaustive.hip.  Same condition on the scene
// as reciprocal().  The remainder is about 2^-47 |a|: for a numerator below 2^-90 it underflows and the
// last bit of q can be off (seen up to |a| = 5.2e-28 in that sweep), so numerators below 2^-80, zero
// among them, take the ordinary division; above that the sweep finds no mismatch.
constexpr float kSmallNumerator = 0x1p-80f;
__device__ __forceinline__ float quotient(float a, float s, float r) {
    if (fabsf(a) < kSmallNumerator) return a / s;
    const float q0 = a * r;
    const float rem = __builtin_fmaf(-s, q0, a);
    return __builtin_fmaf(rem, r, q0);
}
template <int FAST>
__device__ __forceinline__ V3 divide(V3 v, float s) {
    if (FAST) {
        const float r = reciprocal(s, 1);
        return v3(quotient(v.x, s, r), quotient(v.y, s, r), quotient(v.z, s, r));
    }
    return v / s;
}
template <int FAST>
__device__ __forceinline__ V3 unit_vector(V3 a) { return divide<FAST>(a, length(a)); }   // vec3.cuh unit(): v / length

__device__ __forceinline__ void to_object_space(const float* xf, V3 o, V3 d, V3& oo, V3& od) {   // hittable.cuh:429-430
    V3 anchor = v3(xf[9], xf[10], xf[11]), trans = v3(xf[12], xf[13], xf[14]);
    oo = matvec(xf, o - anchor - trans) + anchor;
    od = matvec(xf, d);
}
// A primitive's transform index with this bit set: rotation and inverse rotation are exactly the
// identity matrix (a translated object, or any sphere: the reference anchors a sphere at its
// centre), only anchor and translation act.  The two matrix products are then left out:
// 1*x + 0*y + 0*z is x for every finite vector (up to the sign of a zero), so (o - anchor -
// translation) + anchor and the untouched direction are what hittable.cuh:429-430 computes.
// Without kFeatPlainXf the scene has no such transform and the test for it is compiled out.
constexpr int kXfIdentityRotation = 1 << 30;
template <int FEAT = kFeatAll>
__device__ __forceinline__ const float* xform_record(const float* xforms, int xf) {
    return at32(xforms, ((FEAT & kFeatPlainXf) ? (xf & ~kXfIdentityRotation) : xf) * kXformFloats);
}
__device__ __forceinline__ void to_object_space(const float* X, bool plain, V3 o, V3 d, V3& oo, V3& od) {
    if (plain) {
        V3 anchor = v3(X[9], X[10], X[11]), trans = v3(X[12], X[13], X[14]);
        oo = (o - anchor - trans) + anchor;
        od = d;
    } else {
        to_object_space(X, o, d, oo, od);
    }
}

// aabb.cuh:140 -- the reciprocal drops the sign for tiny components (always +1e8)
__device__ __forceinline__ V3 slab_reciprocal(V3 d, int fast) {
    return v3((fabsf(d.x) > 1e-12f) ? reciprocal(d.x, fast) : 1e8f, (fabsf(d.y) > 1e-12f) ? reciprocal(d.y, fast) : 1e8f,
              (fabsf(d.z) > 1e-12f) ? reciprocal(d.z, fast) : 1e8f);
}

// hit_tri, hittable.cuh:314-335 (acceptance part).  Interval test is inclusive.
__device__ __forceinline__ bool tri_test(V3 o, V3 d, V3 point, V3 e1, V3 e2, float tmin, float tmax, int fast, float& t, float& bu, float& bv) {
    V3 pvec = cross(d, e2);
    float det = dot(e1, pvec);
    if (fabsf(det) < 1e-8f) return false;
    float inv_det = reciprocal(det, fast);
    V3 tvec = o - point;
    float u = dot(tvec, pvec) * inv_det;
    if (u < 0.0f || u > 1.0f) return false;
    V3 qvec = cross(tvec, e1);
    float v = dot(d, qvec) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return false;
    float tt = dot(e2, qvec) * inv_det;
    if (!(tmin <= tt && tt <= tmax)) return false;
    t = tt; bu = u; bv = v;
    return true;
}

// hit_sphere, hittable.cuh:272-290 (acceptance part).  Interval test is exclusive.
template <int FAST>
__device__ __forceinline__ bool sphere_test(V3 o, V3 d, float time, V3 c0, V3 cd, float radius, float tmin, float tmax, float& t) {
    V3 center = c0 + cd * time;
    V3 oc = o - center;
    float a = length_squared(d);
    float h = dot(oc, d);
    float c = length_squared(oc) - radius * radius;
    float disc = h * h - a * c;
    if (disc < 0.0f) return false;
    float sq = sqrtf(disc);
    const float ra = FAST ? reciprocal(a, 1) : 0.0f;
    float root = FAST ? quotient(-h - sq, a, ra) : (-h - sq) / a;
    if (!(tmin < root && root < tmax)) {
        root = FAST ? quotient(-h + sq, a, ra) : (-h + sq) / a;
        if (!(tmin < root && root < tmax)) return false;
    }
    t = root;
    return true;
}

// Order-independent form of "if (hit(r, [tmin, closest])) closest = rec.t" (bvh.cuh:407-411).
// The closest hit does not depend on the order of the tests except when two candidates have
// exactly the same t.  The reference visits leaf slots in a fixed depth-first order that does
// not depend on the ray (bvh.cuh:387-419: slots 0..3 in order, children popped last-in
// first-out), accepts a triangle on the closed interval (hittable.cuh:335) and a sphere on the
// open one (:287-289).  With `rank` = position of a primitive in that order, "tested later wins
// a tie if it is a triangle, never if it is a sphere" reproduces the reference's choice in any
// test order.  best.obj holds the rank of the best candidate so far.
// One refinement for the tree walks, which may meet the later-ranked of two tied candidates first: the
// reference would have met the earlier one first, shrunk the ray interval to t, and then tested the later one
// only if its box is still hit on [tmin, t] (bvh.cuh:398-411) - which fails when the hit lies, by rounding,
// just before the computed entry distance of its own box (a ray through an edge that lies in a box face).
// `lo` = box entry distance of the candidate, `best_lo` = that of the best so far; the flat mode passes -inf.
// ALL_TRI: the scene holds nothing but triangles, so the best candidate so far is one and `best_tri` is
// neither read nor written (a flag less to carry round the loops).
template <bool ALL_TRI = false>
__device__ __forceinline__ void consider(Hit& best, bool& best_tri, float& best_lo, bool ok, float t, float lo, int rank, bool is_tri, float bu, float bv) {
    const bool best_would_be_culled = best_lo > t;             // (only read when t == best.t)
    const bool win_tie = best.obj < 0 ? true : ((rank > best.obj) ? (ALL_TRI || is_tri) : ((ALL_TRI ? false : !best_tri) || best_would_be_culled));
    if (ok && (t < best.t || (t == best.t && win_tie))) {
        best.t = t; best.obj = rank; best.bu = bu; best.bv = bv; best_lo = lo;
        if (!ALL_TRI) best_tri = is_tri;
    }
}

// The same decision for a walk that has to agree with the reference's ordered walk (bvh.cuh:387-419) even where a hit
// lies, by rounding, just in front of its own box.  The reference meets its candidates in rank order and tests one
// only if its box is hit on [tmin, closest so far]; so of two candidates the later-ranked one counts iff its box
// entry does not lie beyond the earlier one's t AND its own t does not (triangle: closed interval, sphere: open).
// X against the best so far B, in either order of arrival:
//   X ranked after B:   X replaces B iff lo_X <= t_B and t_X <= t_B (sphere: <);
//   X ranked before B:  B would have replaced X iff lo_B <= t_X and t_B <= t_X (sphere B: <); else X stands.
// The walk culls boxes and primitives against the closest hit times (1 + 2^-12), so that both of two near-tied
// candidates reach this decision.  (Chains of three candidates within rounding of each other are not modelled.)
__device__ __forceinline__ void consider_exact(Hit& best, bool& best_tri, float& best_lo, bool ok, float t, float lo, int rank, bool is_tri, float bu, float bv) {
    const bool x_after = lo <= best.t && (is_tri ? t <= best.t : t < best.t);
    const bool b_stays = best_lo <= t && (best_tri ? best.t <= t : best.t < t);
    const bool take = best.obj < 0 ? true : (rank > best.obj ? x_after : !b_stays);
    if (ok && take) {
        best.t = t; best.obj = rank; best.bu = bu; best.bv = bv; best_lo = lo;
        best_tri = is_tri;
    }
}
// Upper end of the ray interval boxes and primitives are culled against.
template <int FEAT>
__device__ __forceinline__ float cull_distance(float closest) { return (FEAT & kFeatExactTies) ? closest * 1.000244140625f : closest; }

// Closest hit over the flat BVH4, resumable.
//
// The walk differs from bvh.cuh:387-419 in the ORDER nodes are visited (children nearest along
// the ray first, from a per-node table indexed by the ray's direction octant) and gives the
// same hit: the tie rule of consider() makes the result independent of the order, and a box is
// only skipped when it lies entirely beyond the closest hit so far (hi < lo exactly as in
// aabb.cuh:139-154, with the interval's upper end at the current closest t).
//
// For the wave the work is split into two alternating phases ("while-while"):
//   node phase  every lane that has no pending primitives descends (box tests, pushes, pop)
//               until it reaches a node with leaf candidates or runs out of nodes;
//   prim phase  every lane tests its pending candidates, one per round, in slot order.
// Lanes reach leaves at different times; testing primitives the moment one lane finds them
// would run the long intersection code for a handful of lanes, while batching them keeps both
// phases fuller.
//
// Rays need very different numbers of steps, so a wave that waits for its slowest ray runs
// mostly empty.  tree_run() therefore returns as soon as fewer than `min_active` lanes are
// still walking; the caller shades the finished lanes, gives them new rays and calls again,
// and the unfinished lanes simply continue from their saved state (registers + LDS stack).
struct TreeState {
    int node, sp;                       // next node to visit as a stack entry: index * kNodeEntryScale + leafy (-1: none left), stack height
    unsigned leaf_mask;                 // pending leaf slots of the last visited node
                                        // (its slot codes and box entry distances: LaneStack::code, ::entry)
    Hit best;                           // best.obj = visiting rank while walking
    int best_obj;                       // object index of the best candidate
    bool best_tri;
    float best_lo;                      // entry distance of the best candidate's box (exact ties, see consider())
    __device__ __forceinline__ void begin(float tmax, int root_entry) {
        node = root_entry; sp = 0; leaf_mask = 0;
        best.t = tmax; best.obj = -1; best.bu = 0.0f; best.bv = 0.0f;
        best_obj = -1; best_tri = false; best_lo = -INFINITY;
    }
    __device__ __forceinline__ Hit result() const { Hit h = best; h.obj = best_obj; return h; }
};

// Keeps a loaded record whole: without it the compiler moves the loads of fields that are only read
// under later branches (a triangle's first vertex after the determinant test, its rank after the
// interval test) down into those branches, and a primitive test then waits for memory three times.
__device__ __forceinline__ void keep_loaded(float4& q) { asm volatile("" : "+v"(q.x), "+v"(q.y), "+v"(q.z), "+v"(q.w)); }

template <int FAST, int FEAT = kFeatAll, class Mem>
__device__ __forceinline__ void tree_prim(const Mem& S, int obj, float lo, const Ray& r, float tmin, TreeState& ts) {
    float4 q0 = S.prim(obj, 0), q1 = S.prim(obj, 1), q2 = S.prim(obj, 2);
    keep_loaded(q0); keep_loaded(q1); keep_loaded(q2);
    const int xf = __float_as_int(q2.y), type = __float_as_int(q2.z), rank = __float_as_int(q2.w);
    V3 o = r.o, d = r.d;
    if ((FEAT & kFeatXform) && xf >= 0) { CRT_PROF(kProfXform); to_object_space(xform_record<FEAT>(S.xforms, xf), (FEAT & kFeatPlainXf) && (xf & kXfIdentityRotation), r.o, r.d, o, d); }
    const int before = ts.best.obj;
    if (!(FEAT & kFeatSphere) || type == 2) {
        float t = 0.0f, bu = 0.0f, bv = 0.0f;
        const bool ok = tri_test(o, d, v3(q0.x, q0.y, q0.z), v3(q0.w, q1.x, q1.y), v3(q1.z, q1.w, q2.x), tmin, cull_distance<FEAT>(ts.best.t), FAST, t, bu, bv);
        if constexpr ((FEAT & kFeatExactTies) != 0) consider_exact(ts.best, ts.best_tri, ts.best_lo, ok, t, lo, rank, true, bu, bv);
        else consider<!(FEAT & kFeatSphere)>(ts.best, ts.best_tri, ts.best_lo, ok, t, lo, rank, true, bu, bv);
    } else {
        // the root the reference would take does not depend on the interval's upper end: the
        // near root if it is beyond tmin, else the far one (hittable.cuh:283-289)
        float t = 0.0f;
        const bool ok = sphere_test<FAST>(o, d, r.time, v3(q0.x, q0.y, q0.z), v3(q1.x, q1.y, q1.z), q0.w, tmin, INFINITY, t);
        if constexpr ((FEAT & kFeatExactTies) != 0) consider_exact(ts.best, ts.best_tri, ts.best_lo, ok, t, lo, rank, false, 0.0f, 0.0f);
        else consider(ts.best, ts.best_tri, ts.best_lo, ok, t, lo, rank, false, 0.0f, 0.0f);
    }
    if (ts.best.obj != before) ts.best_obj = obj;
}

// The two per-lane tests of tree_run's loop as ONE compare each (a test that combines several compares reaches the
// lane mask through scalar and/or instructions and a detour through a vector register).
// `next` is a stack entry (node * kNodeEntryScale + leafy, >= 0) or -1.
//   can_step: a node to visit, and no untested candidates in front of a leafy node: the blocked case sets the sign bit
__device__ __forceinline__ bool can_step(int next, unsigned leaf_mask) {
    const unsigned blocked = (unsigned)next & (leaf_mask < 1u ? leaf_mask : 1u);
    return (int)((unsigned)next | (blocked << 31)) >= 0;
}
//   still_walking: a node to visit or candidates to test
__device__ __forceinline__ bool still_walking(int next, unsigned leaf_mask) { return (((unsigned)next + 1u) | leaf_mask) != 0u; }

// One wave-wide loop; in every round
//   * each lane that can takes ONE node step (box tests of the four slots, pushes, pop).  A lane
//     that already holds untested leaf candidates keeps descending through internal nodes and
//     only stops in front of the next node that has leaf slots ("leafy": flagged in the
//     reference to it, bit 0 of a stack entry);
//   * leaf candidates are tested, one per lane, once at least `prim_batch` lanes hold some (or
//     nobody can take a node step): the long intersection code then runs for a batch of lanes
//     instead of the one or two that happen to reach a leaf in a given round.
template <int FAST, int FEAT = kFeatAll, class Mem>
__device__ __forceinline__ void tree_run(const Mem& S, const Ray& r, float tmin, TreeState& ts, bool& walking, LaneStack stk, int min_active, int prim_batch) {
    const V3 inv = slab_reciprocal(r.d, FAST);
    const float ix = inv.x, iy = inv.y, iz = inv.z;
    // field numbers of the near / far planes per axis (swap when the reciprocal is negative)
    const int nx = ix < 0.0f ? 3 : 0, fx = 3 - nx;
    const int ny = iy < 0.0f ? 4 : 1, fy = 5 - ny;
    const int nz = iz < 0.0f ? 5 : 2, fz = 7 - nz;
    const int octant = (ix < 0.0f ? 1 : 0) | (iy < 0.0f ? 2 : 0) | (iz < 0.0f ? 4 : 0);
    // A node's last quad: {order of octants 0,1 | order of octants 2,3 | leaf slots (bits 0-3) and
    // internal slots (bits 4-7) | -}.  An octant's order is 16 bits: for slot i, bits 4i..4i+3 are
    // the slots visited AFTER slot i (farther along the octant's diagonal).  Octants 4-7 (rays
    // towards -z) are the mirror images of octants 3-0 and read their entry backwards: the slots
    // visited after i are then the others (and i itself, taken off again through `self`).
    const unsigned flip = (octant & 4) ? (unsigned)S.order_flip : 0u;
    const int self = -(int)(flip & 1u);
    const int oct_entry = (octant & 4) ? (octant ^ 7) : octant;
    const bool oct_word = (oct_entry & 2) != 0;
    const int oct_shift = (oct_entry & 1) * 16;

    // A lane that is not walking holds no node and no candidates, so `walking` drops out of every test
    // of the loop.  The lanes that take a node step are kept as a lane mask in scalar registers: it is
    // evaluated once per round (it also decides whether the candidates are tested now) and again only
    // after a round of primitive tests; the branch around the node step takes it as it is.
    int next = walking ? ts.node : -1;
    unsigned leaf_mask = walking ? ts.leaf_mask : 0u;
    unsigned long long stepping = lanes_where(can_step(next, leaf_mask));
    while (true) {                                                  // wave-uniform
        if (__builtin_amdgcn_inverse_ballot_w64(stepping)) {
            CRT_PROF(kProfNode);
            float4 bnx = S.node_field(next, nx), bfx = S.node_field(next, fx);
            float4 bny = S.node_field(next, ny), bfy = S.node_field(next, fy);
            float4 bnz = S.node_field(next, nz), bfz = S.node_field(next, fz);
            float4 codef = S.node_field(next, 6), orderf = S.node_field(next, 7);
            const float nxs[4] = {bnx.x, bnx.y, bnx.z, bnx.w}, fxs[4] = {bfx.x, bfx.y, bfx.z, bfx.w};
            const float nys[4] = {bny.x, bny.y, bny.z, bny.w}, fys[4] = {bfy.x, bfy.y, bfy.z, bfy.w};
            const float nzs[4] = {bnz.x, bnz.y, bnz.z, bnz.w}, fzs[4] = {bfz.x, bfz.y, bfz.z, bfz.w};
            const int codes[4] = {__float_as_int(codef.x), __float_as_int(codef.y), __float_as_int(codef.z), __float_as_int(codef.w)};
            float los[4];
            unsigned hit_mask = 0;                                  // bit i: the ray meets slot i's box before the closest hit so far
            const float closest = cull_distance<FEAT>(ts.best.t);
#pragma unroll
            for (int i = 3; i >= 0; i--) {
                float t0 = (nxs[i] - r.o.x) * ix, t1 = (fxs[i] - r.o.x) * ix;
                float lo = fmaxf(tmin, t0), hi = fminf(closest, t1);
                t0 = (nys[i] - r.o.y) * iy; t1 = (fys[i] - r.o.y) * iy;
                lo = fmaxf(lo, t0); hi = fminf(hi, t1);
                t0 = (nzs[i] - r.o.z) * iz; t1 = (fzs[i] - r.o.z) * iz;
                lo = fmaxf(lo, t0); hi = fminf(hi, t1);
                los[i] = lo;
                unsigned long long hit;                             // !(hi < lo), from the compare's lane mask straight into the bit word
                asm("v_cmp_nlt_f32_e64 %0, %1, %2" : "=s"(hit) : "v"(hi), "v"(lo));
                (void)shift_in(hit_mask, hit);
            }
            // (all three words are read here: used under the branch only, the two order words would be
            // fetched by a load of their own after the box tests)
            unsigned kinds = (unsigned)__float_as_int(orderf.z);
            unsigned order = (unsigned)__float_as_int(oct_word ? orderf.y : orderf.x) >> oct_shift;
            asm volatile("" : "+v"(kinds), "+v"(order));
            const unsigned leaf_hits = hit_mask & kinds & 15u, push_mask = hit_mask & (kinds >> 4);
            // internal children onto the stack, farthest first, so that the nearest is popped first:
            // a slot's place above the old top is the number of pushed slots visited after it
            if (push_mask) {
                if (!lanes_where(ts.sp + 4 > stk.lds_entries)) {        // decided for the wave: two instructions instead of a divergent if / else
                    // every slot writes, without a branch: the ones that are not pushed to a word above the new top
                    const int first = ts.sp + self, unused = ts.sp + 3;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const unsigned after = ((order >> (4 * i)) & 15u) ^ flip;
                        const int place = __builtin_popcount(after & push_mask) + first;
                        const bool pushed = (push_mask >> i) & 1u;
                        stk.base[(pushed ? place : unused) * stk.stride] = codes[i];
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const unsigned after = ((order >> (4 * i)) & 15u) ^ flip;
                        const int place = __builtin_popcount(after & push_mask) + self;
                        if ((push_mask >> i) & 1u) stk.put(ts.sp + place, codes[i]);
                    }
                }
                ts.sp += __builtin_popcount(push_mask);
            }
            if (leaf_hits) {                                        // only reached with no candidates pending
                leaf_mask = leaf_hits;
                stk.code[0] = codes[0]; stk.code[stk.stride] = codes[1]; stk.code[2 * stk.stride] = codes[2]; stk.code[3 * stk.stride] = codes[3];
                stk.entry[0] = los[0]; stk.entry[stk.stride] = los[1]; stk.entry[2 * stk.stride] = los[2]; stk.entry[3 * stk.stride] = los[3];
            }
            // pop; -1 from the row below an empty stack (sp is then -1 until the lane's next walk begins)
            --ts.sp;
            if (!lanes_where(ts.sp >= stk.lds_entries)) next = stk.base[ts.sp * stk.stride];
            else next = stk.get(ts.sp);
        }
        const bool has_prim = leaf_mask != 0;
        const int pending = __builtin_popcountll(lanes_where(has_prim));
        stepping = lanes_where(can_step(next, leaf_mask));
        if (pending && (pending >= prim_batch || !stepping)) {
            if (has_prim) {
                const int i = __builtin_ctz(leaf_mask);
                leaf_mask &= leaf_mask - 1;
                const int code = stk.code[i * stk.stride];
                const float lo = stk.entry[i * stk.stride];
                // an earlier candidate may have shrunk the interval below this box's entry
                if (!(cull_distance<FEAT>(ts.best.t) < lo)) {
                    CRT_PROF(kProfPrim);
                    tree_prim<FAST, FEAT>(S, code, lo, r, tmin, ts);
                }
            }
            stepping = lanes_where(can_step(next, leaf_mask));
        }
        if (__builtin_popcountll(lanes_where(still_walking(next, leaf_mask))) < min_active) break;
    }
    ts.node = next;
    ts.leaf_mask = leaf_mask;
    walking = still_walking(next, leaf_mask);
}

// One whole walk (used where no lane is refilled in between).
template <class Mem>
__device__ __forceinline__ Hit traverse(const Mem& S, const Ray& r, float tmin, float tmax, LaneStack stk) {
    TreeState ts;
    ts.begin(tmax, S.root_entry);
    bool walking = true;
    tree_run<0>(S, r, tmin, ts, walking, stk, 1, 16);
    return ts.result();
}

// ------------------------------------------------------------------ flat traversal ------
// Small scenes (at most 64 box slots).  Measured on the 36-triangle Cornell box, the tree walk
// spends its time on divergence, not on arithmetic: the inner boxes reject almost nothing
// (31 box tests per ray through the tree against 36 leaves) while lanes sit on different
// nodes and reach leaves at different times (8-20 % of the lanes active in the primitive
// tests).  The flat mode removes the tree from the inner loop:
//
//   1. BOX PHASE, uniform across the wave.  Every box slot is tested by every lane in one
//      loop whose box data comes through scalar loads (constant address space); there is no
//      stack, no per-lane addressing, no divergence.  A hit sets one bit in the lane's mask.
//   2. PRIMITIVE PHASE.  Each lane walks its set bits; one round of the loop tests one slot
//      per lane.
//
// What a slot is.  Triangles that share a bit-identical box and transform are paired (a quad
// is two such triangles, and quads/boxes are what scenes here are made of), so one box test
// and one round cover both.  Triangles under a rotation get their box in OBJECT space (tight
// slab, padded) and are tested with the ray transformed once per rotation ("group") instead
// of the world-space box of the rotated slab, which for a rotated cube face is almost the
// whole cube: that alone cuts the candidates of a ray that passes a rotated block from up to
// 12 to about 2.  Everything else (untransformed triangles, spheres) keeps the world box the
// reference stores in its BVH slot.
//
// Same result as the reference's ordered walk.  The closest hit does not depend on the order
// of the tests except when two candidates have exactly the same t.  The reference visits leaf
// slots in a fixed depth-first order that does not depend on the ray (bvh.cuh:387-419: slots
// 0..3 in order, children popped last-in first-out), accepts a triangle on the closed interval
// (hittable.cuh:335) and a sphere on the open one (:287-289).  So with `rank` = position of a
// primitive in that order, "tested later wins a tie if it is a triangle, never if it is a
// sphere" reproduces the reference's choice in any test order; it is applied in consider().
typedef float NativeQuad __attribute__((ext_vector_type(4)));
typedef const NativeQuad __attribute__((address_space(4))) * ConstQuadPtr;
typedef const float __attribute__((address_space(4))) * ConstFloatPtr;

constexpr int kSlotQuads = 5;          // float4 per slot record of the primitive phase
constexpr int kMaxFlatSlots = 64;
constexpr int kMaxFlatGroups = 2;      // rotations that get their own object-space group

struct FlatView {                      // part of the kernel arguments
    const float4* boxes;               // [box_end[last]][2]: {minx,maxx,miny,maxy},{minz,maxz,0,0} of the slots outside cuboids,
                                       // then, from quad cuboid_quad0 on, the cuboid records: world group 5 quads each (flat_cuboid()),
                                       // rotated groups 4 quads each (flat_cuboid_shared())
    const float4* slots;               // [slot_count][5]: meta, transform, triangle A, triangle B (or a sphere), see traverse_flat()
    const int32_t* ent_obj;            // rank in the reference's visiting order -> object index
    const float* group_xf;             // [groups][24] transform of group 1.. (rinv, anchor, trans, rot)
    int32_t slot_count;
    // Slots in the order of their box tests: group 0 (world space), group 1, group 2; within a group
    // the faces of its cuboids, then its other slots.  Running totals per group:
    int32_t box_end[kMaxFlatGroups + 1];     // box records:    [0,b0) world, [b0,b1) group 1, [b1,b2) group 2
    int32_t cuboid_end[kMaxFlatGroups + 1];  // cuboid records: [0,c0) world, [c0,c1) group 1, [c1,c2) group 2
    int32_t cuboid_quad0;
    int32_t groups;
    // Phases of the primitive loop (traverse_flat()): candidate bits (low word) that are sphere slots, and per
    // rotated group the bit of face zB of its last cuboid when that cuboid's far faces may wait for the near
    // ones (-1: not in use, see flat_cuboid_shared()).
    uint32_t sphere_bits;
    int32_t far_shift[kMaxFlatGroups + 1];
    // The world group's last cuboid (its faces keep their own intervals and may be absent): candidate bit of face
    // A and of face B per axis (0 = absent or not in use), for the same purpose.
    uint32_t world_face_a[3], world_face_b[3];
};

struct FlatLds {
    const float4* slots;               // [slot_count][5] copy in LDS, slots in reverse order (slot of candidate bit b at b)
    const float* xforms;               // transforms of world-group primitives that carry one
    int slot_count;
};

// One slab test of the box phase; the mask of the lanes whose ray hits the box on [tmin, +inf).
// For a non-empty box min <= max, so the reference's swap on a negative reciprocal
// (aabb.cuh:144-148) selects exactly fmin/fmax of the two plane distances.
__device__ __forceinline__ unsigned long long flat_box_hit(NativeQuad b0, NativeQuad b1, V3 o, V3 inv, float tmin) {
    const float ax = (b0.x - o.x) * inv.x, bx = (b0.y - o.x) * inv.x;
    const float ay = (b0.z - o.y) * inv.y, by = (b0.w - o.y) * inv.y;
    const float az = (b1.x - o.z) * inv.z, bz = (b1.y - o.z) * inv.z;
    const float lo = fmaxf(tmin, fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz)));
    const float hi = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    // !(hi < lo) as a lane mask in scalar registers (written as an instruction: a wave-wide
    // ballot in the source would keep the slot loop from being unrolled)
    unsigned long long hit;
    asm("v_cmp_nlt_f32_e64 %0, %1, %2" : "=s"(hit) : "v"(hi), "v"(lo));
    return hit;
}
// Box tests of slots [first, last) -> bits shifted into (hi:lo) from the right, so that slot s
// ends up at bit (slot_count-1-s): the first slot is the highest set bit.  WIDE = more than 32
// slots in the scene (the high word is in use).
template <bool WIDE>
__device__ __forceinline__ void flat_box_range(ConstQuadPtr boxes, int first, int last, V3 o, V3 inv, float tmin,
                                               unsigned& lo_word, unsigned& hi_word) {
    auto one = [&](int s) {
        const unsigned long long hit = flat_box_hit(boxes[2 * s], boxes[2 * s + 1], o, inv, tmin);
        const unsigned long long carry = shift_in(lo_word, hit);
        if constexpr (WIDE) (void)shift_in(hi_word, carry);
    };
    int s = first;
    for (; s + 4 <= last; s += 4) { one(s); one(s + 1); one(s + 2); one(s + 3); }    // unrolled by hand: the
    for (; s < last; s++) one(s);                        // instructions above count as convergent operations
}

// A cuboid: up to six slots of one group whose boxes share their planes the way the faces of an
// axis-aligned box do (the quads of a box object in its own space, the walls of a room).  The box
// of face "A of axis x" is [thin interval] x [outer y] x [outer z], and so on: 18 distinct plane
// values instead of 36, and the outer slabs' entry and exit distances are shared by four faces each.
// Every distance is computed from its plane exactly as flat_box_hit() does and minima / maxima are
// exact, so each face gets the same bit as its own slab test would give it.
// Record, 5 quads: {outer xlo,xhi,ylo,yhi} {outer zlo,zhi, present-face bits, -} {xA lo,hi, xB lo,hi} {yA.., yB..} {zA.., zB..};
// bits are shifted in for the present faces only, in the order xA xB yA yB zA zB.
// Returns the ray parameter in front of which no triangle of a far face can be hit (see flat_cuboid_shared());
// -inf without a margin (record word [1].w: +inf).
template <bool WIDE>
__device__ __forceinline__ float flat_cuboid(ConstQuadPtr rec, V3 o, V3 inv, float tmin, unsigned& lo_word, unsigned& hi_word) {
    const NativeQuad c0 = rec[0], c1 = rec[1];
    const int present = __float_as_int(c1.z);
    const float ax = (c0.x - o.x) * inv.x, bx = (c0.y - o.x) * inv.x;
    const float ay = (c0.z - o.y) * inv.y, by = (c0.w - o.y) * inv.y;
    const float az = (c1.x - o.z) * inv.z, bz = (c1.y - o.z) * inv.z;
    // tmin folded into two of the three entry distances: every face reads at least one of them
    const float nx = fmaxf(tmin, fminf(ax, bx)), ny = fmaxf(tmin, fminf(ay, by)), nz = fminf(az, bz);
    const float fx = fmaxf(ax, bx), fy = fmaxf(ay, by), fz = fmaxf(az, bz);
    auto face = [&](float plane_lo, float plane_hi, float origin, float rcp, float n1, float n2, float f1, float f2) {
        const float a = (plane_lo - origin) * rcp, b = (plane_hi - origin) * rcp;
        const float lo = fmaxf(fmaxf(fminf(a, b), n1), n2);
        const float hi = fminf(fminf(fmaxf(a, b), f1), f2);
        unsigned long long hit;
        asm("v_cmp_nlt_f32_e64 %0, %1, %2" : "=s"(hit) : "v"(hi), "v"(lo));
        const unsigned long long carry = shift_in(lo_word, hit);
        if constexpr (WIDE) (void)shift_in(hi_word, carry);
    };
    const NativeQuad tx = rec[2], ty = rec[3], tz = rec[4];
    if (present & 1) face(tx.x, tx.y, o.x, inv.x, ny, nz, fy, fz);
    if (present & 2) face(tx.z, tx.w, o.x, inv.x, ny, nz, fy, fz);
    if (present & 4) face(ty.x, ty.y, o.y, inv.y, nx, nz, fx, fz);
    if (present & 8) face(ty.z, ty.w, o.y, inv.y, nx, nz, fx, fz);
    if (present & 16) face(tz.x, tz.y, o.z, inv.z, nx, ny, fx, fy);
    if (present & 32) face(tz.z, tz.w, o.z, inv.z, nx, ny, fx, fy);
    // The far face of an axis is B for a ray that runs up the axis, A for one that runs down; its slab begins at
    // the plane facing the ray's origin (B's lower, A's upper one).  The host stores -inf / +inf there for an absent
    // face, which then bars nothing.
    const float margin = c1.w;
    const float px = inv.x < 0.0f ? tx.y : tx.z, py = inv.y < 0.0f ? ty.y : ty.z, pz = inv.z < 0.0f ? tz.y : tz.z;
    const float qx = (px - o.x) * inv.x, qy = (py - o.y) * inv.y, qz = (pz - o.z) * inv.z;
    return fminf(fminf(__builtin_fmaf(-margin, fabsf(inv.x), qx), __builtin_fmaf(-margin, fabsf(inv.y), qy)), __builtin_fmaf(-margin, fabsf(inv.z), qz));
}
// A cuboid of a rotated group.  The boxes of a group's slots are not the reference's (they only
// have to contain their triangles), so here the six face boxes are laid out on FOUR planes per axis,
// lo-p < lo+p < hi-p < hi+p: outer interval [lo-p, hi+p], face A [lo-p, lo+p], face B [hi-p, hi+p].
// With LO / HI the entry / exit distance of the outer box and q the distance to a face's inner plane,
// the face whose outer plane the ray meets first is hit iff the box is hit and LO <= q, the other
// one iff the box is hit and q <= HI (distances to ordered planes are ordered, rounding is monotone):
// one inner plane per face instead of a slab test per face.  `pos` = lanes whose reciprocal is
// positive on the axis (face A is then the near one).
// Record, 4 quads: {outer xlo,xhi,ylo,yhi} {outer zlo,zhi, present-face bits, -} {xA,xB,yA,yB inner planes} {zA,zB,margin,-}.
//
// Returns a ray parameter below which no triangle of a FAR face (the face of an axis whose planes the ray
// meets last) can be accepted by hit_tri.  The far face's triangles lie in its slab, which the ray enters at the
// slab's plane facing the origin (parameter qf).  Two cases, `margin` (from the host) covering both:
//   * the ray is not nearly parallel to the face: the computed t differs from the exact one by a rounding error
//     that, measured along the face's axis, is at most about 20 x 2^-24 x reach / sin(angle between the triangle's
//     edges), reach = the largest distance between a ray origin and a point of the scene (the 1/|d_axis| of the
//     error of t cancels against the d_axis that turns t into a distance along the axis); margin is 1024 times that;
//   * the ray is so nearly parallel that the computed determinant is mostly rounding error (this includes the
//     directions whose reciprocal is the substitute of aabb.cuh:140): a safe distance > 0 puts the origin more than
//     `margin` away from the slab, and then one of the computed barycentrics is at least margin / (50 x 2^-24 x edge
//     length) >> 1 in magnitude: the candidate is rejected whatever its t.
// So safe = min over the axes of qf - margin * |1/d_axis|: a candidate whose t cannot be below `safe` cannot
// replace a closest hit that is, and traverse_flat() tests the far faces last and only where the closest hit so
// far is not already in front of them.  margin = +inf gives -inf: nothing is skipped.
template <bool WIDE>
__device__ __forceinline__ float flat_cuboid_shared(ConstQuadPtr rec, V3 o, V3 inv, float tmin, unsigned long long pos_x, unsigned long long pos_y,
                                                    unsigned long long pos_z, unsigned& lo_word, unsigned& hi_word) {
    const NativeQuad c0 = rec[0], c1 = rec[1], c2 = rec[2], c3 = rec[3];
    const int present = __float_as_int(c1.z);
    const float ax = (c0.x - o.x) * inv.x, bx = (c0.y - o.x) * inv.x;
    const float ay = (c0.z - o.y) * inv.y, by = (c0.w - o.y) * inv.y;
    const float az = (c1.x - o.z) * inv.z, bz = (c1.y - o.z) * inv.z;
    const float LO = fmaxf(tmin, fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz)));
    const float HI = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    // a missed box: LO and HI become NaN, which fails both ordered comparisons below
    const bool box = !(HI < LO);
    const float lo_box = box ? LO : __builtin_nanf(""), hi_box = box ? HI : __builtin_nanf("");
    auto axis = [&](float plane_a, float plane_b, float origin, float rcp, unsigned long long pos, int bits) {
        const float ta = (plane_a - origin) * rcp, tb = (plane_b - origin) * rcp;
        const float qn = fminf(ta, tb), qf = fmaxf(ta, tb);
        unsigned long long near_hit, far_hit;
        asm("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(near_hit) : "v"(lo_box), "v"(qn));
        asm("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(far_hit) : "v"(qf), "v"(hi_box));
        const unsigned long long swap = (near_hit ^ far_hit) & pos;      // face A: pos ? near : far; face B the other way round
        if (bits & 1) {
            const unsigned long long carry = shift_in(lo_word, far_hit ^ swap);
            if constexpr (WIDE) (void)shift_in(hi_word, carry);
        }
        if (bits & 2) {
            const unsigned long long carry = shift_in(lo_word, near_hit ^ swap);
            if constexpr (WIDE) (void)shift_in(hi_word, carry);
        }
        return qf;
    };
    float qx, qy, qz;
    if (present == 63) {                                                 // the usual case without the six tests
        qx = axis(c2.x, c2.y, o.x, inv.x, pos_x, 3);
        qy = axis(c2.z, c2.w, o.y, inv.y, pos_y, 3);
        qz = axis(c3.x, c3.y, o.z, inv.z, pos_z, 3);
    } else {
        qx = axis(c2.x, c2.y, o.x, inv.x, pos_x, present);
        qy = axis(c2.z, c2.w, o.y, inv.y, pos_y, present >> 2);
        qz = axis(c3.x, c3.y, o.z, inv.z, pos_z, present >> 4);
    }
    const float margin = c3.z;
    return fminf(fminf(__builtin_fmaf(-margin, fabsf(inv.x), qx), __builtin_fmaf(-margin, fabsf(inv.y), qy)), __builtin_fmaf(-margin, fabsf(inv.z), qz));
}
// Box tests of one group: its cuboids, then its other slots.  SHARED: the group's cuboids are of the
// four-planes-per-axis kind (rotated groups), 4 quads each; else 5 quads each.
template <bool WIDE, bool SHARED>
__device__ __forceinline__ float flat_group_boxes(const FlatView& F, int g, V3 o, V3 inv, float tmin, unsigned& lo_word, unsigned& hi_word) {
    ConstQuadPtr boxes = (ConstQuadPtr)F.boxes;
    const int first = g ? F.cuboid_end[g - 1] : 0, last = F.cuboid_end[g];
    float safe = -INFINITY;                                              // of the group's last cuboid (flat_cuboid_shared())
    if constexpr (SHARED) {
        if (first < last) {
            ConstQuadPtr cuboids = boxes + F.cuboid_quad0 + 5 * F.cuboid_end[0];
            unsigned long long pos_x, pos_y, pos_z;
            asm("v_cmp_lt_f32_e64 %0, 0, %1" : "=s"(pos_x) : "v"(inv.x));
            asm("v_cmp_lt_f32_e64 %0, 0, %1" : "=s"(pos_y) : "v"(inv.y));
            asm("v_cmp_lt_f32_e64 %0, 0, %1" : "=s"(pos_z) : "v"(inv.z));
            for (int c = first; c < last; c++) safe = flat_cuboid_shared<WIDE>(cuboids + 4 * (c - F.cuboid_end[0]), o, inv, tmin, pos_x, pos_y, pos_z, lo_word, hi_word);
        }
    } else {
        ConstQuadPtr cuboids = boxes + F.cuboid_quad0;
        for (int c = first; c < last; c++) safe = flat_cuboid<WIDE>(cuboids + 5 * c, o, inv, tmin, lo_word, hi_word);
    }
    flat_box_range<WIDE>(boxes, g ? F.box_end[g - 1] : 0, F.box_end[g], o, inv, tmin, lo_word, hi_word);
    return safe;
}
// The six face bits xA xB yA yB zA zB (xA the highest) of a rotated group's cuboid that are FAR faces for a ray
// with reciprocal direction `inv`: face B of an axis the ray runs up, face A of one it runs down.
__device__ __forceinline__ unsigned far_faces(V3 inv) {
    const unsigned nx = __float_as_uint(inv.x) >> 31, ny = __float_as_uint(inv.y) >> 31, nz = __float_as_uint(inv.z) >> 31;
    return 21u + (nx << 4) + (ny << 2) + nz;
}

struct TriEval {
    bool ok;
    float t, u, v;
};
// hit_tri (hittable.cuh:314-335) without early exits: same tests, same arithmetic.
__device__ __forceinline__ TriEval tri_eval(V3 o, V3 d, V3 point, V3 e1, V3 e2, float tmin, float tmax, int fast) {
    TriEval r;
    V3 pvec = cross(d, e2);
    float det = dot(e1, pvec);
    bool ok = !(fabsf(det) < 1e-8f);
    float inv_det = reciprocal(det, fast);                       // never read for a rejected candidate
    V3 tvec = o - point;
    r.u = dot(tvec, pvec) * inv_det;
    ok = ok && !(r.u < 0.0f || r.u > 1.0f);
    V3 qvec = cross(tvec, e1);
    r.v = dot(d, qvec) * inv_det;
    ok = ok && !(r.v < 0.0f || r.u + r.v > 1.0f);
    r.t = dot(e2, qvec) * inv_det;
    r.ok = ok && (tmin <= r.t && r.t <= tmax);
    return r;
}

// Which kinds of slot a round of the primitive loop may meet (a compile-time tag: the code of the other kind is left out)
template <int K> struct SlotKind { static constexpr int value = K; };
enum { kSlotAny = 0, kSlotSphere = 1, kSlotTri = 2 };

template <int GROUPS, int FAST, int FEAT = kFeatAll>
__device__ __forceinline__ Hit traverse_flat(const FlatLds& L, const FlatView& F, const Ray& r, float tmin, float tmax, SectionClock& clk) {
    ConstFloatPtr gx = (ConstFloatPtr)F.group_xf;
    unsigned cand_lo = 0, cand_hi = 0;
    const bool wide = (FEAT & kFeatWide) && F.slot_count > 32;   // uniform over the launch
    const V3 inv0 = slab_reciprocal(r.d, FAST);
    float safe0;
    if (wide) safe0 = flat_group_boxes<true, false>(F, 0, r.o, inv0, tmin, cand_lo, cand_hi);
    else safe0 = flat_group_boxes<false, false>(F, 0, r.o, inv0, tmin, cand_lo, cand_hi);
    // far faces of the world group's last cuboid, at their final bit positions
    const unsigned world_faces = F.world_face_a[0] | F.world_face_b[0] | F.world_face_a[1] | F.world_face_b[1] | F.world_face_a[2] | F.world_face_b[2];
    unsigned far0 = 0;
    if (world_faces) far0 = (inv0.x < 0.0f ? F.world_face_a[0] : F.world_face_b[0]) | (inv0.y < 0.0f ? F.world_face_a[1] : F.world_face_b[1]) |
                            (inv0.z < 0.0f ? F.world_face_a[2] : F.world_face_b[2]);
    // rays of the rotated groups (hittable.cuh:429-430), transform data through scalar loads
    V3 o1 = r.o, d1 = r.d, o2 = r.o, d2 = r.d;
    // far faces of the groups' cuboids: their candidate bits, and the ray parameter in front of which none of them can be hit
    unsigned far = 0;
    float safe1 = -INFINITY, safe2 = -INFINITY;
    if constexpr (GROUPS >= 1) {
        float X[15];
#pragma unroll
        for (int k = 0; k < 15; k++) X[k] = gx[k];
        to_object_space(X, r.o, r.d, o1, d1);
        const V3 inv1 = slab_reciprocal(d1, FAST);
        if (wide) safe1 = flat_group_boxes<true, true>(F, 1, o1, inv1, tmin, cand_lo, cand_hi);
        else safe1 = flat_group_boxes<false, true>(F, 1, o1, inv1, tmin, cand_lo, cand_hi);
        if (F.far_shift[1] >= 0) far = far_faces(inv1);
    }
    if constexpr (GROUPS >= 2) {
        float X[15];
#pragma unroll
        for (int k = 0; k < 15; k++) X[k] = gx[kXformFloats + k];
        to_object_space(X, r.o, r.d, o2, d2);
        const V3 inv2 = slab_reciprocal(d2, FAST);
        if (wide) safe2 = flat_group_boxes<true, true>(F, 2, o2, inv2, tmin, cand_lo, cand_hi);
        else safe2 = flat_group_boxes<false, true>(F, 2, o2, inv2, tmin, cand_lo, cand_hi);
        // (group 2's boxes were tested after group 1's: its bits lie below, and group 1's have moved up by as many)
        if (F.far_shift[1] >= 0) far <<= (F.far_shift[1] - (F.far_shift[2] >= 0 ? F.far_shift[2] : 0));
        if (F.far_shift[2] >= 0) far |= far_faces(inv2);
        if (F.far_shift[2] > 0) far <<= F.far_shift[2];
    } else if constexpr (GROUPS == 1) {
        if (F.far_shift[1] > 0) far <<= F.far_shift[1];
    }

    Hit best;
    best.t = tmax; best.obj = -1; best.bu = 0.0f; best.bv = 0.0f;
    bool best_tri = false;
    float no_lo = -INFINITY;                                   // (no box entry distances in this mode: consider())
    CRT_CLOCK(clk, kSecPrim, cand_lo);
    // One candidate slot (bit `s` of the candidate words = position s of the LDS copy, stored last slot first).
    auto test_slot = [&](int s, auto kind) {
        constexpr int KIND = decltype(kind)::value;
        CRT_PROF(kProfPrim);
        const float4* rec = L.slots + s * kSlotQuads;
        // record: {meta, transform, A.point.xy} {A.point.z, A.edge1} {A.edge2, B.point.x} {B.point.yz, B.edge1.xy} {B.edge1.z, B.edge2}
        // (sphere: {meta, transform, centre.xy} {centre.z, radius, centre direction.xy} {centre direction.z, -, -, -}
        //          {anchor, translation.x} {translation.yz, -, -}).
        // Triangle B's part is only loaded after triangle A has been dealt with: eight registers fewer at the loop's peak.
        const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
        const int meta = __float_as_int(q0.x), xf = __float_as_int(q0.y);
        const int rank_a = meta & 63, rank_b = (meta >> 6) & 63, group = (meta >> 15) & 3;
        const bool has_b = (meta >> 12) & 1, is_tri = ((meta >> 13) & 3) == 2;
        if (KIND == kSlotTri || (KIND == kSlotAny && (!(FEAT & kFeatSphere) || is_tri))) {
            // the ray of the slot's group, component by component (selects in registers: written with
            // branches, the compiler may turn the three rays into an indexed array in scratch memory)
            V3 o = r.o, d = r.d;
            if constexpr (GROUPS >= 1) {
                const bool g1 = group == 1;
                o.x = g1 ? o1.x : o.x; o.y = g1 ? o1.y : o.y; o.z = g1 ? o1.z : o.z;
                d.x = g1 ? d1.x : d.x; d.y = g1 ? d1.y : d.y; d.z = g1 ? d1.z : d.z;
            }
            if constexpr (GROUPS >= 2) {
                const bool g2 = group == 2;
                o.x = g2 ? o2.x : o.x; o.y = g2 ? o2.y : o.y; o.z = g2 ? o2.z : o.z;
                d.x = g2 ? d2.x : d.x; d.y = g2 ? d2.y : d.y; d.z = g2 ? d2.z : d.z;
            }
            if ((FEAT & kFeatSlotXf) && xf >= 0) { CRT_PROF(kProfXform); to_object_space(xform_record<FEAT>(L.xforms, xf), (FEAT & kFeatPlainXf) && (xf & kXfIdentityRotation), r.o, r.d, o, d); }
            const TriEval A = tri_eval(o, d, v3(q0.z, q0.w, q1.x), v3(q1.y, q1.z, q1.w), v3(q2.x, q2.y, q2.z), tmin, tmax, FAST);
            consider<!(FEAT & kFeatSphere)>(best, best_tri, no_lo, A.ok, A.t, -INFINITY, rank_a, true, A.u, A.v);
            float bx = q2.w;
            asm volatile("" : "+v"(bx) : : "memory");              // keeps the two loads below behind triangle A
            const float4 q3 = rec[3], q4 = rec[4];
            const TriEval B = tri_eval(o, d, v3(bx, q3.x, q3.y), v3(q3.z, q3.w, q4.x), v3(q4.y, q4.z, q4.w), tmin, tmax, FAST);
            consider<!(FEAT & kFeatSphere)>(best, best_tri, no_lo, B.ok && has_b, B.t, -INFINITY, rank_b, true, B.u, B.v);
        } else if constexpr ((FEAT & kFeatSphere) != 0 && KIND != kSlotTri) {
            // Sphere slot (always in the world group).  Its transform is nearly always the rotation-free one the reference
            // gives every sphere (anchor = centre): anchor and translation then come with the slot record.
            V3 o = r.o, d = r.d;
            if ((FEAT & kFeatSlotXf) && xf >= 0) {
                if ((FEAT & kFeatPlainXf) && (xf & kXfIdentityRotation)) {
                    const float4 q3 = rec[3], q4 = rec[4];
                    const V3 anchor = v3(q3.x, q3.y, q3.z), trans = v3(q3.w, q4.x, q4.y);
                    o = (r.o - anchor - trans) + anchor;           // hittable.cuh:429 with the identity matrix
                } else {
                    CRT_PROF(kProfXform);
                    to_object_space(xform_record<FEAT>(L.xforms, xf), false, r.o, r.d, o, d);
                }
            }
            // The root the reference would take does not depend on the upper bound: the near root if it is
            // beyond tmin, else the far one.
            V3 center = v3(q0.z, q0.w, q1.x) + v3(q1.z, q1.w, q2.x) * r.time;
            V3 oc = o - center;
            float a = length_squared(d);
            float h = dot(oc, d);
            float c = length_squared(oc) - q1.y * q1.y;
            float disc = h * h - a * c;
            float sq = sqrtf(disc);
            const float ra = FAST ? reciprocal(a, 1) : 0.0f;
            float r1 = FAST ? quotient(-h - sq, a, ra) : (-h - sq) / a, r2 = FAST ? quotient(-h + sq, a, ra) : (-h + sq) / a;
            float root = (tmin < r1 && r1 < tmax) ? r1 : r2;
            bool ok = !(disc < 0.0f) && (tmin < root && root < tmax);
            consider(best, best_tri, no_lo, ok, root, -INFINITY, rank_a, false, 0.0f, 0.0f);
        }
    };
    // Any order of the tests gives the same hit (consider()), so the candidates are taken in three phases, each a
    // loop of its own in which every lane runs the same code: sphere slots; triangle slots; the far faces of the
    // rotated groups' cuboids, and those only for rays whose closest hit so far is not already in front of them
    // (flat_cuboid_shared(): a ray that crosses a box meets two of its faces, the far one to no purpose).
    far |= far0;
    far &= cand_lo;
    cand_lo ^= far;
#if defined(CRT_PROFILE) && !defined(CRT_PROFILE_CLOCK)
    {   // candidates per lane (slots 16-23), their maximum over the wave (24-31), and which slots the lanes with three or more hold (64-127)
        const int n = __builtin_popcount(cand_lo) + __builtin_popcount(cand_hi) + __builtin_popcount(far);
        atomicAdd(&g_prof[16 + (n < 7 ? n : 7)], 1ull);
        int most = n;
        for (int off = 32; off; off >>= 1) most = max(most, __shfl_xor(most, off));
        if ((int)(threadIdx.x & 63) == __builtin_ctzll(lanes_where(true))) atomicAdd(&g_prof[24 + (most < 7 ? most : 7)], 1ull);
        if (n >= 3)
            for (unsigned m = cand_lo | far; m; m &= m - 1u) atomicAdd(&g_prof[64 + __builtin_ctz(m)], 1ull);
    }
#endif
    if constexpr ((FEAT & kFeatSphere) != 0) {
        unsigned spheres = cand_lo & F.sphere_bits;
        cand_lo ^= spheres;
        while (spheres) {
            const int bit = __builtin_ctz(spheres);
            spheres &= spheres - 1u;
            test_slot(bit, SlotKind<kSlotSphere>());
        }
    }
    // A lane turns to its far faces when its own near candidates are used up.
    const unsigned faces1 = F.far_shift[1] >= 0 ? 63u << F.far_shift[1] : 0u, faces2 = F.far_shift[2] >= 0 ? 63u << F.far_shift[2] : 0u;
    auto next_slot = [&]() {
        int bit;
        if (!(FEAT & kFeatWide) || cand_lo) { bit = __builtin_ctz(cand_lo); cand_lo &= cand_lo - 1u; }
        else { bit = 32 + __builtin_ctz(cand_hi); cand_hi &= cand_hi - 1u; }
        // (beyond 32 slots the high word may hold spheres)
        if constexpr ((FEAT & kFeatWide) != 0) test_slot(bit, SlotKind<kSlotAny>());
        else test_slot(bit, SlotKind<kSlotTri>());
    };
    while (cand_lo | cand_hi | far) {
        if (cand_lo == 0u && far != 0u) {
            if (best.t < safe0) far &= ~world_faces;
            if (GROUPS >= 1 && best.t < safe1) far &= ~faces1;
            if (GROUPS >= 2 && best.t < safe2) far &= ~faces2;
            cand_lo = far;
            far = 0u;
        }
        if (cand_lo | cand_hi) next_slot();
    }
    return best;
}

// ------------------------------------------------------------------ textures ------------
// Software tex2D: wrap addressing, bilinear, normalised coordinates, 8-bit -> [0,1]
// (sampler state at texture.cuh:212-217).
// i modulo n for -n <= i < 2n.  tex2d() only asks for floor(x) and floor(x) + 1 with x in
// [-0.5, n - 0.5], i.e. -1 .. n: one conditional add and subtract instead of an integer division.
__device__ __forceinline__ int wrapi(int i, int n) { i = i < 0 ? i + n : i; return i >= n ? i - n : i; }
__device__ __forceinline__ void tex2d(const DevImage& img, float u, float v, float out[4]) {
    float x = (u - floorf(u)) * (float)img.width - 0.5f;
    float y = (v - floorf(v)) * (float)img.height - 0.5f;
    float xf = floorf(x), yf = floorf(y);
    float ax = x - xf, ay = y - yf;
    int x0 = wrapi((int)xf, img.width), x1 = wrapi((int)xf + 1, img.width);
    int y0 = wrapi((int)yf, img.height), y1 = wrapi((int)yf + 1, img.height);
    out[0] = out[1] = out[2] = out[3] = 0.0f;
    if (img.channels == 4) {
        // (the pointer comes out of memory, so the compiler cannot tell that it points to global memory)
        const uint32_t __attribute__((address_space(1)))* p = (const uint32_t __attribute__((address_space(1)))*)img.pixels;
        const uint32_t a = p[(uint32_t)(y0 * img.width + x0)], b = p[(uint32_t)(y0 * img.width + x1)];     // r | g << 8 | b << 16 | a << 24
        const uint32_t c = p[(uint32_t)(y1 * img.width + x0)], d = p[(uint32_t)(y1 * img.width + x1)];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float t00 = (float)((a >> (8 * k)) & 255u) * (1.0f / 255.0f), t10 = (float)((b >> (8 * k)) & 255u) * (1.0f / 255.0f);
            float t01 = (float)((c >> (8 * k)) & 255u) * (1.0f / 255.0f), t11 = (float)((d >> (8 * k)) & 255u) * (1.0f / 255.0f);
            out[k] = (1.0f - ax) * (1.0f - ay) * t00 + ax * (1.0f - ay) * t10 + (1.0f - ax) * ay * t01 + ax * ay * t11;
        }
    } else {
        const uint8_t __attribute__((address_space(1)))* p = (const uint8_t __attribute__((address_space(1)))*)img.pixels;
        float t00 = p[(uint32_t)(y0 * img.width + x0)] * (1.0f / 255.0f), t10 = p[(uint32_t)(y0 * img.width + x1)] * (1.0f / 255.0f);
        float t01 = p[(uint32_t)(y1 * img.width + x0)] * (1.0f / 255.0f), t11 = p[(uint32_t)(y1 * img.width + x1)] * (1.0f / 255.0f);
        out[0] = (1.0f - ax) * (1.0f - ay) * t00 + ax * (1.0f - ay) * t10 + (1.0f - ax) * ay * t01 + ax * ay * t11;
    }
}

// _texture::value (texture.cuh:313-324).  Checker recursion (texture.cuh:267-274) becomes a
// walk down the texture indices; the host guarantees the texture graph is acyclic.
template <int FEAT = kFeatAll>
__device__ __forceinline__ V3 texture_value(const DevScene& sc, int idx, float u, float v, V3 p) {
    float4 t0 = *at32(sc.texs, idx * 2);
    if (!(FEAT & kFeatShading)) return v3(t0.y, t0.z, t0.w);     // every texture of the scene is one colour
    int type = __float_as_int(t0.x);
    for (int guard = 0; type == 1 && guard < sc.texture_count; guard++) {
        float4 t1 = *at32(sc.texs, idx * 2 + 1);
        int xi = (int)floorf(t1.z * p.x), yi = (int)floorf(t1.z * p.y), zi = (int)floorf(t1.z * p.z);
        bool even = (xi + yi + zi) % 2 == 0;
        idx = even ? __float_as_int(t1.y) : __float_as_int(t1.x);
        t0 = *at32(sc.texs, idx * 2);
        type = __float_as_int(t0.x);
    }
    if (type == 0) return v3(t0.y, t0.z, t0.w);
    if (type == 2) {
        float4 t1 = *at32(sc.texs, idx * 2 + 1);
        const DevImage img = *at32(sc.images, __float_as_int(t1.w));
        u = fminf(fmaxf(u, 0.0f), 1.0f);
        v = fminf(fmaxf(v, 0.0f), 1.0f);
        float px[4];
        tex2d(img, u, v, px);
        if (img.channels == 1) return v3(px[0], px[0], px[0]);
        if (img.channels == 4) return v3(px[0], px[1], px[2]);
        return v3(1.0f, 0.0f, 1.0f);
    }
    return v3(0.0f, 0.0f, 0.0f);
}

// ------------------------------------------------------------------ hit record ----------
struct Surface {
    V3 p, normal;
    float u, v;
    bool front;
    int mat;
    bool has_unit;       // unit_normal is unit(normal), taken from the shading record instead of three divisions
    V3 unit_normal;
};

// Rebuild the record of the closest hit: hit_tri / hit_sphere tail, object->world transform,
// normal map (hittable.cuh:292-299, 338-356, 367-386, 438-444).
// ALWAYS_UV = false: the (u, v) of a sphere, an acosf and an atan2f, are only computed for objects
// on which something reads them (flag bit 1 of the shading record: an image texture under the
// material or a normal map); the render never looks at them otherwise.
template <bool ALWAYS_UV, int FEAT = kFeatAll, int FAST = 0>
__device__ __forceinline__ Surface make_surface(const DevScene& sc, const Ray& r, const Hit& h) {
    const float4* pq = at32(sc.prims, h.obj * kPrimQuads);
    const float4* sq = at32(sc.shade, h.obj * kShadeQuads);
    float4 q0 = pq[0], q1 = pq[1], q2 = pq[2];
    float4 s0 = sq[0], s1 = sq[1];
    const int xf = (FEAT & kFeatXform) ? __float_as_int(q2.y) : -1, type = __float_as_int(q2.z);
    const int packed = __float_as_int(s1.w);
    // without kFeatShading every object is a flat triangle without a normal map (flag bit 2) or a sphere
    const int flags = (FEAT & kFeatShading) ? (packed & 7) : 4;
    const int normal_idx = (FEAT & kFeatShading) ? (packed >> 3) - 1 : -1;
    const float* X = xform_record<FEAT>(sc.xforms, xf >= 0 ? xf : 0);
    const bool plain = (FEAT & kFeatPlainXf) && xf >= 0 && (xf & kXfIdentityRotation);
    V3 o = r.o, d = r.d;
    if (xf >= 0) to_object_space(X, plain, r.o, r.d, o, d);
    Surface s;
    s.has_unit = false;
    s.unit_normal = v3(0.0f, 0.0f, 0.0f);
    s.mat = __float_as_int(s0.w);
    V3 p = o + d * h.t;
    V3 tangent = v3(0, 0, 0), bitangent = v3(0, 0, 0);
    if (!(FEAT & kFeatSphere) || type == 2) {
        float w = 1.0f - h.bu - h.bv;
        // vertex uvs, third vertex normal, tangent frame: only where something reads them
        float4 c0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), c1 = c0, c2 = c0, c3 = c0;
        if (ALWAYS_UV || (flags & 3) || normal_idx >= 0) {
            const float4* cq = at32(sc.shade_cold, h.obj * kShadeColdQuads);
            c0 = cq[0]; c1 = cq[1]; c2 = cq[2]; c3 = cq[3];
        }
        V3 n;
        if (flags & 1) n = unit_vector<FAST>(w * v3(s0.x, s0.y, s0.z) + h.bu * v3(s1.x, s1.y, s1.z) + h.bv * v3(c0.x, c0.y, c0.z));
        else n = v3(s0.x, s0.y, s0.z);
        s.front = dot(d, n) < 0.0f;
        s.normal = s.front ? n : -n;
        if (flags & 4) {
            // flat triangle without a normal map: the host has stored unit(world normal) in the slot of
            // the second vertex normal; negation commutes with every step of rotating and normalising
            const V3 pre = v3(s1.x, s1.y, s1.z);
            s.unit_normal = s.front ? pre : -pre;
            s.has_unit = true;
        }
        s.u = 0.0f; s.v = 0.0f;
        if (FEAT & kFeatShading) {
            s.u = w * c1.x + h.bu * c1.z + h.bv * c2.x;
            s.v = w * c1.y + h.bu * c1.w + h.bv * c2.y;
            tangent = v3(c2.z, c2.w, c3.x);
            bitangent = v3(c3.y, c3.z, c3.w);
        }
    } else {
        V3 c0 = v3(q0.x, q0.y, q0.z), cd = v3(q1.x, q1.y, q1.z);
        float radius = q0.w;
        V3 center = c0 + cd * r.time;
        V3 outward = divide<FAST>(p - center, radius);
        s.front = dot(d, outward) < 0.0f;
        s.normal = s.front ? outward : -outward;
        s.u = 0.0f; s.v = 0.0f;
        if (ALWAYS_UV || ((FEAT & kFeatShading) && (flags & 2))) {
            float theta = crt_det::acos(-outward.y);                     // get_sphere_uv
            float phi = crt_det::atan2(-outward.z, outward.x) + kPi;
            s.u = phi / (2 * kPi);
            s.v = theta / kPi;
        }
        if (normal_idx >= 0) {                                           // get_tangent_bitangent: centre at rec.t
            V3 local_p = p - (c0 + cd * h.t);
            float ph = crt_det::atan2(-local_p.z, local_p.x) + kPi;
            float sin_ph, cos_ph;
            crt_det::sincos(ph, sin_ph, cos_ph);
            tangent = unit(v3(-radius * sin_ph, 0.0f, radius * cos_ph));
            bitangent = unit(cross(s.normal, tangent));
        }
    }
    if (xf >= 0) {
        V3 anchor = v3(X[9], X[10], X[11]), trans = v3(X[12], X[13], X[14]);
        if (plain) {                                                     // hittable.cuh:439-441 with the identity matrix
            p = (p - anchor) + anchor;
            p = p + trans;
        } else {
            p = matvec(X + 15, p - anchor) + anchor;
            p = p + trans;
            s.normal = matvec(X + 15, s.normal);
        }
    }
    s.p = p;
    if (normal_idx >= 0) {                                               // apply_normal_map
        float4 t1 = *at32(sc.texs, normal_idx * 2 + 1);
        const DevImage img = *at32(sc.images, __float_as_int(t1.w));
        V3 pixel = v3(0.0f, 0.0f, 1.0f);
        float px[4];
        if (img.channels == 1) {
            tex2d(img, s.u, s.v, px);
            float value = px[0] * 2.0f - 1.0f;
            pixel = v3(value, value, value);
        } else if (img.channels == 4) {
            tex2d(img, s.u, s.v, px);
            pixel = v3(px[0] * 2.0f - 1.0f, px[1] * 2.0f - 1.0f, px[2] * 2.0f - 1.0f);
        }
        V3 wn = v3(tangent.x * pixel.x + bitangent.x * pixel.y + s.normal.x * pixel.z,
                   tangent.y * pixel.x + bitangent.y * pixel.y + s.normal.y * pixel.z,
                   tangent.z * pixel.x + bitangent.z * pixel.y + s.normal.z * pixel.z);
        s.normal = unit_vector<FAST>(wn);
    }
    return s;
}

// ------------------------------------------------------------------ materials -----------
__device__ __forceinline__ V3 reflect(V3 v, V3 n) { return v - 2.0f * dot(v, n) * n; }
__device__ __forceinline__ V3 refract(V3 uv, V3 n, float ni_over_nt) {          // vec3.cuh:166-171
    float cos_theta = fminf(dot(-uv, n), 1.0f);
    V3 perp = ni_over_nt * (uv + cos_theta * n);
    V3 parallel = -sqrtf(fabsf(1.0f - length_squared(perp))) * n;
    return perp + parallel;
}
// (1-cos)^5 of the Schlick terms (material.cuh:152,186) as a multiplication chain.
__device__ __forceinline__ float pow5(float x) { float x2 = x * x; return x2 * x2 * x; }

// Wave-cooperative form of the rejection samplers vec3::random_unit_vector (vec3.cuh:204-212,
// in_sphere = false) and vec3::random_in_unit_sphere (vec3.cuh:235-241, in_sphere = true).
//
// A lane accepts its first candidate with probability pi/6, so a loop that runs until the last
// of 64 lanes has accepted takes about 7 rounds with a quarter of the lanes active.  Candidate c
// of a lane is a pure function of its stream position advanced by 3c draws, so ANY lane can evaluate it:
// after every lane has tried its own first candidate, the lanes still pending publish their
// stream position in LDS and all 64 lanes evaluate the next m candidates of each of them in
// one round (m = 64 / pending, at most 8).  The pending lane takes the FIRST accepted candidate
// in stream order and advances its counter by what the sequential loop would have consumed,
// so the result and the draw count are identical to the sequential sampler's.
// Must be called by the whole wave in uniform control flow; `pub` (128 words) and `res`
// (192 floats) are this wave's LDS scratch.
__device__ __forceinline__ V3 candidate_point(uint32_t pos) {
    Rng r;
    r.pos = pos;
    return r.range3(-1.0f, 1.0f);
}
__device__ __forceinline__ bool sample_accepted(V3 p, bool in_sphere) {
    const float lensq = length_squared(p);
    return in_sphere ? (lensq < 1.0f) : (1e-30f < lensq && lensq <= 1.0f);
}
// Lanes exchange data through LDS inside one wave: LDS executes a wave's instructions in order,
// so all that is needed between a store and another lane's load is that the compiler keeps
// them in place (fence + wave barrier).  The pointers are plain, not volatile: a volatile access
// through a generic pointer is compiled to a flat, system-coherent load or store with a full
// wait after each, ten of them in a row per round.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// (IN_SPHERE_TOO = false: every caller asks for unit vectors, and the flag that tells the two kinds apart is not published)
template <int FAST = 0, bool IN_SPHERE_TOO = true>
__device__ __forceinline__ V3 coop_sphere_sample(bool need, bool in_sphere, Rng& rng, uint32_t* pub, float* res, int lane) {
    if constexpr (!IN_SPHERE_TOO) in_sphere = false;
    V3 out = v3(0.0f, 0.0f, 0.0f);
    bool pending = false;
    if (need) {
        CRT_PROF(kProfReject);
        out = candidate_point(rng.pos);
        rng.skip(3);
        pending = !sample_accepted(out, in_sphere);
    }
    unsigned long long mask = lanes_where(pending);
    while (mask) {                                          // wave-uniform
        CRT_PROF(kProfReject);
        const int pending_count = __builtin_popcountll(mask);
        const int log2m = pending_count <= 8 ? 3 : (pending_count <= 16 ? 2 : (pending_count <= 32 ? 1 : 0));
        const int m = 1 << log2m;
        const int rank = __builtin_popcountll(mask & ((1ull << lane) - 1ull));
        if (pending) {
            pub[2 * rank] = rng.pos;
            if constexpr (IN_SPHERE_TOO) pub[2 * rank + 1] = in_sphere ? 1u : 0u;
        }
        wave_lds_sync();
        const int target = lane >> log2m, c = lane & (m - 1);
        bool ok = false;
        if (target < pending_count) {
            const uint32_t at = pub[2 * target];
            const V3 p = candidate_point(at + (uint32_t)c * (3u * kRngStep));
            ok = sample_accepted(p, IN_SPHERE_TOO && pub[2 * target + 1] != 0u);
            res[3 * lane] = p.x; res[3 * lane + 1] = p.y; res[3 * lane + 2] = p.z;
        }
        const unsigned long long okmask = lanes_where(ok);
        wave_lds_sync();
        if (pending) {
            const unsigned sub = (unsigned)(okmask >> (rank << log2m)) & ((1u << m) - 1u);
            if (sub) {
                const int first = __builtin_ctz(sub);
                const int src = (rank << log2m) + first;
                out = v3(res[3 * src], res[3 * src + 1], res[3 * src + 2]);
                rng.skip(3u * (uint32_t)(first + 1));
                pending = false;
            } else {
                rng.skip(3u * (uint32_t)m);
            }
        }
        wave_lds_sync();                                    // the next round overwrites pub and res
        mask = lanes_where(pending);
    }
    if (need && !in_sphere) out = unit_vector<FAST>(out);
    return out;
}

// Material record, two quads.  First: {kind, texture, fuzz, ior}; a lambertian material or light whose texture is
// one colour carries that colour instead: {kind | kMatOneColour, r, g, b} (no second lookup for what most hits need).
// Second: metal {albedo r, g, b, -}; dielectric {1 / ior, r0 of a front-face hit, r0 of a back-face hit, -}, the
// Schlick constants ((1 - ratio) / (1 + ratio))^2 of material.cuh:183-187 for ratio = 1 / ior and ratio = ior,
// evaluated on the host with the same float operations.
constexpr int kMatOneColour = 16;
__device__ __forceinline__ int material_kind(float4 m0) { return __float_as_int(m0.x) & 15; }

// material::scatter (material.cuh:359-372) with the sphere sample already drawn (see
// coop_sphere_sample): `rv` is the random unit vector for a lambertian surface, the in-sphere
// vector for a metal.  Returns false when the material does not scatter (light, or a metal
// bounce that points into the surface).
template <int FEAT = kFeatAll, int FAST = 0>
__device__ __forceinline__ bool scatter_with(const DevScene& sc, const Ray& in, const Surface& s, float4 m0, V3 rv, Rng& rng, V3& attenuation, Ray& out) {
    const int type = material_kind(m0);
    out.time = in.time;
    if (type == 0) {                                           // lambertian :290-297
        V3 dir = s.normal + rv;
        if (fabsf(dir.x) < 1e-8f && fabsf(dir.y) < 1e-8f && fabsf(dir.z) < 1e-8f) dir = s.normal;
        out.o = s.p + s.normal * 0.001f;
        out.d = dir;
        if (!(FEAT & kFeatShading) || (__float_as_int(m0.x) & kMatOneColour)) attenuation = v3(m0.y, m0.z, m0.w);
        else attenuation = texture_value<FEAT>(sc, __float_as_int(m0.y), s.u, s.v, s.p);
        return true;
    }
    if (!(FEAT & kFeatMaterials)) return false;                // neither metal nor glass in the scene: a light
    if (type != 1 && type != 2) return false;
    // metal and glass both start from the unit direction and the cosine at the surface
    const float4 m1 = *at32(sc.mats, s.mat * 2 + 1);
    const V3 unit_dir = unit_vector<FAST>(in.d);
    const float cos_theta = fminf(dot(-unit_dir, s.normal), 1.0f);
    out.o = s.p;
    if (type == 1) {                                           // metal :311-322
        V3 reflected = reflect(unit_dir, s.normal);
        out.d = reflected + m0.z * rv;
        V3 albedo = v3(m1.x, m1.y, m1.z);
        attenuation = albedo + (v3(1.0f, 1.0f, 1.0f) - albedo) * pow5(1.0f - cos_theta);
        return dot(out.d, s.normal) > 0.0f;
    }
    // dielectric :336-356
    attenuation = v3(1.0f, 1.0f, 1.0f);
    const float ratio = s.front ? m1.x : m0.w;                 // 1 / ior : ior
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    bool do_reflect = ratio * sin_theta > 1.0f;
    if (!do_reflect) {                                         // the uniform is drawn only when refraction is possible
        const float r0 = s.front ? m1.y : m1.z;
        float reflectance = r0 + (1.0f - r0) * pow5(1.0f - cos_theta);
        do_reflect = reflectance > rng.uniform();
    }
    out.d = do_reflect ? reflect(unit_dir, s.normal) : refract(unit_dir, s.normal, ratio);
    return true;
}

// material::emit (material.cuh:375-381); m0 = the first quad of the hit's material record
template <int FEAT = kFeatAll>
__device__ __forceinline__ V3 emit(const DevScene& sc, const Surface& s, float4 m0) {
    if (material_kind(m0) != 3) return v3(0.0f, 0.0f, 0.0f);
    if (!(FEAT & kFeatShading) || (__float_as_int(m0.x) & kMatOneColour)) return v3(m0.y, m0.z, m0.w);
    return texture_value<FEAT>(sc, __float_as_int(m0.y), s.u, s.v, s.p);
}
// Environment lookup on a miss (render.cuh:77-82)
template <int FEAT = kFeatAll, int FAST = 0>
__device__ __forceinline__ V3 environment(const DevScene& sc, V3 dir) {
    float4 t0 = *at32(sc.texs, sc.background_index * 2);
    if (!(FEAT & kFeatEnvTexture) || __float_as_int(t0.x) == 0) return v3(t0.y, t0.z, t0.w);   // solid colour: (u,v,p) are not read
    V3 p = unit_vector<FAST>(dir);
    float theta = crt_det::acos(-p.y);
    float phi = crt_det::atan2(-p.z, p.x) + kPi;
    float u = phi / (2 * kPi);
    float v = theta / kPi;
    return texture_value(sc, sc.background_index, u, v, p);
}

}  // namespace crt

#endif
