// acos, atan2, sin and cos in plain float arithmetic, the same on the host and on the GPU.
//
// The reference calls acosf / atan2f for the (u, v) of a sphere and of the environment
// (hittable.cuh:183-186, render.cuh:78-80) and sinf / cosf for a sphere's tangent (hittable.cuh:
// 196-199).  Those come from CUDA's device library there; every math library rounds them a little
// differently (1-2 ulp), so a path through them is not reproducible bit for bit across machines.
// The functions here are single-precision evaluations in the style of the Cephes library (argument
// reduction + short polynomials, a few ulp at worst): written with one rounding per operation, built
// without contraction and calling nothing but IEEE +, -, *, / and sqrt, they give the same bits
// wherever they are compiled.  The GPU kernels use them, and the CPU oracle can be switched to them
// (its default stays the C library, like the reference).
#ifndef CRT_DET_MATH_H
#define CRT_DET_MATH_H

#include <math.h>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define CRT_DM __host__ __device__ __forceinline__
#else
#define CRT_DM inline
#endif

namespace crt_det {

constexpr float kPiF = 3.14159265358979323846f;
constexpr float kHalfPiF = 1.57079632679489661923f;
constexpr float kQuarterPiF = 0.78539816339744830962f;

// asin for |x| <= 0.5: x + x^3 P(x^2)
CRT_DM float asin_core(float x) {
    const float z = x * x;
    float p = 4.2163199048e-2f;
    p = p * z + 2.4181311049e-2f;
    p = p * z + 4.5470025998e-2f;
    p = p * z + 7.4953002686e-2f;
    p = p * z + 1.6666752422e-1f;
    return p * z * x + x;
}

// acos(x) on [-1, 1]; NaN outside (through the square root of a negative number), like the C library
CRT_DM float acos(float x) {
    if (x < -0.5f) return kPiF - 2.0f * asin_core(sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * asin_core(sqrtf(0.5f * (1.0f - x)));
    return kHalfPiF - asin_core(x);
}

// atan2(y, x) in (-pi, pi].  Zeros: atan2(+-0, x > 0) = +-0, atan2(+-0, x < 0) = +-pi,
// atan2(y, 0) = +-pi/2, atan2(0, 0) = 0; NaN in, NaN out.
// The angle of (|x|, |y|) comes from ONE division: the octant decides which ratio is formed
// (|y|/|x| below tan(pi/8), (|y|-|x|)/(|y|+|x|) up to tan(3 pi/8), -|x|/|y| above), then
// atan(t) = t + t^3 P(t^2) on |t| <= tan(pi/8).
CRT_DM float atan2(float y, float x) {
    if (x != x || y != y) return x + y;
    const bool y_negative = __builtin_signbit(y) != 0;
    if (x == 0.0f) {
        if (y == 0.0f) return 0.0f;
        return y_negative ? -kHalfPiF : kHalfPiF;
    }
    if (y == 0.0f) {
        if (x > 0.0f) return y;
        return y_negative ? -kPiF : kPiF;
    }
    const float ax = fabsf(x), ay = fabsf(y);
    float base = 0.0f, num = ay, den = ax;
    if (ay > ax * 2.414213562373095f) {          // tan(3 pi / 8)
        base = kHalfPiF; num = -ax; den = ay;
    } else if (ay > ax * 0.4142135623730950f) {  // tan(pi / 8)
        base = kQuarterPiF; num = ay - ax; den = ay + ax;
    }
    const float t = num / den;
    const float z = t * t;
    float p = 8.05374449538e-2f;
    p = p * z - 1.38776856032e-1f;
    p = p * z + 1.99777106478e-1f;
    p = p * z - 3.33329491539e-1f;
    float a = base + (p * z * t + t);            // angle of (|x|, |y|) in [0, pi/2]
    if (x < 0.0f) a = kPiF - a;
    return y_negative ? -a : a;
}

// sin and cos for |x| up to a few thousand (the kernels pass angles in [0, 2 pi]): the octant of
// x, then x minus the octant's multiple of pi/4 in three steps, then a polynomial in the octant.
CRT_DM void sincos(float x, float& s, float& c) {
    if (!(fabsf(x) < 8192.0f)) {             // NaN, infinities and arguments outside the reduction's range
        s = x - x;                           // NaN for NaN / inf, 0 for a huge finite argument (not used by the kernels)
        c = s + 1.0f;
        return;
    }
    const float ax = fabsf(x);
    int j = (int)(ax * 1.27323954473516f);   // 4 / pi
    j += j & 1;                              // odd octants fold onto the next even one
    const float fj = (float)j;
    float r = ((ax - fj * 0.78515625f) - fj * 2.4187564849853515625e-4f) - fj * 3.77489497744594108e-8f;
    const float z = r * r;
    // sin r and cos r on [-pi/4, pi/4]
    float ps = -1.9515295891e-4f;
    ps = ps * z + 8.3321608736e-3f;
    ps = ps * z - 1.6666654611e-1f;
    const float sin_r = ps * z * r + r;
    float pc = 2.443315711809948e-5f;
    pc = pc * z - 1.388731625493765e-3f;
    pc = pc * z + 4.166664568298827e-2f;
    const float cos_r = pc * z * z - 0.5f * z + 1.0f;
    const int quadrant = (j >> 1) & 3;       // x = r + quadrant * pi / 2
    float sv = (quadrant & 1) ? cos_r : sin_r;
    float cv = (quadrant & 1) ? sin_r : cos_r;
    if (quadrant == 2 || quadrant == 3) sv = -sv;
    if (quadrant == 1 || quadrant == 2) cv = -cv;
    s = x < 0.0f ? -sv : sv;
    c = cv;
}

}  // namespace crt_det

#endif
