// Small float3 toolkit shared by the host scene layer and the HIP kernels.
//
// Every function spells out its operations in the order the reference's vec3/mat3 do
// (src/vec3.cuh:77-143, src/mat3.cuh:82-86): sums associate left to right, unit() divides by
// the length (no rsqrt).  Built with -ffp-contract=off this makes host geometry and device
// intersection results reproducible bit for bit against a scalar CPU evaluation.
#ifndef CRT_VECMATH_H
#define CRT_VECMATH_H

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define CRT_HD __host__ __device__ __forceinline__
#else
#define CRT_HD inline
#endif

struct V3 {
    float x, y, z;
};

CRT_HD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
CRT_HD V3 v3(const float* p) { V3 r; r.x = p[0]; r.y = p[1]; r.z = p[2]; return r; }
CRT_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
CRT_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
CRT_HD V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
CRT_HD V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
CRT_HD V3 operator*(float s, V3 a) { return v3(a.x * s, a.y * s, a.z * s); }
CRT_HD V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
CRT_HD V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
CRT_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
CRT_HD V3 cross(V3 a, V3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
CRT_HD float length_squared(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
CRT_HD float length(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
CRT_HD V3 unit(V3 a) { return a / length(a); }

// Row-major 3x3 times vector, rows summed left to right (mat3.cuh:82-86).
CRT_HD V3 matvec(const float* m, V3 u) {
    return v3(m[0] * u.x + m[1] * u.y + m[2] * u.z,
              m[3] * u.x + m[4] * u.y + m[5] * u.z,
              m[6] * u.x + m[7] * u.y + m[8] * u.z);
}

#endif
