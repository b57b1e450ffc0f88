// TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's path-tracing hot path.
//
// Nothing in the product (the HIP library, the Python package, the CLI) may include, link
// or call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
// do, and only as the checker.  It follows JasperJeuken/CudaRayTracer function by function
// (file:line given at each function), keeping the reference's structure on purpose:
// an int stack[256] traversal that tests the four slots in storage order, a slab test that
// divides per box, a per-candidate world->object transform, a hit record that is filled at
// every accepted hit, recursive texture evaluation, and a running mean over samples.
//
// Two things differ from the CUDA build by design and are shared with the HIP library as
// a written specification (not as code):
//   * curand XORWOW is replaced by a keyed counter RNG: state = H(H(1984+pixel) + n*PHI)
//     for sample number n (1-based), draw k (0-based) = H(state + (k+1)*PHI),
//     H = the 32-bit "lowbias32" finaliser, uniform = ((x>>8)+1)*2^-24 in (0,1].
//   * tex2D (wrap, bilinear, normalised float) is restated in software with float weights.
// Parity pinning: the reference has no tests; this oracle is pinned against the six
// published orthographic passes and the published raw materials render (tests/golden).
// The reference itself cannot be compiled here without writing stand-in CUDA headers, so
// it is treated as unbuildable and no reference binary is used.
//
// Build: g++ -O2 -ffp-contract=off -fopenmp -shared -fPIC (see oracle/Makefile).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/crt.h"
#include "../cudaraytracer_amd/csrc/det_math.h"   // plain-float acos/atan2/sin/cos shared with the kernels (oracle_set_det_math)

namespace {

const float k_pi = 3.1415926535897932385f;                       // utility.cuh:27-29
const float k_inf = std::numeric_limits<float>::infinity();      // utility.cuh:31-33

bool g_pow5_mulchain = false;   // see oracle_set_pow5_mulchain()
bool g_det_math = false;        // see oracle_set_det_math()
inline float o_acos(float x) { return g_det_math ? crt_det::acos(x) : acosf(x); }
inline float o_atan2(float y, float x) { return g_det_math ? crt_det::atan2(y, x) : atan2f(y, x); }
inline void o_sincos(float x, float& s, float& c) {
    if (g_det_math) { crt_det::sincos(x, s, c); return; }
    s = sinf(x); c = cosf(x);
}

// ---------------------------------------------------------------- vec3.cuh:19-483 ------
struct vec3 {
    float e[3];
    vec3() : e{0, 0, 0} {}
    vec3(float a, float b, float c) : e{a, b, c} {}
    explicit vec3(const float* p) : e{p[0], p[1], p[2]} {}
    float x() const { return e[0]; }
    float y() const { return e[1]; }
    float z() const { return e[2]; }
    float operator[](int i) const { return e[i]; }
    float& operator[](int i) { return e[i]; }
    vec3 operator-() const { return vec3(-e[0], -e[1], -e[2]); }
    vec3& operator+=(const vec3& o) { e[0] += o.e[0]; e[1] += o.e[1]; e[2] += o.e[2]; return *this; }
    vec3& operator*=(const vec3& o) { e[0] *= o.e[0]; e[1] *= o.e[1]; e[2] *= o.e[2]; return *this; }
    float length_squared() const { return e[0] * e[0] + e[1] * e[1] + e[2] * e[2]; }          // :77
    float length() const { return std::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]); }       // :86
    bool near_zero() const {                                                                  // :96
        const float s = 1e-8;
        return (std::fabs(e[0]) < s) && (std::fabs(e[1]) < s) && (std::fabs(e[2]) < s);
    }
};
inline vec3 operator+(const vec3& u, const vec3& v) { return vec3(u.e[0] + v.e[0], u.e[1] + v.e[1], u.e[2] + v.e[2]); }
inline vec3 operator-(const vec3& u, const vec3& v) { return vec3(u.e[0] - v.e[0], u.e[1] - v.e[1], u.e[2] - v.e[2]); }
inline vec3 operator*(const vec3& u, const vec3& v) { return vec3(u.e[0] * v.e[0], u.e[1] * v.e[1], u.e[2] * v.e[2]); }
inline vec3 operator*(const vec3& u, float s) { return vec3(u.e[0] * s, u.e[1] * s, u.e[2] * s); }
inline vec3 operator*(float s, const vec3& u) { return vec3(u.e[0] * s, u.e[1] * s, u.e[2] * s); }
inline vec3 operator/(const vec3& u, float s) { return vec3(u.e[0] / s, u.e[1] / s, u.e[2] / s); }
inline float dot(const vec3& u, const vec3& v) { return u.e[0] * v.e[0] + u.e[1] * v.e[1] + u.e[2] * v.e[2]; }  // :118
inline vec3 cross(const vec3& u, const vec3& v) {                                                             // :129
    return vec3(u.e[1] * v.e[2] - u.e[2] * v.e[1], u.e[2] * v.e[0] - u.e[0] * v.e[2], u.e[0] * v.e[1] - u.e[1] * v.e[0]);
}
inline vec3 unit_vector(const vec3& v) { return v / v.length(); }                                             // :141
inline float minf(float a, float b) { return a < b ? a : b; }                                                 // utility.cuh:71
inline vec3 reflect(const vec3& v, const vec3& n) { return v - 2.0f * dot(v, n) * n; }                        // :152
inline vec3 refract(const vec3& uv, const vec3& n, float ni_over_nt) {                                        // :166
    float cos_theta = minf(dot(-uv, n), 1.0f);
    vec3 r_out_perp = ni_over_nt * (uv + cos_theta * n);
    vec3 r_out_parallel = -std::sqrt(std::fabs(1.0f - r_out_perp.length_squared())) * n;
    return r_out_perp + r_out_parallel;
}

// ---------------------------------------------------------------- RNG (replaces curand) --
inline uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
struct rng_state {
    uint32_t state, draws;
};
// curand_init(1984 + pixel_index, samples_done, 0) at render.cuh:117, made per-sample.
inline void rng_init(rng_state& r, int pixel_index, int sample_number) {
    uint32_t s = lowbias32(1984u + (uint32_t)pixel_index);
    r.state = lowbias32(s + (uint32_t)sample_number * 0x9E3779B9u);
    r.draws = 0;
}
// curand_uniform: (0,1]
inline float uniform(rng_state& r) {
    r.draws++;
    uint32_t x = lowbias32(r.state + r.draws * 0x9E3779B9u);
    return (float)((x >> 8) + 1u) * (1.0f / 16777216.0f);
}
inline vec3 random_range(rng_state& r, float lo, float hi) {        // vec3.cuh:191-196
    float x = lo + (hi - lo) * uniform(r);
    float y = lo + (hi - lo) * uniform(r);
    float z = lo + (hi - lo) * uniform(r);
    return vec3(x, y, z);
}
inline vec3 random_unit_vector(rng_state& r) {                      // vec3.cuh:204-212
    while (true) {
        vec3 p = random_range(r, -1.0f, 1.0f);
        float lensq = p.length_squared();
        if (1e-30f < lensq && lensq <= 1) return unit_vector(p);
    }
}
inline vec3 random_in_unit_sphere(rng_state& r) {                   // vec3.cuh:235-241
    vec3 p = random_range(r, -1.0f, 1.0f);
    while (p.length_squared() >= 1.0f) p = random_range(r, -1.0f, 1.0f);
    return p;
}
inline vec3 random_in_unit_disk(rng_state& r) {                     // vec3.cuh:249-255
    float a = uniform(r), b = uniform(r);
    vec3 p = 2.0f * vec3(a, b, 0) - vec3(1, 1, 0);
    while (dot(p, p) >= 1.0f) {
        a = uniform(r); b = uniform(r);
        p = 2.0f * vec3(a, b, 0) - vec3(1, 1, 0);
    }
    return p;
}

// ---------------------------------------------------------------- interval.cuh, ray.cuh --
struct interval {
    float min, max;
    interval() : min(k_inf), max(-k_inf) {}
    interval(float a, float b) : min(a), max(b) {}
    interval(const interval& a, const interval& b) {               // interval.cuh:51-54
        min = a.min <= b.min ? a.min : b.min;
        max = a.max >= b.max ? a.max : b.max;
    }
    float size() const { return max - min; }
    bool contains(float x) const { return min <= x && x <= max; }  // :63
    bool surrounds(float x) const { return min < x && x < max; }   // :68
    interval expand(float x) const { return interval(min - x, max + x); }  // :78
};
struct ray {
    vec3 orig, dir;
    float tm;
    ray() : tm(0) {}
    ray(const vec3& o, const vec3& d, float t) : orig(o), dir(d), tm(t) {}
    vec3 at(float t) const { return orig + dir * t; }              // ray.cuh:55
};

// ---------------------------------------------------------------- aabb.cuh ---------------
struct aabb {
    interval x, y, z;
    aabb() {}
    aabb(const interval& a, const interval& b, const interval& c) : x(a), y(b), z(c) { pad_to_minimums(); }
    aabb(const aabb& a, const aabb& b) {                            // :87-92
        x = interval(a.x, b.x); y = interval(a.y, b.y); z = interval(a.z, b.z);
        pad_to_minimums();
    }
    aabb(const float* mn, const float* mx) {                        // :101-107
        x = interval(mn[0], mx[0]); y = interval(mn[1], mx[1]); z = interval(mn[2], mx[2]);
        pad_to_minimums();
    }
    const interval& operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    bool hit(const ray& r, interval ray_t) const {                  // :132-156
        float t_min = ray_t.min, t_max = ray_t.max;
        for (int i = 0; i < 3; i++) {
            float inv_d = (std::fabs(r.dir[i]) > 1e-12f) ? (1.0f / r.dir[i]) : 1e8f;
            float t0 = ((*this)[i].min - r.orig[i]) * inv_d;
            float t1 = ((*this)[i].max - r.orig[i]) * inv_d;
            if (inv_d < 0.0f) { float tmp = t0; t0 = t1; t1 = tmp; }
            t_min = std::fmax(t_min, t0);
            t_max = std::fmin(t_max, t1);
            if (t_max < t_min) return false;
        }
        return true;
    }
    int longest_axis() const {                                      // :163-166
        if (x.size() > y.size()) return x.size() > z.size() ? 0 : 2;
        return y.size() > z.size() ? 1 : 2;
    }
    void pad_to_minimums(float delta = 0.001f) {                    // :219-223
        if (x.size() < delta) x = x.expand(delta);
        if (y.size() < delta) y = y.expand(delta);
        if (z.size() < delta) z = z.expand(delta);
    }
    static aabb empty() { return aabb(interval(k_inf, -k_inf), interval(k_inf, -k_inf), interval(k_inf, -k_inf)); }
};

// ---------------------------------------------------------------- mat3.cuh:82-86 ---------
inline vec3 matvec(const float* m, const vec3& u) {
    return vec3(m[0] * u.x() + m[1] * u.y() + m[2] * u.z(),
                m[3] * u.x() + m[4] * u.y() + m[5] * u.z(),
                m[6] * u.x() + m[7] * u.y() + m[8] * u.z());
}

// ---------------------------------------------------------------- material.cuh:29-50 -----
struct hit_record {
    vec3 p, normal;
    float t, u, v;
    bool front_face;
    int mat_idx;
    vec3 tangent, bitangent;
    int obj_idx;   // bookkeeping for the parity tables; not read by the algorithm
    void set_face_normal(const ray& r, const vec3& outward_normal) {
        front_face = dot(r.dir, outward_normal) < 0;
        normal = front_face ? outward_normal : -outward_normal;
    }
};

struct counters {
    uint64_t rays, node_pops, box_tests, prim_tests, accepts;
    int max_stack;
};

// ---------------------------------------------------------------- texture.cuh ------------
// Software tex2D: cudaAddressModeWrap, cudaFilterModeLinear, normalised coordinates,
// cudaReadModeNormalizedFloat (texture.cuh:212-217).
inline int wrapi(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }
void tex2d(const crt_texture& img, float u, float v, float out[4]) {
    float x = (u - std::floor(u)) * (float)img.width - 0.5f;
    float y = (v - std::floor(v)) * (float)img.height - 0.5f;
    float xf = std::floor(x), yf = std::floor(y);
    float ax = x - xf, ay = y - yf;
    int x0 = wrapi((int)xf, img.width), x1 = wrapi((int)xf + 1, img.width);
    int y0 = wrapi((int)yf, img.height), y1 = wrapi((int)yf + 1, img.height);
    int ch = img.channels;
    for (int c = 0; c < 4; c++) out[c] = 0.0f;
    for (int c = 0; c < ch; c++) {
        float t00 = img.pixels[((size_t)y0 * img.width + x0) * ch + c] * (1.0f / 255.0f);
        float t10 = img.pixels[((size_t)y0 * img.width + x1) * ch + c] * (1.0f / 255.0f);
        float t01 = img.pixels[((size_t)y1 * img.width + x0) * ch + c] * (1.0f / 255.0f);
        float t11 = img.pixels[((size_t)y1 * img.width + x1) * ch + c] * (1.0f / 255.0f);
        out[c] = (1.0f - ax) * (1.0f - ay) * t00 + ax * (1.0f - ay) * t10 + (1.0f - ax) * ay * t01 + ax * ay * t11;
    }
}

vec3 texture_value(const crt_scene_desc* sc, int idx, float u, float v, const vec3& p) {   // :313-324
    const crt_texture& t = sc->textures[idx];
    switch (t.type) {
    case CRT_TEX_SOLID_COLOR:                                       // :252-255
        return vec3(t.color);
    case CRT_TEX_CHECKER: {                                         // :267-274
        int x_int = int(std::floor(t.inv_scale * p.x()));
        int y_int = int(std::floor(t.inv_scale * p.y()));
        int z_int = int(std::floor(t.inv_scale * p.z()));
        bool is_even = (x_int + y_int + z_int) % 2 == 0;
        return is_even ? texture_value(sc, t.even_idx, u, v, p) : texture_value(sc, t.odd_idx, u, v, p);
    }
    case CRT_TEX_IMAGE: {                                           // :286-302
        u = std::fmin(std::fmax(u, 0.0f), 1.0f);
        v = std::fmin(std::fmax(v, 0.0f), 1.0f);
        float px[4];
        if (t.channels == 1) { tex2d(t, u, v, px); return vec3(px[0], px[0], px[0]); }
        if (t.channels == 4) { tex2d(t, u, v, px); return vec3(px[0], px[1], px[2]); }
        return vec3(1, 0, 1);
    }
    default:
        return vec3(0, 0, 0);
    }
}

// ---------------------------------------------------------------- hittable.cuh -----------
bool hit_sphere(const ray& r, interval ray_t, hit_record& rec, const crt_object& o) {       // :272-301
    vec3 c0(o.p0), cd(o.p1);
    float radius = o.p2[0];
    vec3 current_center = c0 + cd * r.tm;                           // center.at(r.time())
    vec3 oc = r.orig - current_center;
    float a = r.dir.length_squared();
    float h = dot(oc, r.dir);
    float c = oc.length_squared() - radius * radius;
    float discriminant = h * h - a * c;
    if (discriminant < 0) return false;
    float sqrt_discriminant = std::sqrt(discriminant);
    float root = (-h - sqrt_discriminant) / a;
    if (!ray_t.surrounds(root)) {
        root = (-h + sqrt_discriminant) / a;
        if (!ray_t.surrounds(root)) return false;
    }
    rec.t = root;
    rec.p = r.at(rec.t);
    vec3 outward_normal = (rec.p - current_center) / radius;
    rec.set_face_normal(r, outward_normal);
    {                                                               // get_sphere_uv :179-184
        float theta = o_acos(-outward_normal.y());
        float phi = o_atan2(-outward_normal.z(), outward_normal.x()) + k_pi;
        rec.u = phi / (2 * k_pi);
        rec.v = theta / k_pi;
    }
    rec.mat_idx = o.mat_idx;
    {                                                               // get_tangent_bitangent :191-197
        vec3 local_p = rec.p - (c0 + cd * rec.t);                   // center.at(rec.t) -- t, not time
        float phi = o_atan2(-local_p.z(), local_p.x()) + k_pi;
        float sin_phi, cos_phi;
        o_sincos(phi, sin_phi, cos_phi);
        rec.tangent = unit_vector(vec3(-radius * sin_phi, 0, radius * cos_phi));
        rec.bitangent = unit_vector(cross(rec.normal, rec.tangent));
    }
    return true;
}

bool hit_tri(const ray& r, interval ray_t, hit_record& rec, const crt_object& o, float epsilon = 1e-8f) {  // :314-359
    vec3 point(o.p0), edge1(o.p1), edge2(o.p2);
    vec3 pvec = cross(r.dir, edge2);
    float det = dot(edge1, pvec);
    if (std::fabs(det) < epsilon) return false;
    float inv_det = 1.0f / det;
    vec3 tvec = r.orig - point;
    float u = dot(tvec, pvec) * inv_det;
    if (u < 0.0f || u > 1.0f) return false;
    vec3 qvec = cross(tvec, edge1);
    float v = dot(r.dir, qvec) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = dot(edge2, qvec) * inv_det;
    if (!ray_t.contains(t)) return false;
    rec.t = t;
    rec.p = r.at(t);
    rec.mat_idx = o.mat_idx;
    rec.tangent = vec3(o.tangent);
    rec.bitangent = vec3(o.bitangent);
    float w = 1.0f - u - v;
    if (o.shade_smooth) {
        rec.normal = unit_vector(w * vec3(o.n0) + u * vec3(o.n1) + v * vec3(o.n2));
    } else {
        rec.normal = vec3(o.n0);
    }
    rec.set_face_normal(r, rec.normal);
    // vec2 uv = w*uv0 + u*uv1 + v*uv2
    rec.u = w * o.uv0[0] + u * o.uv1[0] + v * o.uv2[0];
    rec.v = w * o.uv0[1] + u * o.uv1[1] + v * o.uv2[1];
    return true;
}

void apply_normal_map(const crt_scene_desc* sc, hit_record& rec, int tex_idx) {             // :367-386
    const crt_texture& nm = sc->textures[tex_idx];
    vec3 pixel(0, 0, 1);
    float px[4];
    if (nm.channels == 1) {
        tex2d(nm, rec.u, rec.v, px);
        float value = px[0] * 2.0f - 1.0f;
        pixel = vec3(value, value, value);
    } else if (nm.channels == 4) {
        tex2d(nm, rec.u, rec.v, px);
        pixel = vec3(px[0] * 2.0f - 1.0f, px[1] * 2.0f - 1.0f, px[2] * 2.0f - 1.0f);
    }
    vec3 world_normal(
        rec.tangent[0] * pixel[0] + rec.bitangent[0] * pixel[1] + rec.normal[0] * pixel[2],
        rec.tangent[1] * pixel[0] + rec.bitangent[1] * pixel[1] + rec.normal[1] * pixel[2],
        rec.tangent[2] * pixel[0] + rec.bitangent[2] * pixel[1] + rec.normal[2] * pixel[2]);
    rec.normal = unit_vector(world_normal);
}

bool object_hit_transformed(const crt_scene_desc* sc, const crt_object& o, const ray& r, interval ray_t,
                            hit_record& rec, counters& cnt) {                               // hittable::hit :426-446
    vec3 anchor(o.anchor), translation(o.translation);
    vec3 transformed_origin = matvec(o.rotation_inv, r.orig - anchor - translation) + anchor;
    vec3 transformed_direction = matvec(o.rotation_inv, r.dir);
    ray transformed_r(transformed_origin, transformed_direction, r.tm);
    cnt.prim_tests++;
    bool hit;
    switch (o.type) {                                               // object_hit :398-413
    case CRT_OBJ_SPHERE: hit = hit_sphere(transformed_r, ray_t, rec, o); break;
    case CRT_OBJ_TRI: hit = hit_tri(transformed_r, ray_t, rec, o); break;
    default: hit = false;
    }
    if (!hit) return hit;
    cnt.accepts++;
    rec.p = matvec(o.rotation, rec.p - anchor) + anchor;
    rec.p += translation;
    rec.normal = matvec(o.rotation, rec.normal);
    if (o.normal_idx >= 0) apply_normal_map(sc, rec, o.normal_idx);
    return hit;
}

// ---------------------------------------------------------------- bvh.cuh:380-421 --------
bool bvh4_hit(const crt_scene_desc* sc, const ray& r, interval ray_t, hit_record& rec, counters& cnt) {
    const crt_bvh4& tree = sc->bvh;
    int stack[256];
    int stack_ptr = 0;
    stack[stack_ptr++] = 0;
    bool hit_anything = false;
    float closest_t = ray_t.max;
    cnt.rays++;
    while (stack_ptr > 0) {
        if (stack_ptr > cnt.max_stack) cnt.max_stack = stack_ptr;
        int node_index = stack[--stack_ptr];
        if (node_index == -1) continue;
        cnt.node_pops++;
        for (int i = 0; i < 4; i++) {
            int child_index = node_index * 4 + i;
            aabb child_box(tree.min_bounds + 3 * child_index, tree.max_bounds + 3 * child_index);
            cnt.box_tests++;
            if (!child_box.hit(r, ray_t)) continue;
            int object_index = tree.object_index[child_index];
            int children_index = tree.children[child_index];
            if (object_index >= 0) {
                if (object_hit_transformed(sc, sc->objects[object_index], r, ray_t, rec, cnt)) {
                    hit_anything = true;
                    closest_t = rec.t;
                    ray_t.max = closest_t;
                    rec.obj_idx = object_index;
                }
            }
            if (children_index >= 0) stack[stack_ptr++] = children_index;
        }
    }
    return hit_anything;
}

// ---------------------------------------------------------------- material.cuh -----------
inline float pow5(float x) {
    if (g_pow5_mulchain) { float x2 = x * x; return x2 * x2 * x; }
    return powf(x, 5.0f);                                            // pow(1-cos, 5.0f) :152,:186
}
bool scatter(const crt_scene_desc* sc, const ray& r_in, const hit_record& rec, vec3& attenuation, ray& scattered,
             rng_state& rs) {                                                               // :359-372
    const crt_material& m = sc->materials[rec.mat_idx];
    switch (m.type) {
    case CRT_MAT_LAMBERTIAN: {                                      // :290-297
        vec3 scatter_direction = rec.normal + random_unit_vector(rs);
        if (scatter_direction.near_zero()) scatter_direction = rec.normal;
        scattered = ray(rec.p + rec.normal * 0.001f, scatter_direction, r_in.tm);
        attenuation = texture_value(sc, m.tex_idx, rec.u, rec.v, rec.p);
        return true;
    }
    case CRT_MAT_METAL: {                                           // :311-322
        vec3 reflected = reflect(unit_vector(r_in.dir), rec.normal);
        scattered = ray(rec.p, reflected + m.fuzz * random_in_unit_sphere(rs), r_in.tm);
        vec3 unit_direction = unit_vector(r_in.dir);
        float cos_theta = std::min(dot(-unit_direction, rec.normal), 1.0f);
        vec3 albedo(m.albedo);
        attenuation = albedo + (vec3(1, 1, 1) - albedo) * pow5(1.0f - cos_theta);          // :151-153
        return dot(scattered.dir, rec.normal) > 0.0f;
    }
    case CRT_MAT_DIELECTRIC: {                                      // :336-356
        attenuation = vec3(1.0, 1.0, 1.0);
        float refraction_ratio = rec.front_face ? (1.0f / m.ior) : m.ior;
        vec3 unit_direction = unit_vector(r_in.dir);
        float cos_theta = minf(dot(-unit_direction, rec.normal), 1.0f);
        float sin_theta = std::sqrt(1.0f - cos_theta * cos_theta);
        bool cannot_refract = refraction_ratio * sin_theta > 1.0f;
        vec3 direction;
        bool do_reflect = cannot_refract;
        if (!do_reflect) {                                           // || short-circuit: the uniform is
            float r0 = (1.0f - refraction_ratio) / (1.0f + refraction_ratio);  // only drawn when it can refract
            r0 = r0 * r0;
            float reflectance = r0 + (1.0f - r0) * pow5(1.0f - cos_theta);                 // :183-187
            do_reflect = reflectance > uniform(rs);
        }
        if (do_reflect) direction = reflect(unit_direction, rec.normal);
        else direction = refract(unit_direction, rec.normal, refraction_ratio);
        scattered = ray(rec.p, direction, r_in.tm);
        return true;
    }
    default:
        return false;
    }
}
vec3 emit(const crt_scene_desc* sc, const hit_record& rec) {                                // :375-381
    const crt_material& m = sc->materials[rec.mat_idx];
    if (m.type == CRT_MAT_DIFFUSE_LIGHT) return texture_value(sc, m.tex_idx, rec.u, rec.v, rec.p);
    return vec3(0, 0, 0);
}

// ---------------------------------------------------------------- camera.cuh:240-278 -----
ray get_ray(const crt_camera& cam, int i, int j, rng_state& rs) {
    vec3 offset = random_range(rs, -0.5f, 0.5f);                    // sample_square :93-96 (3 draws)
    vec3 llc(cam.lower_left_corner), du(cam.pixel_delta_u), dv(cam.pixel_delta_v);
    if (cam.type == CRT_CAM_PERSPECTIVE) {
        vec3 pixel_sample = llc + (i + offset.x()) * du + (j + offset.y()) * dv;
        vec3 ray_origin;
        if (cam.defocus_angle <= 0) {
            ray_origin = vec3(cam.center);
        } else {                                                    // defocus_disk_sample :170-173
            vec3 v = random_in_unit_disk(rs);
            ray_origin = vec3(cam.center) + (v[0] * vec3(cam.defocus_disk_u)) + (v[1] * vec3(cam.defocus_disk_v));
        }
        vec3 ray_direction = pixel_sample - ray_origin;
        float ray_time = uniform(rs);
        return ray(ray_origin, ray_direction, ray_time);
    }
    vec3 ray_origin = llc + (i + offset.x()) * du + (j + offset.y()) * dv;
    vec3 ray_direction(cam.w);
    float ray_time = uniform(rs);
    return ray(ray_origin, ray_direction, ray_time);
}

// ---------------------------------------------------------------- render.cuh:19-98 -------
struct ray_info {
    vec3 col, albedo, emission, normal;
    float depth, opacity;
};
static thread_local bool g_trace_path = false;
void get_ray_info(const crt_scene_desc* sc, const ray& r, rng_state& rs, ray_info& result, counters& cnt) {
    ray cur_ray = r;
    vec3 cur_attenuation(1, 1, 1);
    hit_record rec;
    int bounces = 0;
    int max_bounces = sc->render.max_bounces;
    while (bounces < max_bounces) {
        const bool any_hit = bvh4_hit(sc, cur_ray, interval(0.001f, k_inf), rec, cnt);
        if (g_trace_path) {             // development aid: ORACLE_TRACE_PIXEL=<pixel index> prints the path
            std::printf("  bounce %d  o %.9g %.9g %.9g  d %.9g %.9g %.9g  time %.9g  hit %d t %.9g object %d\n", bounces,
                        cur_ray.orig.x(), cur_ray.orig.y(), cur_ray.orig.z(), cur_ray.dir.x(), cur_ray.dir.y(), cur_ray.dir.z(),
                        cur_ray.tm, (int)any_hit, any_hit ? rec.t : 0.0f, any_hit ? rec.obj_idx : -1);
        }
        if (any_hit) {
            ray scattered;
            vec3 attenuation;
            if (scatter(sc, cur_ray, rec, attenuation, scattered, rs)) {
                if (bounces == 0) {
                    result.albedo = attenuation;
                    result.normal = unit_vector(rec.normal);
                    result.depth = rec.t * cur_ray.dir.length();
                    result.opacity = 1.0f;
                }
                cur_attenuation *= attenuation;
                cur_ray = scattered;
                bounces++;
            } else {
                vec3 emitted = emit(sc, rec);
                result.col = cur_attenuation * emitted;
                if (bounces == 0) {
                    result.albedo = emitted;
                    result.emission = emitted;
                    result.normal = unit_vector(rec.normal);
                    result.depth = rec.t * cur_ray.dir.length();
                    result.opacity = 1.0f;
                }
                return;
            }
        } else {
            vec3 p = unit_vector(cur_ray.dir);
            float theta = o_acos(-p.y());
            float phi = o_atan2(-p.z(), p.x()) + k_pi;
            float u = phi / (2 * k_pi);
            float v = theta / k_pi;
            vec3 c = texture_value(sc, sc->background_index, u, v, p);
            result.col = cur_attenuation * c;
            if (bounces == 0) {
                result.albedo = result.col;
                result.emission = vec3(0, 0, 0);
                result.normal = vec3(0.0f, 0.0f, 0.0f);
                result.depth = nanf("");
                result.opacity = 0.0f;
            }
            return;
        }
    }
    result.col = vec3(0, 0, 0);
    result.albedo = vec3(0, 0, 0);
}

// render_kernel body for one pixel, render.cuh:108-154
void render_pixel(const crt_scene_desc* sc, int i, int j, int samples, int samples_done, float* color, float* albedo,
                  float* emission, float* normal, float* depth, float* opacity, counters& cnt) {
    int W = sc->render.image_width;
    int pixel_index = j * W + i;
    vec3* C = reinterpret_cast<vec3*>(color) + pixel_index;
    vec3* A = reinterpret_cast<vec3*>(albedo) + pixel_index;
    vec3* E = reinterpret_cast<vec3*>(emission) + pixel_index;
    vec3* N = reinterpret_cast<vec3*>(normal) + pixel_index;
    for (int s = 0; s < samples; s++) {
        int current_sample = samples_done + s + 1;
        rng_state rs;
        rng_init(rs, pixel_index, current_sample);
        ray_info result;
        result.depth = 0.0f; result.opacity = 0.0f;   // uninitialised in the reference; always written when max_bounces >= 1
        ray r = get_ray(sc->camera, i, j, rs);
        static const char* trace_env = std::getenv("ORACLE_TRACE_PIXEL");
        g_trace_path = trace_env && std::atoi(trace_env) == pixel_index;
        if (g_trace_path) std::printf("pixel %d sample %d\n", pixel_index, current_sample);
        get_ray_info(sc, r, rs, result, cnt);
        g_trace_path = false;
        if (current_sample == 1) {
            *C = result.col; *A = result.albedo; *E = result.emission; *N = result.normal;
            depth[pixel_index] = result.depth; opacity[pixel_index] = result.opacity;
            continue;
        }
        float n1 = (float)(current_sample - 1), n = (float)current_sample;
        *C = (*C * n1 + result.col) / n;
        *A = (*A * n1 + result.albedo) / n;
        *E = (*E * n1 + result.emission) / n;
        *N = (*N * n1 + result.normal) / n;
        depth[pixel_index] = (depth[pixel_index] * n1 + result.depth) / n;
        opacity[pixel_index] = (opacity[pixel_index] * n1 + result.opacity) / n;
    }
}

// ---------------------------------------------------------------- bvh.cuh:42-310 (host) --
struct build_obj {
    aabb bbox;
    bool visible;
};
struct build_node {
    build_node* child_node[4];
    int child_obj[4];       // leaf children: index into the ORIGINAL object vector
    bool present[4];
    bool is_leaf[4];
    aabb box;
};
aabb compute_bounding_box(const std::vector<build_obj>& all, const std::vector<int>& objs, size_t start, size_t end) {  // :192-198
    aabb bbox = aabb::empty();
    for (size_t i = start; i < end; i++) bbox = aabb(bbox, all[objs[i]].bbox);
    return bbox;
}
// bvh4_node_recursive(std::vector<hittable*> objects, start, end)  :68-133.
// `objects` holds indices into `all` instead of pointers; the comparator reads the same key.
build_node* build_recursive(const std::vector<build_obj>& all, std::vector<int> objects, size_t start, size_t end) {
    build_node* node = new build_node();
    std::vector<int> visible_objects;
    for (size_t i = start; i < end; i++)
        if (all[objects[i]].visible) visible_objects.push_back(objects[i]);
    size_t object_span = visible_objects.size();
    if (object_span == 0) {
        for (int i = 0; i < 4; i++) { node->present[i] = false; node->is_leaf[i] = false; node->child_node[i] = nullptr; }
        node->box = aabb::empty();
        return node;
    }
    if (object_span <= 4) {
        for (size_t i = 0; i < object_span; i++) { node->present[i] = true; node->is_leaf[i] = true; node->child_obj[i] = visible_objects[i]; node->child_node[i] = nullptr; }
        for (size_t i = object_span; i < 4; i++) { node->present[i] = false; node->is_leaf[i] = false; node->child_node[i] = nullptr; }
        node->box = compute_bounding_box(all, visible_objects, 0, object_span);
        return node;
    }
    aabb bbox = compute_bounding_box(all, visible_objects, 0, object_span);
    int axis = bbox.longest_axis();
    std::sort(visible_objects.begin(), visible_objects.end(),
              [&](int a, int b) { return all[a].bbox[axis].min < all[b].bbox[axis].min; });   // :145-149
    size_t mid1 = 1 * object_span / 4, mid2 = 2 * object_span / 4, mid3 = 3 * object_span / 4;
    node->child_node[0] = build_recursive(all, visible_objects, 0, mid1);
    node->child_node[1] = build_recursive(all, visible_objects, mid1, mid2);
    node->child_node[2] = build_recursive(all, visible_objects, mid2, mid3);
    node->child_node[3] = build_recursive(all, visible_objects, mid3, object_span);
    for (int i = 0; i < 4; i++) { node->present[i] = true; node->is_leaf[i] = false; }
    node->box = aabb::empty();
    for (int i = 0; i < 4; i++) node->box = aabb(node->box, node->child_node[i]->box);
    return node;
}
int count_nodes(const build_node* n) {                              // :251-262
    if (!n) return 0;
    int c = 1;
    for (int i = 0; i < 4; i++)
        if (!n->is_leaf[i] && n->present[i]) c += count_nodes(n->child_node[i]);
    return c;
}
int flatten(const build_node* node, const std::vector<build_obj>& all, int& current_index, float* mn, float* mx,
            int32_t* children, int32_t* object_index) {             // :210-243
    int node_index = current_index++;
    for (int i = 0; i < 4; i++) {
        int ci = node_index * 4 + i;
        if (node->present[i]) {
            if (node->is_leaf[i]) {
                const aabb& b = all[node->child_obj[i]].bbox;
                mn[3 * ci] = b.x.min; mn[3 * ci + 1] = b.y.min; mn[3 * ci + 2] = b.z.min;
                mx[3 * ci] = b.x.max; mx[3 * ci + 1] = b.y.max; mx[3 * ci + 2] = b.z.max;
                object_index[ci] = node->child_obj[i];
                children[ci] = -1;
            } else {
                const build_node* ch = node->child_node[i];
                mn[3 * ci] = ch->box.x.min; mn[3 * ci + 1] = ch->box.y.min; mn[3 * ci + 2] = ch->box.z.min;
                mx[3 * ci] = ch->box.x.max; mx[3 * ci + 1] = ch->box.y.max; mx[3 * ci + 2] = ch->box.z.max;
                object_index[ci] = -1;
                children[ci] = flatten(ch, all, current_index, mn, mx, children, object_index);
            }
        } else {
            mn[3 * ci] = mn[3 * ci + 1] = mn[3 * ci + 2] = k_inf;
            mx[3 * ci] = mx[3 * ci + 1] = mx[3 * ci + 2] = -k_inf;
            object_index[ci] = -1;
            children[ci] = -1;
        }
    }
    return node_index;
}
void free_nodes(build_node* n) {
    if (!n) return;
    for (int i = 0; i < 4; i++) free_nodes(n->child_node[i]);
    delete n;
}

}  // namespace

extern "C" {

void oracle_set_pow5_mulchain(int on) { g_pow5_mulchain = on != 0; }
// 1: acos/atan2/sin/cos through the plain-float evaluations the kernels use (bit-exact comparisons of
// textured spheres and environment images); 0 (default): the C library's, like the reference's calls.
void oracle_set_det_math(int on) { g_det_math = on != 0; }
// the four functions themselves, for accuracy tests against the C library in double precision
float oracle_det_acos(float x) { return crt_det::acos(x); }
float oracle_det_atan2(float y, float x) { return crt_det::atan2(y, x); }
float oracle_det_sin(float x) { float s, c; crt_det::sincos(x, s, c); return s; }
float oracle_det_cos(float x) { float s, c; crt_det::sincos(x, s, c); return c; }

// render() over the whole image; buffers are host arrays in the reference layout and are
// read-modified-written exactly like render_kernel does.  stats: rays, node_pops, box_tests,
// prim_tests, accepts, max_stack (6 x uint64, may be NULL).
int oracle_render(const crt_scene_desc* sc, int samples, int samples_done, float* color, float* albedo,
                  float* emission, float* normal, float* depth, float* opacity, uint64_t* stats, int threads) {
    int W = sc->render.image_width, H = sc->render.image_height;
    if (samples < 0) samples = sc->render.samples_per_pixel;                       // render.cuh:120-124
    else samples = (int)minf((float)samples, (float)sc->render.samples_per_pixel);
    counters total = {0, 0, 0, 0, 0, 0};
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        counters cnt = {0, 0, 0, 0, 0, 0};
#pragma omp for schedule(dynamic, 4)
        for (int j = 0; j < H; j++)
            for (int i = 0; i < W; i++)
                render_pixel(sc, i, j, samples, samples_done, color, albedo, emission, normal, depth, opacity, cnt);
#pragma omp critical
        {
            total.rays += cnt.rays; total.node_pops += cnt.node_pops; total.box_tests += cnt.box_tests;
            total.prim_tests += cnt.prim_tests; total.accepts += cnt.accepts;
            total.max_stack = std::max(total.max_stack, cnt.max_stack);
        }
    }
    if (stats) {
        stats[0] = total.rays; stats[1] = total.node_pops; stats[2] = total.box_tests;
        stats[3] = total.prim_tests; stats[4] = total.accepts; stats[5] = (uint64_t)total.max_stack;
    }
    return 0;
}

// Bounce-0 closest hit of sample number `sample` for every pixel.
int oracle_trace_primary(const crt_scene_desc* sc, int sample, int32_t* object_index, float* t, float* u, float* v,
                         int32_t* front_face) {
    int W = sc->render.image_width, H = sc->render.image_height;
#pragma omp parallel for schedule(dynamic, 4)
    for (int j = 0; j < H; j++) {
        for (int i = 0; i < W; i++) {
            int pixel_index = j * W + i;
            rng_state rs;
            rng_init(rs, pixel_index, sample);
            ray r = get_ray(sc->camera, i, j, rs);
            hit_record rec;
            counters cnt = {0, 0, 0, 0, 0, 0};
            bool hit = bvh4_hit(sc, r, interval(0.001f, k_inf), rec, cnt);
            if (object_index) object_index[pixel_index] = hit ? rec.obj_idx : -1;
            if (t) t[pixel_index] = hit ? rec.t : k_inf;
            if (u) u[pixel_index] = hit ? rec.u : 0.0f;
            if (v) v[pixel_index] = hit ? rec.v : 0.0f;
            if (front_face) front_face[pixel_index] = hit ? (rec.front_face ? 1 : 0) : 0;
        }
    }
    return 0;
}

// First `count` uniforms of the stream of (pixel, sample) -- pins the RNG specification.
void oracle_rng_stream(int pixel_index, int sample_number, int count, float* out) {
    rng_state rs;
    rng_init(rs, pixel_index, sample_number);
    for (int k = 0; k < count; k++) out[k] = uniform(rs);
}

// Camera ray of (i, j, sample): origin[3], direction[3], time.
void oracle_camera_ray(const crt_scene_desc* sc, int i, int j, int sample, float* out7) {
    rng_state rs;
    rng_init(rs, j * sc->render.image_width + i, sample);
    ray r = get_ray(sc->camera, i, j, rs);
    for (int k = 0; k < 3; k++) { out7[k] = r.orig[k]; out7[3 + k] = r.dir[k]; }
    out7[6] = r.tm;
}

// create_bvh4 (bvh.cuh:299-310).  Returns the slot count; arrays sized by the caller
// (4 * (number of nodes) slots; call with NULL arrays to get the count).
int oracle_bvh4_build(int n, const float* bbox_min, const float* bbox_max, const int32_t* visible, int capacity_slots,
                      float* min_bounds, float* max_bounds, int32_t* children, int32_t* object_index) {
    std::vector<build_obj> all(n);
    std::vector<int> objects(n);
    for (int i = 0; i < n; i++) {
        all[i].bbox.x = interval(bbox_min[3 * i], bbox_max[3 * i]);
        all[i].bbox.y = interval(bbox_min[3 * i + 1], bbox_max[3 * i + 1]);
        all[i].bbox.z = interval(bbox_min[3 * i + 2], bbox_max[3 * i + 2]);
        all[i].visible = visible ? visible[i] != 0 : true;
        objects[i] = i;
    }
    build_node* root = build_recursive(all, objects, 0, (size_t)n);
    int slots = count_nodes(root) * 4;
    if (min_bounds && max_bounds && children && object_index) {
        if (capacity_slots < slots) { free_nodes(root); return -1; }
        int current_index = 0;
        flatten(root, all, current_index, min_bounds, max_bounds, children, object_index);
    }
    free_nodes(root);
    return slots;
}

// Single-function probes for known-answer tests -------------------------------------------
int oracle_aabb_hit(const float* mn, const float* mx, const float* orig, const float* dir, float tmin, float tmax) {
    aabb b(mn, mx);
    return b.hit(ray(vec3(orig), vec3(dir), 0.0f), interval(tmin, tmax)) ? 1 : 0;
}
// out: t, u(surface), v(surface), front_face, normal[3], p[3]  (10 floats)
int oracle_object_hit(const crt_scene_desc* sc, int obj, const float* orig, const float* dir, float time, float tmin,
                      float tmax, float* out10) {
    hit_record rec;
    counters cnt = {0, 0, 0, 0, 0, 0};
    bool h = object_hit_transformed(sc, sc->objects[obj], ray(vec3(orig), vec3(dir), time), interval(tmin, tmax), rec, cnt);
    if (h && out10) {
        out10[0] = rec.t; out10[1] = rec.u; out10[2] = rec.v; out10[3] = rec.front_face ? 1.0f : 0.0f;
        for (int k = 0; k < 3; k++) { out10[4 + k] = rec.normal[k]; out10[7 + k] = rec.p[k]; }
    }
    return h ? 1 : 0;
}
void oracle_texture_value(const crt_scene_desc* sc, int tex, float u, float v, const float* p, float* out3) {
    vec3 c = texture_value(sc, tex, u, v, vec3(p));
    out3[0] = c[0]; out3[1] = c[1]; out3[2] = c[2];
}

}  // extern "C"
