// Host scene layer: YAML scene file -> flat scene description + BVH4.
//
// Mirrors the behaviour of the reference's loader so that every index (texture, material,
// object) and every float of the geometry comes out the same:
//   load_scene order  camera/render -> textures -> materials -> objects -> environment
//                     (src/scene.cuh:650-690); inline textures/materials are appended
//                     where they are met, an object's `normal` texture before its material
//                     (scene.cuh:422-427).
//   object factories  quad / box / uv_sphere / model (src/objects.cuh:36-193)
//   transforms        translate / rotate on the `hittable` base (src/hittable.cuh:77-127)
//   cameras           src/camera.cuh:121-151 (perspective), :204-220 (orthographic)
// All arithmetic is single precision in the reference's operation order.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "crt_internal.h"
#include "vecmath.h"
#include "yaml_lite.h"

using yamllite::Node;

namespace {

const float k_pi = 3.1415926535897932385f;

struct SceneError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
[[noreturn]] void fail(const std::string& msg) { throw SceneError(msg); }
std::string at_line(const Node& n) { return " (line " + std::to_string(n.line) + ")"; }

// parse_required / parse_optional of utility.cuh:294-325
template <typename F>
auto required(const Node& node, const char* field, const std::string& message, F conv) -> decltype(conv(node)) {
    try {
        return conv(node[field]);
    } catch (const yamllite::ParseError&) {
        fail(message);
    }
}
template <typename T, typename F>
T optional(const Node& node, const char* field, T def, F conv) {
    const Node& f = node[field];
    if (!f.defined()) return def;
    try {
        return conv(f);
    } catch (const yamllite::ParseError&) {
        fail(std::string("Field '") + field + "' is invalid");
    }
}
float to_float(const Node& n) { return n.as_float(); }
int to_int(const Node& n) { return n.as_int(); }
bool to_bool(const Node& n) { return n.as_bool(); }
std::string to_string(const Node& n) { return n.as_string(); }

V3 vec3_of(const Node& n) {
    if (!n.is_seq() || n.size() < 3) throw yamllite::ParseError("not a 3-vector", n.line);
    return v3(n[0].as_float(), n[1].as_float(), n[2].as_float());
}
V3 required_vec3(const Node& node, const char* field, const std::string& message) {
    try {
        return vec3_of(node[field]);
    } catch (const yamllite::ParseError&) {
        fail(message);
    }
}
V3 optional_vec3(const Node& node, const char* field, V3 def) {
    const Node& f = node[field];
    if (!f.defined()) return def;
    try {
        return vec3_of(f);
    } catch (const yamllite::ParseError&) {
        fail(std::string("Field '") + field + "' is invalid");
    }
}
void optional_vec2(const Node& node, const char* field, float def0, float def1, float out[2]) {
    const Node& f = node[field];
    out[0] = def0; out[1] = def1;
    if (!f.defined()) return;
    try {
        if (!f.is_seq() || f.size() < 2) throw yamllite::ParseError("not a 2-vector", f.line);
        out[0] = f[0].as_float(); out[1] = f[1].as_float();
    } catch (const yamllite::ParseError&) {
        fail(std::string("Field '") + field + "' is invalid");
    }
}

void put(float* dst, V3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }
float deg2rad(float degrees) { return degrees * k_pi / 180.0f; }   // utility.cuh:98-104

void set_identity(float* m) {
    for (int i = 0; i < 9; i++) m[i] = 0.0f;
    m[0] = m[4] = m[8] = 1.0f;
}
// mat3::operator*(mat3) (mat3.cuh:94-104)
void matmul(const float* a, const float* b, float* out) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) out[3 * i + j] = a[3 * i + 0] * b[0 + j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}
// mat3::from_euler (mat3.cuh:112-125): Rz * Ry * Rx
void from_euler(V3 ang, float* out) {
    float cx = std::cos(ang.x), sx = std::sin(ang.x);
    float cy = std::cos(ang.y), sy = std::sin(ang.y);
    float cz = std::cos(ang.z), sz = std::sin(ang.z);
    float rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx};
    float ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy};
    float rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
    float zy[9];
    matmul(rz, ry, zy);
    matmul(zy, rx, out);
}
// mat3::inverse (mat3.cuh:143-161), cofactors times 1/det
void inverse3(const float* m, float* r) {
#define M(i, j) m[3 * (i) + (j)]
    float det = M(0, 0) * (M(1, 1) * M(2, 2) - M(1, 2) * M(2, 1)) - M(0, 1) * (M(1, 0) * M(2, 2) - M(1, 2) * M(2, 0)) +
                M(0, 2) * (M(1, 0) * M(2, 1) - M(1, 1) * M(2, 0));
    float inv_det = 1.0f / det;
    r[0] = (M(1, 1) * M(2, 2) - M(1, 2) * M(2, 1)) * inv_det;
    r[1] = (M(0, 2) * M(2, 1) - M(0, 1) * M(2, 2)) * inv_det;
    r[2] = (M(0, 1) * M(1, 2) - M(0, 2) * M(1, 1)) * inv_det;
    r[3] = (M(1, 2) * M(2, 0) - M(1, 0) * M(2, 2)) * inv_det;
    r[4] = (M(0, 0) * M(2, 2) - M(0, 2) * M(2, 0)) * inv_det;
    r[5] = (M(1, 0) * M(0, 2) - M(0, 0) * M(1, 2)) * inv_det;
    r[6] = (M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1)) * inv_det;
    r[7] = (M(2, 0) * M(0, 1) - M(0, 0) * M(2, 1)) * inv_det;
    r[8] = (M(0, 0) * M(1, 1) - M(1, 0) * M(0, 1)) * inv_det;
#undef M
}

crt::Box box_of(const crt_object& o) {
    crt::Box b;
    for (int a = 0; a < 3; a++) { b.mn[a] = o.bbox_min[a]; b.mx[a] = o.bbox_max[a]; }
    return b;
}
void set_box(crt_object& o, const crt::Box& b) {
    for (int a = 0; a < 3; a++) { o.bbox_min[a] = b.mn[a]; o.bbox_max[a] = b.mx[a]; }
}

crt_object blank_object(int type, int mat_idx, bool visible) {   // hittable ctor, hittable.cuh:32-47
    crt_object o;
    std::memset(&o, 0, sizeof(o));
    o.type = type;
    o.mat_idx = mat_idx;
    o.normal_idx = -1;
    o.visible = visible ? 1 : 0;
    set_identity(o.rotation);
    set_identity(o.rotation_inv);
    return o;
}

// tri::tri (hittable.cuh:227-246)
crt_object make_tri(V3 v0, V3 v1, V3 v2, V3 n0, V3 n1, V3 n2, const float uv0[2], const float uv1[2], const float uv2[2],
                    int mat_idx, bool shade_smooth = false, bool visible = true) {
    crt_object o = blank_object(CRT_OBJ_TRI, mat_idx, visible);
    V3 edge1 = v1 - v0, edge2 = v2 - v0;
    put(o.p0, v0); put(o.p1, edge1); put(o.p2, edge2);
    float a[3], b[3], c[3];
    put(a, v0); put(b, v1); put(c, v2);
    set_box(o, crt::box_from_points(a, b, c));
    o.shade_smooth = shade_smooth ? 1 : 0;
    if (!shade_smooth) {
        n0 = unit(cross(edge1, edge2));
        n1 = n0;
        n2 = n0;
    }
    put(o.n0, n0); put(o.n1, n1); put(o.n2, n2);
    for (int k = 0; k < 2; k++) { o.uv0[k] = uv0[k]; o.uv1[k] = uv1[k]; o.uv2[k] = uv2[k]; }
    float d1x = uv1[0] - uv0[0], d1y = uv1[1] - uv0[1];
    float d2x = uv2[0] - uv0[0], d2y = uv2[1] - uv0[1];
    float f = 1.0f / (d1x * d2y - d2x * d1y);
    put(o.tangent, f * (d2y * edge1 - d1y * edge2));
    put(o.bitangent, f * (-d2x * edge1 + d1x * edge2));
    return o;
}

// quad() (objects.cuh:100-106)
void make_quad(V3 corner, V3 u, V3 v, int mat_idx, bool visible, std::vector<crt_object>& out) {
    V3 normal = unit(cross(u, v));
    const float a[2] = {0, 0}, b[2] = {1, 0}, c[2] = {1, 1}, d[2] = {0, 1};
    out.push_back(make_tri(corner, corner + u, corner + u + v, normal, normal, normal, a, b, c, mat_idx, false, visible));
    out.push_back(make_tri(corner, corner + u + v, corner + v, normal, normal, normal, a, c, d, mat_idx, false, visible));
}

// box() (objects.cuh:117-146)
void make_box(V3 c1, V3 c2, int mat_idx, bool visible, std::vector<crt_object>& out) {
    V3 mn = v3(fminf(c1.x, c2.x), fminf(c1.y, c2.y), fminf(c1.z, c2.z));
    V3 mx = v3(fmaxf(c1.x, c2.x), fmaxf(c1.y, c2.y), fmaxf(c1.z, c2.z));
    V3 dx = v3(mx.x - mn.x, 0, 0), dy = v3(0, mx.y - mn.y, 0), dz = v3(0, 0, mx.z - mn.z);
    make_quad(mn, dx, dy, mat_idx, visible, out);
    make_quad(mn, dx, dz, mat_idx, visible, out);
    make_quad(mn, dy, dz, mat_idx, visible, out);
    make_quad(mx, -dx, -dy, mat_idx, visible, out);
    make_quad(mx, -dx, -dz, mat_idx, visible, out);
    make_quad(mx, -dy, -dz, mat_idx, visible, out);
}

// uv_sphere() (objects.cuh:36-88)
void make_uv_sphere(V3 center, float radius, int mat_idx, int rings, int segments, bool shade_smooth, bool visible,
                    std::vector<crt_object>& out) {
    std::vector<V3> verts, normals;
    std::vector<float> uvs;
    for (int lat = 0; lat <= rings; lat++) {
        float theta = k_pi * lat / rings;
        float sin_theta = sinf(theta), cos_theta = cosf(theta);
        for (int lon = 0; lon <= segments; lon++) {
            float phi = 2.0f * k_pi * lon / segments;
            float sin_phi = sinf(phi), cos_phi = cosf(phi);
            V3 vert = center + radius * v3(sin_theta * cos_phi, cos_theta, sin_theta * sin_phi);
            verts.push_back(vert);
            normals.push_back(unit(vert - center));
            uvs.push_back(phi / (2.0f * k_pi));
            uvs.push_back((1.0f - cos_theta) / 2.0f);
        }
    }
    for (int lat = 0; lat < rings; lat++) {
        for (int lon = 0; lon < segments; lon++) {
            int i0 = lat * (segments + 1) + lon, i1 = i0 + 1, i2 = i0 + (segments + 1), i3 = i2 + 1;
            out.push_back(make_tri(verts[i0], verts[i2], verts[i1], normals[i0], normals[i2], normals[i1], &uvs[2 * i0],
                                   &uvs[2 * i2], &uvs[2 * i1], mat_idx, shade_smooth, visible));
            out.push_back(make_tri(verts[i1], verts[i2], verts[i3], normals[i1], normals[i2], normals[i3], &uvs[2 * i1],
                                   &uvs[2 * i2], &uvs[2 * i3], mat_idx, shade_smooth, visible));
        }
    }
}

// sphere::sphere(center1, center2, radius, mat) (hittable.cuh:164-170)
crt_object make_sphere(V3 c1, V3 c2, float radius, int mat_idx) {
    crt_object o = blank_object(CRT_OBJ_SPHERE, mat_idx, true);
    V3 dir = c2 - c1;
    put(o.p0, c1); put(o.p1, dir);
    o.p2[0] = radius;
    V3 rvec = v3(radius, radius, radius);
    V3 at0 = c1 + dir * 0.0f, at1 = c1 + dir * 1.0f;
    float lo[3], hi[3];
    put(lo, at0 - rvec); put(hi, at0 + rvec);
    crt::Box b1 = crt::box_from_points(lo, hi);
    put(lo, at1 - rvec); put(hi, at1 + rvec);
    crt::Box b2 = crt::box_from_points(lo, hi);
    set_box(o, crt::box_union(b1, b2));
    put(o.anchor, (c1 + c2) / 2.0f);
    return o;
}

// hittable::translate (hittable.cuh:77-81)
void translate_object(crt_object& o, V3 offset) {
    float off[3];
    put(off, offset);
    put(o.translation, v3(o.translation) + offset);
    set_box(o, crt::box_offset(box_of(o), off));
    put(o.anchor, v3(o.anchor) + offset);
}

// hittable::rotate (hittable.cuh:90-127)
void rotate_object(crt_object& o, V3 angles, V3 anchor) {
    put(o.anchor, anchor);
    V3 rad = v3(deg2rad(angles.x), deg2rad(angles.y), deg2rad(angles.z));
    from_euler(rad, o.rotation);
    inverse3(o.rotation, o.rotation_inv);
    const float* mn = o.bbox_min;
    const float* mx = o.bbox_max;
    V3 corners[8] = {v3(mn[0], mn[1], mn[2]), v3(mx[0], mn[1], mn[2]), v3(mn[0], mx[1], mn[2]), v3(mn[0], mn[1], mx[2]),
                     v3(mx[0], mx[1], mn[2]), v3(mn[0], mx[1], mx[2]), v3(mx[0], mn[1], mx[2]), v3(mx[0], mx[1], mx[2])};
    V3 rotated[8];
    for (int i = 0; i < 8; i++) rotated[i] = matvec(o.rotation, corners[i] - anchor) + anchor;
    float new_min[3], new_max[3];
    put(new_min, rotated[0]);
    put(new_max, rotated[0]);
    for (int i = 0; i < 8; i++) {
        float rc[3];
        put(rc, rotated[i]);
        for (int c = 0; c < 3; c++) {
            new_min[c] = fminf(new_min[c], rc[c]);
            new_max[c] = fmaxf(new_max[c], rc[c]);
        }
    }
    float lo[3], hi[3];
    put(lo, v3(new_min) + v3(o.translation));
    put(hi, v3(new_max) + v3(o.translation));
    set_box(o, crt::box_from_points(lo, hi));
}

// ------------------------------------------------------------------ OBJ files -----------
// Reproduces the vertex/triangle emission of the OBJ loader the reference vendors
// (external/OBJ_Loader/OBJ_Loader.h:431-713 file walk, :727-834 vertex assembly, :838-1003
// triangulation): faces in file order, triangles as they are, quads as (0,1,3),(1,2,3) by
// ascending position match, larger polygons ear-clipped from the front of the list.
struct ObjVertex {
    V3 pos;
    float uv[2];
    V3 normal;
};
bool same_pos(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

std::vector<std::string> split_keep_empty(const std::string& in, char token) {
    // behaviour of algorithm::split for single-character tokens (OBJ_Loader.h:321-357)
    std::vector<std::string> out;
    std::string temp;
    for (size_t i = 0; i < in.size(); i++) {
        if (in[i] == token) {
            if (!temp.empty()) { out.push_back(temp); temp.clear(); }
            else out.push_back("");
        } else if (i + 1 >= in.size()) {
            temp += in[i];
            out.push_back(temp);
            break;
        } else {
            temp += in[i];
        }
    }
    return out;
}
std::string first_token(const std::string& in) {
    size_t s = in.find_first_not_of(" \t");
    if (s == std::string::npos) return "";
    size_t e = in.find_first_of(" \t", s);
    return e == std::string::npos ? in.substr(s) : in.substr(s, e - s);
}
std::string tail_of(const std::string& in) {
    size_t token_start = in.find_first_not_of(" \t");
    if (token_start == std::string::npos) return "";
    size_t space_start = in.find_first_of(" \t", token_start);
    if (space_start == std::string::npos) return "";
    size_t tail_start = in.find_first_not_of(" \t", space_start);
    size_t tail_end = in.find_last_not_of(" \t\r");
    if (tail_start == std::string::npos || tail_end == std::string::npos || tail_end < tail_start) return "";
    return in.substr(tail_start, tail_end - tail_start + 1);
}
template <class T>
const T& obj_element(const std::vector<T>& elements, const std::string& index) {
    int idx = std::stoi(index);
    if (idx < 0) idx = (int)elements.size() + idx;
    else idx--;
    if (idx < 0 || idx >= (int)elements.size()) throw std::out_of_range("OBJ index out of range");
    return elements[idx];
}
float obj_magnitude(V3 v) { return sqrtf(powf(v.x, 2) + powf(v.y, 2) + powf(v.z, 2)); }
bool obj_same_side(V3 p1, V3 p2, V3 a, V3 b) {
    V3 cp1 = cross(b - a, p1 - a), cp2 = cross(b - a, p2 - a);
    return (cp1.x * cp2.x) + (cp1.y * cp2.y) + (cp1.z * cp2.z) >= 0;
}
bool obj_in_triangle(V3 point, V3 t1, V3 t2, V3 t3) {
    bool within = obj_same_side(point, t1, t2, t3) && obj_same_side(point, t2, t1, t3) && obj_same_side(point, t3, t1, t2);
    if (!within) return false;
    V3 n = cross(t2 - t1, t3 - t1);
    V3 bn = n / obj_magnitude(n);
    float d = (point.x * bn.x) + (point.y * bn.y) + (point.z * bn.z);
    V3 proj = bn * d;
    return obj_magnitude(proj) == 0;
}
void obj_triangulate(const std::vector<ObjVertex>& iv, std::vector<unsigned>& out) {
    if (iv.size() < 3) return;
    if (iv.size() == 3) { out.push_back(0); out.push_back(1); out.push_back(2); return; }
    std::vector<ObjVertex> tv = iv;
    auto emit = [&](V3 cur, V3 prev, V3 next, size_t limit) {
        for (size_t j = 0; j < limit; j++) {
            if (same_pos(iv[j].pos, cur)) out.push_back((unsigned)j);
            if (same_pos(iv[j].pos, prev)) out.push_back((unsigned)j);
            if (same_pos(iv[j].pos, next)) out.push_back((unsigned)j);
        }
    };
    while (true) {
        for (int i = 0; i < (int)tv.size(); i++) {
            V3 prev = (i == 0) ? tv[tv.size() - 1].pos : tv[i - 1].pos;
            V3 cur = tv[i].pos;
            V3 next = (i == (int)tv.size() - 1) ? tv[0].pos : tv[i + 1].pos;
            if (tv.size() == 3) {
                emit(cur, prev, next, tv.size());
                tv.clear();
                break;
            }
            if (tv.size() == 4) {
                emit(cur, prev, next, iv.size());
                V3 other = v3(0, 0, 0);
                for (size_t j = 0; j < tv.size(); j++)
                    if (!same_pos(tv[j].pos, cur) && !same_pos(tv[j].pos, prev) && !same_pos(tv[j].pos, next)) { other = tv[j].pos; break; }
                for (size_t j = 0; j < iv.size(); j++) {
                    if (same_pos(iv[j].pos, prev)) out.push_back((unsigned)j);
                    if (same_pos(iv[j].pos, next)) out.push_back((unsigned)j);
                    if (same_pos(iv[j].pos, other)) out.push_back((unsigned)j);
                }
                tv.clear();
                break;
            }
            // (the loader's interior-angle test can never reject a vertex)
            bool in_tri = false;
            for (size_t j = 0; j < iv.size(); j++) {
                if (obj_in_triangle(iv[j].pos, prev, cur, next) && !same_pos(iv[j].pos, prev) && !same_pos(iv[j].pos, cur) &&
                    !same_pos(iv[j].pos, next)) { in_tri = true; break; }
            }
            if (in_tri) continue;
            emit(cur, prev, next, iv.size());
            for (size_t j = 0; j < tv.size(); j++)
                if (same_pos(tv[j].pos, cur)) { tv.erase(tv.begin() + j); break; }
            i = -1;
        }
        if (out.empty()) break;
        if (tv.empty()) break;
    }
}

// load_model() (objects.cuh:156-193)
void load_model(const std::string& filename, int mat_idx, float scale, std::vector<crt_object>& out) {
    if (filename.size() < 4 || filename.substr(filename.size() - 4) != ".obj") fail("Unsupported model type");
    std::ifstream file(filename);
    if (!file.is_open()) fail("Could not find object file '" + filename + "'");
    std::vector<V3> positions, normals;
    std::vector<std::pair<float, float>> tcoords;
    std::string line;
    size_t before = out.size();
    int line_no = 0;
    try {
        while (std::getline(file, line)) {
            line_no++;
            if (!line.empty() && line.back() == '\r') line.pop_back();
            std::string tok = first_token(line);
            if (tok == "v") {
                auto s = split_keep_empty(tail_of(line), ' ');
                positions.push_back(v3(std::stof(s.at(0)), std::stof(s.at(1)), std::stof(s.at(2))));
            } else if (tok == "vt") {
                auto s = split_keep_empty(tail_of(line), ' ');
                tcoords.emplace_back(std::stof(s.at(0)), std::stof(s.at(1)));
            } else if (tok == "vn") {
                auto s = split_keep_empty(tail_of(line), ' ');
                normals.push_back(v3(std::stof(s.at(0)), std::stof(s.at(1)), std::stof(s.at(2))));
            } else if (tok == "f") {
                std::vector<ObjVertex> fv;
                bool no_normal = false;
                for (const std::string& sv : split_keep_empty(tail_of(line), ' ')) {
                    if (sv.empty()) continue;
                    auto parts = split_keep_empty(sv, '/');
                    ObjVertex v;
                    v.uv[0] = v.uv[1] = 0.0f;
                    v.normal = v3(0, 0, 0);
                    if (parts.size() == 1) {
                        v.pos = obj_element(positions, parts[0]);
                        no_normal = true;
                    } else if (parts.size() == 2) {
                        v.pos = obj_element(positions, parts[0]);
                        auto t = obj_element(tcoords, parts[1]);
                        v.uv[0] = t.first; v.uv[1] = t.second;
                        no_normal = true;
                    } else if (parts.size() == 3) {
                        v.pos = obj_element(positions, parts[0]);
                        if (!parts[1].empty()) {
                            auto t = obj_element(tcoords, parts[1]);
                            v.uv[0] = t.first; v.uv[1] = t.second;
                        }
                        v.normal = obj_element(normals, parts[2]);
                    } else {
                        continue;
                    }
                    fv.push_back(v);
                }
                if (no_normal && fv.size() >= 3) {
                    V3 n = cross(fv[0].pos - fv[1].pos, fv[2].pos - fv[1].pos);
                    for (auto& v : fv) v.normal = n;
                }
                std::vector<unsigned> idx;
                obj_triangulate(fv, idx);
                for (size_t i = 0; i + 2 < idx.size(); i += 3) {
                    const ObjVertex &a = fv[idx[i]], &b = fv[idx[i + 1]], &c = fv[idx[i + 2]];
                    out.push_back(make_tri(a.pos * scale, b.pos * scale, c.pos * scale, a.normal, b.normal, c.normal, a.uv,
                                           b.uv, c.uv, mat_idx));
                }
            }
        }
    } catch (const std::exception& e) {
        fail("Malformed OBJ file '" + filename + "' (line " + std::to_string(line_no) + "): " + e.what());
    }
    if (out.size() == before && positions.empty()) fail("Could not find object file '" + filename + "'");
}

// ------------------------------------------------------------------ scene parsing -------
struct Loader {
    crt_host_scene& hs;
    std::string base_dir;
    std::unordered_map<std::string, int> texture_map, material_map;

    explicit Loader(crt_host_scene& s) : hs(s) {}

    std::string resolve(const std::string& file) const {
        if (file.empty() || file[0] == '/' || base_dir.empty()) return file;
        return base_dir + "/" + file;
    }

    int add_texture(const crt_texture& t, std::vector<uint8_t>&& pixels = {}) {
        hs.textures.push_back(t);
        hs.image_store.push_back(std::move(pixels));
        return (int)hs.textures.size() - 1;
    }

    // parse_texture (scene.cuh:241-290)
    int parse_texture(const Node& node) {
        if (node.is_scalar()) {
            std::string name = node.as_string();
            auto it = texture_map.find(name);
            if (it == texture_map.end()) fail("Texture '" + name + "' not found" + at_line(node));
            return it->second;
        }
        std::string type = required(node, "type", "Texture type missing or invalid" + at_line(node), to_string);
        crt_texture t;
        std::memset(&t, 0, sizeof(t));
        t.odd_idx = t.even_idx = -1;
        if (type == "solid_color") {
            V3 c = required_vec3(node, "color", "Solid color 'color' vector missing or invalid" + at_line(node));
            t.type = CRT_TEX_SOLID_COLOR;
            put(t.color, c);
            return add_texture(t);
        }
        if (type == "checker") {
            const Node& odd = node["odd"];
            const Node& even = node["even"];
            if (!odd.defined() || !even.defined()) fail("Checker texture must have 'odd' and 'even' fields" + at_line(node));
            int odd_idx = parse_texture(odd);
            int even_idx = parse_texture(even);
            float scale = optional<float>(node, "scale", 1.0f, to_float);
            t.type = CRT_TEX_CHECKER;
            t.odd_idx = odd_idx;
            t.even_idx = even_idx;
            t.inv_scale = 1.0f / scale;                      // texture.cuh:111
            return add_texture(t);
        }
        if (type == "image") {
            std::string filename = required(node, "filename", "Image texture 'filename' string missing or invalid" + at_line(node), to_string);
            float hdr_gamma = optional<float>(node, "gamma", 2.2f, to_float);
            float hdr_scale = optional<float>(node, "scale", 1.0f, to_float);
            int desired_channels = optional<int>(node, "channels", 0, to_int);
            bool flip_y = optional<bool>(node, "flip_y", false, to_bool);
            int w = 0, h = 0, ch = 0;
            std::vector<uint8_t> px;
            std::string err;
            if (!crt::load_image_file(resolve(filename), hdr_gamma, hdr_scale, desired_channels, flip_y, w, h, ch, px, err))
                fail("Could not load image '" + filename + "': " + err);
            if (desired_channels != 0) ch = desired_channels;   // stbi_load returns the converted layout
            if (ch == 3) {                                      // image::add_alpha_channel (texture.cuh:163-183)
                std::vector<uint8_t> rgba((size_t)w * h * 4);
                for (size_t i = 0; i < (size_t)w * h; i++) {
                    rgba[4 * i + 0] = px[3 * i + 0]; rgba[4 * i + 1] = px[3 * i + 1]; rgba[4 * i + 2] = px[3 * i + 2];
                    rgba[4 * i + 3] = 255;
                }
                px.swap(rgba);
                ch = 4;
            } else if (ch != 1 && ch != 4) {
                fail("Images with 2 channels not supported");
            }
            t.type = CRT_TEX_IMAGE;
            t.width = w; t.height = h; t.channels = ch;
            return add_texture(t, std::move(px));
        }
        fail("Unknown texture type '" + type + "'" + at_line(node));
    }

    // parse_material (scene.cuh:327-379)
    int parse_material(const Node& node) {
        if (node.is_scalar()) {
            std::string name = node.as_string();
            auto it = material_map.find(name);
            if (it == material_map.end()) fail("Material '" + name + "' not found" + at_line(node));
            return it->second;
        }
        std::string type = required(node, "type", "Material type missing or invalid" + at_line(node), to_string);
        crt_material m;
        std::memset(&m, 0, sizeof(m));
        m.tex_idx = -1;
        if (type == "lambertian") {
            const Node& tex = node["texture"];
            if (!tex.defined()) fail("Lambertian material must have 'texture' field" + at_line(node));
            m.type = CRT_MAT_LAMBERTIAN;
            m.tex_idx = parse_texture(tex);
        } else if (type == "metal") {
            V3 c = required_vec3(node, "color", "Metal material 'color' vector missing or invalid" + at_line(node));
            m.type = CRT_MAT_METAL;
            put(m.albedo, c);
            m.fuzz = optional<float>(node, "fuzz", 0.0f, to_float);
        } else if (type == "dielectric") {
            m.type = CRT_MAT_DIELECTRIC;
            m.ior = optional<float>(node, "ior", 1.5f, to_float);
        } else if (type == "diffuse_light") {
            const Node& tex = node["texture"];
            if (!tex.defined()) fail("Diffuse light material must have 'texture' field" + at_line(node));
            m.type = CRT_MAT_DIFFUSE_LIGHT;
            m.tex_idx = parse_texture(tex);
        } else {
            fail("Unknown material type '" + type + "'" + at_line(node));
        }
        hs.materials.push_back(m);
        return (int)hs.materials.size() - 1;
    }

    int material_of(const Node& node, const char* what) {
        const Node& mat = node["material"];
        if (!mat.defined()) fail(std::string(what) + " object must have 'material' field" + at_line(node));
        return parse_material(mat);
    }

    std::vector<int> append(std::vector<crt_object>& made, int normal_idx) {
        std::vector<int> idx;
        for (crt_object& o : made) {
            o.normal_idx = normal_idx;
            hs.objects.push_back(o);
            idx.push_back((int)hs.objects.size() - 1);
        }
        return idx;
    }

    // parse_object (scene.cuh:419-595)
    std::vector<int> parse_object(const Node& node) {
        std::string type = required(node, "type", "Object type missing or invalid" + at_line(node), to_string);
        int normal_idx = -1;
        const Node& norm = node["normal"];
        if (norm.defined()) normal_idx = parse_texture(norm);
        std::vector<crt_object> made;

        if (type == "sphere") {
            V3 center = required_vec3(node, "center", "Sphere 'center' vector missing or invalid" + at_line(node));
            V3 center2 = optional_vec3(node, "center2", center);
            float radius = required(node, "radius", "Sphere 'radius' float missing or invalid" + at_line(node), to_float);
            int mat = material_of(node, "Sphere");
            made.push_back(make_sphere(center, center2, radius, mat));
            return append(made, normal_idx);
        }
        if (type == "uv_sphere") {
            V3 center = required_vec3(node, "center", "UV sphere 'center' vector missing or invalid" + at_line(node));
            float radius = required(node, "radius", "UV sphere 'radius' float missing or invalid" + at_line(node), to_float);
            int rings = optional<int>(node, "rings", 5, to_int);
            int segments = optional<int>(node, "segments", 10, to_int);
            bool smooth = optional<bool>(node, "shade_smooth", false, to_bool);
            int mat = material_of(node, "UV sphere");
            // The reference passes normal_idx in the `visible` slot (scene.cuh:460 vs
            // objects.cuh:36): a UV sphere whose normal texture has index 0 is invisible.
            bool visible = normal_idx != 0;
            make_uv_sphere(center, radius, mat, rings, segments, smooth, visible, made);
            return append(made, normal_idx);
        }
        if (type == "tri") {
            V3 v0 = required_vec3(node, "v0", "Tri 'v0' vector missing or invalid" + at_line(node));
            V3 v1 = required_vec3(node, "v1", "Tri 'v1' vector missing or invalid" + at_line(node));
            V3 v2 = required_vec3(node, "v2", "Tri 'v2' vector missing or invalid" + at_line(node));
            float uv0[2], uv1[2], uv2[2];
            optional_vec2(node, "uv0", 0, 1, uv0);
            optional_vec2(node, "uv1", 1, 0, uv1);
            optional_vec2(node, "uv2", 0, 0, uv2);
            V3 normal = unit(cross(v1 - v0, v2 - v0));
            int mat = material_of(node, "Tri");
            made.push_back(make_tri(v0, v1, v2, normal, normal, normal, uv0, uv1, uv2, mat));
            return append(made, normal_idx);
        }
        if (type == "quad") {
            V3 corner = required_vec3(node, "corner", "Quad 'corner' vector missing or invalid" + at_line(node));
            V3 u = required_vec3(node, "edge1", "Quad 'edge1' vector missing or invalid" + at_line(node));
            V3 v = required_vec3(node, "edge2", "Quad 'edge2' vector missing or invalid" + at_line(node));
            int mat = material_of(node, "Quad");
            make_quad(corner, u, v, mat, true, made);
            return append(made, normal_idx);
        }
        if (type == "box") {
            V3 c1 = required_vec3(node, "corner1", "Box 'corner1' vector missing or invalid" + at_line(node));
            V3 c2 = required_vec3(node, "corner2", "Box 'corner2' vector missing or invalid" + at_line(node));
            int mat = material_of(node, "Box");
            make_box(c1, c2, mat, true, made);
            return append(made, normal_idx);
        }
        if (type == "model") {
            std::string filename = required(node, "filename", "Model object 'filename' string missing or invalid" + at_line(node), to_string);
            float scale = optional<float>(node, "scale", 1.0f, to_float);
            int mat = material_of(node, "Model");
            load_model(resolve(filename), mat, scale, made);
            return append(made, normal_idx);
        }
        if (type == "translate") {
            V3 offset = required_vec3(node, "offset", "Translate 'offset' vector missing or invalid" + at_line(node));
            const Node& obj = node["object"];
            if (!obj.defined()) fail("Translate object must have 'object' field" + at_line(node));
            std::vector<int> idx = parse_object(obj);
            for (int i : idx) translate_object(hs.objects[i], offset);
            return idx;
        }
        if (type == "rotate") {
            V3 angles = required_vec3(node, "angles", "Rotate 'angles' vector missing or invalid" + at_line(node));
            const Node& obj = node["object"];
            if (!obj.defined()) fail("Rotate object must have 'object' field" + at_line(node));
            std::vector<int> idx = parse_object(obj);
            V3 anchor = v3(0, 0, 0);                       // calculate_center (objects.cuh:202-214)
            if (!idx.empty()) {
                crt::Box combined = box_of(hs.objects[idx[0]]);
                for (size_t i = 1; i < idx.size(); i++) combined = crt::box_union(combined, box_of(hs.objects[idx[i]]));
                anchor = v3(0.5f * (combined.mn[0] + combined.mx[0]), 0.5f * (combined.mn[1] + combined.mx[1]),
                            0.5f * (combined.mn[2] + combined.mx[2]));
            }
            const Node& an = node["anchor"];
            if (an.defined() && an.is_seq()) anchor = required_vec3(node, "anchor", "Rotate 'anchor' vector invalid" + at_line(node));
            for (int i : idx) rotate_object(hs.objects[i], angles, anchor);
            return idx;
        }
        fail("Unknown object type '" + type + "'" + at_line(node));
    }

    void load(const Node& config) {
        // parse_settings (scene.cuh:204-231)
        const Node& cam = config["camera"];
        if (!cam.defined() || !cam.is_map()) fail("Camera settings missing or invalid");
        const Node& ren = config["render"];
        if (!ren.defined() || !ren.is_map()) fail("Render settings missing or invalid");
        std::string cam_type = required(cam, "type", "Camera type missing or invalid" + at_line(cam), to_string);
        put(hs.cam_from, required_vec3(cam, "from", "Camera 'from' vector missing or invalid" + at_line(cam)));
        put(hs.cam_to, required_vec3(cam, "to", "Camera 'to' vector missing or invalid" + at_line(cam)));
        put(hs.cam_up, optional_vec3(cam, "up", v3(0, 1, 0)));
        if (cam_type == "perspective") {
            hs.cam_type = CRT_CAM_PERSPECTIVE;
            hs.cam_vfov = required(cam, "vfov", "Camera 'vfov' float missing or invalid" + at_line(cam), to_float);
            hs.cam_defocus_angle = optional<float>(cam, "defocus_angle", 0.0f, to_float);
            hs.cam_focus_dist = optional<float>(cam, "focus_dist", 10.0f, to_float);
        } else if (cam_type == "orthographic") {
            hs.cam_type = CRT_CAM_ORTHOGRAPHIC;
            hs.cam_width = required(cam, "width", "Camera 'width' float missing or invalid" + at_line(cam), to_float);
        } else {
            fail("Unknown camera type '" + cam_type + "', must be either 'perspective' or 'orthographic'" + at_line(cam));
        }
        hs.render.image_width = required(ren, "width", "Render 'width' integer missing or invalid" + at_line(ren), to_int);
        hs.render.image_height = required(ren, "height", "Render 'height' integer missing or invalid" + at_line(ren), to_int);
        hs.render.samples_per_pixel = required(ren, "samples", "Render 'samples' integer missing or invalid" + at_line(ren), to_int);
        hs.render.max_bounces = optional<int>(ren, "max_bounces", 50, to_int);
        hs.render.gamma = optional<float>(ren, "gamma", 2.0f, to_float);
        if (hs.render.image_width <= 0 || hs.render.image_height <= 0) fail("Render 'width'/'height' must be positive");
        hs.rebuild_camera();

        // parse_texture_list (scene.cuh:299-315)
        const Node& texs = config["textures"];
        if (texs.defined()) {
            if (!texs.is_seq()) fail("Textures must be a sequence" + at_line(texs));
            for (const Node& n : texs.seq) {
                std::string name = required(n, "name", "Texture 'name' string missing or invalid" + at_line(n), to_string);
                texture_map[name] = parse_texture(n);
            }
        }
        // parse_material_list (scene.cuh:390-406)
        const Node& mats = config["materials"];
        if (mats.defined()) {
            if (!mats.is_seq()) fail("Materials must be a sequence" + at_line(mats));
            for (const Node& n : mats.seq) {
                std::string name = required(n, "name", "Material 'name' string missing or invalid" + at_line(n), to_string);
                material_map[name] = parse_material(n);
            }
        }
        // parse_object_list (scene.cuh:607-618)
        const Node& objs = config["objects"];
        if (!objs.defined() || !objs.is_seq()) fail("Objects missing or invalid");
        for (const Node& n : objs.seq) parse_object(n);
        // parse_environment (scene.cuh:628-642)
        const Node& env = config["environment"];
        if (!env.defined() || !env.is_map()) fail("Environment missing or invalid");
        const Node& env_tex = env["texture"];
        if (!env_tex.defined()) fail("Environment must have 'texture' field");
        hs.background_index = parse_texture(env_tex);
    }
};

int finish_scene(crt_host_scene* hs) {
    std::vector<crt::Box> boxes(hs->objects.size());
    std::vector<uint8_t> vis(hs->objects.size());
    for (size_t i = 0; i < hs->objects.size(); i++) {
        boxes[i] = box_of(hs->objects[i]);
        vis[i] = hs->objects[i].visible != 0;
    }
    crt::build_bvh4(boxes, vis, hs->bvh_min, hs->bvh_max, hs->bvh_children, hs->bvh_object_index);
    hs->refresh_desc();
    return CRT_OK;
}

int parse_common(const Node& root, const char* base_dir, crt_host_scene** out) {
    crt_host_scene* hs = new crt_host_scene();
    try {
        Loader loader(*hs);
        if (base_dir) loader.base_dir = base_dir;
        loader.load(root);
        finish_scene(hs);
    } catch (const SceneError& e) {
        delete hs;
        return crt::set_error(CRT_ERR_PARSE, e.what());
    } catch (const yamllite::ParseError& e) {
        delete hs;
        return crt::set_error(CRT_ERR_PARSE, std::string("YAML error (line ") + std::to_string(e.line) + "): " + e.what());
    } catch (const std::exception& e) {
        delete hs;
        return crt::set_error(CRT_ERR_PARSE, e.what());
    }
    *out = hs;
    return CRT_OK;
}

}  // namespace

void crt_host_scene::rebuild_camera() {
    crt_camera& c = camera;
    std::memset(&c, 0, sizeof(c));
    c.type = cam_type;
    const int W = render.image_width, H = render.image_height;
    V3 from = v3(cam_from), to = v3(cam_to), up = v3(cam_up);
    if (cam_type == CRT_CAM_PERSPECTIVE) {                       // camera.cuh:121-151
        V3 center = from;
        float theta = deg2rad(cam_vfov);
        float h = std::tan(theta / 2);
        float viewport_height = 2 * h * cam_focus_dist;
        float viewport_width = viewport_height * (float(W) / H);
        V3 w = unit(from - to);
        V3 u = unit(cross(up, w));
        V3 v = cross(w, u);
        V3 viewport_u = viewport_width * u;
        V3 viewport_v = viewport_height * -v;
        V3 du = viewport_u / (float)W;
        V3 dv = viewport_v / (float)H;
        V3 upper_left = center - (cam_focus_dist * w) - viewport_u / 2.0f - viewport_v / 2.0f;
        V3 llc = upper_left + 0.5f * (du + dv);
        float defocus_radius = cam_focus_dist * std::tan(deg2rad(cam_defocus_angle / 2));
        put(c.center, center); put(c.lower_left_corner, llc); put(c.pixel_delta_u, du); put(c.pixel_delta_v, dv);
        put(c.u, u); put(c.v, v); put(c.w, w);
        put(c.defocus_disk_u, u * defocus_radius);
        put(c.defocus_disk_v, v * defocus_radius);
        c.defocus_angle = cam_defocus_angle;
    } else {                                                     // camera.cuh:204-220
        float aspect_ratio = float(W) / float(H);
        float viewport_width = cam_width;
        float viewport_height = viewport_width / aspect_ratio;
        V3 w = -unit(from - to);
        V3 u = -unit(cross(up, w));
        V3 v = -cross(w, u);
        V3 viewport_u = viewport_width * u;
        V3 viewport_v = viewport_height * -v;
        V3 du = viewport_u / (float)W;
        V3 dv = viewport_v / (float)H;
        V3 upper_left = from - viewport_u / 2.0f - viewport_v / 2.0f;
        V3 llc = upper_left + 0.5f * (du + dv);
        put(c.lower_left_corner, llc); put(c.pixel_delta_u, du); put(c.pixel_delta_v, dv);
        put(c.u, u); put(c.v, v); put(c.w, w);
        c.defocus_angle = 0.0f;
    }
}

void crt_host_scene::refresh_desc() {
    for (size_t i = 0; i < textures.size(); i++)
        textures[i].pixels = (textures[i].type == CRT_TEX_IMAGE && !image_store[i].empty()) ? image_store[i].data() : nullptr;
    desc.object_count = (int)objects.size();
    desc.material_count = (int)materials.size();
    desc.texture_count = (int)textures.size();
    desc.background_index = background_index;
    desc.objects = objects.data();
    desc.materials = materials.data();
    desc.textures = textures.data();
    desc.bvh.slot_count = (int)bvh_children.size();
    desc.bvh.min_bounds = bvh_min.data();
    desc.bvh.max_bounds = bvh_max.data();
    desc.bvh.children = bvh_children.data();
    desc.bvh.object_index = bvh_object_index.data();
    desc.camera = camera;
    desc.render = render;
}

extern "C" {

int crt_scene_load_yaml(const char* scene_file, const char* base_dir, crt_host_scene** out) {
    crt::clear_error();
    if (!scene_file || !out) return crt::set_error(CRT_ERR_INVALID, "crt_scene_load_yaml: null argument");
    Node root;
    try {
        root = yamllite::parse_file(scene_file);
    } catch (const yamllite::ParseError& e) {
        return crt::set_error(CRT_ERR_PARSE, std::string("Failed to load scene file: ") + e.what() + " (line " + std::to_string(e.line) + ")");
    }
    return parse_common(root, base_dir, out);
}

int crt_scene_parse_yaml(const char* yaml_text, const char* base_dir, crt_host_scene** out) {
    crt::clear_error();
    if (!yaml_text || !out) return crt::set_error(CRT_ERR_INVALID, "crt_scene_parse_yaml: null argument");
    Node root;
    try {
        root = yamllite::parse(yaml_text);
    } catch (const yamllite::ParseError& e) {
        return crt::set_error(CRT_ERR_PARSE, std::string("Failed to load scene file: ") + e.what() + " (line " + std::to_string(e.line) + ")");
    }
    return parse_common(root, base_dir, out);
}

int crt_host_scene_set_render(crt_host_scene* hs, int width, int height, int samples, int max_bounces) {
    crt::clear_error();
    if (!hs) return crt::set_error(CRT_ERR_INVALID, "crt_host_scene_set_render: null scene");
    if (width > 0) hs->render.image_width = width;
    if (height > 0) hs->render.image_height = height;
    if (samples > 0) hs->render.samples_per_pixel = samples;
    if (max_bounces > 0) hs->render.max_bounces = max_bounces;
    hs->rebuild_camera();
    hs->refresh_desc();
    return CRT_OK;
}

const crt_scene_desc* crt_host_scene_desc(const crt_host_scene* hs) { return hs ? &hs->desc : nullptr; }

void crt_host_scene_destroy(crt_host_scene* hs) { delete hs; }

}  // extern "C"
