// CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see rt_oracle.hpp for the full header note).
// PARITY UNPINNED against a reference binary; pinned by hand-derived known-answer tests.
//
// Restates, class by class, the Rust sources under /root/reference/raytracer-rs/src/.
// Deliberate, documented departures from a literal restatement:
//   * rand::thread_rng (OS-seeded) -> counter-based streams, section "RNG";
//   * ConstantMedium draws its free-flight sample from a stream position keyed on
//     (path, bounce, medium) instead of "the next thread_rng value", so that the
//     draw does not depend on traversal order;
//   * world.intersect evaluates every ConstantMedium AFTER all surfaces, in creation
//     order (world_intersect below).  A medium hit overrides the closest hit found so
//     far even when it lies beyond it (shapes.rs:676-683 divides by the ray length
//     twice, so final_t may exceed t_max; scene.rs:33-37 and bvh_node.rs:158-163 take
//     the later result unchecked), which makes the reference's own answer depend on
//     where its randomly built tree happens to visit the medium.  "Media last" is the
//     order a Scene gets when its media are added last, as cornell_smoke does;
//   * BVHNode::intersect resolves EXACT t ties between its two children towards the
//     primitive created later (higher id).  The reference resolves them towards
//     whichever child its randomly built tree visits second, i.e. not reproducibly;
//   * BVHNode::from_vector sorts with a strict '<' on the box minimum (the
//     reference's epsilon comparator is not a total order and its outcome is
//     unspecified).
#include "rt_oracle.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr float INF = std::numeric_limits<float>::infinity();
constexpr float F32_EPSILON = 1.1920929e-7f;   // f32::EPSILON
constexpr float PI_F = 3.14159265358979323846f;  // std::f32::consts::PI

// ----------------------------------------------------------------------------
// glam::Vec3A restated (lane-wise f32, no FMA).
// ----------------------------------------------------------------------------
struct V3 {
    float x, y, z;
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator/(V3 a, V3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
// glam dot: (x*x + y*y) + z*z
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float length_squared(V3 a) { return dot(a, a); }
inline float length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(V3 a) { return a / length(a); }
inline V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// SSE minps/maxps semantics (second operand when unordered)
inline float sse_min(float a, float b) { return a < b ? a : b; }
inline float sse_max(float a, float b) { return a > b ? a : b; }
inline V3 vmin(V3 a, V3 b) { return {sse_min(a.x, b.x), sse_min(a.y, b.y), sse_min(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return {sse_max(a.x, b.x), sse_max(a.y, b.y), sse_max(a.z, b.z)}; }
inline float axis_of(V3 a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }
// Rust f32::min / f32::max (NaN-ignoring)
inline float rmin(float a, float b) { return std::fmin(a, b); }
inline float rmax(float a, float b) { return std::fmax(a, b); }
// Rust f32::clamp
inline float rclamp(float v, float lo, float hi) {
    if (v < lo) return lo;
    if (v > hi) return hi;
    return v;
}

// glam::Mat4, column major.  Only what the reference calls.
struct M4 {
    float c[4][4];  // c[col][row]
};
inline M4 m4_identity() {
    M4 m{};
    for (int i = 0; i < 4; i++) m.c[i][i] = 1.0f;
    return m;
}
// Mat4::from_rotation_y: cols (cos,0,-sin,0) (0,1,0,0) (sin,0,cos,0) (0,0,0,1)
inline M4 m4_rotation_y(float a) {
    float s = std::sin(a), co = std::cos(a);
    M4 m = m4_identity();
    m.c[0][0] = co;
    m.c[0][2] = -s;
    m.c[2][0] = s;
    m.c[2][2] = co;
    return m;
}
inline M4 m4_translation(V3 t) {
    M4 m = m4_identity();
    m.c[3][0] = t.x;
    m.c[3][1] = t.y;
    m.c[3][2] = t.z;
    return m;
}
// Mat4 * Mat4: result column j = sum_k a.col[k] * b.c[j][k], accumulated left to right
inline M4 m4_mul(const M4& a, const M4& b) {
    M4 r{};
    for (int j = 0; j < 4; j++)
        for (int row = 0; row < 4; row++) {
            float acc = a.c[0][row] * b.c[j][0];
            acc = acc + a.c[1][row] * b.c[j][1];
            acc = acc + a.c[2][row] * b.c[j][2];
            acc = acc + a.c[3][row] * b.c[j][3];
            r.c[j][row] = acc;
        }
    return r;
}
// transform_point3a: ((x_axis*v.x + y_axis*v.y) + z_axis*v.z) + w_axis   (no divide)
inline V3 m4_point(const M4& m, V3 v) {
    V3 r;
    float* o[3] = {&r.x, &r.y, &r.z};
    for (int row = 0; row < 3; row++)
        *o[row] = ((m.c[0][row] * v.x + m.c[1][row] * v.y) + m.c[2][row] * v.z) + m.c[3][row];
    return r;
}
inline V3 m4_vector(const M4& m, V3 v) {
    V3 r;
    float* o[3] = {&r.x, &r.y, &r.z};
    for (int row = 0; row < 3; row++)
        *o[row] = (m.c[0][row] * v.x + m.c[1][row] * v.y) + m.c[2][row] * v.z;
    return r;
}
// Mat4::inverse: general cofactor inverse in the formulation of GLM's compute_inverse,
// which glam's Mat4::inverse follows.
inline M4 m4_inverse(const M4& mm) {
    const float(*m)[4] = mm.c;
    float c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    float f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11};
    float f3[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
    float v0[4] = {m[1][0], m[0][0], m[0][0], m[0][0]}, v1[4] = {m[1][1], m[0][1], m[0][1], m[0][1]};
    float v2[4] = {m[1][2], m[0][2], m[0][2], m[0][2]}, v3_[4] = {m[1][3], m[0][3], m[0][3], m[0][3]};
    const float sa[4] = {1, -1, 1, -1}, sb[4] = {-1, 1, -1, 1};
    M4 inv{};
    for (int i = 0; i < 4; i++) {
        float i0 = (v1[i] * f0[i] - v2[i] * f1[i]) + v3_[i] * f2[i];
        float i1 = (v0[i] * f0[i] - v2[i] * f3[i]) + v3_[i] * f4[i];
        float i2 = (v0[i] * f1[i] - v1[i] * f3[i]) + v3_[i] * f5[i];
        float i3 = (v0[i] * f2[i] - v1[i] * f4[i]) + v2[i] * f5[i];
        inv.c[0][i] = i0 * sa[i];
        inv.c[1][i] = i1 * sb[i];
        inv.c[2][i] = i2 * sa[i];
        inv.c[3][i] = i3 * sb[i];
    }
    float d0 = m[0][0] * inv.c[0][0], d1 = m[0][1] * inv.c[1][0];
    float d2 = m[0][2] * inv.c[2][0], d3 = m[0][3] * inv.c[3][0];
    float det = (d0 + d1) + (d2 + d3);
    float rcp = 1.0f / det;
    for (int j = 0; j < 4; j++)
        for (int r = 0; r < 4; r++) inv.c[j][r] = inv.c[j][r] * rcp;
    return inv;
}

// ----------------------------------------------------------------------------
// RNG.  Specification (the HIP path implements the same spec independently):
//   mix32(x): x^=x>>16; x*=0x7feb352d; x^=x>>15; x*=0x846ca68b; x^=x>>16
//   path stream (seed,pixel,sample): h=mix32(seed^0x51ED270B); k0=mix32(pixel^h);
//        k1=mix32(k0^mix32(sample+0x68E31DA4));  draw at counter c = mix32((k0+c)^k1)
//   counters of one path: every consumer starts at a fixed counter and counts up from there, so no
//   consumer's draws depend on how many draws another one's rejection loop took:
//        pixel jitter s, t, then the ray time     c = 0xFF000000, 0xFF000001, 0xFF000002...
//        lens disk (random_in_unit_disk)           c = 0xFE000000...
//        Material::scatter at bounce b (b < 254)   c = b<<24 ...
//   f32 in [0,1)   (rand gen::<f32>):        (u32>>8) * 2^-24
//   f32 in [lo,hi) (rand gen_range):          v=bits(0x3f800000|(u32>>9))-1; r=v*(hi-lo)+lo; redraw if r>=hi
//   f64 in [0,1)   (rand gen_range(0.0..1.0)): u64=(hi<<32|lo); bits(0x3ff0..|(u64>>12))-1
//   medium m at bounce b: salt=mix32(m+0x3C6EF372);
//        word j = mix32((k0+0x80000000+2b+j)^k1^salt), hi=word0, lo=word1
//   index in [0,n): (u32*n)>>32
// ----------------------------------------------------------------------------
inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
inline float bits_f32(uint32_t b) {
    float f;
    std::memcpy(&f, &b, 4);
    return f;
}
inline double bits_f64(uint64_t b) {
    double f;
    std::memcpy(&f, &b, 8);
    return f;
}
struct Stream {
    uint32_t k0, k1, n;
    uint32_t next_u32() {
        uint32_t r = mix32((k0 + n) ^ k1);
        n++;
        return r;
    }
    float gen_f32() { return (float)(next_u32() >> 8) * (1.0f / 16777216.0f); }
    float gen_range(float lo, float hi) {
        float scale = hi - lo;
        for (;;) {
            float v12 = bits_f32(0x3f800000U | (next_u32() >> 9));
            float v01 = v12 - 1.0f;
            float r = v01 * scale + lo;
            if (r < hi) return r;
        }
    }
    uint32_t gen_index(uint32_t n_) { return (uint32_t)(((uint64_t)next_u32() * n_) >> 32); }
};
constexpr uint32_t CTR_CAMERA = 0xFF000000U, CTR_LENS = 0xFE000000U;
inline Stream path_stream(uint32_t seed, uint32_t pixel, uint32_t sample) {
    uint32_t h = mix32(seed ^ 0x51ED270BU);
    uint32_t k0 = mix32(pixel ^ h);
    uint32_t k1 = mix32(k0 ^ mix32(sample + 0x68E31DA4U));
    return Stream{k0, k1, 0};
}
inline double medium_xi(const Stream& s, uint32_t bounce, uint32_t medium) {
    uint32_t salt = mix32(medium + 0x3C6EF372U);
    uint32_t hi = mix32((s.k0 + 0x80000000U + 2 * bounce) ^ s.k1 ^ salt);
    uint32_t lo = mix32((s.k0 + 0x80000001U + 2 * bounce) ^ s.k1 ^ salt);
    uint64_t u = ((uint64_t)hi << 32) | lo;
    return bits_f64((u >> 12) | 0x3FF0000000000000ULL) - 1.0;
}
// scene construction / BVH axis streams (stand-ins for thread_rng on the main thread)
inline Stream scene_stream(uint32_t seed) { return Stream{mix32(seed ^ 0xA511E9B3U), mix32(seed ^ 0x2545F491U), 0}; }
inline Stream bvh_stream(uint32_t seed) { return Stream{mix32(seed ^ 0x1B873593U), mix32(seed ^ 0xCC9E2D51U), 0}; }

// "thread_rng()" of the path being traced on this thread
struct PathCtx {
    Stream rng;
    uint32_t bounce;
    bool fixed_xi;  // known-answer probes only
    double xi;
};
thread_local PathCtx* g_path = nullptr;
thread_local bool g_media_deferred = false;  // see world_intersect
thread_local uint64_t g_cnt_rays = 0, g_cnt_box = 0, g_cnt_prim = 0;
// RT_ORACLE_FAST (the build bench.py times as the CPU baseline, oracle/Makefile): the box-test and primitive-test
// counters are compiled out -- the reference has none -- while the ray counter, one increment per world.intersect call,
// stays (it is the numerator of Mrays/s).  Results are the same bit for bit (tests/test_cpu_suite.py).
#ifdef RT_ORACLE_FAST
#define ORC_COUNT(c) ((void)0)
#else
#define ORC_COUNT(c) ((c)++)
#endif

// src/vec3_helpers.rs:7-60
inline bool is_near_zero(V3 v) {
    return std::fabs(v.x) < F32_EPSILON && std::fabs(v.y) < F32_EPSILON && std::fabs(v.z) < F32_EPSILON;
}
inline V3 random_vec(Stream& r) {
    float x = r.gen_f32(), y = r.gen_f32(), z = r.gen_f32();
    return {x, y, z};
}
inline V3 random_range(Stream& r, float lo, float hi) {
    float x = r.gen_range(lo, hi), y = r.gen_range(lo, hi), z = r.gen_range(lo, hi);
    return {x, y, z};
}
inline V3 random_in_unit_sphere(Stream& r) {
    V3 p = random_range(r, -1.0f, 1.0f);
    while (length_squared(p) >= 1.0f) p = random_range(r, -1.0f, 1.0f);
    return p;
}
inline V3 random_unit_vector(Stream& r) { return normalize(random_in_unit_sphere(r)); }
// src/vec3_helpers.rs:47-60 -- returns the first point with |p|^2 >= 1 (sic)
inline V3 random_in_unit_disk(Stream& r) {
    for (;;) {
        float x = r.gen_range(-1.0f, 1.0f);
        float y = r.gen_range(-1.0f, 1.0f);
        V3 p{x, y, 0.0f};
        if (length_squared(p) >= 1.0f) return p;
    }
}
// src/vec3_helpers.rs:62-75
inline V3 reflect(V3 v, V3 n) {
    float scalar = 2.0f * dot(v, n);
    V3 scaled = scalar * n;
    return v - scaled;
}
inline V3 refract(V3 uv, V3 n, float etai_over_etat) {
    float dot_p = dot(n, -uv);
    float cos_theta = rmin(dot_p, 1.0f);
    V3 perp = etai_over_etat * (uv + cos_theta * n);
    V3 par = (-std::sqrt(std::fabs(1.0f - length_squared(perp)))) * n;
    return perp + par;
}

// src/ray.rs:5-16
struct Ray {
    V3 origin, direction;
    float time;
    V3 at(float t) const { return origin + (t * direction); }
};

// src/aabb.rs:8-44
struct AABB {
    V3 min{INF, INF, INF}, max{-INF, -INF, -INF};
    bool hit(const Ray& ray, float /*t_min*/, float /*t_max*/) const {
        ORC_COUNT(g_cnt_box);
        V3 recip{1.0f / ray.direction.x, 1.0f / ray.direction.y, 1.0f / ray.direction.z};
        V3 mn = (min - ray.origin) * recip;
        V3 mx = (max - ray.origin) * recip;
        float t_min = rmax(rmax(rmin(mn.x, mx.x), rmin(mn.y, mx.y)), rmin(mn.z, mx.z));
        float t_max = rmin(rmin(rmax(mn.x, mx.x), rmax(mn.y, mx.y)), rmax(mn.z, mx.z));
        if (t_max < 0.0f || t_min >= t_max) return false;
        return true;
    }
    static AABB surrounding_box(const AABB& a, const AABB& b) {
        AABB r;
        r.min = vmin(a.min, b.min);
        r.max = vmax(a.max, b.max);
        return r;
    }
};

// ----------------------------------------------------------------------------
// Textures  (src/texture.rs, src/perlin.rs)
// ----------------------------------------------------------------------------
struct Texture {
    virtual ~Texture() {}
    virtual V3 value(float u, float v, V3 p) const = 0;
};
struct SolidColor : Texture {
    V3 color;
    explicit SolidColor(V3 c) : color(c) {}
    V3 value(float, float, V3) const override { return color; }
};
struct CheckeredTexture : Texture {  // src/texture.rs:52-62
    std::shared_ptr<Texture> odd, even;
    CheckeredTexture(V3 o, V3 e) : odd(std::make_shared<SolidColor>(o)), even(std::make_shared<SolidColor>(e)) {}
    V3 value(float u, float v, V3 p) const override {
        float sines = std::sin(10.0f * p.x) * std::sin(10.0f * p.y) * std::sin(10.0f * p.z);
        if (sines < 0.0f) return odd->value(u, v, p);
        return even->value(u, v, p);
    }
};
struct Perlin {  // src/perlin.rs:12-103
    V3 rand_vec[256];
    int32_t perm_x[256], perm_y[256], perm_z[256];
    static void generate_perm(Stream& r, int32_t* p) {
        for (int i = 0; i < 256; i++) p[i] = i;
        for (int i = 255; i >= 1; i--) {  // rand 0.8 SliceRandom::shuffle
            uint32_t j = r.gen_index((uint32_t)i + 1);
            std::swap(p[i], p[j]);
        }
    }
    void init(Stream& r) {
        for (int i = 0; i < 256; i++) rand_vec[i] = random_range(r, -1.0f, 1.0f);
        generate_perm(r, perm_x);
        generate_perm(r, perm_y);
        generate_perm(r, perm_z);
    }
    float noise(V3 p) const {
        float fx = std::floor(p.x), fy = std::floor(p.y), fz = std::floor(p.z);
        float u = p.x - fx, v = p.y - fy, w = p.z - fz;
        int32_t i = (int32_t)fx, j = (int32_t)fy, k = (int32_t)fz;
        V3 c[2][2][2];
        for (int di = 0; di < 2; di++)
            for (int dj = 0; dj < 2; dj++)
                for (int dk = 0; dk < 2; dk++)
                    c[di][dj][dk] = rand_vec[perm_x[(uint32_t)(i + di) & 255] ^ perm_y[(uint32_t)(j + dj) & 255] ^
                                             perm_z[(uint32_t)(k + dk) & 255]];
        float uu = u * u * (3.0f - 2.0f * u);
        float vv = v * v * (3.0f - 2.0f * v);
        float ww = w * w * (3.0f - 2.0f * w);
        float acc = 0.0f;
        for (int di = 0; di < 2; di++)
            for (int dj = 0; dj < 2; dj++)
                for (int dk = 0; dk < 2; dk++) {
                    V3 weight{u - (float)di, v - (float)dj, w - (float)dk};
                    float fi = (float)di, fj = (float)dj, fk = (float)dk;
                    acc += (fi * uu + (1.0f - fi) * (1.0f - uu)) * (fj * vv + (1.0f - fj) * (1.0f - vv)) *
                           (fk * ww + (1.0f - fk) * (1.0f - ww)) * dot(weight, c[di][dj][dk]);
                }
        return acc;
    }
    float turb(V3 p, int depth) const {
        float acc = 0.0f;
        V3 temp = p;
        float weight = 1.0f;
        for (int i = 0; i < depth; i++) {
            acc += weight * noise(temp);
            weight *= 0.5f;
            temp = 2.0f * temp;
        }
        return std::fabs(acc);
    }
};
struct NoiseTexture : Texture {  // src/texture.rs:64-82
    Perlin noise;
    float frequency;
    NoiseTexture(Stream& r, float f) : frequency(f) { noise.init(r); }
    V3 value(float, float, V3 p) const override {
        return v3(0.5f, 0.5f, 0.5f) * (1.0f + std::sin(frequency * p.z + 10.0f * noise.turb(p, 7)));
    }
};
inline V3 image_value(const uint8_t* rgb, uint32_t width, uint32_t height, float cu, float cv) {  // src/texture.rs:96-117
    float u = rclamp(cu, 0.0f, 1.0f);
    float v = 1.0f - rclamp(cv, 0.0f, 1.0f);
    auto to_u32 = [](float f) -> uint32_t {  // Rust `as u32`: saturating, NaN -> 0
        if (!(f == f)) return 0;
        if (f <= 0.0f) return 0;
        if (f >= 4294967296.0f) return 0xFFFFFFFFu;
        return (uint32_t)f;
    };
    uint32_t i = to_u32(u * (float)width);
    uint32_t j = to_u32(v * (float)height);
    if (i >= width) i = width - 1;
    if (j >= height) j = height - 1;
    const uint8_t* px = rgb + 3 * ((size_t)j * width + i);
    return {(float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f};
}
struct ImageTexture : Texture {
    std::vector<uint8_t> rgb;
    uint32_t width, height;
    ImageTexture(const uint8_t* data, int w, int h) : rgb(data, data + (size_t)w * h * 3), width(w), height(h) {}
    V3 value(float u, float v, V3) const override { return image_value(rgb.data(), width, height, u, v); }
};

// ----------------------------------------------------------------------------
// HitRecord (src/hit_record.rs:11-42) and Materials (src/materials.rs)
// ----------------------------------------------------------------------------
struct Material;
struct HitRecord {
    V3 point{0, 0, 0}, normal{0, 0, 0};
    float t = INF;
    float u = 0, v = 0;
    std::shared_ptr<const Material> material;  // Arc<dyn Material>
    bool front_face = false;
    int prim_id = -1;  // oracle bookkeeping: which primitive / medium produced the record
    int inst_id = -1;
    void set_face_normal(const Ray& ray, V3 outward) {
        front_face = dot(ray.direction, outward) < 0.0f;
        normal = front_face ? outward : -outward;
    }
};
inline uint32_t tie_key(const HitRecord& r) {
    return r.prim_id >= 0 ? (uint32_t)r.prim_id : 0x40000000u + (uint32_t)(-(r.prim_id) - 2);
}

struct Material {
    virtual ~Material() {}
    virtual bool scatter(const Ray& ray, const HitRecord& rec, V3& att, Ray& scattered) const = 0;
    virtual V3 emitted(float, float, V3) const { return {0, 0, 0}; }
};
struct LambertianMat : Material {  // src/materials.rs:45-61
    std::shared_ptr<Texture> albedo;
    explicit LambertianMat(std::shared_ptr<Texture> t) : albedo(std::move(t)) {}
    explicit LambertianMat(V3 c) : albedo(std::make_shared<SolidColor>(c)) {}
    bool scatter(const Ray& ray, const HitRecord& rec, V3& att, Ray& scattered) const override {
        V3 sc = rec.normal + random_unit_vector(g_path->rng);
        if (is_near_zero(sc)) sc = rec.normal;
        scattered = Ray{rec.point, sc, ray.time};
        att = albedo->value(rec.u, rec.v, rec.point);
        return true;
    }
};
struct MetalMat : Material {  // src/materials.rs:64-95
    V3 albedo;
    float fuzz;
    MetalMat(V3 a, float f) : albedo(a), fuzz(f < 1.0f ? f : 1.0f) {}
    bool scatter(const Ray& ray, const HitRecord& rec, V3& att, Ray& scattered) const override {
        V3 reflected = reflect(normalize(ray.direction), rec.normal);
        scattered = Ray{rec.point, reflected + fuzz * random_in_unit_sphere(g_path->rng), ray.time};
        att = albedo;
        return dot(scattered.direction, rec.normal) > 0.0f;
    }
};
inline float reflectance(float cosine, float ref_index) {  // src/materials.rs:109-114
    float r0 = (1.0f - ref_index) / (1.0f + ref_index);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * std::pow(1.0f - cosine, 5.0f);
}
struct DielectricMat : Material {  // src/materials.rs:98-148
    float index_refraction;
    explicit DielectricMat(float i) : index_refraction(i) {}
    bool scatter(const Ray& ray, const HitRecord& rec, V3& att, Ray& scattered) const override {
        att = {1, 1, 1};
        float ratio = rec.front_face ? 1.0f / index_refraction : index_refraction;
        V3 unit = normalize(ray.direction);
        float cos_theta = rmin(dot(rec.normal, -unit), 1.0f);
        float sin_theta = std::sqrt(1.0f - cos_theta * cos_theta);
        bool cannot_refract = ratio * sin_theta > 1.0f;
        V3 dir;
        if (cannot_refract || reflectance(cos_theta, ratio) > g_path->rng.gen_f32())
            dir = reflect(unit, rec.normal);
        else
            dir = refract(unit, rec.normal, ratio);
        scattered = Ray{rec.point, dir, ray.time};
        return true;
    }
};
struct DiffuseLight : Material {  // src/materials.rs:161-184
    std::shared_ptr<Texture> emit;
    explicit DiffuseLight(V3 c) : emit(std::make_shared<SolidColor>(c)) {}
    bool scatter(const Ray&, const HitRecord&, V3&, Ray&) const override { return false; }
    V3 emitted(float u, float v, V3 p) const override { return emit->value(u, v, p); }
};
struct IsotropicMat : Material {  // src/materials.rs:186-215
    std::shared_ptr<Texture> albedo;
    explicit IsotropicMat(V3 c) : albedo(std::make_shared<SolidColor>(c)) {}
    bool scatter(const Ray& ray, const HitRecord& rec, V3& att, Ray& scattered) const override {
        scattered = Ray{rec.point, normalize(random_in_unit_sphere(g_path->rng)), ray.time};
        // Rust evaluates struct fields in source order: direction first, then origin, time
        att = albedo->value(rec.u, rec.v, rec.point);
        return true;
    }
};

// ----------------------------------------------------------------------------
// Hittables (src/shapes.rs, src/scene.rs, src/bvh_node.rs)
// ----------------------------------------------------------------------------
struct Hittable;
struct Ids {
    int prim = 0, inst = 0, medium = 0;
    std::vector<const Hittable*> media;  // every ConstantMedium, in creation order
    // for each of them the TransformedObjects it stands inside, outermost first (empty for a medium in the world)
    std::vector<std::vector<const Hittable*>> media_inside;
    std::vector<const Hittable*> inside_now;  // (while the ids are handed out)
};
struct Hittable {
    virtual ~Hittable() {}
    virtual bool intersect(const Ray& ray, float t_min, float t_max, HitRecord& out) const = 0;
    virtual bool bounding_box(float t0, float t1, AABB& out) const = 0;
    virtual void assign_ids(Ids& ids) = 0;
};
using HPtr = std::shared_ptr<Hittable>;
using MPtr = std::shared_ptr<const Material>;

inline void get_sphere_uv(V3 p, float& u, float& v) {  // src/shapes.rs:19-24
    float theta = std::acos(-p.y);
    float phi = std::atan2(-p.z, p.x) + PI_F;
    u = phi / (2.0f * PI_F);
    v = theta / PI_F;
}

struct Sphere : Hittable {  // src/shapes.rs:99-165
    V3 center;
    float radius;
    MPtr material;
    int id = -1;
    Sphere(V3 c, float r, MPtr m) : center(c), radius(r), material(std::move(m)) {}
    bool intersect(const Ray& ray, float t_min, float t_max, HitRecord& rec) const override {
        ORC_COUNT(g_cnt_prim);
        V3 oc = ray.origin - center;
        float a = length_squared(ray.direction);
        float half_b = dot(oc, ray.direction);
        float c = length_squared(oc) - radius * radius;
        float disc = half_b * half_b - a * c;
        if (disc < 0.0f) return false;
        float sqrtd = std::sqrt(disc);
        float root = (-half_b - sqrtd) / a;
        if (root < t_min || root > t_max) {
            root = (-half_b + sqrtd) / a;
            if (root < t_min || root > t_max) return false;
        }
        V3 point = ray.at(root);
        V3 outward = (point - center) / radius;
        rec.t = root;
        rec.point = point;
        rec.normal = outward;
        rec.material = material;
        get_sphere_uv(outward, rec.u, rec.v);
        rec.front_face = true;
        rec.prim_id = id;
        rec.inst_id = -1;
        rec.set_face_normal(ray, outward);
        return true;
    }
    bool bounding_box(float, float, AABB& out) const override {
        V3 off{radius, radius, radius};
        out.min = center - off;
        out.max = center + off;
        return true;
    }
    void assign_ids(Ids& ids) override { id = ids.prim++; }
};

struct MovingSphere : Hittable {  // src/shapes.rs:167-250
    V3 c0, c1;
    float radius, t_start, t_end;
    MPtr material;
    int id = -1;
    MovingSphere(V3 a, V3 b, float r, float ts, float te, MPtr m)
        : c0(a), c1(b), radius(r), t_start(ts), t_end(te), material(std::move(m)) {}
    V3 center(float time) const { return c0 + (((time - t_start) / (t_end - t_start)) * (c1 - c0)); }
    bool intersect(const Ray& ray, float t_min, float t_max, HitRecord& rec) const override {
        ORC_COUNT(g_cnt_prim);
        V3 oc = ray.origin - center(ray.time);
        float a = length_squared(ray.direction);
        float half_b = dot(oc, ray.direction);
        float c = length_squared(oc) - radius * radius;
        float disc = half_b * half_b - a * c;
        if (disc < 0.0f) return false;
        float sqrtd = std::sqrt(disc);
        float root = (-half_b - sqrtd) / a;
        if (root < t_min || root > t_max) {
            root = (-half_b + sqrtd) / a;
            if (root < t_min || root > t_max) return false;
        }
        V3 point = ray.at(root);
        V3 outward = (point - center(ray.time)) / radius;
        rec.t = root;
        rec.point = ray.at(root);
        rec.normal = outward;
        rec.material = material;
        get_sphere_uv(outward, rec.u, rec.v);
        rec.front_face = true;
        rec.prim_id = id;
        rec.inst_id = -1;
        rec.set_face_normal(ray, outward);
        return true;
    }
    bool bounding_box(float t0, float t1, AABB& out) const override {
        V3 off{radius, radius, radius};
        AABB a, b;
        a.min = center(t0) - off;
        a.max = center(t0) + off;
        b.min = center(t1) - off;
        b.max = center(t1) + off;
        out = AABB::surrounding_box(a, b);
        return true;
    }
    void assign_ids(Ids& ids) override { id = ids.prim++; }
};

// XYRect / XZRect / YZRect (src/shapes.rs:252-442).  AXIS = the constant axis:
// 2 -> XYRect (in-plane x,y), 1 -> XZRect (x,z), 0 -> YZRect (y,z).
template <int AXIS>
struct Rect : Hittable {
    float min0, min1, max0, max1, offset;
    MPtr material;
    int id = -1;
    Rect(float mn0, float mn1, float mx0, float mx1, float off, MPtr m)
        : min0(mn0), min1(mn1), max0(mx0), max1(mx1), offset(off), material(std::move(m)) {}
    bool intersect(const Ray& ray, float t_min, float t_max, HitRecord& rec) const override {
        ORC_COUNT(g_cnt_prim);
        constexpr int A0 = (AXIS == 0) ? 1 : 0;
        constexpr int A1 = (AXIS == 2) ? 1 : 2;
        float t = (offset - axis_of(ray.origin, AXIS)) / axis_of(ray.direction, AXIS);
        if (t < t_min || t > t_max) return false;
        float x = axis_of(ray.origin, A0) + t * axis_of(ray.direction, A0);
        float y = axis_of(ray.origin, A1) + t * axis_of(ray.direction, A1);
        if (x < min0 || x > max0 || y < min1 || y > max1) return false;
        rec.t = t;
        rec.point = ray.at(t);
        rec.u = (x - min0) / (max0 - min0);
        rec.v = (y - min1) / (max1 - min1);
        V3 n = AXIS == 2 ? v3(0, 0, 1) : (AXIS == 1 ? v3(0, 1, 0) : v3(1, 0, 0));
        rec.normal = n;
        rec.material = material;
        rec.front_face = true;
        rec.prim_id = id;
        rec.inst_id = -1;
        rec.set_face_normal(ray, n);
        return true;
    }
    bool bounding_box(float, float, AABB& out) const override {
        if (AXIS == 2) {
            out.min = {min0, min1, offset - 0.0001f};
            out.max = {max0, max1, offset + 0.0001f};
        } else if (AXIS == 1) {
            out.min = {min0, offset - 0.0001f, min1};
            out.max = {max0, offset + 0.0001f, max1};
        } else {
            out.min = {offset - 0.0001f, min0, min1};
            out.max = {offset + 0.0001f, max0, max1};
        }
        return true;
    }
    void assign_ids(Ids& ids) override { id = ids.prim++; }
};
using XYRect = Rect<2>;
using XZRect = Rect<1>;
using YZRect = Rect<0>;

struct Scene : Hittable {  // src/scene.rs:8-69
    std::vector<HPtr> shapes;
    void add_shape(HPtr s) { shapes.push_back(std::move(s)); }
    bool intersect(const Ray& ray, float t_min, float t_max, HitRecord& out) const override {
        float closest = t_max;
        bool any = false;
        for (const auto& s : shapes) {
            HitRecord rec;
            if (s->intersect(ray, t_min, closest, rec)) {
                closest = rec.t;
                out = rec;
                any = true;
            }
        }
        return any;
    }
    bool bounding_box(float t0, float t1, AABB& out) const override {
        if (shapes.empty()) return false;
        bool first = true;
        AABB result;
        for (const auto& s : shapes) {
            AABB temp;
            if (!s->bounding_box(t0, t1, temp)) return false;
            if (first) {
                result = temp;
                first = false;
            } else {
                result = AABB::surrounding_box(temp, result);
            }
        }
        out = result;
        return true;
    }
    void assign_ids(Ids& ids) override {
        for (auto& s : shapes) s->assign_ids(ids);
    }
};

struct Box2 : Hittable {  // src/shapes.rs:444-488
    V3 min, max;
    Scene sides;
    Box2(V3 mn, V3 mx, MPtr m) : min(mn), max(mx) {
        sides.add_shape(std::make_shared<XYRect>(mn.x, mn.y, mx.x, mx.y, mn.z, m));
        sides.add_shape(std::make_shared<XYRect>(mn.x, mn.y, mx.x, mx.y, mx.z, m));
        sides.add_shape(std::make_shared<XZRect>(mn.x, mn.z, mx.x, mx.z, mn.y, m));
        sides.add_shape(std::make_shared<XZRect>(mn.x, mn.z, mx.x, mx.z, mx.y, m));
        sides.add_shape(std::make_shared<YZRect>(mn.y, mn.z, mx.y, mx.z, mn.x, m));
        sides.add_shape(std::make_shared<YZRect>(mn.y, mn.z, mx.y, mx.z, mx.x, m));
    }
    bool intersect(const Ray& ray, float t_min, float t_max, HitRecord& out) const override {
        return sides.intersect(ray, t_min, t_max, out);
    }
    bool bounding_box(float, float, AABB& out) const override {
        out.min = min;
        out.max = max;
        return true;
    }
    void assign_ids(Ids& ids) override { sides.assign_ids(ids); }
};

struct TransformedObject : Hittable {  // src/shapes.rs:26-97
    HPtr object;
    M4 transform, inv_transform;
    AABB aabb;
    int id = -1;
    TransformedObject(HPtr obj, const M4& m) : object(std::move(obj)), transform(m), inv_transform(m4_inverse(m)) {
        AABB base;
        object->bounding_box(0.0f, 1.0f, base);
        V3 mn = base.min, mx = base.max;
        V3 p0 = m4_point(m, mn);
        V3 p1 = m4_point(m, v3(mx.x, mn.y, mn.z));
        V3 p2 = m4_point(m, v3(mx.x, mn.y, mx.z));
        V3 p3 = m4_point(m, v3(mn.x, mn.y, mx.z));
        V3 p4 = m4_point(m, v3(mn.x, mx.y, mn.z));
        V3 p5 = m4_point(m, v3(mx.x, mx.y, mn.z));
        V3 p6 = m4_point(m, mx);
        V3 p7 = m4_point(m, v3(mn.x, mx.y, mx.z));
        aabb.min = vmin(p0, vmin(p1, vmin(p2, vmin(p3, vmin(p4, vmin(p5, vmin(p6, p7)))))));
        aabb.max = vmax(p0, vmax(p1, vmax(p2, vmax(p3, vmax(p4, vmax(p5, vmax(p6, p7)))))));
    }
    Ray to_local(const Ray& ray) const {
        return Ray{m4_point(inv_transform, ray.origin), m4_vector(inv_transform, ray.direction), ray.time};
    }
    // src/shapes.rs:74-88: the object's record taken out into the space of `ray`
    void to_outside(const Ray& ray, const HitRecord& rec, HitRecord& out) const {
        V3 normal = normalize(m4_vector(transform, rec.normal));
        V3 point = m4_point(transform, rec.point);
        out = rec;
        out.point = point;
        out.normal = normal;
        out.inst_id = id;
        out.set_face_normal(ray, normal);
    }
    bool intersect(const Ray& ray, float t_min, float t_max, HitRecord& out) const override {
        Ray local = to_local(ray);
        HitRecord rec;
        if (!object->intersect(local, t_min, t_max, rec)) return false;
        to_outside(ray, rec, out);
        return true;
    }
    bool bounding_box(float, float, AABB& out) const override {
        out = aabb;
        return true;
    }
    void assign_ids(Ids& ids) override {
        id = ids.inst++;
        ids.inside_now.push_back(this);
        object->assign_ids(ids);
        ids.inside_now.pop_back();
    }
};

struct ConstantMedium : Hittable {  // src/shapes.rs:624-705
    HPtr boundary;
    double negative_density;  // -1/density, f64
    MPtr material;
    int id = -1;
    ConstantMedium(HPtr b, float density, V3 color)
        : boundary(std::move(b)), negative_density(-1.0 / (double)density), material(std::make_shared<IsotropicMat>(color)) {}
    bool intersect(const Ray& ray, float t_min, float t_max, HitRecord& out) const override {
        if (g_media_deferred) return false;
        HitRecord r1, r2;
        if (!boundary->intersect(ray, -INF, INF, r1)) return false;
        if (!boundary->intersect(ray, r1.t + 0.0001f, INF, r2)) return false;
        if (r1.t < t_min) r1.t = t_min;
        if (r2.t > t_max) r2.t = t_max;
        if (r1.t >= r2.t) return false;
        if (r1.t < 0.0f) r1.t = 0.0f;
        double len = (double)length(ray.direction);
        double distance_inside = (double)(r2.t - r1.t) / len;
        double xi = g_path->fixed_xi ? g_path->xi : medium_xi(g_path->rng, g_path->bounce, (uint32_t)id);
        double hit_distance = negative_density * std::log(xi);
        if (hit_distance > distance_inside) return false;
        float final_t = r1.t + (float)(hit_distance / len);
        out = HitRecord();
        out.t = final_t;
        out.point = ray.at(final_t);
        out.normal = {1, 0, 0};
        out.front_face = true;
        out.material = material;
        out.u = 0;
        out.v = 0;
        out.prim_id = -(id + 2);
        out.inst_id = -1;
        return true;
    }
    bool bounding_box(float t0, float t1, AABB& out) const override { return boundary->bounding_box(t0, t1, out); }
    void assign_ids(Ids& ids) override {
        id = ids.medium++;
        ids.media.push_back(this);
        ids.media_inside.push_back(ids.inside_now);
        // (the boundary's own objects are not "inside" anything as far as the media are concerned)
        std::vector<const Hittable*> keep;
        keep.swap(ids.inside_now);
        boundary->assign_ids(ids);
        keep.swap(ids.inside_now);
    }
};

struct EmptyHittable : Hittable {  // src/bvh_node.rs:13-24
    bool intersect(const Ray&, float, float, HitRecord&) const override { return false; }
    bool bounding_box(float, float, AABB&) const override { return false; }
    void assign_ids(Ids&) override {}
};

struct BVHNode : Hittable {  // src/bvh_node.rs:26-169
    HPtr left, right;
    AABB box;
    std::vector<HPtr> insertion_order;  // only kept at a tree's root, for id assignment
    static bool box_less(const HPtr& a, const HPtr& b, int axis) {
        AABB ba, bb;
        a->bounding_box(0.0f, 0.0f, ba);
        b->bounding_box(0.0f, 0.0f, bb);
        return axis_of(ba.min, axis) < axis_of(bb.min, axis);
    }
    static std::shared_ptr<BVHNode> from_vector(std::vector<HPtr> objects, float t0, float t1, Stream& rng) {
        auto root = std::make_shared<BVHNode>();
        root->left = std::make_shared<EmptyHittable>();
        root->right = std::make_shared<EmptyHittable>();
        int axis = (int)rng.gen_index(3);
        size_t span = objects.size();
        if (span == 1) {
            root->left = objects[0];
        } else if (span == 2) {
            AABB ba, bb;
            objects[0]->bounding_box(0.0f, 0.0f, ba);
            objects[1]->bounding_box(0.0f, 0.0f, bb);
            if ((axis_of(ba.min, axis) - axis_of(bb.min, axis)) < F32_EPSILON) {
                root->left = objects[0];
                root->right = objects[1];
            } else {
                root->right = objects[0];
                root->left = objects[1];
            }
        } else if (span > 2) {
            std::stable_sort(objects.begin(), objects.end(),
                             [axis](const HPtr& a, const HPtr& b) { return box_less(a, b, axis); });
            size_t mid = span / 2;
            std::vector<HPtr> l(objects.begin(), objects.begin() + mid), r(objects.begin() + mid, objects.end());
            if (l.size() == 1)
                root->left = l[0];
            else
                root->left = from_vector(l, t0, t1, rng);
            if (r.size() == 1)
                root->right = r[0];
            else
                root->right = from_vector(r, t0, t1, rng);
        }
        AABB la, ra;
        AABB tmp;
        if (root->left->bounding_box(t0, t1, tmp)) la = tmp;
        if (root->right->bounding_box(t0, t1, tmp)) ra = tmp;
        root->box = AABB::surrounding_box(la, ra);
        return root;
    }
    static std::shared_ptr<BVHNode> from_scene(const Scene& s, float t0, float t1, Stream& rng) {
        auto n = from_vector(s.shapes, t0, t1, rng);
        n->insertion_order = s.shapes;
        return n;
    }
    bool intersect(const Ray& ray, float t_min, float t_max, HitRecord& out) const override {
        if (!box.hit(ray, t_min, t_max)) return false;
        float t_max0 = t_max;
        HitRecord lrec, rrec;
        bool lhit = left->intersect(ray, t_min, t_max0, lrec);
        if (lhit) t_max0 = lrec.t;
        bool rhit = right->intersect(ray, t_min, t_max0, rrec);
        if (rhit) {
            // exact-tie rule (see file header): later-created primitive wins
            if (lhit && rrec.t == lrec.t && tie_key(lrec) > tie_key(rrec)) {
                out = lrec;
                return true;
            }
            out = rrec;
            return true;
        }
        if (lhit) {
            out = lrec;
            return true;
        }
        return false;
    }
    bool bounding_box(float, float, AABB& out) const override {
        out = box;
        return true;
    }
    void assign_ids(Ids& ids) override {
        for (auto& s : insertion_order) s->assign_ids(ids);
    }
};

// ----------------------------------------------------------------------------
// Camera (src/camera.rs:14-65)
// ----------------------------------------------------------------------------
struct Camera {
    V3 origin, lower_left_corner, horizontal, vertical, u, v, w;
    float lens_radius, time_start, time_end;
    static Camera make(V3 origin, V3 target, V3 up, float vfov, float aspect, float aperture, float focus, float t0,
                       float t1) {
        float h = std::tan(vfov / 2.0f);
        float viewport_height = 2.0f * h;
        float viewport_width = viewport_height * aspect;
        V3 w = normalize(origin - target);
        V3 u = normalize(cross(up, w));
        V3 v = cross(w, u);
        V3 horizontal = (focus * viewport_width) * u;
        V3 vertical = (focus * viewport_height) * v;
        Camera c;
        c.origin = origin;
        c.lower_left_corner = origin - horizontal / 2.0f - vertical / 2.0f - focus * w;
        c.horizontal = horizontal;
        c.vertical = vertical;
        c.u = u;
        c.v = v;
        c.w = w;
        c.lens_radius = aperture / 2.0f;
        c.time_start = t0;
        c.time_end = t1;
        return c;
    }
    Ray get_ray(float s, float t, Stream& rng) const {
        Stream lens{rng.k0, rng.k1, CTR_LENS};
        V3 rd = lens_radius * random_in_unit_disk(lens);
        V3 offset = u * rd.x + v * rd.y;
        Ray r;
        r.origin = origin + offset;
        r.direction = lower_left_corner + s * horizontal + t * vertical - origin - offset;
        r.time = rng.gen_range(time_start, time_end);
        return r;
    }
};
inline Ray camera_sample_ray(const Camera& cam, int w, int h, int i, int j, Stream& rng) {  // src/main.rs:471-474
    rng.n = CTR_CAMERA;
    float u = ((float)i + rng.gen_f32()) / (float)(w - 1);
    float v = ((float)(h - 1 - j) + rng.gen_f32()) / (float)(h - 1);
    return cam.get_ray(u, v, rng);
}

// world.intersect with media evaluated last (see the file header)
// A medium inside TransformedObjects, asked through them as the nested intersect calls of src/shapes.rs:64-92 would:
// each level takes the ray into its object's space and the record back out.
bool medium_through(const std::vector<const Hittable*>& inside, size_t level, const Hittable* m, const Ray& ray, float t_min, float t_max,
                    HitRecord& out) {
    if (level == inside.size()) return m->intersect(ray, t_min, t_max, out);
    const TransformedObject* T = static_cast<const TransformedObject*>(inside[level]);
    HitRecord rec;
    if (!medium_through(inside, level + 1, m, T->to_local(ray), t_min, t_max, rec)) return false;
    T->to_outside(ray, rec, out);
    return true;
}
struct World {
    const Hittable* surfaces;                 // the Scene or its BVHNode
    const std::vector<const Hittable*>* media;  // every ConstantMedium, creation order
    const std::vector<std::vector<const Hittable*>>* media_inside;
    bool intersect(const Ray& ray, float t_min, float t_max, HitRecord& out) const {
        g_media_deferred = true;
        bool any = surfaces->intersect(ray, t_min, t_max, out);
        g_media_deferred = false;
        float closest = any ? out.t : t_max;
        for (size_t k = 0; k < media->size(); k++) {
            HitRecord rec;
            if (medium_through((*media_inside)[k], 0, (*media)[k], ray, t_min, closest, rec)) {  // scene.rs:33-37: taken unchecked
                closest = rec.t;
                out = rec;
                any = true;
            }
        }
        return any;
    }
};

// src/main.rs:44-63
V3 ray_color(const Ray& ray, V3 background, const World& world, int depth, int max_depth) {
    if (depth <= 0) return {0, 0, 0};
    g_path->bounce = (uint32_t)(max_depth - depth);
    g_path->rng.n = g_path->bounce << 24;
    g_cnt_rays++;
    HitRecord rec;
    if (world.intersect(ray, 0.005f, INF, rec)) {
        Ray scattered{{0, 0, 0}, {0, 0, 0}, ray.time};
        V3 att{1, 1, 1};
        V3 emitted = rec.material->emitted(rec.u, rec.v, rec.point);
        if (rec.material->scatter(ray, rec, att, scattered))
            return emitted + att * ray_color(scattered, background, world, depth - 1, max_depth);
        return emitted;
    }
    return background;
}
inline uint8_t float_to_u8_color(float f) {  // src/main.rs:307-309
    float x = 256.0f * rclamp(f, 0.0f, 0.999f);
    if (!(x == x)) return 0;
    if (x <= 0.0f) return 0;
    if (x >= 255.0f) return 255;
    return (uint8_t)x;
}

// ----------------------------------------------------------------------------
// Scene builders (src/main.rs:87-305) and presets (src/main.rs:386-451)
// ----------------------------------------------------------------------------
inline float degree_to_rad(float deg) { return deg * PI_F / 180.0f; }

}  // namespace

struct orc_scene {
    Scene world;                     // the linear list
    std::shared_ptr<BVHNode> bvh;    // BVHNode::from_scene(&world, 0, 1)  (src/main.rs:458)
    Ids ids;
    V3 cam_origin, cam_target, background;
    float vfov_deg, aperture;
};

namespace {

template <class T, class... A>
std::shared_ptr<T> mk(A&&... a) {
    return std::make_shared<T>(std::forward<A>(a)...);
}

void build_random_world(orc_scene& S, Stream& rng) {  // src/main.rs:87-125
    Scene& s = S.world;
    auto ground = mk<LambertianMat>(std::static_pointer_cast<Texture>(
        mk<CheckeredTexture>(v3(0.2f, 0.3f, 0.1f), v3(0.9f, 0.9f, 0.9f))));
    s.add_shape(mk<Sphere>(v3(0.0f, -1000.0f, 0.0f), 1000.0f, ground));
    V3 point{4.0f, 0.2f, 0.0f};
    for (int a = -11; a < 11; a++)
        for (int b = -11; b < 11; b++) {
            float choose_mat = rng.gen_f32();
            float cx = (float)a + 0.9f * rng.gen_f32();
            float cz = (float)b * 0.9f * rng.gen_f32();
            V3 center{cx, 0.2f, cz};
            if (length(center - point) > 0.9f) {
                if (choose_mat < 0.8f) {
                    V3 center2 = center + v3(0.0f, rng.gen_range(0.0f, 0.5f), 0.0f);
                    V3 a1 = random_vec(rng);
                    V3 a2 = random_vec(rng);
                    s.add_shape(mk<MovingSphere>(center, center2, 0.2f, 0.0f, 1.0f, mk<LambertianMat>(a1 * a2)));
                } else if (choose_mat < 0.95f) {
                    V3 alb = random_range(rng, 0.5f, 1.0f);
                    float fuzz = rng.gen_range(0.5f, 1.0f);
                    s.add_shape(mk<Sphere>(center, 0.2f, mk<MetalMat>(alb, fuzz)));
                } else {
                    s.add_shape(mk<Sphere>(center, 0.2f, mk<DielectricMat>(1.5f)));
                }
            }
        }
    s.add_shape(mk<Sphere>(v3(0.0f, 1.0f, 0.0f), 1.0f, mk<DielectricMat>(1.5f)));
    s.add_shape(mk<Sphere>(v3(-4.0f, 1.0f, 0.0f), 1.0f, mk<LambertianMat>(v3(0.4f, 0.2f, 0.1f))));
    s.add_shape(mk<Sphere>(v3(4.0f, 1.0f, 0.0f), 1.0f, mk<MetalMat>(v3(0.7f, 0.6f, 0.5f), 0.0f)));
    S.cam_origin = {13, 2, 3};
    S.cam_target = {0, 0, 0};
    S.vfov_deg = 20;
    S.aperture = 0.1f;
    S.background = {0.70f, 0.80f, 1.00f};
}
void build_two_spheres(orc_scene& S) {  // src/main.rs:127-139
    auto checker = std::static_pointer_cast<Texture>(mk<CheckeredTexture>(v3(0.2f, 0.3f, 0.1f), v3(0.9f, 0.9f, 0.9f)));
    S.world.add_shape(mk<Sphere>(v3(0, 10, 0), 10.0f, mk<LambertianMat>(checker)));
    S.world.add_shape(mk<Sphere>(v3(0, -10, 0), 10.0f, mk<LambertianMat>(checker)));
    S.cam_origin = {13, 2, 3};
    S.cam_target = {0, 0, 0};
    S.vfov_deg = 20;
    S.aperture = 0;
    S.background = {0.7f, 0.8f, 1.0f};
}
void build_two_perlin_spheres(orc_scene& S, Stream& rng) {  // src/main.rs:141-150
    auto noise = std::static_pointer_cast<Texture>(mk<NoiseTexture>(rng, 4.0f));
    S.world.add_shape(mk<Sphere>(v3(0, -1000, 0), 1000.0f, mk<LambertianMat>(noise)));
    S.world.add_shape(mk<Sphere>(v3(0, 2, 0), 2.0f, mk<LambertianMat>(noise)));
    S.cam_origin = {13, 2, 3};
    S.cam_target = {0, 0, 0};
    S.vfov_deg = 20;
    S.aperture = 0;
    S.background = {0.7f, 0.8f, 1.0f};
}
void build_earth(orc_scene& S, const uint8_t* img, int iw, int ih) {  // src/main.rs:152-161
    auto tex = std::static_pointer_cast<Texture>(mk<ImageTexture>(img, iw, ih));
    S.world.add_shape(mk<Sphere>(v3(0, 0, 0), 2.0f, mk<LambertianMat>(tex)));
    S.cam_origin = {13, 2, 3};
    S.cam_target = {0, 0, 0};
    S.vfov_deg = 20;
    S.aperture = 0;
    S.background = {0.7f, 0.8f, 1.0f};
}
void build_simple_light(orc_scene& S, Stream& rng) {  // src/main.rs:163-175
    auto noise = std::static_pointer_cast<Texture>(mk<NoiseTexture>(rng, 4.0f));
    S.world.add_shape(mk<Sphere>(v3(0, -1000, 0), 1000.0f, mk<LambertianMat>(noise)));
    S.world.add_shape(mk<Sphere>(v3(0, 2, 0), 2.0f, mk<LambertianMat>(noise)));
    S.world.add_shape(mk<XYRect>(3.0f, 1.0f, 5.0f, 3.0f, -2.0f, mk<DiffuseLight>(v3(4, 4, 4))));
    S.world.add_shape(mk<Sphere>(v3(0, 7, 0), 2.0f, mk<DiffuseLight>(v3(4, 4, 4))));
    S.cam_origin = {26, 3, 6};
    S.cam_target = {0, 2, 0};
    S.vfov_deg = 20;
    S.aperture = 0;
    S.background = {0, 0, 0};
}
void cornell_walls(Scene& s, V3 light, float lx0, float lz0, float lx1, float lz1) {
    V3 white{0.73f, 0.73f, 0.73f}, green{0.12f, 0.45f, 0.15f}, red{0.65f, 0.05f, 0.05f};
    s.add_shape(mk<YZRect>(0.0f, 0.0f, 555.0f, 555.0f, 555.0f, mk<LambertianMat>(green)));
    s.add_shape(mk<YZRect>(0.0f, 0.0f, 555.0f, 555.0f, 0.0f, mk<LambertianMat>(red)));
    s.add_shape(mk<XZRect>(lx0, lz0, lx1, lz1, 554.0f, mk<DiffuseLight>(light)));
    s.add_shape(mk<XZRect>(0.0f, 0.0f, 555.0f, 555.0f, 555.0f, mk<LambertianMat>(white)));
    s.add_shape(mk<XZRect>(0.0f, 0.0f, 555.0f, 555.0f, 0.0f, mk<LambertianMat>(white)));
    s.add_shape(mk<XYRect>(0.0f, 0.0f, 555.0f, 555.0f, 555.0f, mk<LambertianMat>(white)));
}
HPtr cornell_block(float w, float h, float d, float rot_deg, V3 trans) {
    V3 white{0.73f, 0.73f, 0.73f};
    auto b = mk<Box2>(v3(0, 0, 0), v3(w, h, d), mk<LambertianMat>(white));
    M4 rotation = m4_rotation_y(degree_to_rad(rot_deg));
    M4 translation = m4_translation(trans);
    M4 final_transform = m4_mul(translation, rotation);
    return mk<TransformedObject>(b, final_transform);
}
void cornell_camera(orc_scene& S) {
    S.cam_origin = {278, 278, -800};
    S.cam_target = {278, 278, 0};
    S.vfov_deg = 40;
    S.aperture = 0;
    S.background = {0, 0, 0};
}
void build_cornell_box(orc_scene& S) {  // src/main.rs:177-205
    cornell_walls(S.world, v3(15, 15, 15), 213.0f, 227.0f, 343.0f, 332.0f);
    S.world.add_shape(cornell_block(165, 165, 165, -18.0f, v3(130, 0, 65)));
    S.world.add_shape(cornell_block(165, 330, 165, 15.0f, v3(265, 0, 305)));
    cornell_camera(S);
}
void build_cornell_smoke(orc_scene& S) {  // src/main.rs:207-238
    cornell_walls(S.world, v3(7, 7, 7), 113.0f, 127.0f, 443.0f, 432.0f);
    S.world.add_shape(mk<ConstantMedium>(cornell_block(165, 165, 165, -18.0f, v3(130, 0, 65)), 0.5f, v3(1, 1, 1)));
    S.world.add_shape(mk<ConstantMedium>(cornell_block(165, 330, 165, 15.0f, v3(265, 0, 305)), 0.5f, v3(0, 0, 0)));
    cornell_camera(S);
}
// Not a reference preset: transformed objects inside transformed objects (TransformedObject<T: Hittable>,
// src/shapes.rs:26-41), up to four deep, in the Cornell shell.  Mirrors nested_instances() of the host library.
void build_nested_instances(orc_scene& S, Stream& axis_rng) {
    cornell_walls(S.world, v3(15, 15, 15), 213.0f, 227.0f, 343.0f, 332.0f);
    V3 white{0.73f, 0.73f, 0.73f};
    {
        HPtr box = mk<Box2>(v3(0, 0, 0), v3(165, 165, 165), mk<LambertianMat>(white));
        HPtr turned = mk<TransformedObject>(box, m4_rotation_y(degree_to_rad(-18.0f)));
        S.world.add_shape(mk<TransformedObject>(turned, m4_translation(v3(130, 0, 65))));
    }
    Scene inner;
    inner.add_shape(mk<Sphere>(v3(60, 200, 60), 40.0f, mk<MetalMat>(v3(0.8f, 0.85f, 0.88f), 0.0f)));
    {
        HPtr box = mk<Box2>(v3(0, 0, 0), v3(80, 80, 80), mk<LambertianMat>(v3(0.65f, 0.05f, 0.05f)));
        HPtr turned = mk<TransformedObject>(box, m4_rotation_y(degree_to_rad(30.0f)));
        inner.add_shape(mk<TransformedObject>(turned, m4_translation(v3(20, 120, 20))));
    }
    {
        HPtr ball = mk<Sphere>(v3(0, 0, 0), 30.0f, mk<DielectricMat>(1.5f));
        inner.add_shape(mk<TransformedObject>(ball, m4_translation(v3(120, 60, 100))));
    }
    inner.add_shape(mk<Box2>(v3(0, 0, 0), v3(165, 100, 165), mk<LambertianMat>(white)));
    inner.add_shape(mk<ConstantMedium>(mk<Sphere>(v3(100, 170, 60), 45.0f, mk<DielectricMat>(1.5f)), 0.05f, v3(0.2f, 0.4f, 0.9f)));
    {
        HPtr fog = mk<ConstantMedium>(mk<Box2>(v3(0, 0, 0), v3(120, 90, 120), mk<DielectricMat>(1.5f)), 0.02f, v3(0.9f, 0.9f, 0.9f));
        M4 place = m4_mul(m4_translation(v3(30, 330, 200)), m4_rotation_y(degree_to_rad(25.0f)));
        S.world.add_shape(mk<TransformedObject>(fog, place));
    }
    HPtr turned = mk<TransformedObject>(BVHNode::from_scene(inner, 0.0f, 1.0f, axis_rng), m4_rotation_y(degree_to_rad(15.0f)));
    S.world.add_shape(mk<TransformedObject>(turned, m4_translation(v3(265, 0, 295))));
    cornell_camera(S);
}
// Not a reference preset: media whose boundary is more than one object (ConstantMedium<T: Hittable>,
// src/shapes.rs:624-646).  Mirrors medium_cluster() of the host library.
void build_medium_cluster(orc_scene& S) {
    cornell_walls(S.world, v3(7, 7, 7), 113.0f, 127.0f, 443.0f, 432.0f);
    Scene balls;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            for (int k = 0; k < 3; k++)
                balls.add_shape(mk<Sphere>(v3(218.0f + 60.0f * (float)i, 150.0f + 60.0f * (float)j, 218.0f + 60.0f * (float)k), 45.0f,
                                           mk<DielectricMat>(1.5f)));
    // (a Scene as the boundary: with a BVHNode the answer would depend on which boxes of its random tree lie behind the
    // ray origin, src/aabb.rs:21-33)
    S.world.add_shape(mk<ConstantMedium>(mk<Scene>(balls), 0.02f, v3(0.9f, 0.9f, 0.9f)));
    S.world.add_shape(mk<ConstantMedium>(mk<Box2>(v3(40, 0, 60), v3(160, 120, 180), mk<DielectricMat>(1.5f)), 0.03f, v3(0.1f, 0.1f, 0.4f)));
    cornell_camera(S);
}
void build_final_scene(orc_scene& S, Stream& rng, Stream& axis_rng, const uint8_t* img, int iw, int ih) {  // src/main.rs:240-305
    Scene& s = S.world;
    Scene boxes;
    auto ground = mk<LambertianMat>(v3(0.48f, 0.83f, 0.53f));
    const int boxes_per_side = 20;
    for (int i = 0; i < boxes_per_side; i++)
        for (int j = 0; j < boxes_per_side; j++) {
            float w = 100.0f;
            float x0 = -1000.0f + (float)i * w;
            float z0 = -1000.0f + (float)j * w;
            float y0 = 0.0f;
            float x1 = x0 + w;
            float y1 = rng.gen_range(1.0f, 101.0f);
            float z1 = z0 + w;
            boxes.add_shape(mk<Box2>(v3(x0, y0, z0), v3(x1, y1, z1), ground));
        }
    s.add_shape(BVHNode::from_scene(boxes, 0.0f, 1.0f, axis_rng));
    s.add_shape(mk<XZRect>(123.0f, 147.0f, 423.0f, 412.0f, 554.0f, mk<DiffuseLight>(v3(7, 7, 7))));
    V3 center1{400, 400, 200};
    V3 center2 = center1 + v3(30, 0, 0);
    s.add_shape(mk<MovingSphere>(center1, center2, 50.0f, 0.0f, 1.0f, mk<LambertianMat>(v3(0.7f, 0.3f, 0.1f))));
    s.add_shape(mk<Sphere>(v3(260, 150, 45), 50.0f, mk<DielectricMat>(1.5f)));
    s.add_shape(mk<Sphere>(v3(0, 150, 145), 50.0f, mk<MetalMat>(v3(0.8f, 0.8f, 0.9f), 1.0f)));
    s.add_shape(mk<Sphere>(v3(360, 150, 145), 70.0f, mk<DielectricMat>(1.5f)));
    s.add_shape(mk<ConstantMedium>(mk<Sphere>(v3(360, 150, 145), 70.0f, mk<DielectricMat>(1.5f)), 0.2f,
                                   v3(0.2f, 0.4f, 0.9f)));
    s.add_shape(mk<ConstantMedium>(mk<Sphere>(v3(0, 0, 0), 5000.0f, mk<DielectricMat>(1.5f)), 0.005f, v3(1, 1, 1)));
    auto emat = mk<LambertianMat>(std::static_pointer_cast<Texture>(mk<ImageTexture>(img, iw, ih)));
    s.add_shape(mk<Sphere>(v3(400, 200, 400), 100.0f, emat));
    auto pertext = std::static_pointer_cast<Texture>(mk<NoiseTexture>(rng, 0.1f));
    s.add_shape(mk<Sphere>(v3(220, 280, 300), 80.0f, mk<LambertianMat>(pertext)));
    Scene boxes2;
    for (int i = 0; i < 1000; i++) {
        auto white = mk<LambertianMat>(v3(0.73f, 0.73f, 0.73f));
        boxes2.add_shape(mk<Sphere>(random_range(rng, 0.0f, 165.0f), 10.0f, white));
    }
    M4 rotation = m4_rotation_y(degree_to_rad(15.0f));
    M4 translation = m4_translation(v3(-100, 270, 395));
    M4 final_transform = m4_mul(translation, rotation);
    s.add_shape(mk<TransformedObject>(BVHNode::from_scene(boxes2, 0.0f, 1.0f, axis_rng), final_transform));
    S.cam_origin = {478, 278, -600};
    S.cam_target = {278, 278, 0};
    S.vfov_deg = 40;
    S.aperture = 0;
    S.background = {0, 0, 0};
}
// Synthetic deep-BVH stress scene (no reference builder): n spheres, centres
// U[-1000,1000) x U[0,50) x U[-1000,1000), radius U[0.05,0.5); 70% Lambertian(xi*xi),
// 20% Metal(U[0.5,1)^3, fuzz U[0,0.5)), 10% Dielectric(1.5); ground sphere r=1e5.
void build_synth_spheres(orc_scene& S, Stream& rng, int64_t n) {
    Scene& s = S.world;
    s.add_shape(mk<Sphere>(v3(0, -100000.0f, 0), 100000.0f, mk<LambertianMat>(v3(0.5f, 0.5f, 0.5f))));
    for (int64_t i = 0; i < n; i++) {
        float cx = rng.gen_range(-1000.0f, 1000.0f);
        float cy = rng.gen_range(0.0f, 50.0f);
        float cz = rng.gen_range(-1000.0f, 1000.0f);
        float r = rng.gen_range(0.05f, 0.5f);
        float choose = rng.gen_f32();
        if (choose < 0.7f) {
            V3 a1 = random_vec(rng);
            V3 a2 = random_vec(rng);
            s.add_shape(mk<Sphere>(v3(cx, cy, cz), r, mk<LambertianMat>(a1 * a2)));
        } else if (choose < 0.9f) {
            V3 alb = random_range(rng, 0.5f, 1.0f);
            float fuzz = rng.gen_range(0.0f, 0.5f);
            s.add_shape(mk<Sphere>(v3(cx, cy, cz), r, mk<MetalMat>(alb, fuzz)));
        } else {
            s.add_shape(mk<Sphere>(v3(cx, cy, cz), r, mk<DielectricMat>(1.5f)));
        }
    }
    S.cam_origin = {0, 120, -1400};
    S.cam_target = {0, 0, 0};
    S.vfov_deg = 40;
    S.aperture = 0;
    S.background = {0.7f, 0.8f, 1.0f};
}

Camera scene_camera(const orc_scene& S, float aspect) {  // src/main.rs:454-456
    return Camera::make(S.cam_origin, S.cam_target, v3(0, 1, 0), degree_to_rad(S.vfov_deg), aspect, S.aperture, 10.0f,
                        0.0f, 1.0f);
}
World world_of(const orc_scene& S, int mode) {
    return World{mode == 1 ? static_cast<const Hittable*>(S.bvh.get()) : static_cast<const Hittable*>(&S.world), &S.ids.media, &S.ids.media_inside};
}
Ray ray_from7(const float* r) { return Ray{{r[0], r[1], r[2]}, {r[3], r[4], r[5]}, r[6]}; }
void rec_to_out(bool hit, const HitRecord& rec, float out[11]) {
    for (int i = 0; i < 11; i++) out[i] = 0;
    out[0] = hit ? 1.0f : 0.0f;
    if (!hit) return;
    out[1] = rec.t;
    out[2] = rec.point.x;
    out[3] = rec.point.y;
    out[4] = rec.point.z;
    out[5] = rec.normal.x;
    out[6] = rec.normal.y;
    out[7] = rec.normal.z;
    out[8] = rec.front_face ? 1.0f : 0.0f;
    out[9] = rec.u;
    out[10] = rec.v;
}
template <class F>
void parallel_for(int64_t n, int threads, F f) {
    if (threads <= 1 || n < 2) {
        f(0, n, 0);
        return;
    }
    std::vector<std::thread> pool;
    std::atomic<int64_t> next{0};
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(4096, n / (threads * 8) + 1));
    for (int t = 0; t < threads; t++)
        pool.emplace_back([&, t]() {
            for (;;) {
                int64_t b = next.fetch_add(chunk);
                if (b >= n) break;
                f(b, std::min(n, b + chunk), t);
            }
        });
    for (auto& th : pool) th.join();
}

}  // namespace

// ============================================================================
// C ABI
// ============================================================================
extern "C" {

orc_scene* orc_scene_create(const char* name, uint32_t seed, const uint8_t* img, int iw, int ih, int64_t n_synth) {
    auto S = std::make_unique<orc_scene>();
    Stream rng = scene_stream(seed);
    Stream axis = bvh_stream(seed);
    std::string n(name);
    bool needs_img = (n == "earth" || n == "final_scene");
    if (needs_img && (!img || iw <= 0 || ih <= 0)) return nullptr;
    if (n == "random_world")
        build_random_world(*S, rng);
    else if (n == "two_spheres")
        build_two_spheres(*S);
    else if (n == "two_perlin_spheres")
        build_two_perlin_spheres(*S, rng);
    else if (n == "earth")
        build_earth(*S, img, iw, ih);
    else if (n == "simple_light")
        build_simple_light(*S, rng);
    else if (n == "cornell_box")
        build_cornell_box(*S);
    else if (n == "cornell_smoke")
        build_cornell_smoke(*S);
    else if (n == "nested_instances")
        build_nested_instances(*S, axis);
    else if (n == "medium_cluster")
        build_medium_cluster(*S);
    else if (n == "final_scene")
        build_final_scene(*S, rng, axis, img, iw, ih);
    else if (n == "synth_spheres")
        build_synth_spheres(*S, rng, n_synth);
    else
        return nullptr;
    S->world.assign_ids(S->ids);
    S->bvh = BVHNode::from_scene(S->world, 0.0f, 1.0f, axis);
    return S.release();
}
void orc_scene_destroy(orc_scene* s) { delete s; }
void orc_scene_info(const orc_scene* s, int64_t info[4]) {
    info[0] = s->ids.prim;
    info[1] = s->ids.inst;
    info[2] = s->ids.medium;
    info[3] = (int64_t)s->world.shapes.size();
}
int64_t orc_scene_spheres(const orc_scene* s, int64_t cap, float* out) {
    int64_t n = (int64_t)s->world.shapes.size();
    for (int64_t i = 0; i < n && i < cap; i++) {
        const Sphere* sp = dynamic_cast<const Sphere*>(s->world.shapes[i].get());
        float nan = std::numeric_limits<float>::quiet_NaN();
        out[4 * i] = sp ? sp->center.x : nan;
        out[4 * i + 1] = sp ? sp->center.y : nan;
        out[4 * i + 2] = sp ? sp->center.z : nan;
        out[4 * i + 3] = sp ? sp->radius : nan;
    }
    return n;
}
void orc_scene_preset(const orc_scene* s, float cam[11]) {
    cam[0] = s->cam_origin.x;
    cam[1] = s->cam_origin.y;
    cam[2] = s->cam_origin.z;
    cam[3] = s->cam_target.x;
    cam[4] = s->cam_target.y;
    cam[5] = s->cam_target.z;
    cam[6] = s->vfov_deg;
    cam[7] = s->aperture;
    cam[8] = s->background.x;
    cam[9] = s->background.y;
    cam[10] = s->background.z;
}
static void camera_to24(const Camera& c, float out[24]) {
    const V3* v[7] = {&c.origin, &c.lower_left_corner, &c.horizontal, &c.vertical, &c.u, &c.v, &c.w};
    for (int i = 0; i < 7; i++) {
        out[3 * i] = v[i]->x;
        out[3 * i + 1] = v[i]->y;
        out[3 * i + 2] = v[i]->z;
    }
    out[21] = c.lens_radius;
    out[22] = c.time_start;
    out[23] = c.time_end;
}
static Camera camera_from24(const float in[24]) {
    Camera c;
    V3* v[7] = {&c.origin, &c.lower_left_corner, &c.horizontal, &c.vertical, &c.u, &c.v, &c.w};
    for (int i = 0; i < 7; i++) *v[i] = {in[3 * i], in[3 * i + 1], in[3 * i + 2]};
    c.lens_radius = in[21];
    c.time_start = in[22];
    c.time_end = in[23];
    return c;
}
void orc_camera_new(const float o[3], const float tg[3], const float up[3], float vfov, float aspect, float aperture,
                    float focus, float t0, float t1, float out[24]) {
    Camera c = Camera::make(v3(o[0], o[1], o[2]), v3(tg[0], tg[1], tg[2]), v3(up[0], up[1], up[2]), vfov, aspect,
                            aperture, focus, t0, t1);
    camera_to24(c, out);
}

int orc_intersect_batch(const orc_scene* S, int mode, int64_t n, const float* ox, const float* oy, const float* oz,
                        const float* dx, const float* dy, const float* dz, const float* time, float tmin, float tmax,
                        uint32_t seed, int32_t* prim, int32_t* inst, float* t, float* normal, uint8_t* front, float* uv,
                        float* point, uint64_t* counters, int threads) {
    if (!S || n < 0) return -1;
    const World world = world_of(*S, mode);
    std::atomic<uint64_t> c_rays{0}, c_box{0}, c_prim{0};
    parallel_for(n, threads, [&](int64_t b, int64_t e, int) {
        g_cnt_rays = g_cnt_box = g_cnt_prim = 0;
        for (int64_t i = b; i < e; i++) {
            PathCtx ctx{path_stream(seed, (uint32_t)i, 0), 0, false, 0.0};
            g_path = &ctx;
            Ray ray{{ox[i], oy[i], oz[i]}, {dx[i], dy[i], dz[i]}, time[i]};
            HitRecord rec;
            g_cnt_rays++;
            bool hit = world.intersect(ray, tmin, tmax, rec);
            if (prim) prim[i] = hit ? rec.prim_id : -1;
            if (inst) inst[i] = hit ? rec.inst_id : -1;
            if (t) t[i] = hit ? rec.t : INF;
            if (normal) {
                normal[3 * i] = hit ? rec.normal.x : 0;
                normal[3 * i + 1] = hit ? rec.normal.y : 0;
                normal[3 * i + 2] = hit ? rec.normal.z : 0;
            }
            if (front) front[i] = hit && rec.front_face;
            if (uv) {
                uv[2 * i] = hit ? rec.u : 0;
                uv[2 * i + 1] = hit ? rec.v : 0;
            }
            if (point) {
                point[3 * i] = hit ? rec.point.x : 0;
                point[3 * i + 1] = hit ? rec.point.y : 0;
                point[3 * i + 2] = hit ? rec.point.z : 0;
            }
        }
        g_path = nullptr;
        c_rays += g_cnt_rays;
        c_box += g_cnt_box;
        c_prim += g_cnt_prim;
    });
    if (counters) {
        counters[0] = c_rays;
        counters[1] = c_box;
        counters[2] = c_prim;
    }
    return 0;
}

int orc_gen_rays(const orc_scene* S, int w, int h, float aspect, uint32_t seed, int which, float* rays7,
                 uint8_t* valid) {
    if (!S || w < 2 || h < 2) return -1;
    Camera cam = scene_camera(*S, aspect);
    const World world = world_of(*S, 1);
    int64_t n = (int64_t)w * h;
    for (int64_t p = 0; p < n; p++) {
        PathCtx ctx{path_stream(seed, (uint32_t)p, 0), 0, false, 0.0};
        g_path = &ctx;
        Ray r = camera_sample_ray(cam, w, h, (int)(p % w), (int)(p / w), ctx.rng);
        uint8_t ok = 1;
        if (which == 1) {
            HitRecord rec;
            ok = 0;
            ctx.rng.n = 0;  // scatter counters of bounce 0
            if (world.intersect(r, 0.005f, INF, rec)) {
                Ray sc{{0, 0, 0}, {0, 0, 0}, r.time};
                V3 att{1, 1, 1};
                if (rec.material->scatter(r, rec, att, sc)) {
                    r = sc;
                    ok = 1;
                }
            }
        }
        rays7[0 * n + p] = r.origin.x;
        rays7[1 * n + p] = r.origin.y;
        rays7[2 * n + p] = r.origin.z;
        rays7[3 * n + p] = r.direction.x;
        rays7[4 * n + p] = r.direction.y;
        rays7[5 * n + p] = r.direction.z;
        rays7[6 * n + p] = r.time;
        if (valid) valid[p] = ok;
    }
    g_path = nullptr;
    return 0;
}

int orc_render(const orc_scene* S, int mode, int w, int h, float aspect, int s0, int s1, int max_depth, uint32_t seed,
               int x0, int y0, int x1, int y1, float* sum_rgb, uint64_t* rays_out, uint64_t* counters, int threads) {
    if (!S || w < 2 || h < 2 || s1 < s0 || x0 < 0 || y0 < 0 || x1 > w || y1 > h) return -1;
    Camera cam = scene_camera(*S, aspect);
    const World world = world_of(*S, mode);
    V3 bg = S->background;
    std::atomic<uint64_t> c_rays{0}, c_box{0}, c_prim{0};
    int64_t rows = y1 - y0;
    if (rows <= 0 || x1 <= x0) return 0;
    std::vector<std::thread> pool;
    std::atomic<int64_t> next{0};
    int64_t npix = rows * (int64_t)(x1 - x0);
    auto work = [&]() {
        g_cnt_rays = g_cnt_box = g_cnt_prim = 0;
        for (;;) {
            int64_t b = next.fetch_add(64);
            if (b >= npix) break;
            int64_t e = std::min(npix, b + 64);
            for (int64_t q = b; q < e; q++) {
                int i = x0 + (int)(q % (x1 - x0));
                int j = y0 + (int)(q / (x1 - x0));
                uint32_t pixel = (uint32_t)((int64_t)j * w + i);
                float* acc = sum_rgb + 3 * (size_t)pixel;
                for (int s = s0; s < s1; s++) {
                    PathCtx ctx{path_stream(seed, pixel, (uint32_t)s), 0, false, 0.0};
                    g_path = &ctx;
                    Ray r = camera_sample_ray(cam, w, h, i, j, ctx.rng);
                    V3 c = ray_color(r, bg, world, max_depth, max_depth);
                    acc[0] = acc[0] + c.x;
                    acc[1] = acc[1] + c.y;
                    acc[2] = acc[2] + c.z;
                }
            }
        }
        g_path = nullptr;
        c_rays += g_cnt_rays;
        c_box += g_cnt_box;
        c_prim += g_cnt_prim;
    };
    if (threads <= 1)
        work();
    else {
        for (int t = 0; t < threads; t++) pool.emplace_back(work);
        for (auto& th : pool) th.join();
    }
    if (rays_out) *rays_out += c_rays;
    if (counters) {
        counters[0] = c_rays;
        counters[1] = c_box;
        counters[2] = c_prim;
    }
    return 0;
}

void orc_resolve_u8(const float* sum_rgb, int64_t n_pixels, int spp, uint8_t* out) {
    float inv = 1.0f / (float)spp;
    for (int64_t i = 0; i < n_pixels * 3; i++) out[i] = float_to_u8_color(sum_rgb[i] * inv);
}

// the write_color the reference keeps commented out (main.rs:35-42): square root of the mean, then the same quantisation
void orc_resolve_u8_sqrt(const float* sum_rgb, int64_t n_pixels, int spp, uint8_t* out) {
    float inv = 1.0f / (float)spp;
    for (int64_t i = 0; i < n_pixels * 3; i++) out[i] = float_to_u8_color(std::sqrt(sum_rgb[i] * inv));
}

void orc_rng_u32(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t first, int n, uint32_t* out) {
    Stream s = path_stream(seed, pixel, sample);
    s.n = first;
    for (int i = 0; i < n; i++) out[i] = s.next_u32();
}
void orc_rng_f32(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t first, int n, float* out) {
    Stream s = path_stream(seed, pixel, sample);
    s.n = first;
    for (int i = 0; i < n; i++) out[i] = s.gen_f32();
}
void orc_rng_range(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t first, float lo, float hi, int n, float* out) {
    Stream s = path_stream(seed, pixel, sample);
    s.n = first;
    for (int i = 0; i < n; i++) out[i] = s.gen_range(lo, hi);
}
double orc_rng_medium_xi(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t medium) {
    Stream s = path_stream(seed, pixel, sample);
    return medium_xi(s, bounce, medium);
}

void orc_kat_sphere(const float c[3], float r, const float ray[7], float tmin, float tmax, float out[11]) {
    Sphere s(v3(c[0], c[1], c[2]), r, nullptr);
    HitRecord rec;
    bool hit = s.intersect(ray_from7(ray), tmin, tmax, rec);
    rec_to_out(hit, rec, out);
}
void orc_kat_moving_sphere(const float c0[3], const float c1[3], float r, float t0, float t1, const float ray[7],
                           float tmin, float tmax, float out[11]) {
    MovingSphere s(v3(c0[0], c0[1], c0[2]), v3(c1[0], c1[1], c1[2]), r, t0, t1, nullptr);
    HitRecord rec;
    bool hit = s.intersect(ray_from7(ray), tmin, tmax, rec);
    rec_to_out(hit, rec, out);
}
void orc_kat_rect(int axis, float min0, float min1, float max0, float max1, float offset, const float ray[7],
                  float tmin, float tmax, float out[11]) {
    HitRecord rec;
    bool hit = false;
    Ray r = ray_from7(ray);
    if (axis == 0)
        hit = XYRect(min0, min1, max0, max1, offset, nullptr).intersect(r, tmin, tmax, rec);
    else if (axis == 1)
        hit = XZRect(min0, min1, max0, max1, offset, nullptr).intersect(r, tmin, tmax, rec);
    else
        hit = YZRect(min0, min1, max0, max1, offset, nullptr).intersect(r, tmin, tmax, rec);
    rec_to_out(hit, rec, out);
}
void orc_kat_instanced_box(float w, float h, float d, float rot_y_deg, const float trans[3], const float ray[7],
                           float tmin, float tmax, float out[11], float aabb[6]) {
    auto b = mk<Box2>(v3(0, 0, 0), v3(w, h, d), nullptr);
    M4 m = m4_mul(m4_translation(v3(trans[0], trans[1], trans[2])), m4_rotation_y(degree_to_rad(rot_y_deg)));
    TransformedObject obj(b, m);
    HitRecord rec;
    bool hit = obj.intersect(ray_from7(ray), tmin, tmax, rec);
    rec_to_out(hit, rec, out);
    if (aabb) {
        aabb[0] = obj.aabb.min.x;
        aabb[1] = obj.aabb.min.y;
        aabb[2] = obj.aabb.min.z;
        aabb[3] = obj.aabb.max.x;
        aabb[4] = obj.aabb.max.y;
        aabb[5] = obj.aabb.max.z;
    }
}
int orc_kat_aabb_hit(const float bmin[3], const float bmax[3], const float ray[7], float tmin, float tmax) {
    AABB b;
    b.min = {bmin[0], bmin[1], bmin[2]};
    b.max = {bmax[0], bmax[1], bmax[2]};
    return b.hit(ray_from7(ray), tmin, tmax) ? 1 : 0;
}
void orc_kat_medium_sphere(const float c[3], float r, float density, double xi, const float ray[7], float tmin,
                           float tmax, float out[11]) {
    ConstantMedium m(mk<Sphere>(v3(c[0], c[1], c[2]), r, nullptr), density, v3(1, 1, 1));
    PathCtx ctx{path_stream(0, 0, 0), 0, true, xi};
    g_path = &ctx;
    HitRecord rec;
    bool hit = m.intersect(ray_from7(ray), tmin, tmax, rec);
    g_path = nullptr;
    rec_to_out(hit, rec, out);
}
void orc_kat_reflect(const float v[3], const float n[3], float out[3]) {
    V3 r = reflect(v3(v[0], v[1], v[2]), v3(n[0], n[1], n[2]));
    out[0] = r.x;
    out[1] = r.y;
    out[2] = r.z;
}
void orc_kat_refract(const float uv[3], const float n[3], float ratio, float out[3]) {
    V3 r = refract(v3(uv[0], uv[1], uv[2]), v3(n[0], n[1], n[2]), ratio);
    out[0] = r.x;
    out[1] = r.y;
    out[2] = r.z;
}
float orc_kat_reflectance(float cosine, float ref_idx) { return reflectance(cosine, ref_idx); }
uint8_t orc_kat_quantise(float f) { return float_to_u8_color(f); }
void orc_kat_image_value(const uint8_t* rgb, int w, int h, float u, float v, float out[3]) {
    V3 c = image_value(rgb, (uint32_t)w, (uint32_t)h, u, v);
    out[0] = c.x;
    out[1] = c.y;
    out[2] = c.z;
}
static void perlin_from(const float* vec, const int32_t* perm, Perlin& p) {
    for (int i = 0; i < 256; i++) p.rand_vec[i] = {vec[3 * i], vec[3 * i + 1], vec[3 * i + 2]};
    std::memcpy(p.perm_x, perm, 1024);
    std::memcpy(p.perm_y, perm + 256, 1024);
    std::memcpy(p.perm_z, perm + 512, 1024);
}
float orc_kat_perlin_noise(const float* vec, const int32_t* perm, const float p[3]) {
    Perlin pn;
    perlin_from(vec, perm, pn);
    return pn.noise(v3(p[0], p[1], p[2]));
}
float orc_kat_perlin_turb(const float* vec, const int32_t* perm, const float p[3], int depth) {
    Perlin pn;
    perlin_from(vec, perm, pn);
    return pn.turb(v3(p[0], p[1], p[2]), depth);
}
void orc_kat_checker(const float odd[3], const float even[3], const float p[3], float out[3]) {
    CheckeredTexture t(v3(odd[0], odd[1], odd[2]), v3(even[0], even[1], even[2]));
    V3 c = t.value(0, 0, v3(p[0], p[1], p[2]));
    out[0] = c.x;
    out[1] = c.y;
    out[2] = c.z;
}
void orc_kat_sphere_uv(const float n[3], float out[2]) { get_sphere_uv(v3(n[0], n[1], n[2]), out[0], out[1]); }
void orc_kat_scatter(int kind, const float albedo[3], float param, const float ray[7], const float point[3],
                     const float normal[3], int front, uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t ctr,
                     float out[11]) {
    std::shared_ptr<Material> m;
    V3 alb{albedo[0], albedo[1], albedo[2]};
    if (kind == 0)
        m = mk<LambertianMat>(alb);
    else if (kind == 1)
        m = mk<MetalMat>(alb, param);
    else if (kind == 2)
        m = mk<DielectricMat>(param);
    else
        m = mk<IsotropicMat>(alb);
    PathCtx ctx{path_stream(seed, pixel, sample), 0, false, 0.0};
    ctx.rng.n = ctr;
    g_path = &ctx;
    HitRecord rec;
    rec.point = {point[0], point[1], point[2]};
    rec.normal = {normal[0], normal[1], normal[2]};
    rec.front_face = front != 0;
    Ray r = ray_from7(ray);
    Ray sc{{0, 0, 0}, {0, 0, 0}, r.time};
    V3 att{1, 1, 1};
    bool ok = m->scatter(r, rec, att, sc);
    g_path = nullptr;
    out[0] = ok ? 1.0f : 0.0f;
    out[1] = sc.origin.x;
    out[2] = sc.origin.y;
    out[3] = sc.origin.z;
    out[4] = sc.direction.x;
    out[5] = sc.direction.y;
    out[6] = sc.direction.z;
    out[7] = att.x;
    out[8] = att.y;
    out[9] = att.z;
    out[10] = (float)(ctx.rng.n - ctr);
}
void orc_kat_camera_ray(const float cam24[24], int w, int h, int i, int j, uint32_t seed, uint32_t sample,
                        float ray[7]) {
    Camera c = camera_from24(cam24);
    Stream s = path_stream(seed, (uint32_t)(j * w + i), sample);
    Ray r = camera_sample_ray(c, w, h, i, j, s);
    ray[0] = r.origin.x;
    ray[1] = r.origin.y;
    ray[2] = r.origin.z;
    ray[3] = r.direction.x;
    ray[4] = r.direction.y;
    ray[5] = r.direction.z;
    ray[6] = r.time;
}

}  // extern "C"
