// Device functions of the path tracer: vector helpers, counter-based RNG, primitive tests, BVH traversal,
// hit records, textures and materials.  Included by rt_kernels.hip only.
//
// Arithmetic rules.  Everything that decides WHICH primitive is hit and at what t is written in the
// operation order of the Rust reference and compiled with -ffp-contract=off (rustc never fuses a*b+c);
// HIP's f32 divide and sqrt are correctly rounded by default.  Box slab tests only cull, so they use a
// fast reciprocal and are padded to stay conservative.  No MFMA: nothing here is a dense contraction.
#include "rt_kernels.h"

#include <hip/hip_runtime.h>

using namespace rtd;

#define DEV __device__ __forceinline__

namespace {

constexpr float INF_F = __builtin_huge_valf();
constexpr float F32_EPSILON = 1.1920929e-7f;
constexpr float PI_F = 3.14159265358979323846f;

struct f3 {
    float x, y, z;
};
DEV f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
DEV f3 ld3(const float* p) { return f3{p[0], p[1], p[2]}; }
DEV f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
DEV f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
DEV f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
DEV f3 operator*(float s, f3 a) { return {s * a.x, s * a.y, s * a.z}; }
DEV f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
// Correctly rounded quotients that share one reciprocal.  r = 1/d refined once by Newton; a quotient is
// q = x*r corrected by its exact remainder, q + fma(-d, q, x)*r (Markstein).  This reproduces the IEEE
// division bit for bit (4.3e9 random operand pairs and the unit-vector case checked on the device,
// scripts/div_check.hip) provided nothing over- or underflows on the way, which holds for d and x inside
// 2^-60 .. 2^60; x == 0 is fine as well.  A numerator below 2^-100 could lose its last bit.
DEV float recip_refined(float d) {
    float r = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
}
DEV float div_by(float x, float d, float r) {
    float q = x * r;
    return __builtin_fmaf(__builtin_fmaf(-d, q, x), r, q);
}
DEV bool in_div_window(float d) { return __builtin_fabsf(d) > 0x1p-60f && __builtin_fabsf(d) < 0x1p60f; }
DEV f3 operator/(f3 a, float s) {  // glam: Vec3A / f32 divides each component
    float big = fmaxf(fmaxf(__builtin_fabsf(a.x), __builtin_fabsf(a.y)), __builtin_fabsf(a.z));
    if (in_div_window(s) && big < 0x1p60f) {
        float r = recip_refined(s);
        return {div_by(a.x, s, r), div_by(a.y, s, r), div_by(a.z, s, r)};
    }
    return {a.x / s, a.y / s, a.z / s};  // zero, huge, tiny or non-finite operands
}
DEV f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
DEV float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }  // glam: (x*x + y*y) + z*z
DEV float len3(f3 a) { return __builtin_sqrtf(dot3(a, a)); }
DEV f3 normalize3(f3 a) { return a / len3(a); }
DEV float comp(f3 a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }

// 3x4 row-major affine maps in glam's accumulation order
DEV f3 xf_point(const float* m, f3 v) {
    return {((m[0] * v.x + m[1] * v.y) + m[2] * v.z) + m[3], ((m[4] * v.x + m[5] * v.y) + m[6] * v.z) + m[7],
            ((m[8] * v.x + m[9] * v.y) + m[10] * v.z) + m[11]};
}
DEV f3 xf_vector(const float* m, f3 v) {
    return {(m[0] * v.x + m[1] * v.y) + m[2] * v.z, (m[4] * v.x + m[5] * v.y) + m[6] * v.z,
            (m[8] * v.x + m[9] * v.y) + m[10] * v.z};
}
// The world ray (o, d) as instance g's own object sees it: TransformedObject::intersect (shapes.rs:66-70) of every
// instance around g, outermost first, then of g itself -- the same products in the same order as the nested calls.
DEV void instance_ray(const Scene& S, int g, f3& o, f3& d) {
    const uint32_t depth = S.groups[g].depth;
    for (uint32_t level = 0; level <= depth; level++) {
        int a = g;
        for (uint32_t k = depth - level; k > 0u; k--) a = S.groups[a].parent;
        const Group& G = S.groups[a];
        if (G.has_xform) {
            o = xf_point(G.minv, o);
            d = xf_vector(G.minv, d);
        }
    }
}

// ---- tables the shading kernels stage in LDS ------------------------------------------------------------
// Shading looks its records up with a different index in every lane: hit primitive -> material -> texture ->
// checker child, each fetch depending on the one before, and a Perlin lattice lookup is a gather of 64
// different addresses per wave.  Through the vector L1 a gather costs one cache line per lane and clock (the 98
// gathers of a seven-octave turbulence made the L1 the limit), and every dependent step pays an L1/L2 latency.
// So the calling kernel copies what is small enough into LDS once per block (stage_tables): Perlin table 0
// whenever the scene has one, and the primitive / material / texture records of small (flat) scenes.
typedef const __attribute__((address_space(3))) PerlinTable LdsPerlinTable;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) v4f LdsFloat4;
struct LdsTables {
    LdsPerlinTable* perlin0;
    bool has_perlin0;
    LdsFloat4* prims;  // two float4 per 32-byte record, as in the global arrays
    LdsFloat4* mats;
    LdsFloat4* texs;
    bool has_scene;
};
DEV void load_record(const LdsTables& L, LdsFloat4* lds, const void* global, uint32_t index, float4& r0, float4& r1) {
    if (L.has_scene) {
        const v4f a = lds[2u * index], b = lds[2u * index + 1u];
        r0 = make_float4(a.x, a.y, a.z, a.w);
        r1 = make_float4(b.x, b.y, b.z, b.w);
    } else {
        const float4* g = reinterpret_cast<const float4*>(global) + 2u * index;
        r0 = g[0];
        r1 = g[1];
    }
}

// ---- counter-based RNG ------------------------------------------------------
//   mix32(x): x^=x>>16; x*=0x7feb352d; x^=x>>15; x*=0x846ca68b; x^=x>>16
//   stream (seed,pixel,sample): h=mix32(seed^0x51ED270B); k0=mix32(pixel^h);
//       k1=mix32(k0^mix32(sample+0x68E31DA4)); draw at counter c = mix32((k0+c)^k1)
//   Each consumer of a path's stream counts up from its own fixed counter, so the length of one rejection
//   loop never shifts the draws of another: jitter + ray time from CTR_CAMERA, the lens disk from CTR_LENS,
//   the scatter draws of bounce b from b<<24.
constexpr uint32_t CTR_CAMERA = 0xFF000000U, CTR_LENS = 0xFE000000U;
DEV uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
struct Rng {
    uint32_t k0, k1, n;
    DEV uint32_t next() {
        uint32_t r = mix32((k0 + n) ^ k1);
        n++;
        return r;
    }
    DEV float f32() { return (float)(next() >> 8) * (1.0f / 16777216.0f); }  // rand gen::<f32>()
    DEV float range(float lo, float hi) {                                       // rand gen_range(lo..hi)
        float scale = hi - lo;
        for (;;) {
            float v12 = __uint_as_float(0x3f800000U | (next() >> 9));
            float v01 = v12 - 1.0f;
            float r = v01 * scale + lo;
            if (r < hi) return r;
        }
    }
    // range(-1, 1): (v12 - 1) * 2 - 1 is exact and at most 1 - 2^-22, so the redraw can never fire
    DEV float range_pm1() { return __uint_as_float(0x40000000U | (next() >> 9)) - 3.0f; }
};
DEV void stream_key(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t& k0, uint32_t& k1) {
    uint32_t h = mix32(seed ^ 0x51ED270BU);
    k0 = mix32(pixel ^ h);
    k1 = mix32(k0 ^ mix32(sample + 0x68E31DA4U));
}
// f64 uniform of medium `m` at bounce `b`: rand gen_range(0.0..1.0) from a (hi,lo) word pair
DEV uint32_t medium_salt(uint32_t medium) { return mix32(medium + 0x3C6EF372U); }
DEV double medium_xi_salted(uint32_t k0, uint32_t k1, uint32_t bounce, uint32_t salt) {
    uint32_t hi = mix32((k0 + 0x80000000U + 2u * bounce) ^ k1 ^ salt);
    uint32_t lo = mix32((k0 + 0x80000001U + 2u * bounce) ^ k1 ^ salt);
    unsigned long long u = ((unsigned long long)hi << 32) | lo;
    return __longlong_as_double((long long)((u >> 12) | 0x3FF0000000000000ULL)) - 1.0;
}
DEV double medium_xi(uint32_t k0, uint32_t k1, uint32_t bounce, uint32_t medium) { return medium_xi_salted(k0, k1, bounce, medium_salt(medium)); }
DEV f3 rand_range3(Rng& r, float lo, float hi) {
    float x = r.range(lo, hi), y = r.range(lo, hi), z = r.range(lo, hi);
    return {x, y, z};
}
DEV f3 rand_in_unit_sphere(Rng& r) {  // vec3_helpers.rs:25-31
    for (;;) {
        float x = r.range_pm1(), y = r.range_pm1(), z = r.range_pm1();
        f3 p{x, y, z};
        if (!(dot3(p, p) >= 1.0f)) return p;
    }
}

// ---- primitives ---------------------------------------------------------------
struct Hit {
    float t;
    uint32_t prim;  // index into prims[], HIT_NONE, or HIT_MEDIUM_BIT | group
    int grp;        // instance group of the hit primitive, or -1
    uint32_t key;   // tie key: caller id of the primitive (of the side that was hit, for a box), MEDIUM_KEY_BASE + k for media
};

DEV f3 moving_center(const Scene& S, f3 c0, uint32_t aux, float time) {  // shapes.rs:189-191
    const float4* mp = reinterpret_cast<const float4*>(S.moving + aux);
    float4 m0 = mp[0], m1 = mp[1];
    f3 c1{m0.x, m0.y, m0.z};
    float t0 = m0.w, t1 = m1.x;
    return c0 + (((time - t0) / (t1 - t0)) * (c1 - c0));
}

// One side of a box as the rect record it stands for (Box2::full_box, shapes.rs:459-466): sides 0,1 are the
// XYRects at min.z / max.z, 2,3 the XZRects at min.y / max.y, 4,5 the YZRects at min.x / max.x; a rect record
// is (k, min0, min1, max0 | max1).
DEV void box_side_rect(float4 A, float4 B, uint32_t side, uint32_t& kind, float& k, float& min0, float& min1, float& max0, float& max1) {
    const f3 lo{A.x, A.y, A.z}, hi{A.w, B.x, B.w};
    const int ax = side < 2u ? 2 : (side < 4u ? 1 : 0);
    const int a0 = ax == 0 ? 1 : 0, a1 = ax == 2 ? 1 : 2;
    kind = side < 2u ? PK_RECT_XY : (side < 4u ? PK_RECT_XZ : PK_RECT_YZ);
    k = (side & 1u) ? comp(hi, ax) : comp(lo, ax);
    min0 = comp(lo, a0);
    min1 = comp(lo, a1);
    max0 = comp(hi, a0);
    max1 = comp(hi, a1);
}
// n0 / d and n1 / d -- (k0 - o) / d and (k1 - o) / d, the plane distances of two parallel sides (shapes.rs:280,
// 344, 408), or the two roots of a sphere (shapes.rs:126, 131).  Both quotients share the denominator, so they come
// from one refined reciprocal (bit-identical to the IEEE division inside the window of div_by; a numerator that
// is tiny but not zero takes the plain division).
DEV void quotient_pair(float n0, float n1, float d, float& t0, float& t1) {
    const float big = fmaxf(__builtin_fabsf(n0), __builtin_fabsf(n1));
    const float small = fminf(n0 == 0.0f ? 1.0f : __builtin_fabsf(n0), n1 == 0.0f ? 1.0f : __builtin_fabsf(n1));
    if (in_div_window(d) && big < 0x1p60f && small > 0x1p-60f) {
        const float r = recip_refined(d);
        t0 = div_by(n0, d, r);
        t1 = div_by(n1, d, r);
    } else {
        t0 = n0 / d;
        t1 = n1 / d;
    }
}
DEV void plane_pair(float k0, float k1, float o, float d, float& t0, float& t1) { quotient_pair(k0 - o, k1 - o, d, t0, t1); }
// Sphere::intersect up to the root that is taken (shapes.rs:117-139): the near root if it lies in [tmin, tmax], else the far one.
DEV bool sphere_candidate(f3 c, float r, f3 o, f3 d, float a, float tmin, float tmax, float& t) {
    f3 oc = o - c;
    float half_b = dot3(oc, d);
    float cc = dot3(oc, oc) - r * r;
    float disc = half_b * half_b - a * cc;
    if (disc < 0.0f) return false;
    float sq = __builtin_sqrtf(disc);
    t = (-half_b - sq) / a;
    if (t < tmin || t > tmax) {
        t = (-half_b + sq) / a;
        if (t < tmin || t > tmax) return false;
    }
    return true;
}
// Both roots of a sphere without any range (what the boundary queries of a medium pick from): false when the line misses.
DEV bool sphere_roots(f3 c, float r, f3 o, f3 d, float a, float& c0, float& c1) {
    f3 oc = o - c;
    float half_b = dot3(oc, d);
    float cc = dot3(oc, oc) - r * r;
    float disc = half_b * half_b - a * cc;
    if (disc < 0.0f) return false;
    float sq = __builtin_sqrtf(disc);
    quotient_pair(-half_b - sq, -half_b + sq, a, c0, c1);
    return true;
}
// One rect side inside Scene::intersect over the box's sides (scene.rs:27-41): accepted when t lies in
// [tmin, closest] and the in-plane point inside the bounds, all inclusive; a later side with the same t wins.
DEV void box_side(float t, float p0, float p1, float min0, float min1, float max0, float max1, float tmin, uint32_t s, float& closest,
                  uint32_t& side) {
    if (t < tmin || t > closest) return;
    if (p0 < min0 || p0 > max0 || p1 < min1 || p1 > max1) return;
    closest = t;
    side = s;
}
// Candidate t of one primitive in [tmin, tmax] (inclusive, as the reference's "< t_min || > t_max" rejects).
// id_out: the caller's id of the primitive that was hit -- for a box, of the side that won.
// BOXES: box records can occur (only scenes traced without a tree hold them, see rt_scene_commit); the tree walk is
// compiled without that branch, whose registers would cost it a resident wave per SIMD.
template <bool BOXES>
DEV bool prim_candidate(const Scene& S, uint32_t pi, f3 o, f3 d, float a, float time, float tmin, float tmax, float& t_out,
                        uint32_t& id_out) {
    const float4* pp = reinterpret_cast<const float4*>(S.prims + pi);
    float4 A = pp[0];
    float4 B = pp[1];
    uint32_t info = __float_as_uint(B.y);
    id_out = __float_as_uint(B.z);
    uint32_t kind = info & 7u;
    float t;
    if (kind <= PK_MOVING_SPHERE) {  // shapes.rs:117-139, 195-217
        f3 c{A.x, A.y, A.z};
        if (kind == PK_MOVING_SPHERE) c = moving_center(S, c, __float_as_uint(B.w), time);
        if (!sphere_candidate(c, A.w, o, d, a, tmin, tmax, t)) return false;
    } else if (BOXES && kind == PK_BOX) {  // Box2::intersect = Scene::intersect over the six sides (shapes.rs:474-477)
        const f3 lo{A.x, A.y, A.z}, hi{A.w, B.x, B.w};
        float tz0, tz1, ty0, ty1, tx0, tx1;
        float closest = tmax;
        uint32_t side = 6u;
        plane_pair(lo.z, hi.z, o.z, d.z, tz0, tz1);
        box_side(tz0, o.x + tz0 * d.x, o.y + tz0 * d.y, lo.x, lo.y, hi.x, hi.y, tmin, 0u, closest, side);
        box_side(tz1, o.x + tz1 * d.x, o.y + tz1 * d.y, lo.x, lo.y, hi.x, hi.y, tmin, 1u, closest, side);
        plane_pair(lo.y, hi.y, o.y, d.y, ty0, ty1);
        box_side(ty0, o.x + ty0 * d.x, o.z + ty0 * d.z, lo.x, lo.z, hi.x, hi.z, tmin, 2u, closest, side);
        box_side(ty1, o.x + ty1 * d.x, o.z + ty1 * d.z, lo.x, lo.z, hi.x, hi.z, tmin, 3u, closest, side);
        plane_pair(lo.x, hi.x, o.x, d.x, tx0, tx1);
        box_side(tx0, o.y + tx0 * d.y, o.z + tx0 * d.z, lo.y, lo.z, hi.y, hi.z, tmin, 4u, closest, side);
        box_side(tx1, o.y + tx1 * d.y, o.z + tx1 * d.z, lo.y, lo.z, hi.y, hi.z, tmin, 5u, closest, side);
        if (side == 6u) return false;
        t = closest;
        id_out += side;
    } else {  // shapes.rs:279-292, 343-356, 407-420
        int ax = kind == PK_RECT_XY ? 2 : (kind == PK_RECT_XZ ? 1 : 0);
        int a0 = ax == 0 ? 1 : 0;
        int a1 = ax == 2 ? 1 : 2;
        t = (A.x - comp(o, ax)) / comp(d, ax);
        if (t < tmin || t > tmax) return false;
        float x = comp(o, a0) + t * comp(d, a0);
        float y = comp(o, a1) + t * comp(d, a1);
        if (x < A.y || x > A.w || y < A.z || y > B.x) return false;
    }
    t_out = t;
    return true;
}

// Closest-hit update with the exact-tie rule "higher id wins" (scene.rs:32-38: the later shape wins).
template <bool BOXES>
DEV void test_prim(const Scene& S, uint32_t pi, f3 o, f3 d, float a, float time, float tmin, int grp, Hit& best) {
    float t;
    uint32_t id;
    if (!prim_candidate<BOXES>(S, pi, o, d, a, time, tmin, best.t, t, id)) return;
    if (t == best.t && best.prim != HIT_NONE && id < best.key) return;
    best.t = t;
    best.prim = pi;
    best.grp = grp;
    best.key = id;
}

// Primitives of a medium's boundary: they sit in one run from (root_ref & 0x03FFFFFF) on; nothing stands inside a
// medium, so its Group::depth is free to hold their number.
DEV uint32_t medium_boundary_count(const Group& G) {
    const uint32_t short_count = ((G.root_ref >> 26) & 15u) + 1u;  // up to 15 primitives: in the reference itself; 16 stands for "16 or more"
    return short_count < 16u ? short_count : G.depth;
}
// Scene::intersect over a primitive list (scene.rs:27-41), however long; only t is needed by the medium.
DEV bool list_closest(const Scene& S, uint32_t first, uint32_t cnt, f3 o, f3 d, float a, float time, float tmin, float tmax, float& t_out) {
    float closest = tmax;
    bool any = false;
    for (uint32_t i = 0; i < cnt; i++) {
        float t;
        uint32_t id;
        if (prim_candidate<false>(S, first + i, o, d, a, time, tmin, closest, t, id)) {
            closest = t;
            any = true;
        }
    }
    t_out = closest;
    return any;
}

// The candidate distances of one primitive without any [tmin, tmax]: bit 0 of the result says c0 exists (a
// rect's plane hit inside its bounds, a sphere's near root), bit 1 that c1 exists (the sphere's far root).
// The values are the ones prim_candidate computes; pick_root applies its range logic to them.
DEV uint32_t prim_roots(const Scene& S, uint32_t pi, f3 o, f3 d, float a, float time, float& c0, float& c1) {
    const float4* pp = reinterpret_cast<const float4*>(S.prims + pi);
    float4 A = pp[0];
    float4 B = pp[1];
    uint32_t kind = __float_as_uint(B.y) & 7u;
    c0 = 0.0f;
    c1 = 0.0f;
    if (kind <= PK_MOVING_SPHERE) {  // shapes.rs:117-139, 195-217
        f3 c{A.x, A.y, A.z};
        if (kind == PK_MOVING_SPHERE) c = moving_center(S, c, __float_as_uint(B.w), time);
        return sphere_roots(c, A.w, o, d, a, c0, c1) ? 3u : 0u;
    }
    int ax = kind == PK_RECT_XY ? 2 : (kind == PK_RECT_XZ ? 1 : 0);  // shapes.rs:279-292, 343-356, 407-420
    int a0 = ax == 0 ? 1 : 0;
    int a1 = ax == 2 ? 1 : 2;
    float t = (A.x - comp(o, ax)) / comp(d, ax);
    float x = comp(o, a0) + t * comp(d, a0);
    float y = comp(o, a1) + t * comp(d, a1);
    if (x < A.y || x > A.w || y < A.z || y > B.x) return 0u;
    c0 = t;
    return 1u;
}
DEV bool pick_root(uint32_t have, float c0, float c1, float tmin, float tmax, float& t) {
    if (!(have & 1u)) return false;
    t = c0;
    if (t < tmin || t > tmax) {
        if (!(have & 2u)) return false;
        t = c1;
        if (t < tmin || t > tmax) return false;
    }
    return true;
}
// The two boundary queries of ConstantMedium::intersect (shapes.rs:655-664) over a list of at most 8
// primitives: Scene::intersect twice (scene.rs:27-41, the closest hit so far narrows the range), the second
// time from r1 + 0.0001 on.  Both queries look at the same candidate distances, so those are computed once.
DEV bool boundary_pair(const Scene& S, uint32_t first, uint32_t cnt, f3 o, f3 d, float a, float time, float& r1, float& r2) {
    float c0[8], c1[8];
    uint32_t have[8];
#pragma unroll
    for (uint32_t i = 0; i < 8u; i++) have[i] = i < cnt ? prim_roots(S, first + i, o, d, a, time, c0[i], c1[i]) : 0u;
    float closest = INF_F, t;
    bool any = false;
#pragma unroll
    for (uint32_t i = 0; i < 8u; i++)
        if (pick_root(have[i], c0[i], c1[i], -INF_F, closest, t)) {
            closest = t;
            any = true;
        }
    if (!any) return false;
    r1 = closest;
    const float from = r1 + 0.0001f;
    closest = INF_F;
    any = false;
#pragma unroll
    for (uint32_t i = 0; i < 8u; i++)
        if (pick_root(have[i], c0[i], c1[i], from, closest, t)) {
            closest = t;
            any = true;
        }
    r2 = closest;
    return any;
}

DEV bool slab(f3 lo, f3 hi, f3 o, f3 inv, float tmin, float tbest, float& tn);
DEV f3 fast_rcp3(f3 d);
// The free flight of ConstantMedium::intersect (shapes.rs:666-689) once the boundary is crossed at r1 and r2: the
// distances are clamped to [tmin, closest hit so far], the flight length is -1/density * ln(xi) in f64 with xi the
// medium's draw, and a flight that ends inside gives the hit at t.  len: the ray's length |d| (the reference divides by it twice).
DEV bool medium_flight(float r1, float r2, float tmin, float best_t, double len, double neg_inv_density, double xi, float& t) {
    if (r1 < tmin) r1 = tmin;
    if (r2 > best_t) r2 = best_t;
    if (r1 >= r2) return false;
    if (r1 < 0.0f) r1 = 0.0f;
    double inside = (double)(r2 - r1) / len;
    double hit_distance = neg_inv_density * log(xi);
    if (hit_distance > inside) return false;
    t = r1 + (float)(hit_distance / len);
    return true;
}
// ConstantMedium::intersect (shapes.rs:648-701).  wo/wd: the world ray.  A medium that reports a hit
// replaces the closest hit so far unchecked, as Scene::intersect does (scene.rs:33-37): its final_t may
// lie beyond best.t because shapes.rs:676-683 divides by the ray length twice.
// NESTED: media may stand inside instances (the scene's kernels are compiled for that); such a medium is handed the ray
// of the instance around it, as the nested intersect calls of shapes.rs:64-92 would.
template <bool NESTED>
DEV void test_medium(const Scene& S, uint32_t g, f3 wo, f3 wd, float time, float tmin, uint32_t k0, uint32_t k1,
                     uint32_t bounce, Hit& best) {
    const Group& G = S.groups[g];
    if (NESTED && G.parent >= 0) instance_ray(S, G.parent, wo, wd);  // (the same in every lane)
    f3 o = wo, d = wd;
    if (G.has_xform) {  // TransformedObject as the boundary (shapes.rs:66-70); t is shared between spaces
        o = xf_point(G.minv, wo);
        d = xf_vector(G.minv, wd);
    }
    float a = dot3(d, d);
    float r1, r2;
    // (cnt: 16 stands for "16 or more"; the exact number is only looked up where the long list is walked -- with the
    // count an unknown from memory the short, common forms below compile to slower code)
    const uint32_t first = G.root_ref & 0x03FFFFFFu, cnt = ((G.root_ref >> 26) & 15u) + 1u;
    // The first boundary query runs over the whole line (shapes.rs:655); a line that misses the widened box
    // of the boundary misses every primitive of it, so the exact tests are skipped for it.  (A single primitive
    // rejects a line that misses it as early as the box would.)
    float tn_unused;
    if (cnt > 1u && !slab(ld3(G.blo), ld3(G.bhi), o, fast_rcp3(d), -INF_F, INF_F, tn_unused)) return;
    if (cnt == 1u) {
        // one primitive (a sphere in every reference scene, main.rs:226-235, 281-284): the first query over the whole
        // line takes the primitive's first candidate, the second query from r1 + 0.0001 on the first candidate at or
        // beyond that -- the sphere's far root, nothing for a rect (shapes.rs:126-136, 655-656)
        float c0, c1;
        const uint32_t have = prim_roots(S, first, o, d, a, time, c0, c1);
        if (!pick_root(have, c0, c1, -INF_F, INF_F, r1)) return;
        if (!pick_root(have, c0, c1, r1 + 0.0001f, INF_F, r2)) return;
    } else if (cnt <= 8u) {
        if (!boundary_pair(S, first, cnt, o, d, a, time, r1, r2)) return;
    } else {
        // any longer list (ConstantMedium<T> takes any Hittable, shapes.rs:624-646): both queries test every primitive
        const uint32_t n = medium_boundary_count(G);
        if (!list_closest(S, first, n, o, d, a, time, -INF_F, INF_F, r1)) return;
        if (!list_closest(S, first, n, o, d, a, time, r1 + 0.0001f, INF_F, r2)) return;
    }
    const double len = (double)(G.has_xform ? len3(wd) : __builtin_sqrtf(a));  // (without a boundary transform a is |wd|^2 already)
    float t;
    if (!medium_flight(r1, r2, tmin, best.t, len, G.neg_inv_density, medium_xi(k0, k1, bounce, G.order), t)) return;
    best.t = t;
    best.prim = HIT_MEDIUM_BIT | g;
    best.grp = -1;
    best.key = MEDIUM_KEY_BASE + G.order;
}

DEV float pad_up(float x) { return x + __builtin_fabsf(x) * 6e-7f; }
DEV float pad_down(float x) { return x - __builtin_fabsf(x) * 6e-7f; }
DEV f3 fast_rcp3(f3 d) { return {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z)}; }

// Conservative slab test: entry distance in tn; true when [tn, tf] overlaps [tmin, tbest] after padding.
// fminf/fmaxf ignore NaN (0 * inf when a direction component is zero and the origin lies on a slab plane).
DEV bool slab(f3 lo, f3 hi, f3 o, f3 inv, float tmin, float tbest, float& tn) {
    float x0 = (lo.x - o.x) * inv.x, x1 = (hi.x - o.x) * inv.x;
    float y0 = (lo.y - o.y) * inv.y, y1 = (hi.y - o.y) * inv.y;
    float z0 = (lo.z - o.z) * inv.z, z1 = (hi.z - o.z) * inv.z;
    float n = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
    float f = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    tn = n;
    float fp = pad_up(f);
    return (pad_down(n) <= fp) && (fp >= tmin) && (pad_down(n) <= tbest);
}

// Per-lane traversal stack.  The first lds_cap entries live in LDS, laid out [entry][lane] (conflict free).
// Walking the near child first keeps the stack shallow, so a kernel may give LDS to the first few entries
// only (more resident blocks per CU) and send the rare deeper ones to a global spill area laid out
// [entry][thread]; spill == nullptr means that all S.stack_cap entries are in LDS.
struct Stack {
    uint32_t* lds;    // this lane's column
    uint32_t lds_cap;
    uint32_t* spill;  // this thread's column
    uint32_t spill_stride;
};
template <int STRIDE>
DEV void stack_put(const Stack& st, int sp, uint32_t v) {
    if ((uint32_t)sp < st.lds_cap)
        st.lds[sp * STRIDE] = v;
    else
        st.spill[(size_t)((uint32_t)sp - st.lds_cap) * st.spill_stride] = v;
}
// (The LDS read is issued unconditionally, at a clamped index, and the rare spilled entry replaces its value: a
// select between an LDS and a global ADDRESS would turn every pop into a flat load.)
template <int STRIDE>
DEV uint32_t stack_get(const Stack& st, int sp) {
    const uint32_t at = (uint32_t)sp < st.lds_cap ? (uint32_t)sp : 0u;
    uint32_t v = st.lds[at * STRIDE];
    // (the value is pinned here: otherwise the compiler sinks this read into the branch below and merges the two into
    // one flat load of a selected address after all)
    asm volatile("" : "+v"(v));
    if ((uint32_t)sp >= st.lds_cap) v = st.spill[(size_t)((uint32_t)sp - st.lds_cap) * st.spill_stride];
    return v;
}
// Short-stack traversal of the two-level BVH.  stk points at this lane's column of the LDS stack;
// consecutive entries are STRIDE words apart so that the 64 lanes of a wave hit 64 different banks.
DEV uint32_t lane_prefix(unsigned long long mask);
struct TraceCount {
    uint32_t nodes, prims;  // inner nodes fetched, primitive records tested (instrumented builds only)
    // lane use of the walk: wave-level executions of an inner step / a leaf step (counted by the first active
    // lane) against the lanes that took part (nodes for inner steps, leaf_lanes for leaf steps)
    uint32_t inner_waves, leaf_waves, leaf_lanes;
};
DEV bool first_active_lane() { return lane_prefix(__ballot(true)) == 0u; }
// Traversal state of one ray.  cur == REF_EMPTY with sp == 0 means the walk is over.
struct Walk {
    uint32_t cur;
    int sp;
    f3 o, d;         // the ray in the space of the subtree being walked (world, or an instance's object space)
    f3 inv;          // 1/d (fast reciprocal, clamped to +-1e30)
    f3 oi_n, oi_f;   // -(o * inv) lowered / raised by its own rounding slack: see walk_set_ray
    float a;         // |d|^2
    int grp;         // instance being walked, or -1
    Hit best;
};
// Box tests only cull, so a plane distance (plane - o) / d is evaluated as ONE fma(plane, inv, oi), oi = -(o * inv).
// With inv within 1 ulp of 1/d and oi rounded once, the result is off by at most 2e-7 * |t| plus 6e-8 * |o * inv| of
// that axis.  The second part is folded into the constant: the near plane of an axis (the lower one when inv >= 0)
// uses oi_n = oi - 1.2e-7 * |oi|, the far plane oi_f = oi + 1.2e-7 * |oi|; the first part is the 6e-7 * |t| padding
// of slab_fma.  So a box the exact ray touches is never culled, per axis, however small a direction component is.
// A zero component has inv = +-1e30 (clamped) and culls like any other; where a product still overflows, the
// distance is inf or NaN and v_min/v_max/compare are used so that NaN means "no constraint" / "visit".
DEV void walk_set_ray(Walk& w, f3 o, f3 d) {
    w.o = o;
    w.d = d;
    w.a = dot3(d, d);
    const f3 r = fast_rcp3(d);
    w.inv = {__builtin_amdgcn_fmed3f(r.x, -1e30f, 1e30f), __builtin_amdgcn_fmed3f(r.y, -1e30f, 1e30f), __builtin_amdgcn_fmed3f(r.z, -1e30f, 1e30f)};
    const f3 oi{-(o.x * w.inv.x), -(o.y * w.inv.y), -(o.z * w.inv.z)};
    const f3 e{__builtin_fabsf(oi.x) * 1.2e-7f, __builtin_fabsf(oi.y) * 1.2e-7f, __builtin_fabsf(oi.z) * 1.2e-7f};
    w.oi_n = oi - e;
    w.oi_f = oi + e;
}
// A ray with a NaN or infinite component (one path in ~10^8 produces one) has no defined closest hit: every comparison of
// the reference's tests is false for it, so each primitive "hits" and the last one visited wins -- whichever that is in the
// reference's random tree.  Here such a ray hits NOTHING, in every way of tracing alike: the tree is not walked (with all
// box tests passing it would visit every node, bounce after bounce), the loop over a flat scene is not run, and the
// media are not asked.
DEV bool ray_is_finite(f3 wo, f3 wd) {
    const float chk = ((wo.x + wo.y) + wo.z) + ((wd.x + wd.y) + wd.z);
    return __builtin_fabsf(chk) < INF_F;
}
DEV void walk_begin(const Scene& S, Walk& w, f3 wo, f3 wd, float tmax) {
    w.best.t = tmax;
    w.best.prim = HIT_NONE;
    w.best.grp = -1;
    w.best.key = 0;
    w.cur = S.root_ref;
    walk_set_ray(w, wo, wd);
    w.grp = -1;
    w.sp = 0;
    if (!ray_is_finite(wo, wd)) w.cur = REF_EMPTY;  // (see ray_is_finite)
}
// Conservative slab test of the walk: entry distance in tn; false only when the box certainly holds no hit in
// [tmin, tbest].  The near plane of each axis is picked by the sign of inv (one select instead of a min and a max).
DEV bool slab_fma(f3 lo, f3 hi, const Walk& w, float tmin, float tbest, float& tn) {
    const bool px = w.inv.x >= 0.0f, py = w.inv.y >= 0.0f, pz = w.inv.z >= 0.0f;
    const float nx = __builtin_fmaf(px ? lo.x : hi.x, w.inv.x, w.oi_n.x), fx = __builtin_fmaf(px ? hi.x : lo.x, w.inv.x, w.oi_f.x);
    const float ny = __builtin_fmaf(py ? lo.y : hi.y, w.inv.y, w.oi_n.y), fy = __builtin_fmaf(py ? hi.y : lo.y, w.inv.y, w.oi_f.y);
    const float nz = __builtin_fmaf(pz ? lo.z : hi.z, w.inv.z, w.oi_n.z), fz = __builtin_fmaf(pz ? hi.z : lo.z, w.inv.z, w.oi_f.z);
    const float n = fmaxf(fmaxf(nx, ny), nz);
    const float f = fminf(fminf(fx, fy), fz);
    tn = n;
    const float np = __builtin_fmaf(-__builtin_fabsf(n), 6e-7f, n);
    const float fp = __builtin_fmaf(__builtin_fabsf(f), 6e-7f, f);
    const bool miss = (np > fp) || (fp < tmin) || (np > tbest);
    return !miss;
}
DEV bool walk_done(const Walk& w) { return w.cur == REF_EMPTY; }
// Walks inner nodes only (at most max_steps of them), until cur is a leaf, an instance, a sentinel or empty.
// Near child first; a box is entered only if it can still hold a hit at or before the closest one.
// Orders two (entry distance, reference) pairs, nearer first.
DEV void near_first(float& ka, uint32_t& ra, float& kb, uint32_t& rb) {
    const bool swap = kb < ka;
    const float k = swap ? kb : ka;
    const uint32_t r = swap ? rb : ra;
    kb = swap ? ka : kb;
    rb = swap ? ra : rb;
    ka = k;
    ra = r;
}
// The crown in LDS (kernels compiled with CROWN: the trace kernel puts it at the start of its dynamic LDS, the instance
// table first): a node fetched from there costs no address cycles of the texture path -- a wave's scattered 16-byte loads
// take one cycle per lane and instruction there, which is what bounds the walk -- and a fraction of the latency.
// (The reads are typed as LDS reads, address space 3: through plain pointers the compiler folds "this node from LDS, that
// one from memory" into ONE flat load of a selected address, and flat loads take the texture path whatever they hit.)
typedef float crown_v4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const crown_v4 crown_lds_v4;
typedef __attribute__((address_space(3))) const uint32_t crown_lds_u32;
DEV crown_lds_u32* crown_words() {
    extern __shared__ __attribute__((aligned(16))) uint32_t crown_lds_base[];
    return (crown_lds_u32*)crown_lds_base;
}
// 16-byte piece k of crown record `index` of `pieces` pieces
DEV float4 crown_piece(uint32_t index, uint32_t pieces, uint32_t k) {
    crown_lds_v4* p = (crown_lds_v4*)(crown_words() + CROWN_INST_SLOTS) + index * pieces + k;
    const crown_v4 v = *p;
    return make_float4(v.x, v.y, v.z, v.w);
}
// The same walk over four-child nodes: one 128-byte fetch decides about four boxes, which halves the chain of
// dependent fetches per ray.  The children that are hit are entered nearest first; the others wait on the stack,
// the farthest at the bottom.  Every valid tree gives the same closest hit, so the result equals the binary walk's.
template <int STRIDE, bool COUNT, bool CROWN = false>
DEV void inner_step_wide(const Scene& S, Walk& w, float tmin, const Stack& stk, uint32_t* overflow_flag, TraceCount* cnt) {
    {
        float4 q0, q1, q2, q3, q4, q5, q6;
        const uint32_t ci = w.cur - CROWN_BASE;
        if (CROWN && ci <= CROWN_MAX) {
            q0 = crown_piece(ci, 8u, 0u), q1 = crown_piece(ci, 8u, 1u), q2 = crown_piece(ci, 8u, 2u), q3 = crown_piece(ci, 8u, 3u);
            q4 = crown_piece(ci, 8u, 4u), q5 = crown_piece(ci, 8u, 5u), q6 = crown_piece(ci, 8u, 6u);
        } else {
            const float4* np = reinterpret_cast<const float4*>(S.wide + w.cur);
            q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3], q4 = np[4], q5 = np[5], q6 = np[6];
#ifdef RT_EXTRA_LOAD  // timing experiment: one more 16-byte load per node (the record's unused tail), its value consumed
            {
                const float4 q7 = np[7];
                asm volatile("" ::"v"(q7.x), "v"(q7.y), "v"(q7.z), "v"(q7.w));
            }
#endif
        }
        if (COUNT) {
            cnt->nodes++;
            if (first_active_lane()) cnt->inner_waves++;
        }
        float k0, k1, k2, k3;
        uint32_t r0 = __float_as_uint(q6.x), r1 = __float_as_uint(q6.y), r2 = __float_as_uint(q6.z), r3 = __float_as_uint(q6.w);
        const bool h0 = slab_fma(mk3(q0.x, q0.y, q0.z), mk3(q0.w, q1.x, q1.y), w, tmin, w.best.t, k0) && r0 != REF_EMPTY;
        const bool h1 = slab_fma(mk3(q1.z, q1.w, q2.x), mk3(q2.y, q2.z, q2.w), w, tmin, w.best.t, k1) && r1 != REF_EMPTY;
        const bool h2 = slab_fma(mk3(q3.x, q3.y, q3.z), mk3(q3.w, q4.x, q4.y), w, tmin, w.best.t, k2) && r2 != REF_EMPTY;
        const bool h3 = slab_fma(mk3(q4.z, q4.w, q5.x), mk3(q5.y, q5.z, q5.w), w, tmin, w.best.t, k3) && r3 != REF_EMPTY;
        // a box that is hit has a finite entry distance (or -inf), so +inf sorts the misses behind every hit
        k0 = h0 ? k0 : INF_F;
        k1 = h1 ? k1 : INF_F;
        k2 = h2 ? k2 : INF_F;
        k3 = h3 ? k3 : INF_F;
        r0 = h0 ? r0 : REF_EMPTY;
        r1 = h1 ? r1 : REF_EMPTY;
        r2 = h2 ? r2 : REF_EMPTY;
        r3 = h3 ? r3 : REF_EMPTY;
        near_first(k0, r0, k1, r1);
        near_first(k2, r2, k3, r3);
        near_first(k0, r0, k2, r2);
        near_first(k1, r1, k3, r3);
        near_first(k1, r1, k2, r2);
        if (r0 != REF_EMPTY) {
            w.cur = r0;
            const int n_push = (r1 != REF_EMPTY) + (r2 != REF_EMPTY) + (r3 != REF_EMPTY);
            if (n_push) {
                if (w.sp + n_push <= (int)S.stack_cap) {
                    if (r3 != REF_EMPTY) stack_put<STRIDE>(stk, w.sp++, r3);
                    if (r2 != REF_EMPTY) stack_put<STRIDE>(stk, w.sp++, r2);
                    stack_put<STRIDE>(stk, w.sp++, r1);
                } else {
                    *overflow_flag = 1u;
                }
            }
        } else if (w.sp > 0) {
            w.sp--;
            w.cur = stack_get<STRIDE>(stk, w.sp);
        } else {
            w.cur = REF_EMPTY;
        }
    }
}
// WIDTH: 2 / 4 = the kernel is compiled for the binary / the four-child walk only (the persistent trace kernel: the
// binary walk alone needs 64 instead of 72 registers, one more resident wave per SIMD); 0 = whichever the scene holds.
template <int STRIDE, bool COUNT, int WIDTH = 0, bool CROWN = false>
DEV void inner_step(const Scene& S, Walk& w, float tmin, const Stack& stk, uint32_t* overflow_flag, TraceCount* cnt) {
    if (WIDTH == 4 || (WIDTH == 0 && S.wide)) {  // (the same in every lane)
        inner_step_wide<STRIDE, COUNT, CROWN>(S, w, tmin, stk, overflow_flag, cnt);
        return;
    }
    float4 q0, q1, q2, q3;
    const uint32_t ci = w.cur - CROWN_BASE;
    if (CROWN && ci <= CROWN_MAX) {
        q0 = crown_piece(ci, 4u, 0u), q1 = crown_piece(ci, 4u, 1u), q2 = crown_piece(ci, 4u, 2u), q3 = crown_piece(ci, 4u, 3u);
    } else {
        const float4* np = reinterpret_cast<const float4*>(S.nodes + w.cur);
        q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
#ifdef RT_EXTRA_LOAD  // timing experiment: one more 16-byte load per node (its first piece again), its value consumed
        {
            const float4* again = np;
            asm volatile("" : "+v"(again));
            const float4 q7 = again[0];
            asm volatile("" ::"v"(q7.x), "v"(q7.y), "v"(q7.z), "v"(q7.w));
        }
#endif
    }
    if (COUNT) {
        cnt->nodes++;
        if (first_active_lane()) cnt->inner_waves++;
    }
    float tn0, tn1;
    bool h0 = slab_fma(mk3(q0.x, q0.y, q0.z), mk3(q0.w, q1.x, q1.y), w, tmin, w.best.t, tn0);
    bool h1 = slab_fma(mk3(q1.z, q1.w, q2.x), mk3(q2.y, q2.z, q2.w), w, tmin, w.best.t, tn1);
#ifdef RT_INNER_TWICE  // timing experiment: both box tests a second time (same result)
    {
        asm volatile("" : "+v"(q0.x), "+v"(q1.z));
        float u0, u1;
        const bool g0 = slab_fma(mk3(q0.x, q0.y, q0.z), mk3(q0.w, q1.x, q1.y), w, tmin, w.best.t, u0);
        const bool g1 = slab_fma(mk3(q1.z, q1.w, q2.x), mk3(q2.y, q2.z, q2.w), w, tmin, w.best.t, u1);
        h0 = h0 && g0;
        h1 = h1 && g1;
        tn0 = fminf(tn0, u0);
        tn1 = fminf(tn1, u1);
    }
#endif
    uint32_t c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y);
    h0 = h0 && (c0 != REF_EMPTY);
    h1 = h1 && (c1 != REF_EMPTY);
    if (h0 || h1) {
        bool first1 = h1 && (!h0 || tn1 < tn0);
        w.cur = first1 ? c1 : c0;
        if (h0 && h1) {
            if (w.sp < (int)S.stack_cap) {
                stack_put<STRIDE>(stk, w.sp, first1 ? c0 : c1);
                w.sp++;
            } else {
                *overflow_flag = 1u;
            }
        }
    } else if (w.sp > 0) {
        w.sp--;
        w.cur = stack_get<STRIDE>(stk, w.sp);
    } else {
        w.cur = REF_EMPTY;
    }
}
// Walks inner nodes only (at most max_steps of them), until cur is a leaf, an instance, a sentinel or empty.
template <int STRIDE, bool COUNT, int WIDTH = 0, bool CROWN = false>
DEV void walk_inner(const Scene& S, Walk& w, float tmin, const Stack& stk, uint32_t* overflow_flag, TraceCount* cnt, int max_steps) {
    for (int step = 0; step < max_steps && (w.cur >> 30) == REF_INNER; step++)
        inner_step<STRIDE, COUNT, WIDTH, CROWN>(S, w, tmin, stk, overflow_flag, cnt);
}
// One non-inner step: tests a leaf's primitives, enters an instance or leaves it, then takes the next entry
// from the stack.  No-op when cur is an inner node or empty.
// CROWN: an instance whose root is in the crown is entered there.
template <int STRIDE, int GROUPS, bool COUNT, bool CROWN = false>
DEV void walk_leaf(const Scene& S, Walk& w, f3 wo, f3 wd, float time, float tmin, const Stack& stk, uint32_t* overflow_flag,
                   TraceCount* cnt) {
    const uint32_t type = w.cur >> 30;
    if (w.cur == REF_EMPTY || type == REF_INNER) return;
    if (COUNT) {
        cnt->leaf_lanes++;
        if (first_active_lane()) cnt->leaf_waves++;
    }
    if (type == REF_PRIMS) {
        uint32_t first = w.cur & 0x03FFFFFFu, n_leaf = ((w.cur >> 26) & 15u) + 1u;
        for (uint32_t i = 0; i < n_leaf; i++) test_prim<false>(S, first + i, w.o, w.d, w.a, time, tmin, w.grp, w.best);
#ifdef RT_LEAF_TWICE  // timing experiment: the leaf tests a second time (same result), to price them at their present lane use
        {
            uint32_t f2 = first;
            asm volatile("" : "+v"(f2), "+v"(w.best.t));
            for (uint32_t i = 0; i < n_leaf; i++) test_prim<false>(S, f2 + i, w.o, w.d, w.a, time, tmin, w.grp, w.best);
        }
#endif
        if (COUNT) cnt->prims += n_leaf;
        w.cur = REF_EMPTY;
    } else if (GROUPS && type == REF_INSTANCE) {
        uint32_t g = w.cur & 0x3FFFFFFFu;
        const Group& G = S.groups[g];
        uint32_t root = G.root_ref;
        if (CROWN && g < CROWN_INST_SLOTS) {
            const uint32_t in_crown = crown_words()[g];
            root = in_crown != REF_EMPTY ? in_crown : root;
        }
        w.cur = REF_EMPTY;
        if (root != REF_EMPTY) {
            if (w.sp < (int)S.stack_cap) {
                // shapes.rs:66-70; an instance inside an instance (kernels compiled with GROUPS == 2) is handed the ray
                // of the one around it
                if (G.has_xform) {
                    if (GROUPS == 2 && w.grp >= 0)
                        walk_set_ray(w, xf_point(G.minv, w.o), xf_vector(G.minv, w.d));
                    else
                        walk_set_ray(w, xf_point(G.minv, wo), xf_vector(G.minv, wd));
                }
                stack_put<STRIDE>(stk, w.sp, REF_SENTINEL);
                w.sp++;
                w.grp = (int)g;
                w.cur = root;
            } else {
                *overflow_flag = 1u;
            }
        }
    } else if (GROUPS) {  // REF_SENTINEL: the instance's subtree is finished, back to the ray of whatever is around it
        int parent = -1;
        f3 o = wo, d = wd;
        if (GROUPS == 2) {
            parent = S.groups[w.grp].parent;
            if (parent >= 0) instance_ray(S, parent, o, d);
        }
        walk_set_ray(w, o, d);
        w.grp = parent;
        w.cur = REF_EMPTY;
    }
    if (w.cur == REF_EMPTY && w.sp > 0) {  // next entry from the stack
        w.sp--;
        w.cur = stack_get<STRIDE>(stk, w.sp);
    }
}
// ConstantMedium objects come after all surfaces, in creation order (the order of a Scene whose media were
// added last), and override the closest surface hit unchecked (scene.rs:33-37).
template <bool COUNT, bool NESTED>
DEV void apply_media(const Scene& S, f3 wo, f3 wd, float time, float tmin, uint32_t k0, uint32_t k1, uint32_t bounce, Hit& best,
                     TraceCount* cnt) {
    if (!ray_is_finite(wo, wd)) return;  // (see ray_is_finite)
    for (uint32_t m = 0; m < S.n_media; m++) {
        test_medium<NESTED>(S, S.media[m], wo, wd, time, tmin, k0, k1, bounce, best);
        if (COUNT) cnt->prims += 2u * medium_boundary_count(S.groups[S.media[m]]);  // two boundary queries
    }
}
// Closest surface hit of one ray in a flat scene (Scene::flat): the world primitives, then each instance's
// primitives with the ray taken into its object space (shapes.rs:66-70) -- Scene::intersect (scene.rs:27-41)
// as written, no tree.  Loop counters and addresses are the same in every lane, so the primitive records come
// through the scalar cache and the wave never splits over which node to visit next.  The closest hit with the
// "higher id wins a tie" rule does not depend on the test order, so the result equals the tree walk's.
template <bool GROUPS, bool COUNT>
DEV void trace_flat(const Scene& S, f3 wo, f3 wd, float time, float tmin, float tmax, Hit& best, TraceCount* cnt) {
    best.t = tmax;
    best.prim = HIT_NONE;
    best.grp = -1;
    best.key = 0;
    if (!ray_is_finite(wo, wd)) return;  // (hits nothing, as in the tree walk)
    const float a = dot3(wd, wd);
    for (uint32_t i = 0; i < S.world_count; i++) test_prim<true>(S, S.world_first + i, wo, wd, a, time, tmin, -1, best);
    if (COUNT) cnt->prims += S.world_count;
    if (GROUPS)
        for (uint32_t g = 0; g < S.n_groups; g++) {
            const Group& G = S.groups[g];
            if (G.kind != 1u || G.flat_range == FLAT_NONE) continue;  // media come later; empty instance
            f3 o = wo, d = wd;
            float ag = a;
            if (G.has_xform) {
                o = xf_point(G.minv, wo);
                d = xf_vector(G.minv, wd);
                ag = dot3(d, d);
            }
            const uint32_t first = G.flat_range & 0x03FFFFFFu, n = (G.flat_range >> 26) + 1u;
            for (uint32_t i = 0; i < n; i++) test_prim<true>(S, first + i, o, d, ag, time, tmin, (int)g, best);
            if (COUNT) cnt->prims += n;
        }
}
// Closest hit of one ray: "while-while" -- the inner loop walks inner nodes only, so that lanes of a wave that
// are still descending do not drag the (long) primitive tests of the others along on every iteration.
// FLAT: the scene is traced without a tree (Scene::flat); a template parameter, so that neither form carries the
// registers of the other.
// MEDIA = false: surfaces only (the caller applies the media itself).
template <int STRIDE, int GROUPS, bool FLAT, bool COUNT = false, bool MEDIA = true>
DEV void trace_ray(const Scene& S, f3 wo, f3 wd, float time, float tmin, float tmax, uint32_t k0, uint32_t k1,
                   uint32_t bounce, Hit& best, const Stack& stk, uint32_t* overflow_flag, TraceCount* cnt = nullptr) {
    if (FLAT) {
        trace_flat<GROUPS != 0, COUNT>(S, wo, wd, time, tmin, tmax, best, cnt);
        if (GROUPS && MEDIA) apply_media<COUNT, GROUPS == 2>(S, wo, wd, time, tmin, k0, k1, bounce, best, cnt);
        return;
    }
    Walk w;
    walk_begin(S, w, wo, wd, tmax);
    while (!walk_done(w)) {
        walk_inner<STRIDE, COUNT>(S, w, tmin, stk, overflow_flag, cnt, 0x7fffffff);
        walk_leaf<STRIDE, GROUPS, COUNT>(S, w, wo, wd, time, tmin, stk, overflow_flag, cnt);
    }
    best = w.best;
    if (GROUPS && MEDIA) apply_media<COUNT, GROUPS == 2>(S, wo, wd, time, tmin, k0, k1, bounce, best, cnt);
}

// ---- hit record (hit_record.rs, shapes.rs) --------------------------------------
struct Record {
    f3 point, normal;
    float u, v;
    bool front;
    uint32_t mat;
};
// shapes.rs:74-88 at every instance from grp outwards, innermost first: normal through M (not its inverse transpose),
// point through M, then face-set again against the ray that instance was handed (wo, wd: the world ray).
template <bool NESTED>
DEV void record_to_world(const Scene& S, int grp, f3 wo, f3 wd, f3& point, f3& normal, bool& front) {
    for (int a = grp; a >= 0; a = NESTED ? S.groups[a].parent : -1) {
        const Group& G = S.groups[a];
        if (!G.has_xform) continue;
        f3 nw = normalize3(xf_vector(G.m, normal));
        point = xf_point(G.m, point);
        f3 lo = wo, ld = wd;
        if (NESTED && G.parent >= 0) instance_ray(S, G.parent, lo, ld);
        front = dot3(ld, nw) < 0.0f;
        normal = front ? nw : -nw;
    }
}
DEV void sphere_uv(f3 p, float& u, float& v) {  // shapes.rs:19-24
    float theta = acosf(-p.y);
    float phi = atan2f(-p.z, p.x) + PI_F;
    u = phi / (2.0f * PI_F);
    v = theta / PI_F;
}
// NESTED: the scene has groups inside instances (Scene::nested); kernels are compiled for either case, so that the
// common one does not carry the loops over the levels.
template <bool WANT_UV, bool NESTED>
DEV void make_record(const Scene& S, const LdsTables& L, f3 wo, f3 wd, float time, const Hit& h, Record& rec) {
    if (h.prim & HIT_MEDIUM_BIT) {  // shapes.rs:690-697
        const Group& G = S.groups[h.prim & 0x3FFFFFFFu];
        f3 bo = wo, bd = wd;  // the ray the medium was handed
        const bool inside = NESTED && G.parent >= 0;
        if (inside) instance_ray(S, G.parent, bo, bd);
        rec.point = bo + (h.t * bd);
        rec.normal = {1.0f, 0.0f, 0.0f};
        rec.front = true;
        rec.u = 0.0f;
        rec.v = 0.0f;
        rec.mat = G.mat;
        if (inside) record_to_world<NESTED>(S, G.parent, wo, wd, rec.point, rec.normal, rec.front);
        return;
    }
    f3 o = wo, d = wd;
    const Group* G = nullptr;  // (scenes without nesting) the one instance around the primitive, if it has a transform
    if (h.grp >= 0) {
        if (NESTED) {
            instance_ray(S, h.grp, o, d);
        } else {
            G = S.groups + h.grp;
            if (G->has_xform) {
                o = xf_point(G->minv, wo);
                d = xf_vector(G->minv, wd);
            } else {
                G = nullptr;
            }
        }
    }
    float4 A, B;
    load_record(L, L.prims, S.prims, h.prim, A, B);
    uint32_t info = __float_as_uint(B.y);
    uint32_t kind = info & 7u;
    rec.mat = info >> 8;
    if (kind == PK_BOX) {  // the side that was hit, as the rect it stands for (Hit::key is its id)
        float k, min0, min1, max0, max1;
        box_side_rect(A, B, h.key - __float_as_uint(B.z), kind, k, min0, min1, max0, max1);
        A = make_float4(k, min0, min1, max0);
        B.x = max1;
    }
    f3 point = o + (h.t * d);
    f3 outward;
    rec.u = 0.0f;
    rec.v = 0.0f;
    if (kind <= PK_MOVING_SPHERE) {  // shapes.rs:141-153
        f3 c{A.x, A.y, A.z};
        if (kind == PK_MOVING_SPHERE) c = moving_center(S, c, __float_as_uint(B.w), time);
        outward = (point - c) / A.w;
        if (WANT_UV) sphere_uv(outward, rec.u, rec.v);
    } else {  // shapes.rs:294-303
        int ax = kind == PK_RECT_XY ? 2 : (kind == PK_RECT_XZ ? 1 : 0);
        int a0 = ax == 0 ? 1 : 0;
        int a1 = ax == 2 ? 1 : 2;
        outward = {ax == 0 ? 1.0f : 0.0f, ax == 1 ? 1.0f : 0.0f, ax == 2 ? 1.0f : 0.0f};
        if (WANT_UV) {
            rec.u = (comp(point, a0) - A.y) / (A.w - A.y);
            rec.v = (comp(point, a1) - A.z) / (B.x - A.z);
        }
    }
    bool front = dot3(d, outward) < 0.0f;  // hit_record.rs:32-42
    f3 normal = front ? outward : -outward;
    if (NESTED) {
        record_to_world<NESTED>(S, h.grp, wo, wd, point, normal, front);
    } else if (G) {  // shapes.rs:74-88: normal through M (not its inverse transpose), then face-set again
        f3 nw = normalize3(xf_vector(G->m, normal));
        point = xf_point(G->m, point);
        front = dot3(wd, nw) < 0.0f;
        normal = front ? nw : -nw;
    }
    rec.point = point;
    rec.normal = normal;
    rec.front = front;
}

// ---- textures (texture.rs, perlin.rs) -------------------------------------------
// The noise value only feeds a colour (no hit decision hangs on it), so it is evaluated in the cheapest
// accurate form rather than in the reference's operation order: the eight-corner weighted sum of
// perlin.rs:71-85 is the trilinear interpolation of the corner dot products with the Hermite weights, here
// written as seven fused lerps.  Agreement with the reference order is at the 1e-7 level.
template <class Tab>
DEV float perlin_corner(const Tab& P, int h, float du, float dv, float dw) {
    // (component loads: the table may live in LDS, whose address space a vector cast would drop)
    const float gx = P.vec[h & 255][0], gy = P.vec[h & 255][1], gz = P.vec[h & 255][2];
    return __builtin_fmaf(gz, dw, __builtin_fmaf(gy, dv, gx * du));
}
template <class Tab>
DEV float perlin_noise(const Tab& P, f3 p) {  // perlin.rs:46-88
    float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    int i = (int)fx, j = (int)fy, k = (int)fz;
    float uu = u * u * __builtin_fmaf(-2.0f, u, 3.0f);
    float vv = v * v * __builtin_fmaf(-2.0f, v, 3.0f);
    float ww = w * w * __builtin_fmaf(-2.0f, w, 3.0f);
    const int x0 = P.perm[0][(uint32_t)i & 255u], x1 = P.perm[0][(uint32_t)(i + 1) & 255u];
    const int y0 = P.perm[1][(uint32_t)j & 255u], y1 = P.perm[1][(uint32_t)(j + 1) & 255u];
    const int z0 = P.perm[2][(uint32_t)k & 255u], z1 = P.perm[2][(uint32_t)(k + 1) & 255u];
    const float u1 = u - 1.0f, v1 = v - 1.0f, w1 = w - 1.0f;
    const int h00 = x0 ^ y0, h01 = x0 ^ y1, h10 = x1 ^ y0, h11 = x1 ^ y1;
    float a, b;
    a = perlin_corner(P, h00 ^ z0, u, v, w);
    b = perlin_corner(P, h00 ^ z1, u, v, w1);
    const float c00 = __builtin_fmaf(ww, b - a, a);
    a = perlin_corner(P, h01 ^ z0, u, v1, w);
    b = perlin_corner(P, h01 ^ z1, u, v1, w1);
    const float c01 = __builtin_fmaf(ww, b - a, a);
    a = perlin_corner(P, h10 ^ z0, u1, v, w);
    b = perlin_corner(P, h10 ^ z1, u1, v, w1);
    const float c10 = __builtin_fmaf(ww, b - a, a);
    a = perlin_corner(P, h11 ^ z0, u1, v1, w);
    b = perlin_corner(P, h11 ^ z1, u1, v1, w1);
    const float c11 = __builtin_fmaf(ww, b - a, a);
    const float c0 = __builtin_fmaf(vv, c01 - c00, c00);
    const float c1 = __builtin_fmaf(vv, c11 - c10, c10);
    return __builtin_fmaf(uu, c1 - c0, c0);
}
template <class Tab>
DEV float perlin_turb(const Tab& P, f3 p, int depth) {  // perlin.rs:90-102
    float acc = 0.0f;
    f3 temp = p;
    float weight = 1.0f;
#pragma unroll 1
    for (int i = 0; i < depth; i++) {
        acc = __builtin_fmaf(weight, perlin_noise(P, temp), acc);
        weight *= 0.5f;
        temp = 2.0f * temp;
    }
    return __builtin_fabsf(acc);
}
// sin(x) to ~1e-6 absolute for any finite f32 x: the turn count x/2pi is reduced in f64 (exact to 1e-16 of a
// turn), the hardware sine takes the remaining fraction of a turn.  Used for colours only.
DEV float sin_turns(float x) {
    double t = (double)x * 0.15915494309189533577;
    t -= __builtin_rint(t);
    return __builtin_amdgcn_sinf((float)t);
}
DEV float rust_clamp(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
DEV uint32_t rust_as_u32(float f) {  // saturating, NaN -> 0
    if (!(f == f)) return 0u;
    if (f <= 0.0f) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}
// Sign of sin(x) without evaluating it: x lies in (k*pi, (k+1)*pi) with k = floor(x/pi) taken in f64 (exact
// enough for every f32 x), and sin is negative there iff k is odd, i.e. iff frac(x/2pi) >= 1/2 (the halving
// is exact).  sin(x) is zero only for x == 0.
DEV bool sin_is_negative(float x) {
    double h = (double)x * (0.5 * 0.31830988618379067153776752674502872);
    return __builtin_amdgcn_fract(h) >= 0.5;
}
// `sin(10x)*sin(10y)*sin(10z) < 0` of CheckeredTexture::value (texture.rs:54-56): only the sign of the
// product is used, so three transcendentals reduce to three parities.  A zero or non-finite factor makes
// the product zero or NaN, for which `< 0` is false.
DEV bool checker_is_odd(f3 p) {
    float x = 10.0f * p.x, y = 10.0f * p.y, z = 10.0f * p.z;
    float lo = fminf(fminf(__builtin_fabsf(x), __builtin_fabsf(y)), __builtin_fabsf(z));
    float hi = fmaxf(fmaxf(__builtin_fabsf(x), __builtin_fabsf(y)), __builtin_fabsf(z));
    bool usable = lo > 0.0f && hi < INF_F;  // false as well when any factor is NaN
    bool neg = sin_is_negative(x) ^ sin_is_negative(y) ^ sin_is_negative(z);
    return usable && neg;
}
DEV f3 tex_value(const Scene& S, const LdsTables& L, uint32_t ti, float u, float v, f3 p) {
    float4 T0, T1;
    load_record(L, L.texs, S.texs, ti, T0, T1);
    uint32_t kind = __float_as_uint(T0.x) & 0xFFu;
    for (int it = 0; it < 8 && kind == TK_CHECKER; it++) {  // texture.rs:52-62
        ti = checker_is_odd(p) ? __float_as_uint(T1.x) : __float_as_uint(T1.y);
        load_record(L, L.texs, S.texs, ti, T0, T1);
        kind = __float_as_uint(T0.x) & 0xFFu;
    }
    if (kind == TK_NOISE) {  // texture.rs:78-82
        const uint32_t table = __float_as_uint(T1.w);
        float turb;
        if (L.has_perlin0 && table == 0u)
            turb = perlin_turb(*L.perlin0, p, 7);
        else
            turb = perlin_turb(S.perlin[table], p, 7);
        float s = 1.0f + sin_turns(__builtin_fmaf(10.0f, turb, T1.z * p.z));
        return {0.5f * s, 0.5f * s, 0.5f * s};
    }
    if (kind == TK_IMAGE) {  // texture.rs:96-117
        const Image im = S.images[__float_as_uint(T1.w)];
        float cu = rust_clamp(u, 0.0f, 1.0f);
        float cv = 1.0f - rust_clamp(v, 0.0f, 1.0f);
        uint32_t i = rust_as_u32(cu * (float)im.w);
        uint32_t j = rust_as_u32(cv * (float)im.h);
        if (i >= im.w) i = im.w - 1;
        if (j >= im.h) j = im.h - 1;
        const uint8_t* px = im.rgb + 3ull * ((unsigned long long)j * im.w + i);
        // x / 255 for x = 0..255 through the rounded reciprocal of 255 (0x3B808081): all 256 quotients equal the
        // IEEE division (checked exhaustively)
        const float r255 = 0x1.010102p-8f;
        return {div_by((float)px[0], 255.0f, r255), div_by((float)px[1], 255.0f, r255), div_by((float)px[2], 255.0f, r255)};
    }
    return {T0.y, T0.z, T0.w};  // texture.rs:18-22
}

// ---- materials (materials.rs, vec3_helpers.rs) ------------------------------------
DEV f3 reflect3(f3 v, f3 n) {  // vec3_helpers.rs:62-66
    float scalar = 2.0f * dot3(v, n);
    return v - scalar * n;
}
DEV f3 refract3(f3 uv, f3 n, float ratio) {  // vec3_helpers.rs:68-75
    float dot_p = dot3(n, -uv);
    float cos_theta = fminf(dot_p, 1.0f);
    f3 perp = ratio * (uv + cos_theta * n);
    f3 par = (-__builtin_sqrtf(__builtin_fabsf(1.0f - dot3(perp, perp)))) * n;
    return perp + par;
}
DEV float reflectance(float cosine, float ref_index) {  // materials.rs:109-114
    float r0 = (1.0f - ref_index) / (1.0f + ref_index);
    r0 = r0 * r0;
    float x = 1.0f - cosine, x2 = x * x;  // powf(x, 5.0): within 2 ulp of it, feeds one comparison with a random draw
    return r0 + (1.0f - r0) * ((x2 * x2) * x);
}
// Texture::value of the material's texture (albedo of Lambertian / Isotropic, emission of DiffuseLight); Metal
// carries its colour in the record.  The commit step folds SolidColor textures into the material record
// (MF_SOLID: the colour sits in albedo), which saves a dependent texture fetch.  Evaluated once per hit, ahead
// of the material switch, so that the texture code exists once in the kernel.
DEV f3 material_color(const Scene& S, const LdsTables& L, const Material& M, const Record& rec) {
    if ((M.pad & MF_SOLID) || M.kind == MK_METAL || M.kind == MK_DIELECTRIC) return ld3(M.albedo);
    return tex_value(S, L, M.tex, rec.u, rec.v, rec.point);
}
// returns true when the path continues; att and the scattered direction are set then.
// Lambertian, Metal and Isotropic all start by drawing random_in_unit_sphere() from the bounce's stream
// (vec3_helpers.rs:25-31 through materials.rs:47, 89, 206), and two of them normalise it.  A wave of scattered
// rays holds all three kinds at once, so the rejection loop (by far the longest part of a scatter) and the
// normalisation are done once for all of them ahead of the per-material code, not once per material.
DEV bool scatter(const Material& M, f3 color, f3 wd, const Record& rec, Rng& rng, f3& att, f3& dir) {
    const bool draws = M.kind == MK_LAMBERTIAN || M.kind == MK_METAL || M.kind == MK_ISOTROPIC;
    f3 ball{0.0f, 0.0f, 1.0f}, unit_ball{0.0f, 0.0f, 1.0f};
    if (draws) {
        ball = rand_in_unit_sphere(rng);
        if (M.kind != MK_METAL) unit_ball = normalize3(ball);
    }
    switch (M.kind) {
        case MK_LAMBERTIAN: {  // materials.rs:45-61
            f3 sc = rec.normal + unit_ball;
            if (__builtin_fabsf(sc.x) < F32_EPSILON && __builtin_fabsf(sc.y) < F32_EPSILON && __builtin_fabsf(sc.z) < F32_EPSILON)
                sc = rec.normal;
            dir = sc;
            att = color;
            return true;
        }
        case MK_METAL: {  // materials.rs:84-95 (draws even when fuzz == 0)
            f3 reflected = reflect3(normalize3(wd), rec.normal);
            dir = reflected + M.fuzz * ball;
            att = color;
            return dot3(dir, rec.normal) > 0.0f;
        }
        case MK_DIELECTRIC: {  // materials.rs:117-148
            att = {1.0f, 1.0f, 1.0f};
            float ratio = rec.front ? 1.0f / M.ior : M.ior;
            f3 unit = normalize3(wd);
            float cos_theta = fminf(dot3(rec.normal, -unit), 1.0f);
            float sin_theta = __builtin_sqrtf(1.0f - cos_theta * cos_theta);
            bool cannot_refract = ratio * sin_theta > 1.0f;
            if (cannot_refract || reflectance(cos_theta, ratio) > rng.f32())
                dir = reflect3(unit, rec.normal);
            else
                dir = refract3(unit, rec.normal, ratio);
            return true;
        }
        case MK_ISOTROPIC: {  // materials.rs:204-215
            dir = unit_ball;
            att = color;
            return true;
        }
        default:  // MK_LIGHT, materials.rs:176-179
            return false;
    }
}

DEV Material load_material(const Scene& S, const LdsTables& L, uint32_t mi) {
    float4 m0, m1;
    load_record(L, L.mats, S.mats, mi, m0, m1);
    Material M;
    M.kind = __float_as_uint(m0.x);
    M.tex = __float_as_uint(m0.y);
    M.albedo[0] = m0.z;
    M.albedo[1] = m0.w;
    M.albedo[2] = m1.x;
    M.fuzz = m1.y;
    M.ior = m1.z;
    M.pad = __float_as_uint(m1.w);
    return M;
}
// set at commit: does anything this material evaluates read (u, v)?  (only image textures do)
DEV bool material_needs_uv(const Material& M) { return (M.pad & MF_NEEDS_UV) != 0; }

DEV uint32_t lane_prefix(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}


}  // namespace
