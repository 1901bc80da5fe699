// C++ mirror of the reference's scene-construction API, sitting above the C ABI of include/rtamd.h.
//
// The reference is a Rust crate; no Rust toolchain exists in the build image, so the host side is
// C++.  Types, constructor names, argument order and meaning follow the Rust sources one to one
// (Rust `T::new(..)` is spelled `T::new_(..)` here because `new` is a C++ keyword).  Objects hold
// only what the device needs; `flatten` walks the object graph depth first in insertion order and
// fills the flat arrays of rt_scene_set_*.
//
//   Rust (raytracer-rs/src)                          here
//   shapes.rs:14-17    trait Hittable                 rt::Hittable
//   shapes.rs:107,178  Sphere::new, MovingSphere::new rt::Sphere::new_, rt::MovingSphere::new_
//   shapes.rs:260-404  XY/XZ/YZRect::new(_with_material)
//   shapes.rs:452,456  Box2::new, Box2::full_box
//   shapes.rs:34       TransformedObject::new
//   shapes.rs:639      ConstantMedium::from_color
//   bvh_node.rs:48     BVHNode::from_scene            (a grouping node: the device BVH is built at commit)
//   scene.rs:13-23     Scene::new/add_shape/clear
//   materials.rs       LambertianMat, MetalMat, DielectricMat, DiffuseLight, IsotropicMat
//   texture.rs         SolidColor, CheckeredTexture, NoiseTexture, ImageTexture
//   camera.rs:28       Camera::new
//   vec3_helpers.rs    random, random_range (scene builders use them)
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/rtamd.h"

namespace rt {

// ---- glam subset --------------------------------------------------------------------
struct Vec3A {
    float x = 0, y = 0, z = 0;
    Vec3A() {}
    Vec3A(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    static Vec3A ZERO() { return {0, 0, 0}; }
    static Vec3A ONE() { return {1, 1, 1}; }
    static Vec3A Y() { return {0, 1, 0}; }
    float dot(Vec3A o) const { return (x * o.x + y * o.y) + z * o.z; }
    float length() const { return std::sqrt(dot(*this)); }
    Vec3A normalize() const {
        float l = length();
        return {x / l, y / l, z / l};
    }
    Vec3A cross(Vec3A o) const { return {y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x}; }
};
using Vec3 = Vec3A;
inline Vec3A operator+(Vec3A a, Vec3A b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3A operator-(Vec3A a, Vec3A b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3A operator*(Vec3A a, Vec3A b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline Vec3A operator*(float s, Vec3A a) { return {s * a.x, s * a.y, s * a.z}; }
inline Vec3A operator*(Vec3A a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3A operator/(Vec3A a, float s) { return {a.x / s, a.y / s, a.z / s}; }
struct Vec2 {
    float x = 0, y = 0;
    Vec2() {}
    Vec2(float x_, float y_) : x(x_), y(y_) {}
};

// Mat4, column major (col[j][row])
struct Mat4 {
    float col[4][4];
    static Mat4 IDENTITY() {
        Mat4 m;
        std::memset(m.col, 0, sizeof(m.col));
        for (int i = 0; i < 4; i++) m.col[i][i] = 1.0f;
        return m;
    }
    static Mat4 from_rotation_y(float angle) {
        float s = std::sin(angle), c = std::cos(angle);
        Mat4 m = IDENTITY();
        m.col[0][0] = c;
        m.col[0][2] = -s;
        m.col[2][0] = s;
        m.col[2][2] = c;
        return m;
    }
    static Mat4 from_translation(Vec3 t) {
        Mat4 m = IDENTITY();
        m.col[3][0] = t.x;
        m.col[3][1] = t.y;
        m.col[3][2] = t.z;
        return m;
    }
    Mat4 operator*(const Mat4& o) const {
        Mat4 r;
        for (int j = 0; j < 4; j++)
            for (int i = 0; i < 4; i++) {
                float acc = col[0][i] * o.col[j][0];
                acc = acc + col[1][i] * o.col[j][1];
                acc = acc + col[2][i] * o.col[j][2];
                acc = acc + col[3][i] * o.col[j][3];
                r.col[j][i] = acc;
            }
        return r;
    }
    // general 4x4 inverse by cofactors, laid out as in GLM's compute_inverse (what glam's Mat4::inverse follows)
    Mat4 inverse() const {
        const float(*m)[4] = col;
        float a2323 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
        float a1323 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
        float a1223 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
        float b2313 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
        float b1313 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
        float b1213 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
        float c2312 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
        float c1312 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
        float c1212 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
        float d2303 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
        float d1303 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
        float d1203 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
        float e2302 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
        float e1302 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
        float e1202 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
        float f2301 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
        float f1301 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
        float f1201 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
        const float fa[4] = {a2323, a2323, a1323, a1223}, fb[4] = {b2313, b2313, b1313, b1213};
        const float fc[4] = {c2312, c2312, c1312, c1212}, fd[4] = {d2303, d2303, d1303, d1203};
        const float fe[4] = {e2302, e2302, e1302, e1202}, ff[4] = {f2301, f2301, f1301, f1201};
        const float v0[4] = {m[1][0], m[0][0], m[0][0], m[0][0]}, v1[4] = {m[1][1], m[0][1], m[0][1], m[0][1]};
        const float v2[4] = {m[1][2], m[0][2], m[0][2], m[0][2]}, v3[4] = {m[1][3], m[0][3], m[0][3], m[0][3]};
        Mat4 inv;
        for (int i = 0; i < 4; i++) {
            float sgn_a = (i & 1) ? -1.0f : 1.0f, sgn_b = -sgn_a;
            inv.col[0][i] = ((v1[i] * fa[i] - v2[i] * fb[i]) + v3[i] * fc[i]) * sgn_a;
            inv.col[1][i] = ((v0[i] * fa[i] - v2[i] * fd[i]) + v3[i] * fe[i]) * sgn_b;
            inv.col[2][i] = ((v0[i] * fb[i] - v1[i] * fd[i]) + v3[i] * ff[i]) * sgn_a;
            inv.col[3][i] = ((v0[i] * fc[i] - v1[i] * fe[i]) + v2[i] * ff[i]) * sgn_b;
        }
        float det = (m[0][0] * inv.col[0][0] + m[0][1] * inv.col[1][0]) + (m[0][2] * inv.col[2][0] + m[0][3] * inv.col[3][0]);
        float rcp = 1.0f / det;
        for (int j = 0; j < 4; j++)
            for (int i = 0; i < 4; i++) inv.col[j][i] = inv.col[j][i] * rcp;
        return inv;
    }
    // 3 rows x 4 columns, row major, as rt_group wants it
    void to_rows(float out[12]) const {
        for (int r = 0; r < 3; r++)
            for (int j = 0; j < 4; j++) out[4 * r + j] = col[j][r];
    }
};

// ---- thread_rng stand-in ----------------------------------------------------------------
// Counter-based stream (same construction as the render streams): draw n = mix32((k0+n)^k1).
// The scene stream of seed s has k0 = mix32(s^0xA511E9B3), k1 = mix32(s^0x2545F491).
inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
class ThreadRng {
   public:
    void seed(uint32_t s) {
        k0_ = mix32(s ^ 0xA511E9B3U);
        k1_ = mix32(s ^ 0x2545F491U);
        n_ = 0;
    }
    uint32_t next_u32() { return mix32((k0_ + n_++) ^ k1_); }
    float gen() { return (float)(next_u32() >> 8) * (1.0f / 16777216.0f); }  // rng.gen::<f32>()
    float gen_range(float lo, float hi) {                                      // rng.gen_range(lo..hi)
        float scale = hi - lo;
        for (;;) {
            uint32_t bits = 0x3f800000U | (next_u32() >> 9);
            float v12;
            std::memcpy(&v12, &bits, 4);
            float r = (v12 - 1.0f) * scale + lo;
            if (r < hi) return r;
        }
    }
    uint32_t gen_index(uint32_t n) { return (uint32_t)(((uint64_t)next_u32() * n) >> 32); }

   private:
    uint32_t k0_ = 0, k1_ = 0, n_ = 0;
};
inline ThreadRng& thread_rng() {
    static thread_local ThreadRng r;
    return r;
}
namespace vec3_helpers {
inline Vec3A random() {
    ThreadRng& r = thread_rng();
    float x = r.gen(), y = r.gen(), z = r.gen();
    return {x, y, z};
}
inline Vec3A random_range(float lo, float hi) {
    ThreadRng& r = thread_rng();
    float x = r.gen_range(lo, hi), y = r.gen_range(lo, hi), z = r.gen_range(lo, hi);
    return {x, y, z};
}
}  // namespace vec3_helpers

// ---- flat scene (what goes through the C ABI) ----------------------------------------------
struct FlatScene {
    std::vector<float> sph_c0r, sph_c1, sph_t01;
    std::vector<uint8_t> sph_moving;
    std::vector<uint32_t> sph_mat, sph_id, sph_group;
    std::vector<uint8_t> rect_axis;
    std::vector<float> rect_k, rect_bounds;
    std::vector<uint32_t> rect_mat, rect_id, rect_group;
    std::vector<rt_group> groups;
    std::vector<rt_material> mats;
    std::vector<rt_texture> texs;
    struct PerlinData {
        std::vector<float> vec;
        std::vector<int32_t> perm;
    };
    std::vector<PerlinData> perlin;
    struct ImageData {
        uint32_t w, h;
        std::vector<uint8_t> rgb;
    };
    std::vector<ImageData> images;
    uint32_t next_id = 0;
    uint32_t n_instances = 0, n_media = 0;
    bool any_moving = false;
    std::unordered_map<const void*, uint32_t> mat_index, tex_index;

    void add_sphere(Vec3A c0, float r, uint32_t mat, uint32_t group, bool moving, Vec3A c1, float t0, float t1) {
        sph_c0r.insert(sph_c0r.end(), {c0.x, c0.y, c0.z, r});
        sph_c1.insert(sph_c1.end(), {c1.x, c1.y, c1.z});
        sph_t01.insert(sph_t01.end(), {t0, t1});
        sph_moving.push_back(moving ? 1 : 0);
        any_moving = any_moving || moving;
        sph_mat.push_back(mat);
        sph_id.push_back(next_id++);
        sph_group.push_back(group);
    }
    void add_rect(uint8_t axis, Vec2 mn, Vec2 mx, float k, uint32_t mat, uint32_t group) {
        rect_axis.push_back(axis);
        rect_k.push_back(k);
        rect_bounds.insert(rect_bounds.end(), {mn.x, mn.y, mx.x, mx.y});
        rect_mat.push_back(mat);
        rect_id.push_back(next_id++);
        rect_group.push_back(group);
    }
    // sends everything through the C ABI and commits (BVHNode::from_scene(&world, t0, t1))
    int upload(rt_ctx* ctx, float t0, float t1) const {
        int rc;
        if ((rc = rt_scene_clear(ctx))) return rc;
        if ((rc = rt_scene_set_textures(ctx, (uint32_t)texs.size(), texs.data()))) return rc;
        if ((rc = rt_scene_set_materials(ctx, (uint32_t)mats.size(), mats.data()))) return rc;
        for (size_t i = 0; i < perlin.size(); i++)
            if ((rc = rt_scene_set_perlin(ctx, (uint32_t)i, perlin[i].vec.data(), perlin[i].perm.data()))) return rc;
        for (size_t i = 0; i < images.size(); i++)
            if ((rc = rt_scene_set_image(ctx, (uint32_t)i, images[i].w, images[i].h, images[i].rgb.data()))) return rc;
        if ((rc = rt_scene_set_groups(ctx, (uint32_t)groups.size(), groups.data()))) return rc;
        if ((rc = rt_scene_set_spheres(ctx, (uint32_t)sph_id.size(), sph_c0r.data(), any_moving ? sph_c1.data() : nullptr,
                                       any_moving ? sph_t01.data() : nullptr, any_moving ? sph_moving.data() : nullptr,
                                       sph_mat.data(), sph_id.data(), sph_group.data())))
            return rc;
        if ((rc = rt_scene_set_rects(ctx, (uint32_t)rect_id.size(), rect_axis.data(), rect_k.data(), rect_bounds.data(),
                                     rect_mat.data(), rect_id.data(), rect_group.data())))
            return rc;
        return rt_scene_commit(ctx, t0, t1);
    }
};

// ---- textures (texture.rs) -------------------------------------------------------------------
struct Texture {
    virtual ~Texture() {}
    virtual uint32_t emit(FlatScene& fs) const = 0;
    uint32_t flatten(FlatScene& fs) const {
        auto it = fs.tex_index.find(this);
        if (it != fs.tex_index.end()) return it->second;
        uint32_t idx = emit(fs);
        fs.tex_index[this] = idx;
        return idx;
    }
};
using TexturePtr = std::shared_ptr<const Texture>;
struct SolidColor : Texture {
    Vec3A color;
    explicit SolidColor(Vec3A c) : color(c) {}
    uint32_t emit(FlatScene& fs) const override {
        rt_texture t{};
        t.kind = RT_TEX_SOLID;
        t.color[0] = color.x;
        t.color[1] = color.y;
        t.color[2] = color.z;
        fs.texs.push_back(t);
        return (uint32_t)fs.texs.size() - 1;
    }
};
struct CheckeredTexture : Texture {
    TexturePtr odd, even;
    CheckeredTexture(TexturePtr o, TexturePtr e) : odd(std::move(o)), even(std::move(e)) {}
    template <class S, class T>
    static CheckeredTexture from_texture(S o, T e) {
        return CheckeredTexture(std::make_shared<S>(std::move(o)), std::make_shared<T>(std::move(e)));
    }
    static CheckeredTexture from_shared_texture(TexturePtr o, TexturePtr e) { return CheckeredTexture(std::move(o), std::move(e)); }
    static CheckeredTexture from_color(Vec3A o, Vec3A e) {
        return CheckeredTexture(std::make_shared<SolidColor>(o), std::make_shared<SolidColor>(e));
    }
    uint32_t emit(FlatScene& fs) const override {
        uint32_t o = odd->flatten(fs), e = even->flatten(fs);
        rt_texture t{};
        t.kind = RT_TEX_CHECKER;
        t.odd = o;
        t.even = e;
        fs.texs.push_back(t);
        return (uint32_t)fs.texs.size() - 1;
    }
};
struct Perlin {  // perlin.rs:19-44
    std::vector<float> rand_vec;  // 256 x 3
    std::vector<int32_t> perm;    // x | y | z
    static Perlin new_() {
        Perlin p;
        p.rand_vec.reserve(768);
        for (int i = 0; i < 256; i++) {
            Vec3A v = vec3_helpers::random_range(-1.0f, 1.0f);
            p.rand_vec.insert(p.rand_vec.end(), {v.x, v.y, v.z});
        }
        for (int a = 0; a < 3; a++) {
            std::vector<int32_t> q(256);
            for (int i = 0; i < 256; i++) q[i] = i;
            ThreadRng& r = thread_rng();
            for (int i = 255; i >= 1; i--) std::swap(q[i], q[r.gen_index((uint32_t)i + 1)]);  // SliceRandom::shuffle
            p.perm.insert(p.perm.end(), q.begin(), q.end());
        }
        return p;
    }
};
struct NoiseTexture : Texture {
    Perlin noise;
    float frequency;
    NoiseTexture(Perlin p, float f) : noise(std::move(p)), frequency(f) {}
    static NoiseTexture new_(float freq) { return NoiseTexture(Perlin::new_(), freq); }
    uint32_t emit(FlatScene& fs) const override {
        fs.perlin.push_back({noise.rand_vec, noise.perm});
        rt_texture t{};
        t.kind = RT_TEX_NOISE;
        t.freq = frequency;
        t.table = (uint32_t)fs.perlin.size() - 1;
        fs.texs.push_back(t);
        return (uint32_t)fs.texs.size() - 1;
    }
};
struct ImageTexture : Texture {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgb;
    // texture.rs:89-93 decodes any format the `image` crate knows and panics on failure; rt_image_read knows
    // baseline JPEG (what the reference's earthmap.jpg is) and binary PPM, and this throws instead of panicking.
    static ImageTexture new_(const std::string& path) {
        ImageTexture t;
        int rc = rt_image_read(path.c_str(), &t.width, &t.height, nullptr, 0);
        if (rc == RT_OK) {
            t.rgb.resize((size_t)t.width * t.height * 3);
            rc = rt_image_read(path.c_str(), &t.width, &t.height, t.rgb.data(), t.rgb.size());
        }
        if (rc == RT_ERR_IO) throw std::runtime_error("ImageTexture: cannot open " + path);
        if (rc != RT_OK) throw std::runtime_error("ImageTexture: neither a baseline JPEG nor a binary 8-bit PPM: " + path);
        return t;
    }
    uint32_t emit(FlatScene& fs) const override {
        fs.images.push_back({width, height, rgb});
        rt_texture t{};
        t.kind = RT_TEX_IMAGE;
        t.table = (uint32_t)fs.images.size() - 1;
        fs.texs.push_back(t);
        return (uint32_t)fs.texs.size() - 1;
    }
};

// ---- materials (materials.rs) -------------------------------------------------------------------
struct Material {
    virtual ~Material() {}
    virtual uint32_t emit(FlatScene& fs) const = 0;
    uint32_t flatten(FlatScene& fs) const {
        auto it = fs.mat_index.find(this);
        if (it != fs.mat_index.end()) return it->second;
        uint32_t idx = emit(fs);
        fs.mat_index[this] = idx;
        return idx;
    }
};
using MaterialPtr = std::shared_ptr<const Material>;
struct TexturedMat : Material {
    uint32_t kind;
    TexturePtr tex;
    TexturedMat(uint32_t k, TexturePtr t) : kind(k), tex(std::move(t)) {}
    uint32_t emit(FlatScene& fs) const override {
        rt_material m{};
        m.kind = kind;
        m.tex = tex->flatten(fs);
        fs.mats.push_back(m);
        return (uint32_t)fs.mats.size() - 1;
    }
};
struct LambertianMat : TexturedMat {
    explicit LambertianMat(TexturePtr t) : TexturedMat(RT_MAT_LAMBERTIAN, std::move(t)) {}
    template <class T>
    static LambertianMat from_texture(T albedo) { return LambertianMat(std::make_shared<T>(std::move(albedo))); }
    static LambertianMat from_shared_texture(TexturePtr albedo) { return LambertianMat(std::move(albedo)); }
    static LambertianMat from_color(Vec3A albedo) { return LambertianMat(std::make_shared<SolidColor>(albedo)); }
};
struct DiffuseLight : TexturedMat {
    explicit DiffuseLight(TexturePtr t) : TexturedMat(RT_MAT_DIFFUSE_LIGHT, std::move(t)) {}
    template <class T>
    static DiffuseLight from_texture(T tex) { return DiffuseLight(std::make_shared<T>(std::move(tex))); }
    static DiffuseLight from_color(Vec3A c) { return DiffuseLight(std::make_shared<SolidColor>(c)); }
};
struct IsotropicMat : TexturedMat {
    explicit IsotropicMat(TexturePtr t) : TexturedMat(RT_MAT_ISOTROPIC, std::move(t)) {}
    template <class T>
    static IsotropicMat from_texture(T tex) { return IsotropicMat(std::make_shared<T>(std::move(tex))); }
    static IsotropicMat from_color(Vec3A c) { return IsotropicMat(std::make_shared<SolidColor>(c)); }
};
struct MetalMat : Material {
    Vec3A albedo;
    float fuzz;
    MetalMat(Vec3A a, float f) : albedo(a), fuzz(f) {}
    static MetalMat new_(Vec3A albedo, float fuzz) { return MetalMat(albedo, fuzz < 1.0f ? fuzz : 1.0f); }  // materials.rs:70-81
    uint32_t emit(FlatScene& fs) const override {
        rt_material m{};
        m.kind = RT_MAT_METAL;
        m.albedo[0] = albedo.x;
        m.albedo[1] = albedo.y;
        m.albedo[2] = albedo.z;
        m.fuzz = fuzz;
        fs.mats.push_back(m);
        return (uint32_t)fs.mats.size() - 1;
    }
};
struct DielectricMat : Material {
    float index_refraction;
    explicit DielectricMat(float i) : index_refraction(i) {}
    static DielectricMat new_(float index) { return DielectricMat(index); }
    uint32_t emit(FlatScene& fs) const override {
        rt_material m{};
        m.kind = RT_MAT_DIELECTRIC;
        m.ior = index_refraction;
        fs.mats.push_back(m);
        return (uint32_t)fs.mats.size() - 1;
    }
};

// ---- hittables (shapes.rs, scene.rs, bvh_node.rs) ---------------------------------------------------
struct Hittable {
    virtual ~Hittable() {}
    // group: 0 = world, otherwise index+1 of the rt_group the primitives belong to
    virtual void flatten(FlatScene& fs, uint32_t group) const = 0;
    // TransformedObject reports its matrices so that a ConstantMedium can take them over
    virtual bool take_xform(float /*m*/[12], float /*minv*/[12], const Hittable** /*inner*/) const { return false; }
};
using HittablePtr = std::shared_ptr<const Hittable>;

struct Sphere : Hittable {
    Vec3A center;
    float radius;
    MaterialPtr material;
    Sphere(Vec3A c, float r, MaterialPtr m) : center(c), radius(r), material(std::move(m)) {}
    template <class M>
    static Sphere new_(Vec3A center, float radius, M material) {
        return Sphere(center, radius, std::make_shared<M>(std::move(material)));
    }
    void flatten(FlatScene& fs, uint32_t group) const override {
        fs.add_sphere(center, radius, material->flatten(fs), group, false, Vec3A(), 0, 0);
    }
};
struct MovingSphere : Hittable {
    Vec3A center_start, center_end;
    float radius, time_start, time_end;
    MaterialPtr material;
    MovingSphere(Vec3A c0, Vec3A c1, float r, float t0, float t1, MaterialPtr m)
        : center_start(c0), center_end(c1), radius(r), time_start(t0), time_end(t1), material(std::move(m)) {}
    template <class M>
    static MovingSphere new_(Vec3A c0, Vec3A c1, float radius, float t0, float t1, M material) {
        return MovingSphere(c0, c1, radius, t0, t1, std::make_shared<M>(std::move(material)));
    }
    void flatten(FlatScene& fs, uint32_t group) const override {
        fs.add_sphere(center_start, radius, material->flatten(fs), group, true, center_end, time_start, time_end);
    }
};
template <int AXIS>  // 0 = XY, 1 = XZ, 2 = YZ (the rt_scene_set_rects encoding)
struct AxisRect : Hittable {
    Vec2 min, max;
    float offset;
    MaterialPtr material;
    AxisRect(Vec2 mn, Vec2 mx, float off, MaterialPtr m) : min(mn), max(mx), offset(off), material(std::move(m)) {}
    template <class M>
    static AxisRect new_(Vec2 mn, Vec2 mx, float offset, M material) {
        return AxisRect(mn, mx, offset, std::make_shared<M>(std::move(material)));
    }
    static AxisRect new_with_material(Vec2 mn, Vec2 mx, float offset, MaterialPtr material) {
        return AxisRect(mn, mx, offset, std::move(material));
    }
    void flatten(FlatScene& fs, uint32_t group) const override {
        fs.add_rect((uint8_t)AXIS, min, max, offset, material->flatten(fs), group);
    }
};
using XYRect = AxisRect<0>;
using XZRect = AxisRect<1>;
using YZRect = AxisRect<2>;

struct Scene : Hittable {
    std::vector<HittablePtr> shapes;
    static Scene new_() { return Scene(); }
    template <class S>
    void add_shape(S shape) { shapes.push_back(std::make_shared<S>(std::move(shape))); }
    void clear() { shapes.clear(); }
    void flatten(FlatScene& fs, uint32_t group) const override {
        for (const auto& s : shapes) s->flatten(fs, group);
    }
};

struct Box2 : Hittable {
    Vec3A min, max;
    Scene sides;
    MaterialPtr material;
    template <class M>
    static Box2 new_(float width, float height, float depth, M material) {
        return full_box(Vec3A::ZERO(), Vec3A(width, height, depth), std::make_shared<M>(std::move(material)));
    }
    static Box2 full_box(Vec3A mn, Vec3A mx, MaterialPtr color) {  // shapes.rs:456-474
        Box2 b;
        b.min = mn;
        b.max = mx;
        b.material = color;
        b.sides.add_shape(XYRect::new_with_material(Vec2(mn.x, mn.y), Vec2(mx.x, mx.y), mn.z, color));
        b.sides.add_shape(XYRect::new_with_material(Vec2(mn.x, mn.y), Vec2(mx.x, mx.y), mx.z, color));
        b.sides.add_shape(XZRect::new_with_material(Vec2(mn.x, mn.z), Vec2(mx.x, mx.z), mn.y, color));
        b.sides.add_shape(XZRect::new_with_material(Vec2(mn.x, mn.z), Vec2(mx.x, mx.z), mx.y, color));
        b.sides.add_shape(YZRect::new_with_material(Vec2(mn.y, mn.z), Vec2(mx.y, mx.z), mn.x, color));
        b.sides.add_shape(YZRect::new_with_material(Vec2(mn.y, mn.z), Vec2(mx.y, mx.z), mx.x, color));
        return b;
    }
    void flatten(FlatScene& fs, uint32_t group) const override { sides.flatten(fs, group); }
};

// BVHNode::from_scene(&scene, t0, t1): on the device every level of nesting ends up in one linear BVH,
// so this node only keeps the objects (in insertion order).
struct BVHNode : Hittable {
    std::vector<HittablePtr> objects;
    static BVHNode from_scene(const Scene& scene, float /*t0*/, float /*t1*/) {
        BVHNode n;
        n.objects = scene.shapes;
        return n;
    }
    void flatten(FlatScene& fs, uint32_t group) const override {
        for (const auto& s : objects) s->flatten(fs, group);
    }
};

template <class T>
struct TransformedObject : Hittable {
    T object;
    Mat4 transform, inv_transform;
    TransformedObject(T obj, Mat4 m) : object(std::move(obj)), transform(m), inv_transform(m.inverse()) {}
    static TransformedObject new_(T object, Mat4 transform) { return TransformedObject(std::move(object), transform); }
    bool take_xform(float m[12], float minv[12], const Hittable** inner) const override {
        transform.to_rows(m);
        inv_transform.to_rows(minv);
        *inner = &object;
        return true;
    }
    void flatten(FlatScene& fs, uint32_t group) const override {
        // inside another TransformedObject: an instance group whose parent is that one (at most four stacked);
        // inside a medium's boundary (beyond the one transform the medium takes over itself): not on the device
        uint32_t depth = 0;
        for (uint32_t a = group; a != 0; a = fs.groups[a - 1].parent) {
            if (fs.groups[a - 1].kind != RT_GROUP_INSTANCE)
                throw std::runtime_error("TransformedObject nested inside a ConstantMedium boundary is not supported on the device");
            depth++;
        }
        if (depth >= 4) throw std::runtime_error("TransformedObject nested more than 4 deep is not supported on the device");
        rt_group g{};
        g.kind = RT_GROUP_INSTANCE;
        g.has_xform = 1;
        g.parent = group;
        transform.to_rows(g.m);
        inv_transform.to_rows(g.minv);
        fs.groups.push_back(g);
        fs.n_instances++;
        object.flatten(fs, (uint32_t)fs.groups.size());
    }
};

template <class T>
struct ConstantMedium : Hittable {
    T boundary;
    double negative_density;
    MaterialPtr material;
    ConstantMedium(T b, double nd, MaterialPtr m) : boundary(std::move(b)), negative_density(nd), material(std::move(m)) {}
    static ConstantMedium from_color(T boundary, float density, Vec3A color) {  // shapes.rs:639-645
        return ConstantMedium(std::move(boundary), -1.0 / (double)density, std::make_shared<IsotropicMat>(IsotropicMat::from_color(color)));
    }
    void flatten(FlatScene& fs, uint32_t group) const override {
        // inside TransformedObjects: the medium names the innermost one as its parent; inside another medium: not on the device
        for (uint32_t a = group; a != 0; a = fs.groups[a - 1].parent)
            if (fs.groups[a - 1].kind != RT_GROUP_INSTANCE)
                throw std::runtime_error("ConstantMedium nested inside a ConstantMedium boundary is not supported on the device");
        rt_group g{};
        g.kind = RT_GROUP_MEDIUM;
        g.parent = group;
        g.neg_inv_density = negative_density;
        g.mat = material->flatten(fs);
        const Hittable* inner = &boundary;
        const Hittable* probe = nullptr;
        if (boundary.take_xform(g.m, g.minv, &probe)) {
            g.has_xform = 1;
            inner = probe;
            // the wrapped TransformedObject still counts as an instance in the id walk
            fs.n_instances++;
        }
        fs.groups.push_back(g);
        fs.n_media++;
        inner->flatten(fs, (uint32_t)fs.groups.size());
    }
};

// ---- camera (camera.rs:28-52) -------------------------------------------------------------------
struct Camera {
    rt_camera pod;
    static Camera new_(Vec3A origin, Vec3A target, Vec3A up, float vfov, float aspect_ratio, float aperture,
                       float focus_distance, float time_start, float time_end) {
        float h = std::tan(vfov / 2.0f);
        float viewport_height = 2.0f * h;
        float viewport_width = viewport_height * aspect_ratio;
        Vec3A w = (origin - target).normalize();
        Vec3A u = up.cross(w).normalize();
        Vec3A v = w.cross(u);
        Vec3A horizontal = (focus_distance * viewport_width) * u;
        Vec3A vertical = (focus_distance * viewport_height) * v;
        Vec3A llc = origin - horizontal / 2.0f - vertical / 2.0f - focus_distance * w;
        Camera c;
        auto put = [](float* d, Vec3A s) {
            d[0] = s.x;
            d[1] = s.y;
            d[2] = s.z;
        };
        put(c.pod.origin, origin);
        put(c.pod.lower_left_corner, llc);
        put(c.pod.horizontal, horizontal);
        put(c.pod.vertical, vertical);
        put(c.pod.u, u);
        put(c.pod.v, v);
        put(c.pod.w, w);
        c.pod.lens_radius = aperture / 2.0f;
        c.pod.time_start = time_start;
        c.pod.time_end = time_end;
        return c;
    }
};

}  // namespace rt
