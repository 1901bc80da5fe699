// Host driver above the C ABI: the reference's eight scene builders (main.rs:87-305) and presets
// (main.rs:356-452) written against the C++ mirror of its scene API, plus one synthetic stress scene.
// Exposed to Python through a small C interface (rth_*); all rendering goes through include/rtamd.h.
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>

#include "rt_scene.hpp"

using namespace rt;

namespace {

float degree_to_rad(float deg) { return deg * 3.14159265358979323846f / 180.0f; }  // main.rs:83-85

struct Preset {  // main.rs:386-451
    Vec3A origin, target, background;
    float fov_deg = 20.0f, aperture = 0.0f;
};

struct Built {
    Scene world;
    Preset preset;
    uint32_t n_prims = 0, n_instances = 0, n_media = 0;
};

std::string g_earthmap_path;

Scene generate_random_world() {  // main.rs:87-125
    Scene s = Scene::new_();
    auto material = LambertianMat::from_texture(CheckeredTexture::from_color(Vec3A(0.2f, 0.3f, 0.1f), Vec3A(0.9f, 0.9f, 0.9f)));
    s.add_shape(Sphere::new_(Vec3A(0.0f, -1000.0f, 0.0f), 1000.0f, material));
    ThreadRng& rng = thread_rng();
    Vec3A point(4.0f, 0.2f, 0.0f);
    for (int a = -11; a < 11; a++) {
        for (int b = -11; b < 11; b++) {
            float choose_mat = rng.gen();
            float cx = (float)a + 0.9f * rng.gen();
            float cz = (float)b * 0.9f * rng.gen();
            Vec3A center(cx, 0.2f, cz);
            if ((center - point).length() > 0.9f) {
                if (choose_mat < 0.8f) {
                    Vec3A center2 = center + Vec3A(0.0f, rng.gen_range(0.0f, 0.5f), 0.0f);
                    Vec3A c1 = vec3_helpers::random();
                    Vec3A c2 = vec3_helpers::random();
                    s.add_shape(MovingSphere::new_(center, center2, 0.2f, 0.0f, 1.0f, LambertianMat::from_color(c1 * c2)));
                } else if (choose_mat < 0.95f) {
                    Vec3A albedo = vec3_helpers::random_range(0.5f, 1.0f);
                    float fuzz = rng.gen_range(0.5f, 1.0f);
                    s.add_shape(Sphere::new_(center, 0.2f, MetalMat::new_(albedo, fuzz)));
                } else {
                    s.add_shape(Sphere::new_(center, 0.2f, DielectricMat::new_(1.5f)));
                }
            }
        }
    }
    s.add_shape(Sphere::new_(Vec3A(0.0f, 1.0f, 0.0f), 1.0f, DielectricMat::new_(1.5f)));
    s.add_shape(Sphere::new_(Vec3A(-4.0f, 1.0f, 0.0f), 1.0f, LambertianMat::from_color(Vec3A(0.4f, 0.2f, 0.1f))));
    s.add_shape(Sphere::new_(Vec3A(4.0f, 1.0f, 0.0f), 1.0f, MetalMat::new_(Vec3A(0.7f, 0.6f, 0.5f), 0.0f)));
    return s;
}

Scene two_spheres() {  // main.rs:127-139
    Scene s = Scene::new_();
    TexturePtr checkered = std::make_shared<CheckeredTexture>(CheckeredTexture::from_color(Vec3A(0.2f, 0.3f, 0.1f), Vec3A(0.9f, 0.9f, 0.9f)));
    s.add_shape(Sphere::new_(Vec3A(0.0f, 10.0f, 0.0f), 10.0f, LambertianMat::from_shared_texture(checkered)));
    s.add_shape(Sphere::new_(Vec3A(0.0f, -10.0f, 0.0f), 10.0f, LambertianMat::from_shared_texture(checkered)));
    return s;
}

Scene two_perlin_spheres() {  // main.rs:141-150
    Scene s = Scene::new_();
    TexturePtr noise_texture = std::make_shared<NoiseTexture>(NoiseTexture::new_(4.0f));
    s.add_shape(Sphere::new_(Vec3A(0.0f, -1000.0f, 0.0f), 1000.0f, LambertianMat::from_shared_texture(noise_texture)));
    s.add_shape(Sphere::new_(Vec3A(0.0f, 2.0f, 0.0f), 2.0f, LambertianMat::from_shared_texture(noise_texture)));
    return s;
}

Scene earth() {  // main.rs:152-161
    Scene s = Scene::new_();
    TexturePtr texture = std::make_shared<ImageTexture>(ImageTexture::new_(g_earthmap_path));
    auto surface = LambertianMat::from_shared_texture(texture);
    s.add_shape(Sphere::new_(Vec3A::ZERO(), 2.0f, surface));
    return s;
}

Scene simple_light() {  // main.rs:163-175
    Scene s = Scene::new_();
    TexturePtr noise_texture = std::make_shared<NoiseTexture>(NoiseTexture::new_(4.0f));
    s.add_shape(Sphere::new_(Vec3A(0.0f, -1000.0f, 0.0f), 1000.0f, LambertianMat::from_shared_texture(noise_texture)));
    s.add_shape(Sphere::new_(Vec3A(0.0f, 2.0f, 0.0f), 2.0f, LambertianMat::from_shared_texture(noise_texture)));
    s.add_shape(XYRect::new_(Vec2(3.0f, 1.0f), Vec2(5.0f, 3.0f), -2.0f, DiffuseLight::from_color(Vec3A(4.0f, 4.0f, 4.0f))));
    s.add_shape(Sphere::new_(Vec3A(0.0f, 7.0f, 0.0f), 2.0f, DiffuseLight::from_color(Vec3A(4.0f, 4.0f, 4.0f))));
    return s;
}

void cornell_shell(Scene& s, Vec3A light, Vec2 lmin, Vec2 lmax) {  // main.rs:180-190, 210-220
    Vec3A white(0.73f, 0.73f, 0.73f), green(0.12f, 0.45f, 0.15f), red(0.65f, 0.05f, 0.05f);
    s.add_shape(YZRect::new_(Vec2(0.0f, 0.0f), Vec2(555.0f, 555.0f), 555.0f, LambertianMat::from_color(green)));
    s.add_shape(YZRect::new_(Vec2(0.0f, 0.0f), Vec2(555.0f, 555.0f), 0.0f, LambertianMat::from_color(red)));
    s.add_shape(XZRect::new_(lmin, lmax, 554.0f, DiffuseLight::from_color(light)));
    s.add_shape(XZRect::new_(Vec2(0.0f, 0.0f), Vec2(555.0f, 555.0f), 555.0f, LambertianMat::from_color(white)));
    s.add_shape(XZRect::new_(Vec2(0.0f, 0.0f), Vec2(555.0f, 555.0f), 0.0f, LambertianMat::from_color(white)));
    s.add_shape(XYRect::new_(Vec2(0.0f, 0.0f), Vec2(555.0f, 555.0f), 555.0f, LambertianMat::from_color(white)));
}

TransformedObject<Box2> cornell_block(float w, float h, float d, float deg, Vec3 offset) {  // main.rs:192-196
    Vec3A white(0.73f, 0.73f, 0.73f);
    Box2 b = Box2::new_(w, h, d, LambertianMat::from_color(white));
    Mat4 rotation = Mat4::from_rotation_y(degree_to_rad(deg));
    Mat4 translation = Mat4::from_translation(offset);
    Mat4 final_transform = translation * rotation;
    return TransformedObject<Box2>::new_(b, final_transform);
}

Scene cornell_box() {  // main.rs:177-205
    Scene s = Scene::new_();
    cornell_shell(s, Vec3A(15.0f, 15.0f, 15.0f), Vec2(213.0f, 227.0f), Vec2(343.0f, 332.0f));
    s.add_shape(cornell_block(165.0f, 165.0f, 165.0f, -18.0f, Vec3(130.0f, 0.0f, 65.0f)));
    s.add_shape(cornell_block(165.0f, 330.0f, 165.0f, 15.0f, Vec3(265.0f, 0.0f, 305.0f)));
    return s;
}

Scene cornell_smoke() {  // main.rs:207-238
    Scene s = Scene::new_();
    cornell_shell(s, Vec3A(7.0f, 7.0f, 7.0f), Vec2(113.0f, 127.0f), Vec2(443.0f, 432.0f));
    auto box2 = cornell_block(165.0f, 165.0f, 165.0f, -18.0f, Vec3(130.0f, 0.0f, 65.0f));
    s.add_shape(ConstantMedium<TransformedObject<Box2>>::from_color(box2, 0.5f, Vec3A::ONE()));
    auto box1 = cornell_block(165.0f, 330.0f, 165.0f, 15.0f, Vec3(265.0f, 0.0f, 305.0f));
    s.add_shape(ConstantMedium<TransformedObject<Box2>>::from_color(box1, 0.5f, Vec3A::ZERO()));
    return s;
}

// Not a reference preset: the Cornell shell around transformed objects INSIDE transformed objects, which the
// reference's types allow (TransformedObject<T: Hittable>, shapes.rs:26-41) and none of its scenes uses.  One block is
// a translation around a rotation around a box; the other stacks four transforms (translate, rotate, a BVH of five
// objects, two of them transformed again, one of those twice, one a medium); a third is a medium inside a transform.
Scene nested_instances() {
    Scene s = Scene::new_();
    cornell_shell(s, Vec3A(15.0f, 15.0f, 15.0f), Vec2(213.0f, 227.0f), Vec2(343.0f, 332.0f));
    Vec3A white(0.73f, 0.73f, 0.73f);
    {
        auto turned = TransformedObject<Box2>::new_(Box2::new_(165.0f, 165.0f, 165.0f, LambertianMat::from_color(white)),
                                                    Mat4::from_rotation_y(degree_to_rad(-18.0f)));
        s.add_shape(TransformedObject<TransformedObject<Box2>>::new_(turned, Mat4::from_translation(Vec3(130.0f, 0.0f, 65.0f))));
    }
    Scene inner = Scene::new_();
    inner.add_shape(Sphere::new_(Vec3A(60.0f, 200.0f, 60.0f), 40.0f, MetalMat::new_(Vec3A(0.8f, 0.85f, 0.88f), 0.0f)));
    {
        auto turned = TransformedObject<Box2>::new_(Box2::new_(80.0f, 80.0f, 80.0f, LambertianMat::from_color(Vec3A(0.65f, 0.05f, 0.05f))),
                                                    Mat4::from_rotation_y(degree_to_rad(30.0f)));
        inner.add_shape(TransformedObject<TransformedObject<Box2>>::new_(turned, Mat4::from_translation(Vec3(20.0f, 120.0f, 20.0f))));
    }
    inner.add_shape(TransformedObject<Sphere>::new_(Sphere::new_(Vec3A(0.0f, 0.0f, 0.0f), 30.0f, DielectricMat::new_(1.5f)),
                                                    Mat4::from_translation(Vec3(120.0f, 60.0f, 100.0f))));
    inner.add_shape(Box2::new_(165.0f, 100.0f, 165.0f, LambertianMat::from_color(white)));
    // a medium two transforms deep (it is handed the ray of the inner one), and below one alone inside a transform
    inner.add_shape(ConstantMedium<Sphere>::from_color(Sphere::new_(Vec3A(100.0f, 170.0f, 60.0f), 45.0f, DielectricMat::new_(1.5f)), 0.05f,
                                                       Vec3A(0.2f, 0.4f, 0.9f)));
    {
        auto fog = ConstantMedium<Box2>::from_color(Box2::new_(120.0f, 90.0f, 120.0f, DielectricMat::new_(1.5f)), 0.02f, Vec3A(0.9f, 0.9f, 0.9f));
        Mat4 place = Mat4::from_translation(Vec3(30.0f, 330.0f, 200.0f)) * Mat4::from_rotation_y(degree_to_rad(25.0f));
        s.add_shape(TransformedObject<ConstantMedium<Box2>>::new_(fog, place));
    }
    auto turned = TransformedObject<BVHNode>::new_(BVHNode::from_scene(inner, 0.0f, 1.0f), Mat4::from_rotation_y(degree_to_rad(15.0f)));
    s.add_shape(TransformedObject<TransformedObject<BVHNode>>::new_(turned, Mat4::from_translation(Vec3(265.0f, 0.0f, 295.0f))));
    return s;
}

// Not a reference preset either: media whose boundary is more than one object, which ConstantMedium<T: Hittable>
// (shapes.rs:624-646) allows -- a Scene of 27 overlapping spheres, and an untransformed box -- in the Cornell shell.
// (A Scene, not a BVHNode: the reference's box test rejects boxes behind the ray origin whatever interval it is asked
// about, aabb.rs:21-33, so which objects a BVH boundary offers to the medium's first query over the whole line depends
// on its randomly built tree.  The device evaluates either as the list of objects, Scene::intersect, scene.rs:27-41.)
Scene medium_cluster() {
    Scene s = Scene::new_();
    cornell_shell(s, Vec3A(7.0f, 7.0f, 7.0f), Vec2(113.0f, 127.0f), Vec2(443.0f, 432.0f));
    Scene balls = Scene::new_();
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            for (int k = 0; k < 3; k++)
                balls.add_shape(Sphere::new_(Vec3A(218.0f + 60.0f * (float)i, 150.0f + 60.0f * (float)j, 218.0f + 60.0f * (float)k), 45.0f,
                                             DielectricMat::new_(1.5f)));
    s.add_shape(ConstantMedium<Scene>::from_color(balls, 0.02f, Vec3A(0.9f, 0.9f, 0.9f)));
    s.add_shape(ConstantMedium<Box2>::from_color(Box2::full_box(Vec3A(40.0f, 0.0f, 60.0f), Vec3A(160.0f, 120.0f, 180.0f),
                                                                std::make_shared<DielectricMat>(DielectricMat::new_(1.5f))),
                                                 0.03f, Vec3A(0.1f, 0.1f, 0.4f)));
    return s;
}

Scene final_scene() {  // main.rs:240-305
    Scene s = Scene::new_();
    Scene boxes = Scene::new_();
    MaterialPtr ground = std::make_shared<LambertianMat>(LambertianMat::from_color(Vec3A(0.48f, 0.83f, 0.53f)));
    const int boxes_per_side = 20;
    ThreadRng& rng = thread_rng();
    for (int i = 0; i < boxes_per_side; i++) {
        for (int j = 0; j < boxes_per_side; j++) {
            float w = 100.0f;
            float x0 = -1000.0f + (float)i * w;
            float z0 = -1000.0f + (float)j * w;
            float y0 = 0.0f;
            float x1 = x0 + w;
            float y1 = rng.gen_range(1.0f, 101.0f);
            float z1 = z0 + w;
            boxes.add_shape(Box2::full_box(Vec3A(x0, y0, z0), Vec3A(x1, y1, z1), ground));
        }
    }
    s.add_shape(BVHNode::from_scene(boxes, 0.0f, 1.0f));
    auto light = DiffuseLight::from_color(Vec3A(7.0f, 7.0f, 7.0f));
    s.add_shape(XZRect::new_(Vec2(123.0f, 147.0f), Vec2(423.0f, 412.0f), 554.0f, light));
    Vec3A center1(400.0f, 400.0f, 200.0f);
    Vec3A center2 = center1 + Vec3A(30.0f, 0.0f, 0.0f);
    auto moving_mat = LambertianMat::from_color(Vec3A(0.7f, 0.3f, 0.1f));
    s.add_shape(MovingSphere::new_(center1, center2, 50.0f, 0.0f, 1.0f, moving_mat));
    s.add_shape(Sphere::new_(Vec3A(260.0f, 150.0f, 45.0f), 50.0f, DielectricMat::new_(1.5f)));
    s.add_shape(Sphere::new_(Vec3A(0.0f, 150.0f, 145.0f), 50.0f, MetalMat::new_(Vec3A(0.8f, 0.8f, 0.9f), 1.0f)));
    Sphere boundary = Sphere::new_(Vec3A(360.0f, 150.0f, 145.0f), 70.0f, DielectricMat::new_(1.5f));
    s.add_shape(boundary);
    Sphere boundary2 = Sphere::new_(Vec3A(360.0f, 150.0f, 145.0f), 70.0f, DielectricMat::new_(1.5f));
    s.add_shape(ConstantMedium<Sphere>::from_color(boundary2, 0.2f, Vec3A(0.2f, 0.4f, 0.9f)));
    Sphere boundary3 = Sphere::new_(Vec3A(0.0f, 0.0f, 0.0f), 5000.0f, DielectricMat::new_(1.5f));
    s.add_shape(ConstantMedium<Sphere>::from_color(boundary3, 0.005f, Vec3A(1.0f, 1.0f, 1.0f)));
    auto emat = LambertianMat::from_texture(ImageTexture::new_(g_earthmap_path));
    s.add_shape(Sphere::new_(Vec3A(400.0f, 200.0f, 400.0f), 100.0f, emat));
    auto pertext = NoiseTexture::new_(0.1f);
    s.add_shape(Sphere::new_(Vec3A(220.0f, 280.0f, 300.0f), 80.0f, LambertianMat::from_texture(pertext)));
    Scene boxes2 = Scene::new_();
    for (int i = 0; i < 1000; i++) {
        auto white = LambertianMat::from_color(Vec3A(0.73f, 0.73f, 0.73f));
        boxes2.add_shape(Sphere::new_(vec3_helpers::random_range(0.0f, 165.0f), 10.0f, white));
    }
    Mat4 rotation = Mat4::from_rotation_y(degree_to_rad(15.0f));
    Mat4 translation = Mat4::from_translation(Vec3(-100.0f, 270.0f, 395.0f));
    Mat4 final_transform = translation * rotation;
    s.add_shape(TransformedObject<BVHNode>::new_(BVHNode::from_scene(boxes2, 0.0f, 1.0f), final_transform));
    return s;
}

// Synthetic deep-BVH stress scene (no counterpart in the reference): n spheres with centres in
// U[-1000,1000) x U[0,50) x U[-1000,1000), radius U[0.05,0.5); 70 % Lambertian(xi*xi), 20 % Metal
// (albedo U[0.5,1)^3, fuzz U[0,0.5)), 10 % Dielectric(1.5); ground sphere of radius 1e5 below.
Scene synth_spheres(int64_t n) {
    Scene s = Scene::new_();
    s.add_shape(Sphere::new_(Vec3A(0.0f, -100000.0f, 0.0f), 100000.0f, LambertianMat::from_color(Vec3A(0.5f, 0.5f, 0.5f))));
    ThreadRng& rng = thread_rng();
    s.shapes.reserve((size_t)n + 1);
    for (int64_t i = 0; i < n; i++) {
        float cx = rng.gen_range(-1000.0f, 1000.0f);
        float cy = rng.gen_range(0.0f, 50.0f);
        float cz = rng.gen_range(-1000.0f, 1000.0f);
        float r = rng.gen_range(0.05f, 0.5f);
        float choose = rng.gen();
        if (choose < 0.7f) {
            Vec3A c1 = vec3_helpers::random();
            Vec3A c2 = vec3_helpers::random();
            s.add_shape(Sphere::new_(Vec3A(cx, cy, cz), r, LambertianMat::from_color(c1 * c2)));
        } else if (choose < 0.9f) {
            Vec3A albedo = vec3_helpers::random_range(0.5f, 1.0f);
            float fuzz = rng.gen_range(0.0f, 0.5f);
            s.add_shape(Sphere::new_(Vec3A(cx, cy, cz), r, MetalMat::new_(albedo, fuzz)));
        } else {
            s.add_shape(Sphere::new_(Vec3A(cx, cy, cz), r, DielectricMat::new_(1.5f)));
        }
    }
    return s;
}

void copy_err(char* err, int errlen, const std::string& msg) {
    if (err && errlen > 0) {
        std::strncpy(err, msg.c_str(), (size_t)errlen - 1);
        err[errlen - 1] = 0;
    }
}

}  // namespace

extern "C" {

// Builds one of the named scenes.  Returns an opaque handle or NULL (message in err).
void* rth_scene_build(const char* name, uint32_t seed, const char* earthmap_path, int64_t n_synth, char* err, int errlen) {
    try {
        std::string n(name ? name : "");
        g_earthmap_path = earthmap_path ? earthmap_path : "earthmap.ppm";
        thread_rng().seed(seed);
        auto b = std::make_unique<Built>();
        Preset& p = b->preset;
        p.origin = Vec3A(13.0f, 2.0f, 3.0f);
        p.target = Vec3A(0.0f, 0.0f, 0.0f);
        p.background = Vec3A(0.70f, 0.80f, 1.00f);
        if (n == "random_world") {
            b->world = generate_random_world();
            p.aperture = 0.1f;
        } else if (n == "two_spheres") {
            b->world = two_spheres();
        } else if (n == "two_perlin_spheres") {
            b->world = two_perlin_spheres();
        } else if (n == "earth") {
            b->world = earth();
        } else if (n == "simple_light") {
            b->world = simple_light();
            p.origin = Vec3A(26.0f, 3.0f, 6.0f);
            p.target = Vec3A(0.0f, 2.0f, 0.0f);
            p.background = Vec3A::ZERO();
        } else if (n == "cornell_box" || n == "cornell_smoke" || n == "nested_instances" || n == "medium_cluster") {
            b->world = n == "cornell_box" ? cornell_box()
                                          : (n == "cornell_smoke" ? cornell_smoke() : (n == "nested_instances" ? nested_instances() : medium_cluster()));
            p.origin = Vec3A(278.0f, 278.0f, -800.0f);
            p.target = Vec3A(278.0f, 278.0f, 0.0f);
            p.fov_deg = 40.0f;
            p.background = Vec3A::ZERO();
        } else if (n == "final_scene") {
            b->world = final_scene();
            p.origin = Vec3A(478.0f, 278.0f, -600.0f);
            p.target = Vec3A(278.0f, 278.0f, 0.0f);
            p.fov_deg = 40.0f;
            p.background = Vec3A::ZERO();
        } else if (n == "synth_spheres") {
            if (n_synth < 0) throw std::runtime_error("synth_spheres: negative count");
            b->world = synth_spheres(n_synth);
            p.origin = Vec3A(0.0f, 120.0f, -1400.0f);
            p.target = Vec3A(0.0f, 0.0f, 0.0f);
            p.fov_deg = 40.0f;
        } else {
            throw std::runtime_error("unknown scene '" + n + "'");
        }
        return b.release();
    } catch (const std::exception& e) {
        copy_err(err, errlen, e.what());
        return nullptr;
    }
}

void rth_scene_destroy(void* h) { delete static_cast<Built*>(h); }

// Flattens the scene, sends it through rt_scene_set_* and commits with shutter [0,1] (main.rs:458).
int rth_scene_upload(void* h, rt_ctx* ctx, char* err, int errlen) {
    Built* b = static_cast<Built*>(h);
    if (!b || !ctx) return RT_ERR_INVALID;
    try {
        FlatScene fs;
        b->world.flatten(fs, 0);
        b->n_prims = fs.next_id;
        b->n_instances = fs.n_instances;
        b->n_media = fs.n_media;
        int rc = fs.upload(ctx, 0.0f, 1.0f);
        if (rc) copy_err(err, errlen, rt_last_error(ctx));
        return rc;
    } catch (const std::exception& e) {
        copy_err(err, errlen, e.what());
        return RT_ERR_UNSUPPORTED;
    }
}

// Flattens without touching a GPU and reports what the C ABI would receive (host-logic checks):
// info[0] primitives, [1] instances (TransformedObject count), [2] media, [3] top-level objects,
// [4] spheres, [5] rects, [6] groups, [7] materials, [8] textures, [9] perlin tables, [10] images,
// [11] moving spheres.  ids_ok is set to 1 when the primitive ids are exactly 0..n-1.
int rth_scene_flatten_counts(void* h, int64_t info[12], int* ids_ok, char* err, int errlen) {
    Built* b = static_cast<Built*>(h);
    if (!b) return RT_ERR_INVALID;
    try {
        FlatScene fs;
        b->world.flatten(fs, 0);
        info[0] = fs.next_id;
        info[1] = fs.n_instances;
        info[2] = fs.n_media;
        info[3] = (int64_t)b->world.shapes.size();
        info[4] = (int64_t)fs.sph_id.size();
        info[5] = (int64_t)fs.rect_id.size();
        info[6] = (int64_t)fs.groups.size();
        info[7] = (int64_t)fs.mats.size();
        info[8] = (int64_t)fs.texs.size();
        info[9] = (int64_t)fs.perlin.size();
        info[10] = (int64_t)fs.images.size();
        int64_t moving = 0;
        for (uint8_t m : fs.sph_moving) moving += m;
        info[11] = moving;
        std::vector<uint8_t> seen(fs.next_id, 0);
        bool ok = fs.sph_id.size() + fs.rect_id.size() == fs.next_id;
        for (uint32_t id : fs.sph_id) ok = ok && id < fs.next_id && !seen[id]++;
        for (uint32_t id : fs.rect_id) ok = ok && id < fs.next_id && !seen[id]++;
        if (ids_ok) *ids_ok = ok ? 1 : 0;
        return RT_OK;
    } catch (const std::exception& e) {
        copy_err(err, errlen, e.what());
        return RT_ERR_UNSUPPORTED;
    }
}

// The index-th image the flattened scene would hand to rt_scene_set_image (host-logic checks of ImageTexture::new):
// sets *w, *h and copies w*h*3 bytes when rgb is given and cap is large enough.
int rth_scene_flattened_image(void* h, uint32_t index, uint32_t* w, uint32_t* hgt, uint8_t* rgb, uint64_t cap) {
    Built* b = static_cast<Built*>(h);
    if (!b || !w || !hgt) return RT_ERR_INVALID;
    try {
        FlatScene fs;
        b->world.flatten(fs, 0);
        if (index >= fs.images.size()) return RT_ERR_INVALID;
        const auto& im = fs.images[index];
        *w = im.w;
        *hgt = im.h;
        if (rgb) {
            if (cap < im.rgb.size()) return RT_ERR_INVALID;
            std::memcpy(rgb, im.rgb.data(), im.rgb.size());
        }
        return RT_OK;
    } catch (const std::exception&) {
        return RT_ERR_UNSUPPORTED;
    }
}

// cam[0..2] origin, [3..5] target, [6] vfov in degrees, [7] aperture, [8..10] background
void rth_scene_preset(void* h, float cam[11]) {
    const Preset& p = static_cast<Built*>(h)->preset;
    cam[0] = p.origin.x;
    cam[1] = p.origin.y;
    cam[2] = p.origin.z;
    cam[3] = p.target.x;
    cam[4] = p.target.y;
    cam[5] = p.target.z;
    cam[6] = p.fov_deg;
    cam[7] = p.aperture;
    cam[8] = p.background.x;
    cam[9] = p.background.y;
    cam[10] = p.background.z;
}

// valid after rth_scene_upload: info[0] primitives, [1] instances, [2] media, [3] top-level objects
void rth_scene_counts(void* h, int64_t info[4]) {
    const Built* b = static_cast<Built*>(h);
    info[0] = b->n_prims;
    info[1] = b->n_instances;
    info[2] = b->n_media;
    info[3] = (int64_t)b->world.shapes.size();
}

// Camera::new with vup = Y, focus 10, shutter [0,1] as main.rs:454-456 sets them
void rth_scene_camera(void* h, float aspect, rt_camera* out) {
    const Preset& p = static_cast<Built*>(h)->preset;
    *out = Camera::new_(p.origin, p.target, Vec3A::Y(), degree_to_rad(p.fov_deg), aspect, p.aperture, 10.0f, 0.0f, 1.0f).pod;
}

void rth_camera_new(const float origin[3], const float target[3], const float up[3], float vfov_rad, float aspect,
                    float aperture, float focus, float t0, float t1, rt_camera* out) {
    *out = Camera::new_(Vec3A(origin[0], origin[1], origin[2]), Vec3A(target[0], target[1], target[2]),
                        Vec3A(up[0], up[1], up[2]), vfov_rad, aspect, aperture, focus, t0, t1)
               .pod;
}

}  // extern "C"
