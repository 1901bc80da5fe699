// The reference's main() (raytracer-rs/src/main.rs:335-498) on the GPU path, in C++ against the mirror of the
// reference's scene-construction API (raytracer-rs_amd/host/rt_scene.hpp) and the C ABI (include/rtamd.h):
// build a Scene exactly as the Rust builders do, flatten + upload instead of BVHNode::from_scene, rt_render
// instead of the rayon pixel loop, quantise, write the image.
//
// Built by raytracer-rs_amd/Makefile into raytracer-rs_amd/lib/render_main:
//   render_main [width] [samples_per_pixel] [out.png]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rt_scene.hpp"

using namespace rt;

static float degree_to_rad(float deg) { return deg * 3.14159265358979323846f / 180.0f; }  // main.rs:83-85

// main.rs:163-175 (simple_light), written with the same constructors in the same order
static Scene simple_light() {
    Scene s = Scene::new_();
    TexturePtr noise_texture = std::make_shared<NoiseTexture>(NoiseTexture::new_(4.0f));
    s.add_shape(Sphere::new_(Vec3A(0.0f, -1000.0f, 0.0f), 1000.0f, LambertianMat::from_shared_texture(noise_texture)));
    s.add_shape(Sphere::new_(Vec3A(0.0f, 2.0f, 0.0f), 2.0f, LambertianMat::from_shared_texture(noise_texture)));
    s.add_shape(XYRect::new_(Vec2(3.0f, 1.0f), Vec2(5.0f, 3.0f), -2.0f, DiffuseLight::from_color(Vec3A(4.0f, 4.0f, 4.0f))));
    s.add_shape(Sphere::new_(Vec3A(0.0f, 7.0f, 0.0f), 2.0f, DiffuseLight::from_color(Vec3A(4.0f, 4.0f, 4.0f))));
    return s;
}

int main(int argc, char** argv) {
    const uint32_t image_width = argc > 1 ? (uint32_t)atoi(argv[1]) : 400;
    const uint32_t samples_per_pixel = argc > 2 ? (uint32_t)atoi(argv[2]) : 100;
    const char* out_path = argc > 3 ? argv[3] : "test.png";
    const float aspect_ratio = 16.0f / 9.0f;
    const uint32_t image_height = (uint32_t)((float)image_width / aspect_ratio);  // main.rs:383
    const int max_depth = 50;

    thread_rng().seed(1);  // the Perlin tables of NoiseTexture::new draw from it
    Scene world = simple_light();
    // main.rs:418-425
    const Vec3A origin(26.0f, 3.0f, 6.0f), target(0.0f, 2.0f, 0.0f), vup(0.0f, 1.0f, 0.0f), background = Vec3A::ZERO();
    Camera camera = Camera::new_(origin, target, vup, degree_to_rad(20.0f), aspect_ratio, 0.0f, 10.0f, 0.0f, 1.0f);

    rt_ctx* ctx = nullptr;
    int rc = rt_ctx_create(0, &ctx);
    if (rc) {
        fprintf(stderr, "rt_ctx_create failed (%d): this path needs an MI355X, there is no CPU fallback\n", rc);
        return 1;
    }
    FlatScene fs;
    world.flatten(fs, 0);                       // instead of BVHNode::from_scene(&world, 0.0, 1.0), main.rs:458
    if ((rc = fs.upload(ctx, 0.0f, 1.0f))) {
        fprintf(stderr, "upload failed (%d): %s\n", rc, rt_last_error(ctx));
        return 1;
    }
    rt_render_params params{};
    params.width = image_width;
    params.height = image_height;
    params.sample0 = 0;
    params.sample1 = samples_per_pixel;
    params.max_depth = (uint32_t)max_depth;
    params.background[0] = background.x;
    params.background[1] = background.y;
    params.background[2] = background.z;
    params.seed = 1;
    std::vector<float> sum((size_t)image_width * image_height * 3, 0.0f);
    rt_render_stats stats{};
    auto now = std::chrono::steady_clock::now();
    if ((rc = rt_render(ctx, &camera.pod, &params, sum.data(), &stats))) {
        fprintf(stderr, "rt_render failed (%d): %s\n", rc, rt_last_error(ctx));
        return 1;
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - now).count();
    printf("Time elapsed: %.0f\n", ms);          // the span of main.rs:462,484
    printf("%llu rays, %.1f Mrays/s\n", (unsigned long long)stats.rays, (double)stats.rays / ms / 1e3);
    std::vector<uint8_t> out_data(sum.size());   // vec3_to_rgb(color * inv_samples), main.rs:481
    if ((rc = rt_resolve_u8(ctx, sum.data(), (uint64_t)image_width * image_height, samples_per_pixel, out_data.data())) ||
        (rc = rt_image_write(out_path, image_width, image_height, out_data.data(), RT_IMAGE_PNG))) {  // main.rs:486-497
        fprintf(stderr, "output failed (%d): %s\n", rc, rt_last_error(ctx));
        return 1;
    }
    rt_ctx_destroy(ctx);
    return 0;
}
