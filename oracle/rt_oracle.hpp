// CPU ORACLE -- TEST INFRASTRUCTURE ONLY.
//
// A plain C++ restatement of the render hot path of the CPU path tracer
// jameslen/raytracer-rs (Rust).  It exists to check the HIP implementation and
// to serve as the CPU baseline in bench.py.  It is never linked into, loaded
// by, or called from the product library: only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may use it.
//
// PARITY UNPINNED against a reference binary: the reference is Rust, no Rust
// toolchain exists in the build image, and the reference holds no tests,
// golden vectors or fixtures of any kind.  What pins this file instead are
// hand-derived known-answer tests (tests/test_oracle_kat.py), each derived
// from the reference lines cited next to the function it exercises.
//
// Third-party arithmetic restated from published crate behaviour (sources not
// present in the image): glam ^0.13.1 (Vec3A / Mat4), rand ^0.8 (f32/f64
// uniform conversions, Fisher-Yates shuffle).  rand's thread_rng is OS-seeded
// and therefore not reproducible; it is replaced by the counter-based
// generator specified in the "RNG" section below, which the HIP path
// implements independently from the same specification.
//
// Every class/function cites the reference file:line it follows
// (paths relative to /root/reference/raytracer-rs/).
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off: rustc never fuses
// a*b+c, so neither does the oracle).
#pragma once
#include <cstdint>

extern "C" {

typedef struct orc_scene orc_scene;

// ---- scenes (src/main.rs:87-305 + presets :386-451) -----------------------
// name: random_world | two_spheres | two_perlin_spheres | earth | simple_light
//       | cornell_box | cornell_smoke | final_scene | synth_spheres
// image_rgb: decoded earthmap (w*h*3 u8) for scenes that use ImageTexture, else null.
// n_synth: sphere count for synth_spheres (ignored otherwise).
orc_scene* orc_scene_create(const char* name, uint32_t seed,
                            const uint8_t* image_rgb, int image_w, int image_h,
                            int64_t n_synth);
void orc_scene_destroy(orc_scene*);
// info[0]=n_prims info[1]=n_instances info[2]=n_media info[3]=n_top_level
void orc_scene_info(const orc_scene*, int64_t info[4]);
// Top-level Sphere objects of the world list, in insertion order: out[4*i..] = cx,cy,cz,r
// (NaN for anything that is not a plain Sphere).  Returns the number of top-level objects.
int64_t orc_scene_spheres(const orc_scene*, int64_t cap, float* out);
// preset camera of the scene: cam[0..2]=origin cam[3..5]=target cam[6]=vfov_deg
// cam[7]=aperture cam[8..10]=background
void orc_scene_preset(const orc_scene*, float cam[11]);
// Camera::new (src/camera.rs:28-52): out[0..2] origin, [3..5] lower_left_corner,
// [6..8] horizontal, [9..11] vertical, [12..14] u, [15..17] v, [18..20] w,
// [21] lens_radius, [22] time_start, [23] time_end
void orc_camera_new(const float origin[3], const float target[3], const float up[3],
                    float vfov_rad, float aspect, float aperture, float focus,
                    float t0, float t1, float out[24]);

// ---- hit() path on fixed ray batches --------------------------------------
// mode 0: world as the linear Scene (src/scene.rs:27-41)
// mode 1: world as BVHNode::from_scene (src/bvh_node.rs:48-169)
// Outputs (any may be null): prim (>=0 primitive id, -1 miss, <=-2 medium -(k+2)),
// inst (instance id or -1), t, normal[3n], front, uv[2n], point[3n].
// Media draw their free-flight sample from stream (seed, pixel=ray index, sample 0), bounce 0.
// counters (nullable): [0] rays, [1] box tests, [2] primitive tests
int orc_intersect_batch(const orc_scene*, int mode, int64_t n,
                        const float* ox, const float* oy, const float* oz,
                        const float* dx, const float* dy, const float* dz,
                        const float* time, float tmin, float tmax, uint32_t seed,
                        int32_t* prim, int32_t* inst, float* t, float* normal,
                        uint8_t* front, float* uv, float* point,
                        uint64_t* counters, int threads);

// Fixed ray batches: camera rays of a w x h, 1-spp frame (which=0) or the first
// scattered ray of each of those paths (which=1; valid[i]=0 where the path
// ended at its first hit).  rays = 7 arrays of w*h floats (ox,oy,oz,dx,dy,dz,time).
int orc_gen_rays(const orc_scene*, int w, int h, float aspect, uint32_t seed, int which,
                 float* rays7, uint8_t* valid);

// ---- render loop (src/main.rs:44-63, 460-482) ------------------------------
// Adds samples [s0,s1) of pixels in [x0,x1)x[y0,y1) to sum_rgb (w*h*3 floats,
// row 0 = top), in sample order.  rays_out += number of world.intersect calls.
// counters (nullable) as above.
int orc_render(const orc_scene*, int mode, int w, int h, float aspect,
               int s0, int s1, int max_depth, uint32_t seed,
               int x0, int y0, int x1, int y1,
               float* sum_rgb, uint64_t* rays_out, uint64_t* counters, int threads);
// vec3_to_rgb(color * inv_samples) (src/main.rs:307-313,460,481)
void orc_resolve_u8(const float* sum_rgb, int64_t n_pixels, int spp, uint8_t* out);
void orc_resolve_u8_sqrt(const float* sum_rgb, int64_t n_pixels, int spp, uint8_t* out);

// ---- RNG specification probes ---------------------------------------------
void orc_rng_u32(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t first, int n, uint32_t* out);
void orc_rng_f32(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t first, int n, float* out);
void orc_rng_range(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t first, float lo, float hi, int n, float* out);
double orc_rng_medium_xi(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t medium);

// ---- known-answer probes on single objects ---------------------------------
// out: [0]=hit(0/1) [1]=t [2..4]=point [5..7]=normal [8]=front_face [9]=u [10]=v
void orc_kat_sphere(const float c[3], float r, const float ray[7], float tmin, float tmax, float out[11]);
void orc_kat_moving_sphere(const float c0[3], const float c1[3], float r, float t0, float t1,
                           const float ray[7], float tmin, float tmax, float out[11]);
// axis: 0 = XYRect, 1 = XZRect, 2 = YZRect; (min0,min1),(max0,max1), offset as the constructors take them
void orc_kat_rect(int axis, float min0, float min1, float max0, float max1, float offset,
                  const float ray[7], float tmin, float tmax, float out[11]);
// Box2::new(w,h,d) under translation * rotation_y(deg)   (src/main.rs:192-196)
void orc_kat_instanced_box(float w, float h, float d, float rot_y_deg, const float trans[3],
                           const float ray[7], float tmin, float tmax, float out[11], float aabb[6]);
int orc_kat_aabb_hit(const float bmin[3], const float bmax[3], const float ray[7], float tmin, float tmax);
// ConstantMedium over a sphere with the free-flight sample xi given explicitly
void orc_kat_medium_sphere(const float c[3], float r, float density, double xi,
                           const float ray[7], float tmin, float tmax, float out[11]);
void orc_kat_reflect(const float v[3], const float n[3], float out[3]);
void orc_kat_refract(const float uv[3], const float n[3], float ratio, float out[3]);
float orc_kat_reflectance(float cosine, float ref_idx);
uint8_t orc_kat_quantise(float f);
void orc_kat_image_value(const uint8_t* rgb, int w, int h, float u, float v, float out[3]);
float orc_kat_perlin_noise(const float* vec256x3, const int32_t* perm3x256, const float p[3]);
float orc_kat_perlin_turb(const float* vec256x3, const int32_t* perm3x256, const float p[3], int depth);
void orc_kat_checker(const float odd[3], const float even[3], const float p[3], float out[3]);
void orc_kat_sphere_uv(const float n[3], float out[2]);
// Material::scatter probes: kind 0 lambertian(albedo) 1 metal(albedo,fuzz) 2 dielectric(ior) 4 isotropic(albedo)
// hit record given as point, normal (already face-set), front flag; draws come from stream (seed,pixel,sample)
// starting at counter ctr.  out: [0]=scattered(0/1) [1..3]=origin [4..6]=direction [7..9]=attenuation [10]=draws consumed
void orc_kat_scatter(int kind, const float albedo[3], float param, const float ray[7],
                     const float point[3], const float normal[3], int front,
                     uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t ctr, float out[11]);
// Camera::get_ray with jitter, from stream (seed,pixel,sample) at counter 0 (src/main.rs:471-474)
void orc_kat_camera_ray(const float cam24[24], int w, int h, int i, int j,
                        uint32_t seed, uint32_t sample, float ray[7]);

}  // extern "C"
