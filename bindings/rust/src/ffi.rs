//! Declarations of include/rtamd.h (the C ABI of the MI355X render path), item for item.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct rt_ctx {
    _private: [u8; 0],
}

pub const RT_OK: c_int = 0;
pub const RT_ERR_INVALID: c_int = -1;
pub const RT_ERR_HIP: c_int = -2;
pub const RT_ERR_UNSUPPORTED: c_int = -3;
pub const RT_ERR_NOT_COMMITTED: c_int = -4;
pub const RT_ERR_NO_DEVICE: c_int = -5;
pub const RT_ERR_IO: c_int = -6;

pub const RT_MAT_LAMBERTIAN: u32 = 0;
pub const RT_MAT_METAL: u32 = 1;
pub const RT_MAT_DIELECTRIC: u32 = 2;
pub const RT_MAT_DIFFUSE_LIGHT: u32 = 3;
pub const RT_MAT_ISOTROPIC: u32 = 4;
pub const RT_TEX_SOLID: u32 = 0;
pub const RT_TEX_CHECKER: u32 = 1;
pub const RT_TEX_NOISE: u32 = 2;
pub const RT_TEX_IMAGE: u32 = 3;
pub const RT_GROUP_INSTANCE: u32 = 1;
pub const RT_GROUP_MEDIUM: u32 = 2;
pub const RT_TRANSFER_LINEAR: c_int = 0;
pub const RT_TRANSFER_SQRT: c_int = 1;
pub const RT_IMAGE_PNG: c_int = 0;
pub const RT_IMAGE_PPM: c_int = 1;
pub const RT_NOALLOC_MAX_RAYS: u32 = 65536;
pub const RT_DEFAULT_TILE: u32 = 16;

#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct rt_material {
    pub kind: u32,
    pub tex: u32,
    pub albedo: [f32; 3],
    pub fuzz: f32,
    pub ior: f32,
    pub reserved: u32,
}
#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct rt_texture {
    pub kind: u32,
    pub color: [f32; 3],
    pub odd: u32,
    pub even: u32,
    pub freq: f32,
    pub table: u32,
}
#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct rt_group {
    pub kind: u32,
    pub has_xform: u32,
    pub m: [f32; 12],
    pub minv: [f32; 12],
    pub neg_inv_density: f64,
    pub mat: u32,
    /// 0: in the world; otherwise 1 + index of the instance group this one stands inside
    pub parent: u32,
}
#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct rt_scene_info {
    pub n_prims: u64,
    pub n_nodes: u64,
    pub n_groups: u64,
    pub n_media: u64,
    pub max_depth: u32,
    pub stack_entries: u32,
    pub device_bytes: u64,
    pub build_seconds: f64,
    pub tree_seconds: f64,
    pub builder: u32,
    pub bvh_width: u32,
    pub crown_nodes: u32,
    pub reserved: u32,
}
#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct rt_camera {
    pub origin: [f32; 3],
    pub lower_left_corner: [f32; 3],
    pub horizontal: [f32; 3],
    pub vertical: [f32; 3],
    pub u: [f32; 3],
    pub v: [f32; 3],
    pub w: [f32; 3],
    pub lens_radius: f32,
    pub time_start: f32,
    pub time_end: f32,
}
#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct rt_render_params {
    pub width: u32,
    pub height: u32,
    pub sample0: u32,
    pub sample1: u32,
    pub max_depth: u32,
    pub background: [f32; 3],
    pub seed: u32,
    pub tile_size: u32,
    pub rank: u32,
    pub nranks: u32,
    pub reserved: u32,
}
#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct rt_render_stats {
    pub rays: u64,
    pub paths: u64,
    pub batches: u32,
    pub bounce_launches: u32,
    pub render_ms: f64,
    pub trace_ms: f64,
    pub shade_ms: f64,
    pub raygen_ms: f64,
    pub accumulate_ms: f64,
    pub trace_launches: u64,
    pub shade_launches: u64,
    pub node_visits: u64,
    pub prim_tests: u64,
    pub bounce_ms: f64,
    pub bounce_kernel_launches: u64,
    pub pipeline: u32,
    pub tail_launches: u32,
    pub tail_rays: u64,
    pub tail_ms: f64,
    pub paths_ms: f64,
    pub paths_launches: u64,
    pub settled_rays: u64,
    pub settled_ms: f64,
    pub settled_launches: u64,
    pub settled_entries: u64,
}

extern "C" {
    pub fn rt_ctx_create(device: c_int, out: *mut *mut rt_ctx) -> c_int;
    pub fn rt_ctx_destroy(ctx: *mut rt_ctx);
    pub fn rt_last_error(ctx: *const rt_ctx) -> *const c_char;
    pub fn rt_ctx_set_stream(ctx: *mut rt_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn rt_ctx_set_max_paths(ctx: *mut rt_ctx, max_paths: u64) -> c_int;
    pub fn rt_ctx_reserve(ctx: *mut rt_ctx, n_paths: u64) -> c_int;
    pub fn rt_ctx_synchronize(ctx: *mut rt_ctx) -> c_int;
    pub fn rt_ctx_set_pipeline(ctx: *mut rt_ctx, pipeline: c_int) -> c_int;
    pub fn rt_ctx_set_tail_paths(ctx: *mut rt_ctx, tail_paths: u64) -> c_int;
    pub fn rt_ctx_set_builder(ctx: *mut rt_ctx, builder: c_int) -> c_int;
    pub fn rt_ctx_set_bvh_width(ctx: *mut rt_ctx, width: c_int) -> c_int;
    pub fn rt_ctx_set_fuse_boxes(ctx: *mut rt_ctx, on: c_int) -> c_int;
    pub fn rt_scene_clear(ctx: *mut rt_ctx) -> c_int;
    pub fn rt_scene_set_spheres(ctx: *mut rt_ctx, n: u32, c0_r: *const f32, c1: *const f32, t01: *const f32,
                                moving: *const u8, mat: *const u32, id: *const u32, group: *const u32) -> c_int;
    pub fn rt_scene_set_rects(ctx: *mut rt_ctx, n: u32, axis: *const u8, k: *const f32, bounds: *const f32,
                              mat: *const u32, id: *const u32, group: *const u32) -> c_int;
    pub fn rt_scene_set_groups(ctx: *mut rt_ctx, n: u32, groups: *const rt_group) -> c_int;
    pub fn rt_scene_set_materials(ctx: *mut rt_ctx, n: u32, mats: *const rt_material) -> c_int;
    pub fn rt_scene_set_textures(ctx: *mut rt_ctx, n: u32, texs: *const rt_texture) -> c_int;
    pub fn rt_scene_set_perlin(ctx: *mut rt_ctx, table: u32, vec: *const f32, perm: *const i32) -> c_int;
    pub fn rt_scene_set_image(ctx: *mut rt_ctx, table: u32, w: u32, h: u32, rgb: *const u8) -> c_int;
    pub fn rt_scene_commit(ctx: *mut rt_ctx, shutter_t0: f32, shutter_t1: f32) -> c_int;
    pub fn rt_scene_get_info(ctx: *mut rt_ctx, out: *mut rt_scene_info) -> c_int;
    pub fn rt_scene_read_bvh(ctx: *mut rt_ctx, nodes: *mut c_void, nodes_bytes: u64, prims: *mut c_void, prims_bytes: u64,
                             root_ref: *mut u32) -> c_int;
    pub fn rt_intersect_batch(ctx: *mut rt_ctx, n: u64, ox: *const f32, oy: *const f32, oz: *const f32, dx: *const f32,
                              dy: *const f32, dz: *const f32, time: *const f32, tmin: f32, tmax: f32, seed: u32,
                              prim: *mut i32, inst: *mut i32, t: *mut f32, normal: *mut f32, front: *mut u8,
                              uv: *mut f32, point: *mut f32) -> c_int;
    pub fn rt_intersect_batch_noalloc(ctx: *mut rt_ctx, n: u64, ox: *const f32, oy: *const f32, oz: *const f32,
                                      dx: *const f32, dy: *const f32, dz: *const f32, time: *const f32, tmin: f32,
                                      tmax: f32, seed: u32, prim: *mut i32, inst: *mut i32, t: *mut f32, normal: *mut f32,
                                      front: *mut u8, uv: *mut f32, point: *mut f32) -> c_int;
    pub fn rt_render(ctx: *mut rt_ctx, cam: *const rt_camera, params: *const rt_render_params, accum_rgb: *mut f32,
                     stats: *mut rt_render_stats) -> c_int;
    pub fn rt_profile_enable(ctx: *mut rt_ctx, flags: c_int) -> c_int;
    pub fn rt_resolve_u8(ctx: *mut rt_ctx, sum_rgb: *const f32, n_pixels: u64, spp: u32, out: *mut u8) -> c_int;
    pub fn rt_resolve_u8_transfer(ctx: *mut rt_ctx, sum_rgb: *const f32, n_pixels: u64, spp: u32, transfer: c_int,
                                  out: *mut u8) -> c_int;
    pub fn rt_image_write(path: *const c_char, width: u32, height: u32, rgb: *const u8, format: c_int) -> c_int;
    pub fn rt_image_read(path: *const c_char, width: *mut u32, height: *mut u32, rgb: *mut u8, cap_bytes: u64) -> c_int;
    pub fn rt_accum_write(path: *const c_char, width: u32, height: u32, samples_done: u32, seed: u32,
                          accum_rgb: *const f32) -> c_int;
    pub fn rt_accum_read(path: *const c_char, width: *mut u32, height: *mut u32, samples_done: *mut u32, seed: *mut u32,
                         accum_rgb: *mut f32, cap_floats: u64) -> c_int;
    pub fn rt_accum_write_tagged(path: *const c_char, width: u32, height: u32, samples_done: u32, seed: u32, tag: u64,
                                 accum_rgb: *const f32) -> c_int;
    pub fn rt_accum_read_tagged(path: *const c_char, width: *mut u32, height: *mut u32, samples_done: *mut u32,
                                seed: *mut u32, tag: *mut u64, accum_rgb: *mut f32, cap_floats: u64) -> c_int;
    pub fn rt_owned_pixels(width: u32, height: u32, tile_size: u32, rank: u32, nranks: u32, out: *mut u32, cap: u64,
                           n: *mut u64) -> c_int;
    // frame exchange between ranks (one RCCL all-gather per frame behind rt_gather_frame)
    pub fn rt_comm_unique_id(out: *mut u8) -> c_int;
    pub fn rt_comm_create(ctx: *mut rt_ctx, unique_id: *const u8, rank: u32, nranks: u32) -> c_int;
    pub fn rt_comm_attach(ctx: *mut rt_ctx, nccl_comm: *mut c_void, rank: u32, nranks: u32) -> c_int;
    pub fn rt_comm_destroy(ctx: *mut rt_ctx) -> c_int;
    pub fn rt_gather_frame(ctx: *mut rt_ctx, width: u32, height: u32, tile_size: u32, accum_rgb: *const f32,
                           frame_rgb: *mut f32) -> c_int;
    pub fn rt_frame_block_pixels(width: u32, height: u32, tile_size: u32, nranks: u32, out: *mut u64) -> c_int;
    pub fn rt_pack_owned(ctx: *mut rt_ctx, width: u32, height: u32, tile_size: u32, rank: u32, nranks: u32,
                         accum_rgb: *const f32, packed: *mut f32) -> c_int;
    pub fn rt_scatter_gathered(ctx: *mut rt_ctx, width: u32, height: u32, tile_size: u32, nranks: u32,
                               gathered: *const f32, frame_rgb: *mut f32) -> c_int;
}
