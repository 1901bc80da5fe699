//! Replacement for main.rs:458-482 (BVH build + rayon pixel loop): flatten, upload, render on the GPU,
//! quantise.  Returns the pixels in the shape main.rs:486-497 goes on to interleave and save.
use std::time::Instant;

use glam::Vec3A;

use crate::camera::Camera;
use crate::ffi::*;
use crate::flatten::{check, FlatScene};
use crate::scene::Scene;
use crate::shapes::Hittable;

pub fn render(world: &Scene, camera: &Camera, background: Vec3A, image_width: u32, image_height: u32,
              samples_per_pixel: u32, max_depth: i32) -> Vec<(u8, u8, u8)> {
    let mut fs = FlatScene::default();
    world.flatten(&mut fs, 0);
    let mut ctx: *mut rt_ctx = std::ptr::null_mut();
    let rc = unsafe { rt_ctx_create(0, &mut ctx) };
    assert_eq!(rc, RT_OK, "rt_ctx_create failed ({rc}): no usable MI355X?");
    fs.upload(ctx, 0.0, 1.0).unwrap();                       // shutter interval of main.rs:458

    let cam = camera.to_ffi();
    let params = rt_render_params { width: image_width, height: image_height, sample0: 0, sample1: samples_per_pixel,
                                    max_depth: max_depth as u32, background: background.into(), seed: 1, tile_size: 16,
                                    rank: 0, nranks: 1, reserved: 0 };
    let n = (image_width * image_height) as usize;
    let mut sum = vec![0f32; 3 * n];
    let mut stats = rt_render_stats::default();
    let now = Instant::now();
    check(ctx, unsafe { rt_render(ctx, &cam, &params, sum.as_mut_ptr(), &mut stats) }).unwrap();
    println!("Time elapsed: {}", now.elapsed().as_millis());  // same span as main.rs:462,484
    println!("{} rays, {:.1} Mrays/s", stats.rays, stats.rays as f64 / stats.render_ms / 1e3);

    let mut bytes = vec![0u8; 3 * n];                         // vec3_to_rgb(color * inv_samples), main.rs:481
    check(ctx, unsafe { rt_resolve_u8(ctx, sum.as_ptr(), n as u64, samples_per_pixel, bytes.as_mut_ptr()) }).unwrap();
    unsafe { rt_ctx_destroy(ctx) };
    bytes.chunks_exact(3).map(|p| (p[0], p[1], p[2])).collect()
}
