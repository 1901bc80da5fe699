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

/// The same render split over the GPUs of one node: one process per GPU, `rank` of `nranks`.  Every rank flattens the
/// same scene (it is replicated), renders the tiles it owns (tile index % nranks == rank, rt_render) and takes part in
/// ONE RCCL all-gather per frame (rt_gather_frame), after which every rank holds the complete sample sums -- what the
/// single Vec of main.rs:463-482 holds in the reference.  `unique_id`: the 128 bytes rank 0 got from
/// `comm_unique_id()` and sent to the other ranks over whatever channel the host program has (a file, a socket, MPI).
pub fn comm_unique_id() -> [u8; 128] {
    let mut id = [0u8; 128];
    let rc = unsafe { rt_comm_unique_id(id.as_mut_ptr()) };
    assert_eq!(rc, RT_OK, "rt_comm_unique_id failed ({rc}): no RCCL library on the loader path?");
    id
}

#[allow(clippy::too_many_arguments)]
pub fn render_ranked(world: &Scene, camera: &Camera, background: Vec3A, image_width: u32, image_height: u32,
                     samples_per_pixel: u32, max_depth: i32, device: i32, rank: u32, nranks: u32,
                     unique_id: &[u8; 128]) -> Vec<(u8, u8, u8)> {
    let mut fs = FlatScene::default();
    world.flatten(&mut fs, 0);
    let mut ctx: *mut rt_ctx = std::ptr::null_mut();
    let rc = unsafe { rt_ctx_create(device, &mut ctx) };
    assert_eq!(rc, RT_OK, "rt_ctx_create failed ({rc}) on device {device}");
    fs.upload(ctx, 0.0, 1.0).unwrap();
    // collective: every rank calls it with the same id
    check(ctx, unsafe { rt_comm_create(ctx, unique_id.as_ptr(), rank, nranks) }).unwrap();

    let tile = if nranks > 4 { 8 } else { RT_DEFAULT_TILE };  // eight ranks want the finer deal (load balance)
    let cam = camera.to_ffi();
    let params = rt_render_params { width: image_width, height: image_height, sample0: 0, sample1: samples_per_pixel,
                                    max_depth: max_depth as u32, background: background.into(), seed: 1, tile_size: tile,
                                    rank, nranks, reserved: 0 };
    let n = (image_width * image_height) as usize;
    let mut own = vec![0f32; 3 * n];     // this rank's tiles, zero elsewhere
    let mut frame = vec![0f32; 3 * n];   // the complete frame, on every rank
    let mut stats = rt_render_stats::default();
    let now = Instant::now();
    check(ctx, unsafe { rt_render(ctx, &cam, &params, own.as_mut_ptr(), &mut stats) }).unwrap();
    check(ctx, unsafe { rt_gather_frame(ctx, image_width, image_height, tile, own.as_ptr(), frame.as_mut_ptr()) }).unwrap();
    println!("rank {rank}: time elapsed: {} ({} rays)", now.elapsed().as_millis(), stats.rays);

    let mut bytes = vec![0u8; 3 * n];
    check(ctx, unsafe { rt_resolve_u8(ctx, frame.as_ptr(), n as u64, samples_per_pixel, bytes.as_mut_ptr()) }).unwrap();
    unsafe {
        rt_comm_destroy(ctx);
        rt_ctx_destroy(ctx);
    }
    bytes.chunks_exact(3).map(|p| (p[0], p[1], p[2])).collect()
}
