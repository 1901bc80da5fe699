//! Flattening of the reference's object graph (Arc<dyn Hittable> / Arc<dyn Material> / Arc<dyn Texture>)
//! into the flat arrays of rt_scene_set_*.  Same walk as the C++ mirror (raytracer-rs_amd/host/rt_scene.hpp):
//! depth first, in insertion order; every primitive gets the next id (ties at equal t go to the higher id,
//! the "later shape wins" rule of scene.rs:32-38); group 0 is the world, group g > 0 is groups[g-1].
//!
//! The three traits gain one method each:
//!     trait Hittable { ...; fn flatten(&self, fs: &mut FlatScene, group: u32);
//!                           fn as_transformed(&self) -> Option<(Mat4, Mat4, &dyn Hittable)> { None } }
//!     trait Material { ...; fn flatten(&self, fs: &mut FlatScene) -> u32; }
//!     trait Texture  { ...; fn flatten(&self, fs: &mut FlatScene) -> u32; }
//! and the impl bodies below go into the modules that own the (partly private) fields.
use std::collections::HashMap;
use std::sync::Arc;

use glam::{Mat4, Vec2, Vec3A};

use crate::ffi::*;

#[derive(Default)]
pub struct FlatScene {
    pub sph_c0r: Vec<f32>,
    pub sph_c1: Vec<f32>,
    pub sph_t01: Vec<f32>,
    pub sph_moving: Vec<u8>,
    pub sph_mat: Vec<u32>,
    pub sph_id: Vec<u32>,
    pub sph_group: Vec<u32>,
    pub rect_axis: Vec<u8>,
    pub rect_k: Vec<f32>,
    pub rect_bounds: Vec<f32>,
    pub rect_mat: Vec<u32>,
    pub rect_id: Vec<u32>,
    pub rect_group: Vec<u32>,
    pub groups: Vec<rt_group>,
    pub mats: Vec<rt_material>,
    pub texs: Vec<rt_texture>,
    pub perlin: Vec<(Vec<f32>, Vec<i32>)>, // (256 x 3 gradients, perm_x | perm_y | perm_z)
    pub images: Vec<(u32, u32, Vec<u8>)>,  // (w, h, rgb8 rows top to bottom)
    pub next_id: u32,
    mat_seen: HashMap<*const (), u32>,
    tex_seen: HashMap<*const (), u32>,
}

/// glam's column-major Mat4 -> the 3 x 4 row-major block the device uses
pub fn rows3x4(m: &Mat4) -> [f32; 12] {
    let c = m.to_cols_array();
    let mut r = [0f32; 12];
    for row in 0..3 {
        for col in 0..4 {
            r[row * 4 + col] = c[col * 4 + row];
        }
    }
    r
}

impl FlatScene {
    /// index of a shared material, flattened on first sight (Arc identity)
    pub fn material(&mut self, m: &Arc<dyn crate::materials::Material>) -> u32 {
        let key = Arc::as_ptr(m) as *const ();
        if let Some(&i) = self.mat_seen.get(&key) {
            return i;
        }
        let i = m.flatten(self);
        self.mat_seen.insert(key, i);
        i
    }
    pub fn texture(&mut self, t: &Arc<dyn crate::texture::Texture>) -> u32 {
        let key = Arc::as_ptr(t) as *const ();
        if let Some(&i) = self.tex_seen.get(&key) {
            return i;
        }
        let i = t.flatten(self);
        self.tex_seen.insert(key, i);
        i
    }
    fn id(&mut self) -> u32 {
        let i = self.next_id;
        self.next_id += 1;
        i
    }
    pub fn add_sphere(&mut self, c0: Vec3A, r: f32, mat: u32, group: u32, moving: Option<(Vec3A, f32, f32)>) {
        self.sph_c0r.extend_from_slice(&[c0.x, c0.y, c0.z, r]);
        let (c1, t0, t1) = moving.unwrap_or((Vec3A::ZERO, 0.0, 0.0));
        self.sph_c1.extend_from_slice(&[c1.x, c1.y, c1.z]);
        self.sph_t01.extend_from_slice(&[t0, t1]);
        self.sph_moving.push(moving.is_some() as u8);
        self.sph_mat.push(mat);
        let id = self.id();
        self.sph_id.push(id);
        self.sph_group.push(group);
    }
    /// axis: 0 = XY (z = k), 1 = XZ (y = k), 2 = YZ (x = k); min/max exactly as the Rust constructors take them
    pub fn add_rect(&mut self, axis: u8, min: Vec2, max: Vec2, k: f32, mat: u32, group: u32) {
        self.rect_axis.push(axis);
        self.rect_k.push(k);
        self.rect_bounds.extend_from_slice(&[min.x, min.y, max.x, max.y]);
        self.rect_mat.push(mat);
        let id = self.id();
        self.rect_id.push(id);
        self.rect_group.push(group);
    }
    /// parent: 0, or the group (1 + index) of the TransformedObject this one stands inside
    fn add_group(&mut self, kind: u32, xform: Option<(Mat4, Mat4)>, neg_inv_density: f64, mat: u32, parent: u32) -> u32 {
        let mut g = rt_group { kind, neg_inv_density, mat, parent, ..Default::default() };
        if let Some((m, minv)) = xform {
            g.has_xform = 1;
            g.m = rows3x4(&m);
            g.minv = rows3x4(&minv);
        }
        self.groups.push(g);
        self.groups.len() as u32
    }

    /// rt_scene_set_* + rt_scene_commit (stands in for BVHNode::from_scene(&world, t0, t1), main.rs:458)
    pub fn upload(&self, ctx: *mut rt_ctx, t0: f32, t1: f32) -> Result<(), String> {
        let opt = |v: &Vec<u8>| if v.iter().any(|&m| m != 0) { v.as_ptr() } else { std::ptr::null() };
        let moving = opt(&self.sph_moving);
        unsafe {
            check(ctx, rt_scene_clear(ctx))?;
            check(ctx, rt_scene_set_materials(ctx, self.mats.len() as u32, self.mats.as_ptr()))?;
            check(ctx, rt_scene_set_textures(ctx, self.texs.len() as u32, self.texs.as_ptr()))?;
            for (i, (vec, perm)) in self.perlin.iter().enumerate() {
                check(ctx, rt_scene_set_perlin(ctx, i as u32, vec.as_ptr(), perm.as_ptr()))?;
            }
            for (i, (w, h, rgb)) in self.images.iter().enumerate() {
                check(ctx, rt_scene_set_image(ctx, i as u32, *w, *h, rgb.as_ptr()))?;
            }
            check(ctx, rt_scene_set_groups(ctx, self.groups.len() as u32, self.groups.as_ptr()))?;
            check(ctx, rt_scene_set_spheres(ctx, self.sph_mat.len() as u32, self.sph_c0r.as_ptr(),
                                            if moving.is_null() { std::ptr::null() } else { self.sph_c1.as_ptr() },
                                            if moving.is_null() { std::ptr::null() } else { self.sph_t01.as_ptr() },
                                            moving, self.sph_mat.as_ptr(), self.sph_id.as_ptr(), self.sph_group.as_ptr()))?;
            check(ctx, rt_scene_set_rects(ctx, self.rect_mat.len() as u32, self.rect_axis.as_ptr(), self.rect_k.as_ptr(),
                                          self.rect_bounds.as_ptr(), self.rect_mat.as_ptr(), self.rect_id.as_ptr(),
                                          self.rect_group.as_ptr()))?;
            check(ctx, rt_scene_commit(ctx, t0, t1))
        }
    }
}

pub fn check(ctx: *mut rt_ctx, rc: i32) -> Result<(), String> {
    if rc == RT_OK {
        return Ok(());
    }
    let msg = unsafe {
        let p = rt_last_error(ctx);
        if p.is_null() { String::new() } else { std::ffi::CStr::from_ptr(p).to_string_lossy().into_owned() }
    };
    Err(format!("librtamd error {rc}: {msg}"))
}

// ---------------------------------------------------------------------------------------------------
// impl bodies, by the module they go into
// ---------------------------------------------------------------------------------------------------

// ---- shapes.rs ----
// impl Hittable for Sphere (shapes.rs:116):
//     fn flatten(&self, fs: &mut FlatScene, group: u32) {
//         let mat = fs.material(&self.material);
//         fs.add_sphere(self.center, self.radius, mat, group, None);
//     }
// impl Hittable for MovingSphere (shapes.rs:193):
//     fn flatten(&self, fs: &mut FlatScene, group: u32) {
//         let mat = fs.material(&self.material);
//         fs.add_sphere(self.center_start, self.radius, mat, group,
//                       Some((self.center_end, self.time_start, self.time_end)));
//     }
// impl Hittable for XYRect / XZRect / YZRect (shapes.rs:278, 342, 406) with axis 0 / 1 / 2:
//     fn flatten(&self, fs: &mut FlatScene, group: u32) {
//         let mat = fs.material(&self.material);
//         fs.add_rect(AXIS, self.min, self.max, self.offset, mat, group);
//     }
// impl Hittable for Box2 (shapes.rs:470): the six sides in the order shapes.rs:459-466 adds them
//     fn flatten(&self, fs: &mut FlatScene, group: u32) { self.sides.flatten(fs, group) }
// impl<T: Hittable> Hittable for TransformedObject<T> (shapes.rs:64):
//     fn flatten(&self, fs: &mut FlatScene, group: u32) {
//         // inside another TransformedObject: parent = that group (the library takes up to 4 stacked instances and
//         // refuses one inside a medium's boundary with RT_ERR_UNSUPPORTED at rt_scene_commit)
//         let g = fs.add_group(RT_GROUP_INSTANCE, Some((self.transform, self.inv_transform)), 0.0, 0, group);
//         self.object.flatten(fs, g);
//     }
//     fn as_transformed(&self) -> Option<(Mat4, Mat4, &dyn Hittable)> {
//         Some((self.transform, self.inv_transform, &self.object))
//     }
// impl<T: Hittable> Hittable for ConstantMedium<T> (shapes.rs:648): the boundary's primitives become the
// group's members; a TransformedObject boundary hands its matrices to the group (shapes.rs:226-235 wraps
// Transformed(Box2)); the boundary's own material is never used (shapes.rs:690-697)
//     fn flatten(&self, fs: &mut FlatScene, group: u32) {
//         // group != 0: the medium stands inside TransformedObjects and names the innermost one as its parent (the
//         // library refuses a medium inside another medium's boundary with RT_ERR_UNSUPPORTED at rt_scene_commit)
//         let mat = fs.material(&self.material);                      // the Isotropic phase function
//         match self.boundary.as_transformed() {
//             Some((m, minv, inner)) => { let g = fs.add_group(RT_GROUP_MEDIUM, Some((m, minv)), self.negative_density, mat, group);
//                                         inner.flatten(fs, g); }
//             None => { let g = fs.add_group(RT_GROUP_MEDIUM, None, self.negative_density, mat, group);
//                       self.boundary.flatten(fs, g); }
//         }
//     }
//
// ---- scene.rs ----
// impl Hittable for Scene (scene.rs:26):
//     fn flatten(&self, fs: &mut FlatScene, group: u32) { for s in &self.shapes { s.flatten(fs, group) } }
//
// ---- bvh_node.rs ----
// BVHNode gains `objects: Vec<Arc<dyn Hittable>>`, a clone of the vector from_vector receives
// (bvh_node.rs:52-53 clones it anyway); the device builds its own tree over the primitives:
//     fn flatten(&self, fs: &mut FlatScene, group: u32) { for o in &self.objects { o.flatten(fs, group) } }
//
// ---- materials.rs ----
// impl Material for LambertianMat:  fn flatten(&self, fs) -> u32 { let tex = fs.texture(&self.albedo);
//         fs.mats.push(rt_material { kind: RT_MAT_LAMBERTIAN, tex, ..Default::default() }); fs.mats.len() as u32 - 1 }
// impl Material for MetalMat:       fn flatten(&self, fs) -> u32 { fs.mats.push(rt_material { kind: RT_MAT_METAL,
//         albedo: self.albedo.into(), fuzz: self.fuzz /* already clamped by MetalMat::new, materials.rs:70-81 */,
//         ..Default::default() }); fs.mats.len() as u32 - 1 }
// impl Material for DielectricMat:  fn flatten(&self, fs) -> u32 { fs.mats.push(rt_material { kind: RT_MAT_DIELECTRIC,
//         ior: self.index_refraction, ..Default::default() }); fs.mats.len() as u32 - 1 }
// impl Material for DiffuseLight:   as LambertianMat with kind RT_MAT_DIFFUSE_LIGHT and self.emit
// impl Material for IsotropicMat:   as LambertianMat with kind RT_MAT_ISOTROPIC and self.albedo
// impl Material for NoMaterial:     never reached by a flattened scene (only the boundary of a ConstantMedium
//                                   carries it, and that material is not flattened): unreachable!()
//
// ---- texture.rs ----
// impl Texture for SolidColor:       fs.texs.push(rt_texture { kind: RT_TEX_SOLID, color: self.color.into(), ..Default::default() })
// impl Texture for CheckeredTexture: let (o, e) = (fs.texture(&self.odd), fs.texture(&self.even));
//                                    fs.texs.push(rt_texture { kind: RT_TEX_CHECKER, odd: o, even: e, ..Default::default() })
// impl Texture for NoiseTexture:     let mut vec = Vec::with_capacity(768);
//                                    for v in &self.noise.rand_vec { vec.extend_from_slice(&[v.x, v.y, v.z]); }   // perlin.rs:13
//                                    let perm = [&self.noise.perm_x[..], &self.noise.perm_y[..], &self.noise.perm_z[..]].concat();
//                                    fs.perlin.push((vec, perm));
//                                    fs.texs.push(rt_texture { kind: RT_TEX_NOISE, freq: self.frequency,
//                                                              table: fs.perlin.len() as u32 - 1, ..Default::default() })
// impl Texture for ImageTexture:     fs.images.push((self.image.width(), self.image.height(), self.image.as_raw().clone()));
//                                    fs.texs.push(rt_texture { kind: RT_TEX_IMAGE, table: fs.images.len() as u32 - 1, ..Default::default() })
// each returns fs.texs.len() as u32 - 1.
//
// ---- camera.rs ----
// impl Camera { pub fn to_ffi(&self) -> rt_camera { rt_camera { origin: self.origin.into(),
//         lower_left_corner: self.lower_left_corner.into(), horizontal: self.horizontal.into(),
//         vertical: self.vertical.into(), u: self.u.into(), v: self.v.into(), w: self.w.into(),
//         lens_radius: self.lens_radius, time_start: self.time_start, time_end: self.time_end } } }
