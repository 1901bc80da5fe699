/*
 * rtamd.h -- C ABI of the MI355X render hot path for raytracer-rs scenes.
 *
 * The reference (jameslen/raytracer-rs) is a binary-only Rust crate with no FFI
 * of its own; every entry point below therefore names the reference code it
 * stands in for (paths relative to raytracer-rs/src/).  All functions return 0
 * (RT_OK) or a negative rt_status; none aborts or throws across the boundary.
 * rt_last_error() gives the text of the last failure on a context.
 *
 * Pointers marked "host or device" are probed with hipPointerGetAttributes.
 * One rt_ctx owns one GPU and one HIP stream and must not be used from two
 * host threads at once; different contexts are independent.
 */
#ifndef RTAMD_H
#define RTAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ctx rt_ctx;

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,       /* bad argument */
    RT_ERR_HIP = -2,           /* a HIP runtime call failed */
    RT_ERR_UNSUPPORTED = -3,   /* scene shape the device path does not handle */
    RT_ERR_NOT_COMMITTED = -4, /* work call before rt_scene_commit */
    RT_ERR_NO_DEVICE = -5,     /* no usable GPU */
    RT_ERR_IO = -6             /* a file could not be written or read */
} rt_status;

/* ---- context ----------------------------------------------------------- */
int rt_ctx_create(int device, rt_ctx** out);
void rt_ctx_destroy(rt_ctx* ctx);
const char* rt_last_error(const rt_ctx* ctx);
/* Use an existing hipStream_t (e.g. torch's current stream); NULL = the context's own stream. */
int rt_ctx_set_stream(rt_ctx* ctx, void* hip_stream);
/* Upper bound on paths in flight per wavefront batch (default 512Mi paths = up to 72 GB of wavefront buffers,
 * never more than 60% of the free device memory; env RTAMD_MAX_PATHS_MI).  A render allocates only what it needs. */
int rt_ctx_set_max_paths(rt_ctx* ctx, uint64_t max_paths);
/* Allocates the path queues for renders of up to n_paths camera samples (pixels x samples of one call) now, so
 * that the first rt_render does not pay for it.  Optional: rt_render grows the buffers itself when it has to. */
int rt_ctx_reserve(rt_ctx* ctx, uint64_t n_paths);
int rt_ctx_synchronize(rt_ctx* ctx);
/* Kernel pipeline of rt_render: -1 = pick per scene (default), 0 = split (ray generation, then a trace and
 * a shade kernel per bounce), 1 = fused (one kernel per bounce), 2 = hybrid (bounce 0 fused, later bounces
 * split), 3 = whole paths (one launch per batch: every lane follows a path to its end and then starts the next
 * one; no path queues).  All produce the same bits.  -1 takes 3 for small scenes traced without a tree and with
 * cheap shading, 1 for other scenes of up to 64 nodes, 0 for trees whose nodes outgrow the 256 MB Infinity Cache
 * and 2 for everything in between. */
int rt_ctx_set_pipeline(rt_ctx* ctx, int pipeline);
/* Once a batch has at most tail_paths live paths, one kernel launch walks each of them to its end
 * instead of one launch per bounce (default 262144, env RTAMD_TAIL_PATHS; 0 = never).  The image does not depend on it. */
int rt_ctx_set_tail_paths(rt_ctx* ctx, uint64_t tail_paths);

/* ---- scene description --------------------------------------------------
 * Replaces the object graph of Arc<dyn Hittable>/Arc<dyn Material>/Arc<dyn Texture>
 * (shapes.rs:14-17, materials.rs:13-18, texture.rs:10-12) by flat arrays.
 * Every primitive carries
 *   id    : its position in a depth-first walk of the scene in insertion order
 *           (Scene.shapes order, Box2 sides in shapes.rs:459-466 order); reported by
 *           rt_intersect_batch and used to break exact t ties (higher id wins, the
 *           "later shape wins" rule of scene.rs:32-38);
 *   group : 0 = top level world, g > 0 = member of groups[g-1].
 * A group is what TransformedObject<T> (shapes.rs:26-97) or ConstantMedium<T>
 * (shapes.rs:624-705) wraps.  Groups may stand inside instance groups (rt_group.parent): transformed objects
 * inside transformed objects, up to four deep, and media inside them. */
typedef enum rt_material_kind {
    RT_MAT_LAMBERTIAN = 0,    /* materials.rs:45-61  : tex */
    RT_MAT_METAL = 1,         /* materials.rs:84-95  : albedo, fuzz (caller clamps to <=1 as MetalMat::new does) */
    RT_MAT_DIELECTRIC = 2,    /* materials.rs:117-148: ior */
    RT_MAT_DIFFUSE_LIGHT = 3, /* materials.rs:176-184: tex */
    RT_MAT_ISOTROPIC = 4      /* materials.rs:204-215: tex */
} rt_material_kind;

typedef struct rt_material {
    uint32_t kind;
    uint32_t tex;
    float albedo[3];
    float fuzz;
    float ior;
    uint32_t reserved;
} rt_material;

typedef enum rt_texture_kind {
    RT_TEX_SOLID = 0,   /* texture.rs:18-22 : color */
    RT_TEX_CHECKER = 1, /* texture.rs:52-62 : odd, even = texture indices */
    RT_TEX_NOISE = 2,   /* texture.rs:78-82 : freq, table = perlin table id */
    RT_TEX_IMAGE = 3    /* texture.rs:96-117: table = image id */
} rt_texture_kind;

typedef struct rt_texture {
    uint32_t kind;
    float color[3];
    uint32_t odd, even;
    float freq;
    uint32_t table;
} rt_texture;

typedef enum rt_group_kind { RT_GROUP_INSTANCE = 1, RT_GROUP_MEDIUM = 2 } rt_group_kind;

typedef struct rt_group {
    uint32_t kind;           /* rt_group_kind */
    uint32_t has_xform;      /* 0: members are in world space */
    float m[12];             /* object->world, 3 rows x 4 columns (row major) */
    float minv[12];          /* world->object */
    double neg_inv_density;  /* media: -1/density in f64 (shapes.rs:642) */
    uint32_t mat;            /* media: the Isotropic phase material */
    /* 0: the group stands in the world.  Otherwise 1 + index of the instance group it stands inside
     * (TransformedObject<T> takes any Hittable, shapes.rs:26-41: a transformed object inside a transformed object).
     * The enclosing group comes earlier in the array and is an instance; at most 4 instances are stacked.  A
     * primitive's `group` names the innermost instance around it.  A medium may stand inside instances (it is then
     * handed their ray); nothing may stand inside a medium beyond the one transform of its boundary (RT_ERR_UNSUPPORTED). */
    uint32_t parent;
} rt_group;

int rt_scene_clear(rt_ctx* ctx);
/* Sphere / MovingSphere (shapes.rs:99-250).  c0_r: n x (cx,cy,cz,r).  c1: n x 3 and t01: n x (t0,t1)
 * and moving: n flags may be NULL when nothing moves. */
int rt_scene_set_spheres(rt_ctx* ctx, uint32_t n, const float* c0_r, const float* c1, const float* t01,
                         const uint8_t* moving, const uint32_t* mat, const uint32_t* id, const uint32_t* group);
/* XYRect / XZRect / YZRect (shapes.rs:252-442).  axis: 0 = XY (z = k), 1 = XZ (y = k), 2 = YZ (x = k);
 * bounds: n x (min.x, min.y, max.x, max.y) exactly as the Rust constructors take them. */
int rt_scene_set_rects(rt_ctx* ctx, uint32_t n, const uint8_t* axis, const float* k, const float* bounds,
                       const uint32_t* mat, const uint32_t* id, const uint32_t* group);
int rt_scene_set_groups(rt_ctx* ctx, uint32_t n, const rt_group* groups);
int rt_scene_set_materials(rt_ctx* ctx, uint32_t n, const rt_material* mats);
int rt_scene_set_textures(rt_ctx* ctx, uint32_t n, const rt_texture* texs);
/* Perlin tables (perlin.rs:12-44): vec = 256 x 3 gradients, perm = perm_x|perm_y|perm_z, 3 x 256. */
int rt_scene_set_perlin(rt_ctx* ctx, uint32_t table, const float* vec, const int32_t* perm);
/* Decoded RGB8 image, row 0 = top (texture.rs:84-93). */
int rt_scene_set_image(rt_ctx* ctx, uint32_t table, uint32_t w, uint32_t h, const uint8_t* rgb);
/* Stands in for BVHNode::from_scene(&world, t0, t1) (bvh_node.rs:48-120, main.rs:458): builds the
 * linear two-level BVH and uploads everything.  Instances' own trees are built on the host; the world-level
 * tree is built on the host (binned SAH) or, for large worlds, on the device (Morton-sorted linear BVH). */
int rt_scene_commit(rt_ctx* ctx, float shutter_t0, float shutter_t1);
/* Builder of the world-level tree at the next rt_scene_commit: -1 = by size (default: device from 262144 world
 * primitives on, env RTAMD_GPU_BUILD_MIN), 0 = host, 1 = device (env RTAMD_BUILDER).  Every valid tree gives
 * the same hits; the builders differ in build time and in node visits per ray. */
int rt_ctx_set_builder(rt_ctx* ctx, int builder);
/* Children per inner node the kernels walk from the next rt_scene_commit on: 2 (64-byte nodes holding both child
 * boxes), 4 (the finished tree, whichever builder made it, is also written in a 128-byte four-child form -- the
 * grandchildren of every node -- which halves the chain of dependent node fetches per ray) or -1 = by tree size
 * (default: 4 from 65536 nodes on, i.e. once the nodes outgrow one XCD's L2; env RTAMD_BVH_WIDTH, RTAMD_WIDE_MIN_NODES).
 * The reference's tree is binary (bvh_node.rs:26-30); the width changes no hit.  A tree whose four-child walk could
 * need more than 64 stack entries keeps the binary walk (rt_scene_info.bvh_width tells which one is in use). */
int rt_ctx_set_bvh_width(rt_ctx* ctx, int width);

/* Six rects that are exactly what Box2::full_box emits (shapes.rs:456-466: consecutive ids, one material, side
 * order XY(min.z), XY(max.z), XZ(min.y), XZ(max.y), YZ(min.x), YZ(max.x), bounds equal bit for bit to the corners)
 * are kept as ONE record and one BVH leaf from the next rt_scene_commit on (default on; env RTAMD_FUSE_BOXES=0).
 * The kernels still run the six rect tests in that order with the reference's arithmetic, so hits, reported ids
 * and ties do not change; the tree gets two to three levels shallower where boxes dominate.  Medium boundaries
 * keep their rects. */
int rt_ctx_set_fuse_boxes(rt_ctx* ctx, int on);

typedef struct rt_scene_info {
    uint64_t n_prims, n_nodes, n_groups, n_media;
    uint32_t max_depth;    /* deepest stack the traversal can need */
    uint32_t stack_entries; /* traversal stack entries per lane the kernels are launched with (1 for flat scenes) */
    uint64_t device_bytes;
    double build_seconds;  /* all of rt_scene_commit */
    double tree_seconds;   /* of which: the world-level tree (build + upload of nodes and primitives) */
    uint32_t builder;      /* 0 host, 1 device */
    uint32_t bvh_width;    /* children per inner node the kernels walk: 2 or 4 */
    uint32_t crown_nodes;  /* inner nodes the persistent trace kernel keeps in LDS (the whole tree of a small scene), 0 = none */
    uint32_t reserved;
} rt_scene_info;
int rt_scene_get_info(rt_ctx* ctx, rt_scene_info* out);
/* Inspection (tests, tools): copies the committed tree to host memory.  nodes: n_nodes records of 64 bytes
 * {float lo0[3], hi0[3], lo1[3], hi1[3]; uint32 c0, c1, pad[2]} (both child boxes, two child references);
 * prims: n_prims records of 32 bytes {float a[4], b0; uint32 info, id, aux} in leaf order.  A reference has
 * its type in bits 31..30 (0 inner node, 1 primitive range: bits 29..26 = count-1, bits 25..0 = first,
 * 2 instance group); 0xFFFFFFFF = empty.  Either pointer may be NULL. */
int rt_scene_read_bvh(rt_ctx* ctx, void* nodes, uint64_t nodes_bytes, void* prims, uint64_t prims_bytes,
                      uint32_t* root_ref);

/* ---- camera (camera.rs:14-25; built by Camera::new, camera.rs:28-52) ------ */
/* time_start..time_end is the interval Camera::get_ray draws ray times from (camera.rs:63).  When the scene has
 * moving spheres it must lie inside the shutter interval given to rt_scene_commit, whose boxes cover only that
 * interval (rt_render returns RT_ERR_INVALID otherwise). */
typedef struct rt_camera {
    float origin[3];
    float lower_left_corner[3];
    float horizontal[3];
    float vertical[3];
    float u[3], v[3], w[3];
    float lens_radius;
    float time_start, time_end;
} rt_camera;

/* ---- hit() path on a ray batch -------------------------------------------
 * world.intersect(ray, tmin, tmax) (bvh_node.rs:145-164 / scene.rs:27-41) for n rays given as
 * seven arrays.  Outputs (each may be NULL): prim (>= 0 primitive id, -1 miss, <= -2 medium
 * -(k+2) for the k-th medium group), inst (index of the instance group hit, else -1), t,
 * normal (n x 3, face-set as hit_record.rs:32-42), front, uv (n x 2), point (n x 3).
 * Media take their free-flight sample from RNG stream (seed, pixel = ray index, sample 0), bounce 0.
 * All pointers: host or device, all of one kind (a mix is RT_ERR_INVALID). */
int rt_intersect_batch(rt_ctx* ctx, uint64_t n, const float* ox, const float* oy, const float* oz,
                       const float* dx, const float* dy, const float* dz, const float* time, float tmin,
                       float tmax, uint32_t seed, int32_t* prim, int32_t* inst, float* t, float* normal,
                       uint8_t* front, float* uv, float* point);

/* The same call for callers that must not allocate (tight serving loops, harnesses that forbid hipMalloc, processes
 * whose device memory is spoken for).  It has no counterpart in the reference.  rt_intersect_batch allocates: its
 * overflow flag on first use, and staging buffers on every call that passes host arrays.  This variant keeps both in a
 * static device array of 5 MiB (256 bytes of flags + 76 bytes of staging per ray) that each GPU the library runs on
 * holds until the process ends, and calls no hipMalloc / hipFree.  Like rt_intersect_batch it waits for its stream
 * and reads the overflow flag back before it returns, so it cannot be recorded into a HIP graph.  Limit: n <= 65536
 * rays per call (RT_ERR_INVALID beyond it); same arguments, same results bit for bit.  Every call on a GPU shares that
 * GPU's array: calls that overlap on one GPU (two contexts on two host threads) race -- serialise them. */
#define RT_NOALLOC_MAX_RAYS 65536u
int rt_intersect_batch_noalloc(rt_ctx* ctx, uint64_t n, const float* ox, const float* oy, const float* oz,
                               const float* dx, const float* dy, const float* dz, const float* time, float tmin,
                               float tmax, uint32_t seed, int32_t* prim, int32_t* inst, float* t, float* normal,
                               uint8_t* front, float* uv, float* point);

/* ---- render loop ------------------------------------------------------------
 * The pixel/sample loop of main.rs:463-482 over ray_color (main.rs:44-63) as a wavefront
 * path tracer.  Adds samples [sample0, sample1) of every pixel this rank owns to
 * accum_rgb (height x width x 3 floats, row 0 = top; host or device), one sample at a
 * time in sample order, so that splitting a render by samples or by ranks changes no bit.
 * Ownership: tile_size x tile_size tiles dealt round-robin, tile index % nranks == rank.  The tile size also sets the
 * order in which a rank's pixels are walked (a wave's 64 camera rays: 4 rows of a 16-pixel tile).  Measured on
 * final_scene 800x800: 16 is 2 % faster than 32 on one GPU and splits the frame evenly over 2 or 4 ranks (32: one
 * rank of two gets 8 % more work, the expensive objects falling into its tiles); 8 evens out 8 ranks. */
#define RT_DEFAULT_TILE 16u
typedef struct rt_render_params {
    uint32_t width, height;
    uint32_t sample0, sample1;
    uint32_t max_depth; /* 1..253; 50 in every reference preset */
    float background[3];
    uint32_t seed;
    uint32_t tile_size; /* 0 -> RT_DEFAULT_TILE */
    uint32_t rank, nranks; /* nranks 0 -> 1 */
    uint32_t reserved;
} rt_render_params;

typedef struct rt_render_stats {
    uint64_t rays;        /* world.intersect calls issued by ray_color (primary + scattered) */
    uint64_t paths;       /* camera samples */
    uint32_t batches;
    uint32_t bounce_launches;
    double render_ms;     /* wall time inside rt_render, device-synchronised */
    double trace_ms, shade_ms, raygen_ms, accumulate_ms; /* HIP-event kernel time (profile flag 1) */
    uint64_t trace_launches, shade_launches;
    uint64_t node_visits; /* inner BVH nodes fetched   (profile flag 2, instrumented trace kernel) */
    uint64_t prim_tests;  /* primitive records tested  (profile flag 2) */
    double bounce_ms;     /* fused pipeline: HIP-event time of the bounce kernel (profile flag 1) */
    uint64_t bounce_kernel_launches;
    uint32_t pipeline;    /* the pipeline used: 0 split, 1 fused, 2 hybrid (see rt_ctx_set_pipeline) */
    uint32_t tail_launches; /* batches whose last live paths were finished by the single-launch tail kernel */
    uint64_t tail_rays;   /* rays traced by the tail kernel (included in rays) */
    double tail_ms;       /* HIP-event time of the tail kernel (profile flag 1) */
    double paths_ms;      /* pipeline 3: HIP-event time of the whole-path kernel (profile flag 1) */
    uint64_t paths_launches;
    uint64_t settled_rays; /* rays whose hit in the first medium was found by the bounce before them, without the tree walk (included in rays) */
    double settled_ms;     /* HIP-event time of the kernel that follows those paths in registers (profile flag 1) */
    uint64_t settled_launches;
    uint64_t settled_entries; /* path-queue entries that kernel read (all it moves through memory besides the paths that leave it) */
} rt_render_stats;

int rt_render(rt_ctx* ctx, const rt_camera* cam, const rt_render_params* params, float* accum_rgb,
              rt_render_stats* stats);
/* flags: 1 = time every kernel launch with HIP events on the context's stream; 2 = run the instrumented
 * trace kernel that counts node fetches and primitive tests (slower).  0 = off (default). */
int rt_profile_enable(rt_ctx* ctx, int flags);

/* vec3_to_rgb(color * inv_samples) (main.rs:307-313,460,481): out[i] = (256*clamp(sum[i]/spp,0,0.999)) as u8.
 * sum_rgb and out: host or device. */
int rt_resolve_u8(rt_ctx* ctx, const float* sum_rgb, uint64_t n_pixels, uint32_t spp, uint8_t* out);

/* The same with a transfer curve: RT_TRANSFER_LINEAR is rt_resolve_u8 (what main.rs:307-313 does);
 * RT_TRANSFER_SQRT takes the square root of the mean first -- the gamma-2 write_color the reference keeps
 * commented out (main.rs:35-42): (256*clamp(sqrt(sum[i]/spp),0,0.999)) as u8. */
typedef enum rt_transfer { RT_TRANSFER_LINEAR = 0, RT_TRANSFER_SQRT = 1 } rt_transfer;
int rt_resolve_u8_transfer(rt_ctx* ctx, const float* sum_rgb, uint64_t n_pixels, uint32_t spp, int transfer, uint8_t* out);

/* ---- output files (host only; no GPU needed) -----------------------------------------
 * Stands in for main.rs:486-497 (interleave + image::save("./test.png")): writes height x width x 3 bytes,
 * row 0 = top, as an 8-bit RGB PNG or a binary PPM (P6). */
typedef enum rt_image_format { RT_IMAGE_PNG = 0, RT_IMAGE_PPM = 1 } rt_image_format;
int rt_image_write(const char* path, uint32_t width, uint32_t height, const uint8_t* rgb, int format);
/* ---- texture files (host only; no GPU needed) -----------------------------------------
 * Stands in for ImageTexture::new (texture.rs:89-93: image::open(path).unwrap().into_rgb8()): decodes a baseline
 * JPEG (8 bit, Huffman; greyscale or YCbCr with chroma sampled 1x1, 2x1 or 2x2; restart markers) or a binary PPM
 * into height x width x 3 bytes, row 0 = top -- the layout rt_scene_desc's images take.  With rgb == NULL only
 * *width and *height are set; otherwise cap_bytes must hold width*height*3.  A missing file is RT_ERR_IO, a format
 * outside the above (progressive or arithmetic-coded JPEG, PNG, ...) RT_ERR_UNSUPPORTED where the reference panics.
 * The decode follows libjpeg's default arithmetic (accurate integer IDCT, triangle-filter chroma upsampling). */
int rt_image_read(const char* path, uint32_t* width, uint32_t* height, uint8_t* rgb, uint64_t cap_bytes);
/* Checkpoint of a progressive render (the reference has none; its render is all-or-nothing): the f32 sample
 * sums (host, height x width x 3) together with the number of samples they hold and the seed.  Continue with
 * rt_render(sample0 = samples_done, ...) into the loaded buffer: the result equals the uninterrupted render
 * bit for bit.  rt_accum_read with accum_rgb == NULL only reads the header. */
int rt_accum_write(const char* path, uint32_t width, uint32_t height, uint32_t samples_done, uint32_t seed,
                   const float* accum_rgb);
int rt_accum_read(const char* path, uint32_t* width, uint32_t* height, uint32_t* samples_done, uint32_t* seed,
                  float* accum_rgb, uint64_t cap_floats);
/* The same with a caller-supplied 64-bit tag (a hash of whatever must not change between the two halves of a
 * render: scene, camera, max_depth, background) stored in the file and returned for comparison, since the sums
 * only continue bit-exactly under the same scene and camera.  rt_accum_write stores tag 0.  The file is written
 * to "<path>.tmp", flushed to the disk and renamed over the target, so an interruption never destroys the
 * previous checkpoint. */
int rt_accum_write_tagged(const char* path, uint32_t width, uint32_t height, uint32_t samples_done, uint32_t seed,
                          uint64_t tag, const float* accum_rgb);
int rt_accum_read_tagged(const char* path, uint32_t* width, uint32_t* height, uint32_t* samples_done, uint32_t* seed,
                         uint64_t* tag, float* accum_rgb, uint64_t cap_floats);

/* Pixels owned by `rank` under the tile deal of rt_render, in render order (tile-major). Writes up to
 * cap pixel indices (y*width+x) to out (host) and returns the count in *n. */
int rt_owned_pixels(uint32_t width, uint32_t height, uint32_t tile_size, uint32_t rank, uint32_t nranks,
                    uint32_t* out, uint64_t cap, uint64_t* n);

/* ---- frame exchange between ranks ---------------------------------------------------------
 * The reference collects every pixel of the frame in one Vec (main.rs:463-482).  With the frame's tiles dealt over
 * the ranks (rt_render: rank, nranks, tile_size) each rank's accumulation buffer holds only its own tiles; ONE
 * collective per frame completes it everywhere: every rank packs its pixels (in rt_owned_pixels order, padded to the
 * largest share), one ncclAllGather over the RCCL communicator of the context (each peer's block arrives over its
 * own xGMI link), and a scatter back to frame order.  RCCL is not linked into the library: its entry points are
 * taken from the RCCL the process has loaded (or librccl.so on the loader path), so a communicator the host program
 * made with its own RCCL binding can be adopted (rt_comm_attach) as well as one made here (rt_comm_create).
 *   rank 0:     rt_comm_unique_id(id);  ... send the 128 bytes to every rank by any means ...
 *   every rank: rt_comm_create(ctx, id, rank, nranks);          (collective: all ranks must call it)
 *   per frame:  rt_render(..., rank, nranks, ...);  rt_gather_frame(ctx, w, h, tile_size, accum, frame);
 * rt_gather_frame with a single rank needs no communicator and no RCCL.  accum_rgb, frame_rgb: height x width x 3
 * floats, host or device, may not overlap. */
int rt_comm_unique_id(uint8_t out[128]);
int rt_comm_create(rt_ctx* ctx, const uint8_t unique_id[128], uint32_t rank, uint32_t nranks);
/* Adopts an ncclComm_t the caller made (not destroyed with the context); NULL with nranks == 1 detaches. */
int rt_comm_attach(rt_ctx* ctx, void* nccl_comm, uint32_t rank, uint32_t nranks);
int rt_comm_destroy(rt_ctx* ctx);
int rt_gather_frame(rt_ctx* ctx, uint32_t width, uint32_t height, uint32_t tile_size, const float* accum_rgb,
                    float* frame_rgb);
/* The two device steps on their own (device pointers; for callers that run the collective themselves, and for
 * tests with several virtual ranks on one GPU).  A rank's block holds rt_frame_block_pixels(...) pixels (3 floats
 * each): the largest share of any rank; rt_pack_owned writes this rank's pixels to the front of `packed`;
 * rt_scatter_gathered reads nranks such blocks, rank r's at gathered + 3 * r * block_pixels. */
int rt_frame_block_pixels(uint32_t width, uint32_t height, uint32_t tile_size, uint32_t nranks, uint64_t* out);
int rt_pack_owned(rt_ctx* ctx, uint32_t width, uint32_t height, uint32_t tile_size, uint32_t rank, uint32_t nranks,
                  const float* accum_rgb, float* packed);
int rt_scatter_gathered(rt_ctx* ctx, uint32_t width, uint32_t height, uint32_t tile_size, uint32_t nranks,
                        const float* gathered, float* frame_rgb);

#ifdef __cplusplus
}
#endif
#endif /* RTAMD_H */
