// C ABI of the MI355X render path: context, scene upload, BVH commit, ray-batch and render drivers.
// Compiled with hipcc; there is no CPU fallback: every work call needs a GPU and fails with
// RT_ERR_NO_DEVICE / RT_ERR_HIP otherwise.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <queue>
#include <string>
#include <vector>

#include "../../include/rtamd.h"
#include "rt_bvh.h"
#include "rt_device.h"
#include "rt_kernels.h"
#include "rt_lbvh.h"

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

struct HostSpheres {
    std::vector<float> c0r, c1, t01;
    std::vector<uint8_t> moving;
    std::vector<uint32_t> mat, id, group;
    uint32_t n = 0;
};
struct HostRects {
    std::vector<uint8_t> axis;
    std::vector<float> k, bounds;
    std::vector<uint32_t> mat, id, group;
    uint32_t n = 0;
};
struct HostImage {
    uint32_t w = 0, h = 0;
    std::vector<uint8_t> rgb;
};

enum KernelSlot { KS_RAYGEN = 0, KS_TRACE = 1, KS_SHADE = 2, KS_ACCUM = 3, KS_BOUNCE = 4, KS_TAIL = 5, KS_PATHS = 6, KS_SETTLED = 7, KS_COUNT = 8 };

}  // namespace

struct rt_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;

    // host-side scene description
    HostSpheres sph;
    HostRects rect;
    std::vector<rt_group> groups;
    std::vector<rt_material> mats;
    std::vector<rt_texture> texs;
    std::map<uint32_t, std::vector<float>> perlin_vec;
    std::map<uint32_t, std::vector<int32_t>> perlin_perm;
    std::map<uint32_t, HostImage> images;

    // device scene
    bool committed = false;
    float shutter_t0 = 0.0f, shutter_t1 = 1.0f;  // of the last commit: the interval the moving spheres' boxes cover
    bool has_moving = false;
    DevBuf d_prims, d_moving, d_nodes, d_groups, d_mats, d_texs, d_perlin, d_images, d_media;
    std::vector<DevBuf> d_image_data;
    rtd::Scene scene{};
    int stack_entries = 16;
    bool has_groups = false;
    rt_scene_info info{};

    // wavefront buffers
    uint64_t max_paths = 512ull << 20;
    uint64_t capacity = 0;
    uint32_t cap_shard = 0;
    int pipeline = -1;  // -1 choose per scene, 0 split (raygen + trace + shade), 1 fused (one kernel per bounce), 2 hybrid
    int deep_pipeline = 2;  // what trees above fused_node_limit take
    uint32_t n_perlin_tables = 0;  // of the committed scene
    bool whole_paths_auto = true;  // pipeline -1 may pick 3 (whole paths) for flat scenes; env RTAMD_WHOLE_PATHS=0 turns that off
    uint64_t fused_node_limit = 64;  // scenes with more BVH nodes than this take deep_pipeline
    uint64_t split_min_node_bytes = 256ull << 20;  // trees whose nodes outgrow the 256 MB Infinity Cache take the split pipeline
    int builder = -1;  // world-level BVH: -1 by size, 0 host (binned SAH), 1 device (linear BVH)
    uint64_t gpu_build_min = 1u << 18;  // builder -1: worlds with at least this many primitives are built on the device
    bool fuse_boxes = true;  // six rects that form a Box2 become one box record at commit (env RTAMD_FUSE_BOXES=0: keep the rects)
    int bvh_width = -1;  // children per inner node the kernels walk: 2, 4 (four-child form made at commit), -1 by tree size
    uint64_t wide_min_nodes = 1u << 16;  // bvh_width -1: trees of at least this many nodes (4 MiB, one XCD's L2) are walked four-wide
    DevBuf d_wide;
    DevBuf d_crown;  // rtd::Scene::crown
    bool debug_live = false;
    bool settle_ahead = true;  // k_shade may settle the first medium's answer for the ray it makes (RTAMD_SETTLE_AHEAD=0: never)
    uint64_t tail_paths = 256u << 10;  // a batch with at most this many live paths is finished by one k_finish launch
    rtk::SpillArea trace_spill;  // deep traversal-stack entries of k_trace (launch_trace grows it)
    // A wave's 64 camera rays are 64 >> this neighbouring pixels x 1 << this samples of each (env RTAMD_SLOT_SAMPLES_LOG2).
    // 4 pixels x 16 samples: the rays of a wave, and the paths that sit together in the queues afterwards, start from the
    // same few surface points and meet the same materials -- against 64 pixels x 1 sample: random_world -4 %, two_spheres
    // -6 %, 10 M spheres -5 %, final_scene -0.6 %, cornell_smoke +1 %; same bits (a path's draws are keyed by pixel and sample).
    uint32_t slot_samples_log2 = 4;
    DevBuf d_state[2][5], d_hits, d_radiance, d_counts, d_rays, d_overflow, d_pixels, d_accum;  // ([4]: PathState::settled, split pipeline only)
    // Knobs, all read from the environment once in rt_ctx_create: launch shapes ...
    rtk::Tuning tune;
    // ... and what rt_scene_commit / rt_render consult
    uint32_t flat_max = 48;      // RTAMD_FLAT_MAX: scenes of at most this many primitives are traced without a tree
    uint32_t leaf_max = 1;       // RTAMD_LEAF_MAX: primitives per leaf of the host-built trees
    bool near_list = true;       // RTAMD_NEAR_LIST=0: never make the near list of the first medium (no settling ahead)
    bool lds_tables = true;      // RTAMD_LDS_TABLES=0: shading kernels read every table from global memory
    bool print_hist = false;     // RTAMD_HIST (builds with -DRT_HIST): what late unsettled rays hit
    bool print_sections = false; // RTAMD_SECTION_CYCLES: wave cycles per section of the instrumented fused kernel
    bool print_trace_lanes = false;  // RTAMD_TRACE_LANES: lane use of the instrumented persistent trace kernel
    uint64_t pixels_key[5] = {0, 0, 0, 0, 0};
    uint64_t n_owned = 0;
    // frame exchange between ranks (rt_comm_*, rt_gather_frame)
    void* comm = nullptr;       // ncclComm_t of the RCCL library this process has loaded
    bool comm_owned = false;    // made by rt_comm_create (destroyed with the context) or adopted from the caller
    uint32_t comm_rank = 0, comm_nranks = 1;
    DevBuf d_tile_off, d_packed, d_gathered, d_xin, d_xout;
    uint64_t deal_key[4] = {0, 0, 0, 0};
    uint64_t deal_block_pixels = 0;  // pixels of the largest rank's share (every rank's block is padded to it)
    unsigned char* noalloc_scratch = nullptr;  // this device's copy of the static scratch array (rt_intersect_batch_noalloc)
    uint32_t* h_counts = nullptr;  // pinned: NSHARD queue counts, the overflow flag (+15 spare), NSHARD counts of settled paths
    unsigned long long* h_rays = nullptr;

    // profiling
    bool profile = false;
    bool count_visits = false;
    std::vector<hipEvent_t> ev_pool;
    std::vector<std::pair<int, std::pair<int, int>>> ev_used;  // slot, (start idx, stop idx)
    size_t ev_next = 0;
};

namespace {

int fail(rt_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}
int hip_fail(rt_ctx* c, hipError_t e, const char* what) {
    (void)hipGetLastError();
    return fail(c, RT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(ctx, call)                                   \
    do {                                                     \
        hipError_t e_ = (call);                              \
        if (e_ != hipSuccess) return hip_fail(ctx, e_, #call); \
    } while (0)

int ensure(rt_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.bytes >= bytes) return RT_OK;
    b.release();
    HIP_TRY(c, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return RT_OK;
}
int upload(rt_ctx* c, DevBuf& b, const void* src, size_t bytes) {
    int rc = ensure(c, b, bytes);
    if (rc) return rc;
    if (bytes) HIP_TRY(c, hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return RT_OK;
}
bool is_device_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// ---- flattening of the host description into device records -------------------------
struct PrimSrc {
    bool sphere;
    uint32_t idx;
    bool box = false;  // rects idx .. idx+5 of the caller's arrays are the six sides of one Box2 (see fuse_boxes)
};

void sphere_box(const rt_ctx* c, uint32_t i, float t0, float t1, float lo[3], float hi[3]) {
    const float* a = &c->sph.c0r[4 * (size_t)i];
    float r = std::fabs(a[3]);
    for (int k = 0; k < 3; k++) {
        lo[k] = a[k] - r;
        hi[k] = a[k] + r;
    }
    if (!c->sph.moving.empty() && c->sph.moving[i]) {
        // MovingSphere::bounding_box(t0,t1) (shapes.rs:236-249): union of the boxes at both shutter ends
        const float* c1 = &c->sph.c1[3 * (size_t)i];
        float s0 = c->sph.t01[2 * (size_t)i], s1 = c->sph.t01[2 * (size_t)i + 1];
        for (int e = 0; e < 2; e++) {
            float tt = e ? t1 : t0;
            float f = (tt - s0) / (s1 - s0);
            for (int k = 0; k < 3; k++) {
                float ck = a[k] + f * (c1[k] - a[k]);
                lo[k] = e ? std::min(lo[k], ck - r) : ck - r;
                hi[k] = e ? std::max(hi[k], ck + r) : ck + r;
            }
        }
    }
}
void rect_box(const rt_ctx* c, uint32_t i, float lo[3], float hi[3]) {
    // bounding_box of XY/XZ/YZRect (shapes.rs:308-313,372-377,436-441): +-1e-4 on the thin axis
    const float* b = &c->rect.bounds[4 * (size_t)i];
    float k = c->rect.k[i];
    int ax = c->rect.axis[i] == 0 ? 2 : (c->rect.axis[i] == 1 ? 1 : 0);
    int a0 = ax == 0 ? 1 : 0, a1 = ax == 2 ? 1 : 2;
    lo[ax] = k - 0.0001f;
    hi[ax] = k + 0.0001f;
    lo[a0] = std::min(b[0], b[2]);
    hi[a0] = std::max(b[0], b[2]);
    lo[a1] = std::min(b[1], b[3]);
    hi[a1] = std::max(b[1], b[3]);
}
void box_box(const rt_ctx* c, uint32_t i, float lo[3], float hi[3]) {
    // union of the six side rects' boxes (each padded by 1e-4 on its thin axis, as above); the sides are stored
    // XY(min.z), XY(max.z), XZ(min.y), XZ(max.y), YZ(min.x), YZ(max.x) with bounds (min0, min1, max0, max1)
    const float* xy = &c->rect.bounds[4 * (size_t)i];
    const float mn[3] = {xy[0], xy[1], c->rect.k[i]}, mx[3] = {xy[2], xy[3], c->rect.k[i + 1]};
    for (int k = 0; k < 3; k++) {
        lo[k] = std::min(mn[k], mx[k]) - 0.0001f;
        hi[k] = std::max(mn[k], mx[k]) + 0.0001f;
    }
}
// Is the run of six rects starting at s[0] exactly what Box2::full_box emits (shapes.rs:459-466)?  Consecutive
// ids, one material, sides in the order XY(min.z), XY(max.z), XZ(min.y), XZ(max.y), YZ(min.x), YZ(max.x), and
// every bound equal bit for bit to the corner it comes from.  Only then do six rect tests in this order with
// these numbers equal the box record's.
bool is_box2(const rt_ctx* c, const PrimSrc* s) {
    const HostRects& r = c->rect;
    for (int k = 0; k < 6; k++) {
        if (s[k].sphere || s[k].box) return false;
        if (r.id[s[k].idx] != r.id[s[0].idx] + (uint32_t)k || r.mat[s[k].idx] != r.mat[s[0].idx]) return false;
        if (r.axis[s[k].idx] != (uint8_t)(k / 2)) return false;
        if (s[k].idx != s[0].idx + (uint32_t)k) return false;  // the record reads k[i], k[i+1] and bounds[i] of the caller's arrays
    }
    const uint32_t i = s[0].idx;
    const float* b = r.bounds.data();
    auto same = [](float x, float y) { return memcmp(&x, &y, sizeof(float)) == 0; };
    const float x0 = b[4 * i], y0 = b[4 * i + 1], x1 = b[4 * i + 2], y1 = b[4 * i + 3], z0 = r.k[i], z1 = r.k[i + 1];
    const float want[6][5] = {{x0, y0, x1, y1, z0}, {x0, y0, x1, y1, z1}, {x0, z0, x1, z1, y0},
                              {x0, z0, x1, z1, y1}, {y0, z0, y1, z1, x0}, {y0, z0, y1, z1, x1}};
    for (int k = 0; k < 6; k++) {
        for (int j = 0; j < 4; j++)
            if (!same(b[4 * (i + k) + j], want[k][j])) return false;
        if (!same(r.k[i + k], want[k][4])) return false;
    }
    return true;
}
// Replaces every run of six rects that is a Box2 by one box source (in place, order kept).
void fuse_boxes(const rt_ctx* c, std::vector<PrimSrc>& v) {
    std::vector<PrimSrc> out;
    out.reserve(v.size());
    for (size_t i = 0; i < v.size();) {
        if (i + 6 <= v.size() && is_box2(c, &v[i])) {
            PrimSrc s = v[i];
            s.box = true;
            out.push_back(s);
            i += 6;
        } else {
            out.push_back(v[i++]);
        }
    }
    v.swap(out);
}
// Boxes only cull; widen them a little so that rounding in the primitive tests (which follow the
// reference's arithmetic exactly) can never place an accepted hit outside its leaf box.
void widen(float lo[3], float hi[3]) {
    for (int k = 0; k < 3; k++) {
        float m = std::max(std::max(std::fabs(lo[k]), std::fabs(hi[k])), hi[k] - lo[k]);
        float pad = 2e-6f * m + 1e-30f;
        lo[k] -= pad;
        hi[k] += pad;
    }
}
void xform_box(const float* m, const float lo[3], const float hi[3], float olo[3], float ohi[3]) {
    for (int k = 0; k < 3; k++) {
        olo[k] = INFINITY;
        ohi[k] = -INFINITY;
    }
    for (int corner = 0; corner < 8; corner++) {
        float p[3] = {(corner & 1) ? hi[0] : lo[0], (corner & 2) ? hi[1] : lo[1], (corner & 4) ? hi[2] : lo[2]};
        for (int r = 0; r < 3; r++) {
            float v = m[4 * r] * p[0] + m[4 * r + 1] * p[1] + m[4 * r + 2] * p[2] + m[4 * r + 3];
            olo[r] = std::min(olo[r], v);
            ohi[r] = std::max(ohi[r], v);
        }
    }
}

rtd::Prim make_prim(const rt_ctx* c, const PrimSrc& s, std::vector<rtd::Moving>& moving) {
    rtd::Prim p{};
    if (s.sphere) {
        const float* a = &c->sph.c0r[4 * (size_t)s.idx];
        for (int k = 0; k < 4; k++) p.a[k] = a[k];
        bool mv = !c->sph.moving.empty() && c->sph.moving[s.idx];
        p.info = (mv ? rtd::PK_MOVING_SPHERE : rtd::PK_SPHERE) | (c->sph.mat[s.idx] << 8);
        p.id = c->sph.id[s.idx];
        if (mv) {
            rtd::Moving m{};
            for (int k = 0; k < 3; k++) m.c1[k] = c->sph.c1[3 * (size_t)s.idx + k];
            m.t0 = c->sph.t01[2 * (size_t)s.idx];
            m.t1 = c->sph.t01[2 * (size_t)s.idx + 1];
            p.aux = (uint32_t)moving.size();
            moving.push_back(m);
        }
    } else if (s.box) {
        const float* b = &c->rect.bounds[4 * (size_t)s.idx];  // side 0: XY rect (min.x, min.y, max.x, max.y) at z = min.z
        p.a[0] = b[0];
        p.a[1] = b[1];
        p.a[2] = c->rect.k[s.idx];
        p.a[3] = b[2];
        p.b0 = b[3];
        const float max_z = c->rect.k[s.idx + 1];
        memcpy(&p.aux, &max_z, sizeof(float));
        p.info = rtd::PK_BOX | (c->rect.mat[s.idx] << 8);
        p.id = c->rect.id[s.idx];
    } else {
        const float* b = &c->rect.bounds[4 * (size_t)s.idx];
        p.a[0] = c->rect.k[s.idx];
        p.a[1] = b[0];  // min0
        p.a[2] = b[1];  // min1
        p.a[3] = b[2];  // max0
        p.b0 = b[3];    // max1
        uint32_t kind = c->rect.axis[s.idx] == 0 ? rtd::PK_RECT_XY : (c->rect.axis[s.idx] == 1 ? rtd::PK_RECT_XZ : rtd::PK_RECT_YZ);
        p.info = kind | (c->rect.mat[s.idx] << 8);
        p.id = c->rect.id[s.idx];
    }
    return p;
}

uint32_t env_u32(const char* name, uint32_t dflt) {
    const char* v = getenv(name);
    if (!v || !*v) return dflt;
    return (uint32_t)strtoul(v, nullptr, 10);
}

// Wavefront buffers for batches of up to `cap` paths.  The path queues are split into rtk::NSHARD shards
// of cap_shard entries; the hit buffer is only needed by the split pipeline.
int alloc_wavefront(rt_ctx* c, uint64_t cap, bool need_hits) {
    int rc;
    if (c->capacity < cap) {
        const uint64_t tiles = (cap + rtk::BLOCK - 1) / rtk::BLOCK;
        const uint64_t cap_shard = ((tiles + rtk::NSHARD - 1) / rtk::NSHARD) * rtk::BLOCK;
        const uint64_t total = cap_shard * rtk::NSHARD;
        for (int s = 0; s < 2; s++) {
            for (int a = 0; a < 3; a++)
                if ((rc = ensure(c, c->d_state[s][a], total * sizeof(float4)))) return rc;
            if ((rc = ensure(c, c->d_state[s][3], total * sizeof(uint2)))) return rc;
        }
        for (int s = 0; s < 2; s++) c->d_state[s][4].release();  // (sized with the hit buffer below)
        if ((rc = ensure(c, c->d_radiance, cap * sizeof(float4)))) return rc;
        if ((rc = ensure(c, c->d_counts, 6 * rtk::NSHARD * sizeof(uint32_t)))) return rc;  // (rows 3..5: the settled paths, rtk::Queues)
        if ((rc = ensure(c, c->d_rays, 256))) return rc;
        if ((rc = ensure(c, c->d_overflow, 16))) return rc;
        c->capacity = cap;
        c->cap_shard = (uint32_t)cap_shard;
    }
    if (need_hits) {
        const uint64_t total = (uint64_t)c->cap_shard * rtk::NSHARD;
        if ((rc = ensure(c, c->d_hits, total * sizeof(float4)))) return rc;
        for (int s = 0; s < 2; s++)
            if ((rc = ensure(c, c->d_state[s][4], total * sizeof(float)))) return rc;
    }
    return RT_OK;
}
rtk::PathState path_state(rt_ctx* c, int s) {
    rtk::PathState p;
    p.o_time = (float4*)c->d_state[s][0].p;
    p.d_slot = (float4*)c->d_state[s][1].p;
    p.thr_meta = (float4*)c->d_state[s][2].p;
    p.key = (uint2*)c->d_state[s][3].p;
    p.settled = (float*)c->d_state[s][4].p;  // (null in the fused pipelines, which never settle)
    return p;
}

// Wavefront buffers for a render of n_paths camera samples: one batch if it fits, else the largest that does.
// Large batches amortise the per-bounce launches and tails (one batch of 512 Mi paths renders a 1080p frame at
// 256 spp ~20% faster than eight of 64 Mi), but they must fit: at most 60% of what the device has free now plus
// what the wavefront buffers already hold (152 bytes per path).
int reserve_paths(rt_ctx* c, uint64_t n_paths, bool need_hits) {
    uint64_t cap = std::min<uint64_t>(c->max_paths, n_paths);
    if (cap > c->capacity) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const uint64_t held = c->capacity * 152ull;
            const uint64_t fit = (uint64_t)(0.6 * (double)(free_b + held)) / 152ull;
            cap = std::min<uint64_t>(cap, std::max<uint64_t>(fit, 1ull << 20));
        } else {
            (void)hipGetLastError();
        }
    }
    return alloc_wavefront(c, std::max<uint64_t>(cap, 1024), need_hits);
}

// ---- event profiling -------------------------------------------------------------
int ev_get(rt_ctx* c, int& idx) {
    if (c->ev_next == c->ev_pool.size()) {
        hipEvent_t e;
        HIP_TRY(c, hipEventCreate(&e));
        c->ev_pool.push_back(e);
    }
    idx = (int)c->ev_next++;
    return RT_OK;
}
struct ScopedTimer {
    rt_ctx* c;
    int slot, i0 = -1, i1 = -1;
    bool on;
    ScopedTimer(rt_ctx* ctx, int s) : c(ctx), slot(s), on(ctx->profile) {
        if (on && ev_get(c, i0) == RT_OK) (void)hipEventRecord(c->ev_pool[i0], c->stream);
    }
    ~ScopedTimer() {
        if (on && i0 >= 0 && ev_get(c, i1) == RT_OK) {
            (void)hipEventRecord(c->ev_pool[i1], c->stream);
            c->ev_used.push_back({slot, {i0, i1}});
        }
    }
};

// The crown of a committed tree (rtd::CROWN_BASE): up to max_nodes inner nodes, picked greedily by the area of their box
// from the root downwards -- for rays spread evenly over the scene the chance of a visit goes with that area -- through
// instance references into the instances' own trees as well.  The records are read back from the device one by one (a
// few hundred small copies; the trees may have been built there) and written as Scene::crown describes.
// words stays empty when the root is not an inner node.
int build_crown(rt_ctx* c, bool wide, uint32_t root_ref, const std::vector<rtd::Group>& groups, uint32_t max_nodes, bool partial,
                std::vector<uint32_t>& words, uint32_t& crown_root) {
    words.clear();
    crown_root = root_ref;
    if (max_nodes == 0 || root_ref == rtd::REF_EMPTY || (root_ref >> 30) != rtd::REF_INNER) return RT_OK;
    max_nodes = std::min(max_nodes, rtd::CROWN_MAX);
    const size_t rec_words = wide ? sizeof(rtd::WideNode) / 4 : sizeof(rtd::Node) / 4;
    const char* dev = (const char*)(wide ? c->d_wide.p : c->d_nodes.p);
    struct Cand {
        float area;
        uint32_t node;
        bool operator<(const Cand& o) const { return area < o.area; }
    };
    std::priority_queue<Cand> todo;
    todo.push({INFINITY, root_ref & 0x3FFFFFFFu});
    std::map<uint32_t, uint32_t> place;  // node index -> place in the crown
    std::vector<uint32_t> recs;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    while (!todo.empty() && place.size() < max_nodes) {
        const Cand cand = todo.top();
        todo.pop();
        if (place.count(cand.node)) continue;
        place[cand.node] = (uint32_t)place.size();
        recs.resize(recs.size() + rec_words);
        uint32_t* rec = recs.data() + recs.size() - rec_words;
        HIP_TRY(c, hipMemcpy(rec, dev + (size_t)cand.node * rec_words * 4, rec_words * 4, hipMemcpyDeviceToHost));
        const int n_children = wide ? 4 : 2;
        for (int k = 0; k < n_children; k++) {
            const float* box = reinterpret_cast<const float*>(rec) + 6 * k;
            uint32_t ref = rec[(wide ? 24 : 12) + k];
            if (ref == rtd::REF_EMPTY) continue;
            if ((ref >> 30) == rtd::REF_INSTANCE) {  // on into the instance's own tree
                const uint32_t g = ref & 0x3FFFFFFFu;
                if (g >= rtd::CROWN_INST_SLOTS || g >= groups.size()) continue;
                ref = groups[g].root_ref;
                if (ref == rtd::REF_EMPTY) continue;
            }
            if ((ref >> 30) != rtd::REF_INNER) continue;
            const float dx = box[3] - box[0], dy = box[4] - box[1], dz = box[5] - box[2];
            const float area = dx * dy + dy * dz + dz * dx;
            todo.push({area == area ? area : 0.0f, ref & 0x3FFFFFFFu});
        }
    }
    // Measured: a crown that holds only the top of a larger tree buys nothing (neighbouring lanes share the nodes near the
    // root, and loads of the same address cost the texture path a quarter) while the large blocks cost a few per cent; with
    // the whole tree in LDS the persistent walk of a few-hundred-sphere scene runs a fifth faster.
    while (!todo.empty() && place.count(todo.top().node)) todo.pop();
    if (!todo.empty() && !partial) return RT_OK;
    words.assign(rtd::CROWN_INST_SLOTS, rtd::REF_EMPTY);
    for (uint32_t g = 0; g < rtd::CROWN_INST_SLOTS && g < groups.size(); g++) {
        const uint32_t r = groups[g].root_ref;
        if (groups[g].kind == 1u && r != rtd::REF_EMPTY && (r >> 30) == rtd::REF_INNER && place.count(r & 0x3FFFFFFFu))
            words[g] = rtd::CROWN_BASE + place[r & 0x3FFFFFFFu];
    }
    const size_t n = place.size();
    for (size_t j = 0; j < n; j++) {
        uint32_t* rec = recs.data() + j * rec_words;
        for (int k = 0; k < (wide ? 4 : 2); k++) {
            uint32_t& ref = rec[(wide ? 24 : 12) + k];
            if (ref != rtd::REF_EMPTY && (ref >> 30) == rtd::REF_INNER) {
                auto it = place.find(ref & 0x3FFFFFFFu);
                if (it != place.end()) ref = rtd::CROWN_BASE + it->second;
            }
        }
    }
    words.insert(words.end(), recs.begin(), recs.end());
    crown_root = rtd::CROWN_BASE + place[root_ref & 0x3FFFFFFFu];
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_ctx_create(int device, rt_ctx** out) {
    if (!out) return RT_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return RT_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) return RT_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return RT_ERR_HIP;
    rt_ctx* c = new rt_ctx();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return RT_ERR_HIP;
    }
    c->stream = c->own_stream;
    if (hipHostMalloc((void**)&c->h_counts, (2 * rtk::NSHARD + 32) * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&c->h_rays, 256, hipHostMallocDefault) != hipSuccess) {
        delete c;
        return RT_ERR_HIP;
    }
    // device address of this GPU's copy of the static scratch array, looked up once per context with its device current
    if (rtk::noalloc_scratch_address((void**)&c->noalloc_scratch) != hipSuccess || !c->noalloc_scratch) {
        (void)hipGetLastError();
        (void)hipHostFree(c->h_counts);
        (void)hipHostFree(c->h_rays);
        (void)hipStreamDestroy(c->own_stream);
        delete c;
        return RT_ERR_HIP;
    }
    c->max_paths = (uint64_t)env_u32("RTAMD_MAX_PATHS_MI", 512) << 20;
    if (getenv("RTAMD_PIPELINE")) c->pipeline = (int)std::min(env_u32("RTAMD_PIPELINE", 1), 3u);  // 0 split, 1 fused, 2 hybrid, 3 whole paths
    if (getenv("RTAMD_DEEP_PIPELINE")) c->deep_pipeline = (int)std::min(env_u32("RTAMD_DEEP_PIPELINE", 2), 2u);
    if (getenv("RTAMD_FUSED_NODE_LIMIT")) c->fused_node_limit = env_u32("RTAMD_FUSED_NODE_LIMIT", 0);
    if (getenv("RTAMD_SPLIT_MIN_NODE_MB")) c->split_min_node_bytes = (uint64_t)env_u32("RTAMD_SPLIT_MIN_NODE_MB", 256) << 20;
    if (getenv("RTAMD_WHOLE_PATHS")) c->whole_paths_auto = env_u32("RTAMD_WHOLE_PATHS", 1) != 0;
    if (getenv("RTAMD_TAIL_PATHS")) c->tail_paths = env_u32("RTAMD_TAIL_PATHS", 0);
    c->settle_ahead = env_u32("RTAMD_SETTLE_AHEAD", 1) != 0;
    c->debug_live = env_u32("RTAMD_DEBUG_LIVE", 0) != 0;
    if (getenv("RTAMD_SLOT_SAMPLES_LOG2")) c->slot_samples_log2 = std::min(env_u32("RTAMD_SLOT_SAMPLES_LOG2", 0), 6u);
    if (getenv("RTAMD_BUILDER")) c->builder = (int)std::min(env_u32("RTAMD_BUILDER", 0), 1u);
    if (getenv("RTAMD_GPU_BUILD_MIN")) c->gpu_build_min = env_u32("RTAMD_GPU_BUILD_MIN", 0);
    if (getenv("RTAMD_BVH_WIDTH")) {
        const uint32_t w = env_u32("RTAMD_BVH_WIDTH", 0);
        c->bvh_width = w == 4 ? 4 : (w == 2 ? 2 : -1);
    }
    if (getenv("RTAMD_FUSE_BOXES")) c->fuse_boxes = env_u32("RTAMD_FUSE_BOXES", 1) != 0;
    if (getenv("RTAMD_WIDE_MIN_NODES")) c->wide_min_nodes = env_u32("RTAMD_WIDE_MIN_NODES", 0);
    c->flat_max = env_u32("RTAMD_FLAT_MAX", c->flat_max);
    c->leaf_max = env_u32("RTAMD_LEAF_MAX", c->leaf_max);
    c->near_list = env_u32("RTAMD_NEAR_LIST", 1) != 0;
    c->lds_tables = env_u32("RTAMD_LDS_TABLES", 1) != 0;
    c->print_hist = getenv("RTAMD_HIST") != nullptr;
    c->print_sections = getenv("RTAMD_SECTION_CYCLES") != nullptr;
    c->print_trace_lanes = getenv("RTAMD_TRACE_LANES") != nullptr;
    rtk::Tuning& tn = c->tune;
    auto knob = [](const char* name, int dflt, int lo, int hi) { return (int)std::min<uint32_t>((uint32_t)hi, std::max<uint32_t>((uint32_t)lo, env_u32(name, (uint32_t)dflt))); };
    tn.oversub = knob("RTAMD_OVERSUB", tn.oversub, 1, 64);
    tn.trace_oversub = knob("RTAMD_TRACE_OVERSUB", tn.trace_oversub, 1, 64);
    tn.stack_lds = knob("RTAMD_STACK_LDS", tn.stack_lds, 1, 64);
    tn.trace_min_run = 8 * knob("RTAMD_TRACE_MIN_RUN", tn.trace_min_run / 8, 1, 64);
    tn.refill = knob("RTAMD_REFILL", tn.refill, 1, 64);
    tn.inner_steps = knob("RTAMD_INNER_STEPS", tn.inner_steps, 1, 64);
    tn.regen = knob("RTAMD_REGEN", tn.regen, 1, 64);
    tn.paths_run = 24 * knob("RTAMD_PATHS_RUN", tn.paths_run / 24, 1, 64);
    tn.settle_entry = (uint32_t)knob("RTAMD_SETTLE_ENTRY", (int)tn.settle_entry, 1, 64);
    tn.settle_near_hits = env_u32("RTAMD_SETTLE_NEAR_HITS", tn.settle_near_hits) != 0 ? 1u : 0u;
    tn.settled_refill = knob("RTAMD_SETTLED_REFILL", tn.settled_refill, 1, 64);
    tn.settled_min_run = knob("RTAMD_SETTLED_MIN_RUN", tn.settled_min_run, 64, 1 << 20);
    tn.settled_slow_min = knob("RTAMD_SETTLED_SLOW_MIN", tn.settled_slow_min, 1, 64);
    tn.settled_fast = knob("RTAMD_SETTLED_FAST", tn.settled_fast, 0, 1);
    tn.settled_late_rounds = knob("RTAMD_SETTLED_LATE_ROUNDS", tn.settled_late_rounds, 1, 255);
    tn.settled_drain = knob("RTAMD_SETTLED_DRAIN", tn.settled_drain, 1, 64);
    tn.settled_oversub = knob("RTAMD_SETTLED_OVERSUB", tn.settled_oversub, 1, 64);
    tn.crown_nodes = knob("RTAMD_CROWN_NODES", tn.crown_nodes, 0, 65535);
    tn.crown_partial = knob("RTAMD_CROWN_PARTIAL", tn.crown_partial, 0, 1);
    tn.crown_min_blocks = knob("RTAMD_CROWN_MIN_BLOCKS", tn.crown_min_blocks, 0, 1 << 20);
    *out = c;
    return RT_OK;
}

void rt_ctx_destroy(rt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)rt_comm_destroy(c);
    DevBuf* xbufs[] = {&c->d_tile_off, &c->d_packed, &c->d_gathered, &c->d_xin, &c->d_xout};
    for (DevBuf* b : xbufs) b->release();
    DevBuf* bufs[] = {&c->d_prims, &c->d_moving, &c->d_nodes, &c->d_wide, &c->d_crown, &c->d_groups, &c->d_mats, &c->d_texs, &c->d_perlin,
                      &c->d_images, &c->d_media, &c->d_hits, &c->d_radiance, &c->d_counts, &c->d_rays, &c->d_overflow, &c->d_pixels,
                      &c->d_accum};
    for (DevBuf* b : bufs) b->release();
    for (auto& b : c->d_image_data) b.release();
    for (int s = 0; s < 2; s++)
        for (int a = 0; a < 5; a++) c->d_state[s][a].release();
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->h_counts) (void)hipHostFree(c->h_counts);
    if (c->h_rays) (void)hipHostFree(c->h_rays);
    if (c->trace_spill.p) (void)hipFree(c->trace_spill.p);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

const char* rt_last_error(const rt_ctx* c) { return c ? c->err.c_str() : "null context"; }

int rt_ctx_set_stream(rt_ctx* c, void* hip_stream) {
    if (!c) return RT_ERR_INVALID;
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return RT_OK;
}
int rt_ctx_set_max_paths(rt_ctx* c, uint64_t max_paths) {
    if (!c || max_paths < 1024 || max_paths > (1ull << 31)) return fail(c, RT_ERR_INVALID, "max_paths out of range");
    c->max_paths = max_paths;
    return RT_OK;
}
int rt_ctx_set_pipeline(rt_ctx* c, int pipeline) {
    if (!c || pipeline < -1 || pipeline > 3) return fail(c, RT_ERR_INVALID, "pipeline must be -1, 0, 1, 2 or 3");
    c->pipeline = pipeline;
    return RT_OK;
}
int rt_ctx_set_builder(rt_ctx* c, int builder) {
    if (!c || builder < -1 || builder > 1) return fail(c, RT_ERR_INVALID, "builder must be -1, 0 or 1");
    c->builder = builder;
    return RT_OK;
}
int rt_ctx_set_bvh_width(rt_ctx* c, int width) {
    if (!c || (width != 2 && width != 4 && width != -1)) return fail(c, RT_ERR_INVALID, "bvh width must be -1, 2 or 4");
    c->bvh_width = width;
    return RT_OK;
}
int rt_ctx_set_fuse_boxes(rt_ctx* c, int on) {
    if (!c) return RT_ERR_INVALID;
    c->fuse_boxes = on != 0;
    return RT_OK;
}
int rt_ctx_reserve(rt_ctx* c, uint64_t n_paths) {
    if (!c) return RT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    return reserve_paths(c, n_paths, true);
}
int rt_ctx_set_tail_paths(rt_ctx* c, uint64_t tail_paths) {
    if (!c) return RT_ERR_INVALID;
    c->tail_paths = tail_paths;
    return RT_OK;
}
int rt_ctx_synchronize(rt_ctx* c) {
    if (!c) return RT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_scene_clear(rt_ctx* c) {
    if (!c) return RT_ERR_INVALID;
    c->sph = HostSpheres();
    c->rect = HostRects();
    c->groups.clear();
    c->mats.clear();
    c->texs.clear();
    c->perlin_vec.clear();
    c->perlin_perm.clear();
    c->images.clear();
    c->committed = false;
    return RT_OK;
}

int rt_scene_set_spheres(rt_ctx* c, uint32_t n, const float* c0_r, const float* c1, const float* t01,
                         const uint8_t* moving, const uint32_t* mat, const uint32_t* id, const uint32_t* group) {
    if (!c) return RT_ERR_INVALID;
    if (n && (!c0_r || !mat || !id)) return fail(c, RT_ERR_INVALID, "rt_scene_set_spheres: null array");
    if (moving && (!c1 || !t01)) return fail(c, RT_ERR_INVALID, "rt_scene_set_spheres: moving flags need c1 and t01");
    HostSpheres& s = c->sph;
    s = HostSpheres();
    s.n = n;
    s.c0r.assign(c0_r, c0_r + 4 * (size_t)n);
    if (moving) {
        s.moving.assign(moving, moving + n);
        s.c1.assign(c1, c1 + 3 * (size_t)n);
        s.t01.assign(t01, t01 + 2 * (size_t)n);
    }
    s.mat.assign(mat, mat + n);
    s.id.assign(id, id + n);
    if (group)
        s.group.assign(group, group + n);
    else
        s.group.assign(n, 0u);
    c->committed = false;
    return RT_OK;
}

int rt_scene_set_rects(rt_ctx* c, uint32_t n, const uint8_t* axis, const float* k, const float* bounds,
                       const uint32_t* mat, const uint32_t* id, const uint32_t* group) {
    if (!c) return RT_ERR_INVALID;
    if (n && (!axis || !k || !bounds || !mat || !id)) return fail(c, RT_ERR_INVALID, "rt_scene_set_rects: null array");
    for (uint32_t i = 0; i < n; i++)
        if (axis[i] > 2) return fail(c, RT_ERR_INVALID, "rt_scene_set_rects: axis must be 0, 1 or 2");
    HostRects& r = c->rect;
    r = HostRects();
    r.n = n;
    r.axis.assign(axis, axis + n);
    r.k.assign(k, k + n);
    r.bounds.assign(bounds, bounds + 4 * (size_t)n);
    r.mat.assign(mat, mat + n);
    r.id.assign(id, id + n);
    if (group)
        r.group.assign(group, group + n);
    else
        r.group.assign(n, 0u);
    c->committed = false;
    return RT_OK;
}

int rt_scene_set_groups(rt_ctx* c, uint32_t n, const rt_group* groups) {
    if (!c || (n && !groups)) return fail(c, RT_ERR_INVALID, "rt_scene_set_groups: null array");
    for (uint32_t i = 0; i < n; i++)
        if (groups[i].kind != RT_GROUP_INSTANCE && groups[i].kind != RT_GROUP_MEDIUM)
            return fail(c, RT_ERR_INVALID, "rt_scene_set_groups: unknown group kind");
    c->groups.assign(groups, groups + n);
    c->committed = false;
    return RT_OK;
}
int rt_scene_set_materials(rt_ctx* c, uint32_t n, const rt_material* mats) {
    if (!c || (n && !mats)) return fail(c, RT_ERR_INVALID, "rt_scene_set_materials: null array");
    for (uint32_t i = 0; i < n; i++)
        if (mats[i].kind > RT_MAT_ISOTROPIC) return fail(c, RT_ERR_INVALID, "rt_scene_set_materials: unknown kind");
    if (n >= (1u << 24)) return fail(c, RT_ERR_UNSUPPORTED, "more than 2^24 materials");
    c->mats.assign(mats, mats + n);
    c->committed = false;
    return RT_OK;
}
int rt_scene_set_textures(rt_ctx* c, uint32_t n, const rt_texture* texs) {
    if (!c || (n && !texs)) return fail(c, RT_ERR_INVALID, "rt_scene_set_textures: null array");
    for (uint32_t i = 0; i < n; i++)
        if (texs[i].kind > RT_TEX_IMAGE) return fail(c, RT_ERR_INVALID, "rt_scene_set_textures: unknown kind");
    c->texs.assign(texs, texs + n);
    c->committed = false;
    return RT_OK;
}
int rt_scene_set_perlin(rt_ctx* c, uint32_t table, const float* vec, const int32_t* perm) {
    if (!c || !vec || !perm || table >= 1024) return fail(c, RT_ERR_INVALID, "rt_scene_set_perlin: bad argument");
    for (int i = 0; i < 768; i++)
        if (perm[i] < 0 || perm[i] > 255) return fail(c, RT_ERR_INVALID, "rt_scene_set_perlin: perm entry outside 0..255");
    c->perlin_vec[table].assign(vec, vec + 768);
    c->perlin_perm[table].assign(perm, perm + 768);
    c->committed = false;
    return RT_OK;
}
int rt_scene_set_image(rt_ctx* c, uint32_t table, uint32_t w, uint32_t h, const uint8_t* rgb) {
    if (!c || !rgb || !w || !h || table >= 1024) return fail(c, RT_ERR_INVALID, "rt_scene_set_image: bad argument");
    HostImage& im = c->images[table];
    im.w = w;
    im.h = h;
    im.rgb.assign(rgb, rgb + 3 * (size_t)w * h);
    c->committed = false;
    return RT_OK;
}

int rt_scene_commit(rt_ctx* c, float shutter_t0, float shutter_t1) {
    if (!c) return RT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    auto tstart = std::chrono::steady_clock::now();
    c->committed = false;
    const uint32_t ng = (uint32_t)c->groups.size();
    const uint64_t total_prims = (uint64_t)c->sph.n + c->rect.n;
    if (total_prims >= rtd::MAX_PRIMS) return fail(c, RT_ERR_UNSUPPORTED, "more than 2^26 primitives");
    // validate references
    const uint32_t nm = (uint32_t)c->mats.size(), nt = (uint32_t)c->texs.size();
    for (uint32_t i = 0; i < c->sph.n; i++)
        if (c->sph.mat[i] >= nm || c->sph.group[i] > ng) return fail(c, RT_ERR_INVALID, "sphere: material or group out of range");
    for (uint32_t i = 0; i < c->rect.n; i++)
        if (c->rect.mat[i] >= nm || c->rect.group[i] > ng) return fail(c, RT_ERR_INVALID, "rect: material or group out of range");
    for (const rt_material& m : c->mats)
        if ((m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_DIFFUSE_LIGHT || m.kind == RT_MAT_ISOTROPIC) && m.tex >= nt)
            return fail(c, RT_ERR_INVALID, "material: texture out of range");
    uint32_t n_perlin = 0, n_images = 0;
    for (const rt_texture& t : c->texs) {
        if (t.kind == RT_TEX_CHECKER && (t.odd >= nt || t.even >= nt)) return fail(c, RT_ERR_INVALID, "checker: child texture out of range");
        if (t.kind == RT_TEX_NOISE) {
            if (!c->perlin_vec.count(t.table)) return fail(c, RT_ERR_INVALID, "noise texture: perlin table not set");
            n_perlin = std::max(n_perlin, t.table + 1);
        }
        if (t.kind == RT_TEX_IMAGE) {
            if (!c->images.count(t.table)) return fail(c, RT_ERR_INVALID, "image texture: image not set");
            n_images = std::max(n_images, t.table + 1);
        }
    }
    for (const rt_group& g : c->groups)
        if (g.kind == RT_GROUP_MEDIUM && g.mat >= nm) return fail(c, RT_ERR_INVALID, "medium: material out of range");
    // Groups inside instances (rt_group::parent): the enclosing group comes first in the array and is an instance; at
    // most MAX_NESTING instances are stacked.  A medium may stand inside instances; nothing stands inside a medium.
    std::vector<uint32_t> nest_depth(ng, 0);
    std::vector<std::vector<uint32_t>> children(ng);
    bool nested = false;
    for (uint32_t g = 0; g < ng; g++) {
        const rt_group& G = c->groups[g];
        if (G.parent == 0) continue;
        if (G.parent > g) return fail(c, RT_ERR_INVALID, "group: parent must be a group earlier in the array");
        const rt_group& P = c->groups[G.parent - 1];
        if (P.kind != RT_GROUP_INSTANCE)
            return fail(c, RT_ERR_UNSUPPORTED, "a group nested inside a medium (its boundary or the medium itself) is not supported");
        nest_depth[g] = nest_depth[G.parent - 1] + 1;
        nested = true;
        if (G.kind == RT_GROUP_MEDIUM) continue;  // a medium inside instances: handed their ray, evaluated after the surfaces like any medium
        if (nest_depth[g] >= rtd::MAX_NESTING) return fail(c, RT_ERR_UNSUPPORTED, "instances nested more than 4 deep");
        children[G.parent - 1].push_back(g);
    }

    // bucket primitives by group, in id order inside a group (Scene::intersect order for medium boundaries)
    std::vector<std::vector<PrimSrc>> members(ng + 1);
    for (uint32_t i = 0; i < c->sph.n; i++) members[c->sph.group[i]].push_back({true, i});
    for (uint32_t i = 0; i < c->rect.n; i++) members[c->rect.group[i]].push_back({false, i});
    auto id_of = [&](const PrimSrc& s) { return s.sphere ? c->sph.id[s.idx] : c->rect.id[s.idx]; };
    for (auto& v : members) std::sort(v.begin(), v.end(), [&](const PrimSrc& a, const PrimSrc& b) { return id_of(a) < id_of(b); });
    // Scenes of a few dozen primitives are traced without the tree (see rtd::Scene::flat): incoherent rays take
    // different branches of even a four-node tree, which costs more lane time than testing everything in step.
    bool flat_ok = total_prims <= c->flat_max;
    for (uint32_t g = 1; g <= ng; g++)
        if (members[g].size() > 64) flat_ok = false;
    if (nested) flat_ok = false;  // (the loop that stands in for the tree takes every instance's ray from the world ray)
    // In such a scene a Box2 (six rects, shapes.rs:456-466) becomes ONE record: its six tests share three reciprocals
    // (cornell_box 32.0 -> 26.1 ms per 64 spp).  A scene that takes the tree keeps its rects: there the box test's
    // registers cost the walk a resident wave per SIMD, which outweighs the shallower tree (final_scene 120 against
    // 110 ms per 256 spp), so the tree-walking kernels are compiled without it.  Medium boundaries keep their rects
    // too (the boundary queries want the separate roots).
    if (c->fuse_boxes && flat_ok)
        for (uint32_t g = 0; g <= ng; g++)
            if (g == 0 || c->groups[g - 1].kind != RT_GROUP_MEDIUM) fuse_boxes(c, members[g]);

    const uint32_t leaf_max = std::max(1u, std::min(c->leaf_max, rtd::MAX_LEAF_PRIMS));
    std::vector<rtd::Node> nodes;
    std::vector<rtd::Prim> prims;
    std::vector<rtd::Moving> moving;
    std::vector<rtd::Group> dgroups(ng);
    prims.reserve(total_prims);
    auto prim_box = [&](const PrimSrc& s, float lo[3], float hi[3]) {
        if (s.sphere)
            sphere_box(c, s.idx, shutter_t0, shutter_t1, lo, hi);
        else if (s.box)
            box_box(c, s.idx, lo, hi);
        else
            rect_box(c, s.idx, lo, hi);
        widen(lo, hi);
    };

    std::vector<rtb::Item> top;
    std::vector<uint32_t> media_groups;
    uint32_t blas_depth = 0, blas_depth_wide = 0, n_media = 0, n_inst = 0;
    // stack entries the walk of an instance can need from its root down, instances inside it included (binary / four-child walk)
    std::vector<uint32_t> chain_bin(ng, 0), chain_wide(ng, 0);
    std::vector<rtb::Item> group_item(ng);  // the leaf that stands for the group in the tree around it
    std::vector<uint8_t> group_has_item(ng, 0);
    for (uint32_t g = 0; g < ng; g++) dgroups[g].order = c->groups[g].kind == RT_GROUP_MEDIUM ? n_media++ : n_inst++;
    // groups first: each gets a contiguous primitive range and (for instances) its own BVH.  Last to first, so that
    // the instances inside an instance (which come after it) are there when its tree is built.
    for (uint32_t g = ng; g-- > 0;) {
        const rt_group& G = c->groups[g];
        const auto& mem = members[g + 1];
        rtd::Group& D = dgroups[g];
        D.parent = (int32_t)G.parent - 1;
        D.depth = nest_depth[g];
        memcpy(D.m, G.m, sizeof(D.m));
        memcpy(D.minv, G.minv, sizeof(D.minv));
        D.neg_inv_density = G.neg_inv_density;
        D.mat = G.mat;
        D.kind = G.kind;
        D.has_xform = G.has_xform ? 1u : 0u;
        D.root_ref = rtd::REF_EMPTY;
        D.flat_range = rtd::FLAT_NONE;
        for (int k = 0; k < 3; k++) {
            D.blo[k] = INFINITY;   // empty group: nothing can be hit
            D.bhi[k] = -INFINITY;
        }
        // a box record costs six rect tests: size decisions count it as six
        size_t mem_tests = 0;
        for (const PrimSrc& ps : mem) mem_tests += ps.box ? 6 : 1;
        std::vector<rtb::Item> items(mem.size());
        float glo[3] = {INFINITY, INFINITY, INFINITY}, ghi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t i = 0; i < mem.size(); i++) {
            prim_box(mem[i], items[i].lo, items[i].hi);
            items[i].payload = (uint32_t)i;
            items[i].type = rtd::REF_PRIMS;
        }
        uint32_t below_bin = 0, below_wide = 0;  // the deepest instance inside this one, its sentinel included
        for (uint32_t ch : children[g])
            if (group_has_item[ch]) {
                items.push_back(group_item[ch]);
                below_bin = std::max(below_bin, 1 + chain_bin[ch]);
                below_wide = std::max(below_wide, 1 + chain_wide[ch]);
            }
        if (items.empty()) continue;
        for (const rtb::Item& it : items)
            for (int k = 0; k < 3; k++) {
                glo[k] = std::min(glo[k], it.lo[k]);
                ghi[k] = std::max(ghi[k], it.hi[k]);
            }
        const bool medium = G.kind == RT_GROUP_MEDIUM;
        // (a boundary is tested primitive by primitive, twice per ray: the limit keeps a mistake from hanging the device)
        if (medium && mem.size() > 65536) return fail(c, RT_ERR_UNSUPPORTED, "medium boundary with more than 65536 primitives");
        if (medium && media_groups.size() >= 64) return fail(c, RT_ERR_UNSUPPORTED, "more than 64 media");
        const bool single = medium || (mem_tests <= 8 && items.size() == mem.size());
        std::vector<uint32_t> order;
        const uint32_t base = (uint32_t)prims.size();
        rtb::Result r;
        if (medium) {  // the boundary's primitives in id order (Scene::intersect order), no tree
            for (const PrimSrc& ps : mem) prims.push_back(make_prim(c, ps, moving));
            r.root_ref = rtd::make_prim_ref(base, (uint32_t)std::min<size_t>(mem.size(), rtd::MAX_LEAF_PRIMS));
            r.depth = 0;
            D.depth = (uint32_t)mem.size();
        } else {
            r = rtb::build(items, leaf_max, single, base, nodes, order);
            for (uint32_t oi : order) prims.push_back(make_prim(c, mem[oi], moving));
        }
        D.root_ref = r.root_ref;
        if (!mem.empty() && mem.size() <= 64) D.flat_range = base | ((uint32_t)(mem.size() - 1) << 26);
        memcpy(D.blo, glo, sizeof(glo));
        memcpy(D.bhi, ghi, sizeof(ghi));
        chain_bin[g] = r.depth + below_bin;
        chain_wide[g] = 3u * ((r.depth + 1) / 2) + below_wide;
        if (!medium && G.parent == 0) {
            blas_depth = std::max(blas_depth, chain_bin[g]);
            blas_depth_wide = std::max(blas_depth_wide, chain_wide[g]);
        }
        rtb::Item it;
        if (G.has_xform) {
            xform_box(G.m, glo, ghi, it.lo, it.hi);
            widen(it.lo, it.hi);
        } else {
            memcpy(it.lo, glo, sizeof(glo));
            memcpy(it.hi, ghi, sizeof(ghi));
        }
        it.payload = g;
        it.type = rtd::REF_INSTANCE;
        if (medium) {
            media_groups.push_back(g);  // evaluated after the BVH walk, not a leaf
        } else if (G.parent) {
            group_item[g] = it;
            group_has_item[g] = 1;
        } else {
            top.push_back(it);
        }
    }
    std::reverse(media_groups.begin(), media_groups.end());  // (collected last to first) creation order
    std::reverse(top.begin(), top.end());
    // top level: world primitives + groups
    const auto& world = members[0];
    const size_t n_group_items = top.size();
    const uint32_t base = (uint32_t)prims.size();
    // Large worlds are built on the device (rt_lbvh.h); a flat scene has no tree, and a world of one primitive
    // has no inner node.
    const bool device_build = !flat_ok && world.size() >= 2 &&
                              (c->builder == 1 || (c->builder < 0 && world.size() >= c->gpu_build_min));
    auto ttree = std::chrono::steady_clock::now();
    rtb::Result top_r;
    std::vector<uint32_t> near0;
    bool near0_known = false, near0_ball = false, settle_fast = false;
    uint32_t lbvh_depth = 0;
    uint64_t n_nodes_total = 0;
    int rc;
    if (!device_build) {
        top.resize(n_group_items + world.size());
        for (size_t i = 0; i < world.size(); i++) {
            rtb::Item& it = top[n_group_items + i];
            prim_box(world[i], it.lo, it.hi);
            it.payload = (uint32_t)i;
            it.type = rtd::REF_PRIMS;
        }
        std::vector<uint32_t> order;
        top_r = rtb::build(top, leaf_max, false, base, nodes, order);
        for (uint32_t oi : order) prims.push_back(make_prim(c, world[oi], moving));
        // the world primitives near the first medium (rtd::Scene::near0_n), by their place in leaf order
        if (!flat_ok && !nested && !media_groups.empty() && c->near_list) {
            const rtd::Group& M0 = dgroups[media_groups[0]];
            // a boundary that is one resting sphere: what matters is the (slightly larger) ball, not its box
            const auto& bmem = members[media_groups[0] + 1];
            const bool ball = bmem.size() == 1 && bmem[0].sphere && !(!c->sph.moving.empty() && c->sph.moving[bmem[0].idx]);
            float bc[3] = {0, 0, 0}, br = 0;
            if (ball) {
                const float* a = &c->sph.c0r[4 * (size_t)bmem[0].idx];
                for (int k = 0; k < 3; k++) bc[k] = a[k];
                br = std::fabs(a[3]) * 1.001f;
            }
            auto meets = [&](const float lo[3], const float hi[3]) {
                for (int k = 0; k < 3; k++)
                    if (!(lo[k] <= M0.bhi[k] && hi[k] >= M0.blo[k])) return false;
                if (ball) {
                    double d2 = 0;
                    for (int k = 0; k < 3; k++) {
                        const double d = bc[k] < lo[k] ? lo[k] - bc[k] : (bc[k] > hi[k] ? bc[k] - hi[k] : 0.0);
                        d2 += d * d;
                    }
                    return d2 <= (double)br * br;
                }
                return true;
            };
            bool known = !M0.has_xform && M0.parent < 0;
            for (const rtb::Item& it : top)
                if (it.type == rtd::REF_INSTANCE && meets(it.lo, it.hi)) known = false;  // an instance reaches into the boundary
            for (size_t k = 0; known && k < order.size(); k++) {
                float lo[3], hi[3];
                prim_box(world[order[k]], lo, hi);
                if (meets(lo, hi)) near0.push_back(base + (uint32_t)k);
                if (near0.size() > 32) known = false;
            }
            near0_known = known;
            near0_ball = known && ball;
            // rtd::Scene::settle_fast: nothing but resting spheres and plain colours around the settled paths
            auto resting_sphere = [&](const PrimSrc& ps) { return ps.sphere && !(!c->sph.moving.empty() && c->sph.moving[ps.idx]); };
            settle_fast = near0_ball && near0.size() <= 1 && media_groups.size() <= 4;
            if (settle_fast && near0.size() == 1) {  // ... of glass
                const PrimSrc& ps = world[order[near0[0] - base]];
                settle_fast = resting_sphere(ps) && c->sph.mat[ps.idx] < c->mats.size() && c->mats[c->sph.mat[ps.idx]].kind == RT_MAT_DIELECTRIC;
            }
            for (uint32_t g : media_groups) {
                const rt_group& G = c->groups[g];
                const auto& mem = members[g + 1];
                settle_fast = settle_fast && !G.has_xform && G.parent == 0 && mem.size() == 1 && resting_sphere(mem[0]) && G.mat < c->mats.size() &&
                              c->mats[G.mat].kind == RT_MAT_ISOTROPIC && c->mats[G.mat].tex < c->texs.size() &&
                              c->texs[c->mats[G.mat].tex].kind == RT_TEX_SOLID;
            }
        }
        if ((rc = upload(c, c->d_prims, prims.data(), prims.size() * sizeof(rtd::Prim)))) return rc;
        if ((rc = upload(c, c->d_moving, moving.data(), moving.size() * sizeof(rtd::Moving)))) return rc;
        if ((rc = upload(c, c->d_nodes, nodes.data(), nodes.size() * sizeof(rtd::Node)))) return rc;
        n_nodes_total = nodes.size();
    } else {
        // A linear BVH sorts by box centre only, so one huge primitive (a ground sphere) would inflate every
        // box on its path to the root.  Primitives whose box is large against the spread of the centres are
        // therefore kept out of the sort and sit in the small host-built top level next to the instances.
        const uint32_t nw_all = (uint32_t)world.size();
        std::vector<float> side(nw_all);
        float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = 0; i < nw_all; i++) {
            float ctr[3];
            if (world[i].sphere) {
                const float* a = &c->sph.c0r[4 * (size_t)world[i].idx];
                for (int k = 0; k < 3; k++) ctr[k] = a[k];
                side[i] = 2.0f * std::fabs(a[3]);
            } else {
                float lo[3], hi[3];
                if (world[i].box)
                    box_box(c, world[i].idx, lo, hi);
                else
                    rect_box(c, world[i].idx, lo, hi);
                side[i] = 0.0f;
                for (int k = 0; k < 3; k++) {
                    ctr[k] = 0.5f * (lo[k] + hi[k]);
                    side[i] = std::max(side[i], hi[k] - lo[k]);
                }
            }
            for (int k = 0; k < 3; k++) {
                cmin[k] = std::min(cmin[k], ctr[k]);
                cmax[k] = std::max(cmax[k], ctr[k]);
            }
        }
        const float spread = std::max(std::max(cmax[0] - cmin[0], cmax[1] - cmin[1]), cmax[2] - cmin[2]);
        std::vector<uint32_t> large;
        for (uint32_t i = 0; i < nw_all; i++)
            if (side[i] > 0.125f * spread) large.push_back(i);
        if (large.size() > 4096 || nw_all - large.size() < 2) large.clear();  // no few-big / many-small structure: sort everything
        std::vector<uint8_t> is_large(nw_all, 0);
        for (uint32_t i : large) is_large[i] = 1;
        const uint32_t nw = nw_all - (uint32_t)large.size();
        std::vector<rtd::Prim> wprims;
        wprims.reserve(nw);
        for (uint32_t i = 0; i < nw_all; i++)
            if (!is_large[i]) wprims.push_back(make_prim(c, world[i], moving));
        // moving[] grows while the large primitives' records are made too; they are made now so that one upload holds all
        std::vector<rtd::Prim> large_prims(large.size());
        for (size_t k = 0; k < large.size(); k++) {
            large_prims[k] = make_prim(c, world[large[k]], moving);
            rtb::Item it;
            prim_box(world[large[k]], it.lo, it.hi);
            it.payload = (uint32_t)k;
            it.type = rtd::REF_PRIMS;
            top.push_back(it);
        }
        if ((rc = upload(c, c->d_moving, moving.data(), moving.size() * sizeof(rtd::Moving)))) return rc;
        if ((rc = ensure(c, c->d_prims, ((size_t)base + nw_all) * sizeof(rtd::Prim)))) return rc;
        if (base) HIP_TRY(c, hipMemcpy(c->d_prims.p, prims.data(), (size_t)base * sizeof(rtd::Prim), hipMemcpyHostToDevice));
        DevBuf unsorted;
        if ((rc = upload(c, unsorted, wprims.data(), (size_t)nw * sizeof(rtd::Prim)))) {
            unsorted.release();
            return rc;
        }
        rtl::Workspace ws;
        float wlo[3], whi[3];
        hipError_t e = rtl::sort_world(c->stream, ws, (const rtd::Prim*)unsorted.p, (const rtd::Moving*)c->d_moving.p, nw, shutter_t0,
                                       shutter_t1, (rtd::Prim*)c->d_prims.p + base, wlo, whi);
        unsorted.release();
        uint32_t node_base = (uint32_t)nodes.size();
        if (e == hipSuccess && !top.empty()) {
            // instances and large primitives share the top level with the world: a small host tree over them and one
            // item that stands for the whole device-built tree (its reference is patched in once the node index is known)
            constexpr uint32_t PLACEHOLDER = 0x3FFFFFFDu;
            rtb::Item it;
            memcpy(it.lo, wlo, sizeof(wlo));
            memcpy(it.hi, whi, sizeof(whi));
            it.payload = PLACEHOLDER;
            it.type = rtd::REF_INNER;
            top.push_back(it);
            std::vector<uint32_t> order;
            top_r = rtb::build(top, leaf_max, false, base + nw, nodes, order);
            if (!order.empty()) {
                std::vector<rtd::Prim> in_leaf_order(order.size());
                for (size_t k = 0; k < order.size(); k++) in_leaf_order[k] = large_prims[order[k]];
                e = hipMemcpy((rtd::Prim*)c->d_prims.p + base + nw, in_leaf_order.data(), order.size() * sizeof(rtd::Prim), hipMemcpyHostToDevice);
            }
            node_base = (uint32_t)nodes.size();
            const uint32_t from = rtd::make_ref(rtd::REF_INNER, PLACEHOLDER), to = rtd::make_ref(rtd::REF_INNER, node_base);
            for (rtd::Node& nd : nodes) {
                if (nd.c0 == from) nd.c0 = to;
                if (nd.c1 == from) nd.c1 = to;
            }
            if (top_r.root_ref == from) top_r.root_ref = to;
        } else {
            top_r.root_ref = rtd::make_ref(rtd::REF_INNER, node_base);
            top_r.depth = 0;
        }
        n_nodes_total = (uint64_t)node_base + nw - 1;
        if (e == hipSuccess && n_nodes_total >= (1ull << 30)) {
            ws.release();
            return fail(c, RT_ERR_UNSUPPORTED, "more than 2^30 BVH nodes");
        }
        if (e == hipSuccess) {
            if ((rc = ensure(c, c->d_nodes, n_nodes_total * sizeof(rtd::Node)))) {
                ws.release();
                return rc;
            }
            if (node_base) e = hipMemcpy(c->d_nodes.p, nodes.data(), (size_t)node_base * sizeof(rtd::Node), hipMemcpyHostToDevice);
        }
        if (e == hipSuccess)
            e = rtl::build_tree(c->stream, ws, (const rtd::Prim*)c->d_prims.p + base, (const rtd::Moving*)c->d_moving.p, nw, shutter_t0,
                                shutter_t1, base, (rtd::Node*)c->d_nodes.p + node_base, node_base, &lbvh_depth);
        ws.release();
        if (e != hipSuccess) return hip_fail(c, e, "device BVH build");
    }
    c->info.tree_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - ttree).count();
    const uint64_t n_prims_total = device_build ? (uint64_t)base + world.size() : prims.size();

    uint32_t need = top_r.depth + lbvh_depth + (n_inst ? 1 + blas_depth : 0) + 1;
    if (need > 64) return fail(c, RT_ERR_UNSUPPORTED, "BVH deeper than the 64-entry traversal stack");
    // Four-child walk: a node covers two binary levels and leaves up to three entries behind.  The top level, the
    // device-built world tree and the instances' trees are widened separately (each from its own root).
    const uint32_t need_wide = 3u * ((top_r.depth + 1) / 2 + (lbvh_depth + 1) / 2) + (n_inst ? 1 + blas_depth_wide : 0) + 1;
    const bool want_wide = c->bvh_width == 4 || (c->bvh_width < 0 && n_nodes_total >= c->wide_min_nodes);
    const bool wide = want_wide && !flat_ok && n_nodes_total > 0 && need_wide <= 64;
    if (wide) {
        need = need_wide;
        if ((rc = ensure(c, c->d_wide, n_nodes_total * sizeof(rtd::WideNode)))) return rc;
        hipError_t e = rtl::widen_tree(c->stream, (const rtd::Node*)c->d_nodes.p, (uint32_t)n_nodes_total, (rtd::WideNode*)c->d_wide.p);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return hip_fail(c, e, "four-child form of the BVH");
    } else {
        c->d_wide.release();
    }
    c->stack_entries = (int)(flat_ok ? 1u : need);
    c->has_groups = ng > 0;
    std::vector<uint32_t> crown_words;
    uint32_t crown_root = top_r.root_ref;
    {
        const uint32_t cap = rtk::crown_capacity(c->tune, ng > 0, wide, need);
        const bool worth = c->tune.crown_partial || n_nodes_total <= cap;  // (instances beyond CROWN_INST_SLOTS aside, it all fits)
        if (!flat_ok && !nested && n_nodes_total > 0 && n_nodes_total < rtd::CROWN_BASE && cap > 0 && worth &&
            (rc = build_crown(c, wide, top_r.root_ref, dgroups, cap, c->tune.crown_partial != 0, crown_words, crown_root)))
            return rc;
    }
    if ((rc = upload(c, c->d_crown, crown_words.data(), crown_words.size() * sizeof(uint32_t)))) return rc;

    // upload
    if ((rc = upload(c, c->d_groups, dgroups.data(), dgroups.size() * sizeof(rtd::Group)))) return rc;
    {
        std::vector<uint32_t> media_words(media_groups);  // media[], then the primitives near the first medium
        if (near0_known) media_words.insert(media_words.end(), near0.begin(), near0.end());
        if ((rc = upload(c, c->d_media, media_words.data(), media_words.size() * sizeof(uint32_t)))) return rc;
    }
    static_assert(sizeof(rt_material) == sizeof(rtd::Material), "material layout");
    static_assert(sizeof(rt_texture) == sizeof(rtd::Texture), "texture layout");
    // Device material records: a SolidColor texture is folded into the record (one dependent fetch less per
    // shade) and "reads (u,v)" is resolved here once, through checker children, instead of per hit.
    std::vector<rtd::Material> dmats(c->mats.size());
    for (size_t i = 0; i < c->mats.size(); i++) {
        const rt_material& m = c->mats[i];
        rtd::Material& d = dmats[i];
        d.kind = m.kind;
        d.tex = m.tex;
        memcpy(d.albedo, m.albedo, sizeof(d.albedo));
        d.fuzz = m.fuzz;
        d.ior = m.ior;
        d.pad = 0;
        if (m.kind == RT_MAT_METAL || m.kind == RT_MAT_DIELECTRIC) continue;
        const rt_texture& t = c->texs[m.tex];
        if (t.kind == RT_TEX_SOLID) {
            d.pad |= rtd::MF_SOLID;
            memcpy(d.albedo, t.color, sizeof(d.albedo));
        }
        // walk the checker tree (bounded: a malformed cyclic description must not hang the host)
        std::vector<uint32_t> todo{m.tex};
        for (size_t k = 0; k < todo.size() && k < 4096; k++) {
            const rt_texture& q = c->texs[todo[k]];
            if (q.kind == RT_TEX_IMAGE) d.pad |= rtd::MF_NEEDS_UV;
            if (q.kind == RT_TEX_CHECKER) {
                todo.push_back(q.odd);
                todo.push_back(q.even);
            }
        }
    }
    if ((rc = upload(c, c->d_mats, dmats.data(), dmats.size() * sizeof(rtd::Material)))) return rc;
    if ((rc = upload(c, c->d_texs, c->texs.data(), c->texs.size() * sizeof(rt_texture)))) return rc;
    std::vector<rtd::PerlinTable> ptab(n_perlin);
    for (uint32_t t = 0; t < n_perlin; t++) {
        memset(&ptab[t], 0, sizeof(rtd::PerlinTable));
        if (!c->perlin_vec.count(t)) continue;
        const auto& v = c->perlin_vec[t];
        const auto& p = c->perlin_perm[t];
        for (int i = 0; i < 256; i++)
            for (int k = 0; k < 3; k++) ptab[t].vec[i][k] = v[3 * i + k];
        for (int a = 0; a < 3; a++)
            for (int i = 0; i < 256; i++) ptab[t].perm[a][i] = p[256 * a + i];
    }
    if ((rc = upload(c, c->d_perlin, ptab.data(), ptab.size() * sizeof(rtd::PerlinTable)))) return rc;
    for (auto& b : c->d_image_data) b.release();
    c->d_image_data.assign(n_images, DevBuf());
    std::vector<rtd::Image> dimg(n_images);
    uint64_t image_bytes = 0;
    for (uint32_t t = 0; t < n_images; t++) {
        dimg[t] = rtd::Image{nullptr, 0, 0};
        if (!c->images.count(t)) continue;
        const HostImage& im = c->images[t];
        if ((rc = upload(c, c->d_image_data[t], im.rgb.data(), im.rgb.size()))) return rc;
        dimg[t] = rtd::Image{(const uint8_t*)c->d_image_data[t].p, im.w, im.h};
        image_bytes += im.rgb.size();
    }
    if ((rc = upload(c, c->d_images, dimg.data(), dimg.size() * sizeof(rtd::Image)))) return rc;

    rtd::Scene& S = c->scene;
    S.prims = (const rtd::Prim*)c->d_prims.p;
    S.moving = (const rtd::Moving*)c->d_moving.p;
    S.nodes = (const rtd::Node*)c->d_nodes.p;
    S.wide = wide ? (const rtd::WideNode*)c->d_wide.p : nullptr;
    S.groups = (const rtd::Group*)c->d_groups.p;
    S.mats = (const rtd::Material*)c->d_mats.p;
    S.texs = (const rtd::Texture*)c->d_texs.p;
    S.perlin = (const rtd::PerlinTable*)c->d_perlin.p;
    S.images = (const rtd::Image*)c->d_images.p;
    S.media = (const uint32_t*)c->d_media.p;
    S.root_ref = top_r.root_ref;
    S.n_prims = (uint32_t)n_prims_total;
    S.n_groups = ng;
    // media[] holds only the media that have a boundary: a ConstantMedium around nothing can never be hit, so it
    // is not evaluated (its Group::order still counts it, which keeps the other media's RNG salts where they were)
    S.n_media = (uint32_t)media_groups.size();
    S.near0_n = near0_known ? (uint32_t)near0.size() : rtd::NEAR_NONE;
    S.near0_ball = near0_ball ? 1u : 0u;
    S.settle_fast = settle_fast ? 1u : 0u;
    S.crown = crown_words.empty() ? nullptr : (const uint32_t*)c->d_crown.p;
    S.crown_n = crown_words.empty() ? 0u : (uint32_t)((crown_words.size() - rtd::CROWN_INST_SLOTS) / (wide ? 32 : 16));
    S.crown_root = crown_root;
    S.flat = flat_ok ? 1u : 0u;
    S.nested = nested ? 1u : 0u;
    S.stack_cap = flat_ok ? 1u : need;  // flat scenes never push
    c->n_perlin_tables = n_perlin;
    S.n_perlin = c->lds_tables ? n_perlin : 0u;  // 0: noise lookups read the table from global memory
    S.n_mats = (uint32_t)c->mats.size();
    S.n_texs = (uint32_t)c->texs.size();
    // small (flat) scenes: primitive, material and texture records follow the Perlin table into LDS (at most 8 KiB)
    S.lds_scene = (flat_ok && c->lds_tables && 32ull * (n_prims_total + S.n_mats + S.n_texs) <= 8192ull) ? 1u : 0u;
    S.world_first = base;
    S.world_count = (uint32_t)world.size();

    c->info.n_prims = n_prims_total;
    c->info.n_nodes = n_nodes_total;
    c->info.builder = device_build ? 1u : 0u;
    c->info.bvh_width = wide ? 4u : 2u;
    c->info.crown_nodes = S.crown_n;
    c->info.reserved = 0;
    c->info.n_groups = ng;
    c->info.n_media = n_media;
    c->info.max_depth = need;
    c->info.stack_entries = (uint32_t)c->stack_entries;
    c->info.device_bytes = n_prims_total * sizeof(rtd::Prim) + moving.size() * sizeof(rtd::Moving) +
                           n_nodes_total * (sizeof(rtd::Node) + (wide ? sizeof(rtd::WideNode) : 0)) + dgroups.size() * sizeof(rtd::Group) +
                           c->mats.size() * sizeof(rt_material) + c->texs.size() * sizeof(rt_texture) +
                           ptab.size() * sizeof(rtd::PerlinTable) + image_bytes;
    c->info.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - tstart).count();
    c->shutter_t0 = shutter_t0;
    c->shutter_t1 = shutter_t1;
    c->has_moving = !moving.empty();
    c->committed = true;
    return RT_OK;
}

int rt_scene_get_info(rt_ctx* c, rt_scene_info* out) {
    if (!c || !out) return RT_ERR_INVALID;
    if (!c->committed) return fail(c, RT_ERR_NOT_COMMITTED, "scene not committed");
    *out = c->info;
    return RT_OK;
}

int rt_scene_read_bvh(rt_ctx* c, void* nodes, uint64_t nodes_bytes, void* prims, uint64_t prims_bytes, uint32_t* root_ref) {
    if (!c) return RT_ERR_INVALID;
    if (!c->committed) return fail(c, RT_ERR_NOT_COMMITTED, "rt_scene_read_bvh before rt_scene_commit");
    if (nodes_bytes < c->info.n_nodes * sizeof(rtd::Node) || prims_bytes < c->info.n_prims * sizeof(rtd::Prim))
        return fail(c, RT_ERR_INVALID, "rt_scene_read_bvh: buffer too small");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (nodes && c->info.n_nodes) HIP_TRY(c, hipMemcpy(nodes, c->d_nodes.p, c->info.n_nodes * sizeof(rtd::Node), hipMemcpyDeviceToHost));
    if (prims && c->info.n_prims) HIP_TRY(c, hipMemcpy(prims, c->d_prims.p, c->info.n_prims * sizeof(rtd::Prim), hipMemcpyDeviceToHost));
    if (root_ref) *root_ref = c->scene.root_ref;
    return RT_OK;
}

namespace {
// true when the non-null pointers among the 14 are all device memory or all host memory; *dev says which
bool one_kind(const void* const in[7], const void* const out[7], bool* dev) {
    int n_dev = 0, n_host = 0;
    for (int k = 0; k < 7; k++) {
        if (in[k]) (is_device_ptr(in[k]) ? n_dev : n_host)++;
        if (out[k]) (is_device_ptr(out[k]) ? n_dev : n_host)++;
    }
    *dev = n_dev > 0;
    return n_dev == 0 || n_host == 0;
}
}  // namespace

int rt_intersect_batch(rt_ctx* c, uint64_t n, const float* ox, const float* oy, const float* oz, const float* dx,
                       const float* dy, const float* dz, const float* time, float tmin, float tmax, uint32_t seed,
                       int32_t* prim, int32_t* inst, float* t, float* normal, uint8_t* front, float* uv, float* point) {
    if (!c) return RT_ERR_INVALID;
    if (!c->committed) return fail(c, RT_ERR_NOT_COMMITTED, "rt_intersect_batch before rt_scene_commit");
    if (n == 0) return RT_OK;
    if (!ox || !oy || !oz || !dx || !dy || !dz || !time) return fail(c, RT_ERR_INVALID, "rt_intersect_batch: null ray array");
    if (n > (1ull << 32)) return fail(c, RT_ERR_INVALID, "rt_intersect_batch: more than 2^32 rays");
    HIP_TRY(c, hipSetDevice(c->device));
    const float* in[7] = {ox, oy, oz, dx, dy, dz, time};
    const void* const all_in[7] = {ox, oy, oz, dx, dy, dz, time};
    const void* const all_out[7] = {prim, inst, t, normal, front, uv, point};
    bool dev = false;
    if (!one_kind(all_in, all_out, &dev)) return fail(c, RT_ERR_INVALID, "rt_intersect_batch: host and device pointers mixed in one call");
    int rc;
    if ((rc = ensure(c, c->d_overflow, 16))) return rc;
    HIP_TRY(c, hipMemsetAsync(c->d_overflow.p, 0, 4, c->stream));
    if (dev) {
        HIP_TRY(c, rtk::launch_intersect_batch(c->stream, rtk::TraceCfg{c->stack_entries, c->has_groups, false, c->scene.nested != 0u}, c->scene, n, in, tmin, tmax,
                                               seed, prim, inst, t, normal, front, uv, point, (uint32_t*)c->d_overflow.p));
    } else {
        // staging for host callers: rays in, requested outputs back
        DevBuf din, dout;
        size_t out_bytes = n * (4 + 4 + 4 + 12 + 4 + 8 + 12);
        if ((rc = ensure(c, din, n * 7 * sizeof(float))) || (rc = ensure(c, dout, out_bytes))) {
            din.release();
            dout.release();
            return rc;
        }
        float* dr = (float*)din.p;
        const float* drays[7];
        hipError_t e = hipSuccess;
        for (int k = 0; k < 7 && e == hipSuccess; k++) {
            drays[k] = dr + (size_t)k * n;
            e = hipMemcpyAsync(dr + (size_t)k * n, in[k], n * sizeof(float), hipMemcpyHostToDevice, c->stream);
        }
        char* o = (char*)dout.p;
        int32_t* d_prim = (int32_t*)o;
        int32_t* d_inst = (int32_t*)(o + 4 * n);
        float* d_t = (float*)(o + 8 * n);
        float* d_normal = (float*)(o + 12 * n);
        uint8_t* d_front = (uint8_t*)(o + 24 * n);
        float* d_uv = (float*)(o + 28 * n);
        float* d_point = (float*)(o + 36 * n);
        if (e == hipSuccess)
            e = rtk::launch_intersect_batch(c->stream, rtk::TraceCfg{c->stack_entries, c->has_groups, false, c->scene.nested != 0u}, c->scene, n, drays, tmin, tmax, seed,
                                            prim ? d_prim : nullptr, inst ? d_inst : nullptr, t ? d_t : nullptr,
                                            normal ? d_normal : nullptr, front ? d_front : nullptr, uv ? d_uv : nullptr,
                                            point ? d_point : nullptr, (uint32_t*)c->d_overflow.p);
        auto back = [&](void* dst, const void* src, size_t bytes) {
            if (dst && e == hipSuccess) e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream);
        };
        back(prim, d_prim, 4 * n);
        back(inst, d_inst, 4 * n);
        back(t, d_t, 4 * n);
        back(normal, d_normal, 12 * n);
        back(front, d_front, n);
        back(uv, d_uv, 8 * n);
        back(point, d_point, 12 * n);
        hipError_t e2 = hipStreamSynchronize(c->stream);
        din.release();
        dout.release();
        if (e != hipSuccess) return hip_fail(c, e, "rt_intersect_batch staging");
        if (e2 != hipSuccess) return hip_fail(c, e2, "rt_intersect_batch sync");
    }
    HIP_TRY(c, hipMemcpyAsync(c->h_counts + rtk::NSHARD, c->d_overflow.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->h_counts[rtk::NSHARD]) return fail(c, RT_ERR_UNSUPPORTED, "traversal stack overflow");
    return RT_OK;
}

int rt_intersect_batch_noalloc(rt_ctx* c, uint64_t n, const float* ox, const float* oy, const float* oz, const float* dx,
                               const float* dy, const float* dz, const float* time, float tmin, float tmax, uint32_t seed,
                               int32_t* prim, int32_t* inst, float* t, float* normal, uint8_t* front, float* uv, float* point) {
    if (!c) return RT_ERR_INVALID;
    if (!c->committed) return fail(c, RT_ERR_NOT_COMMITTED, "rt_intersect_batch_noalloc before rt_scene_commit");
    if (n == 0) return RT_OK;
    if (n > rtk::NOALLOC_MAX_RAYS)
        return fail(c, RT_ERR_INVALID, "rt_intersect_batch_noalloc: more than 65536 rays in one call (the limit of its static scratch); "
                                       "split the batch or use rt_intersect_batch");
    if (!ox || !oy || !oz || !dx || !dy || !dz || !time) return fail(c, RT_ERR_INVALID, "rt_intersect_batch_noalloc: null ray array");
    HIP_TRY(c, hipSetDevice(c->device));
    // every use of the scratch goes through the device address kept at rt_ctx_create, never through the array's name
    unsigned char* const sc = c->noalloc_scratch;
    uint32_t* const flag = (uint32_t*)sc;
    const float* in[7] = {ox, oy, oz, dx, dy, dz, time};
    const rtk::TraceCfg cfg{c->stack_entries, c->has_groups, false, c->scene.nested != 0u};
    const void* const all_in[7] = {ox, oy, oz, dx, dy, dz, time};
    const void* const all_out[7] = {prim, inst, t, normal, front, uv, point};
    bool dev = false;
    if (!one_kind(all_in, all_out, &dev)) return fail(c, RT_ERR_INVALID, "rt_intersect_batch_noalloc: host and device pointers mixed in one call");
    HIP_TRY(c, hipMemsetAsync(flag, 0, 4, c->stream));  // the array keeps the last call's contents
    if (dev) {
        HIP_TRY(c, rtk::launch_intersect_batch(c->stream, cfg, c->scene, n, in, tmin, tmax, seed, prim, inst, t, normal, front, uv, point, flag));
    } else {
        // host arrays: staged through the scratch
        float* din = (float*)(sc + rtk::NOALLOC_FLAG_BYTES);
        unsigned char* o = sc + rtk::NOALLOC_FLAG_BYTES + rtk::NOALLOC_MAX_RAYS * 7 * 4;
        const float* drays[7];
        for (int k = 0; k < 7; k++) {
            drays[k] = din + (size_t)k * rtk::NOALLOC_MAX_RAYS;
            HIP_TRY(c, hipMemcpyAsync(din + (size_t)k * rtk::NOALLOC_MAX_RAYS, in[k], n * sizeof(float), hipMemcpyHostToDevice, c->stream));
        }
        const uint64_t M = rtk::NOALLOC_MAX_RAYS;
        int32_t* d_prim = (int32_t*)o;
        int32_t* d_inst = (int32_t*)(o + 4 * M);
        float* d_t = (float*)(o + 8 * M);
        float* d_normal = (float*)(o + 12 * M);
        uint8_t* d_front = (uint8_t*)(o + 24 * M);
        float* d_uv = (float*)(o + 28 * M);
        float* d_point = (float*)(o + 36 * M);
        HIP_TRY(c, rtk::launch_intersect_batch(c->stream, cfg, c->scene, n, drays, tmin, tmax, seed, prim ? d_prim : nullptr,
                                               inst ? d_inst : nullptr, t ? d_t : nullptr, normal ? d_normal : nullptr,
                                               front ? d_front : nullptr, uv ? d_uv : nullptr, point ? d_point : nullptr, flag));
        if (prim) HIP_TRY(c, hipMemcpyAsync(prim, d_prim, 4 * n, hipMemcpyDeviceToHost, c->stream));
        if (inst) HIP_TRY(c, hipMemcpyAsync(inst, d_inst, 4 * n, hipMemcpyDeviceToHost, c->stream));
        if (t) HIP_TRY(c, hipMemcpyAsync(t, d_t, 4 * n, hipMemcpyDeviceToHost, c->stream));
        if (normal) HIP_TRY(c, hipMemcpyAsync(normal, d_normal, 12 * n, hipMemcpyDeviceToHost, c->stream));
        if (front) HIP_TRY(c, hipMemcpyAsync(front, d_front, n, hipMemcpyDeviceToHost, c->stream));
        if (uv) HIP_TRY(c, hipMemcpyAsync(uv, d_uv, 8 * n, hipMemcpyDeviceToHost, c->stream));
        if (point) HIP_TRY(c, hipMemcpyAsync(point, d_point, 12 * n, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipMemcpyAsync(c->h_counts + rtk::NSHARD, flag, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->h_counts[rtk::NSHARD]) return fail(c, RT_ERR_UNSUPPORTED, "traversal stack overflow");
    return RT_OK;
}

int rt_owned_pixels(uint32_t width, uint32_t height, uint32_t tile_size, uint32_t rank, uint32_t nranks, uint32_t* out,
                    uint64_t cap, uint64_t* n_out) {
    if (!n_out || !width || !height) return RT_ERR_INVALID;
    if (!tile_size) tile_size = RT_DEFAULT_TILE;
    if (!nranks) nranks = 1;
    if (rank >= nranks) return RT_ERR_INVALID;
    const uint32_t tx = (width + tile_size - 1) / tile_size, ty = (height + tile_size - 1) / tile_size;
    uint64_t n = 0;
    for (uint64_t tile = rank; tile < (uint64_t)tx * ty; tile += nranks) {
        uint32_t x0 = (uint32_t)(tile % tx) * tile_size, y0 = (uint32_t)(tile / tx) * tile_size;
        uint32_t x1 = std::min(width, x0 + tile_size), y1 = std::min(height, y0 + tile_size);
        for (uint32_t y = y0; y < y1; y++)
            for (uint32_t x = x0; x < x1; x++) {
                if (out && n < cap) out[n] = y * width + x;
                n++;
            }
    }
    *n_out = n;
    return RT_OK;
}

// ---- frame exchange ---------------------------------------------------------------------------------------------
// RCCL is not linked: its entry points are looked up in the process at first use (the library the caller already
// loaded, if any -- a communicator must be used with the library that made it -- else librccl.so from the loader path).
namespace {
struct UniqueId {
    char internal[128];  // ncclUniqueId: NCCL_UNIQUE_ID_BYTES, passed by value
};
struct Rccl {
    std::once_flag once;
    bool ok = false;
    bool from_process = false;  // the symbols are those of an RCCL the process had loaded already
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
// Resolved once per process (call_once: contexts may be driven from several host threads).  An RCCL the process already
// holds wins -- whatever its soname or path, or linked statically: a communicator the host made (rt_comm_attach) must be
// driven by the library that made it -- and only when the process has none is librccl.so opened from the loader path.
// (RTLD_DEFAULT is a null handle on glibc, so "found in the process" is tracked apart from the dlopen handle.)
bool rccl_load() {
    std::call_once(g_rccl.once, [] {
        Rccl& r = g_rccl;
        r.from_process = dlsym(RTLD_DEFAULT, "ncclAllGather") != nullptr;
        void* h = nullptr;
        if (!r.from_process) {
            const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
            for (int k = 0; !h && k < 3; k++) h = dlopen(names[k], RTLD_NOW | RTLD_GLOBAL);
            if (!h) return;
        }
        auto sym = [&](const char* name) { return dlsym(r.from_process ? RTLD_DEFAULT : h, name); };
        r.GetUniqueId = (int (*)(void*))sym("ncclGetUniqueId");
        r.CommInitRank = (int (*)(void**, int, UniqueId, int))sym("ncclCommInitRank");
        r.CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
        r.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))sym("ncclAllGather");
        r.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather;
    });
    return g_rccl.ok;
}
int rccl_fail(rt_ctx* c, int code, const char* what) {
    return fail(c, RT_ERR_HIP, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(code) : "RCCL error"));
}
constexpr int NCCL_FLOAT32 = 7;  // ncclFloat32 of rccl.h

// Tile offsets of the deal (width, height, tile, nranks) on the device, cached per context.
int prepare_deal(rt_ctx* c, uint32_t width, uint32_t height, uint32_t ts, uint32_t nranks, rtk::TileDeal& d) {
    const uint32_t tx = (width + ts - 1) / ts, ty = (height + ts - 1) / ts;
    const uint64_t key[4] = {width, height, ts, nranks};
    if (memcmp(key, c->deal_key, sizeof(key)) != 0 || !c->d_tile_off.p) {
        std::vector<uint32_t> off((size_t)tx * ty);
        std::vector<uint64_t> fill(nranks, 0);
        for (uint64_t tile = 0; tile < (uint64_t)tx * ty; tile++) {
            const uint32_t x0 = (uint32_t)(tile % tx) * ts, y0 = (uint32_t)(tile / tx) * ts;
            const uint32_t tw = std::min(width, x0 + ts) - x0, th = std::min(height, y0 + ts) - y0;
            uint64_t& f = fill[tile % nranks];
            off[tile] = (uint32_t)f;
            f += (uint64_t)tw * th;
        }
        int rc = upload(c, c->d_tile_off, off.data(), off.size() * sizeof(uint32_t));
        if (rc) return rc;
        c->deal_block_pixels = *std::max_element(fill.begin(), fill.end());
        memcpy(c->deal_key, key, sizeof(key));
    }
    d = rtk::TileDeal{width, height, ts, tx, nranks, (const uint32_t*)c->d_tile_off.p};
    return RT_OK;
}
int check_frame_args(rt_ctx* c, uint32_t width, uint32_t height, uint32_t nranks, uint32_t rank, const void* a, const void* b) {
    if (!c) return RT_ERR_INVALID;
    if (!a || !b) return fail(c, RT_ERR_INVALID, "frame exchange: null buffer");
    if (width < 1 || height < 1 || width > 65535 || height > 65535 || (uint64_t)width * height > (1ull << 31))
        return fail(c, RT_ERR_INVALID, "frame exchange: image size out of range");
    if (nranks < 1 || rank >= nranks) return fail(c, RT_ERR_INVALID, "frame exchange: rank >= nranks");
    return RT_OK;
}
}  // namespace

int rt_frame_block_pixels(uint32_t width, uint32_t height, uint32_t tile_size, uint32_t nranks, uint64_t* out) {
    if (!out || !width || !height) return RT_ERR_INVALID;
    if (!nranks) nranks = 1;
    uint64_t most = 0;
    for (uint32_t r = 0; r < nranks; r++) {
        uint64_t n = 0;
        int rc = rt_owned_pixels(width, height, tile_size, r, nranks, nullptr, 0, &n);
        if (rc) return rc;
        most = std::max(most, n);
    }
    *out = most;
    return RT_OK;
}

int rt_pack_owned(rt_ctx* c, uint32_t width, uint32_t height, uint32_t tile_size, uint32_t rank, uint32_t nranks, const float* accum_rgb,
                  float* packed) {
    if (!nranks) nranks = 1;
    int rc = check_frame_args(c, width, height, nranks, rank, accum_rgb, packed);
    if (rc) return rc;
    if (!is_device_ptr(accum_rgb) || !is_device_ptr(packed)) return fail(c, RT_ERR_INVALID, "rt_pack_owned: device pointers expected");
    HIP_TRY(c, hipSetDevice(c->device));
    rtk::TileDeal d;
    if ((rc = prepare_deal(c, width, height, tile_size ? tile_size : RT_DEFAULT_TILE, nranks, d))) return rc;
    HIP_TRY(c, rtk::launch_pack_owned(c->stream, d, rank, accum_rgb, packed));
    return RT_OK;
}

int rt_scatter_gathered(rt_ctx* c, uint32_t width, uint32_t height, uint32_t tile_size, uint32_t nranks, const float* gathered,
                        float* frame_rgb) {
    if (!nranks) nranks = 1;
    int rc = check_frame_args(c, width, height, nranks, 0, gathered, frame_rgb);
    if (rc) return rc;
    if (!is_device_ptr(gathered) || !is_device_ptr(frame_rgb)) return fail(c, RT_ERR_INVALID, "rt_scatter_gathered: device pointers expected");
    HIP_TRY(c, hipSetDevice(c->device));
    rtk::TileDeal d;
    if ((rc = prepare_deal(c, width, height, tile_size ? tile_size : RT_DEFAULT_TILE, nranks, d))) return rc;
    HIP_TRY(c, rtk::launch_scatter_gathered(c->stream, d, gathered, c->deal_block_pixels, frame_rgb));
    return RT_OK;
}

int rt_comm_unique_id(uint8_t out[128]) {
    if (!out) return RT_ERR_INVALID;
    if (!rccl_load()) return RT_ERR_UNSUPPORTED;
    UniqueId id;
    memset(&id, 0, sizeof(id));
    if (g_rccl.GetUniqueId(&id) != 0) return RT_ERR_HIP;
    memcpy(out, id.internal, 128);
    return RT_OK;
}

int rt_comm_create(rt_ctx* c, const uint8_t unique_id[128], uint32_t rank, uint32_t nranks) {
    if (!c || !unique_id) return RT_ERR_INVALID;
    if (nranks < 1 || rank >= nranks) return fail(c, RT_ERR_INVALID, "rt_comm_create: rank >= nranks");
    if (!rccl_load()) return fail(c, RT_ERR_UNSUPPORTED, "rt_comm_create: no RCCL library (librccl.so) in this process or on the loader path");
    HIP_TRY(c, hipSetDevice(c->device));
    (void)rt_comm_destroy(c);
    UniqueId id;
    memcpy(id.internal, unique_id, 128);
    void* comm = nullptr;
    const int r = g_rccl.CommInitRank(&comm, (int)nranks, id, (int)rank);
    if (r != 0 || !comm) return rccl_fail(c, r, "ncclCommInitRank");
    c->comm = comm;
    c->comm_owned = true;
    c->comm_rank = rank;
    c->comm_nranks = nranks;
    return RT_OK;
}

int rt_comm_attach(rt_ctx* c, void* nccl_comm, uint32_t rank, uint32_t nranks) {
    if (!c) return RT_ERR_INVALID;
    if (nranks < 1 || rank >= nranks) return fail(c, RT_ERR_INVALID, "rt_comm_attach: rank >= nranks");
    if (nccl_comm && !rccl_load()) return fail(c, RT_ERR_UNSUPPORTED, "rt_comm_attach: no RCCL library in this process");
    if (!nccl_comm && nranks != 1) return fail(c, RT_ERR_INVALID, "rt_comm_attach: a null communicator only stands for a single rank");
    (void)rt_comm_destroy(c);
    c->comm = nccl_comm;
    c->comm_owned = false;
    c->comm_rank = rank;
    c->comm_nranks = nranks;
    return RT_OK;
}

int rt_comm_destroy(rt_ctx* c) {
    if (!c) return RT_ERR_INVALID;
    if (c->comm && c->comm_owned && g_rccl.ok) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        (void)g_rccl.CommDestroy(c->comm);
    }
    c->comm = nullptr;
    c->comm_owned = false;
    c->comm_rank = 0;
    c->comm_nranks = 1;
    return RT_OK;
}

int rt_gather_frame(rt_ctx* c, uint32_t width, uint32_t height, uint32_t tile_size, const float* accum_rgb, float* frame_rgb) {
    if (!c) return RT_ERR_INVALID;
    const uint32_t nranks = c->comm_nranks, rank = c->comm_rank;
    int rc = check_frame_args(c, width, height, nranks, rank, accum_rgb, frame_rgb);
    if (rc) return rc;
    if (nranks > 1 && !c->comm) return fail(c, RT_ERR_INVALID, "rt_gather_frame: no communicator (rt_comm_create / rt_comm_attach)");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t frame_bytes = (size_t)width * height * 3 * sizeof(float);
    const bool in_dev = is_device_ptr(accum_rgb), out_dev = is_device_ptr(frame_rgb);
    const float* d_in = accum_rgb;
    float* d_out = frame_rgb;
    if (!in_dev) {
        if ((rc = ensure(c, c->d_xin, frame_bytes))) return rc;
        HIP_TRY(c, hipMemcpyAsync(c->d_xin.p, accum_rgb, frame_bytes, hipMemcpyHostToDevice, c->stream));
        d_in = (const float*)c->d_xin.p;
    }
    if (!out_dev) {
        if ((rc = ensure(c, c->d_xout, frame_bytes))) return rc;
        d_out = (float*)c->d_xout.p;
    }
    rtk::TileDeal d;
    if ((rc = prepare_deal(c, width, height, tile_size ? tile_size : RT_DEFAULT_TILE, nranks, d))) return rc;
    const uint64_t block = c->deal_block_pixels;  // every rank's share, padded to the largest
    if ((rc = ensure(c, c->d_gathered, (size_t)nranks * block * 3 * sizeof(float)))) return rc;
    float* gathered = (float*)c->d_gathered.p;
    if (nranks == 1 && !c->comm) {
        // one rank owns every tile: its packed block is the gathered buffer
        HIP_TRY(c, rtk::launch_pack_owned(c->stream, d, 0, d_in, gathered));
    } else {  // (a communicator of one rank runs the collective all the same: the code N ranks run, on one GPU)
        if ((rc = ensure(c, c->d_packed, (size_t)block * 3 * sizeof(float)))) return rc;
        HIP_TRY(c, hipMemsetAsync(c->d_packed.p, 0, (size_t)block * 3 * sizeof(float), c->stream));  // the padding of a smaller share
        HIP_TRY(c, rtk::launch_pack_owned(c->stream, d, rank, d_in, (float*)c->d_packed.p));
        // ONE collective per frame: every rank's block goes to every rank, each over its own xGMI link
        const int r = g_rccl.AllGather(c->d_packed.p, gathered, (size_t)block * 3, NCCL_FLOAT32, c->comm, c->stream);
        if (r != 0) return rccl_fail(c, r, "ncclAllGather");
    }
    HIP_TRY(c, rtk::launch_scatter_gathered(c->stream, d, gathered, block, d_out));
    if (!out_dev) HIP_TRY(c, hipMemcpyAsync(frame_rgb, d_out, frame_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_profile_enable(rt_ctx* c, int flags) {
    if (!c) return RT_ERR_INVALID;
    c->profile = (flags & 1) != 0;
    c->count_visits = (flags & 2) != 0;
    return RT_OK;
}

int rt_render(rt_ctx* c, const rt_camera* cam, const rt_render_params* p, float* accum_rgb, rt_render_stats* stats) {
    if (!c) return RT_ERR_INVALID;
    if (!c->committed) return fail(c, RT_ERR_NOT_COMMITTED, "rt_render before rt_scene_commit");
    if (!cam || !p || !accum_rgb) return fail(c, RT_ERR_INVALID, "rt_render: null argument");
    if (p->width < 2 || p->height < 2 || p->width > 65535 || p->height > 65535 || (uint64_t)p->width * p->height > (1ull << 31))
        return fail(c, RT_ERR_INVALID, "rt_render: image size out of range (2..65535 per side, at most 2^31 pixels)");
    if (p->sample1 < p->sample0) return fail(c, RT_ERR_INVALID, "rt_render: sample1 < sample0");
    if (p->max_depth == 0 || p->max_depth > 253) return fail(c, RT_ERR_INVALID, "rt_render: max_depth must be 1..253");
    // camera.rs:63 draws the ray time with gen_range(time_start..time_end), which panics on an empty range
    if (!(cam->time_start < cam->time_end) || !std::isfinite(cam->time_end - cam->time_start))
        return fail(c, RT_ERR_INVALID, "rt_render: camera needs finite time_start < time_end");
    if (!std::isfinite(cam->lens_radius)) return fail(c, RT_ERR_INVALID, "rt_render: camera lens_radius is not finite");
    // The boxes of moving spheres cover the shutter interval given to rt_scene_commit (MovingSphere::bounding_box,
    // shapes.rs:236-249, over the t0..t1 of BVHNode::from_scene): a ray time outside it would be culled by the tree
    // and still hit in a flat scene.  gen_range(time_start..time_end) draws from [time_start, time_end).
    if (c->has_moving && !(cam->time_start >= std::min(c->shutter_t0, c->shutter_t1) && cam->time_end <= std::max(c->shutter_t0, c->shutter_t1)))
        return fail(c, RT_ERR_INVALID, "rt_render: camera time range lies outside the shutter interval of rt_scene_commit (the scene has moving spheres)");
    const uint32_t ts = p->tile_size ? p->tile_size : RT_DEFAULT_TILE, nranks = p->nranks ? p->nranks : 1;
    if (p->rank >= nranks) return fail(c, RT_ERR_INVALID, "rt_render: rank >= nranks");
    HIP_TRY(c, hipSetDevice(c->device));
    auto wall0 = std::chrono::steady_clock::now();
    int rc;

    // owned pixel list (cached per frame geometry)
    uint64_t key[5] = {p->width, p->height, ts, p->rank, nranks};
    if (memcmp(key, c->pixels_key, sizeof(key)) != 0 || !c->d_pixels.p) {
        uint64_t n = 0;
        rt_owned_pixels(p->width, p->height, ts, p->rank, nranks, nullptr, 0, &n);
        std::vector<uint32_t> px(n);
        rt_owned_pixels(p->width, p->height, ts, p->rank, nranks, px.data(), n, &n);
        for (auto& v : px) v = ((v / p->width) << 16) | (v % p->width);  // the kernels want row << 16 | column
        if ((rc = upload(c, c->d_pixels, px.data(), n * sizeof(uint32_t)))) return rc;
        memcpy(c->pixels_key, key, sizeof(key));
        c->n_owned = n;
    }
    const uint64_t n_owned = c->n_owned;
    const uint32_t n_samples_total = p->sample1 - p->sample0;

    const bool accum_dev = is_device_ptr(accum_rgb);
    const size_t accum_bytes = (size_t)p->width * p->height * 3 * sizeof(float);
    float* d_accum = accum_rgb;
    if (!accum_dev) {
        if ((rc = ensure(c, c->d_accum, accum_bytes))) return rc;
        d_accum = (float*)c->d_accum.p;
        HIP_TRY(c, hipMemcpyAsync(d_accum, accum_rgb, accum_bytes, hipMemcpyHostToDevice, c->stream));
    }

    rtk::RenderConst rcst;
    static_assert(sizeof(rt_camera) == sizeof(rtd::Camera), "camera layout");
    memcpy(&rcst.cam, cam, sizeof(rt_camera));
    memcpy(rcst.background, p->background, sizeof(float) * 3);
    rcst.width = p->width;
    rcst.height = p->height;
    rcst.seed = p->seed;
    rcst.max_depth = p->max_depth;
    rcst.radiance_cleared = (p->background[0] == 0.0f && p->background[1] == 0.0f && p->background[2] == 0.0f) ? 1u : 0u;

    rt_render_stats st{};
    c->ev_used.clear();
    c->ev_next = 0;
    // Pipeline choice: the fused bounce kernel moves the least data and wins while a ray visits few nodes
    // (flat scenes, random_world's 370 nodes).  Deeper trees take the split pipeline, whose trace kernel keeps
    // its lanes busy across rays at half the registers: final_scene (1.9k nodes) +9%, 10M spheres +55%.
    // mode 2 (hybrid): camera rays are coherent and cost no queue read, so bounce 0 runs fused; the scattered
    // rays of the later bounces go through the split kernels
    // flat scenes whose shading is cheap and whose paths are not held long by media go whole-path (mode 3): the path
    // state never leaves the registers (two_spheres 50.5 -> 33.5 ms per frame).  With a noise texture or media the
    // per-bounce kernels stay ahead: every regeneration round runs the camera code next to a shading step that is
    // already VALU-bound (two_perlin_spheres 39 vs 43 ms), and long paths drain slowly at the end of a wave's run
    // (cornell_smoke 130 vs 165 ms; cornell_box, closed as well, is level at best).  Instances stand for those here.
    const bool whole_paths = c->scene.flat && c->n_perlin_tables == 0 && !c->has_groups && c->whole_paths_auto;
    // Once the nodes the kernels walk no longer fit the Infinity Cache, the fused camera-ray launch (whole stack in
    // LDS, one ray per lane from start to end) falls behind the persistent trace waves, which refill their lanes
    // and keep only eight stack entries in LDS: 10 M spheres 78 vs 91 ms per 16-spp frame with the split pipeline,
    // 1 M level, 0.3 M and smaller 8-25 % the other way.
    const uint64_t walked_node_bytes = c->info.n_nodes * (c->info.bvh_width == 4 ? sizeof(rtd::WideNode) : sizeof(rtd::Node));
    const int deep = walked_node_bytes >= c->split_min_node_bytes ? 0 : c->deep_pipeline;
    const int mode = c->pipeline >= 0 ? c->pipeline
                                      : (whole_paths ? 3 : (c->info.n_nodes <= c->fused_node_limit ? 1 : deep));
    // (flat scenes have no nodes to speak of and always fall on the fused side)
    const bool fused = mode == 1 || mode == 3;  // (the instrumented passes of mode 3 run the fused kernels)
    st.pipeline = (uint32_t)mode;
    if (n_owned && n_samples_total) {
        if ((rc = reserve_paths(c, n_owned * (uint64_t)n_samples_total, !fused))) return rc;
        HIP_TRY(c, hipMemsetAsync(c->d_rays.p, 0, 256, c->stream));
        HIP_TRY(c, hipMemsetAsync(c->d_overflow.p, 0, 4, c->stream));
        const rtk::Queues q{(uint32_t*)c->d_counts.p, c->cap_shard};
        const rtk::TraceCfg tcfg{c->stack_entries, c->has_groups, c->count_visits, c->scene.nested != 0u};
        unsigned long long* d_stats = (unsigned long long*)c->d_rays.p;
        uint32_t* d_ovf = (uint32_t*)c->d_overflow.p;
        float4* d_rad = (float4*)c->d_radiance.p;
        // settled paths (see k_shade / k_settled): only the uninstrumented split kernels of scenes with a near list
        const bool settle = !fused && !c->count_visits && c->settle_ahead && rtk::scene_settles(c->scene);
        const uint64_t pix_chunk = std::min<uint64_t>(n_owned, c->capacity);
        for (uint64_t p0 = 0; p0 < n_owned; p0 += pix_chunk) {
            const uint32_t npix = (uint32_t)std::min<uint64_t>(pix_chunk, n_owned - p0);
            const uint32_t s_batch = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_samples_total, c->capacity / npix));
            for (uint32_t s = 0; s < n_samples_total; s += s_batch) {
                const uint32_t ns = std::min(s_batch, n_samples_total - s);
                // samples per run of 64 slots (see rtk::Batch); falls back to one where the batch does not divide evenly
                uint32_t q_log2 = c->slot_samples_log2;
                while (q_log2 && ((npix & ((64u >> q_log2) - 1u)) || (ns & ((1u << q_log2) - 1u)))) q_log2--;
                const rtk::Batch batch{(const uint32_t*)c->d_pixels.p + p0, npix, p->sample0 + s, ns, 1.0 / ((double)npix * (double)(1u << q_log2)), q_log2};
                const uint32_t n_batch = npix * ns;
                // blocks per shard for the next launches, from the largest queue known so far
                uint32_t bps = (n_batch + rtk::BLOCK * rtk::NSHARD - 1) / (rtk::BLOCK * rtk::NSHARD);
                st.batches++;
                st.paths += n_batch;
                if (rcst.radiance_cleared) HIP_TRY(c, hipMemsetAsync(d_rad, 0, (size_t)n_batch * sizeof(float4), c->stream));
                if (mode == 3 && !c->count_visits) {
                    // whole paths with regeneration: one launch, nothing queued
                    {
                        ScopedTimer tm(c, KS_PATHS);
                        HIP_TRY(c, rtk::launch_paths(c->stream, c->tune, tcfg, c->scene, rcst, batch, d_rad, d_stats, d_ovf));
                    }
                    st.bounce_launches++;
                    ScopedTimer tm(c, KS_ACCUM);
                    HIP_TRY(c, rtk::launch_accumulate(c->stream, (const float4*)c->d_radiance.p, batch, p->width, d_accum));
                    continue;
                }
                if (mode != 0) {
                    HIP_TRY(c, hipMemsetAsync(q.counts, 0, 6 * rtk::NSHARD * sizeof(uint32_t), c->stream));
                } else {
                    HIP_TRY(c, hipMemsetAsync(q.counts + 3 * rtk::NSHARD, 0, 3 * rtk::NSHARD * sizeof(uint32_t), c->stream));
                    ScopedTimer tm(c, KS_RAYGEN);
                    HIP_TRY(c, rtk::launch_raygen(c->stream, c->tune, rcst, batch, path_state(c, 0), q));
                }
                for (uint32_t b = 0; b < p->max_depth; b++) {
                    const rtk::PathState in = path_state(c, b & 1), out = path_state(c, (b & 1) ^ 1);
                    if (fused || (mode == 2 && b == 0)) {
                        ScopedTimer tm(c, KS_BOUNCE);
                        HIP_TRY(c, rtk::launch_bounce(c->stream, c->tune, tcfg, c->scene, rcst, batch, in, out, q, b, bps, d_rad, d_stats, d_ovf));
                    } else {
                        {
                            ScopedTimer tm(c, KS_TRACE);
                            HIP_TRY(c, rtk::launch_trace(c->stream, c->tune, tcfg, c->scene, in, q, b, bps, (float4*)c->d_hits.p, d_stats, d_ovf, c->trace_spill));
                        }
                        {
                            ScopedTimer tm(c, KS_SHADE);
                            HIP_TRY(c, rtk::launch_shade(c->stream, c->tune, c->scene, rcst, in, out, (const float4*)c->d_hits.p, q, b, bps, d_rad, d_stats,
                                                         settle));
                        }
                        if (settle && b > 0) {
                            // the paths the launch before left at the far end of `in`, each followed for as long as it stays settled
                            ScopedTimer tm(c, KS_SETTLED);
                            HIP_TRY(c, rtk::launch_settled(c->stream, c->tune, c->scene, rcst, in, out, q, b, d_rad, d_stats));
                        }
                    }
                    st.bounce_launches++;
                    // read the queue lengths back now and then: stop when no path is left, shrink the grid
                    const bool check = b < 4 || (b & 3) == 3;
                    if (check && b + 1 < p->max_depth) {
                        HIP_TRY(c, hipMemcpyAsync(c->h_counts, q.counts + ((b + 1) % 3) * rtk::NSHARD, rtk::NSHARD * sizeof(uint32_t),
                                                  hipMemcpyDeviceToHost, c->stream));
                        uint32_t* h_settled = c->h_counts + rtk::NSHARD + 16;
                        if (!fused)
                            HIP_TRY(c, hipMemcpyAsync(h_settled, q.counts + (3 + (b + 1) % 3) * rtk::NSHARD, rtk::NSHARD * sizeof(uint32_t),
                                                      hipMemcpyDeviceToHost, c->stream));
                        HIP_TRY(c, hipStreamSynchronize(c->stream));
                        uint32_t longest = 0;
                        uint64_t live = 0;
                        for (int k = 0; k < rtk::NSHARD; k++) {
                            // (a shard's queue holds its settled paths at the far end: k_settled takes those with a grid of its
                            // own; each that leaves it joins the front of the next queue, so the front can grow by that many)
                            longest = std::max(longest, c->h_counts[k] + (fused ? 0u : h_settled[k]));
                            live += c->h_counts[k] + (fused ? 0u : h_settled[k]);
                        }
                        if (c->debug_live) {  // RTAMD_DEBUG_LIVE=1: paths alive after each checked bounce (front of the queues / settled end)
                            uint64_t settled = 0;
                            for (int k = 0; k < rtk::NSHARD && !fused; k++) settled += h_settled[k];
                            fprintf(stderr, "[rtamd] after bounce %u: %llu paths alive, %llu of them settled ahead\n", b, (unsigned long long)live,
                                    (unsigned long long)settled);
                        }
                        if (longest == 0) break;
                        bps = (longest + rtk::BLOCK - 1) / rtk::BLOCK;
                        if (live <= c->tail_paths && !c->count_visits) {
                            // few paths left: one launch walks each of them to its end (same draws, same bits).  The
                            // tail takes about as long as its longest path run serially (0.1 ms on a flat scene, ~0.9 ms
                            // through a tree), which is still well below the ~45 launches + read-backs it replaces
                            // (C1, 400x225x10: 3.1 -> 1.4 ms per frame).
                            ScopedTimer tm(c, KS_TAIL);
                            HIP_TRY(c, rtk::launch_finish(c->stream, tcfg, c->scene, rcst, out, q, b + 1, bps, d_rad, d_stats, d_ovf));
                            st.tail_launches++;
                            break;
                        }
                    }
                }
                {
                    ScopedTimer tm(c, KS_ACCUM);
                    HIP_TRY(c, rtk::launch_accumulate(c->stream, (const float4*)c->d_radiance.p, batch, p->width, d_accum));
                }
            }
        }
        HIP_TRY(c, hipMemcpyAsync(c->h_rays, c->d_rays.p, 256, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->h_counts + rtk::NSHARD, c->d_overflow.p, 4, hipMemcpyDeviceToHost, c->stream));
    }
    if (!accum_dev) HIP_TRY(c, hipMemcpyAsync(accum_rgb, d_accum, accum_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (n_owned && n_samples_total) {
        st.tail_rays = c->h_rays[8];
        st.settled_rays = c->h_rays[16];
        st.settled_entries = c->h_rays[30];
        if (c->print_hist) { fprintf(stderr, "[rtamd] hist:"); for (int k = 17; k < 32; k++) fprintf(stderr, " %llu", c->h_rays[k]); fprintf(stderr, "\n"); }
        st.rays = c->h_rays[0] + st.tail_rays;
        st.node_visits = c->h_rays[1];
        st.prim_tests = c->h_rays[2];
        if (c->print_sections && c->h_rays[29]) {  // builds with -DRT_SECTIONS: wave cycles per section of k_settled's loop
            const unsigned long long* h = c->h_rays;
            fprintf(stderr, "[rtamd] k_settled wave cycles per round: refill %.0f  other media %.0f  shade %.0f  settle test %.0f  (rounds %llu, rays %llu)\n",
                    h[24] / (double)h[29], h[25] / (double)h[29], h[26] / (double)h[29], h[27] / (double)h[29], h[29], h[16]);
        }
        if (c->count_visits && c->print_sections) {
            // instrumented fused kernel: wave cycles per section, summed over waves (words 3..6), wave tiles (7)
            const unsigned long long* h = c->h_rays;
            double tot = (double)(h[3] + h[4] + h[5] + h[6]);
            if (tot > 0)
                fprintf(stderr, "[rtamd] wave cycles per tile: fetch/raygen %.0f  trace %.0f  shade %.0f  compact %.0f  (tiles %llu)\n",
                        h[3] / (double)h[7], h[4] / (double)h[7], h[5] / (double)h[7], h[6] / (double)h[7], h[7]);
        }
        if (c->count_visits && c->print_trace_lanes && c->h_rays[12]) {
            // instrumented persistent trace kernel: how full its waves were
            const unsigned long long* h = c->h_rays;
            fprintf(stderr,
                    "[rtamd] k_trace lanes: rounds %llu, lanes holding a ray %.1f/64 | inner steps %llu at %.1f lanes | leaf steps %llu at %.1f lanes "
                    "| per round %.2f inner + %.2f leaf steps | non-finite rays %llu\n",
                    h[12], h[13] / (double)h[12], h[9], h[14] / (double)std::max(1ull, h[9]), h[10], h[11] / (double)std::max(1ull, h[10]),
                    h[9] / (double)h[12], h[10] / (double)h[12], h[15]);
        }
        if (c->h_counts[rtk::NSHARD]) return fail(c, RT_ERR_UNSUPPORTED, "traversal stack overflow");
    }
    if (c->profile) {
        double ms[KS_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (auto& u : c->ev_used) {
            float t = 0;
            if (hipEventElapsedTime(&t, c->ev_pool[u.second.first], c->ev_pool[u.second.second]) == hipSuccess) ms[u.first] += t;
            if (c->debug_live) fprintf(stderr, "[rtamd] launch slot %d: %.3f ms\n", u.first, t);  // (1 trace, 2 shade, 4 bounce, 7 settled)
            if (u.first == KS_TRACE) st.trace_launches++;
            if (u.first == KS_SHADE) st.shade_launches++;
            if (u.first == KS_BOUNCE) st.bounce_kernel_launches++;
            if (u.first == KS_PATHS) st.paths_launches++;
            if (u.first == KS_SETTLED) st.settled_launches++;
        }
        st.raygen_ms = ms[KS_RAYGEN];
        st.trace_ms = ms[KS_TRACE];
        st.shade_ms = ms[KS_SHADE];
        st.accumulate_ms = ms[KS_ACCUM];
        st.bounce_ms = ms[KS_BOUNCE];
        st.tail_ms = ms[KS_TAIL];
        st.paths_ms = ms[KS_PATHS];
        st.settled_ms = ms[KS_SETTLED];
    }
    st.render_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    if (stats) *stats = st;
    return RT_OK;
}

int rt_resolve_u8(rt_ctx* c, const float* sum_rgb, uint64_t n_pixels, uint32_t spp, uint8_t* out) {
    return rt_resolve_u8_transfer(c, sum_rgb, n_pixels, spp, RT_TRANSFER_LINEAR, out);
}

int rt_resolve_u8_transfer(rt_ctx* c, const float* sum_rgb, uint64_t n_pixels, uint32_t spp, int transfer, uint8_t* out) {
    if (!c || !sum_rgb || !out || !spp) return fail(c, RT_ERR_INVALID, "rt_resolve_u8: bad argument");
    if (transfer != RT_TRANSFER_LINEAR && transfer != RT_TRANSFER_SQRT) return fail(c, RT_ERR_INVALID, "rt_resolve_u8: unknown transfer");
    if (n_pixels == 0) return RT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const uint64_t n = n_pixels * 3;
    const bool in_dev = is_device_ptr(sum_rgb), out_dev = is_device_ptr(out);
    DevBuf din, dout;
    int rc = RT_OK;
    const float* d_in = sum_rgb;
    uint8_t* d_out = out;
    hipError_t e = hipSuccess;
    if (!in_dev) {
        if ((rc = ensure(c, din, n * sizeof(float)))) return rc;
        e = hipMemcpyAsync(din.p, sum_rgb, n * sizeof(float), hipMemcpyHostToDevice, c->stream);
        d_in = (const float*)din.p;
    }
    if (!out_dev && e == hipSuccess) {
        if ((rc = ensure(c, dout, n))) {
            din.release();
            return rc;
        }
        d_out = (uint8_t*)dout.p;
    }
    if (e == hipSuccess) e = rtk::launch_resolve(c->stream, d_in, n, spp, transfer == RT_TRANSFER_SQRT, d_out);
    if (!out_dev && e == hipSuccess) e = hipMemcpyAsync(out, d_out, n, hipMemcpyDeviceToHost, c->stream);
    hipError_t e2 = hipStreamSynchronize(c->stream);
    din.release();
    dout.release();
    if (e != hipSuccess) return hip_fail(c, e, "rt_resolve_u8");
    if (e2 != hipSuccess) return hip_fail(c, e2, "rt_resolve_u8 sync");
    return RT_OK;
}

}  // extern "C"
