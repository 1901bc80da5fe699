// HIP kernels of the wavefront path tracer (gfx950 / CDNA4, wave64, 256-thread blocks).
//
//   k_bounce<PRIMARY>  fused bounce: (camera ray | queued ray) -> closest hit -> shade -> compacted queue
//   k_raygen           camera samples -> path state            (main.rs:471-474, camera.rs:54-65)
//   k_trace            closest hit of every live path          (bvh_node.rs:145-164 + shapes.rs intersect fns)
//   k_shade            hit record, emitted, scatter, compaction (main.rs:44-63, materials.rs, texture.rs)
//   k_finish           the last few live paths, each walked to its end by one lane (no queue, no launches)
//   k_paths            whole paths per lane with regeneration: no queues at all (flat scenes)
//   k_accumulate       per-pixel sum in sample order            (main.rs:478-481)
//   k_resolve          quantise                                 (main.rs:307-313)
//   k_intersect_batch  hit() path on caller-supplied rays
//
// Two pipelines share the device functions of rt_devfn.inl:
//   fused: one k_bounce launch per bounce.  A ray's state is read once and written once per bounce, the
//          hit record never touches memory and bounce 0 generates its camera rays in registers.
//   split: k_raygen, then k_trace + k_shade per bounce.  k_trace keeps persistent waves busy across rays
//          (free lanes take the next rays of the wave's run), which pays when rays walk deep trees.
// Scenes of a few dozen primitives are traced without the tree in both (trace_flat).
// Live paths sit in NSHARD queues (see rt_kernels.h); queue s is served by blocks s, s+NSHARD, ...
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>

#include "rt_devfn.inl"

namespace {

#ifndef PATHS_WAVES
#define PATHS_WAVES 6
#endif
#ifndef BOUNCE_WAVES
#define BOUNCE_WAVES 5
#endif
struct PathRegs {
    f3 o, d, T;
    float time;
    uint32_t k0, k1, slot;
    // Bounce number of the ray (o, d): how many scatterings lie behind it.  In most launches every path is at the launch's
    // own bounce; paths that k_settled has followed in registers are ahead of it, so the number travels with the path.
    uint32_t bounce = 0u;
    // != 0: the hit of this ray, found out by the kernel that made the ray ("settled ahead", see k_shade): > 0 the distance at
    // which the first medium takes it, < 0 minus the distance at which it meets the one primitive near that medium.  Such
    // a path waits at the far end of its queue, where the trace kernel does not look.
    float settled = 0.0f;
};

// Jitter + Camera::get_ray for path slot `slot` of the batch (main.rs:471-474, camera.rs:54-65).
// slot = sample_local * npix + pixel_local: neighbouring lanes are neighbouring pixels of one tile.
DEV void camera_path(const rtk::RenderConst& rc, const rtk::Batch& b, uint32_t slot, PathRegs& p) {
    // slot / npix through the f64 reciprocal: the product is within 2^-20 of the quotient, so the truncation is
    // off by at most one, which the remainder test repairs (an integer divide costs ~30 instructions)
    const uint32_t qn = b.npix << b.q_log2;
    uint32_t sl = (uint32_t)((double)slot * b.inv_qnpix);
    uint32_t pl = slot - sl * qn;
    if ((int32_t)pl < 0) {
        sl--;
        pl += qn;
    } else if (pl >= qn) {
        sl++;
        pl -= qn;
    }
    if (b.q_log2) {  // (the same in every lane) see rtk::Batch: 64 slots = a few pixels x a few samples
        const uint32_t lane = pl & 63u, w_log2 = 6u - b.q_log2;
        sl = (sl << b.q_log2) + (lane >> w_log2);
        pl = ((pl >> 6) << w_log2) + (lane & ((1u << w_log2) - 1u));
    }
    const uint32_t sample = b.sample0 + sl;
    const uint32_t packed = b.pixel_list[pl];  // row << 16 | column
    const uint32_t i = packed & 0xFFFFu, j = packed >> 16;
    const uint32_t pixel = j * rc.width + i;
    Rng rng;
    stream_key(rc.seed, pixel, sample, rng.k0, rng.k1);
    rng.n = CTR_CAMERA;
    // main.rs:472-473; the denominators are at least 1 and below 2^16, the numerators zero or at least 2^-24
    const float wd = (float)(rc.width - 1), hd = (float)(rc.height - 1);
    float u = div_by((float)i + rng.f32(), wd, recip_refined(wd));
    float v = div_by((float)(rc.height - 1 - j) + rng.f32(), hd, recip_refined(hd));
    // vec3_helpers.rs:47-60: the first point with |p|^2 >= 1 (sic).  The reference draws it even when the
    // aperture is 0 and then scales it by 0; the lens draws have their own counters, so skipping them there
    // changes nothing else.
    f3 offset{0.0f, 0.0f, 0.0f};
    if (rc.cam.lens_radius != 0.0f) {
        Rng lens{rng.k0, rng.k1, CTR_LENS};
        float px, py;
        for (;;) {
            px = lens.range_pm1();
            py = lens.range_pm1();
            if ((px * px + py * py) + 0.0f * 0.0f >= 1.0f) break;
        }
        float rdx = rc.cam.lens_radius * px, rdy = rc.cam.lens_radius * py;
        offset = ld3(rc.cam.u) * rdx + ld3(rc.cam.v) * rdy;
    }
    p.o = ld3(rc.cam.origin) + offset;
    p.d = ld3(rc.cam.llc) + u * ld3(rc.cam.horizontal) + v * ld3(rc.cam.vertical) - ld3(rc.cam.origin) - offset;
    p.time = rng.range(rc.cam.time_start, rc.cam.time_end);
    p.T = {1.0f, 1.0f, 1.0f};
    p.k0 = rng.k0;
    p.k1 = rng.k1;
    p.slot = slot;
    p.bounce = 0u;
}

DEV void load_path(const rtk::PathState& st, uint32_t idx, PathRegs& p) {
    float4 ot = st.o_time[idx], ds = st.d_slot[idx], tm = st.thr_meta[idx];
    uint2 key = st.key[idx];
    p.o = {ot.x, ot.y, ot.z};
    p.time = ot.w;
    p.d = {ds.x, ds.y, ds.z};
    p.slot = __float_as_uint(ds.w);
    p.T = {tm.x, tm.y, tm.z};
    p.k0 = key.x;
    p.k1 = key.y;
    p.bounce = __float_as_uint(tm.w);
    p.settled = 0.0f;
}
DEV void store_path(const rtk::PathState& st, uint32_t idx, const PathRegs& p) {
    st.o_time[idx] = make_float4(p.o.x, p.o.y, p.o.z, p.time);
    st.d_slot[idx] = make_float4(p.d.x, p.d.y, p.d.z, __uint_as_float(p.slot));
    st.thr_meta[idx] = make_float4(p.T.x, p.T.y, p.T.z, __uint_as_float(p.bounce));
    st.key[idx] = make_uint2(p.k0, p.k1);
}

// One step of ray_color (main.rs:44-63) after the hit is known.  Returns true when the path goes on (p is
// then the scattered ray of bounce + 1 with the updated throughput); otherwise its radiance has been written.
// Radiance of a path that has ended.  Adding a zero (of either sign) to a pixel's sum changes nothing, so zeros are
// not written when the buffer starts out cleared.
DEV void end_path(const rtk::RenderConst& rc, float4* __restrict__ radiance, uint32_t slot, f3 L) {
    if (rc.radiance_cleared && L.x == 0.0f && L.y == 0.0f && L.z == 0.0f) return;
    radiance[slot] = make_float4(L.x, L.y, L.z, 0.0f);
}
template <bool NESTED>
DEV bool shade_path(const Scene& S, const LdsTables& tabs, const rtk::RenderConst& rc, PathRegs& p, const Hit& h, uint32_t bounce,
                    float4* __restrict__ radiance) {
    if (h.prim == HIT_NONE) {  // main.rs:60-62
        end_path(rc, radiance, p.slot, p.T * ld3(rc.background));
        return false;
    }
    // material first, so that uv (acos/atan2) is only evaluated for textures that read it
    uint32_t mi;
    if (h.prim & HIT_MEDIUM_BIT)
        mi = S.groups[h.prim & 0x3FFFFFFFu].mat;
    else
    {
        float4 A, B;
        load_record(tabs, tabs.prims, S.prims, h.prim, A, B);
        mi = __float_as_uint(B.y) >> 8;
    }
    const Material M = load_material(S, tabs, mi);
    Record rec;
    if (material_needs_uv(M))
        make_record<true, NESTED>(S, tabs, p.o, p.d, p.time, h, rec);
    else
        make_record<false, NESTED>(S, tabs, p.o, p.d, p.time, h, rec);
    Rng rng{p.k0, p.k1, bounce << 24};
    f3 att{1.0f, 1.0f, 1.0f}, dir{0.0f, 0.0f, 0.0f};
    const f3 color = material_color(S, tabs, M, rec);
    if (scatter(M, color, p.d, rec, rng, att, dir)) {
        if (bounce + 1u >= rc.max_depth) {  // main.rs:45-47: the next ray_color call returns zero
            end_path(rc, radiance, p.slot, mk3(0.0f, 0.0f, 0.0f));
            return false;
        }
        p.T = p.T * att;
        p.o = rec.point;
        p.d = dir;
        p.bounce = bounce + 1u;
        return true;
    }
    f3 E{0.0f, 0.0f, 0.0f};  // main.rs:57-59
    if (M.kind == MK_LIGHT) E = color;
    end_path(rc, radiance, p.slot, p.T * E);
    return false;
}

// Block-wide compaction of the survivors into queue `shard` of `out`: wave ballot + prefix, one atomic per
// block.  Must be reached by every thread of the block; `parity` alternates between consecutive calls.
struct CompactLds {
    uint32_t wave_tot[2][rtk::BLOCK / 64];
    uint32_t block_base[2];
    uint32_t wave_tot_b[2][rtk::BLOCK / 64];  // compact_store2: the second destination
    uint32_t block_base_b[2];
};
DEV void compact_store(CompactLds& lds, uint32_t parity, bool alive, const PathRegs& p, const rtk::PathState& out,
                       uint32_t* out_count, uint32_t shard_base) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned long long mask = __ballot(alive);
    uint32_t prefix = lane_prefix(mask);
    if (lane == 0) lds.wave_tot[parity][wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < rtk::BLOCK / 64; w++) tot += lds.wave_tot[parity][w];
        lds.block_base[parity] = tot ? atomicAdd(out_count, tot) : 0u;
    }
    __syncthreads();
    if (alive) {
        uint32_t off = lds.block_base[parity] + prefix;
        for (uint32_t w = 0; w < wave; w++) off += lds.wave_tot[parity][w];
        store_path(out, shard_base + off, p);
    }
}

// compact_store with two destinations: survivors of class A are appended to the shard's queue from its start (out_a
// counts them), those of class B from its far end downwards (out_b): a path is in one of the two, so they never meet.
DEV void compact_store2(CompactLds& lds, uint32_t parity, bool a, bool b, const PathRegs& p, const rtk::PathState& out,
                        uint32_t* out_a, uint32_t* out_b, uint32_t shard_base, uint32_t cap_shard) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long mask_a = __ballot(a), mask_b = __ballot(b);
    if (lane == 0) {
        lds.wave_tot[parity][wave] = (uint32_t)__popcll(mask_a);
        lds.wave_tot_b[parity][wave] = (uint32_t)__popcll(mask_b);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot_a = 0, tot_b = 0;
        for (int w = 0; w < rtk::BLOCK / 64; w++) {
            tot_a += lds.wave_tot[parity][w];
            tot_b += lds.wave_tot_b[parity][w];
        }
        lds.block_base[parity] = tot_a ? atomicAdd(out_a, tot_a) : 0u;
        lds.block_base_b[parity] = tot_b ? atomicAdd(out_b, tot_b) : 0u;
    }
    __syncthreads();
    if (a) {
        uint32_t off = lds.block_base[parity] + lane_prefix(mask_a);
        for (uint32_t w = 0; w < wave; w++) off += lds.wave_tot[parity][w];
        store_path(out, shard_base + off, p);
    } else if (b) {
        uint32_t off = lds.block_base_b[parity] + lane_prefix(mask_b);
        for (uint32_t w = 0; w < wave; w++) off += lds.wave_tot_b[parity][w];
        store_path(out, shard_base + cap_shard - 1u - off, p);
        out.settled[shard_base + cap_shard - 1u - off] = p.settled;
    }
}

// Fills the block's dynamic LDS behind the traversal stacks with the tables of LdsTables (the launchers size the
// allocation: table_bytes).  Must be reached by every thread of the block.
DEV void copy_words(uint32_t* dst, const void* src, uint32_t words) {
    const uint32_t* s = reinterpret_cast<const uint32_t*>(src);
    for (uint32_t i = threadIdx.x; i < words; i += rtk::BLOCK) dst[i] = s[i];
}
DEV LdsTables stage_tables(const Scene& S, uint32_t* dyn_lds) {
    uint32_t* at = dyn_lds + S.stack_cap * rtk::BLOCK;
    LdsTables L;
    L.perlin0 = (LdsPerlinTable*)at;
    L.has_perlin0 = S.n_perlin != 0u;
    if (S.n_perlin) {
        copy_words(at, S.perlin, sizeof(PerlinTable) / 4u);
        at += sizeof(PerlinTable) / 4u;
    }
    L.has_scene = S.lds_scene != 0u;
    L.prims = (LdsFloat4*)at;
    L.mats = (LdsFloat4*)(at + 8u * S.n_prims);
    L.texs = (LdsFloat4*)(at + 8u * (S.n_prims + S.n_mats));
    if (S.lds_scene) {
        copy_words(at, S.prims, 8u * S.n_prims);
        copy_words(at + 8u * S.n_prims, S.mats, 8u * S.n_mats);
        copy_words(at + 8u * (S.n_prims + S.n_mats), S.texs, 8u * S.n_texs);
    }
    if (S.n_perlin || S.lds_scene) __syncthreads();
    return L;
}

// slots of a batch of n paths that fall into shard s (tiles of BLOCK slots are dealt round-robin)
DEV uint32_t primary_count(uint32_t n, uint32_t s) {
    uint32_t tiles = (n + rtk::BLOCK - 1) / rtk::BLOCK;
    uint32_t nt = tiles / rtk::NSHARD + (s < tiles % rtk::NSHARD ? 1u : 0u);
    uint32_t cnt = nt * rtk::BLOCK;
    uint32_t rem = n % rtk::BLOCK;
    if (rem && nt && (tiles - 1) % rtk::NSHARD == s) cnt -= rtk::BLOCK - rem;
    return cnt;
}

// ---- media that are plain spheres of plain colour, from LDS (Scene::settle_fast; see k_settled) ------------------------
struct FastMedium {
    float cx, cy, cz, r;
    double neg_inv_density;
    uint32_t salt, hit, key, pad;
    float col[3];
    float inv_density_lo;  // a float just below 1 / density (fast_medium_misses)
};
struct FastLds {
    FastMedium m[4];
    float blo[3], bhi[3];        // widened box of the first medium's boundary (Group::blo / bhi)
    float ncx, ncy, ncz, nr;     // the one primitive near the first medium (a resting sphere of glass), if the scene has one
    float near_ior;
    uint32_t same_sphere;        // that sphere equals the first medium's boundary bit for bit (final_scene: main.rs:279-282)
};
// The free flight of such a medium once the roots of its sphere are known (test_medium's single-sphere case).
DEV bool fast_medium_flight(const FastMedium& M, bool line_hits, float c0, float c1, float a, float tmin, uint32_t k0, uint32_t k1, uint32_t bounce,
                            float best_t, float& t) {
    float r1, r2;
    const uint32_t have = line_hits ? 3u : 0u;
    if (!pick_root(have, c0, c1, -INF_F, INF_F, r1)) return false;
    if (!pick_root(have, c0, c1, r1 + 0.0001f, INF_F, r2)) return false;
    return medium_flight(r1, r2, tmin, best_t, (double)__builtin_sqrtf(a), M.neg_inv_density, medium_xi_salted(k0, k1, bounce, M.salt), t);
}
// ConstantMedium::intersect for such a medium: test_medium's single-sphere case with the record in LDS.
DEV bool fast_medium(const FastMedium& M, f3 o, f3 d, float a, float tmin, uint32_t k0, uint32_t k1, uint32_t bounce, float best_t, float& t) {
    float c0, c1;
    const bool line_hits = sphere_roots(mk3(M.cx, M.cy, M.cz), M.r, o, d, a, c0, c1);
    return fast_medium_flight(M, line_hits, c0, c1, a, tmin, k0, k1, bounce, best_t, t);
}
// true: fast_medium certainly returns false for a closest hit at best_t (> 0), whatever the sphere -- decided from the
// medium's draw alone, in f32.  The medium takes a ray only if its flight -ln(xi) / density ends within the part of the ray
// inside it, which is shorter than best_t / |d| (medium_flight: r1 >= tmin > 0, r2 <= best_t).  The hardware log2 of the
// rounded draw is within 2e-7 relative and 6e-8 absolute of ln(xi) / ln 2; both sides carry that slack, so "misses" is never
// said wrongly, and a draw of exactly 0 (flight +inf) misses.  For a thin medium this spares nearly every ray the f64
// logarithm and the two f64 divisions.
DEV bool fast_medium_misses(const FastMedium& M, float a, uint32_t k0, uint32_t k1, uint32_t bounce, float best_t) {
    const float xi = (float)medium_xi_salted(k0, k1, bounce, M.salt);
    const float flight_lo = M.inv_density_lo * (__builtin_amdgcn_logf(xi) * -0.69314647f - 2e-7f);  // (ln 2 rounded down a little)
    const float inside_hi = best_t * __builtin_amdgcn_rsqf(a) * 1.000002f;
    return flight_lo > inside_hi;
}
// Fills FastLds from the scene (Scene::settle_fast holds); every thread of the block must reach it.
DEV void fill_fast_lds(const Scene& S, FastLds& fl) {
    if (threadIdx.x < S.n_media) {
        const uint32_t g = S.media[threadIdx.x];
        const Group& G = S.groups[g];
        const float4 sph = *reinterpret_cast<const float4*>(S.prims + (G.root_ref & 0x03FFFFFFu));
        const Material& M = S.mats[G.mat];
        FastMedium& F = fl.m[threadIdx.x];
        F.cx = sph.x;
        F.cy = sph.y;
        F.cz = sph.z;
        F.r = sph.w;
        F.neg_inv_density = G.neg_inv_density;
        F.salt = medium_salt(G.order);
        F.hit = HIT_MEDIUM_BIT | g;
        F.key = MEDIUM_KEY_BASE + G.order;
        F.col[0] = M.albedo[0];
        F.col[1] = M.albedo[1];
        F.col[2] = M.albedo[2];
        F.inv_density_lo = (float)(-G.neg_inv_density) * 0.9999995f;
        if (threadIdx.x == 0u)
            for (int k = 0; k < 3; k++) {
                fl.blo[k] = G.blo[k];
                fl.bhi[k] = G.bhi[k];
            }
    }
    if (threadIdx.x == 64u) {
        float4 sph = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        float ior = 1.0f;
        uint32_t same = 0u;
        if (S.near0_n == 1u) {
            const float4* rec = reinterpret_cast<const float4*>(S.prims + S.media[S.n_media]);
            sph = rec[0];
            ior = S.mats[__float_as_uint(rec[1].y) >> 8].ior;
            const float4 b = *reinterpret_cast<const float4*>(S.prims + (S.groups[S.media[0]].root_ref & 0x03FFFFFFu));
            same = (__float_as_uint(b.x) == __float_as_uint(sph.x) && __float_as_uint(b.y) == __float_as_uint(sph.y) &&
                    __float_as_uint(b.z) == __float_as_uint(sph.z) && __float_as_uint(b.w) == __float_as_uint(sph.w)) ? 1u : 0u;
        }
        fl.ncx = sph.x;
        fl.ncy = sph.y;
        fl.ncz = sph.z;
        fl.nr = sph.w;
        fl.near_ior = ior;
        fl.same_sphere = same;
    }
    __syncthreads();
}
// settle_next with the records in LDS: the same tests in the same order with the same arithmetic.
DEV float settle_next_fast(const Scene& S, const FastLds& fl, const PathRegs& p, uint32_t near_prim) {
    const f3 c0 = mk3(fl.m[0].cx, fl.m[0].cy, fl.m[0].cz);
    const float rr = fl.m[0].r * 1.0005f;
    const f3 oc = p.o - c0;
    const bool inside = p.o.x >= fl.blo[0] && p.o.x <= fl.bhi[0] && p.o.y >= fl.blo[1] && p.o.y <= fl.bhi[1] && p.o.z >= fl.blo[2] &&
                        p.o.z <= fl.bhi[2] && dot3(oc, oc) <= rr * rr;
    if (!inside) return 0.0f;
    const float a2 = dot3(p.d, p.d);
    float q0, q1, tg = INF_F, tm;
    const bool line_hits = sphere_roots(c0, fl.m[0].r, p.o, p.d, a2, q0, q1);
    bool hit_g = false;
    if (S.near0_n == 1u) {
        // (the same sphere: its roots are the boundary's -- quotient_pair gives the quotients of sphere_candidate bit for
        // bit -- and the root it takes is pick_root's)
        if (fl.same_sphere) hit_g = pick_root(line_hits ? 3u : 0u, q0, q1, 0.005f, INF_F, tg);
        else hit_g = sphere_candidate(mk3(fl.ncx, fl.ncy, fl.ncz), fl.nr, p.o, p.d, a2, 0.005f, INF_F, tg);
    }
    if (fast_medium_flight(fl.m[0], line_hits, q0, q1, a2, 0.005f, p.k0, p.k1, p.bounce, hit_g ? tg : INF_F, tm)) return tm > 0.0f ? tm : 0.0f;
    if (hit_g && tg > 0.0f && near_prim != 0x7FFFFFFFu) {
        const f3 pc = (p.o + tg * p.d) - c0;
        if (dot3(pc, pc) <= rr * rr) return -tg;
    }
    return 0.0f;
}


// =================================================================================
// Kernels.  Grid = NSHARD * blocks_per_shard; block b serves shard b % NSHARD.
// =================================================================================

// Held to 128 VGPRs (4 waves per SIMD): one more resident wave than the compiler's own choice hides more of
// the queue loads and the compaction barrier than the few spilled values cost.
// FAST (Scene::settle_fast): the media are plain spheres of plain colour, tested from the records in LDS.
template <int GROUPS, bool COUNT, bool PRIMARY, bool FLAT, bool FAST = false>
__global__ __launch_bounds__(rtk::BLOCK) __attribute__((amdgpu_waves_per_eu(GROUPS ? 4 : BOUNCE_WAVES))) void k_bounce(Scene S, rtk::RenderConst rc, rtk::Batch batch, rtk::PathState in,
                                                        rtk::PathState out, rtk::Queues q, uint32_t bounce,
                                                        float4* __restrict__ radiance, unsigned long long* stats,
                                                        uint32_t* overflow_flag) {
    extern __shared__ __attribute__((aligned(16))) uint32_t stack[];  // S.stack_cap entries per lane, [entry][lane]
    __shared__ CompactLds clds;
    const uint32_t shard = blockIdx.x % rtk::NSHARD, j0 = blockIdx.x / rtk::NSHARD, stride = gridDim.x / rtk::NSHARD;
    const uint32_t n_batch = batch.npix * batch.n_samples;
    const uint32_t n = PRIMARY ? primary_count(n_batch, shard) : q.counts[(bounce % 3u) * rtk::NSHARD + shard];
    uint32_t* out_count = q.counts + ((bounce + 1u) % 3u) * rtk::NSHARD + shard;
    const uint32_t shard_base = shard * q.cap_shard;
    if (j0 == 0 && threadIdx.x == 0) {
        q.counts[((bounce + 2u) % 3u) * rtk::NSHARD + shard] = 0;  // the next bounce appends there
        if (n) atomicAdd(stats, (unsigned long long)n);
    }
    const Stack stk{stack + threadIdx.x, S.stack_cap, nullptr, 0u};
    const LdsTables L = stage_tables(S, stack);
    __shared__ FastLds fl;
    if (FAST) fill_fast_lds(S, fl);
    TraceCount tc{0u, 0u, 0u, 0u, 0u};
    uint32_t parity = 0;
    // instrumented build only: wave cycles spent per section (s_memtime ticks), and wave tiles
    unsigned long long cyc[4] = {0, 0, 0, 0}, tiles = 0, t0 = 0, t1 = 0;
    for (uint32_t base = j0 * rtk::BLOCK; base < n; base += stride * rtk::BLOCK, parity ^= 1u) {
        const uint32_t i = base + threadIdx.x;
        bool alive = false;
        PathRegs p;
        if (COUNT) t0 = __builtin_amdgcn_s_memtime();
        if (i < n) {
            if (PRIMARY)
                camera_path(rc, batch, ((base / rtk::BLOCK) * rtk::NSHARD + shard) * rtk::BLOCK + threadIdx.x, p);
            else
                load_path(in, shard_base + i, p);
            if (COUNT) {
                // the section ends when the loaded values are there
                asm volatile("" ::"v"(p.o.x), "v"(p.d.x), "v"(p.T.x), "v"(p.k0));
                t1 = __builtin_amdgcn_s_memtime();
                cyc[0] += t1 - t0;
                t0 = t1;
            }
            Hit h;
            trace_ray<rtk::BLOCK, GROUPS, FLAT, COUNT, !FAST>(S, p.o, p.d, p.time, 0.005f, INF_F, p.k0, p.k1, bounce, h, stk,
                                                               overflow_flag, &tc);
            if (FAST && ray_is_finite(p.o, p.d)) {  // apply_media from the records in LDS
                const float a = dot3(p.d, p.d);
                for (uint32_t m = 0; m < S.n_media; m++) {
                    float tm;
                    if (fast_medium(fl.m[m], p.o, p.d, a, 0.005f, p.k0, p.k1, bounce, h.t, tm)) {
                        h.t = tm;
                        h.prim = fl.m[m].hit;
                        h.grp = -1;
                        h.key = fl.m[m].key;
                    }
                }
            }
            if (COUNT) {
                asm volatile("" ::"v"(h.t), "v"(h.prim));
                t1 = __builtin_amdgcn_s_memtime();
                cyc[1] += t1 - t0;
                t0 = t1;
            }
            alive = shade_path<GROUPS == 2>(S, L, rc, p, h, bounce, radiance);
            if (COUNT) {
                asm volatile("" ::"v"(p.o.x), "v"(p.d.x), "v"(p.T.x));
                t1 = __builtin_amdgcn_s_memtime();
                cyc[2] += t1 - t0;
                t0 = t1;
            }
        }
        compact_store(clds, parity, alive, p, out, out_count, shard_base);
        if (COUNT) {
            t1 = __builtin_amdgcn_s_memtime();
            cyc[3] += t1 - t0;
            tiles++;
        }
    }
    if (COUNT) {
        if (tc.nodes) atomicAdd(stats + 1, (unsigned long long)tc.nodes);
        if (tc.prims) atomicAdd(stats + 2, (unsigned long long)tc.prims);
        if ((threadIdx.x & 63u) == 0 && tiles) {
            for (int k = 0; k < 4; k++) atomicAdd(stats + 3 + k, cyc[k]);
            atomicAdd(stats + 7, tiles);
        }
    }
}

__global__ __launch_bounds__(rtk::BLOCK) void k_raygen(rtk::RenderConst rc, rtk::Batch batch, rtk::PathState out, rtk::Queues q) {
    const uint32_t shard = blockIdx.x % rtk::NSHARD, j0 = blockIdx.x / rtk::NSHARD, stride = gridDim.x / rtk::NSHARD;
    const uint32_t n = primary_count(batch.npix * batch.n_samples, shard);
    if (j0 == 0 && threadIdx.x == 0) {
        q.counts[shard] = n;
        q.counts[rtk::NSHARD + shard] = 0;
        q.counts[2 * rtk::NSHARD + shard] = 0;
    }
    for (uint32_t base = j0 * rtk::BLOCK; base < n; base += stride * rtk::BLOCK) {
        const uint32_t i = base + threadIdx.x;
        if (i < n) {
            PathRegs p;
            camera_path(rc, batch, ((base / rtk::BLOCK) * rtk::NSHARD + shard) * rtk::BLOCK + threadIdx.x, p);
            store_path(out, shard * q.cap_shard + i, p);
        }
    }
}

#ifndef TRACE_WAVES
#define TRACE_WAVES 7
#endif
#ifndef CROWN_BLOCK
#define CROWN_BLOCK 1024  // threads per block of the trace kernel that keeps the crown in LDS
#endif
// WIDTH: children per inner node the walk is compiled for (2 or 4; a launch picks the one the scene holds)
// TB: threads per block.  CROWN: the block keeps the scene's crown (Scene::crown, rtd::CROWN_BASE) at the start of its LDS and
// is launched with S.root_ref = S.crown_root; blocks are large then (one copy of the crown serves 16 waves).
template <int GROUPS, bool COUNT, bool FLAT, int WIDTH, int TB = rtk::BLOCK, bool CROWN = false>
__global__ __launch_bounds__(TB) __attribute__((amdgpu_waves_per_eu(CROWN ? 8 : TRACE_WAVES))) void k_trace(Scene S, rtk::PathState in, rtk::Queues q, uint32_t bounce,
                                                       float4* __restrict__ hits, unsigned long long* stats,
                                                       uint32_t* overflow_flag, uint32_t REFILL, int INNER_STEPS, uint32_t lds_entries,
                                                       uint32_t* spill) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_all[];
    // the crown (instance table, then the records), then lds_entries stack entries per lane, [entry][lane]; deeper ones spill
    const uint32_t crown_words_n = CROWN ? CROWN_INST_SLOTS + S.crown_n * (WIDTH == 4 ? 32u : 16u) : 0u;
    uint32_t* stack = lds_all + crown_words_n;
    if (CROWN) {
        const uint4* src = reinterpret_cast<const uint4*>(S.crown);
        uint4* dst = reinterpret_cast<uint4*>(lds_all);
        for (uint32_t i = threadIdx.x; i < crown_words_n / 4u; i += TB) dst[i] = src[i];
        __syncthreads();
    }
    const uint32_t shard = blockIdx.x % rtk::NSHARD, j0 = blockIdx.x / rtk::NSHARD, stride = gridDim.x / rtk::NSHARD;
    const uint32_t n = q.counts[(bounce % 3u) * rtk::NSHARD + shard];
    const uint32_t shard_base = shard * q.cap_shard;
    if (j0 == 0 && threadIdx.x == 0) {
        q.counts[((bounce + 2u) % 3u) * rtk::NSHARD + shard] = 0;
        if (n) atomicAdd(stats, (unsigned long long)n);
    }
    // blockIdx.x * BLOCK + threadIdx.x is this thread's column of the spill area (gridDim.x * BLOCK columns)
    const Stack stk{stack + threadIdx.x, lds_entries, spill + (size_t)blockIdx.x * TB + threadIdx.x, gridDim.x * (uint32_t)TB};
    TraceCount tc{0u, 0u, 0u, 0u, 0u};
    if (FLAT) {  // no tree, no divergence between rays: plain strided loop
        for (uint32_t i = j0 * TB + threadIdx.x; i < n; i += stride * TB) {
            const uint32_t at = shard_base + i;
            const float4 ot = in.o_time[at], ds = in.d_slot[at];
            Hit h;
            trace_flat<GROUPS != 0, COUNT>(S, mk3(ot.x, ot.y, ot.z), mk3(ds.x, ds.y, ds.z), ot.w, 0.005f, INF_F, h, &tc);
            hits[at] = make_float4(h.t, __uint_as_float(h.prim), __int_as_float(h.grp), __uint_as_float(h.key));
        }
        if (COUNT) {
            if (tc.prims) atomicAdd(stats + 2, (unsigned long long)tc.prims);
        }
        return;
    }
    // Each wave owns a contiguous run of its shard's queue and keeps its lanes busy across rays: incoherent
    // rays differ several-fold in how many nodes they visit, and a wave that waits for its longest ray runs at
    // ~27% lane use.  So a lane whose ray is finished does not wait for the tile: once REFILL lanes are free
    // (or none is busy), they take the next rays of the run.  Surfaces only: the media are applied by k_shade.
    const uint32_t nw = stride * (TB / 64u), wi = j0 * (TB / 64u) + (threadIdx.x >> 6);
    const uint32_t per = (((n + nw - 1u) / nw) + 63u) & ~63u;
    uint32_t next = __builtin_amdgcn_readfirstlane(wi * per);
    const uint32_t end = __builtin_amdgcn_readfirstlane(next + per < n ? next + per : n);
    bool have = false;
    uint32_t idx = 0;
    f3 wo{0, 0, 0}, wd{0, 0, 0};
    // With instances the world ray is needed again wherever the walk enters or leaves one.  It waits in this lane's own
    // LDS words behind the stacks ([component][lane]) rather than in six registers: those cost a resident wave per SIMD.
    float* world_ray = reinterpret_cast<float*>(stack + (size_t)lds_entries * TB) + threadIdx.x;
    float time = 0.0f;
    Walk w;
    walk_begin(S, w, wo, wd, INF_F);
    w.cur = REF_EMPTY;
    uint32_t media_prims = 0;
    if (COUNT && GROUPS)
        for (uint32_t m = 0; m < S.n_media; m++) media_prims += 2u * medium_boundary_count(S.groups[S.media[m]]);
    uint32_t rounds = 0, busy_lanes = 0;  // instrumented build: loop rounds of this wave and the lanes that held a ray in them
    for (;;) {
        const unsigned long long idle = __ballot(!have);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        if (n_idle == 64u && next >= end) break;
        if (next < end && (n_idle >= REFILL || n_idle == 64u)) {
            const uint32_t mine = next + lane_prefix(idle);
            next += n_idle;
            if (!have && mine < end) {
                idx = shard_base + mine;
                const float4 ot = in.o_time[idx];
                const float4 ds = in.d_slot[idx];
                wo = mk3(ot.x, ot.y, ot.z);
                wd = mk3(ds.x, ds.y, ds.z);
                time = ot.w;
                if (GROUPS) {
                    world_ray[0 * TB] = wo.x;
                    world_ray[1 * TB] = wo.y;
                    world_ray[2 * TB] = wo.z;
                    world_ray[3 * TB] = wd.x;
                    world_ray[4 * TB] = wd.y;
                    world_ray[5 * TB] = wd.z;
                }
                walk_begin(S, w, wo, wd, INF_F);
                have = true;
                if (COUNT) {
                    tc.prims += media_prims;
                    if (!ray_is_finite(wo, wd)) atomicAdd(stats + 15, 1ull);  // rays with a NaN or infinite component
                }
            }
        }
        if (COUNT) {
            if ((threadIdx.x & 63u) == 0u) rounds++;
            if (have) busy_lanes++;
        }
        if (have) {
            walk_inner<TB, COUNT, WIDTH, CROWN>(S, w, 0.005f, stk, overflow_flag, &tc, INNER_STEPS);
            // (without instances the walk never leaves world space: its own copy of the ray is the world ray, and keeping
            // a second one would cost the kernel a resident wave per SIMD)
            if (GROUPS) {
                const uint32_t type = w.cur >> 30;
                if (w.cur != REF_EMPTY && type != REF_INNER && type != REF_PRIMS) {  // instance or sentinel
                    const f3 lo = mk3(world_ray[0 * TB], world_ray[1 * TB], world_ray[2 * TB]);
                    const f3 ld = mk3(world_ray[3 * TB], world_ray[4 * TB], world_ray[5 * TB]);
                    walk_leaf<TB, GROUPS, COUNT, CROWN>(S, w, lo, ld, time, 0.005f, stk, overflow_flag, &tc);
                } else {
                    walk_leaf<TB, 0, COUNT>(S, w, w.o, w.d, time, 0.005f, stk, overflow_flag, &tc);
                }
            } else {
                walk_leaf<TB, GROUPS, COUNT>(S, w, w.o, w.d, time, 0.005f, stk, overflow_flag, &tc);
            }
            if (walk_done(w)) {
                hits[idx] = make_float4(w.best.t, __uint_as_float(w.best.prim), __int_as_float(w.best.grp), __uint_as_float(w.best.key));
                have = false;
            }
        }
    }
    if (COUNT) {
        if (tc.nodes) atomicAdd(stats + 1, (unsigned long long)tc.nodes);
        if (tc.prims) atomicAdd(stats + 2, (unsigned long long)tc.prims);
        // lane use of the persistent walk (words 9..14; read with RTAMD_TRACE_LANES set)
        if (tc.inner_waves) atomicAdd(stats + 9, (unsigned long long)tc.inner_waves);
        if (tc.leaf_waves) atomicAdd(stats + 10, (unsigned long long)tc.leaf_waves);
        if (tc.leaf_lanes) atomicAdd(stats + 11, (unsigned long long)tc.leaf_lanes);
        if (rounds) atomicAdd(stats + 12, (unsigned long long)rounds);
        if (busy_lanes) atomicAdd(stats + 13, (unsigned long long)busy_lanes);
        if (tc.nodes) atomicAdd(stats + 14, (unsigned long long)tc.nodes);
    }
}

// Settled ahead.  A path that has just scattered in the first medium will most likely scatter there again a few units on:
// final_scene's blue sphere (main.rs:281-284: mean free path 5 in a radius of 70, nothing absorbed) holds 5 % of the paths
// for all 50 bounces, a third of the frame's rays.  While such a ray starts inside the boundary, the only surfaces it can
// meet before it leaves are the few whose box meets the boundary (Scene::near0_n, found at commit), and the medium sees
// surfaces only through min(r2, t) with r2 on the boundary.  So the kernel that has just made the ray tests it against those
// primitives and then lets the first medium decide, exactly as the next bounce would after the tree walk: if the medium
// takes the ray, that IS the next bounce's answer for this medium, and the tree is never walked for it.
//
// settle_next: the test itself, for the ray p now holds (bounce p.bounce).  Returns 0 when nothing is settled (the origin is
// outside the boundary, or the ray leaves or meets another surface: it then goes through the tree as usual), > 0 the distance
// at which the first medium takes it, < 0 minus the distance at which it meets the one primitive near a ball-shaped boundary
// (near_prim; the glass around final_scene's medium, from inside) with the hit point still in the ball: no other primitive
// reaches in there, so that hit is the tree's answer too.
template <bool NESTED>
DEV float settle_next(const Scene& S, const PathRegs& p, uint32_t medium0, uint32_t near_prim) {
    const Group& G = S.groups[S.media[0]];
    bool inside = p.o.x >= G.blo[0] && p.o.x <= G.bhi[0] && p.o.y >= G.blo[1] && p.o.y <= G.bhi[1] && p.o.z >= G.blo[2] &&
                  p.o.z <= G.bhi[2];
    float4 sph = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (S.near0_ball) {  // (the list holds what meets the ball of 1.001 radii: the origin within 1.0005 of them)
        sph = *reinterpret_cast<const float4*>(S.prims + (G.root_ref & 0x03FFFFFFu));
        const f3 oc = p.o - mk3(sph.x, sph.y, sph.z);
        const float rr = sph.w * 1.0005f;
        inside = inside && dot3(oc, oc) <= rr * rr;
    }
    if (!inside) return 0.0f;
    Hit nh;
    nh.t = INF_F;
    nh.prim = HIT_NONE;
    nh.grp = -1;
    nh.key = 0;
    const float a = dot3(p.d, p.d);
    for (uint32_t k = 0; k < S.near0_n; k++) test_prim<false>(S, S.media[S.n_media + k], p.o, p.d, a, p.time, 0.005f, -1, nh);
    test_medium<NESTED>(S, S.media[0], p.o, p.d, p.time, 0.005f, p.k0, p.k1, p.bounce, nh);
    if (nh.prim == medium0 && nh.t > 0.0f) return nh.t;
    if (nh.prim == near_prim && nh.t > 0.0f) {
        const f3 pc = (p.o + nh.t * p.d) - mk3(sph.x, sph.y, sph.z);
        const float rr = sph.w * 1.0005f;
        if (dot3(pc, pc) <= rr * rr) return -nh.t;
    }
    return 0.0f;
}

// k_shade.  SETTLE: compiled with the test above (launched for scenes whose first medium has a known near list): a path
// that has just scattered in the first medium (or off the one primitive near it) and whose next hit is settled is queued
// at the far end of its shard with that distance (a second count per shard, rtk::Queues), where the trace kernel does not
// look; k_settled takes it from there.  Paths are tested when at least `entry_lanes` lanes of their wave qualify (the test
// costs the wave the same for one lane as for 64).  Every path carries its own bounce number (see PathRegs).
// FAST (Scene::settle_fast; implies SETTLE, never NESTED): the media are tested and the next ray settled from the records in LDS.
template <bool NESTED, bool SETTLE, bool FAST>
__global__ __launch_bounds__(rtk::BLOCK) __attribute__((amdgpu_waves_per_eu(4))) void k_shade(Scene S, rtk::RenderConst rc, rtk::PathState in, rtk::PathState out,
                                                       const float4* __restrict__ hits, rtk::Queues q, uint32_t bounce,
                                                       float4* __restrict__ radiance, unsigned long long* stats, uint32_t entry_lanes,
                                                       uint32_t near_hits) {
    extern __shared__ __attribute__((aligned(16))) uint32_t stack[];  // no traversal here: only the staged tables (launched with stack_cap = 0 in mind)
    __shared__ CompactLds clds;
    __shared__ FastLds fl;
    Scene St = S;
    St.stack_cap = 0;
    const LdsTables L = stage_tables(St, stack);
    if (FAST) fill_fast_lds(S, fl);
    const uint32_t shard = blockIdx.x % rtk::NSHARD, j0 = blockIdx.x / rtk::NSHARD, stride = gridDim.x / rtk::NSHARD;
    const uint32_t n = q.counts[(bounce % 3u) * rtk::NSHARD + shard];
    uint32_t* out_count = q.counts + ((bounce + 1u) % 3u) * rtk::NSHARD + shard;
    const uint32_t shard_base = shard * q.cap_shard;
    uint32_t parity = 0;
    if (!SETTLE) {
        for (uint32_t base = j0 * rtk::BLOCK; base < n; base += stride * rtk::BLOCK, parity ^= 1u) {
            const uint32_t i = base + threadIdx.x;
            bool alive = false;
            PathRegs p;
            if (i < n) {
                load_path(in, shard_base + i, p);
                float4 hv = hits[shard_base + i];
                Hit h;
                h.t = hv.x;
                h.prim = __float_as_uint(hv.y);
                h.grp = __float_as_int(hv.z);
                h.key = __float_as_uint(hv.w);
                if (S.n_media) apply_media<false, NESTED>(S, p.o, p.d, p.time, 0.005f, p.k0, p.k1, bounce, h, nullptr);
                alive = shade_path<NESTED>(S, L, rc, p, h, bounce, radiance);
            }
            compact_store(clds, parity, alive, p, out, out_count, shard_base);
        }
        return;
    }
    const uint32_t medium0 = HIT_MEDIUM_BIT | S.media[0];
    // the single primitive near a ball-shaped boundary, whose hits from inside can be settled as well (else: no such index)
    const uint32_t near_prim = (S.near0_ball && S.near0_n == 1u && near_hits) ? S.media[S.n_media] : 0x7FFFFFFFu;
    uint32_t* out_settled = q.counts + (3u + (bounce + 1u) % 3u) * rtk::NSHARD + shard;
    if (j0 == 0 && threadIdx.x == 0) q.counts[(3u + (bounce + 2u) % 3u) * rtk::NSHARD + shard] = 0;  // the next bounce appends there
    for (uint32_t base = j0 * rtk::BLOCK; base < n; base += stride * rtk::BLOCK, parity ^= 1u) {
        const uint32_t i = base + threadIdx.x;
        bool alive = false, candidate = false;
        PathRegs p;
        if (i < n) {
            load_path(in, shard_base + i, p);
            float4 hv = hits[shard_base + i];
            Hit h;
            h.t = hv.x;
            h.prim = __float_as_uint(hv.y);
            h.grp = __float_as_int(hv.z);
            h.key = __float_as_uint(hv.w);
            if (ray_is_finite(p.o, p.d)) {  // (as apply_media)
                if (FAST) {
                    const float a = dot3(p.d, p.d);
                    for (uint32_t m = 0; m < S.n_media; m++) {
                        float tm;
                        if (fast_medium(fl.m[m], p.o, p.d, a, 0.005f, p.k0, p.k1, p.bounce, h.t, tm)) {
                            h.t = tm;
                            h.prim = fl.m[m].hit;
                            h.grp = -1;
                            h.key = fl.m[m].key;
                        }
                    }
                } else {
                    for (uint32_t m = 0; m < S.n_media; m++)
                        test_medium<NESTED>(S, S.media[m], p.o, p.d, p.time, 0.005f, p.k0, p.k1, p.bounce, h);
                }
            }
#ifdef RT_HIST
            if (p.bounce >= 16u) {
                uint32_t cls = 4u;
                if (h.prim == HIT_NONE) cls = 0u;
                else if (h.prim == medium0) cls = 1u;
                else if (h.prim & HIT_MEDIUM_BIT) cls = 2u;
                else { for (uint32_t k = 0; k < S.near0_n; k++) if (h.prim == S.media[S.n_media + k]) cls = 3u; }
                if (cls == 4u) { float4 A, B; load_record(L, L.prims, S.prims, h.prim, A, B); const Material M = load_material(S, L, __float_as_uint(B.y) >> 8); cls = 4u + M.kind; if (h.grp >= 0) cls = 10u; }
                atomicAdd(stats + 17 + cls, 1ull);
            }
#endif
            alive = shade_path<NESTED>(S, L, rc, p, h, p.bounce, radiance);
            candidate = alive && (h.prim == medium0 || h.prim == near_prim);
        }
        if ((uint32_t)__popcll(__ballot(candidate)) >= entry_lanes && candidate)
            p.settled = FAST ? settle_next_fast(S, fl, p, near_prim) : settle_next<NESTED>(S, p, medium0, near_prim);
        compact_store2(clds, parity, alive && p.settled == 0.0f, alive && p.settled != 0.0f, p, out, out_count, out_settled, shard_base,
                       q.cap_shard);
    }
}

// The settled paths, followed in registers.  Launch `bounce` takes the paths that launch `bounce - 1` of k_shade (or of this
// kernel's predecessor) left at the far end of `in`.  Each wave owns a run of them and keeps its lanes busy across paths, as
// k_trace does: a lane makes the hit from the stored distance, lets the other media have their say, shades (one step of
// ray_color, main.rs:44-63), and tests the ray it has just made with settle_next; while that settles the next hit too
// (about nineteen times in twenty inside final_scene's blue sphere) the path stays in its lane and its state never touches
// memory.  A path that leaves (its ray goes out of the boundary or meets another surface) waits in its lane until the wave
// next refills, when all such paths of the wave are appended to the FRONT of `out` with one atomic: the next launch of
// k_trace walks the tree for them, at whatever bounce each has reached.  Draws are keyed by (stream, bounce), so the image
// is the one the bounce-per-launch kernels give, bit for bit.  Counted in stats[0] and stats[16]: one ray per step.
//
// FAST (Scene::settle_fast, decided at commit): every medium of the scene is a resting sphere in world space filled with a
// plain colour, and at most one primitive, a resting sphere too, lies near the first medium (final_scene: the blue ball
// inside its glass shell and the fog around everything, main.rs:279-284).  Then a step in the first medium needs no record
// from memory at all: the few numbers that describe the media sit in LDS (FastLds, filled once per block), the arithmetic
// is the same device functions in the same order as the general code (sphere_roots / pick_root / medium_flight for a
// medium, sphere_candidate for the primitive, the isotropic scatter of shade_path), so the bits are the same.  What such
// a step cannot do -- shade a hit on the primitive near the medium (glass: the whole material code) -- is left to the general
// step, and lanes that need that one WAIT until `slow_min` of their wave do (or nothing else is left to do): run for two or
// three lanes in every round it would cost each round more than the fast step itself.
#ifndef SETTLED_WAVES
#define SETTLED_WAVES 4
#endif
template <bool NESTED, bool FAST>
__global__ __launch_bounds__(rtk::BLOCK) __attribute__((amdgpu_waves_per_eu(SETTLED_WAVES))) void k_settled(
    Scene S, rtk::RenderConst rc, rtk::PathState in, rtk::PathState out, rtk::Queues q, uint32_t bounce, float4* __restrict__ radiance,
    unsigned long long* stats, uint32_t REFILL, uint32_t min_run, uint32_t DRAIN, uint32_t near_hits, uint32_t slow_min, uint32_t late_max) {
    extern __shared__ __attribute__((aligned(16))) uint32_t stack[];  // only the staged tables
    __shared__ FastLds fl;
    const uint32_t shard = blockIdx.x % rtk::NSHARD, j0 = blockIdx.x / rtk::NSHARD, stride = gridDim.x / rtk::NSHARD;
    const uint32_t n = q.counts[(3u + bounce % 3u) * rtk::NSHARD + shard];
    // runs of at least min_run paths: a wave refills from its own run only, and a short run ends in a long, thin drain
    const uint32_t nw = stride * (rtk::BLOCK / 64u), wi = j0 * (rtk::BLOCK / 64u) + (threadIdx.x >> 6);
    uint32_t per = (((n + nw - 1u) / nw) + 63u) & ~63u;
    if (per < min_run) per = min_run;
    // (the grid is sized for a full queue; most launches late in a batch find a few paths per shard, and a block none of
    // whose waves gets a run leaves before it stages anything)
    if ((unsigned long long)j0 * (rtk::BLOCK / 64u) * per >= n) return;
    Scene St = S;
    St.stack_cap = 0;
    const LdsTables L = stage_tables(St, stack);
    uint32_t* out_count = q.counts + ((bounce + 1u) % 3u) * rtk::NSHARD + shard;
    uint32_t* out_settled = q.counts + (3u + (bounce + 1u) % 3u) * rtk::NSHARD + shard;
    const uint32_t shard_base = shard * q.cap_shard, lane = threadIdx.x & 63u;
    const uint32_t medium0 = HIT_MEDIUM_BIT | S.media[0];
    const uint32_t near_prim = (S.near0_ball && S.near0_n == 1u && near_hits) ? S.media[S.n_media] : 0x7FFFFFFFu;
    if (FAST) fill_fast_lds(S, fl);
    const unsigned long long first = (unsigned long long)wi * per;
    uint32_t next = __builtin_amdgcn_readfirstlane(first < n ? (uint32_t)first : n);
    const uint32_t end = __builtin_amdgcn_readfirstlane(n - next > per ? next + per : n);
    // a lane: have = it holds a path whose next hit is settled (p.settled != 0); leaving = it holds one that has to go back
    // to the tree (waits for the wave's next refill)
    bool have = false, leaving = false;
    PathRegs p;
    p.o = p.d = p.T = mk3(0.0f, 0.0f, 0.0f);
    p.time = 0.0f;
    p.k0 = p.k1 = p.slot = 0u;
    uint32_t rays = 0, taken = 0, late_rounds = 0;
    const bool last_launch = bounce + 1u >= rc.max_depth;
    const uint32_t drain = last_launch ? 1u : DRAIN;
#ifdef RT_SECTIONS  // wave cycles per section of the loop (s_memtime), summed into stats[24..28]; [29] counts loop rounds
    unsigned long long cyc[5] = {0, 0, 0, 0, 0}, rounds = 0, t0 = 0, t1 = 0;
#define RT_SECTION(k, ...) do { asm volatile("" :: __VA_ARGS__); t1 = __builtin_amdgcn_s_memtime(); cyc[k] += t1 - t0; t0 = t1; } while (0)
#else
#define RT_SECTION(k, ...) do { } while (0)
#endif
    if (next < end) {
        for (;;) {
#ifdef RT_SECTIONS
            t0 = __builtin_amdgcn_s_memtime();
            rounds++;
#endif
            const unsigned long long busy = __ballot(have);
            const uint32_t n_idle = 64u - (uint32_t)__popcll(busy);
            const bool more = next < end;
            // Once its run is used up a wave does not drain to the last path (a few lanes for up to max_depth rounds): with
            // fewer than DRAIN paths left it hands them on, still settled, to the far end of `out`, where the next launch of
            // this kernel finds them next to the paths k_shade has settled meanwhile.  (Not in the last launch of a batch, which
            // nothing follows: there every path is at the depth limit and ends with its next step.)
            // A wave also hands on, however many paths it holds, `late_max` rounds after its run was used up: a small launch
            // (a few paths per shard, late in a batch or on one rank's share of a frame) is a chain of serial rounds on a
            // nearly empty GPU, and the next launch comes anyway.
            if (!more) late_rounds++;
            if ((!more && (n_idle > 64u - drain || (late_rounds > late_max && !last_launch))) || (more && n_idle >= REFILL)) {
                const unsigned long long out_mask = __ballot(leaving);
                if (out_mask) {
                    uint32_t at = 0;
                    if (lane == 0u) at = atomicAdd(out_count, (uint32_t)__popcll(out_mask));
                    at = __builtin_amdgcn_readfirstlane(at);
                    if (leaving) {
                        p.settled = 0.0f;
                        store_path(out, shard_base + at + lane_prefix(out_mask), p);
                        leaving = false;
                    }
                }
                if (!more) {
                    if (busy) {
                        uint32_t at = 0;
                        if (lane == 0u) at = atomicAdd(out_settled, 64u - n_idle);
                        at = __builtin_amdgcn_readfirstlane(at);
                        if (have) {
                            const uint32_t to = shard_base + q.cap_shard - 1u - (at + lane_prefix(busy));
                            store_path(out, to, p);
                            out.settled[to] = p.settled;
                        }
                    }
                    break;
                }
                const uint32_t mine = next + lane_prefix(~busy);
                next += n_idle;
                if (!have && mine < end) {
                    const uint32_t at = shard_base + q.cap_shard - 1u - mine;
                    load_path(in, at, p);
                    p.settled = in.settled[at];
                    have = true;
                    taken++;
                }
            }
            RT_SECTION(0, "v"(p.o.x), "v"(p.d.x), "v"(p.T.x), "v"(p.k0), "v"(p.settled));
            // FAST: every path is stepped here; the general step below runs only in kernels compiled without it
            bool general = have && !FAST;
            if (FAST) {
                // the hit at the stored distance -- in the first medium, or on the glass near it -- unless a later medium takes the
                // ray first.  Nearly always none does and the draw alone says so (fast_medium_misses); the few lanes that need the
                // real test wait until `slow_min` of their wave do, or nothing else is left to do: run for one or two lanes in
                // every round, the f64 arithmetic would cost the round a quarter of its time.
                float t = __builtin_fabsf(p.settled);
                uint32_t which = p.settled > 0.0f ? 0u : 4u;  // 0..3: that medium; 4: the sphere near the first medium
                const float a = dot3(p.d, p.d);
                bool clear = have;
                for (uint32_t m = 1u; m < S.n_media; m++) clear = clear && fast_medium_misses(fl.m[m], a, p.k0, p.k1, p.bounce, t);
                const unsigned long long wait_mask = __ballot(have && !clear);
                if (wait_mask && ((uint32_t)__popcll(wait_mask) >= slow_min || __ballot(clear) == 0ull || next >= end)) {
                    // (never left waiting once the wave's run is used up: a path must not be handed on to the next launch
                    // without having made a step in this one -- every path of launch L is at bounce L at least, and the last
                    // launch ends them all)
                    if (have && !clear) {
                        for (uint32_t m = 1u; m < S.n_media; m++) {
                            float tm;
                            if (fast_medium(fl.m[m], p.o, p.d, a, 0.005f, p.k0, p.k1, p.bounce, t, tm)) {
                                t = tm;
                                which = m;
                            }
                        }
                        clear = true;
                    }
                }
                RT_SECTION(1, "v"(t), "v"(which));
                if (clear) {
                    rays++;
                    const f3 point = p.o + (t * p.d);
                    Rng rng{p.k0, p.k1, p.bounce << 24};
                    f3 att{1.0f, 1.0f, 1.0f}, dir;
                    if (which < 4u) {  // shade_path for a medium with a plain colour (shapes.rs:690-697, materials.rs:204-215)
                        dir = normalize3(rand_in_unit_sphere(rng));
                        att = mk3(fl.m[which].col[0], fl.m[which].col[1], fl.m[which].col[2]);
                    } else {  // ... and for a sphere of glass (make_record: shapes.rs:141-153, hit_record.rs:32-42; materials.rs:117-148)
                        Record rec;
                        const f3 outward = (point - mk3(fl.ncx, fl.ncy, fl.ncz)) / fl.nr;
                        rec.front = dot3(p.d, outward) < 0.0f;
                        rec.normal = rec.front ? outward : -outward;
                        rec.point = point;
                        rec.u = rec.v = 0.0f;
                        Material M;
                        M.kind = MK_DIELECTRIC;
                        M.ior = fl.near_ior;
                        M.fuzz = 0.0f;
                        M.tex = M.pad = 0u;
                        M.albedo[0] = M.albedo[1] = M.albedo[2] = 0.0f;
                        (void)scatter(M, att, p.d, rec, rng, att, dir);  // (always goes on, attenuation 1)
                    }
                    if (p.bounce + 1u >= rc.max_depth) {  // main.rs:45-47: the next ray_color call returns zero
                        end_path(rc, radiance, p.slot, mk3(0.0f, 0.0f, 0.0f));
                        have = false;
                    } else {
                        p.T = p.T * att;
                        p.o = point;
                        p.d = dir;
                        p.bounce++;
                    }
                    RT_SECTION(2, "v"(p.o.x), "v"(p.d.x), "v"(p.T.x));
                    if (have) {
                        p.settled = settle_next_fast(S, fl, p, near_prim);
                        have = p.settled != 0.0f;
                        leaving = !have;
                    }
                    RT_SECTION(3, "v"(p.settled));
                }
            }
            if (general) {
                Hit h;  // (> 0: the first medium at that distance; < 0: the one primitive near it)
                h.t = __builtin_fabsf(p.settled);
                h.prim = p.settled > 0.0f ? medium0 : near_prim;
                h.grp = -1;
                h.key = 0;
                for (uint32_t m = 1u; m < S.n_media; m++)
                    test_medium<NESTED>(S, S.media[m], p.o, p.d, p.time, 0.005f, p.k0, p.k1, p.bounce, h);
                RT_SECTION(1, "v"(h.t), "v"(h.prim));
                rays++;
                have = false;
                const bool alive = shade_path<NESTED>(S, L, rc, p, h, p.bounce, radiance);
                RT_SECTION(2, "v"(p.o.x), "v"(p.d.x), "v"(p.T.x));
                if (alive) {
                    p.settled = settle_next<NESTED>(S, p, medium0, near_prim);
                    have = p.settled != 0.0f;
                    leaving = !have;
                }
                RT_SECTION(3, "v"(p.settled));
            }
        }
    }
#ifdef RT_SECTIONS
    if (lane == 0u && rounds) {
        for (int k = 0; k < 4; k++) atomicAdd(stats + 24 + k, cyc[k]);
        atomicAdd(stats + 29, rounds);
    }
#endif
    for (int off = 32; off > 0; off >>= 1) {
        rays += __shfl_down(rays, off, 64);
        taken += __shfl_down(taken, off, 64);
    }
    if (lane == 0u && rays) {
        atomicAdd(stats, (unsigned long long)rays);
        atomicAdd(stats + 16, (unsigned long long)rays);
        atomicAdd(stats + 30, (unsigned long long)taken);
    }
}

// Tail of a batch.  Once only a few paths are alive a bounce costs a launch (and now and then a host
// read-back of the queue lengths) rather than work, so the remaining paths are finished here in one go:
// each lane takes one queued path and runs the ray_color loop (main.rs:44-63) to its end.  A path's draws
// are keyed by (stream, bounce), so the result is the one the bounce-per-launch kernels would have produced.
template <int GROUPS, bool FLAT>
__global__ __launch_bounds__(rtk::BLOCK) void k_finish(Scene S, rtk::RenderConst rc, rtk::PathState in, rtk::Queues q,
                                                        uint32_t bounce, float4* __restrict__ radiance,
                                                        unsigned long long* stats, uint32_t* overflow_flag) {
    extern __shared__ __attribute__((aligned(16))) uint32_t stack[];  // S.stack_cap entries per lane, [entry][lane]
    const uint32_t shard = blockIdx.x % rtk::NSHARD, j0 = blockIdx.x / rtk::NSHARD, stride = gridDim.x / rtk::NSHARD;
    const uint32_t n = q.counts[(bounce % 3u) * rtk::NSHARD + shard];
    const uint32_t shard_base = shard * q.cap_shard;
    const Stack stk{stack + threadIdx.x, S.stack_cap, nullptr, 0u};
    const LdsTables L = stage_tables(S, stack);
    // (paths whose next hit k_shade has settled wait at the far end of the queue; here they are walked like the others)
    const uint32_t n_settled = q.counts[(3u + bounce % 3u) * rtk::NSHARD + shard];
    uint32_t rays = 0;
    for (uint32_t i = j0 * rtk::BLOCK + threadIdx.x; i < n + n_settled; i += stride * rtk::BLOCK) {
        PathRegs p;
        load_path(in, shard_base + (i < n ? i : q.cap_shard - 1u - (i - n)), p);
        for (uint32_t b = p.bounce;; b++) {  // (the path's own bounce number; shade_path ends the path at the depth limit at the latest)
            Hit h;
            trace_ray<rtk::BLOCK, GROUPS, FLAT, false>(S, p.o, p.d, p.time, 0.005f, INF_F, p.k0, p.k1, b, h, stk, overflow_flag, nullptr);
            rays++;
            if (!shade_path<GROUPS == 2>(S, L, rc, p, h, b, radiance)) break;
        }
    }
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off, 64);
    if ((threadIdx.x & 63u) == 0 && rays) atomicAdd(stats + 8, (unsigned long long)rays);
}

// Whole paths, one launch: each wave owns a run of the batch's path slots; a lane generates a camera ray, follows
// the path bounce by bounce in registers (the ray_color loop, main.rs:44-63) and, when the path has ended, takes
// the next slot of the run ("path regeneration": dead lanes wait until at least REGEN of them can generate their
// camera rays together).  Nothing but the final radiance leaves the chip: no path queues, no compaction barriers,
// no launch per bounce.  That only pays where every lane of a wave does the same work per bounce -- scenes traced
// flat; through a tree the lanes' walks diverge and the queue-based kernels, which re-pack rays between the walk
// and the shading, stay ahead.  Draws are keyed by (stream, bounce): the image is the one the other pipelines give.
template <int GROUPS, bool FLAT>
__global__ __launch_bounds__(rtk::BLOCK) __attribute__((amdgpu_waves_per_eu(GROUPS ? 4 : PATHS_WAVES))) void k_paths(
    Scene S, rtk::RenderConst rc, rtk::Batch batch, float4* __restrict__ radiance, unsigned long long* stats,
    uint32_t* overflow_flag, uint32_t REGEN) {
    extern __shared__ __attribute__((aligned(16))) uint32_t stack[];  // S.stack_cap entries per lane, [entry][lane]
    const Stack stk{stack + threadIdx.x, S.stack_cap, nullptr, 0u};
    const LdsTables L = stage_tables(S, stack);
    const uint32_t n = batch.npix * batch.n_samples;
    const uint32_t nw = gridDim.x * (rtk::BLOCK / 64u), wi = blockIdx.x * (rtk::BLOCK / 64u) + (threadIdx.x >> 6);
    const uint32_t per = (((n + nw - 1u) / nw) + 63u) & ~63u;
    uint32_t next = __builtin_amdgcn_readfirstlane(wi * per);
    const uint32_t end = __builtin_amdgcn_readfirstlane(next + per < n ? next + per : n);
    bool have = false;
    PathRegs p;
    p.o = p.d = p.T = mk3(0.0f, 0.0f, 0.0f);
    p.time = 0.0f;
    p.k0 = p.k1 = p.slot = 0u;
    uint32_t bounce = 0, rays = 0;
    if (wi * per < n) {
        for (;;) {
            const unsigned long long idle = __ballot(!have);
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            if (n_idle == 64u && next >= end) break;
            if (next < end && (n_idle >= REGEN || n_idle == 64u)) {
                const uint32_t mine = next + lane_prefix(idle);
                next += n_idle;
                if (!have && mine < end) {
                    camera_path(rc, batch, mine, p);
                    bounce = 0;
                    have = true;
                }
            }
            if (have) {
                Hit h;
                trace_ray<rtk::BLOCK, GROUPS, FLAT, false>(S, p.o, p.d, p.time, 0.005f, INF_F, p.k0, p.k1, bounce, h, stk, overflow_flag, nullptr);
                rays++;
                if (shade_path<GROUPS == 2>(S, L, rc, p, h, bounce, radiance))
                    bounce++;
                else
                    have = false;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off, 64);
    if ((threadIdx.x & 63u) == 0 && rays) atomicAdd(stats, (unsigned long long)rays);
}

// accum[pixel] += radiance of each sample, in sample order (main.rs:478: reduce(|c, src| c + src))
__global__ __launch_bounds__(rtk::BLOCK) void k_accumulate(const float4* __restrict__ radiance, rtk::Batch batch, uint32_t width,
                                                            float* __restrict__ accum_rgb) {
    for (uint32_t pl = blockIdx.x * rtk::BLOCK + threadIdx.x; pl < batch.npix; pl += gridDim.x * rtk::BLOCK) {
        const uint32_t packed = batch.pixel_list[pl];
        const uint32_t pixel = (packed >> 16) * width + (packed & 0xFFFFu);
        float* a = accum_rgb + 3ull * pixel;
        float r = a[0], g = a[1], b = a[2];
        const uint32_t w_log2 = 6u - batch.q_log2, q_mask = (1u << batch.q_log2) - 1u;
        const unsigned long long at_pixel = (unsigned long long)(pl >> w_log2) * 64u + (pl & ((1u << w_log2) - 1u));
        for (uint32_t s = 0; s < batch.n_samples; s++) {
            float4 L = radiance[(unsigned long long)(s >> batch.q_log2) * ((unsigned long long)batch.npix << batch.q_log2) + at_pixel +
                                ((s & q_mask) << w_log2)];
            r = r + L.x;
            g = g + L.y;
            b = b + L.z;
        }
        a[0] = r;
        a[1] = g;
        a[2] = b;
    }
}

// Frame exchange (main.rs:463-482 collects every pixel in one Vec; with the frame's tiles dealt over ranks, each rank
// holds only its own).  Pixels of a rank's tiles are packed in the order rt_owned_pixels lists them: tiles by index,
// rows inside a tile, so tile_off[tile] + (y - y0) * tile_width + (x - x0) is a pixel's place in its owner's block.
DEV uint32_t exchange_slot(const rtk::TileDeal& d, uint32_t p, uint32_t& owner) {
    const uint32_t y = p / d.width, x = p - y * d.width;
    const uint32_t tyi = y / d.tile, txi = x / d.tile;
    const uint32_t tile = tyi * d.tiles_x + txi;
    const uint32_t x0 = txi * d.tile, y0 = tyi * d.tile;
    const uint32_t tw = (x0 + d.tile <= d.width ? d.tile : d.width - x0);
    owner = tile % d.nranks;
    return d.tile_off[tile] + (y - y0) * tw + (x - x0);
}
__global__ __launch_bounds__(rtk::BLOCK) void k_pack_owned(rtk::TileDeal d, uint32_t rank, const float* __restrict__ accum_rgb,
                                                            float* __restrict__ packed) {
    const uint32_t n = d.width * d.height;
    for (uint32_t p = blockIdx.x * rtk::BLOCK + threadIdx.x; p < n; p += gridDim.x * rtk::BLOCK) {
        uint32_t owner;
        const uint32_t slot = exchange_slot(d, p, owner);
        if (owner != rank) continue;
        packed[3ull * slot] = accum_rgb[3ull * p];
        packed[3ull * slot + 1] = accum_rgb[3ull * p + 1];
        packed[3ull * slot + 2] = accum_rgb[3ull * p + 2];
    }
}
__global__ __launch_bounds__(rtk::BLOCK) void k_scatter_gathered(rtk::TileDeal d, const float* __restrict__ gathered,
                                                                  unsigned long long block_pixels, float* __restrict__ frame_rgb) {
    const uint32_t n = d.width * d.height;
    for (uint32_t p = blockIdx.x * rtk::BLOCK + threadIdx.x; p < n; p += gridDim.x * rtk::BLOCK) {
        uint32_t owner;
        const uint32_t slot = exchange_slot(d, p, owner);
        const float* src = gathered + 3ull * (owner * block_pixels + slot);
        frame_rgb[3ull * p] = src[0];
        frame_rgb[3ull * p + 1] = src[1];
        frame_rgb[3ull * p + 2] = src[2];
    }
}

template <bool SQRT>
__global__ __launch_bounds__(rtk::BLOCK) void k_resolve(const float* __restrict__ sum_rgb, unsigned long long n, float inv,
                                                         uint8_t* __restrict__ out) {
    for (unsigned long long i = (unsigned long long)blockIdx.x * rtk::BLOCK + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * rtk::BLOCK) {
        float f = sum_rgb[i] * inv;                      // main.rs:481
        if (SQRT) f = __builtin_sqrtf(f);                // main.rs:37-39 (commented-out write_color)
        float x = 256.0f * rust_clamp(f, 0.0f, 0.999f);  // main.rs:307-309
        uint32_t qv = rust_as_u32(x);
        out[i] = (uint8_t)(qv > 255u ? 255u : qv);
    }
}

template <int GROUPS, bool FLAT>
__global__ __launch_bounds__(rtk::BLOCK) void k_intersect_batch(Scene S, unsigned long long n, const float* __restrict__ ox,
                                                                 const float* __restrict__ oy, const float* __restrict__ oz,
                                                                 const float* __restrict__ dx, const float* __restrict__ dy,
                                                                 const float* __restrict__ dz, const float* __restrict__ tm,
                                                                 float tmin, float tmax, uint32_t seed, int32_t* prim,
                                                                 int32_t* inst, float* t, float* normal, uint8_t* front,
                                                                 float* uv, float* point, uint32_t* overflow_flag) {
    extern __shared__ __attribute__((aligned(16))) uint32_t stack[];  // S.stack_cap entries per lane, [entry][lane]
    const Stack stk{stack + threadIdx.x, S.stack_cap, nullptr, 0u};
    LdsTables no_tables;  // this kernel stages nothing: every record comes from global memory
    no_tables.has_perlin0 = false;
    no_tables.has_scene = false;
    no_tables.perlin0 = nullptr;
    no_tables.prims = no_tables.mats = no_tables.texs = nullptr;
    for (unsigned long long i = (unsigned long long)blockIdx.x * rtk::BLOCK + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.x * rtk::BLOCK) {
        f3 wo{ox[i], oy[i], oz[i]}, wd{dx[i], dy[i], dz[i]};
        float time = tm[i];
        uint32_t k0, k1;
        stream_key(seed, (uint32_t)i, 0u, k0, k1);
        Hit h;
        trace_ray<rtk::BLOCK, GROUPS, FLAT>(S, wo, wd, time, tmin, tmax, k0, k1, 0u, h, stk, overflow_flag);
        bool hit = h.prim != HIT_NONE;
        Record rec;
        rec.point = {0, 0, 0};
        rec.normal = {0, 0, 0};
        rec.u = rec.v = 0.0f;
        rec.front = false;
        if (hit) make_record<true, GROUPS == 2>(S, no_tables, wo, wd, time, h, rec);
        if (prim) {
            int32_t id = -1;
            if (hit) {
                if (h.prim & HIT_MEDIUM_BIT)
                    id = -(int32_t)(S.groups[h.prim & 0x3FFFFFFFu].order + 2u);
                else
                    id = (int32_t)h.key;  // the primitive's id; for a box, the id of the side that was hit
            }
            prim[i] = id;
        }
        if (inst) {  // the instance the record came out of last: the outermost one around the primitive (shapes.rs:74-88)
            int g = hit ? h.grp : -1;
            if (hit && (h.prim & HIT_MEDIUM_BIT)) g = S.groups[h.prim & 0x3FFFFFFFu].parent;  // a medium inside instances
            while (g >= 0 && S.groups[g].parent >= 0) g = S.groups[g].parent;
            inst[i] = g >= 0 ? (int32_t)S.groups[g].order : -1;
        }
        if (t) t[i] = hit ? h.t : INF_F;
        if (normal) {
            normal[3 * i] = rec.normal.x;
            normal[3 * i + 1] = rec.normal.y;
            normal[3 * i + 2] = rec.normal.z;
        }
        if (front) front[i] = (hit && rec.front) ? 1 : 0;
        if (uv) {
            uv[2 * i] = rec.u;
            uv[2 * i + 1] = rec.v;
        }
        if (point) {
            point[3 * i] = rec.point.x;
            point[3 * i + 1] = rec.point.y;
            point[3 * i + 2] = rec.point.z;
        }
    }
}

// Scratch of the entry point that allocates nothing (rt_intersect_batch_noalloc): one array per GPU, held from the
// moment the code object is loaded until the process ends.  Layout: 256 bytes of flags (word 0: traversal stack
// overflow), then the staging area for callers that pass host arrays -- seven input arrays of NOALLOC_MAX_RAYS floats,
// then the output arrays (prim, inst, t: 4 bytes per ray each; normal 12; front 1 padded to 4; uv 8; point 12).
// In host code the name below is only a handle: the device address comes from hipGetSymbolAddress, once per device.
static __device__ __align__(256) unsigned char scratch[rtk::NOALLOC_SCRATCH_BYTES];

inline int grid_flat(unsigned long long n) {
    unsigned long long b = (n + rtk::BLOCK - 1) / rtk::BLOCK;
    if (b < 1) b = 1;
    if (b > (unsigned long long)(rtk::NSHARD * rtk::MAX_BLOCKS_PER_SHARD)) b = rtk::NSHARD * rtk::MAX_BLOCKS_PER_SHARD;
    return (int)b;
}
// Grid size as a multiple of the resident blocks.  Tiles differ a lot in cost (rays that leave the scene at
// once next to rays that walk the whole tree), and blocks own fixed strides of tiles, so many short blocks
// balance far better than one round of long ones: measured gains level off near 16.  The trace kernel of the
// split pipeline balances inside each wave instead and wants long runs per wave: 2.
// (the factors come from rtk::Tuning: oversub, trace_oversub)
// Blocks of kernel `fn` that are resident at once on the current device.
uint32_t resident_blocks(const void* fn, size_t smem, int threads = rtk::BLOCK) {
    static std::mutex mu;
    static std::map<std::tuple<const void*, int, size_t>, uint32_t> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1024;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(std::make_tuple(fn, dev, smem));
    if (it != cache.end()) return it->second;
    int per_cu = 0, cus = 0;
    uint32_t n = 1024;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, smem) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && per_cu > 0 && cus > 0)
        n = (uint32_t)(per_cu * cus);
    else
        (void)hipGetLastError();
    cache[std::make_tuple(fn, dev, smem)] = n;
    return n;
}
inline int grid_sharded(uint32_t blocks_per_shard, const void* fn, int oversub, size_t smem = 0, int threads = rtk::BLOCK) {
    uint32_t lim = std::max<uint32_t>(1u, resident_blocks(fn, smem, threads) * (uint32_t)oversub / rtk::NSHARD);
    if (lim > (uint32_t)rtk::MAX_BLOCKS_PER_SHARD) lim = rtk::MAX_BLOCKS_PER_SHARD;
    if (blocks_per_shard < 1) blocks_per_shard = 1;
    if (blocks_per_shard > lim) blocks_per_shard = lim;
    return (int)(blocks_per_shard * rtk::NSHARD);
}

}  // namespace

namespace rtk {

// groups x instrumented -> template instance
// (the first template argument: 0 no groups, 1 instances / media, 2 instances inside instances as well)
#define RT_DISPATCH_GC(cfg, CALL, F)                   \
    do {                                               \
        if ((cfg).has_groups) {                        \
            if ((cfg).count_visits) CALL(1, true, F);  \
            else CALL(1, false, F);                    \
        } else {                                       \
            if ((cfg).count_visits) CALL(0, true, F);  \
            else CALL(0, false, F);                    \
        }                                              \
    } while (0)
// groups x instrumented x traced-without-a-tree -> template instance
// (a scene with nested instances always takes the tree)
#define RT_DISPATCH_TRACE_CFG(cfg, flat, CALL)             \
    do {                                                   \
        if (flat) RT_DISPATCH_GC(cfg, CALL, true);         \
        else if ((cfg).nested) {                           \
            if ((cfg).count_visits) CALL(2, true, false);  \
            else CALL(2, false, false);                    \
        } else RT_DISPATCH_GC(cfg, CALL, false);           \
    } while (0)

// LDS bytes of the per-lane traversal stacks of one block
inline size_t stack_bytes(const Scene& scene) { return (size_t)std::max(1u, scene.stack_cap) * BLOCK * sizeof(uint32_t); }
// LDS bytes of the tables a shading kernel stages (stage_tables)
inline size_t table_bytes(const Scene& scene) {
    return (scene.n_perlin ? sizeof(PerlinTable) : 0) +
           (scene.lds_scene ? 32u * (size_t)(scene.n_prims + scene.n_mats + scene.n_texs) : 0);
}

hipError_t launch_raygen(hipStream_t s, const Tuning& tn, const RenderConst& rc, const Batch& batch, PathState out, Queues q) {
    unsigned long long n = (unsigned long long)batch.npix * batch.n_samples;
    uint32_t bps = (uint32_t)((n + (unsigned long long)BLOCK * NSHARD - 1) / ((unsigned long long)BLOCK * NSHARD));
    hipLaunchKernelGGL(k_raygen, dim3(grid_sharded(bps, (const void*)&k_raygen, tn.oversub)), dim3(BLOCK), 0, s, rc, batch, out, q);
    return hipGetLastError();
}

// Entries of the traversal stack k_trace keeps in LDS; deeper entries spill to global memory.  The walk takes
// the near child first, so a ray rarely holds more than a few entries, while LDS sized for the worst case
// (tree depth) is what limits the resident waves of this latency-bound kernel.
uint32_t trace_lds_entries(const Tuning& tn, const Scene& scene) {
    return std::min<uint32_t>(std::max(1u, scene.stack_cap), (uint32_t)std::max(1, tn.stack_lds));
}
// Makes sure `bytes` of spill columns exist.
static hipError_t ensure_spill(SpillArea& spill, size_t need) {
    if (need <= spill.bytes) return hipSuccess;
    if (spill.p) (void)hipFree(spill.p);
    spill.p = nullptr;
    spill.bytes = 0;
    hipError_t e = hipMalloc(&spill.p, need);
    if (e == hipSuccess) spill.bytes = need;
    return e;
}
// Dynamic LDS beyond 64 KiB has to be asked for once per kernel.
static hipError_t allow_lds(const void* fn, size_t smem) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> allowed;  // (kernel, device): the attribute belongs to the device's copy of the code
    if (smem <= 64u * 1024u) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    auto it = allowed.find(std::make_pair(fn, dev));
    if (it != allowed.end() && it->second >= smem) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e == hipSuccess) allowed[std::make_pair(fn, dev)] = smem;
    return e;
}
uint32_t crown_capacity(const Tuning& tn, bool has_groups, bool wide, uint32_t stack_cap) {
    if (tn.crown_nodes <= 0) return 0u;
    const size_t lds_entries = std::min<uint32_t>(std::max(1u, stack_cap), (uint32_t)std::max(1, tn.stack_lds));
    const size_t per_block = 80u * 1024u;  // two blocks per compute unit (160 KiB)
    const size_t fixed = CROWN_INST_SLOTS * sizeof(uint32_t) + lds_entries * CROWN_BLOCK * sizeof(uint32_t) +
                         (has_groups ? (size_t)CROWN_BLOCK * 6 * sizeof(float) : 0);
    if (fixed >= per_block) return 0u;
    const size_t fit = (per_block - fixed) / (wide ? sizeof(WideNode) : sizeof(Node));
    return (uint32_t)std::min<size_t>(std::min<size_t>(fit, (size_t)tn.crown_nodes), CROWN_MAX);
}
// Whether launch_trace walks with the crown in LDS (see rtd::CROWN_BASE).
bool trace_uses_crown(const Tuning& tn, const TraceCfg& cfg, const Scene& scene) {
    return tn.crown_nodes > 0 && scene.crown_n > 0u && !scene.flat && !cfg.nested && !cfg.count_visits;
}
hipError_t launch_trace(hipStream_t s, const Tuning& tn, const TraceCfg& cfg, const Scene& scene, PathState in, Queues q, uint32_t bounce,
                        uint32_t blocks_per_shard, float4* hits, unsigned long long* stats, uint32_t* overflow_flag,
                        SpillArea& spill) {
    const uint32_t lds_entries = trace_lds_entries(tn, scene);
    const uint32_t deep = std::max(1u, scene.stack_cap) - lds_entries;
    // A wave refills its free lanes from its own run of the queue, so a run of one tile gives it nothing to refill
    // from.  When few rays are queued, launch fewer blocks rather than shorter runs: at least min_run rays per wave.
    const int min_run = std::max(8, tn.trace_min_run);
    // (launches of a few hundred thousand rays run a little faster in the small blocks: the crown has to be copied into
    // every block before its first ray)
    if (trace_uses_crown(tn, cfg, scene) && blocks_per_shard >= (uint32_t)tn.crown_min_blocks) {
        // Large blocks, so that one copy of the crown serves many waves: [crown][stacks][world rays]
        constexpr int TB = CROWN_BLOCK;
        const size_t smem = (size_t)(CROWN_INST_SLOTS + scene.crown_n * (scene.wide ? 32u : 16u)) * sizeof(uint32_t) +
                            (size_t)lds_entries * TB * sizeof(uint32_t) + (cfg.has_groups ? (size_t)TB * 6 * sizeof(float) : 0);
        Scene sc = scene;
        sc.root_ref = scene.crown_root;
        const uint32_t bps = std::max(1u, (blocks_per_shard * BLOCK + (TB / 64) * min_run - 1) / ((TB / 64) * min_run));
#define RT_TRACE_CROWN(G, W)                                                                                                          \
    do {                                                                                                                              \
        const void* fn = (const void*)&k_trace<G, false, false, W, TB, true>;                                                         \
        hipError_t e = allow_lds(fn, smem);                                                                                           \
        if (e != hipSuccess) return e;                                                                                                \
        const int grid = grid_sharded(bps, fn, tn.trace_oversub, smem, TB);                                                           \
        if ((e = ensure_spill(spill, (size_t)deep * grid * TB * sizeof(uint32_t))) != hipSuccess) return e;                           \
        hipLaunchKernelGGL((k_trace<G, false, false, W, TB, true>), dim3(grid), dim3(TB), smem, s, sc, in, q, bounce, hits, stats,    \
                           overflow_flag, (uint32_t)tn.refill, tn.inner_steps, lds_entries, (uint32_t*)spill.p);                      \
    } while (0)
        if (cfg.has_groups) {
            if (scene.wide) RT_TRACE_CROWN(1, 4);
            else RT_TRACE_CROWN(1, 2);
        } else {
            if (scene.wide) RT_TRACE_CROWN(0, 4);
            else RT_TRACE_CROWN(0, 2);
        }
#undef RT_TRACE_CROWN
        return hipGetLastError();
    }
    const dim3 block(BLOCK);
    // stacks, then (scenes with instances) each lane's world ray
    const size_t smem = (size_t)lds_entries * BLOCK * sizeof(uint32_t) + (cfg.has_groups ? (size_t)BLOCK * 6 * sizeof(float) : 0);
    blocks_per_shard = std::max(1u, (blocks_per_shard * BLOCK + (BLOCK / 64) * min_run - 1) / ((BLOCK / 64) * min_run));
#define RT_CALL(G, C, F)                                       \
    do {                                                       \
        if (scene.wide) RT_TRACE(G, C, F, 4);                  \
        else RT_TRACE(G, C, F, 2);                             \
    } while (0)
#define RT_TRACE(G, C, F, W)                                                                                                     \
    do {                                                                                                                         \
        const int grid = grid_sharded(blocks_per_shard, (const void*)&k_trace<G, C, F, W>, tn.trace_oversub, smem);      \
        hipError_t e = ensure_spill(spill, (size_t)deep * grid * BLOCK * sizeof(uint32_t));                                      \
        if (e != hipSuccess) return e;                                                                                           \
        hipLaunchKernelGGL((k_trace<G, C, F, W>), dim3(grid), block, smem, s, scene, in, q, bounce, hits, stats, overflow_flag,  \
                           (uint32_t)tn.refill, tn.inner_steps, lds_entries,                                                     \
                           (uint32_t*)spill.p);                                                                                  \
    } while (0)
    RT_DISPATCH_TRACE_CFG(cfg, scene.flat != 0u, RT_CALL);
#undef RT_CALL
#undef RT_TRACE
    return hipGetLastError();
}

bool scene_settles(const Scene& scene) { return scene.n_media != 0u && scene.near0_n != NEAR_NONE; }

hipError_t launch_shade(hipStream_t s, const Tuning& tn, const Scene& scene, const RenderConst& rc, PathState in, PathState out,
                        const float4* hits, Queues q, uint32_t bounce, uint32_t blocks_per_shard, float4* radiance,
                        unsigned long long* stats, bool may_settle) {
    const size_t smem = table_bytes(scene);
    // lanes of a wave that have to qualify for its paths to be settled ahead (see k_shade)
    const uint32_t entry_lanes = std::min(64u, std::max(1u, tn.settle_entry));
    const bool settle = may_settle && scene_settles(scene);
    const uint32_t near_hits = tn.settle_near_hits;
#define RT_SHADE(N, T, F)                                                                                                                 \
    hipLaunchKernelGGL((k_shade<N, T, F>), dim3(grid_sharded(blocks_per_shard, (const void*)&k_shade<N, T, F>, tn.oversub, smem)), \
                       dim3(BLOCK), smem, s, scene, rc, in, out, hits, q, bounce, radiance, stats, entry_lanes, near_hits)
    if (scene.nested) {
        if (settle) RT_SHADE(true, true, false);
        else RT_SHADE(true, false, false);
    } else {
        if (settle && scene.settle_fast && tn.settled_fast) RT_SHADE(false, true, true);
        else if (settle) RT_SHADE(false, true, false);
        else RT_SHADE(false, false, false);
    }
#undef RT_SHADE
    return hipGetLastError();
}

hipError_t launch_settled(hipStream_t s, const Tuning& tn, const Scene& scene, const RenderConst& rc, PathState in, PathState out, Queues q,
                          uint32_t bounce, float4* radiance, unsigned long long* stats) {
    const size_t smem = table_bytes(scene);
    const uint32_t refill = (uint32_t)std::min(64, std::max(1, tn.settled_refill));
    const uint32_t min_run = (uint32_t)std::max(64, tn.settled_min_run) & ~63u;
    const uint32_t drain = (uint32_t)std::min(64, std::max(1, tn.settled_drain));
    // persistent waves: as many blocks as are resident at once (times settled_oversub); the kernel cuts the far end of every
    // shard into one run per wave and gives waves beyond the last run nothing
    const uint32_t slow_min = (uint32_t)std::min(64, std::max(1, tn.settled_slow_min));
#define RT_SETTLED(N, F)                                                                                                                      \
    hipLaunchKernelGGL((k_settled<N, F>), dim3(grid_sharded(MAX_BLOCKS_PER_SHARD, (const void*)&k_settled<N, F>, tn.settled_oversub, smem)), \
                       dim3(BLOCK), smem, s, scene, rc, in, out, q, bounce, radiance, stats, refill, min_run, drain, tn.settle_near_hits, slow_min, \
                       (uint32_t)std::max(1, tn.settled_late_rounds))
    if (scene.nested) RT_SETTLED(true, false);
    else if (scene.settle_fast && tn.settled_fast) RT_SETTLED(false, true);
    else RT_SETTLED(false, false);
#undef RT_SETTLED
    return hipGetLastError();
}

hipError_t launch_bounce(hipStream_t s, const Tuning& tn, const TraceCfg& cfg, const Scene& scene, const RenderConst& rc, const Batch& batch,
                         PathState in, PathState out, Queues q, uint32_t bounce, uint32_t blocks_per_shard, float4* radiance,
                         unsigned long long* stats, uint32_t* overflow_flag) {
    const dim3 block(BLOCK);
    const size_t smem = stack_bytes(scene) + table_bytes(scene);
    if (bounce == 0 && scene.settle_fast && tn.settled_fast && cfg.has_groups && !cfg.count_visits && !cfg.nested && !scene.flat) {
        hipLaunchKernelGGL((k_bounce<1, false, true, false, true>),
                           dim3(grid_sharded(blocks_per_shard, (const void*)&k_bounce<1, false, true, false, true>, tn.oversub, smem)), block, smem, s,
                           scene, rc, batch, in, out, q, bounce, radiance, stats, overflow_flag);
    } else if (bounce == 0) {
#define RT_CALL(G, C, F)                                                                                                  \
    hipLaunchKernelGGL((k_bounce<G, C, true, F>),                                                                         \
                       dim3(grid_sharded(blocks_per_shard, (const void*)&k_bounce<G, C, true, F>, tn.oversub, smem)), block, \
                       smem, s, scene, rc, batch, in, out, q, bounce, radiance, stats, overflow_flag)
        RT_DISPATCH_TRACE_CFG(cfg, scene.flat != 0u, RT_CALL);
#undef RT_CALL
    } else {
#define RT_CALL(G, C, F)                                                                                                   \
    hipLaunchKernelGGL((k_bounce<G, C, false, F>),                                                                         \
                       dim3(grid_sharded(blocks_per_shard, (const void*)&k_bounce<G, C, false, F>, tn.oversub, smem)), block, \
                       smem, s, scene, rc, batch, in, out, q, bounce, radiance, stats, overflow_flag)
        RT_DISPATCH_TRACE_CFG(cfg, scene.flat != 0u, RT_CALL);
#undef RT_CALL
    }
    return hipGetLastError();
}

hipError_t launch_finish(hipStream_t s, const TraceCfg& cfg, const Scene& scene, const RenderConst& rc, PathState in, Queues q,
                         uint32_t bounce, uint32_t blocks_per_shard, float4* radiance, unsigned long long* stats,
                         uint32_t* overflow_flag) {
    const size_t smem = stack_bytes(scene) + table_bytes(scene);
    if (blocks_per_shard < 1) blocks_per_shard = 1;
    if (blocks_per_shard > (uint32_t)MAX_BLOCKS_PER_SHARD) blocks_per_shard = MAX_BLOCKS_PER_SHARD;
    const dim3 grid(blocks_per_shard * NSHARD), block(BLOCK);
#define RT_FINISH(G, F) hipLaunchKernelGGL((k_finish<G, F>), grid, block, smem, s, scene, rc, in, q, bounce, radiance, stats, overflow_flag)
    if (cfg.nested) {
        RT_FINISH(2, false);
    } else if (cfg.has_groups) {
        if (scene.flat) RT_FINISH(1, true);
        else RT_FINISH(1, false);
    } else {
        if (scene.flat) RT_FINISH(0, true);
        else RT_FINISH(0, false);
    }
#undef RT_FINISH
    return hipGetLastError();
}

hipError_t launch_paths(hipStream_t s, const Tuning& tn, const TraceCfg& cfg, const Scene& scene, const RenderConst& rc, const Batch& batch,
                        float4* radiance, unsigned long long* stats, uint32_t* overflow_flag) {
    const size_t smem = stack_bytes(scene) + table_bytes(scene);
    const unsigned long long n = (unsigned long long)batch.npix * batch.n_samples;
    const uint32_t regen = (uint32_t)std::min(64, std::max(1, tn.regen));
    // Grid: runs of about 1536 path slots per wave, but never fewer blocks than fit on the chip at once.  Runs differ a
    // lot in cost (sky next to glass) and the block scheduler balances them, so short runs are good -- until a run is
    // so short that the lanes idling while its last paths drain (up to max_depth bounces) weigh against it.
    const unsigned long long run = (unsigned long long)std::max(24, tn.paths_run);
    const unsigned long long want = (n + (BLOCK / 64) * run - 1) / ((BLOCK / 64) * run);
#define RT_PATHS(G, F)                                                                                                      \
    do {                                                                                                                    \
        const unsigned long long fit = resident_blocks((const void*)&k_paths<G, F>, smem);                                  \
        const unsigned long long all = (n + BLOCK - 1) / BLOCK;                                                             \
        /* small batches: 16 rounds of blocks if that leaves runs of 512 slots, for balance */                              \
        const unsigned long long small = std::min(16 * fit, (n + (BLOCK / 64) * 512ull - 1) / ((BLOCK / 64) * 512ull));     \
        const dim3 grid((uint32_t)std::max<unsigned long long>(1, std::min(all, std::min(std::max({want, fit, small}), 1ull << 20)))); \
        hipLaunchKernelGGL((k_paths<G, F>), grid, dim3(BLOCK), smem, s, scene, rc, batch, radiance, stats, overflow_flag, regen); \
    } while (0)
    if (cfg.nested) {
        RT_PATHS(2, false);
    } else if (cfg.has_groups) {
        if (scene.flat) RT_PATHS(1, true);
        else RT_PATHS(1, false);
    } else {
        if (scene.flat) RT_PATHS(0, true);
        else RT_PATHS(0, false);
    }
#undef RT_PATHS
    return hipGetLastError();
}

hipError_t launch_accumulate(hipStream_t s, const float4* radiance, const Batch& batch, uint32_t width, float* accum_rgb) {
    hipLaunchKernelGGL(k_accumulate, dim3(grid_flat(batch.npix)), dim3(BLOCK), 0, s, radiance, batch, width, accum_rgb);
    return hipGetLastError();
}

hipError_t launch_resolve(hipStream_t s, const float* sum_rgb, uint64_t n_values, uint32_t spp, bool sqrt_transfer, uint8_t* out) {
    float inv = 1.0f / (float)spp;  // main.rs:460
    if (sqrt_transfer)
        hipLaunchKernelGGL(k_resolve<true>, dim3(grid_flat(n_values)), dim3(BLOCK), 0, s, sum_rgb, (unsigned long long)n_values, inv, out);
    else
        hipLaunchKernelGGL(k_resolve<false>, dim3(grid_flat(n_values)), dim3(BLOCK), 0, s, sum_rgb, (unsigned long long)n_values, inv, out);
    return hipGetLastError();
}

hipError_t launch_pack_owned(hipStream_t s, const TileDeal& d, uint32_t rank, const float* accum_rgb, float* packed) {
    hipLaunchKernelGGL(k_pack_owned, dim3(grid_flat((unsigned long long)d.width * d.height)), dim3(BLOCK), 0, s, d, rank, accum_rgb, packed);
    return hipGetLastError();
}
hipError_t launch_scatter_gathered(hipStream_t s, const TileDeal& d, const float* gathered, uint64_t block_pixels, float* frame_rgb) {
    hipLaunchKernelGGL(k_scatter_gathered, dim3(grid_flat((unsigned long long)d.width * d.height)), dim3(BLOCK), 0, s, d, gathered,
                       (unsigned long long)block_pixels, frame_rgb);
    return hipGetLastError();
}

hipError_t noalloc_scratch_address(void** out) {
    // with the device current whose copy of the array is wanted
    return hipGetSymbolAddress(out, HIP_SYMBOL(scratch));
}

hipError_t launch_intersect_batch(hipStream_t s, const TraceCfg& cfg, const Scene& scene, uint64_t n, const float* const rays[7],
                                  float tmin, float tmax, uint32_t seed, int32_t* prim, int32_t* inst, float* t, float* normal,
                                  uint8_t* front, float* uv, float* point, uint32_t* overflow_flag) {
    const dim3 grid(grid_flat(n)), block(BLOCK);
    TraceCfg c = cfg;
    c.count_visits = false;
#define RT_CALL(G, C, F)                                                                                                       \
    hipLaunchKernelGGL((k_intersect_batch<G, F>), grid, block, stack_bytes(scene), s, scene, (unsigned long long)n, rays[0], rays[1], \
                       rays[2], rays[3], rays[4], rays[5], rays[6], tmin, tmax, seed, prim, inst, t, normal, front, uv, point,  \
                       overflow_flag)
    RT_DISPATCH_TRACE_CFG(c, scene.flat != 0u, RT_CALL);
#undef RT_CALL
    return hipGetLastError();
}

}  // namespace rtk
