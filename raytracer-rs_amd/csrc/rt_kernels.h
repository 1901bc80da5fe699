// Launchers of the HIP kernels (implemented in rt_kernels.hip).  Host code calls only these.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_device.h"

namespace rtk {

constexpr int BLOCK = 256;
// Live paths are kept in NSHARD independent queues.  A block appends its survivors to the queue it read
// from with ONE atomic per 256 paths; with a single queue that word saturates (about 90 returning
// atomics per microsecond on one address) and bounds the whole shade step, with 64 words it does not.
#ifndef RT_NSHARD
#define RT_NSHARD 64
#endif
constexpr int NSHARD = RT_NSHARD;
constexpr int MAX_BLOCKS_PER_SHARD = 512;  // upper bound of any grid; launches use the resident block count

// Wavefront path state, structure of arrays, 56 B per path; element (shard s, i) lives at s*cap_shard + i.
struct PathState {
    float4* o_time;    // origin.xyz, time
    float4* d_slot;    // direction.xyz, bits(slot)
    float4* thr_meta;  // throughput.xyz, bits(bounce number of the ray this entry describes)
    uint2* key;        // RNG stream key (k0, k1)
    // Entries at the far end of a shard only (paths whose next hit is settled, see k_shade): > 0 the distance at which the
    // first medium takes the ray, < 0 minus the distance at which it meets the one primitive near that medium.
    float* settled;
};

// counts: 6 x NSHARD words.  Bounce b reads counts[b % 3], appends to counts[(b + 1) % 3] and clears
// counts[(b + 2) % 3] for the bounce after it.  Rows 3..5 count, in the same rotation, the paths k_shade has queued
// at the far end of each shard (entry cap_shard - 1 - i): their next hit is settled already and the trace kernel
// passes them by.
struct Queues {
    uint32_t* counts;
    uint32_t cap_shard;  // entries per shard, multiple of BLOCK
};

struct RenderConst {
    rtd::Camera cam;
    float background[3];
    uint32_t width, height;
    uint32_t seed;
    uint32_t max_depth;
    // != 0: the radiance buffer of the batch was cleared before the first bounce, and a path that ends with zero
    // radiance writes nothing.  rt_render sets it when the background is black: then nearly every path ends that way,
    // and one streaming clear is far cheaper than a scattered 16-byte write per path.
    uint32_t radiance_cleared;
};

// one wavefront batch: npix pixels (pixel_list) x n_samples samples starting at sample0
struct Batch {
    const uint32_t* pixel_list;  // owned pixels of this chunk, row << 16 | column
    uint32_t npix, sample0, n_samples;
    // Slot layout of a batch.  A run of 64 slots (the camera rays of one wave) holds 64 >> q_log2 neighbouring pixels
    // of the list times 1 << q_log2 consecutive samples: slot = (s / q) * q * npix + (p / w) * 64 + (s % q) * w + p % w
    // with q = 1 << q_log2, w = 64 / q.  q_log2 = 0: plainly slot = s * npix + p.  (q > 1 needs npix % w == 0 and
    // n_samples % q == 0.)
    double inv_qnpix;            // 1.0 / (q * npix)
    uint32_t q_log2;
};

// Launch-shape knobs.  A context reads them from the environment ONCE, when it is created (rt_ctx_create), and hands them
// to every launcher; nothing on a launch path reads the environment.
struct Tuning {
    int oversub = 16;        // RTAMD_OVERSUB: grid of the per-bounce kernels, in rounds of resident blocks
    int trace_oversub = 2;   // RTAMD_TRACE_OVERSUB: the same for the persistent trace kernel
    int stack_lds = 10;      // RTAMD_STACK_LDS: traversal-stack entries k_trace keeps in LDS per lane (deeper ones spill)
    int trace_min_run = 128; // RTAMD_TRACE_MIN_RUN (x8): fewest rays a wave of k_trace is given to refill from
    int refill = 24;         // RTAMD_REFILL: idle lanes at which a wave of k_trace fetches new rays
    int inner_steps = 4;     // RTAMD_INNER_STEPS: inner nodes per round of k_trace before the leaf step
    int regen = 8;           // RTAMD_REGEN: idle lanes at which a wave of k_paths starts new camera rays
    int paths_run = 1536;    // RTAMD_PATHS_RUN (x24): path slots per wave of k_paths
    uint32_t settle_entry = 4;      // RTAMD_SETTLE_ENTRY: lanes of a wave that must qualify for its paths to be settled ahead
    uint32_t settle_near_hits = 1;  // RTAMD_SETTLE_NEAR_HITS: hits on the one primitive near a ball-shaped medium are settled too
    int settled_refill = 16;        // RTAMD_SETTLED_REFILL: idle lanes at which a wave of k_settled takes new paths
    int settled_min_run = 128;      // RTAMD_SETTLED_MIN_RUN: fewest paths a wave of k_settled is given
    int settled_drain = 24;         // RTAMD_SETTLED_DRAIN: a wave of k_settled whose run is used up hands its paths on once fewer are left
    int settled_late_rounds = 2;    // RTAMD_SETTLED_LATE_ROUNDS: rounds a wave of k_settled may still run once its run is used up
    int settled_slow_min = 4;       // RTAMD_SETTLED_SLOW_MIN: lanes of a wave of k_settled that must wait for the general step before it runs
    int settled_fast = 1;           // RTAMD_SETTLED_FAST=0: never the step specialised for sphere-shaped media (Scene::settle_fast)
    int settled_oversub = 1;        // RTAMD_SETTLED_OVERSUB: grid of k_settled, in rounds of resident blocks
    int crown_nodes = 4096;         // RTAMD_CROWN_NODES: most inner nodes k_trace may keep in LDS (0: none); the LDS left beside the stacks bounds it too
    int crown_min_blocks = 256;     // RTAMD_CROWN_MIN_BLOCKS: launches with fewer 256-ray tiles per queue shard (x 64 shards) walk without the crown
    int crown_partial = 0;          // RTAMD_CROWN_PARTIAL: 1 = keep the top of a tree that does not fit as a whole (measured: no gain)
};

struct TraceCfg {
    int stack_entries;  // per-lane traversal stack in LDS (Scene::stack_cap)
    bool has_groups;    // scene holds instances / media
    bool count_visits;  // instrumented traversal (node / primitive counters)
    bool nested = false;  // some instance stands inside another one (Scene::nested): kernels compiled for that walk
};

// stats: 64-bit words: [0] += rays traced by the per-bounce kernels; with count_visits also [1] += inner nodes
// fetched, [2] += primitive records tested ([3..7]: section cycles of the instrumented fused kernel);
// [8] += rays traced by the tail kernel; [16] += rays k_settled followed without the tree walk (also in [0]);
// [30] += queue entries k_settled took up (a path it hands on to the next launch is taken up again there).

// ---- split pipeline: raygen, then per bounce trace + shade --------------------------------------------
hipError_t launch_raygen(hipStream_t s, const Tuning& tn, const RenderConst& rc, const Batch& batch, PathState out, Queues q);
// Global memory behind the LDS part of k_trace's traversal stacks; grown by launch_trace as needed, owned by the caller.
struct SpillArea {
    void* p = nullptr;
    size_t bytes = 0;
};
hipError_t launch_trace(hipStream_t s, const Tuning& tn, const TraceCfg& cfg, const rtd::Scene& scene, PathState in, Queues q,
                        uint32_t bounce, uint32_t blocks_per_shard, float4* hits, unsigned long long* stats,
                        uint32_t* overflow_flag, SpillArea& spill);
hipError_t launch_shade(hipStream_t s, const Tuning& tn, const rtd::Scene& scene, const RenderConst& rc, PathState in, PathState out,
                        const float4* hits, Queues q, uint32_t bounce, uint32_t blocks_per_shard, float4* radiance,
                        unsigned long long* stats, bool may_settle);
// true when launch_shade(..., may_settle = true) queues settled paths for this scene, i.e. launch_settled has to follow
bool scene_settles(const rtd::Scene& scene);
// Inner nodes (64-byte records, or 128-byte ones of the four-child form) that fit into the LDS of a k_trace block beside its
// traversal stacks and world rays, with two blocks of 1024 threads per compute unit.
uint32_t crown_capacity(const Tuning& tn, bool has_groups, bool wide, uint32_t stack_cap);
// The paths launch `bounce - 1` of k_shade queued at the far end of `in`: each is followed in registers, bounce after
// bounce, for as long as its next hit can be settled without the tree; a path that leaves is appended to the front of `out`.
hipError_t launch_settled(hipStream_t s, const Tuning& tn, const rtd::Scene& scene, const RenderConst& rc, PathState in, PathState out,
                          Queues q, uint32_t bounce, float4* radiance, unsigned long long* stats);
// ---- fused pipeline: one kernel per bounce (bounce 0 generates its camera rays itself) ---------------------
hipError_t launch_bounce(hipStream_t s, const Tuning& tn, const TraceCfg& cfg, const rtd::Scene& scene, const RenderConst& rc,
                         const Batch& batch, PathState in, PathState out, Queues q, uint32_t bounce,
                         uint32_t blocks_per_shard, float4* radiance, unsigned long long* stats,
                         uint32_t* overflow_flag);
// ---- tail: every queued path of bounce `bounce` is walked to its end by one lane -----------------------------
hipError_t launch_finish(hipStream_t s, const TraceCfg& cfg, const rtd::Scene& scene, const RenderConst& rc, PathState in,
                         Queues q, uint32_t bounce, uint32_t blocks_per_shard, float4* radiance, unsigned long long* stats,
                         uint32_t* overflow_flag);
// ---- whole paths with regeneration (flat scenes): one launch per batch, no queues --------------------------------
hipError_t launch_paths(hipStream_t s, const Tuning& tn, const TraceCfg& cfg, const rtd::Scene& scene, const RenderConst& rc, const Batch& batch,
                        float4* radiance, unsigned long long* stats, uint32_t* overflow_flag);

hipError_t launch_accumulate(hipStream_t s, const float4* radiance, const Batch& batch, uint32_t width, float* accum_rgb);
hipError_t launch_resolve(hipStream_t s, const float* sum_rgb, uint64_t n_values, uint32_t spp, bool sqrt_transfer, uint8_t* out);
// ---- frame exchange: pixels of a rank's tiles <-> its block of the gathered buffer ---------------------------------
struct TileDeal {
    uint32_t width, height, tile, tiles_x, nranks;
    const uint32_t* tile_off;  // per tile: position of its first pixel in its owner's block (device)
};
hipError_t launch_pack_owned(hipStream_t s, const TileDeal& d, uint32_t rank, const float* accum_rgb, float* packed);
hipError_t launch_scatter_gathered(hipStream_t s, const TileDeal& d, const float* gathered, uint64_t block_pixels, float* frame_rgb);

// rt_intersect_batch_noalloc: at most this many rays per call; its per-GPU static scratch (flags + staging for
// host callers: 28 bytes in and 48 bytes out per ray)
constexpr uint64_t NOALLOC_MAX_RAYS = 65536;
constexpr uint64_t NOALLOC_FLAG_BYTES = 256;
constexpr uint64_t NOALLOC_SCRATCH_BYTES = NOALLOC_FLAG_BYTES + NOALLOC_MAX_RAYS * (7 * 4 + 48);
// device address of the current device's copy of that scratch array
hipError_t noalloc_scratch_address(void** out);
hipError_t launch_intersect_batch(hipStream_t s, const TraceCfg& cfg, const rtd::Scene& scene, uint64_t n,
                                  const float* const rays[7], float tmin, float tmax, uint32_t seed, int32_t* prim,
                                  int32_t* inst, float* t, float* normal, uint8_t* front, float* uv, float* point,
                                  uint32_t* overflow_flag);

}  // namespace rtk
