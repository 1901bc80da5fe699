// Launchers of the HIP kernels (implemented in rt_kernels.hip).  Host code calls only these.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_device.h"

namespace rtk {

// Wavefront path state, structure of arrays, 56 B per path.
struct PathState {
    float4* o_time;    // origin.xyz, time
    float4* d_slot;    // direction.xyz, bits(slot)
    float4* thr_meta;  // throughput.xyz, bits(bounce << 24 | rng counter)
    uint2* key;        // RNG stream key (k0, k1)
};

struct RenderConst {
    rtd::Camera cam;
    float background[3];
    uint32_t width, height;
    uint32_t seed;
    uint32_t max_depth;
};

constexpr int BLOCK = 256;
constexpr int MAX_GRID = 256 * 8;  // 8 blocks per CU; larger batches grid-stride

// stack_entries: 16, 32 or 64 (per-lane traversal stack in LDS); has_groups: scene holds instances/media
hipError_t launch_raygen(hipStream_t s, const RenderConst& rc, const uint32_t* pixel_list, uint32_t npix_batch,
                         uint32_t sample0, uint32_t n_samples, PathState out, uint32_t* counts, float4* radiance);
// rays_total: three 64-bit words: [0] += rays traced; with count_visits also [1] += inner nodes fetched,
// [2] += primitive records tested (instrumented, slower kernel; used to price the algorithmic bytes per ray)
hipError_t launch_trace(hipStream_t s, int stack_entries, bool has_groups, bool count_visits, const rtd::Scene& scene,
                        PathState in, uint32_t* counts, uint32_t which, uint32_t n_bound, float4* hits,
                        unsigned long long* rays_total, uint32_t* overflow_flag);
hipError_t launch_shade(hipStream_t s, const rtd::Scene& scene, const RenderConst& rc, PathState in, PathState out,
                        const float4* hits, uint32_t* counts, uint32_t which, uint32_t n_bound, float4* radiance);
hipError_t launch_accumulate(hipStream_t s, const float4* radiance, const uint32_t* pixel_list, uint32_t npix_batch,
                             uint32_t n_samples, float* accum_rgb);
hipError_t launch_resolve(hipStream_t s, const float* sum_rgb, uint64_t n_values, uint32_t spp, uint8_t* out);
hipError_t launch_intersect_batch(hipStream_t s, int stack_entries, bool has_groups, const rtd::Scene& scene,
                                  uint64_t n, const float* const rays[7], float tmin, float tmax, uint32_t seed,
                                  int32_t* prim, int32_t* inst, float* t, float* normal, uint8_t* front, float* uv,
                                  float* point, uint32_t* overflow_flag);

}  // namespace rtk
