// Microbenchmark: what a wave pays for fetching one 128-byte (or smaller) record PER LANE from scattered addresses, the
// access shape of a BVH walk.  Each lane chases indices through a table (the next index comes out of the record, so the
// fetches of one lane are dependent, as in a tree walk); waves per SIMD and the table size (L2 / Infinity Cache / HBM) are
// arguments.  Forms:
//   P<k>  every lane loads k x 16 bytes of its own record (k global_load_dwordx4 with 64 different lines per instruction)
//   C     the wave loads the 64 records of its lanes with 8 instructions in which 8 neighbouring lanes read one 128-byte
//         line, passes them through LDS (stride 144 B), and every lane reads its own 8 x 16 bytes back
// Prints ns per lane-step per CU and lane-addresses per CU clock.
//   hipcc --offload-arch=gfx950 -O3 -o ta_probe scripts/ta_probe.hip && ./ta_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// active_mod: only lanes with (lane % active_mod) == 0 ... no: lanes whose hash falls below a threshold stay idle for the
// whole kernel (fixed set), so that the cost per ACTIVE lane can be read off.
// share: lanes in groups of `share` neighbours follow the same chain (the same address in every step)
template <int K>
__global__ __launch_bounds__(256) void chase_per_lane(const float4* __restrict__ table, uint32_t mask, int steps, uint32_t active_of_256,
                                                      float* out, uint32_t share = 1) {
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    if ((mix(tid * 7u + 1u) & 255u) >= active_of_256) return;
    const uint32_t chain = tid / share;
    uint32_t idx = mix(chain) & mask;
    float acc = 0.f;
    for (int s = 0; s < steps; s++) {
        const float4* p = table + (size_t)idx * 8u;
        float4 q[K];
#pragma unroll
        for (int k = 0; k < K; k++) q[k] = p[k];
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < K; k++) sum += q[k].y + q[k].z + q[k].w + (k ? q[k].x : 0.f);
        acc += sum;
        idx = (__float_as_uint(q[0].x) + chain) & mask;
    }
    out[tid] = acc;
}

__global__ __launch_bounds__(256) void chase_coop(const float4* __restrict__ table, uint32_t mask, int steps, uint32_t active_of_256, float* out) {
    __shared__ float4 stage[4][64 * 9];
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool active = (mix(tid * 7u + 1u) & 255u) < active_of_256;
    uint32_t idx = mix(tid) & mask;
    float acc = 0.f;
    float4* my = stage[wave];
    for (int s = 0; s < steps; s++) {
        // records of lanes that are idle are not fetched: their group of 8 loader lanes is switched off
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t src = 8u * i + (lane >> 3);
            const uint32_t nidx = __shfl(idx, src);
            const bool on = __shfl((int)active, src);
            if (on) my[src * 9u + (lane & 7u)] = table[(size_t)nidx * 8u + (lane & 7u)];
        }
        __builtin_amdgcn_wave_barrier();
        if (active) {
            float4 q[8];
#pragma unroll
            for (int k = 0; k < 8; k++) q[k] = my[lane * 9u + k];
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 8; k++) sum += q[k].y + q[k].z + q[k].w + (k ? q[k].x : 0.f);
            acc += sum;
            idx = (__float_as_uint(q[0].x) + tid) & mask;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (active) out[tid] = acc;
}

// The same in two halves of 32 records (half the LDS: 4.5 KiB per wave)
__global__ __launch_bounds__(256) void chase_coop2(const float4* __restrict__ table, uint32_t mask, int steps, uint32_t active_of_256, float* out) {
    __shared__ float4 stage[4][32 * 9];
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool active = (mix(tid * 7u + 1u) & 255u) < active_of_256;
    uint32_t idx = mix(tid) & mask;
    float acc = 0.f;
    float4* my = stage[wave];
    for (int s = 0; s < steps; s++) {
        float4 q[8];
#pragma unroll
        for (int half = 0; half < 2; half++) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t rec = 8u * i + (lane >> 3), src = 32u * half + rec;
                const uint32_t nidx = __shfl(idx, src);
                const bool on = __shfl((int)active, src);
                if (on) my[rec * 9u + (lane & 7u)] = table[(size_t)nidx * 8u + (lane & 7u)];
            }
            __builtin_amdgcn_wave_barrier();
            if (active && (lane >> 5) == (uint32_t)half) {
#pragma unroll
                for (int k = 0; k < 8; k++) q[k] = my[(lane & 31u) * 9u + k];
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (active) {
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 8; k++) sum += q[k].y + q[k].z + q[k].w + (k ? q[k].x : 0.f);
            acc += sum;
            idx = (__float_as_uint(q[0].x) + tid) & mask;
        }
    }
    if (active) out[tid] = acc;
}

int main(int argc, char** argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 400;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;
    printf("%s, %d CUs, %.2f GHz, %d steps\n", prop.name, cus, clk * 1e-9, steps);
    const size_t max_nodes = 1u << 24;  // 2 GiB of 128-byte records
    float4* table;
    CHECK(hipMalloc(&table, max_nodes * 128));
    {
        std::vector<uint32_t> h(max_nodes * 32);
        uint32_t x = 12345u;
        for (size_t i = 0; i < h.size(); i++) {
            x = x * 1664525u + 1013904223u;
            h[i] = (i % 32 == 0) ? (x >> 4) : 0u;  // word 0: the next index (masked by the kernel), the rest zeros
        }
        CHECK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    float* out;
    const int max_waves = 8;
    CHECK(hipMalloc(&out, (size_t)cus * max_waves * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const uint32_t sizes_log2[] = {13, 19, 24};  // 1 MiB (every XCD's L2), 64 MiB (Infinity Cache), 2 GiB (HBM)
    const char* size_name[] = {"1MiB", "64MiB", "2GiB"};
    const uint32_t actives[] = {256, 128};
    for (int si = 0; si < 3; si++)
        for (uint32_t act : actives)
            for (int wps : {4, 7}) {
                const uint32_t mask = (1u << sizes_log2[si]) - 1u;
                const int grid = cus * wps;  // blocks of 4 waves: wps waves per SIMD
                auto run = [&](const char* name, int k_loads, auto launch) {
                    launch();  // warm-up
                    CHECK(hipDeviceSynchronize());
                    CHECK(hipEventRecord(e0));
                    launch();
                    CHECK(hipEventRecord(e1));
                    CHECK(hipEventSynchronize(e1));
                    float ms;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    const double lanes = (double)grid * 256 * act / 256.0;
                    const double lane_steps_per_cu = lanes * steps / cus;
                    const double cyc = ms * 1e-3 * clk;
                    printf("%-6s table %-6s active %3u/256 waves/SIMD %d: %8.3f ms  %6.2f clk per lane-step per CU  %5.2f lane-loads per CU clk  %7.1f GB/s per CU\n",
                           name, size_name[si], act, wps, ms, cyc / lane_steps_per_cu, lane_steps_per_cu * k_loads / cyc,
                           lane_steps_per_cu * k_loads * 16 / (ms * 1e-3) * 1e-9);
                };
                run("P1", 1, [&] { hipLaunchKernelGGL(chase_per_lane<1>, dim3(grid), dim3(256), 0, 0, table, mask, steps, act, out); });
                run("P2", 2, [&] { hipLaunchKernelGGL(chase_per_lane<2>, dim3(grid), dim3(256), 0, 0, table, mask, steps, act, out); });
                run("P4", 4, [&] { hipLaunchKernelGGL(chase_per_lane<4>, dim3(grid), dim3(256), 0, 0, table, mask, steps, act, out); });
                run("P7", 7, [&] { hipLaunchKernelGGL(chase_per_lane<7>, dim3(grid), dim3(256), 0, 0, table, mask, steps, act, out); });
                run("P8", 8, [&] { hipLaunchKernelGGL(chase_per_lane<8>, dim3(grid), dim3(256), 0, 0, table, mask, steps, act, out); });
                if (wps <= 4) run("C", 8, [&] { hipLaunchKernelGGL(chase_coop, dim3(grid), dim3(256), 0, 0, table, mask, steps, act, out); });
                run("C2", 8, [&] { hipLaunchKernelGGL(chase_coop2, dim3(grid), dim3(256), 0, 0, table, mask, steps, act, out); });
                CHECK(hipGetLastError());
                if (si == 0 && act == 256 && wps == 7)
                    for (uint32_t share : {2u, 4u, 8u, 16u, 64u}) {
                        char name[16];
                        snprintf(name, sizeof name, "P4/%u", share);
                        run(name, 4, [&] { hipLaunchKernelGGL(chase_per_lane<4>, dim3(grid), dim3(256), 0, 0, table, mask, steps, act, out, share); });
                    }
            }
    return 0;
}
