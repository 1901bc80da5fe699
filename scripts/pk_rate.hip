// Micro-benchmark: does one v_pk_fma_f32 (two f32 fmas per lane) issue as fast as one v_fma_f32?
// Build: hipcc --offload-arch=gfx950 -O3 -o pk_rate scripts/pk_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096;
constexpr int CHAINS = 8;

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
    v2f f[CHAINS];
    const v2f m = {1.0000001f, 0.9999999f}, a = {0.25f, 0.125f};
    for (int c = 0; c < CHAINS; ++c) f[c] = v2f{seed + threadIdx.x * 1e-3f + c, seed - c};
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if (OP == 0) {  // two scalar fmas
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[c].x) : "v"(m.x), "v"(a.x));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[c].y) : "v"(m.y), "v"(a.y));
            }
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(f[c]) : "v"(m), "v"(a));  // one packed fma
            if (OP == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[c].x) : "v"(m.x), "v"(a.x));  // one scalar fma
            if (OP == 3) {  // packed fma + select pair (the slab test's shape)
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(f[c]) : "v"(m), "v"(a));
                asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[c].x) : "v"(f[c].y));
            }
        }
    }
    float acc = 0;
    for (int c = 0; c < CHAINS; ++c) acc += f[c].x + f[c].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int OP>
int run(const char* name, float* out, int blocks_per_cu) {
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    k<OP><<<blocks, 256>>>(out, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    k<OP><<<blocks, 256>>>(out, 2.0f);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    double wave_ops = (double)blocks * 4 / 1024.0 * ITERS * CHAINS;  // per SIMD
    printf("%-34s %d blocks/CU %8.3f ms  %6.2f cycles per group per wave (at 2.4 GHz)\n", name, blocks_per_cu, ms, ms * 1e-3 * 2.4e9 / wave_ops);
    return 0;
}

int main() {
    float* out;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    for (int bpc : {1, 2, 8}) {
        run<2>("1 x v_fma_f32", out, bpc);
        run<0>("2 x v_fma_f32", out, bpc);
        run<1>("1 x v_pk_fma_f32", out, bpc);
        run<3>("v_pk_fma_f32 + v_max_f32", out, bpc);
    }
    return 0;
}
