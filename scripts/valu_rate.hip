// Micro-benchmark: issue rate of the VALU instructions the path tracer leans on
// (32-bit integer multiply, shifts/xors, 24-bit multiply, f32 divide/sqrt, f64 fma).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate scripts/valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;
constexpr int CHAINS = 8;

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
    uint32_t x[CHAINS];
    float f[CHAINS];
    double d[CHAINS];
    for (int c = 0; c < CHAINS; ++c) {
        x[c] = seed + threadIdx.x * 2654435761u + c * 40503u;
        f[c] = 1.0f + (float)(x[c] & 1023) * 1e-3f;
        d[c] = (double)f[c];
    }
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if (OP == 0) x[c] = (x[c] * 0x7feb352du) ^ (uint32_t)i;           // v_mul_lo_u32 + xor
            if (OP == 1) x[c] = (x[c] ^ (x[c] >> 15)) + (uint32_t)i;          // shift + xor + add
            if (OP == 2) x[c] = __umul24(x[c], 0x00eb352du) + (uint32_t)i;     // v_mad_u32_u24
            if (OP == 11) x[c] = (x[c] ^ (uint32_t)i) + 0x9E3779B9u;        // xor + add
            if (OP == 3) x[c] = __umulhi(x[c], 0x7feb352du) + i;          // v_mul_hi_u32 (+add)
            if (OP == 4) f[c] = f[c] * 1.0000001f + 0.25f;                // mul + add (unfused)
            if (OP == 5) f[c] = 1.5f / f[c] + 0.5f;                       // IEEE divide (+add)
            if (OP == 6) f[c] = sqrtf(f[c]) + 1.5f;                       // IEEE sqrt (+add)
            if (OP == 7) d[c] = fma(d[c], 1.0000001, 0.25);               // v_fma_f64
            if (OP == 8) f[c] = __builtin_amdgcn_rcpf(f[c]) + 0.5f;       // v_rcp_f32 (+add)
            if (OP == 9) d[c] = log(d[c]) + 3.0;                          // f64 log (+add)
            if (OP == 10) f[c] = __logf(f[c]) + 3.0f;                     // fast f32 log (+add)
        }
    }
    uint32_t acc = 0;
    for (int c = 0; c < CHAINS; ++c) acc ^= x[c] ^ __float_as_uint(f[c]) ^ (uint32_t)__double_as_longlong(d[c]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int OP>
int run(const char* name, uint32_t* out) {
    const int blocks = 256 * 8;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    k<OP><<<blocks, 256>>>(out, 1);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    k<OP><<<blocks, 256>>>(out, 2);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    // wave-instructions per SIMD: blocks*4 waves / 1024 SIMDs * ITERS * CHAINS
    double wave_ops = (double)blocks * 4 / 1024.0 * ITERS * CHAINS;
    double cyc = ms * 1e-3 * 2.4e9 / wave_ops;
    printf("%-28s %8.3f ms  %6.2f cycles per wave-op-group (at 2.4 GHz)\n", name, ms, cyc);
    return 0;
}

int main() {
    uint32_t* out;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    run<0>("v_mul_lo_u32 + xor", out);
    run<11>("xor + add (2 ops)", out);
    run<1>("shift+xor+add (3 ops)", out);
    run<2>("v_mad_u32_u24", out);
    run<3>("v_mul_hi_u32 + add", out);
    run<4>("f32 mul + add (2 ops)", out);
    run<5>("f32 IEEE div + add", out);
    run<6>("f32 IEEE sqrt + add", out);
    run<7>("v_fma_f64", out);
    run<8>("v_rcp_f32 + add", out);
    run<9>("f64 log + add", out);
    run<10>("fast f32 log + add", out);
    CHECK(hipFree(out));
    return 0;
}
