// Microbenchmark: sustained rate of v_mfma_f32_32x32x16_bf16 (and the fp32 32x32x2) with W waves per SIMD on every CU,
// s_memtime ticks vs wall time.  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool BF, bool RAND = false>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* ticks, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b, av[4], bv[4];
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(threadIdx.x * 2 + j); }
    // RAND: operands with random mantissa bits that differ per MFMA (data toggling as in a real GEMM)
    unsigned seed = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int q = 0; q < 4; ++q)
        for (int j = 0; j < 8; ++j) {
            seed = seed * 1664525u + 1013904223u;
            av[q][j] = (__bf16)(((int)(seed >> 8) % 2001 - 1000) * 1e-3f);
            seed = seed * 1664525u + 1013904223u;
            bv[q][j] = (__bf16)(((int)(seed >> 8) % 2001 - 1000) * 1e-3f);
        }
    float fa = threadIdx.x, fb = threadIdx.x * 0.5f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (BF && RAND) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[(i + u) & 3], bv[(i * 2 + u) & 3], acc[i], 0, 0, 0);
                else if (BF) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i], 0, 0, 0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int NACC, bool BF, bool RAND = false>
void run(int blocks, int iters, const char* name, int threads = 256) {
    float* out; unsigned long long* ticks;
    hipMalloc(&out, blocks * 1024 * 4); hipMalloc(&ticks, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, BF, RAND><<<blocks, threads>>>(out, ticks, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC, BF, RAND><<<blocks, threads>>>(out, ticks, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[4]; hipMemcpy(h, ticks, 32, hipMemcpyDeviceToHost);
    double nm = (double)iters * NACC;                       // MFMAs per wave
    double per_simd = nm * (blocks / 256.0);                // MFMAs per SIMD (4 waves per block = 1 per SIMD)
    printf("%s blocks=%d: %.3f ms, ticks/wave=%llu -> %.1f ticks per MFMA per wave; wall: %.1f ns per MFMA per SIMD => %.2f GHz-equivalent at 32 cyc; ticks/us=%.0f\n",
           name, blocks, ms, h[0], h[0] / nm, ms * 1e6 / per_simd, 32.0 / (ms * 1e6 / per_simd), h[0] / (ms * 1e3) * (blocks <= 256 ? 1.0 : 256.0 / blocks));
    hipFree(out); hipFree(ticks);
}

int main() {
    run<3, true>(256, 20000, "bf16 3acc 1w/simd");
    run<3, true, true>(256, 20000, "bf16 3acc 1w/simd RANDOM operands");
    run<3, true, true>(256, 20000, "bf16 3acc RANDOM, 512-thread blocks (2 waves/SIMD co-resident)", 512);
    run<3, true, true>(256, 20000, "bf16 3acc RANDOM, 768-thread blocks (3 waves/SIMD co-resident)", 768);
    run<3, false>(256, 10000, "f32 3acc, 512-thread blocks (2 waves/SIMD co-resident)", 512);
    run<3, true>(512, 20000, "bf16 3acc 2w/simd");
    run<6, true>(256, 10000, "bf16 6acc 1w/simd");
    run<3, false>(256, 10000, "f32 3acc 1w/simd");
    run<3, false>(512, 10000, "f32 3acc 2w/simd");
    return 0;
}
