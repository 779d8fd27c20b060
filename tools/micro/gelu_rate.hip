// Vector-ALU cost of the exact GELU epilogue (gemm_epi.h) per element on gfx950, two waves per SIMD: scalar one element at a time,
// N elements interleaved (independent chains side by side), and the packed-fp32 pair version.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -Itinyvc_amd/csrc tools/micro/gelu_rate.hip -o tools/micro/gelu_rate_bin
#include "gemm_epi.h"
#include <cstdio>
using namespace tvc;

template <int IL>
__device__ __forceinline__ void gelu_il(float (&o)[IL]) {       // IL independent evaluations, operation by operation
    float a[IL], t[IL], s[IL], r[IL], u[IL], p[IL], e[IL];
#pragma unroll
    for (int i = 0; i < IL; ++i) { a[i] = o[i] * 0.70710678118654752f; t[i] = fabsf(a[i]); s[i] = a[i] * a[i]; }
#pragma unroll
    for (int i = 0; i < IL; ++i) r[i] = fmaf(-1.72853470e-5f, t[i], 3.83197126e-4f);
#pragma unroll
    for (int i = 0; i < IL; ++i) u[i] = fmaf(-3.88396438e-3f, t[i], 2.42546219e-2f);
#pragma unroll
    for (int i = 0; i < IL; ++i) r[i] = fmaf(r[i], s[i], u[i]);
#pragma unroll
    for (int i = 0; i < IL; ++i) p[i] = fmaf(-5.96761703e-4f, s[i], 4.99119423e-3f);
#pragma unroll
    for (int i = 0; i < IL; ++i) r[i] = fmaf(r[i], t[i], -1.06777877e-1f);
#pragma unroll
    for (int i = 0; i < IL; ++i) p[i] = fmaf(p[i], s[i], -2.67681349e-2f);
#pragma unroll
    for (int i = 0; i < IL; ++i) r[i] = fmaf(r[i], t[i], -6.34846687e-1f);
#pragma unroll
    for (int i = 0; i < IL; ++i) p[i] = fmaf(p[i], s[i], 1.12819925e-1f);
#pragma unroll
    for (int i = 0; i < IL; ++i) r[i] = fmaf(r[i], t[i], -1.28717512e-1f);
#pragma unroll
    for (int i = 0; i < IL; ++i) p[i] = fmaf(p[i], s[i], -3.76125336e-1f);
#pragma unroll
    for (int i = 0; i < IL; ++i) r[i] = fmaf(r[i], t[i], -t[i]);
#pragma unroll
    for (int i = 0; i < IL; ++i) p[i] = fmaf(p[i], s[i], 1.28379166e-1f);
#pragma unroll
    for (int i = 0; i < IL; ++i) e[i] = expf(r[i]);
#pragma unroll
    for (int i = 0; i < IL; ++i) p[i] = fmaf(p[i], a[i], a[i]);
#pragma unroll
    for (int i = 0; i < IL; ++i) r[i] = copysignf(1.0f - e[i], a[i]);
#pragma unroll
    for (int i = 0; i < IL; ++i) o[i] = 0.5f * o[i] * (1.f + (t[i] > 0.927734375f ? r[i] : p[i]));
}

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, float seed) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed * (threadIdx.x + 1) * 0.01f + i * 0.37f - 1.5f;
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { float x = v[i]; asm volatile("" : "+v"(x)); v[i] = act_apply(x, ACT_GELU) + 0.25f; }
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; i += 2) { f32x2e x = {v[i], v[i + 1]}; asm volatile("" : "+v"(x)); x = gelu_pair(x); v[i] = x[0] + 0.25f; v[i + 1] = x[1] + 0.25f; }
        } else if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 8; i += 4) { float x[4] = {v[i], v[i + 1], v[i + 2], v[i + 3]}; gelu_il<4>(x); for (int j = 0; j < 4; ++j) v[i + j] = x[j] + 0.25f; }
        } else if (MODE == 8) {
            gelu_il<8>(v);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += 0.25f;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
static void run(const char* name) {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 512 * 4);
    hipMalloc(&cyc, 64);
    const int iters = 500;
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(512), 0, 0, out, cyc, iters, 1.0f);
    hipDeviceSynchronize();
    unsigned long long h[8];
    float ho[512];
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    hipMemcpy(ho, out, 2048, hipMemcpyDeviceToHost);
    double cs = 0;
    for (int i = 0; i < 512; ++i) cs += ho[i];
    printf("%-28s SIMD cycles per element (two waves): %.1f   checksum %.9g\n", name, (double)(h[4] > h[0] ? h[4] : h[0]) / (iters * 8.0 * 2.0), cs);
}

int main() {
    run<1>("scalar, one at a time");
    run<2>("packed pairs");
    run<4>("4 interleaved");
    run<8>("8 interleaved");
    return 0;
}
