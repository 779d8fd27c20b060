// Do MFMA and vector-ALU work overlap on a gfx950 SIMD?  One workgroup of 8 waves per CU (two waves per SIMD), cycle counts by s_memtime.
//   mode 0: every wave runs NM MFMAs (4 independent accumulators)            -> matrix-pipe time
//   mode 1: every wave runs NV*8 fmas (8 independent chains)                  -> vector time
//   mode 2: every wave runs both, interleaved in program order (1 MFMA + 8 fma)
//   mode 3: waves 0-3 run the MFMAs, waves 4-7 the fmas (one of each per SIMD)
//   mode 4: like 2 but all MFMAs first, then all fmas (per 64-MFMA block)
// hipcc --offload-arch=gfx950 -O3 tools/micro/overlap.hip -o tools/micro/overlap_bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma_a(f16x8 a, f16x8 b, f32x16 c) {      // accumulator in AccVGPRs
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    return c;
}
__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) {      // 16x16x32: four of them = the work of one 32x32x16
    typedef float f32x4q __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4q t = {c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]};
        t = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, t, 0, 0, 0);
        c[4 * q] = t[0]; c[4 * q + 1] = t[1]; c[4 * q + 2] = t[2]; c[4 * q + 3] = t[3];
    }
    return c;
}
#if defined(USE_K8)
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x8f16(__builtin_shufflevector(a, a, 0, 1, 2, 3), __builtin_shufflevector(b, b, 0, 1, 2, 3), c, 0, 0, 0)
#elif defined(USE_16)
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) mfma16(a, b, c)
#elif defined(USE_AGPR)
#define MFMA(a, b, c) mfma_a(a, b, c)
#else
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#endif
template <int MODE, bool PK>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    const int wave = threadIdx.x >> 6;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 c[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) c[j][r] = 0.f;
    float v[8];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 pv[8];
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 0.01f + i; pv[i] = f32x2{v[i], v[i] + 1.f}; }
    const float m = 1.0001f, ad = 0.5f;
    const bool do_m = MODE == 0 || MODE == 2 || MODE == 4 || (MODE == 3 && wave < 4);
    const bool do_v = MODE == 1 || MODE == 2 || MODE == 4 || (MODE == 3 && wave >= 4);
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        if (MODE == 4) {
            if (do_m)
#pragma unroll
                for (int q = 0; q < 16; ++q) c[q & 3] = MFMA(a, b, c[q & 3]);
            __builtin_amdgcn_sched_barrier(0);
            if (do_v)
#pragma unroll
                for (int q = 0; q < 16; ++q)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (PK) pv[i] = __builtin_elementwise_fma(pv[i], f32x2{m, m}, f32x2{ad, ad});
                        else v[i] = fmaf(v[i], m, ad);
                    }
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (do_m) c[q & 3] = MFMA(a, b, c[q & 3]);
                if (do_v)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (PK) pv[i] = __builtin_elementwise_fma(pv[i], f32x2{m, m}, f32x2{ad, ad});
                        else v[i] = fmaf(v[i], m, ad);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += c[j][r];
    for (int i = 0; i < 8; ++i) s += v[i] + pv[i][0] + pv[i][1];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, bool PK>
static void run(const char* name, int blocks) {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, blocks * 512 * 4);
    hipMalloc(&cyc, blocks * 8 * 8);
    const int iters = 200;
    hipLaunchKernelGGL((k<MODE, PK>), dim3(blocks), dim3(512), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-34s blocks %3d  cycles per (1 MFMA + 8 fma) slot: wave0 %.1f  wave4 %.1f\n", name, blocks, h[0] / (iters * 16.0), h[4] / (iters * 16.0));
    hipFree(out);
    hipFree(cyc);
}

int main() {
    for (int blocks : {1, 256}) {
        run<0, false>("0 MFMA only (2 waves/SIMD)", blocks);
        run<1, false>("1 fma only", blocks);
        run<1, true>("1 pk_fma only", blocks);
        run<2, false>("2 interleaved in every wave", blocks);
        run<2, true>("2 interleaved, pk_fma", blocks);
        run<3, false>("3 MFMA waves | fma waves", blocks);
        run<3, true>("3 MFMA waves | pk_fma waves", blocks);
        run<4, false>("4 blocks of 16 MFMA then 128 fma", blocks);
    }
    return 0;
}
