// Streaming / per-column kernels shared by several stages (HBM-bound, no matrix work).
#pragma once
#include <hip/hip_runtime.h>

namespace tvc {

// ATen's linear-interpolation arithmetic (UpSampleKernel.cpp, align_corners=False), verified
// bit-equal against F.interpolate on the build host: src = fma(scale, dst + 0.5, -0.5) clamped at
// 0; i0 = min(floor(src), n-1); lambda = clamp(src - i0, 0, 1); out = fma(1 - lambda, x0, lambda*x1).
struct Lerp {
    int i0, i1;
    float w0, w1;
};
__device__ __forceinline__ Lerp lerp_coord(int dst, float scale, int n_in) {
    float src = fmaf(scale, (float)dst + 0.5f, -0.5f);
    src = src < 0.f ? 0.f : src;
    int i0 = (int)floorf(src);
    i0 = i0 > n_in - 1 ? n_in - 1 : i0;
    float lam = src - (float)i0;
    lam = lam < 0.f ? 0.f : (lam > 1.f ? 1.f : lam);
    Lerp r;
    r.i0 = i0;
    r.i1 = i0 + 1 > n_in - 1 ? n_in - 1 : i0 + 1;
    r.w0 = 1.f - lam;
    r.w1 = lam;
    return r;
}
__device__ __forceinline__ float lerp_eval(const Lerp& c, float x0, float x1) {
    return fmaf(c.w0, x0, __fmul_rn(c.w1, x1));
}

// y[row][d] = lerp(x[row][:]) — F.interpolate(mode='linear') on `rows` independent rows.
static __global__ void lerp_resize_kernel(const float* __restrict__ x, float* __restrict__ y, long rows,
                                          int n_in, int n_out, float scale) {
    long total = rows * (long)n_out;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long row = i / n_out;
        int d = (int)(i - row * n_out);
        Lerp c = lerp_coord(d, scale, n_in);
        const float* xr = x + row * n_in;
        y[i] = lerp_eval(c, xr[c.i0], xr[c.i1]);
    }
}

// max over consecutive windows: y[row][j] = max_{i<win} x[row][j*win + i]   (F.max_pool1d(k=win, s=win))
static __global__ void window_max_kernel(const float* __restrict__ x, float* __restrict__ y, long rows,
                                         int n_out, int win) {
    // one wavefront per output window: coalesced reads, shuffle max
    long total = rows * (long)n_out;
    const int lane = threadIdx.x & 63;
    long wave = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    for (long i = wave; i < total; i += nwaves) {
        const float* p = x + i * win;
        float m = -INFINITY;
        for (int k = lane; k < win; k += 64) m = fmaxf(m, p[k]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) y[i] = m;
    }
}

inline unsigned grid_for(long total, int block = 256, long cap = 256L * 16) {
    long g = (total + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace tvc
