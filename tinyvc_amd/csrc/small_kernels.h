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

// (sin, cos)(a) for |a| <= 2 pi - the oscillator's phase is reduced to one turn before the sine, the noise phases are drawn in
// [-pi, pi) -: two-term Cody-Waite reduction by pi / 2 with fma and the degree-7 / degree-8 kernels of Cephes' sinf / cosf (the cosine
// finished with one rounding).  libm's sinf / sincosf carry their large-argument path and twice the selects.  Checked against the
// correctly rounded values on every float of [2^-20, 2 pi] (sine) and of +-[2^-20, pi] (both): within 1 ulp like libm's, 2.2 % of the
// sines not correctly rounded against libm's 3.1 %, largest absolute error 2^-24 for both (tools/micro/sin2pi.hip, profiles/r03_sin2pi.txt).
__device__ __forceinline__ float2 sincos_small(float a) {
    const float qf = rintf(a * 0.63661977236758134f);
    float r = fmaf(-qf, 1.57079637050628662109375f, a);
    r = fmaf(-qf, -4.37113900018624283e-8f, r);
    const int q = (int)qf;
    const float z = r * r;
    float s = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    s = fmaf(s * z, r, r);
    float c = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    c = fmaf(z, fmaf(z, c, -0.5f), 1.0f);
    const float vs = (q & 1) ? c : s, vc = (q & 1) ? s : c;
    return make_float2((q & 2) ? -vs : vs, ((q + 1) & 2) ? -vc : vc);
}

// shift_frequency of one value (frontend.hip has the notes on its arithmetic; also applied by the pitch decoder when the caller wants the shifted copy)
__device__ __forceinline__ float shift_frequency_one(float f, float shift) {
    float r = __fdiv_rn(f, 440.f);
    r = r < 0.f ? 0.f : r;                       // relu; NaN stays NaN as in F.relu
    const float lg = (float)log2((double)__fadd_rn(r, 1e-6f));
    float midi = __fadd_rn(__fmul_rn(lg, 12.f), 69.f);
    midi = __fadd_rn(midi, shift);
    const float e = __fdiv_rn(__fsub_rn(midi, 69.f), 12.f);
    return __fmul_rn(440.f, (float)exp2((double)e));
}

// The library's own noise phases (noise_angle = NULL): the phase of (utterance row, bin, frame) is a counter-based hash of the call's seed and
// of those three numbers alone - whatever else is in the batch and however a ragged call is cut into in-kernel batches -, uniform in [-pi, pi).
__device__ __forceinline__ float noise_phase_hash(unsigned long long seed, int row, int k, int t) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)row + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z += 0x9E3779B97F4A7C15ull * (((unsigned long long)k << 32) + (unsigned long long)t + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);  // [0, 1)
    return fmaf(u, 6.2831854820251465f, -3.1415927410125732f);      // one rounding (what -ffp-contract=on made of `u * 2 pi - pi`; tinyvc_amd/synth.py noise_phase_hash restates it)
}

// y[row][d] = lerp(x[row][:]) — F.interpolate(mode='linear') on `rows` independent rows.
// A thread owns 4 consecutive outputs: their source coordinates depend on d only, so they are computed once and
// reused for every row the thread walks (blockIdx.y strides the rows); outputs leave as one 16-byte store.
static __global__ __launch_bounds__(256) void lerp_resize_kernel(const float* __restrict__ x, float* __restrict__ y, long rows,
                                                                 int n_in, int n_out, float scale, int tx) {
    // tx (power of two <= 256) threads span a row's 4-output groups, the other 256 / tx thread rows take different rows
    const int sx = threadIdx.x & (tx - 1), sy = threadIdx.x / tx, ny = 256 / tx;
    const int d0 = (blockIdx.x * tx + sx) * 4;
    if (d0 >= n_out) return;
    Lerp c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u] = lerp_coord(d0 + u < n_out ? d0 + u : n_out - 1, scale, n_in);
    const bool vec = (n_out & 3) == 0;                 // rows stay 16-byte aligned
    for (long row = (long)blockIdx.y * ny + sy; row < rows; row += (long)gridDim.y * ny) {
        const float* xr = x + row * n_in;
        float o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] = lerp_eval(c[u], xr[c[u].i0], xr[c[u].i1]);
        float* yr = y + row * n_out + d0;
        if (vec) {
            *reinterpret_cast<float4*>(yr) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (d0 + u < n_out) yr[u] = o[u];
        }
    }
}
struct LerpLaunch {
    dim3 grid;
    int tx;
};
inline LerpLaunch lerp_launch(long rows, int n_out) {
    const int groups = (n_out + 3) / 4;
    int tx = 256;
    while (tx > 16 && tx / 2 >= groups) tx /= 2;
    const unsigned gx = (unsigned)((groups + tx - 1) / tx);
    const int ny = 256 / tx;
    long gy = (rows + ny - 1) / ny;
    const long want = 256L * 16 / (gx ? gx : 1);       // ~16 workgroups per CU in flight
    if (gy > want) gy = want < 1 ? 1 : want;
    return {dim3(gx, (unsigned)gy), tx};
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
