// Small block-fused FilterNet kernels: downs[0] (17 -> 24, full rate) on an LDS tile + MFMA, and a stand-alone
// output_layer kernel (used only when the last level is not fused, see filter_up24.hip).
#include "igemm.h"
#include "small_kernels.h"
#include "tvc_common.h"

namespace tvc {

__device__ __forceinline__ float lrelu01(float v) { return v > 0.f ? v : 0.1f * v; }

__device__ __forceinline__ void copy_to_lds(float* dst, const float* __restrict__ src, int nfloats, int tid, int nthr) {
    for (int i = tid * 4; i < nfloats; i += nthr * 4)
        *reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(src + i);
}

// acc += sum_{tap, ci} W[tap*CIN+ci][m] * act(Xs[ci][clamp(colc + (tap - TAPS/2)*dil)])
// Ws: LDS [TAPS*CIN][32]; Xs: LDS [CIN][xs]; one 32(m) x 32(n) tile per call.
template <int CIN, int TAPS, bool LRELU>
__device__ __forceinline__ void conv_mma_tile(f32x16& acc, const float* Ws, const float* Xs, int xs, int colc,
                                              int dil, int lo, int hi, int l31, int lh) {
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
        int c = colc + (tap - TAPS / 2) * dil;
        c = c < lo ? lo : (c > hi ? hi : c);
        const float* xp = Xs + lh * xs + c;
        const float* wp = Ws + (tap * CIN + lh) * 32 + l31;
#pragma unroll
        for (int ci = 0; ci < CIN; ci += 2) {
            float a = wp[ci * 32];
            float b = xp[ci * xs];
            if (LRELU) b = lrelu01(b);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
}

// FilterNet.downs[0] (decoder.py:206,224,227): Conv1d(17 -> 24, k3, replicate) over cat[source, energy]
// at the full sample rate.  HBM-bound (reads 17 rows, writes 24): one LDS tile per workgroup, the
// 51-deep contraction on the matrix pipe (channel count padded to 18 with a zero row).
struct Down0Args {
    const float* source;  // [B][16][L]
    const float* energy;  // [B][1][L]
    const float* wt;      // tap-major image [3*17][32]
    const float* bias;
    float* out;           // [B][24][L]
    int len, tiles_per_utt;
};

static __global__ __launch_bounds__(256) void down0_kernel(Down0Args a) {
    constexpr int W = 512, XS = W + 4, CP = 18;
    __shared__ __attribute__((aligned(16))) float Xs[CP * XS];
    __shared__ __attribute__((aligned(16))) float Ws[3 * CP * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.x / a.tiles_per_utt;
    const int t0 = (blockIdx.x - b * a.tiles_per_utt) * W;
    const int len = a.len;
    const float* sb = a.source + (long)b * 16 * len;
    const float* eb = a.energy + (long)b * len;
    for (int i = tid; i < CP * (W + 2); i += 256) {
        int ci = i / (W + 2), c = i - ci * (W + 2);
        int p = t0 - 1 + c;
        p = p < 0 ? 0 : (p > len - 1 ? len - 1 : p);
        float v = ci < 16 ? sb[(long)ci * len + p] : (ci == 16 ? eb[p] : 0.f);
        Xs[ci * XS + c] = v;
    }
    for (int i = tid; i < 3 * CP * 32; i += 256) {
        int row = i >> 5, m = i & 31;
        int tap = row / CP, ci = row - tap * CP;
        Ws[i] = ci < 17 ? a.wt[(tap * 17 + ci) * 32 + m] : 0.f;
    }
    __syncthreads();
    const int lo = 1 - t0 > 0 ? 1 - t0 : 0;
    const int hi = (len - t0) < (W + 1) ? (len - t0) : (W + 1);     // column of position len-1 is len-1-t0+1
    for (int nt = wave; nt < W / 32; nt += 4) {
        const int n = nt * 32 + l31;
        const int t = t0 + n;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        conv_mma_tile<CP, 3, false>(acc, Ws, Xs, XS, n + 1, 1, lo, hi, l31, lh);
        if (t < len) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < 24) a.out[((long)b * 24 + m) * len + t] = acc[r] + a.bias[m];
            }
        }
    }
}

int run_down0(tvc_ctx* ctx, hipStream_t s, const PackedW& w, const float* source, const float* energy, float* out, int B, int len) {
    Down0Args a{source, energy, w.At_tap, w.bias, out, len, (len + 511) / 512};
    hipLaunchKernelGGL(down0_kernel, dim3((unsigned)(B * a.tiles_per_utt)), dim3(256), 0, s, a);
    return launch_check(ctx, "down0");
}

// FilterNet.output_layer (decoder.py:220,233): Conv1d(24 -> 1, k7, replicate).  One output channel:
// a 168-tap dot product per sample, HBM-bound (reads 24 rows once) -> plain VALU, weights in LDS.
static __global__ __launch_bounds__(256) void out_conv7_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ y,
                                                               int C, int len) {
    __shared__ float ws[24 * 7];
    for (int i = threadIdx.x; i < C * 7; i += 256) ws[i] = w[i];
    __syncthreads();
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= len) return;
    const float* xb = x + (long)b * C * len;
    int tt[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        int p = t + j - 3;
        tt[j] = p < 0 ? 0 : (p > len - 1 ? len - 1 : p);
    }
    float acc = 0.f;
    for (int c = 0; c < C; ++c) {
        const float* xr = xb + (long)c * len;
#pragma unroll
        for (int j = 0; j < 7; ++j) acc = fmaf(ws[c * 7 + j], xr[tt[j]], acc);
    }
    y[(long)b * len + t] = acc + bias[0];
}

int run_out_conv7(tvc_ctx* ctx, hipStream_t s, const float* x, const float* w_raw, const float* bias, float* y, int B, int C, int len) {
    hipLaunchKernelGGL(out_conv7_kernel, dim3((len + 255) / 256, B), dim3(256), 0, s, x, w_raw, bias, y, C, len);
    return launch_check(ctx, "out_conv7");
}

}  // namespace tvc
