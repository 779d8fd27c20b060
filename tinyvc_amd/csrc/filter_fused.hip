// Block-fused FilterNet kernels for the narrow (24-channel, full-rate) level — the level that carries
// 54 % of FilterNet's layer-boundary bytes.
//
// One Upsample block (decoder.py:173-190) = two launches, each a chain that never leaves the CU:
//   half A:  x_up = interp(x, xf) -> lrelu -> c1(d1) -> lrelu -> c2(d3) -> FiLM1(cond) -> + x_up     => x1
//   half B:  x1 -> lrelu -> c3(d9) -> lrelu -> c4(d27) -> FiLM2(cond) -> + x1 -> c5 (1x1)             => out
// A workgroup owns W output samples of one utterance.  The input tile (with the chain's halo) and the
// intermediate activation live in LDS as [C][cols]; the convs are implicit GEMMs on
// v_mfma_f32_32x32x2_f32 with both operands read from LDS (weights: per-stage image At[tap*C+ci][32];
// activations: rows ci, columns shifted by (tap-1)*dilation and clamped to the utterance = replicate
// padding of *that layer's* input).  FiLM's two 1x1 convs run on the same MFMA tiles with the cond
// fragments fetched straight from HBM (each cond element is used exactly once per block), so scale,
// shift and the conv accumulator share one lane layout and combine in registers.
// HBM traffic per block: x (or x1) tile + cond tile in, one tile out — the block-fused lower bound.
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

template <int C_, int W_, int D1_, int D2_, bool SECOND_>
struct UpHalfCfg {
    static constexpr int C = C_, W = W_, D1 = D1_, D2 = D2_;
    static constexpr bool SECOND = SECOND_;
    static constexpr int H = D1 + D2;                      // halo of the two-conv chain
    static constexpr int XW = W + 2 * H;                   // input tile columns
    static constexpr int HW = W + 2 * D2;                  // intermediate columns needed
    static constexpr int HWr = (HW + 31) / 32 * 32;        // rounded to MFMA n-tiles
    static constexpr int XWa = (XW + 3) / 4 * 4;
    static constexpr int WCONV = 3 * C * 32;               // one k3 conv image
    static constexpr int W1x1 = C * 32;
    static constexpr int WBUF = WCONV + 2 * W1x1;          // conv + FiLM scale + FiLM shift
    static constexpr int LDS_FLOATS = C * XWa + C * HWr + WBUF;
    static constexpr int NWAVES = 8;
};

struct UpHalfArgs {
    const float* x;      // half A: low-rate input [B][C][len/xf]; half B: x1 [B][C][len]
    const float* cond;   // [B][C][len]
    float* out;          // half A: x1 [B][C][len]; half B: [B][cout][len]
    const float* wa;     // first conv, tap-major image [3C][32]
    const float* ba;
    const float* wb;     // second conv
    const float* bb;
    const float* wsc;    // FiLM to_scale [C][32]
    const float* bsc;
    const float* wsh;    // FiLM to_shift
    const float* bsh;
    const float* w5;     // half B: c5 [C][32]
    const float* b5;
    int len, xf, cout, tiles_per_utt;
    float interp_scale;
};

template <class CF>
__global__ __launch_bounds__(512) void up_half_kernel(UpHalfArgs a) {
    constexpr int C = CF::C, W = CF::W, D1 = CF::D1, D2 = CF::D2, H = CF::H;
    constexpr int XW = CF::XWa, HWr = CF::HWr, NW = CF::NWAVES;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                 // [C][XW]   input tile (raw), later x2 in place (half B)
    float* Hs = Xs + C * XW;          // [C][HWr]  lrelu(first conv)
    float* Wc = Hs + C * HWr;         // conv weights image
    float* Wsc = Wc + CF::WCONV;      // FiLM scale / later c5
    float* Wsh = Wsc + CF::W1x1;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.x / a.tiles_per_utt;
    const int t0 = (blockIdx.x - b * a.tiles_per_utt) * W;
    const int len = a.len;

    // ---- S0: stage the input tile (positions t0-H .. t0+W+H, clamped into the utterance) -----------
    if (CF::SECOND) {
        const float* xb = a.x + (long)b * C * len;
        for (int i = tid; i < C * XW; i += 512) {
            int ci = i / XW, c = i - ci * XW;
            int p = t0 - H + c;
            p = p < 0 ? 0 : (p > len - 1 ? len - 1 : p);
            Xs[i] = xb[(long)ci * len + p];
        }
    } else {
        const int lin = len / a.xf;
        const float* xb = a.x + (long)b * C * lin;
        for (int i = tid; i < C * XW; i += 512) {
            int ci = i / XW, c = i - ci * XW;
            int p = t0 - H + c;
            p = p < 0 ? 0 : (p > len - 1 ? len - 1 : p);
            Lerp lc = lerp_coord(p, a.interp_scale, lin);
            const float* xr = xb + (long)ci * lin;
            Xs[i] = lerp_eval(lc, xr[lc.i0], xr[lc.i1]);
        }
    }
    copy_to_lds(Wc, a.wa, CF::WCONV, tid, 512);
    __syncthreads();

    // ---- S1: Hs = lrelu(conv_a(lrelu(x)) + ba) over the extended range -----------------------------
    {
        const int lo = H - t0 > 0 ? H - t0 : 0;
        const int hi = (len - 1 - t0 + H) < (CF::XW - 1) ? (len - 1 - t0 + H) : (CF::XW - 1);
        for (int nt = wave; nt < HWr / 32; nt += NW) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            conv_mma_tile<C, 3, true>(acc, Wc, Xs, XW, nt * 32 + l31 + D1, D1, lo, hi, l31, lh);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < C) Hs[m * HWr + nt * 32 + l31] = lrelu01(acc[r] + a.ba[m]);
            }
        }
    }
    __syncthreads();
    copy_to_lds(Wc, a.wb, CF::WCONV, tid, 512);
    copy_to_lds(Wsc, a.wsc, CF::W1x1, tid, 512);
    copy_to_lds(Wsh, a.wsh, CF::W1x1, tid, 512);
    __syncthreads();

    // ---- S2: (conv_b(Hs) + bb) * scale + shift + x --------------------------------------------------
    {
        const int lo = D2 - t0 > 0 ? D2 - t0 : 0;
        const int hi = (len - 1 - t0 + D2) < (CF::HW - 1) ? (len - 1 - t0 + D2) : (CF::HW - 1);
        const float* cb = a.cond + (long)b * C * len;
        for (int nt = wave; nt < W / 32; nt += NW) {
            const int n = nt * 32 + l31;
            const int t = t0 + n;
            const int tc = t < len ? t : len - 1;
            float cf[C / 2];
#pragma unroll
            for (int s = 0; s < C / 2; ++s) cf[s] = cb[(long)(2 * s + lh) * len + tc];
            f32x16 acc, asc, ash;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = asc[r] = ash[r] = 0.f;
            conv_mma_tile<C, 3, false>(acc, Wc, Hs, HWr, n + D2, D2, lo, hi, l31, lh);
#pragma unroll
            for (int s = 0; s < C / 2; ++s) {
                float ws = Wsc[(2 * s + lh) * 32 + l31];
                float wh = Wsh[(2 * s + lh) * 32 + l31];
                asc = __builtin_amdgcn_mfma_f32_32x32x2f32(ws, cf[s], asc, 0, 0, 0);
                ash = __builtin_amdgcn_mfma_f32_32x32x2f32(wh, cf[s], ash, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < C) {
                    float hval = acc[r] + a.bb[m];
                    float sc = asc[r] + a.bsc[m];
                    float sh = ash[r] + a.bsh[m];
                    float res = Xs[m * XW + n + H];
                    float v = __fadd_rn(__fadd_rn(__fmul_rn(hval, sc), sh), res);
                    if (CF::SECOND)
                        Xs[m * XW + n + H] = v;                       // x2 stays on chip for c5
                    else if (t < len)
                        a.out[((long)b * C + m) * len + t] = v;       // x1
                }
            }
        }
    }
    if (!CF::SECOND) return;
    __syncthreads();
    copy_to_lds(Wc, a.w5, CF::W1x1, tid, 512);
    __syncthreads();

    // ---- S3 (half B): out = c5(x2) + b5 --------------------------------------------------------------
    for (int nt = wave; nt < W / 32; nt += NW) {
        const int n = nt * 32 + l31;
        const int t = t0 + n;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        conv_mma_tile<C, 1, false>(acc, Wc, Xs, XW, n + H, 0, 0, CF::XW - 1, l31, lh);
        if (t < len) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < a.cout) a.out[((long)b * a.cout + m) * len + t] = acc[r] + a.b5[m];
            }
        }
    }
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

template <class CF>
static int launch_up_half(tvc_ctx* ctx, hipStream_t s, const UpHalfArgs& a, int B) {
    static bool attr_set = false;
    const size_t lds = (size_t)CF::LDS_FLOATS * sizeof(float);
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)up_half_kernel<CF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(ctx, TVC_ERR_HIP, "hipFuncSetAttribute(up_half): %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((up_half_kernel<CF>), dim3((unsigned)(B * a.tiles_per_utt)), dim3(512), lds, s, a);
    return launch_check(ctx, "up_half");
}

// Upsample block with cin == 24: x [B][24][len/f], cond [B][24][len] -> out [B][cout][len]; x1 is scratch [B][24][len]
int run_up24_fused(tvc_ctx* ctx, hipStream_t s, const UpW& u, const float* x, const float* cond, float* x1, float* out,
                   int B, int len) {
    using CA = UpHalfCfg<24, 256, 1, 3, false>;
    using CB = UpHalfCfg<24, 256, 9, 27, true>;
    UpHalfArgs a{};
    a.len = len;
    a.xf = u.factor;
    a.cout = u.cout;
    a.tiles_per_utt = (len + 255) / 256;
    a.interp_scale = (float)(1.0 / (double)u.factor);
    a.cond = cond;
    a.x = x;
    a.out = x1;
    a.wa = u.c1.At_tap; a.ba = u.c1.bias;
    a.wb = u.c2.At_tap; a.bb = u.c2.bias;
    a.wsc = u.sc1.At; a.bsc = u.sc1.bias;
    a.wsh = u.sh1.At; a.bsh = u.sh1.bias;
    TVC_CHECK(launch_up_half<CA>(ctx, s, a, B));
    a.x = x1;
    a.out = out;
    a.wa = u.c3.At_tap; a.ba = u.c3.bias;
    a.wb = u.c4.At_tap; a.bb = u.c4.bias;
    a.wsc = u.sc2.At; a.bsc = u.sc2.bias;
    a.wsh = u.sh2.At; a.bsh = u.sh2.bias;
    a.w5 = u.c5.At; a.b5 = u.c5.bias;
    return launch_up_half<CB>(ctx, s, a, B);
}

}  // namespace tvc
