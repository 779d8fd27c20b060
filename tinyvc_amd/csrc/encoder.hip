// Encoder (encoder.py:75-116) and the ConvNeXt-v2 layer shared with SourceNet (convnext.py:7-58).
#include "conv3s.h"
#include "gemm_s2.h"
#include "cnx_s3.h"
#include "small_kernels.h"
#include "tvc_common.h"

namespace tvc {

// conv3s tile of the GEMM launches gemm_s2's preconditions exclude (the ragged 961-row input layer): 128 x 128, two workgroups per CU
constexpr int ENC_MTB = 4, ENC_NWV = 4, ENC_BPC = 2;

// -------------------------------------------------------------------------------------------------
// [depthwise k7 dilated replicate-padded conv] + LayerNorm over channels, per time column.
// One workgroup = 64 consecutive time steps of one utterance; its 16 waves split the channels
// (c = wave, wave+16, ...), lanes run along time (coalesced).  Two-pass moments (mean, then
// centred variance) through a 16x64 LDS exchange; every thread re-reads only values it wrote.
// CPT = channels per thread (C = 16 * CPT): the conv outputs stay in registers across the two moment passes and are
// written once, normalised (the first version wrote them, re-read them twice and wrote again).
template <bool DW, int CPT>
static __global__ __launch_bounds__(1024) void dwconv_ln_kernel(const float* x, float* y,
                                                               const float* __restrict__ dw_w, const float* __restrict__ dw_b,
                                                               const float* __restrict__ g, const float* __restrict__ bta,
                                                               int C, int T, int dil, RagDev rg) {
    constexpr int NW = 16;
    // ragged batch (ragged.h): x / y are [C][T] over the whole batch (T = row stride), utterance blockIdx.y owns columns pre[b] ... + tb[b]
    const int rs = T;
    long ubase = (long)blockIdx.y * C * T;
    if (rg.tb) {
        T = rg.tb[blockIdx.y];
        ubase = rg.pre[blockIdx.y];
        if ((int)blockIdx.x * 64 >= T) return;
    }
    __shared__ float red[NW][64];
    // the per-channel constants - seven taps + bias, gamma, beta - are read by every lane of a wave alike: staged once per workgroup and
    // read back as LDS broadcasts instead of 10 vector loads per channel and thread (a thread's 24 channels: 240 of its 408 loads)
    // (C = 384: 29 -> 25 us; at C = 128 - 8 channels per thread - the staging round costs more than it saves: 14 -> 19 us, so not there)
    constexpr bool STAGE = CPT >= 16;
    __shared__ __attribute__((aligned(16))) float Wl[STAGE ? 16 * CPT * 8 : 8];
    __shared__ float Gl[2][STAGE ? 16 * CPT : 1];
    if (STAGE) {
        for (int i = threadIdx.x; i < C; i += 1024) {
            if (DW) {
#pragma unroll
                for (int j = 0; j < 7; ++j) Wl[i * 8 + j] = dw_w[i * 7 + j];
                Wl[i * 8 + 7] = dw_b[i];
            }
            Gl[0][i] = g[i];
            Gl[1][i] = bta[i];
        }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + lane;
    const bool ok = t < T;
    const int tc = ok ? t : T - 1;
    const float* xb = x + ubase;
    float* yb = y + ubase;

    int tt[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        int q = tc + (j - 3) * dil;
        tt[j] = q < 0 ? 0 : (q >= T ? T - 1 : q);
    }
    float v[CPT];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = wave + i * NW;
        if (DW) {
            const float* xr = xb + (long)c * rs;
            float wj[7], a;
            if (STAGE) {
                const float4 w0 = *reinterpret_cast<const float4*>(Wl + c * 8), w1 = *reinterpret_cast<const float4*>(Wl + c * 8 + 4);
                wj[0] = w0.x; wj[1] = w0.y; wj[2] = w0.z; wj[3] = w0.w; wj[4] = w1.x; wj[5] = w1.y; wj[6] = w1.z;
                a = w1.w;
            } else {
#pragma unroll
                for (int j = 0; j < 7; ++j) wj[j] = dw_w[c * 7 + j];
                a = dw_b[c];
            }
#pragma unroll
            for (int j = 0; j < 7; ++j) a = fmaf(wj[j], xr[tt[j]], a);
            v[i] = a;
        } else {
            v[i] = xb[(long)c * rs + tc];
        }
        sum += v[i];
    }
    red[wave][lane] = sum;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += red[w][lane];
    const float mean = tot / (float)C;
    __syncthreads();
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const float d = v[i] - mean;
        sq = fmaf(d, d, sq);
    }
    red[wave][lane] = sq;
    __syncthreads();
    float tot2 = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot2 += red[w][lane];
    const float var = tot2 / (float)C;
    const float rstd = 1.f / sqrtf(var + 1e-5f);
    if (!ok) return;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = wave + i * NW;
        yb[(long)c * rs + t] = fmaf((v[i] - mean) * rstd, STAGE ? Gl[0][c] : g[c], STAGE ? Gl[1][c] : bta[c]);
    }
}
template <bool DW>
static int dwconv_ln_launch(tvc_ctx* ctx, hipStream_t s, const float* x, float* y, const float* dw_w, const float* dw_b, const float* g,
                            const float* bta, int B, int C, int T, int dil) {
    RagDev rg;
    TVC_CHECK(rag_view(ctx, s, 1, 0, &rg, nullptr));
    const dim3 grid(((ctx->rag ? ctx->rag->Tlong : T) + 63) / 64, ctx->rag ? ctx->rag->B : B), blk(1024);
    if (C == 384) hipLaunchKernelGGL((dwconv_ln_kernel<DW, 24>), grid, blk, 0, s, x, y, dw_w, dw_b, g, bta, C, T, dil, rg);
    else if (C == 128) hipLaunchKernelGGL((dwconv_ln_kernel<DW, 8>), grid, blk, 0, s, x, y, dw_w, dw_b, g, bta, C, T, dil, rg);
    else return fail(ctx, TVC_ERR_ARG, "dwconv_ln: unsupported channel count %d", C);
    return 0;
}

// gx[b][c] = || h[b][c][:] ||_2   (one wavefront per row)
// ragged batch (ragged.h): h is [C2][T] over the whole batch; row r = (utterance r / C2, channel r % C2) sums its own tb[b] columns
static __global__ void grn_norm_kernel(const float* __restrict__ h, float* __restrict__ gx, long rows, int T, RagDev rg, int C2) {
    const int lane = threadIdx.x & 63;
    long wave = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    for (long r = wave; r < rows; r += nwaves) {
        const float* p = h + r * T;
        int n = T;
        if (rg.tb) {
            const int b = (int)(r / C2), c = (int)(r - (long)b * C2);
            p = h + (long)c * T + rg.pre[b];
            n = rg.tb[b];
        }
        float s = 0.f;
        for (int t = lane; t < n; t += 64) s = fmaf(p[t], p[t], s);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) gx[r] = sqrtf(s);
    }
}

// s[b][c] = 1 + gamma[c] * gx[b][c] / (mean_c gx[b][:] + 1e-6)   (one workgroup per utterance):
// GRN(x) = gamma*(x*nx) + beta + x = x*s + beta; the beta term is folded into the next conv's bias.
// hmax[b] = max_c gx[b][c] |s[b][c]| >= |h s| anywhere in the utterance (a row's largest magnitude is at most its L2 norm): the
// |max| slot of the second 1x1's input (block-floating-point guard of the fp16 split), without another pass over h.
static __global__ __launch_bounds__(256) void grn_finalize_kernel(const float* __restrict__ gx, const float* __restrict__ gamma,
                                                                  float* __restrict__ nx, int C2, float* __restrict__ hmax) {
    __shared__ float red[4];
    __shared__ float redm[4];
    const int b = blockIdx.x;
    const float* g = gx + (long)b * C2;
    float s = 0.f;
    for (int c = threadIdx.x; c < C2; c += 256) s += g[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)C2;
    const float den = mean + 1e-6f;
    float mx = 0.f;
    for (int c = threadIdx.x; c < C2; c += 256) {
        const float f = fmaf(gamma[c], g[c] / den, 1.f);
        nx[(long)b * C2 + c] = f;
        mx = fmaxf(mx, g[c] * fabsf(f));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) redm[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) hmax[b] = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
}

int run_layernorm(tvc_ctx* ctx, hipStream_t s, float* x, const float* g, const float* b, int B, int C, int T) {
    TVC_CHECK(dwconv_ln_launch<false>(ctx, s, x, x, nullptr, nullptr, g, b, B, C, T, 1));
    return launch_check(ctx, "layernorm");
}

// gp[b][0][c] = sum of gp[b][t][c] over utterance b's tiles, in cnx2_kernel's own order (long utterances: added up once, not per workgroup)
static __global__ void grn_tiles_kernel(float* __restrict__ gp, int gp_tiles, int K, int T, RagDev rg) {
    const int b = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= K) return;
    const int Tb = rg.tb ? rg.tb[b] : T, n = (Tb + 63) / 64;
    float* g = gp + (long)b * gp_tiles * K + c;
    float s = g[0];
    for (int t = 1; t < n; ++t) s += g[(long)t * K];
    g[0] = s;
}

// The two fused launches of cnx_s3.h; false = the layer is outside their preconditions (run_convnext's five launches take it)
template <int C>
static bool cnx_try(int* rc, tvc_ctx* ctx, hipStream_t s, const ConvNeXtW& w, float* x, float* h, float* gx, int B, int T, float* amax_out) {
    if (w.C != C || !(w.ln_bound < 32768.f) || w.c2.MT6 != 2 * C / 32 || w.c3.MT6 != C / 32 || w.c2.S6 < C / 16 || w.c3.S6 < 2 * C / 16) return false;
    const int NB = ctx->rag ? ctx->rag->B : B, Tl = ctx->rag ? ctx->rag->Tlong : T, rs = T;      // (ragged: T = all frames = the row stride)
    if ((long)rs * 8 * 4 >= (1L << 31)) return false;                                             // 32-bit lane offsets
    constexpr int KP = C == 384 ? 2 : 1;
    static int ncu_dev[64] = {};
    int& ncu = ncu_dev[ctx->device & 63];
    if (!ncu) {
        hipDeviceProp_t prop;
        hipError_t e = hipGetDeviceProperties(&prop, ctx->device);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)cnx1_kernel<C, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, Cnx1<C, 1>::LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)cnx1_kernel<C, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, Cnx1<C, 2>::LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)cnx2_kernel<C, KP>, hipFuncAttributeMaxDynamicSharedMemorySize, Cnx2<C, KP>::LDS_BYTES);
        if (e != hipSuccess) {
            *rc = fail(ctx, TVC_ERR_HIP, "cnx setup: %s", hipGetErrorString(e));
            return true;
        }
        ncu = prop.multiProcessorCount;
    }
    CnxArgs a{};
    a.x = x;
    a.h = h;
    a.T = T;
    a.rs = rs;
    if ((*rc = rag_view(ctx, s, 1, 0, &a.rg, nullptr)) != 0) return true;
    // rows of the weight image per workgroup: the whole image while the column tiles alone cover the chip, else split over blockIdx.z
    auto split = [&](int MT, int tiles) {
        int d = 1;
        for (int c = 1; c <= MT; ++c)
            if (MT % c == 0 && (long)tiles * c <= ncu) d = c;
        return d;
    };
    {
        a.A6 = reinterpret_cast<const uint4*>(w.c2.A6);
        a.wsc = w.c2.wscale;
        a.bias = w.c2.bias;
        a.MT = w.c2.MT6;
        a.dw_w = w.dw_w;
        a.dw_b = w.dw_b;
        a.ln_g = w.ln_g;
        a.ln_b = w.ln_b;
        a.dil = w.dilation;
        const int nt = Tl <= 32 ? 1 : 2, tx = (Tl + 32 * nt - 1) / (32 * nt);
        a.gp = gx;
        a.gp_tiles = tx;
        const int d = split(a.MT, tx * NB);
        a.mt_per_wg = a.MT / d;
        const dim3 grid((unsigned)tx, (unsigned)NB, (unsigned)d);
        constexpr int nthr = Cnx1<C, 2>::NTHR, lds1 = Cnx1<C, 1>::LDS_BYTES, lds2 = Cnx1<C, 2>::LDS_BYTES;
        if (nt == 1) hipLaunchKernelGGL((cnx1_kernel<C, 1>), grid, dim3(nthr), lds1, s, a);
        else hipLaunchKernelGGL((cnx1_kernel<C, 2>), grid, dim3(nthr), lds2, s, a);
    }
    a.gp_sum = a.gp_tiles > CNX_GP_INLINE;
    if (a.gp_sum) hipLaunchKernelGGL(grn_tiles_kernel, dim3((2 * C + 255) / 256, (unsigned)NB), dim3(256), 0, s, gx, a.gp_tiles, 2 * C, T, a.rg);
    {
        a.A6 = reinterpret_cast<const uint4*>(w.c3.A6);
        a.wsc = w.c3.wscale;
        a.bias = w.c3_bias_grn;
        a.MT = w.c3.MT6;
        a.grn_g = w.grn_g;
        a.amax_y = amax_out;
        const int tx = (Tl + 63) / 64;
        const int d = split(a.MT, tx * NB);
        a.mt_per_wg = a.MT / d;
        constexpr int nthr = Cnx2<C, KP>::NTHR, lds = Cnx2<C, KP>::LDS_BYTES;
        hipLaunchKernelGGL((cnx2_kernel<C, KP>), dim3((unsigned)tx, (unsigned)NB, (unsigned)d), dim3(nthr), lds, s, a);
    }
    *rc = launch_check(ctx, "convnext (fused)");
    return true;
}

// amax_out: optional per-utterance |max| slot of the layer's output (zeroed by the caller), for the contraction that reads it next
int run_convnext(tvc_ctx* ctx, hipStream_t s, Ws& ws, bool dry, const ConvNeXtW& w, float* x, int B, int T, float* amax_out) {
    const int C = w.C, C2 = 2 * w.C, ncols = B * T;
    const int NB = ctx->rag ? ctx->rag->B : B;      // utterances (a ragged batch runs as B = 1, T = all its frames: ragged.h)
    size_t mk = ws.mark();
    float* y = ws.get<float>((size_t)B * C * T);
    float* h = ws.get<float>((size_t)B * C2 * T);
    const int Tlong = ctx->rag ? ctx->rag->Tlong : T;
    float* gx = ws.get<float>((size_t)NB * C2 * ((Tlong + 63) / 64));      // row norms; the fused launches keep one sum of squares per 64-column tile
    float* nx = ws.get<float>((size_t)NB * C2);
    float* ymax = ws.get<float>((size_t)NB);      // |max| slots of the two 1x1s' inputs (block-floating-point guard, conv3s.h)
    float* hmax = ws.get<float>((size_t)NB);
    ws.release(mk);
    if (dry) return 0;
#ifndef TVC_CNX_OLD
    {
        int rc = 0;
        if (C == 384 ? cnx_try<384>(&rc, ctx, s, w, x, h, gx, B, T, amax_out) : (C == 128 && cnx_try<128>(&rc, ctx, s, w, x, h, gx, B, T, amax_out))) return rc;
    }
#endif
    {
        TVC_CHECK(dwconv_ln_launch<true>(ctx, s, x, y, w.dw_w, w.dw_b, w.ln_g, w.ln_b, B, C, T, w.dilation));
    }
    {
        // the LayerNorm output is bounded by the layer's own gamma / beta whatever the data: no slot unless that bound leaves fp16's range
        const float* ym = nullptr;
        if (!(w.ln_bound < 32768.f)) {
            TVC_HIP(ctx, hipMemsetAsync(ymax, 0, (size_t)NB * sizeof(float), s));
            TVC_CHECK(run_amax_rows(ctx, s, y, B, C, T, ymax));
            ym = ymax;
        }
        EpiBias<ACT_GELU, false> ep{h, w.c2.bias, nullptr, C2, T, ncols, (long)C2 * T, 0};
        int rc = 0;
        if (!gemm_s2_try(&rc, ctx, s, w.c2, y, B, C, T, 0, ep, ym)) rc = gemm_s_launch<ENC_MTB, ENC_NWV, ENC_BPC>(ctx, s, w.c2, y, B, C, T, 0, ep, ym);
        TVC_CHECK(rc);
    }
    {
        RagDev rg;
        TVC_CHECK(rag_view(ctx, s, 1, 0, &rg, nullptr));
        hipLaunchKernelGGL(grn_norm_kernel, dim3(grid_for((long)NB * C2 * 64)), dim3(256), 0, s, h, gx, (long)NB * C2, T, rg, C2);
        hipLaunchKernelGGL(grn_finalize_kernel, dim3(NB), dim3(256), 0, s, gx, w.grn_g, nx, C2, hmax);
    }
    {
        EpiBias<ACT_NONE, true> ep{x, w.c3_bias_grn, x, C, T, ncols, (long)C * T, (long)C * T};
        int rc = 0;
        if (!gemm_s2_try<EpiBias<ACT_NONE, true>, true>(&rc, ctx, s, w.c3, h, B, C2, T, 0, ep, hmax, nx))
            rc = gemm_s_launch<ENC_MTB, ENC_NWV, ENC_BPC, EpiBias<ACT_NONE, true>, true>(ctx, s, w.c3, h, B, C2, T, 0, ep, hmax, nx);
        TVC_CHECK(rc);
    }
    if (amax_out) TVC_CHECK(run_amax_rows(ctx, s, x, B, C, T, amax_out));
    return launch_check(ctx, "convnext");
}

// stacked input 1x1 (961 -> 384 | 128): rows < M0 go to y0, the rest to y1
struct EpiSplit {
    static constexpr bool kIgemm = true;
    float* y0;
    float* y1;
    const float* bias;
    int M0, M1, T, ncols;
    __device__ __forceinline__ void store(int n, int m, const float v[4]) const {
        if (n >= ncols) return;
        int b = n / T, t = n - b * T;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int mm = m + r;
            float o = v[r] + bias[mm < M0 + M1 ? mm : 0];
            if (mm < M0)
                y0[((long)b * M0 + mm) * T + t] = o;
            else if (mm < M0 + M1)
                y1[((long)b * M1 + mm - M0) * T + t] = o;
        }
    }
};

// PitchEstimator.decode (encoder.py:61-67): top-4 logits (ties -> lower class id), softmax over
// them, expectation of the class frequencies, <= 20 Hz -> 0.
// One workgroup = 64 consecutive columns: each of the 16 waves scans 32 of the 512 classes with
// lanes along time (coalesced), keeps a per-lane top-4, and the partial lists merge through LDS (the order
// of the merge does not matter: `better` is a total order, ties go to the lower class id).
struct PTop4 {
    float v[4];
    int i[4];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = -INFINITY; i[j] = 0x7fffffff; }
    }
    __device__ __forceinline__ static bool better(float a, int ia, float b, int ib) { return a > b || (a == b && ia < ib); }
    __device__ __forceinline__ void insert(float x, int ix) {
        if (!better(x, ix, v[3], i[3])) return;
        if (better(x, ix, v[0], i[0])) { v[3] = v[2]; i[3] = i[2]; v[2] = v[1]; i[2] = i[1]; v[1] = v[0]; i[1] = i[0]; v[0] = x; i[0] = ix; }
        else if (better(x, ix, v[1], i[1])) { v[3] = v[2]; i[3] = i[2]; v[2] = v[1]; i[2] = i[1]; v[1] = x; i[1] = ix; }
        else if (better(x, ix, v[2], i[2])) { v[3] = v[2]; i[3] = i[2]; v[2] = x; i[2] = ix; }
        else { v[3] = x; i[3] = ix; }
    }
};

constexpr int kPdWaves = 16;
static __global__ __launch_bounds__(kPdWaves * 64) void pitch_decode_kernel(const float* __restrict__ logits, const float* __restrict__ freq,
                                                                  float* __restrict__ f0, int B, int T, float* __restrict__ f0s, float shift) {
    __shared__ float sv[kPdWaves][4][64];     // [wave][entry][lane]: lanes along the fastest axis (the [lane][entry] order was a 4-way bank conflict)
    __shared__ int si[kPdWaves][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long ncols = (long)B * T;
    const long n = blockIdx.x * 64L + lane;
    const bool ok = n < ncols;
    const long nn = ok ? n : ncols - 1;
    const int b = (int)(nn / T), t = (int)(nn - (long)b * T);
    const float* p = logits + (long)b * kPitchClasses * T + t;
    PTop4 top;
    top.init();
    // a wave's 32 classes are requested together, then inserted in ascending order (the same list as the load-insert-load chain this
    // replaces, which was 32 memory latencies long: 23 us of a streaming block for 896 columns)
    constexpr int CPW = kPitchClasses / kPdWaves;
    float xv[CPW];
#pragma unroll
    for (int j = 0; j < CPW; ++j) xv[j] = p[(long)(wave * CPW + j) * T];
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
        const float x = xv[j];
        top.insert(x != x ? INFINITY : x, wave * CPW + j);   // torch.topk orders NaN first; also keeps the list sentinel out of freq[]
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { sv[wave][e][lane] = top.v[e]; si[wave][e][lane] = top.i[e]; }
    __syncthreads();
    if (wave != 0 || !ok) return;
    for (int w = 1; w < kPdWaves; ++w)
#pragma unroll
        for (int e = 0; e < 4; ++e) top.insert(sv[w][e][lane], si[w][e][lane]);
#pragma unroll
    for (int e = 0; e < 4; ++e) top.i[e] = (unsigned)top.i[e] < (unsigned)kPitchClasses ? top.i[e] : 0;
    const float v0 = top.v[0];
    // exp evaluated in fp64 and rounded once: agrees with ATen's Sleef expf bit for bit on ~99 % of inputs (see shift_kernel)
    float e0 = 1.f, e1 = (float)exp((double)(top.v[1] - v0)), e2 = (float)exp((double)(top.v[2] - v0)), e3 = (float)exp((double)(top.v[3] - v0));
    float den = ((e0 + e1) + e2) + e3;
    float acc = __fmul_rn(e0 / den, freq[top.i[0]]);
    acc = __fadd_rn(acc, __fmul_rn(e1 / den, freq[top.i[1]]));
    acc = __fadd_rn(acc, __fmul_rn(e2 / den, freq[top.i[2]]));
    acc = __fadd_rn(acc, __fmul_rn(e3 / den, freq[top.i[3]]));
    const float fv = acc <= 20.f ? 0.f : acc;
    f0[n] = fv;
    if (f0s) f0s[n] = shift_frequency_one(fv, shift);      // the caller's shift_frequency(f0, shift), in the same launch
}

int run_pitch_decode(tvc_ctx* ctx, hipStream_t s, const float* logits, float* f0, int B, int T) {
    if (!ctx->pitch_freq) return fail(ctx, TVC_ERR_STATE, "pitch table not uploaded (tvc_set_pitch_table + tvc_finalize_weights)");
    const long ncols = (long)B * T;
    hipLaunchKernelGGL(pitch_decode_kernel, dim3((unsigned)((ncols + 63) / 64)), dim3(kPdWaves * 64), 0, s, logits, ctx->pitch_freq, f0, B, T, (float*)nullptr, 0.f);
    return launch_check(ctx, "pitch_decode");
}

int run_encoder(tvc_ctx* ctx, hipStream_t s, Ws& ws, bool dry, const float* spec, float* ssl, float* f0,
                float* logits, int B, int T, const float* spec_bound, float* zeroed_slots, float* f0_shifted, float shift) {
    const int ncols = B * T;
    float* xs = ws.get<float>((size_t)B * kSslCh * T);
    float* xp = ws.get<float>((size_t)B * kPitchCh * T);
    float* lg = logits ? logits : ws.get<float>((size_t)B * kPitchClasses * T);
    // |max| slots (block-floating-point guard of the fp16 split, conv3s.h): the spectrogram and the two residual streams the output
    // projections read; the ConvNeXt layers keep their own (run_convnext)
    const int NB = ctx->rag ? ctx->rag->B : B;      // utterances (ragged batch: B = 1, T = all frames)
    float* slots = zeroed_slots ? zeroed_slots : ws.get<float>((size_t)3 * NB);
    float *spec_max = slots, *xs_max = slots + NB, *xp_max = slots + 2 * NB;
    if (!dry) {
        if (!zeroed_slots) TVC_HIP(ctx, hipMemsetAsync(slots, 0, (size_t)3 * NB * sizeof(float), s));
        if (spec_bound) spec_max = const_cast<float*>(spec_bound);      // the caller's bound IS the slot: no pass over the 961 x T tensor
        else TVC_CHECK(run_amax_rows(ctx, s, spec, B, kBins, T, spec_max));
        EpiSplit ep{xs, xp, ctx->enc_in.bias, kSslCh, kPitchCh, T, ncols};
        // 961 input rows: the last slab is clamped to row 960 (zero weights beyond)
        TVC_CHECK((gemm_s_launch_ragged<ENC_MTB, ENC_NWV, ENC_BPC>(ctx, s, ctx->enc_in, spec, B, kBins, T, (long)kBins * T, ep, spec_max)));
        TVC_CHECK(run_layernorm(ctx, s, xs, ctx->ssl_ln_g, ctx->ssl_ln_b, B, kSslCh, T));
        TVC_CHECK(run_layernorm(ctx, s, xp, ctx->pit_ln_g, ctx->pit_ln_b, B, kPitchCh, T));
    }
    // The pitch estimator (4 narrow ConvNeXt layers + logits + decode) and the SSL chain are independent after the
    // stacked input 1x1: fork the pitch chain onto the context's side stream and join before returning, so its
    // small launches fill the gaps of the SSL chain instead of extending the critical path.  Each chain gets its own
    // scratch block (the per-layer mark/release scratch would alias otherwise).
    const int gtiles = ((ctx->rag ? ctx->rag->Tlong : T) + 63) / 64;      // run_convnext: y, h, the GRN tile sums, nx, two slots
    const size_t ssl_scratch = ((size_t)B * kSslCh * T * 3 + (size_t)NB * kSslCh * 2 * (gtiles + 1)) * sizeof(float) + (size_t)NB * 8 + 4096;
    char* ssl_blk = ws.get<char>(ssl_scratch);
    Ws wssl(ssl_blk, ssl_scratch, dry);
    hipStream_t sp = s;
    bool fork = !dry && ctx->side;
    if (fork) {
        // Not while the stream is being captured: a replayed graph with the second branch is slower than the one chain (32-stream block
        // 1.90 -> 1.71 ms; one 4 s utterance 1.36 ... 1.59 ms from box to box with the branch, 1.43 without)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) fork = false;
    }
    if (fork) {
        TVC_HIP(ctx, hipEventRecord(ctx->ev_fork, s));
        TVC_HIP(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
        sp = ctx->side;
    }
    for (int i = 0; i < 4; ++i) TVC_CHECK(run_convnext(ctx, sp, ws, dry, ctx->pit_mid[i], xp, B, T, i == 3 ? xp_max : nullptr));      // the last layer publishes the |max| slot of its output
    if (!dry) {
        EpiBias<ACT_NONE, false> ep{lg, ctx->pit_out.bias, nullptr, kPitchClasses, T, ncols, (long)kPitchClasses * T, 0};
        int rc = 0;
        if (!gemm_s2_try(&rc, ctx, sp, ctx->pit_out, xp, B, kPitchCh, T, 0, ep, xp_max)) rc = gemm_s_launch<ENC_MTB, ENC_NWV, ENC_BPC>(ctx, sp, ctx->pit_out, xp, B, kPitchCh, T, 0, ep, xp_max);
        TVC_CHECK(rc);
        hipLaunchKernelGGL(pitch_decode_kernel, dim3((ncols + 63) / 64), dim3(kPdWaves * 64), 0, sp, lg, ctx->pitch_freq, f0, B, T, f0_shifted, shift);
    }
    if (fork) TVC_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->side));
    for (int i = 0; i < 6; ++i) TVC_CHECK(run_convnext(ctx, s, wssl, dry, ctx->ssl_mid[i], xs, B, T, i == 5 ? xs_max : nullptr));
    if (!wssl.ok()) return fail(ctx, TVC_ERR_WORKSPACE, "encoder: SSL scratch block too small");
    if (dry) return 0;
    {
        EpiBias<ACT_NONE, false> ep{ssl, ctx->ssl_out.bias, nullptr, kSslDim, T, ncols, (long)kSslDim * T, 0};
        int rc = 0;
        if (!gemm_s2_try(&rc, ctx, s, ctx->ssl_out, xs, B, kSslCh, T, 0, ep, xs_max)) rc = gemm_s_launch<ENC_MTB, ENC_NWV, ENC_BPC>(ctx, s, ctx->ssl_out, xs, B, kSslCh, T, 0, ep, xs_max);
        TVC_CHECK(rc);
    }
    if (fork) TVC_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_join, 0));
    return launch_check(ctx, "encoder");
}

}  // namespace tvc

#ifdef S_TRACE
extern "C" int tvc_debug_trace_enc(unsigned long long* host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(tvc::g_trace), sizeof(tvc::g_trace)) == hipSuccess ? 0 : -1;
}
#endif
