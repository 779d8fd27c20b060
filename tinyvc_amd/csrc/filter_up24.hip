// Persistent block-fused kernels for FilterNet's 24-channel full-rate Upsample block (ups[4],
// decoder.py:173-190) and the output layer (decoder.py:220,233) — the level that carries 54 % of the
// stack's layer-boundary bytes.
//
//   half A:  x_up = interp(x, x5) -> lrelu -> c1(d1) -> lrelu -> c2(d3) -> FiLM1(cond) -> + x_up      => x1
//   half B:  x1 -> lrelu -> c3(d9) -> lrelu -> c4(d27) -> FiLM2(cond) -> + x1 -> c5 -> output k7       => wave
//
// One persistent workgroup per CU (16 waves) walks tiles of W output samples:
//   * every stage's weights are staged into LDS ONCE per workgroup (bank-conflict-free paired-row image);
//   * the next tile's input is fetched into registers while the current tile computes and dropped into
//     the other half of a double-buffered LDS input tile at the end of the iteration, so HBM latency is
//     off the critical path; FiLM's cond fragments are fetched at the top of the iteration;
//   * convs are implicit GEMMs on v_mfma_f32_16x16x4_f32 (A lane l -> A[i=l&15][k=l>>4],
//     B lane l -> B[k=l>>4][j=l&15], C reg r -> C[row=4*(l>>4)+r][col=l&15]); 16-column tiles spread
//     evenly over the 4 SIMDs; both operands come from LDS, activations clamp their column to the
//     utterance (= replicate padding of that layer's input);
//   * FiLM scale/shift run on the same tiles so (conv, scale, shift) combine in registers.
// HBM traffic per tile: input tile (+halo) and cond tile in, one tile out.
#include "igemm.h"
#include "small_kernels.h"
#include "tvc_common.h"

namespace tvc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef UP24_EXP
#define UP24_EXP 0   // ablation switch for timing experiments (0 = production): 1 no S4, 2 no prefetch, 3 no S1 MFMA,
#endif               // 4 no S2 MFMA, 5 no barriers (wrong results for != 0)
#ifndef UP24_NT
#define UP24_NT 512   // threads per persistent workgroup (8 waves: 256-VGPR budget, no spills)
#endif

__device__ __forceinline__ float lrelu_p1(float v) { return v > 0.f ? v : 0.1f * v; }

// LDS weight image of At[k][32] (k rows, 32 padded output channels): rows are stored in pairs so that the
// 16-lane groups of a ds_read_b32 (rows k, k+1) hit disjoint banks:
//   img[(k>>1)*64 + mt*32 + (k&1)*16 + c] = At[k][mt*16 + c]
__device__ __forceinline__ int wimg(int k, int mt, int c) { return (k >> 1) * 64 + mt * 32 + (k & 1) * 16 + c; }

__device__ __forceinline__ void stage_weights(float* img, const float* __restrict__ At, int rows, int tid, int nthr) {
    for (int i = tid; i < rows * 32; i += nthr) {
        int k = i >> 5, m = i & 31;
        img[wimg(k, m >> 4, m & 15)] = At[i];
    }
}

// One 32(m) x 16(n) output tile of a TAPS-tap conv: acc[mt] += sum_{tap, ci} W[tap*CIN+ci][mt*16+i] *
// act(Xs[ci][clamp(colc + (tap - TAPS/2)*dil)]).  All TAPS*CIN/4 k-steps' operands (2 weight + 1 activation
// value each) are fetched from LDS first and the MFMAs then issue back to back: hipcc otherwise emits
// ds_read -> s_waitcnt lgkmcnt(0) -> 2 MFMA per k-step and the LDS latency sits on every step.
template <int CIN, int TAPS>
struct ConvOps {
    static constexpr int NS = TAPS * CIN / 4;
    float a0[NS], a1[NS], b[NS];
};

template <int CIN, int TAPS>
__device__ __forceinline__ void conv_load16(ConvOps<CIN, TAPS>& o, const float* Wi, const float* Xs, int xs, int colc,
                                            int dil, int lo, int hi, int l15, int lq) {
    const float* wl = Wi + (lq >> 1) * 64 + (lq & 1) * 16 + l15;     // this lane's row offset within a 4-row group
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
        int c = colc + (tap - TAPS / 2) * dil;
        c = c < lo ? lo : (c > hi ? hi : c);
        const float* xp = Xs + lq * xs + c;
#pragma unroll
        for (int ci = 0; ci < CIN; ci += 4) {
            const int s = (tap * CIN + ci) / 4;
            o.a0[s] = wl[(tap * CIN + ci) / 2 * 64];
            o.a1[s] = wl[(tap * CIN + ci) / 2 * 64 + 32];
            o.b[s] = xp[ci * xs];
        }
    }
}

template <int CIN, int TAPS, bool LRELU>
__device__ __forceinline__ void conv_mma16(f32x4 (&acc)[2], const ConvOps<CIN, TAPS>& o) {
#if !defined(UP24_NOPRELOAD) || !UP24_NOPRELOAD
    __builtin_amdgcn_sched_barrier(0);   // keep the operand fetch block above, the MFMA block below
#endif
#pragma unroll
    for (int s = 0; s < ConvOps<CIN, TAPS>::NS; ++s) {
        float b = o.b[s];
        if (LRELU) b = fmaxf(b, 0.1f * b);           // leaky_relu(0.1): max(v, 0.1 v)
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a0[s], b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a1[s], b, acc[1], 0, 0, 0);
    }
}

template <int W_, int D1_, int D2_, bool SECOND_, int E_, int NT_>
struct Up24Cfg {
    static constexpr int C = 24, W = W_, D1 = D1_, D2 = D2_, E = E_;
    static constexpr bool SECOND = SECOND_;
    static constexpr int H = D1 + D2;
    static constexpr int W2 = W + 2 * E;                    // columns produced by the 2nd conv / c5
    static constexpr int W2r = (W2 + 15) / 16 * 16;
    static constexpr int XW = W2 + 2 * H;                   // input tile columns
    static constexpr int HW = W2 + 2 * D2;                  // 1st-conv columns needed
    static constexpr int HWr = (HW + 15) / 16 * 16;
    // row strides = 16 (mod 32): the two rows read by one 32-lane ds_read group sit on disjoint banks
    static constexpr int XS = ((XW > HWr ? XW : HWr) + 15) / 32 * 32 + 16;
    static constexpr int NT = NT_, NWAVES = NT_ / 64;
    static constexpr int NS2 = (W2r / 16 + NWAVES - 1) / NWAVES;   // 2nd-conv tiles per wave
    static constexpr int XELEMS = C * XW;
    static constexpr int XPER = (XELEMS + NT - 1) / NT;
    // weight image offsets (floats)
    static constexpr int O_WA = 0, O_WB = O_WA + 72 * 32, O_SC = O_WB + 72 * 32, O_SH = O_SC + 24 * 32,
                         O_W5 = O_SH + 24 * 32, O_W7 = O_W5 + 24 * 32, WFLOATS = O_W7 + 24 * 8;
    static constexpr int LDS_FLOATS = C * XS + C * XS + WFLOATS + 5 * 32;
    static_assert(XS >= XW && XS >= HWr && XS >= W2r, "row stride");
};

struct Up24Args {
    const float* x;      // half A: low-rate input [B][24][len/xf]; half B: x1 [B][24][len]
    const float* cond;   // [B][24][len]
    float* out;          // half A: x1 [B][24][len]; half B: waveform [B][len]
    const float* wa;     // first conv, tap-major At[72][32]
    const float* ba;
    const float* wb;     // second conv
    const float* bb;
    const float* wsc;    // FiLM to_scale At[24][32]
    const float* bsc;
    const float* wsh;
    const float* bsh;
    const float* w5;     // half B: c5 At[24][32]
    const float* b5;
    const float* w7;     // output_layer weight raw [24][7], bias [1]
    const float* b7;
    int len, xf, tiles_per_utt, ntiles;
    float interp_scale;
};

template <class CF>
__global__ __launch_bounds__(CF::NT) void up24_kernel(Up24Args a) {
    constexpr int C = CF::C, W = CF::W, D1 = CF::D1, D2 = CF::D2, H = CF::H, E = CF::E, XS = CF::XS, NT = CF::NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xbuf = smem;                        // [C][XS] input tile (the next one waits in registers until the end of the tile)
    float* Hs = smem + C * XS;                 // [C][XS] 1st-conv output; later c5 output (half B)
    float* Wi = Hs + C * XS;                   // resident weight images
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int len = a.len;
    const int lin = CF::SECOND ? len : len / a.xf;

    // ---- once per workgroup: all weights -> LDS -------------------------------------------------------
    stage_weights(Wi + CF::O_WA, a.wa, 72, tid, NT);
    stage_weights(Wi + CF::O_WB, a.wb, 72, tid, NT);
    stage_weights(Wi + CF::O_SC, a.wsc, 24, tid, NT);
    stage_weights(Wi + CF::O_SH, a.wsh, 24, tid, NT);
    if (CF::SECOND) {
        stage_weights(Wi + CF::O_W5, a.w5, 24, tid, NT);
        for (int i = tid; i < 24 * 7; i += NT) Wi[CF::O_W7 + i] = a.w7[i];
    }

    // biases -> LDS (read in every epilogue; keeps VMEM out of the tile loop)
    float* Bi = Wi + CF::WFLOATS;              // [5][32]: ba, bb, bsc, bsh, b5
    if (tid < 32) {
        Bi[tid] = a.ba[tid];
        Bi[32 + tid] = a.bb[tid];
        Bi[64 + tid] = a.bsc[tid];
        Bi[96 + tid] = a.bsh[tid];
        Bi[128 + tid] = CF::SECOND ? a.b5[tid] : 0.f;
    }

    // input tile staging, row-mapped: wave w owns channels w, w + NWAVES, ...; lanes run along columns
    // (coalesced, one add + clamp per element).  Half A keeps both interpolation taps in registers.
    constexpr int RPW = (C + CF::NWAVES - 1) / CF::NWAVES, NCH = (CF::XW + 63) / 64;
    float xr0[RPW][NCH], xr1[RPW][NCH];
    auto fetch = [&](int tile) {
        const int b = tile / a.tiles_per_utt;
        const int px0 = (tile - b * a.tiles_per_utt) * W - E - H;
        const float* xb = a.x + (long)b * C * lin;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = lane + 64 * j;
            int p = px0 + c;
            p = p < 0 ? 0 : (p > len - 1 ? len - 1 : p);
            Lerp lc;
            if (!CF::SECOND) lc = lerp_coord(p, a.interp_scale, lin);
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int ci = wave + CF::NWAVES * r;
                if (c < CF::XW && ci < C) {
                    const float* xrow = xb + (long)ci * lin;
                    if (CF::SECOND) {
                        xr0[r][j] = xrow[p];
                    } else {
                        xr0[r][j] = xrow[lc.i0];
                        xr1[r][j] = xrow[lc.i1];
                    }
                }
            }
        }
    };
    auto deposit = [&](float* Xs, int tile) {
        const int b = tile / a.tiles_per_utt;
        const int px0 = (tile - b * a.tiles_per_utt) * W - E - H;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = lane + 64 * j;
            Lerp lc;
            if (!CF::SECOND) {
                int p = px0 + c;
                p = p < 0 ? 0 : (p > len - 1 ? len - 1 : p);
                lc = lerp_coord(p, a.interp_scale, lin);
            }
#pragma unroll
            for (int r = 0; r < RPW; ++r) {
                const int ci = wave + CF::NWAVES * r;
                if (c < CF::XW && ci < C)
                    Xs[ci * XS + c] = CF::SECOND ? xr0[r][j] : lerp_eval(lc, xr0[r][j], xr1[r][j]);
            }
        }
    };

    int tile = blockIdx.x;
    if (tile < a.ntiles) {
        fetch(tile);
        deposit(Xbuf, tile);
    }
    __syncthreads();

    int cur = 0;
    for (; tile < a.ntiles; tile += gridDim.x, cur ^= 1) {
        float* Xs = Xbuf;
        const int b = tile / a.tiles_per_utt;
        const int t0 = (tile - b * a.tiles_per_utt) * W;
        const int px0 = t0 - E - H;       // position of Xs column 0
        const int ph0 = t0 - E - D2;      // position of Hs column 0
        const int p20 = t0 - E;           // position of 2nd-conv column 0
        const int next = tile + gridDim.x;

        // FiLM cond fragments of this wave's S2 tile (B operand: k = 4 s + lq, column l15)
        float cf[CF::NS2][6];
        {
            const float* cb = a.cond + (long)b * C * len;
#pragma unroll
            for (int j = 0; j < CF::NS2; ++j) {
                int t = p20 + (wave + j * CF::NWAVES) * 16 + l15;
                t = t < 0 ? 0 : (t > len - 1 ? len - 1 : t);
#pragma unroll
                for (int s = 0; s < 6; ++s) cf[j][s] = cb[(long)(4 * s + lq) * len + t];
            }
        }

        // next tile's input: issued after the cond loads so S2's counted vmcnt wait never covers it
        if (UP24_EXP != 2 && next < a.ntiles) fetch(next);   // lands in registers during S1..S4

        // ---- S1: Hs = lrelu(conv_a(lrelu(x)) + ba) over the extended range ---------------------------
        {
            const int lo = -px0 > 0 ? -px0 : 0;
            const int hi = (len - 1 - px0) < (CF::XW - 1) ? (len - 1 - px0) : (CF::XW - 1);
            for (int nt = wave; nt < CF::HWr / 16; nt += CF::NWAVES) {
                f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                ConvOps<C, 3> ops;
                conv_load16<C, 3>(ops, Wi + CF::O_WA, Xs, XS, nt * 16 + l15 + D1, D1, lo, hi, l15, lq);
                if (UP24_EXP != 3) conv_mma16<C, 3, true>(acc, ops);
                else acc[0][0] = ops.a0[0] + ops.b[3] + ops.a1[17];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int m = mt * 16 + 4 * lq + r;
                        if (m < C) Hs[m * XS + nt * 16 + l15] = lrelu_p1(acc[mt][r] + Bi[m]);
                    }
            }
        }
        if (UP24_EXP != 5) __syncthreads();

        // ---- S2: (conv_b(Hs) + bb) * scale + shift + x -----------------------------------------------
        {
            const int lo = -ph0 > 0 ? -ph0 : 0;
            const int hi = (len - 1 - ph0) < (CF::HW - 1) ? (len - 1 - ph0) : (CF::HW - 1);
#pragma unroll
            for (int j = 0; j < CF::NS2; ++j) {
                const int nt = wave + j * CF::NWAVES;
                if (nt >= CF::W2r / 16) break;
                const int n = nt * 16 + l15;
                const int t = p20 + n;
                f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                f32x4 asc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                f32x4 ash[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                ConvOps<C, 3> ops;
                conv_load16<C, 3>(ops, Wi + CF::O_WB, Hs, XS, n + D2, D2, lo, hi, l15, lq);
                float fsc0[6], fsc1[6], fsh0[6], fsh1[6];
                {
                    const float* wl = Wi + (lq >> 1) * 64 + (lq & 1) * 16 + l15;
#pragma unroll
                    for (int s = 0; s < 6; ++s) {
                        fsc0[s] = wl[CF::O_SC + s * 128];
                        fsc1[s] = wl[CF::O_SC + s * 128 + 32];
                        fsh0[s] = wl[CF::O_SH + s * 128];
                        fsh1[s] = wl[CF::O_SH + s * 128 + 32];
                    }
                }
                if (UP24_EXP != 4) conv_mma16<C, 3, false>(acc, ops);
                else acc[0][0] = ops.a0[0] + ops.b[3] + ops.a1[17];
#pragma unroll
                for (int s = 0; s < (UP24_EXP == 4 ? 1 : 6); ++s) {
                    asc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fsc0[s], cf[j][s], asc[0], 0, 0, 0);
                    asc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fsc1[s], cf[j][s], asc[1], 0, 0, 0);
                    ash[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fsh0[s], cf[j][s], ash[0], 0, 0, 0);
                    ash[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fsh1[s], cf[j][s], ash[1], 0, 0, 0);
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int m = mt * 16 + 4 * lq + r;
                        if (m < C && n + H < CF::XW) {
                            float hval = acc[mt][r] + Bi[32 + m];
                            float sc = asc[mt][r] + Bi[64 + m];
                            float sh = ash[mt][r] + Bi[96 + m];
                            float res = Xs[m * XS + n + H];
                            float v = __fadd_rn(__fadd_rn(__fmul_rn(hval, sc), sh), res);
                            if (CF::SECOND)
                                Xs[m * XS + n + H] = v;                       // x2 stays on chip
                            else if (t < len && n < W)
                                a.out[((long)b * C + m) * len + t] = v;       // x1
                        }
                    }
            }
        }
        if (CF::SECOND) {
            __syncthreads();
            // ---- S3: c5(x2) + b5, parked in Hs ------------------------------------------------------
            for (int nt = wave; nt < CF::W2r / 16; nt += CF::NWAVES) {
                const int n = nt * 16 + l15;
                f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                ConvOps<C, 1> ops;
                conv_load16<C, 1>(ops, Wi + CF::O_W5, Xs, XS, n + H, 0, 0, CF::XW - 1, l15, lq);
                conv_mma16<C, 1, false>(acc, ops);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int m = mt * 16 + 4 * lq + r;
                        if (m < C) Hs[m * XS + n] = acc[mt][r] + Bi[128 + m];
                    }
            }
            __syncthreads();
            // ---- S4: output_layer, Conv1d(24 -> 1, k7, replicate) on the parked tile ----------------
            // 8 lanes per group of 4 consecutive outputs, 3 channels each: per channel 10 activations
            // and 7 (broadcast) weights feed 28 FMAs; the 8 partial sums meet through three shuffles.
            {
                const int part = tid & 7;
                const int lo = -p20 > 0 ? -p20 : 0;
                const int hi = (len - 1 - p20) < (CF::W2 - 1) ? (len - 1 - p20) : (CF::W2 - 1);
                for (int g0 = 0; g0 < (W + 3) / 4; g0 += NT / 8) {      // uniform trip count: the shuffles need every lane
                    const int g = g0 + (tid >> 3);
                    float o4[4] = {0.f, 0.f, 0.f, 0.f};
                    if (UP24_EXP != 1 && 4 * g < W) {
                        int cols[10];
#pragma unroll
                        for (int i = 0; i < 10; ++i) {
                            int c = 4 * g + E - 3 + i;
                            cols[i] = c < lo ? lo : (c > hi ? hi : c);
                        }
#pragma unroll
                        for (int cc = 0; cc < 3; ++cc) {
                            const int c = part * 3 + cc;
                            float xv[10], wv[7];
#pragma unroll
                            for (int i = 0; i < 10; ++i) xv[i] = Hs[c * XS + cols[i]];
#pragma unroll
                            for (int j = 0; j < 7; ++j) wv[j] = Wi[CF::O_W7 + c * 7 + j];
#pragma unroll
                            for (int q = 0; q < 4; ++q)
#pragma unroll
                                for (int j = 0; j < 7; ++j) o4[q] = fmaf(wv[j], xv[q + j], o4[q]);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        o4[q] += __shfl_xor(o4[q], 1);
                        o4[q] += __shfl_xor(o4[q], 2);
                        o4[q] += __shfl_xor(o4[q], 4);
                    }
                    const int o = 4 * g + part;                  // lanes 0..3 of a group store outputs 4g..4g+3
                    const float v = part == 0 ? o4[0] : (part == 1 ? o4[1] : (part == 2 ? o4[2] : o4[3]));
                    if (part < 4 && o < W && t0 + o < len) a.out[(long)b * len + t0 + o] = v + a.b7[0];
                }
            }
        }
        // ---- next tile's input: registers -> the other LDS buffer ---------------------------------------
        if (!CF::SECOND && UP24_EXP != 5) __syncthreads();     // half A: S2 still reads the input tile (residual)
        if (UP24_EXP != 2 && next < a.ntiles) deposit(Xbuf, next);
        if (UP24_EXP != 5) __syncthreads();
    }
}

template <class CF>
static int launch_up24(tvc_ctx* ctx, hipStream_t s, Up24Args a, int B) {
    static int ncu_dev[64] = {};                    // per device of this process (the LDS attribute is per function and device)
    int& ncu = ncu_dev[ctx->device & 63];
    const size_t lds = (size_t)CF::LDS_FLOATS * sizeof(float);
    if (!ncu) {
        hipDeviceProp_t prop;
        hipError_t e = hipGetDeviceProperties(&prop, ctx->device);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)up24_kernel<CF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(ctx, TVC_ERR_HIP, "up24 setup: %s", hipGetErrorString(e));
        ncu = prop.multiProcessorCount;
    }
    a.tiles_per_utt = (a.len + CF::W - 1) / CF::W;
    a.ntiles = a.tiles_per_utt * B;
    int grid = a.ntiles < ncu ? a.ntiles : ncu;
    hipLaunchKernelGGL((up24_kernel<CF>), dim3(grid), dim3(CF::NT), lds, s, a);
    return launch_check(ctx, "up24");
}

// Upsample block with cin == 24 followed by FilterNet.output_layer:
// x [B][24][len/f], cond [B][24][len] -> wave [B][len]; x1 is scratch [B][24][len].
int run_up24_fused(tvc_ctx* ctx, hipStream_t s, const UpW& u, const float* x, const float* cond, float* x1, float* wave,
                   int B, int len, const float* w7, const float* b7) {
#ifndef UP24_WA
#define UP24_WA 384
#endif
#ifndef UP24_WB
#define UP24_WB 378
#endif
    using CA = Up24Cfg<UP24_WA, 1, 3, false, 0, UP24_NT>;
    using CB = Up24Cfg<UP24_WB, 9, 27, true, 3, UP24_NT>;
    Up24Args a{};
    a.len = len;
    a.xf = u.factor;
    a.interp_scale = (float)(1.0 / (double)u.factor);
    a.cond = cond;
    a.x = x;
    a.out = x1;
    a.wa = u.c1.At_tap; a.ba = u.c1.bias;
    a.wb = u.c2.At_tap; a.bb = u.c2.bias;
    a.wsc = u.sc1.At; a.bsc = u.sc1.bias;
    a.wsh = u.sh1.At; a.bsh = u.sh1.bias;
    TVC_CHECK(launch_up24<CA>(ctx, s, a, B));
    a.x = x1;
    a.out = wave;
    a.wa = u.c3.At_tap; a.ba = u.c3.bias;
    a.wb = u.c4.At_tap; a.bb = u.c4.bias;
    a.wsc = u.sc2.At; a.bsc = u.sc2.bias;
    a.wsh = u.sh2.At; a.bsh = u.sh2.bias;
    a.w5 = u.c5.At; a.b5 = u.c5.bias;
    a.w7 = w7; a.b7 = b7;
    return launch_up24<CB>(ctx, s, a, B);
}

}  // namespace tvc
