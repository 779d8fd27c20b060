// Conv1d (k = 1 or 3, dilated, replicate padding) as an implicit GEMM on the bf16 matrix pipe with
// fp32-equivalent accuracy: every fp32 operand is split into three bf16 parts (x = x1 + x2 + x3, each
// residual exact in fp32) and the product is accumulated in fp32 from the six part-products of order
// <= 2^-16 (x1 w1, x1 w2, x2 w1, x1 w3, x2 w2, x3 w1; the dropped ones are <= 2^-24 relative).
// v_mfma_f32_32x32x16_bf16 runs 16x the fp32 MFMA rate, so six of them cost 3/8 of the fp32 tile.
// FilterNet Downsample/Upsample convs (decoder.py:143-146, 166-171) and their FiLM (decoder.py:94-97).
//
// Layout.  K is walked in slabs of 16 input channels; a K16 step is (slab, tap).
//   weights   pre-split on the host (api.hip Packer::a6): image [step][m-tile][part][lane][8 bf16], one
//             1 KiB piece per (step, m-tile, part) already in MFMA lane order (row = lane & 31,
//             k = 8 * (lane >> 5) + j); a piece is one 16-byte load + one ds_write_b128 per lane.
//   input     the slab's halo tile is staged once: each thread loads 8 channels of one sample (coalesced
//             along time), applies the pre-activation, splits, and writes three 16-byte rows
//             Xs[part][channel-group][position][8 bf16]; a tap is a row offset, so every ds_read_b128 of
//             the MFMA loop is a contiguous 1 KiB wave access.
//   tile      a wave owns all MTB m-tiles of the workgroup for one 32-sample n-tile.
// Pipeline.  At this MFMA rate a slab's compute (1.7 k cycles per wave) is shorter than the HBM latency
// (~5 k cycles measured), so the latency is hidden by occupancy, not inside one workgroup: small
// workgroups (4 waves, 45 KB LDS) run three to a CU, each with the next slab's loads in flight in
// registers across its own compute.  Barriers are raw s_barrier + lgkmcnt(0): __syncthreads() also
// drains vmcnt, i.e. it would wait for exactly that prefetch.
#pragma once
#include <type_traits>
#include "conv3.h"
#include "tvc_common.h"

namespace tvc {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // register-friendly 16-byte value (HIP's uint4 struct defeats SROA in arrays)

#ifndef S_WPE
#define S_WPE 3     // waves per SIMD the register budget is sized for (plain / FiLM-fused kernels): 12-wave workgroups, no spills
#endif
#ifndef S_WPE_F
#define S_WPE_F 3
#endif
#ifndef S_FB
#define S_FB 2   // fragment register sets: 2 = next tap's LDS reads under this tap's MFMAs, 1 = read, then multiply
#endif
#ifndef S_ABL
#define S_ABL 0   // timing ablations (wrong results): 1 no MFMA, 2 no weight loads, 4 no input loads, 8 no input split/stores, 16 no output stores
#endif

#ifndef S_DBG
#define S_DBG 0   // 1: workgroup 300 wave 0 records s_memtime stamps and the launcher prints them
#endif
#if S_DBG
__device__ unsigned long long g_sdbg[64];
#define S_STAMP(k)                                                                                               \
    do {                                                                                                         \
        if (blockIdx.x == 300 && threadIdx.x == 0 && (k) < 64) g_sdbg[k] = __builtin_amdgcn_s_memtime();         \
    } while (0)
#else
#define S_STAMP(k)
#endif

template <int MTB_, int WM_, int NWV_, int WN_ = 1>
struct SplitTile {
    static constexpr int MTB = MTB_, WM = WM_, NWV = NWV_, WN = WN_;            // m-tiles per workgroup / per wave, waves along time, n-tiles per wave
    static constexpr int MW = MTB / WM, NW = MW * NWV, NTHR = NW * 64;
    static_assert(MTB % WM == 0, "wave rows must tile the workgroup");
    static constexpr int BM = MTB * 32, BN = NWV * WN * 32;
    static constexpr int MAXD = 27, XROW = BN + 2 * MAXD;
    static constexpr int X_U4 = 3 * 2 * XROW;
    static constexpr int X_PER = (2 * XROW + NTHR - 1) / NTHR;                  // staging items per thread
    static constexpr int a_u4(int taps) { return taps * MTB * 3 * 64; }
    static constexpr int lds_bytes(int taps) { return (a_u4(taps) + X_U4) * 16; }
};

struct ConvSArgs {
    const uint4* A6;     // split weight image
    int MT;              // m-tiles in the image
    const float* x;      // [B][Cin][len]
    int Cin, len, dil, tiles_per_utt;
    const uint4* sc6 = nullptr;   // FiLM scale / shift images (1x1 over cond), FILM kernels only
    const uint4* sh6 = nullptr;
    const float* cond = nullptr;
    int Ccond = 0;
};

// three bf16 parts of 8 fp32 values, packed for one 16-byte LDS row each
__device__ __forceinline__ void split8(const float (&v)[8], uint4& p1, uint4& p2, uint4& p3) {
    unsigned o1[4], o2[4], o3[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x2 a = {v[2 * j], v[2 * j + 1]};
        bf16x2 h1 = __builtin_convertvector(a, bf16x2);
        f32x2 r = a - __builtin_convertvector(h1, f32x2);
        bf16x2 h2 = __builtin_convertvector(r, bf16x2);
        f32x2 r2 = r - __builtin_convertvector(h2, f32x2);
        bf16x2 h3 = __builtin_convertvector(r2, bf16x2);
        o1[j] = __builtin_bit_cast(unsigned, h1);
        o2[j] = __builtin_bit_cast(unsigned, h2);
        o3[j] = __builtin_bit_cast(unsigned, h3);
    }
    p1 = make_uint4(o1[0], o1[1], o1[2], o1[3]);
    p2 = make_uint4(o2[0], o2[1], o2[2], o2[3]);
    p3 = make_uint4(o3[0], o3[1], o3[2], o3[3]);
}

// workgroup barrier that drains this wave's LDS traffic but not its global loads
__device__ __forceinline__ void slab_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// acc += W (.) x over all slabs of one input tensor
template <class TL, int TAPS, int A_U4, bool LRELU>
__device__ __forceinline__ void split_phase(f32x16 (&acc)[TL::WM][TL::WN], const uint4* __restrict__ A6, int MT, int mt0,
                                            const float* __restrict__ xb, int Cin, int len, int dil, int t0, uint4* As, uint4* Xs) {
    constexpr int MTB = TL::MTB, WM = TL::WM, WN = TL::WN, NWV = TL::NWV, NW = TL::NW, NTHR = TL::NTHR, XROW = TL::XROW, BN = TL::BN, X_PER = TL::X_PER;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / NWV, wn = wave - wm * NWV;
    const int xw = BN + 2 * dil;

    // staging map of this thread, fixed across slabs: item -> (channel group g, staged column c)
    unsigned xo[X_PER];                                      // utterance-relative element offset of (channel 8 g, position p)
    int xdst[X_PER], xg8[X_PER];
    float xr[X_PER][8];
#pragma unroll
    for (int i = 0; i < X_PER; ++i) {
        int idx = tid + i * NTHR;
        int g = idx / xw, c = idx - g * xw;
        int p = t0 - dil + c;
        p = p < 0 ? 0 : (p > len - 1 ? len - 1 : p);
        xdst[i] = g < 2 ? g * XROW + c : -1;
        g = g < 2 ? g : 1;                                   // idle items still load (valid address), never store
        xg8[i] = 8 * g;
        xo[i] = (unsigned)(8 * g * len + p);
    }
    constexpr int PIECES = TAPS * MTB * 3, A_PER = (PIECES + NW - 1) / NW;
    const uint4* a_src = A6 + (long)mt0 * 192;
    u32x4 ar[A_PER];
    // global -> registers only (no use of the values here: the loads stay in flight behind the MFMAs)
    auto gload = [&](int s) __attribute__((always_inline)) {
        const int ci0 = s * 16;
        if (!(S_ABL & 2)) {
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                int q = wave + i * NW;
                q = q < PIECES ? q : PIECES - 1;
                const int tap = q / (MTB * 3), rem = q - tap * (MTB * 3);
                ar[i] = *reinterpret_cast<const u32x4*>(a_src + ((long)(s * TAPS + tap) * MT * 192 + rem * 64) + lane);
            }
        }
        if (S_ABL & 4) return;
        const float* xc = xb + (long)ci0 * len;              // uniform base, 32-bit lane offsets
        if (ci0 + 16 <= Cin) {
#pragma unroll
            for (int i = 0; i < X_PER; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) xr[i][j] = xc[xo[i] + (unsigned)(j * len)];
        } else {   // ragged last slab: channels >= Cin read as zero
#pragma unroll
            for (int i = 0; i < X_PER; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int cj = ci0 + xg8[i] + j;
                    float v = xc[xo[i] + (unsigned)((cj < Cin ? j : 0) * len)];
                    xr[i][j] = cj < Cin ? v : 0.f;
                }
        }
    };
    auto lstore = [&]() __attribute__((always_inline)) {
        if (!(S_ABL & 2)) {
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                const int q = wave + i * NW;
                if (q < PIECES) *reinterpret_cast<u32x4*>(As + q * 64 + lane) = ar[i];
            }
        }
#pragma unroll
        for (int i = 0; i < X_PER; ++i)
            if (xdst[i] >= 0 && !(S_ABL & 8)) {
                if (LRELU) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) xr[i][j] = fmaxf(xr[i][j], 0.1f * xr[i][j]);   // = leaky_relu(x, 0.1)
                }
                uint4 p1, p2, p3;
                split8(xr[i], p1, p2, p3);
                Xs[xdst[i]] = p1;
                Xs[2 * XROW + xdst[i]] = p2;
                Xs[4 * XROW + xdst[i]] = p3;
            }
    };

    const int nslab = (Cin + 15) / 16;
    const uint4* as = As + wm * WM * 192 + lane;
    const uint4* xs = Xs + lh * XROW + wn * WN * 32 + l31;
    S_STAMP(1);
    gload(0);
    for (int s = 0; s < nslab; ++s) {
        if (TAPS == 3) S_STAMP(2 + 4 * s);
        slab_barrier();                            // every wave is done reading the previous slab
        if (TAPS == 3) S_STAMP(3 + 4 * s);
        lstore();                                  // slab s: registers -> LDS
        if (s + 1 < nslab) gload(s + 1);           // slab s+1 flies across this slab's MFMAs
        if (TAPS == 3) S_STAMP(4 + 4 * s);
        slab_barrier();
        if (TAPS == 3) S_STAMP(5 + 4 * s);
        // fragments of tap t+1 are read while the MFMAs of tap t run
        bf16x8 af[2][WM][3], bf[2][WN][3];
        auto frags = [&](int tap, int fb) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) bf[fb][j][p] = __builtin_bit_cast(bf16x8, xs[2 * p * XROW + j * 32 + tap * dil]);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) af[fb][i][p] = __builtin_bit_cast(bf16x8, as[(tap * MTB * 3 + i * 3 + p) * 64]);
        };
        if (S_FB == 2) frags(0, 0);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int fb = S_FB == 2 ? tap & 1 : 0;
            if (S_FB == 2) {
                if (tap + 1 < TAPS) frags(tap + 1, fb ^ 1);
                __builtin_amdgcn_sched_barrier(0);   // keep the reads above the MFMAs (the scheduler sinks them to just-in-time otherwise)
            } else {
                frags(tap, 0);
            }
            // part-products from the smallest order up; independent accumulators interleaved
            constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        if (!(S_ABL & 1) || q == 0)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[fb][i][PA[q]], bf[fb][j][PB[q]], acc[i][j], 0, 0, 0);
            if (S_FB == 2) __builtin_amdgcn_sched_barrier(0);
#if S_DBG
            if (TAPS == 3 && s == 2) { S_STAMP(41 + tap); __builtin_amdgcn_sched_barrier(0); }
#endif
        }
    }
}

template <class TL, int TAPS, bool LRELU, class Epi, bool FILM>
__global__ __launch_bounds__(TL::NTHR) __attribute__((amdgpu_waves_per_eu(FILM ? S_WPE_F : S_WPE))) void conv3s_kernel(ConvSArgs a, Epi ep) {
    constexpr int MTB = TL::MTB, WM = TL::WM, WN = TL::WN, A_U4 = TL::a_u4(TAPS);
    extern __shared__ __attribute__((aligned(16))) uint4 smem_s[];
    uint4* As = smem_s;
    uint4* Xs = smem_s + A_U4;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / TL::NWV, wn = wave - wm * TL::NWV;
    const int mblocks = a.MT / MTB;
    const int mt0 = (blockIdx.x % mblocks) * MTB;
    const int nt_id = blockIdx.x / mblocks;
    const int b = nt_id / a.tiles_per_utt;
    const int t0 = (nt_id - b * a.tiles_per_utt) * TL::BN;
    const int len = a.len;
    S_STAMP(0);

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    split_phase<TL, TAPS, A_U4, LRELU>(acc, a.A6, a.MT, mt0, a.x + (long)b * a.Cin * len, a.Cin, len, a.dil, t0, As, Xs);
    const int tw = t0 + wn * WN * 32 + l31;        // column of n-tile 0 of this wave

    if constexpr (FILM) {
        // conv -> FiLM -> + residual (decoder.py:94-97,181-182): scale and shift are two more 1x1 phases over
        // cond on the same tiles; the conv result is folded with the scale before the shift phase runs, so
        // at most two accumulator sets are live.
        const float* cb = a.cond + (long)b * a.Ccond * len;
        f32x16 a2[WM][WN];
        auto zero2 = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) a2[i][j][r] = 0.f;
        };
        zero2();
        split_phase<TL, 1, A_U4, false>(a2, a.sc6, a.MT, mt0, cb, a.Ccond, len, 0, t0, As, Xs);
        const int mb = (mt0 + wm * WM) * 32 + 4 * lh;              // row of accumulator register r of m-tile i: mb + 32 i + (r & 3) + 8 (r >> 2)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = mb + i * 32 + (r & 3) + 8 * (r >> 2);
                m = m < ep.M ? m : ep.M - 1;
                const float bm = ep.bias[m], bs = ep.bsc[m];
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j][r] = __fmul_rn(acc[i][j][r] + bm, a2[i][j][r] + bs);
            }
        zero2();
        split_phase<TL, 1, A_U4, false>(a2, a.sh6, a.MT, mt0, cb, a.Ccond, len, 0, t0, As, Xs);
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int t = tw + j * 32;
            if (t < len) {
                float* yb = ep.y + (long)b * ep.M * len + t;          // offsets within one utterance fit 32 bits
                const float* rb = ep.res + (long)b * ep.M * len + t;
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mb + i * 32 + (r & 3) + 8 * (r >> 2);
                        if (m < ep.M) yb[m * len] = __fadd_rn(__fadd_rn(acc[i][j][r], a2[i][j][r] + ep.bsh[m]), rb[m * len]);
                    }
            }
        }
    } else {
        const int mb = (mt0 + wm * WM) * 32 + 4 * lh;
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int t = tw + j * 32;
            if (t < len && !(S_ABL & 16)) {
                float* yb = ep.y + (long)b * ep.M * len + t;          // offsets within one utterance fit 32 bits
                const float* rb = nullptr;
                if constexpr (Epi::kRes) rb = ep.res + (long)b * ep.M * len + t;
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mb + i * 32 + (r & 3) + 8 * (r >> 2);
                        if (m < ep.M) {
                            float o = acc[i][j][r] + ep.bias[m];
                            if constexpr (Epi::kRes) o += rb[m * len];
                            yb[m * len] = o;
                        }
                    }
            }
        }
        S_STAMP(40);
    }
}

template <class TL, int TAPS, bool LRELU, class Epi, bool FILM>
inline int conv3s_launch_t(tvc_ctx* ctx, hipStream_t s, const PackedW& w, const float* x, int B, int Cin, int len, int dil, const Epi& ep,
                           const PackedW* wsc, const PackedW* wsh, const float* cond, int Ccond) {
    static bool ready = false;
    constexpr int lds = TL::lds_bytes(TAPS);
    if (!ready) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3s_kernel<TL, TAPS, LRELU, Epi, FILM>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return fail(ctx, TVC_ERR_HIP, "conv3s setup: %s", hipGetErrorString(e));
        ready = true;
    }
    ConvSArgs a;
    a.A6 = reinterpret_cast<const uint4*>(w.A6);
    a.MT = w.MT6;
    a.x = x;
    a.Cin = Cin;
    a.len = len;
    a.dil = dil;
    a.tiles_per_utt = (len + TL::BN - 1) / TL::BN;
    if (FILM) {
        a.sc6 = reinterpret_cast<const uint4*>(wsc->A6);
        a.sh6 = reinterpret_cast<const uint4*>(wsh->A6);
        a.cond = cond;
        a.Ccond = Ccond;
    }
    dim3 g((unsigned)((a.MT / TL::MTB) * a.tiles_per_utt * B));
    hipLaunchKernelGGL((conv3s_kernel<TL, TAPS, LRELU, Epi, FILM>), g, dim3(TL::NTHR), lds, s, a, ep);
#if S_DBG
    if (!FILM && g.x > 300) {
        static int shown = 0;
        if (shown++ == 40) {
            unsigned long long h[64];
            hipDeviceSynchronize();
            hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sdbg), sizeof(h));
            const int ns = (Cin + 15) / 16;
            fprintf(stderr, "[sdbg] grid %u Cin %d len %d: start->phase %llu;", g.x, Cin, len, h[1] - h[0]);
            for (int k = 0; k < ns; ++k)
                fprintf(stderr, " [slab %d: top %llu bar %llu stage %llu bar %llu]", k, h[2 + 4 * k] - h[0], h[3 + 4 * k] - h[2 + 4 * k],
                        h[4 + 4 * k] - h[3 + 4 * k], h[5 + 4 * k] - h[4 + 4 * k]);
            fprintf(stderr, " end %llu; slab2 after bar->tap0 %llu tap1 %llu tap2 %llu\n", h[40] - h[0], h[41] - h[13], h[42] - h[41], h[43] - h[42]);
        }
    }
#endif
    return 0;
}

#ifndef TVC_S_WN
#define TVC_S_WN 1
#endif
#ifndef TVC_S48_NWV   // waves (= 32-sample n-tiles) per m-tile of the 48-channel workgroups
#define TVC_S48_NWV 6
#endif
#ifndef TVC_S48F_NWV
#define TVC_S48F_NWV 6
#endif
#ifndef TVC_SF_WM   // tile of the FiLM-fused kernels (two accumulator sets live)
#define TVC_SF_WM 1
#endif
#ifndef TVC_SF_NWV
#define TVC_SF_NWV 4
#endif
#ifndef TVC_SF_WN
#define TVC_SF_WN 1
#endif
#ifndef TVC_S_WM
#define TVC_S_WM 1
#endif
#ifndef TVC_S_NWV
#define TVC_S_NWV 4
#endif

// k3 conv on the split path; Mpad = 64 (48 channels) or a multiple of 96 (FilterNet levels with C = 96, 192, 384)
template <bool LRELU, class Epi, bool FILM = false>
inline int conv3s_launch(tvc_ctx* ctx, hipStream_t s, const PackedW& w, const float* x, int B, int Cin, int len, int dil, const Epi& ep,
                         const PackedW* wsc = nullptr, const PackedW* wsh = nullptr, const float* cond = nullptr, int Ccond = 0) {
    if (w.MT6 == 2) {   // 48 output channels: two m-tiles, the second half empty (still 1.4x fewer MFMA cycles than exact fp32 tiles)
        if constexpr (FILM)
            return conv3s_launch_t<SplitTile<2, 1, TVC_S48F_NWV, 1>, 3, LRELU, Epi, FILM>(ctx, s, w, x, B, Cin, len, dil, ep, wsc, wsh, cond, Ccond);
        else
            return conv3s_launch_t<SplitTile<2, 1, TVC_S48_NWV, 1>, 3, LRELU, Epi, FILM>(ctx, s, w, x, B, Cin, len, dil, ep, wsc, wsh, cond, Ccond);
    }
    if constexpr (FILM)
        return conv3s_launch_t<SplitTile<3, TVC_SF_WM, TVC_SF_NWV, TVC_SF_WN>, 3, LRELU, Epi, FILM>(ctx, s, w, x, B, Cin, len, dil, ep, wsc, wsh, cond, Ccond);
    else
        return conv3s_launch_t<SplitTile<3, TVC_S_WM, TVC_S_NWV, TVC_S_WN>, 3, LRELU, Epi, FILM>(ctx, s, w, x, B, Cin, len, dil, ep, wsc, wsh, cond, Ccond);
}

}  // namespace tvc
