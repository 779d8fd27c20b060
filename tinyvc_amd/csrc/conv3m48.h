// k=3 dilated Conv1d for the 48-channel FilterNet level on v_mfma_f32_16x16x4_f32: 48 output channels are
// exactly three 16-row MFMA tiles (the 32x32 kernel pads 48 -> 64 rows and wastes a quarter of the matrix
// pipe).  Same structure as conv3.h — LDS halo tile [8 ch][BN + 2*dil] per slab serving all three taps,
// double-buffered slabs, one barrier per slab — with a 16x16 tiling:
//   A lane l -> A[i = l&15][k = l>>4],  B lane l -> B[k = l>>4][j = l&15],  C reg r -> C[row = 4*(l>>4) + r][col = l&15].
// The 24 k-rows of a slab are held TAP-major in LDS (row = tap*8 + ci_local) so that the four k of one MFMA
// are four consecutive channels of one tap: their LDS rows differ by the row stride (= 16 mod 32 banks), the
// two rows of a 32-lane ds_read group never collide.  A wave owns 48 x (TN*16) outputs = 3*TN accumulator
// quads, so FiLM's scale/shift (two 1x1 contractions over the cond tile) fit in registers beside the conv.
#pragma once
#include "conv3.h"
#include "tvc_common.h"

namespace tvc {

typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int NW_, int TN_, int MT_ = 3>
struct C48Tile {
    static constexpr int NW = NW_, TN = TN_, MT = MT_, BMV = MT_ * 16;          // BMV = valid output rows (48 or 32)
    static constexpr int BM = BMV % 32 == 16 ? BMV : BMV + 16;                   // LDS weight row stride, = 16 (mod 32)
    static constexpr int BN = NW * TN * 16, NTHR = NW * 64;
    static constexpr int KC = 8, KS = 24, MAXD = 27;
    static constexpr int XROW = (BN + 2 * MAXD + 31) / 32 * 32 + 16;   // = 16 (mod 32)
    static constexpr int FROW = BN + 16;                               // FiLM cond tile row stride, = 16 (mod 32)
    static_assert(BN % 32 == 0, "BN multiple of 32");
};

template <class TL, bool LRELU, class Epi, bool FILM>
__global__ __launch_bounds__(TL::NTHR) void conv3m48_kernel(Conv3Args a, Epi ep) {
    constexpr int BM = TL::BM, BMV = TL::BMV, MT = TL::MT, BN = TL::BN, TN = TL::TN, KC = TL::KC, KS = TL::KS, XROW = TL::XROW, NTHR = TL::NTHR;
    constexpr int XBUF = KC * XROW > 8 * TL::FROW ? KC * XROW : 8 * TL::FROW;
    __shared__ __attribute__((aligned(16))) float As[2][KS * BM];
    __shared__ __attribute__((aligned(16))) float Xs[2][XBUF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    const int b = blockIdx.x / a.tiles_per_utt;
    const int t0 = (blockIdx.x - b * a.tiles_per_utt) * BN;
    const int len = a.len, dil = a.dil;
    const int xw = BN + 2 * dil;
    const float* xb = a.x + (long)b * a.split * len;
    const int ncol0 = wave * TN * 16;                 // this wave's first column in the tile

    // A staging: 24 rows x BMV floats per slab; LDS row = tap*8 + ci_local, row stride BM
    constexpr int A_F4 = KS * BMV / 4, A_PER = (A_F4 + NTHR - 1) / NTHR;
    constexpr int X_PER = (KC * XROW + NTHR - 1) / NTHR;
    float4 areg[A_PER];
    float xreg[X_PER];
    int xg[X_PER], xl[X_PER], xr[X_PER];
#pragma unroll
    for (int i = 0; i < X_PER; ++i) {
        int idx = tid + i * NTHR;
        int r = idx / xw, c = idx - r * xw;
        int p = t0 - dil + c;
        p = p < 0 ? 0 : (p > len - 1 ? len - 1 : p);
        xg[i] = r < KC ? p : -1;          // clamped sample position; the channel row is added per slab
        xr[i] = r;
        xl[i] = r * XROW + c;
    }
    const float* x2b = a.x2 ? a.x2 + (long)b * (a.Cin - a.split) * len : nullptr;
    auto load_slab = [&](int ci0) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int idx = tid + i * NTHR;
            if (idx < A_F4) {
                int kk = idx / (BMV / 4), c4 = idx - kk * (BMV / 4);    // kk = global row within the slab = ci_l*3 + tap
                areg[i] = ci0 * 3 + kk < a.Krows ? *reinterpret_cast<const float4*>(a.At + (long)(ci0 * 3 + kk) * a.Mpad + c4 * 4)
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int i = 0; i < X_PER; ++i) {
            float v = 0.f;
            if (xg[i] >= 0) {
                const int ci = ci0 + xr[i];
                if (ci < a.split) v = xb[(long)ci * len + xg[i]];
                else if (ci < a.Cin) v = x2b[(long)(ci - a.split) * len + xg[i]];
            }
            if (LRELU) v = v > 0.f ? v : 0.1f * v;
            xreg[i] = v;
        }
    };
    auto store_slab = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int idx = tid + i * NTHR;
            if (idx < A_F4) {
                int kk = idx / (BMV / 4), c4 = idx - kk * (BMV / 4);
                int cil = kk / 3, tap = kk - 3 * cil;
                *reinterpret_cast<float4*>(&As[buf][(tap * KC + cil) * BM + c4 * 4]) = areg[i];
            }
        }
#pragma unroll
        for (int i = 0; i < X_PER; ++i)
            if (xg[i] >= 0) Xs[buf][xl[i]] = xreg[i];
    };

    f32x4v acc[MT][TN];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

    const int nslab = (a.Cin + KC - 1) / KC;
    load_slab(0);
    store_slab(0);
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
        const int cur = s & 1;
        load_slab((s + 1 < nslab ? s + 1 : s) * KC);
        const float* as = As[cur] + lq * BM + l15;
        const float* xs = Xs[cur] + lq * XROW + ncol0 + l15;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
            for (int c4 = 0; c4 < KC; c4 += 4) {
                float av[MT], bv[TN];
#pragma unroll
                for (int i = 0; i < MT; ++i) av[i] = as[(tap * KC + c4) * BM + i * 16];
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[j] = xs[c4 * XROW + tap * dil + j * 16];
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        store_slab(cur ^ 1);
        __syncthreads();
    }

    if constexpr (FILM) {
        f32x4v asc[MT][TN], ash[MT][TN];
        constexpr int FK = 8, FROW = TL::FROW;
        constexpr int FA_F4 = FK * BMV / 4, FA_PER = (FA_F4 + NTHR - 1) / NTHR;
        constexpr int FB_PER = (FK * BN + NTHR - 1) / NTHR;
        const float* cb = a.cond + (long)b * a.Ccond * len;
        auto film_phase = [&](const float* Wt, f32x4v (&out)[MT][TN]) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) out[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
            float4 fa[FA_PER];
            float fb[FB_PER];
            auto fload = [&](int c0) {
#pragma unroll
                for (int i = 0; i < FA_PER; ++i) {
                    int idx = tid + i * NTHR;
                    if (idx < FA_F4) {
                        int kk = idx / (BMV / 4), c4 = idx - kk * (BMV / 4);
                        fa[i] = *reinterpret_cast<const float4*>(Wt + (long)(c0 + kk) * a.Mpad + c4 * 4);
                    }
                }
#pragma unroll
                for (int i = 0; i < FB_PER; ++i) {
                    int idx = tid + i * NTHR;
                    int r = idx / BN, c = idx - r * BN;
                    int t = t0 + c;
                    t = t > len - 1 ? len - 1 : t;
                    fb[i] = (r < FK && c0 + r < a.Ccond) ? cb[(long)(c0 + r) * len + t] : 0.f;
                }
            };
            auto fstore = [&](int buf) {
#pragma unroll
                for (int i = 0; i < FA_PER; ++i) {
                    int idx = tid + i * NTHR;
                    if (idx < FA_F4) {
                        int kk = idx / (BMV / 4), c4 = idx - kk * (BMV / 4);
                        *reinterpret_cast<float4*>(&As[buf][kk * BM + c4 * 4]) = fa[i];
                    }
                }
#pragma unroll
                for (int i = 0; i < FB_PER; ++i) {
                    int idx = tid + i * NTHR;
                    int r = idx / BN, c = idx - r * BN;
                    if (r < FK) Xs[buf][r * FROW + c] = fb[i];
                }
            };
            const int ns = (a.Ccond + FK - 1) / FK;
            fload(0);
            fstore(0);
            __syncthreads();
            for (int s2 = 0; s2 < ns; ++s2) {
                const int cur = s2 & 1;
                fload((s2 + 1 < ns ? s2 + 1 : s2) * FK);
                const float* as = As[cur] + lq * BM + l15;
                const float* bs = Xs[cur] + lq * FROW + ncol0 + l15;
#pragma unroll
                for (int c4 = 0; c4 < FK; c4 += 4) {
                    float av[MT], bv[TN];
#pragma unroll
                    for (int i = 0; i < MT; ++i) av[i] = as[c4 * BM + i * 16];
#pragma unroll
                    for (int j = 0; j < TN; ++j) bv[j] = bs[c4 * FROW + j * 16];
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            out[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], out[i][j], 0, 0, 0);
                }
                fstore(cur ^ 1);
                __syncthreads();
            }
        };
        film_phase(a.sc_At, asc);
        film_phase(a.sh_At, ash);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int t = t0 + ncol0 + j * 16 + l15;
                if (t < len) {
                    const int m = i * 16 + 4 * lq;
                    float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    float c1[4] = {asc[i][j][0], asc[i][j][1], asc[i][j][2], asc[i][j][3]};
                    float c2[4] = {ash[i][j][0], ash[i][j][1], ash[i][j][2], ash[i][j][3]};
                    ep.store(b, t, m, v, c1, c2);
                }
            }
    } else {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int t = t0 + ncol0 + j * 16 + l15;
                if (t < len) {
                    const int m = i * 16 + 4 * lq;
                    float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    ep.store(b, t, m, v);
                }
            }
    }
}

// Launch for M <= 48 (MT = 3) or M <= 32 (MT = 2); only the first MT*16 columns of each At row are read.
// `x2` (optional): channels [split, Cin) come from a second tensor [B][Cin - split][len] (downs[0]'s cat).
template <int MT, bool LRELU, class Epi, bool FILM = false>
inline void conv3mt_launch(hipStream_t s, const PackedW& w, const float* x, int B, int Cin, int len, int dil, const Epi& ep,
                           const FilmOps& f = FilmOps(), const float* x2 = nullptr, int split = 0) {
#ifndef TVC_C48_NW
#define TVC_C48_NW 8
#endif
#ifndef TVC_C24_TN
#define TVC_C24_TN 2     // column tiles per wave of the 24-row (two m-tile) launches: downs.0 / downs.1 (0.605 -> 0.57 ms)
#endif
#ifndef TVC_C48_TN
#define TVC_C48_TN 1     // 8 waves x 16 columns = 128-sample tiles: measured best (1.40 ms vs 1.51 for TN = 2 on ups.3)
#endif
    using TL = C48Tile<TVC_C48_NW, MT == 2 ? TVC_C24_TN : TVC_C48_TN, MT>;
    Conv3Args a;
    a.At = w.At;
    a.x = x;
    a.Mpad = w.Mpad;
    a.Cin = Cin;
    a.len = len;
    a.dil = dil;
    a.tiles_per_utt = (len + TL::BN - 1) / TL::BN;
    a.sc_At = f.sc_At;
    a.sh_At = f.sh_At;
    a.cond = f.cond;
    a.Ccond = f.Ccond;
    a.x2 = x2;
    a.split = x2 ? split : Cin;
    a.Krows = w.Kpad;
    dim3 g((unsigned)(a.tiles_per_utt * B));
    hipLaunchKernelGGL((conv3m48_kernel<TL, LRELU, Epi, FILM>), g, dim3(TL::NTHR), 0, s, a, ep);
}

template <bool LRELU, class Epi, bool FILM = false>
inline void conv3m48_launch(hipStream_t s, const PackedW& w, const float* x, int B, int Cin, int len, int dil, const Epi& ep,
                            const FilmOps& f = FilmOps()) {
    conv3mt_launch<3, LRELU, Epi, FILM>(s, w, x, B, Cin, len, dil, ep, f);
}

}  // namespace tvc
