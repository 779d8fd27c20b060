// k=3 dilated Conv1d (replicate padding, optional leaky_relu(0.1) pre-activation) as an implicit GEMM
// on v_mfma_f32_32x32x2_f32 with an LDS halo tile: FilterNet's Downsample/Upsample convs
// (decoder.py:143-146, 166-171) for the levels whose activations do not fit on chip as a whole block.
//
// A workgroup owns BM output channels x BN consecutive samples of ONE utterance.  K is walked in slabs
// of KC = 8 input channels: the slab's activations are staged once as Xs[8][BN + 2*dil] (clamped to
// the utterance = replicate padding, pre-activation applied on the way in) and serve all three taps
// from LDS; the weight slab is 24 contiguous rows of At[k = ci*3 + tap][m].  Each lane precomputes
// the 12 LDS offsets (ci_local * row + tap * dil) of its k-steps once, so the inner loop is
// ds_read + MFMA only.  Double-buffered slabs, one barrier per slab.
#pragma once
#include "igemm.h"

namespace tvc {

template <int WM_, int WN_, int TM_, int TN_>
struct Conv3Tile {
    static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
    static constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    static constexpr int KC = 8, KS = 24;            // channels / k-rows per slab
    static constexpr int MAXD = 27;
    static constexpr int XROW = BN + 2 * MAXD + 2;   // LDS row stride of the halo tile
    static constexpr int NW = WM * WN, NTHR = NW * 64;
    static_assert(NW == 4 || NW == 8 || NW == 12 || NW == 16, "4, 8, 12 or 16 waves per workgroup");
};

struct Conv3Args {
    const float* At;    // [Kpad][Mpad], k = ci*3 + tap
    const float* x;     // [B][Cin][len]
    int Mpad, Cin, len, dil, tiles_per_utt;
    // FiLM fused behind the conv (FILM kernels only): scale/shift = 1x1 convs of cond [B][Ccond][len]
    const float* sc_At;  // [Ccond_pad][Mpad]
    const float* sh_At;
    const float* cond;
    int Ccond;
    // conv3m48.h only: optional second input tensor for channels [split, Cin), and the number of At rows
    const float* x2 = nullptr;
    int split = 0, Krows = 0;
};

// Epilogue interface: store(b, t, m, v[4]) for 4 consecutive channels m..m+3 at (b, t); t < len guaranteed.
// F2 = 3 / 5: compile-time decimation factor of the scalar store() path (a runtime division per element costs more than
// the interpolate pass it replaces); the tile_store path of conv3s.h uses the runtime fields y2 / f2.
template <bool RES, int F2 = 0>
struct C3EpiBias {
    static constexpr bool kRes = RES;
    static constexpr bool kIgemm = false;
    float* y;
    const float* bias;
    const float* res;
    int M, len;
    // optional second output: the F.interpolate(scale_factor = 1/f2) copy the next Downsample block starts from
    // (decoder.py:148).  For f2 = 3 / 5 ATen's source coordinate f2 * (d + 0.5) - 0.5 is the integer f2*d + f2/2 (weight
    // exactly 1), for f2 = 4 it is 4d + 1.5 (weights exactly 0.5 / 0.5): the copy is a pick / a two-sample mean of y.
    float* y2 = nullptr;
    int f2 = 0;
    __device__ __forceinline__ void store(int b, int t, int m, const float v[4]) const {
        const int q = F2 > 0 ? t / F2 : 0;
        const bool pick = F2 > 0 && y2 != nullptr && t - q * F2 == (F2 >> 1);   // f2 = 4 needs two samples: tile_store only
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (m + r < M) {
                long i = ((long)b * M + m + r) * len + t;
                float o = v[r] + bias[m + r];
                if (RES) o += res[i];
                y[i] = o;
                if (pick) y2[((long)b * M + m + r) * (len / F2) + q] = o;
            }
    }
};

// conv -> FiLM -> + residual (decoder.py:94-97,181-182); film = stacked [scale ; shift] [B][2M][len]
struct C3EpiFilm {
    float* y;
    const float* bias;
    const float* film;
    const float* res;
    int M, len;
    __device__ __forceinline__ void store(int b, int t, int m, const float v[4]) const {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (m + r < M) {
                long i = ((long)b * M + m + r) * len + t;
                float h = v[r] + bias[m + r];
                float sc = film[((long)b * 2 * M + m + r) * len + t];
                float sh = film[((long)b * 2 * M + M + m + r) * len + t];
                y[i] = __fadd_rn(__fadd_rn(__fmul_rn(h, sc), sh), res[i]);
            }
    }
};

// C3EpiBias whose launch also accumulates a 1x1 conv of a second tensor into the tile (conv3s.h wants_res_conv):
// Downsample's c3 + down_res; `bias` = the two biases summed at pack time.
struct C3EpiBiasResConv : C3EpiBias<false, 0> {
    static constexpr bool kResConv = true;
};

// conv -> FiLM -> + residual with scale/shift computed in-kernel: store(b, t, m, h[4], sc[4], sh[4])
struct C3EpiFilmFused {
    static constexpr bool kIgemm = false;
    float* y;
    const float* bias;
    const float* bsc;
    const float* bsh;
    const float* res;
    int M, len;
    // conv3s.h only: res_lin > 0 -> `res` is the low-rate [B][M][res_lin] tensor and the residual is its
    // F.interpolate(scale_factor) (ATen scale float(1/scale_factor) in res_scale), evaluated in the epilogue
    int res_lin = 0;
    float res_scale = 0.f;
    __device__ __forceinline__ void store(int b, int t, int m, const float h[4], const float sc[4], const float sh[4]) const {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (m + r < M) {
                long i = ((long)b * M + m + r) * len + t;
                float hv = h[r] + bias[m + r];
                y[i] = __fadd_rn(__fadd_rn(__fmul_rn(hv, sc[r] + bsc[m + r]), sh[r] + bsh[m + r]), res[i]);
            }
    }
};

template <class TL, bool LRELU, class Epi, bool FILM = false>
__global__ __launch_bounds__(TL::NTHR) void conv3_kernel(Conv3Args a, Epi ep) {
    constexpr int BM = TL::BM, BN = TL::BN, TM = TL::TM, TN = TL::TN, KC = TL::KC, KS = TL::KS, XROW = TL::XROW, NTHR = TL::NTHR;
    __shared__ __attribute__((aligned(16))) float As[2][KS * BM];
    __shared__ __attribute__((aligned(16))) float Xs[2][KC * XROW];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / TL::WN, wn = wave % TL::WN;
    const int mtiles = a.Mpad / BM;
    const int m0 = (blockIdx.x % mtiles) * BM;
    const int nt_id = blockIdx.x / mtiles;
    const int b = nt_id / a.tiles_per_utt;
    const int t0 = (nt_id - b * a.tiles_per_utt) * BN;
    const int len = a.len, dil = a.dil;
    const int xw = BN + 2 * dil;                      // staged columns: positions t0-dil .. t0+BN+dil
    const float* xb = a.x + (long)b * a.Cin * len;

    // per-lane LDS offsets of the 12 k-steps of a slab: k = 2*ks + lh -> (ci_local, tap)
    int boff[KS / 2];
#pragma unroll
    for (int ks = 0; ks < KS / 2; ++ks) {
        int k = 2 * ks + lh;
        int cil = k / 3, tap = k - 3 * cil;
        boff[ks] = cil * XROW + tap * dil;
    }

    constexpr int A_F4 = KS * BM / 4;
    constexpr int A_PER = (A_F4 + NTHR - 1) / NTHR;
    constexpr int X_PER = (KC * XROW + NTHR - 1) / NTHR;
    float4 areg[A_PER];
    float xreg[X_PER];
    // staging map of this thread, fixed across slabs: element i -> (local channel r, staged column c)
    int xg[X_PER], xl[X_PER];
#pragma unroll
    for (int i = 0; i < X_PER; ++i) {
        int idx = tid + i * NTHR;
        int r = idx / xw, c = idx - r * xw;
        int p = t0 - dil + c;
        p = p < 0 ? 0 : (p > len - 1 ? len - 1 : p);
        xg[i] = r < KC ? r * len + p : -1;
        xl[i] = r * XROW + c;
    }

    auto load_slab = [&](int ci0) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int idx = tid + i * NTHR;
            if (A_F4 % NTHR == 0 || idx < A_F4) {
                int kk = idx / (BM / 4), c4 = idx - kk * (BM / 4);
                areg[i] = *reinterpret_cast<const float4*>(a.At + (long)(ci0 * 3 + kk) * a.Mpad + m0 + c4 * 4);
            }
        }
        const float* xc = xb + (long)ci0 * len;
#pragma unroll
        for (int i = 0; i < X_PER; ++i) {
            float v = xg[i] >= 0 ? xc[xg[i]] : 0.f;
            if (LRELU) v = v > 0.f ? v : 0.1f * v;
            xreg[i] = v;
        }
    };
    auto store_slab = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int idx = tid + i * NTHR;
            if (A_F4 % NTHR == 0 || idx < A_F4) *reinterpret_cast<float4*>(&As[buf][idx * 4]) = areg[i];
        }
#pragma unroll
        for (int i = 0; i < X_PER; ++i)
            if (xg[i] >= 0) Xs[buf][xl[i]] = xreg[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nslab = a.Cin / KC;   // Cin % 8 == 0 for every FilterNet level that uses this kernel
    load_slab(0);
    store_slab(0);
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
        const int cur = s & 1;
        const int snext = s + 1 < nslab ? s + 1 : s;
        load_slab(snext * KC);
        const float* as = As[cur];
        const float* xs = Xs[cur] + l31;
#pragma unroll
        for (int ks = 0; ks < KS / 2; ++ks) {
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = as[(2 * ks + lh) * BM + (wm * TM + i) * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = xs[boff[ks] + (wn * TN + j) * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        store_slab(cur ^ 1);
        __syncthreads();
    }

    if constexpr (FILM) {
        // ---- FiLM: two 1x1 contractions over the cond tile on the same MFMA tiles (8-row slabs, the conv's
        // LDS buffers are reused); results stay in registers next to the conv accumulators -----------------
        f32x16 asc[TM][TN], ash[TM][TN];
        constexpr int FK = 8;                          // cond channels per slab
        constexpr int FA_F4 = FK * BM / 4, FA_PER = (FA_F4 + NTHR - 1) / NTHR;
        constexpr int FB_PER = (FK * BN + NTHR - 1) / NTHR;
        const float* cb = a.cond + (long)b * a.Ccond * len;
        auto film_phase = [&](const float* Wt, f32x16 (&out)[TM][TN]) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) out[i][j][r] = 0.f;
            float4 fa[FA_PER];
            float fb[FB_PER];
            auto fload = [&](int c0) {
#pragma unroll
                for (int i = 0; i < FA_PER; ++i) {
                    int idx = tid + i * NTHR;
                    if (FA_F4 % NTHR == 0 || idx < FA_F4) {
                        int kk = idx / (BM / 4), c4 = idx - kk * (BM / 4);
                        fa[i] = *reinterpret_cast<const float4*>(Wt + (long)(c0 + kk) * a.Mpad + m0 + c4 * 4);
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
                    if (FA_F4 % NTHR == 0 || idx < FA_F4) *reinterpret_cast<float4*>(&As[buf][idx * 4]) = fa[i];
                }
#pragma unroll
                for (int i = 0; i < FB_PER; ++i) {
                    int idx = tid + i * NTHR;
                    if (idx < FK * BN) Xs[buf][idx] = fb[i];
                }
            };
            const int ns = (a.Ccond + FK - 1) / FK;
            fload(0);
            fstore(0);
            __syncthreads();
            for (int s2 = 0; s2 < ns; ++s2) {
                const int cur = s2 & 1;
                fload((s2 + 1 < ns ? s2 + 1 : s2) * FK);
                const float* as = As[cur];
                const float* bs = Xs[cur];
#pragma unroll
                for (int ks = 0; ks < FK / 2; ++ks) {
                    float av[TM], bv[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) av[i] = as[(2 * ks + lh) * BM + (wm * TM + i) * 32 + l31];
#pragma unroll
                    for (int j = 0; j < TN; ++j) bv[j] = bs[(2 * ks + lh) * BN + (wn * TN + j) * 32 + l31];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            out[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], out[i][j], 0, 0, 0);
                }
                fstore(cur ^ 1);
                __syncthreads();
            }
        };
        film_phase(a.sc_At, asc);
        film_phase(a.sh_At, ash);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int t = t0 + (wn * TN + j) * 32 + l31;
                if (t < len) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int m = m0 + (wm * TM + i) * 32 + 8 * q + 4 * lh;
                        float v[4] = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        float c1[4] = {asc[i][j][4 * q], asc[i][j][4 * q + 1], asc[i][j][4 * q + 2], asc[i][j][4 * q + 3]};
                        float c2[4] = {ash[i][j][4 * q], ash[i][j][4 * q + 1], ash[i][j][4 * q + 2], ash[i][j][4 * q + 3]};
                        ep.store(b, t, m, v, c1, c2);
                    }
                }
            }
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int t = t0 + (wn * TN + j) * 32 + l31;
                if (t < len) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int m = m0 + (wm * TM + i) * 32 + 8 * q + 4 * lh;
                        float v[4] = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        ep.store(b, t, m, v);
                    }
                }
            }
    }
}

struct FilmOps {
    const float* sc_At = nullptr;
    const float* sh_At = nullptr;
    const float* cond = nullptr;
    int Ccond = 0;
};

template <class TL, bool LRELU, class Epi, bool FILM = false>
inline void conv3_launch_t(hipStream_t s, const float* At, int Mpad, const float* x, int B, int Cin, int len, int dil, const Epi& ep,
                           const FilmOps& f = FilmOps()) {
    Conv3Args a;
    a.sc_At = f.sc_At;
    a.sh_At = f.sh_At;
    a.cond = f.cond;
    a.Ccond = f.Ccond;
    a.At = At;
    a.x = x;
    a.Mpad = Mpad;
    a.Cin = Cin;
    a.len = len;
    a.dil = dil;
    a.tiles_per_utt = (len + TL::BN - 1) / TL::BN;
    dim3 g((unsigned)((Mpad / TL::BM) * a.tiles_per_utt * B));
    hipLaunchKernelGGL((conv3_kernel<TL, LRELU, Epi, FILM>), g, dim3(TL::NTHR), 0, s, a, ep);
}

// Tile choice: BM from Mpad; BN as wide as still yields enough workgroups to fill 256 CUs several times.
template <bool LRELU, class Epi, bool FILM = false>
inline void conv3_launch(hipStream_t s, const float* At, int Mpad, const float* x, int B, int Cin, int len, int dil, const Epi& ep,
                         const FilmOps& f = FilmOps()) {
    constexpr long kEnough = 1536;
    auto blocks = [&](int BM, int BN) { return (long)(Mpad / BM) * ((len + BN - 1) / BN) * B; };
#if TVC_W8 >= 3
    using C128 = Conv3Tile<4, 4, 1, 1>;      // 128 x 128, 16 waves of 32 x 32
    using C96x256 = Conv3Tile<1, 8, 3, 1>;   // 96 x 256, 8 waves of 96 x 32
    using C64x256 = Conv3Tile<2, 8, 1, 1>;   // 64 x 256, 16 waves of 32 x 32
#elif TVC_W8
    using C128 = Conv3Tile<2, 4, 2, 1>;      // 128 x 128, 8 waves of 64 x 32
    using C96x256 = Conv3Tile<1, 8, 3, 1>;   // 96 x 256, 8 waves of 96 x 32
    using C64x256 = Conv3Tile<1, 8, 2, 1>;   // 64 x 256, 8 waves of 64 x 32
#if TVC_W8 >= 2
    using C96x128 = Conv3Tile<3, 4, 1, 1>;   // 96 x 128, 12 waves of 32 x 32
    using C128x64 = Conv3Tile<4, 2, 1, 1>;   // 128 x 64, 8 waves of 32 x 32
#else
    using C96x128 = Conv3Tile<1, 4, 3, 1>;
    using C128x64 = Conv3Tile<4, 1, 1, 2>;
#endif
#else
    using C96x128 = Conv3Tile<1, 4, 3, 1>;
    using C128x64 = Conv3Tile<4, 1, 1, 2>;
    using C128 = Conv3Tile<2, 2, 2, 2>;
    using C96x256 = Conv3Tile<1, 4, 3, 2>;
    using C64x256 = Conv3Tile<1, 4, 2, 2>;
#endif
    if (Mpad % 128 == 0) {
#ifndef TVC_BN96
#define TVC_BN96 1
#endif
        // short levels (len = 400 for a 4 s utterance): 96-wide tiles split 400 samples into 5 tiles and the
        // whole grid fits in one resident round, where 64-wide tiles need 7 tiles and a second, half-empty round
        if (TVC_BN96 && blocks(128, 128) < kEnough && (len + 95) / 96 * 96 - len < (len + 63) / 64 * 64 - len + 64)
            conv3_launch_t<Conv3Tile<4, 3, 1, 1>, LRELU, Epi, FILM>(s, At, Mpad, x, B, Cin, len, dil, ep, f);                 // 128 x 96
        else if (blocks(128, 128) >= kEnough) conv3_launch_t<C128, LRELU, Epi, FILM>(s, At, Mpad, x, B, Cin, len, dil, ep, f);   // 128 x 128
        else conv3_launch_t<C128x64, LRELU, Epi, FILM>(s, At, Mpad, x, B, Cin, len, dil, ep, f);                               // 128 x 64
    } else if (Mpad % 96 == 0) {
        if (blocks(96, 256) >= kEnough) conv3_launch_t<C96x256, LRELU, Epi, FILM>(s, At, Mpad, x, B, Cin, len, dil, ep, f);    // 96 x 256
        else conv3_launch_t<C96x128, LRELU, Epi, FILM>(s, At, Mpad, x, B, Cin, len, dil, ep, f);                               // 96 x 128
    } else if (Mpad % 64 == 0) {
        conv3_launch_t<C64x256, LRELU, Epi, FILM>(s, At, Mpad, x, B, Cin, len, dil, ep, f);                 // 64 x 256
    } else {
        conv3_launch_t<Conv3Tile<1, 4, 1, 2>, LRELU, Epi, FILM>(s, At, Mpad, x, B, Cin, len, dil, ep, f);                 // 32 x 256
    }
}

}  // namespace tvc
