// Implicit-GEMM core for every channel contraction of the path (1x1 convs, k3 dilated convs,
// forward/inverse DFT) on the gfx950 fp32 matrix pipe.
//
//   Y[m][n] = sum_k A[m][k] * Bop(k, n)         m = output channel, n = flattened (batch, time)
//
// A is a static weight, pre-transposed on the host to At[k][m] so that a K-slab of a tile is BK
// rows of BM contiguous floats (coalesced 16-B loads, 16-B LDS writes).  Bop is produced on the fly
// by a Loader functor (conv taps with replicate clamp, pre-activation, GRN scale, STFT framing, ...),
// so no im2col or activation copy ever goes through HBM.  The accumulator tile is handed to an
// Epilogue functor (bias, activation, residual, FiLM, |.|, ...) in quads of 4 consecutive channels.
//
// Pipeline: LDS double-buffered K-slabs (BK = 16), one barrier per slab; the next slab's global
// loads are issued (branch-free) before the current slab's MFMAs and land in LDS after them.
// Loaders precompute everything that depends only on the thread's column (batch base pointer,
// clamped tap offsets) once, so the per-element cost in the K loop is one mad + one load.
//
// v_mfma_f32_32x32x2_f32: exact fp32 FMA chain at the fp32 vector-peak rate (157 TF on MI355X);
// wave64 layouts: A lane l -> A[i = l&31][k = l>>5], B lane l -> B[k = l>>5][j = l&31],
// C reg r of lane l -> C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
#pragma once
#include <hip/hip_runtime.h>

namespace tvc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WM_, int WN_, int TM_, int TN_>
struct Tile {
    static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
    static constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = 16;
    static constexpr int NW = WM * WN, NTHR = NW * 64;     // 4 or 8 waves per workgroup
    static_assert(NW == 4 || NW == 8 || NW == 12 || NW == 16, "4, 8, 12 or 16 waves per workgroup");
};

// ------------------------------------------------------------------------------------------
// Loaders.  `Ctx ctx(n, ncols, T)` is built once per thread for its column n = (b, t);
// `float get(ctx, k)` returns Bop(k, n).

struct ColCtx {
    const float* base;  // &x[b][0][0] (+ t for plain loaders)
    int b;
    bool ok;
};

// X[b][k][t] (1x1 conv input), batch stride given.
struct LoadPlain {
    const float* x;
    int K, T;
    long bstride;
    typedef ColCtx Ctx;
    __device__ __forceinline__ Ctx ctx(int n, int ncols, int Tt) const {
        Ctx c;
        c.ok = n < ncols;
        int nn = c.ok ? n : 0;
        c.b = nn / Tt;
        c.base = x + c.b * bstride + (nn - c.b * Tt);
        return c;
    }
    __device__ __forceinline__ float get(const Ctx& c, int k) const {
        return (c.ok && k < K) ? c.base[k * T] : 0.f;
    }
};

// GRN folded into the 1x1-conv input (convnext.py:31-34): gamma*(x*nx) + beta + x
// = x * s[b][k] + beta[k] with s = 1 + gamma*nx; the beta term is constant per output channel and
// lives in the packed bias (bias + W.beta), so the loader is one multiply.
struct LoadScaled {
    const float* x;
    const float* s;  // [B][K]
    int K, T;
    struct Ctx {
        const float* base;
        const float* srow;
        bool ok;
    };
    __device__ __forceinline__ Ctx ctx(int n, int ncols, int Tt) const {
        Ctx c;
        c.ok = n < ncols;
        int nn = c.ok ? n : 0;
        int b = nn / Tt;
        c.base = x + (long)b * K * T + (nn - b * Tt);
        c.srow = s + (long)b * K;
        return c;
    }
    __device__ __forceinline__ float get(const Ctx& c, int k) const {
        return (c.ok && k < K) ? c.base[k * T] * c.srow[k] : 0.f;
    }
};

// 3-tap dilated conv input with replicate padding, optional leaky_relu(0.1) pre-activation.
// K is channel-major, k = ci*3 + tap (PyTorch's [cout][cin][tap] order): the three taps of a channel
// sit in the same K-slab and hit the same cache lines.
template <bool LRELU>
struct LoadConv3 {
    const float* x;
    int Cin, T, dil;
    long bstride;
    struct Ctx {
        const float* base;
        int off[3];
        bool ok;
    };
    __device__ __forceinline__ Ctx ctx(int n, int ncols, int Tt) const {
        Ctx c;
        c.ok = n < ncols;
        int nn = c.ok ? n : 0;
        int b = nn / Tt, t = nn - b * Tt;
        c.base = x + b * bstride;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            int tt = t + (j - 1) * dil;
            c.off[j] = tt < 0 ? 0 : (tt >= T ? T - 1 : tt);
        }
        return c;
    }
    __device__ __forceinline__ float get(const Ctx& c, int k) const {
        int ci = k / 3;
        if (!(c.ok && ci < Cin)) return 0.f;
        int tap = k - 3 * ci;
        int o = tap == 0 ? c.off[0] : (tap == 1 ? c.off[1] : c.off[2]);
        float v = c.base[ci * T + o];
        if (LRELU) v = v > 0.f ? v : 0.1f * v;
        return v;
    }
};

// FilterNet downs[0]: 3-tap conv over cat[source (16 ch), energy (1 ch)] (decoder.py:224,227);
// k = ci*3 + tap with Cin = 17.
struct LoadConvCat17 {
    const float* src;     // [B][16][L]
    const float* energy;  // [B][1][L]
    int T;
    struct Ctx {
        const float* sb;
        const float* eb;
        int off[3];
        bool ok;
    };
    __device__ __forceinline__ Ctx ctx(int n, int ncols, int Tt) const {
        Ctx c;
        c.ok = n < ncols;
        int nn = c.ok ? n : 0;
        int b = nn / Tt, t = nn - b * Tt;
        c.sb = src + (long)b * 16 * T;
        c.eb = energy + (long)b * T;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            int tt = t + j - 1;
            c.off[j] = tt < 0 ? 0 : (tt >= T ? T - 1 : tt);
        }
        return c;
    }
    __device__ __forceinline__ float get(const Ctx& c, int k) const {
        int ci = k / 3;
        if (!(c.ok && ci < 17)) return 0.f;
        int tap = k - 3 * ci;
        int o = tap == 0 ? c.off[0] : (tap == 1 ? c.off[1] : c.off[2]);
        return ci < 16 ? c.sb[ci * T + o] : c.eb[o];
    }
};

// A plain [K][ncols] operand matrix (column n contiguous along lanes): the folded STFT frames.
struct LoadMatrix {
    const float* x;
    int K;
    long ld;
    struct Ctx {
        const float* base;
        bool ok;
    };
    __device__ __forceinline__ Ctx ctx(int n, int ncols, int) const {
        Ctx c;
        c.ok = n < ncols;
        c.base = x + (c.ok ? n : 0);
        return c;
    }
    __device__ __forceinline__ float get(const Ctx& c, int k) const { return (c.ok && k < K) ? c.base[k * ld] : 0.f; }
};

// ------------------------------------------------------------------------------------------
// Epilogues: `store(n, m, v)` gets 4 consecutive output channels m..m+3 of column n.

enum { ACT_NONE = 0, ACT_GELU = 1, ACT_ELU1 = 2 };

__device__ __forceinline__ float act_apply(float o, int act) {
    if (act == ACT_GELU) return 0.5f * o * (1.f + erff(o * 0.70710678118654752f));
    if (act == ACT_ELU1) return (o > 0.f ? o : (expf(o) - 1.f)) + 1.f;
    return o;
}

template <int ACT, bool RES>
struct EpiBias {
    static constexpr bool kIgemm = true;   // store(n, m, v[4]) interface (also usable from the split-precision kernel)
    float* y;
    const float* bias;
    const float* res;
    int M, T, ncols;
    long y_bs, res_bs;  // batch strides
    __device__ __forceinline__ void store(int n, int m, const float v[4]) const {
        if (n >= ncols) return;
        int b = n / T, t = n - b * T;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (m + r < M) {
                float o = act_apply(v[r] + bias[m + r], ACT);
                if (RES) o += res[b * res_bs + (long)(m + r) * T + t];
                y[b * y_bs + (long)(m + r) * T + t] = o;
            }
        }
    }
};

// conv -> FiLM -> + residual  (decoder.py:94-97,181-182): (h*scale + shift) + res,
// scale/shift = rows [0,M) / [M,2M) of the stacked FiLM 1x1 output `film` [B][2M][T].
struct EpiFilm {
    float* y;
    const float* bias;
    const float* film;
    const float* res;
    int M, T, ncols;
    __device__ __forceinline__ void store(int n, int m, const float v[4]) const {
        if (n >= ncols) return;
        int b = n / T, t = n - b * T;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (m + r < M) {
                long i = ((long)b * M + m + r) * T + t;
                float h = v[r] + bias[m + r];
                float sc = film[((long)b * 2 * M + m + r) * T + t];
                float sh = film[((long)b * 2 * M + M + m + r) * T + t];
                y[i] = __fadd_rn(__fadd_rn(__fmul_rn(h, sc), sh), res[i]);
            }
        }
    }
};

// content_in(content) + energy_in(e) + f0_in(log(relu(f0)+1e-6))   (decoder.py:128, :223)
// e / lf0 are per-(b,t) scalars feeding 1->M 1x1 convs; `e` may be null (FilterNet has no energy).
struct EpiSumCond {
    static constexpr bool kIgemm = true;   // store(n, m, v[4]) interface (also usable from the split-precision kernel)
    float* y;
    const float* bias;
    const float* e;     // [B][T] or null
    const float* f0;    // [B][T]
    const float* we;
    const float* be;
    const float* wf;
    const float* bf;
    int M, T, ncols;
    __device__ __forceinline__ void store(int n, int m, const float v[4]) const {
        if (n >= ncols) return;
        int b = n / T, t = n - b * T;
        float lf = logf(fmaxf(f0[n], 0.f) + 1e-6f);
        float ev = e ? e[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (m + r < M) {
                float o = v[r] + bias[m + r];
                if (e) o = __fadd_rn(o, __fadd_rn(__fmul_rn(we[m + r], ev), be[m + r]));
                o = __fadd_rn(o, __fadd_rn(__fmul_rn(wf[m + r], lf), bf[m + r]));
                y[((long)b * M + m + r) * T + t] = o;
            }
        }
    }
};

// |STFT| in two passes: pass 1 parks Re X in `spec`; pass 2 holds Im X in its accumulators and
// overwrites spec with sqrt(re^2 + im^2).  Rows are bins.
template <bool FINAL>
struct EpiStftPart {
    static constexpr bool kIgemm = true;   // store(n, m, v[4]) interface (also usable from the split-precision kernel)
    float* spec;
    int T, ncols;
    __device__ __forceinline__ void store(int n, int m, const float v[4]) const {
        if (n >= ncols) return;
        int b = n / T, t = n - b * T;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (m + r < 961) {
                float* p = spec + ((long)b * 961 + m + r) * T + t;
                if (FINAL) {
                    float re = *p;
                    *p = sqrtf(re * re + v[r] * v[r]);
                } else {
                    *p = v[r];
                }
            }
    }
};

// inverse real DFT frames, two passes: pass 1 writes E[n] (n = 0..960) into frames[col][n];
// pass 2 holds O[n] (rows m = n-1, n = 1..959) and finishes x[n] = E - O, x[1920-n] = E + O in place.
template <bool FINAL>
struct EpiFramesPart {
    static constexpr bool kIgemm = true;
    float* frames;
    int ncols;
    __device__ __forceinline__ void store(int n, int m, const float v[4]) const {
        if (n >= ncols) return;
        float* fr = frames + (long)n * 1920;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (FINAL) {
                int nn = m + r + 1;
                if (nn <= 959) {
                    float e = fr[nn];
                    fr[nn] = e - v[r];
                    fr[1920 - nn] = e + v[r];
                }
            } else if (m + r <= 960) {
                fr[m + r] = v[r];
            }
        }
    }
};

// ------------------------------------------------------------------------------------------
template <class TL, class Loader, class Epi>
__global__ __launch_bounds__(TL::NTHR) void igemm_kernel(const float* __restrict__ At, int Mpad, int Kpad,
                                                    int ncols, int T, Loader ld, Epi ep) {
    constexpr int BM = TL::BM, BN = TL::BN, BK = TL::BK;
    constexpr int TM = TL::TM, TN = TL::TN, NTHR = TL::NTHR;
    // B staging uses the first B_THR threads: the largest multiple of BN whose row step divides BK
    constexpr int B_THR = (BK % (NTHR / BN) == 0 && NTHR % BN == 0) ? NTHR : (NTHR >= 4 * BN ? 4 * BN : (NTHR >= 2 * BN ? 2 * BN : BN));
    static_assert(BN <= NTHR && B_THR % BN == 0 && BK % (B_THR / BN) == 0, "one column per staging thread");
    __shared__ __attribute__((aligned(16))) float As[2][BK * BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / TL::WN, wn = wave % TL::WN;
    // 1-D grid, m-tiles fastest: the workgroups that share one B column tile are dispatched together
    const int mtiles = Mpad / BM;
    const int m0 = (blockIdx.x % mtiles) * BM;
    const int n0 = (blockIdx.x / mtiles) * BN;

    constexpr int A_F4 = BK * BM / 4;                 // float4s per A slab
    constexpr int A_PER = (A_F4 + NTHR - 1) / NTHR;
    constexpr int B_RSTEP = B_THR / BN;                 // thread owns column tid % BN, rows brow0 + j*B_RSTEP
    constexpr int B_ROWS_PER = BK / B_RSTEP;

    const int bcol = tid % BN;
    const int brow0 = tid / BN;
    const typename Loader::Ctx lc = ld.ctx(n0 + bcol, ncols, T);

    float4 areg[A_PER];
    float breg[B_ROWS_PER];

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto load_slab = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int idx = tid + i * NTHR;
            if (A_F4 % NTHR == 0 || idx < A_F4) {
                int kk = idx / (BM / 4), c4 = idx - kk * (BM / 4);
                areg[i] = *reinterpret_cast<const float4*>(At + (long)(k0 + kk) * Mpad + m0 + c4 * 4);
            }
        }
        if (B_THR == NTHR || tid < B_THR) {
#pragma unroll
            for (int j = 0; j < B_ROWS_PER; ++j) breg[j] = ld.get(lc, k0 + brow0 + j * B_RSTEP);
        }
    };
    auto store_slab = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int idx = tid + i * NTHR;
            if (A_F4 % NTHR == 0 || idx < A_F4) *reinterpret_cast<float4*>(&As[buf][idx * 4]) = areg[i];
        }
        if (B_THR == NTHR || tid < B_THR) {
#pragma unroll
            for (int j = 0; j < B_ROWS_PER; ++j) Bs[buf][(brow0 + j * B_RSTEP) * BN + bcol] = breg[j];
        }
    };

    const int nk = Kpad / BK;
    load_slab(0);
    store_slab(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const int knext = kt + 1 < nk ? kt + 1 : kt;   // last iteration re-reads its own slab (no branch)
        load_slab(knext * BK);
        const float* as = As[cur];
        const float* bs = Bs[cur];
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            const int k = 2 * ks + lh;
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = as[k * BM + (wm * TM + i) * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = bs[k * BN + (wn * TN + j) * 32 + l31];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        store_slab(cur ^ 1);   // the other buffer was last read before the previous barrier
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + l31;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + (wm * TM + i) * 32 + 8 * q + 4 * lh;
                float v[4] = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2],
                              acc[i][j][4 * q + 3]};
                ep.store(n, m, v);
            }
        }
}

// Host-side launch.  The tile shape is picked from Mpad and from the grid it yields: these GEMMs are
// small for a 256-CU chip (N = B*T is 12 800 columns for 64 x 4 s), so a 128 x 128 tiling can leave
// 300 workgroups for 1024 resident slots; smaller tiles trade operand reuse for a full machine.
template <class TL, class Loader, class Epi>
inline void igemm_launch_t(hipStream_t s, const float* At, int Mpad, int Kpad, int ncols, int T,
                           const Loader& ld, const Epi& ep) {
    dim3 g((unsigned)((Mpad / TL::BM) * ((ncols + TL::BN - 1) / TL::BN)));
    hipLaunchKernelGGL((igemm_kernel<TL, Loader, Epi>), g, dim3(TL::NTHR), 0, s, At, Mpad, Kpad, ncols, T, ld, ep);
}

inline long igemm_blocks(int Mpad, int ncols, int BM, int BN) { return (long)(Mpad / BM) * ((ncols + BN - 1) / BN); }

template <class Loader, class Epi>
inline void igemm_launch(hipStream_t s, const float* At, int Mpad, int Kpad, int ncols, int T,
                         const Loader& ld, const Epi& ep) {
    if (ncols <= 0) return;
    constexpr long kEnough = 1536;   // ~6 workgroups per CU
#ifndef TVC_W8
#define TVC_W8 2                     // many-wave workgroups with small per-wave tiles (<= 32 accumulator registers,
                                     // 8 or 12 waves): measured 10-30 % faster than 4 waves x 64 accumulators
#endif
#ifndef TVC_IG64
#define TVC_IG64 0   // 8-wave 64 x 128 measured 8 % slower than 4 waves of 32 x 64 on the DFT GEMMs
#endif
#if TVC_IG64
    using T64x128 = Tile<2, 4, 1, 1>;   // 64 x 128, 8 waves of 32 x 32
#else
    using T64x128 = Tile<2, 2, 1, 2>;   // 64 x 128, 4 waves of 32 x 64
#endif
#if TVC_W8 >= 3
    using T128 = Tile<4, 4, 1, 1>;   // 128 x 128, 16 waves of 32 x 32
    using T64x256 = Tile<2, 8, 1, 1>;
#elif TVC_W8
    using T128 = Tile<2, 4, 2, 1>;   // 128 x 128, wave = 64 x 32
    using T64x256 = Tile<1, 8, 2, 1>;
#else
    using T128 = Tile<2, 2, 2, 2>;   // 128 x 128, wave = 64 x 64
    using T64x256 = Tile<1, 4, 2, 2>;
#endif
    if (Mpad % 128 == 0) {
        if (igemm_blocks(Mpad, ncols, 128, 128) >= kEnough) igemm_launch_t<T128>(s, At, Mpad, Kpad, ncols, T, ld, ep);
        else if (igemm_blocks(Mpad, ncols, 64, 128) >= kEnough) igemm_launch_t<T64x128>(s, At, Mpad, Kpad, ncols, T, ld, ep);
        else igemm_launch_t<Tile<2, 2, 1, 1>>(s, At, Mpad, Kpad, ncols, T, ld, ep);
    } else if (Mpad % 96 == 0) {
#ifndef TVC_IG12
#define TVC_IG12 1
#endif
#if TVC_IG12
        igemm_launch_t<Tile<3, 4, 1, 1>>(s, At, Mpad, Kpad, ncols, T, ld, ep);   // 96 x 128, 12 waves of 32 x 32
#else
        igemm_launch_t<Tile<1, 4, 3, 1>>(s, At, Mpad, Kpad, ncols, T, ld, ep);   // 96 x 128
#endif
    } else if (Mpad % 64 == 0) {
        if (igemm_blocks(Mpad, ncols, 64, 256) >= kEnough) igemm_launch_t<T64x256>(s, At, Mpad, Kpad, ncols, T, ld, ep);
        else if (igemm_blocks(Mpad, ncols, 64, 128) >= kEnough) igemm_launch_t<T64x128>(s, At, Mpad, Kpad, ncols, T, ld, ep);
        else igemm_launch_t<Tile<2, 2, 1, 1>>(s, At, Mpad, Kpad, ncols, T, ld, ep);
    } else {
        igemm_launch_t<Tile<1, 4, 1, 2>>(s, At, Mpad, Kpad, ncols, T, ld, ep);   // 32 x 256
    }
}

}  // namespace tvc
