// ConvNeXt-v2 layer (convnext.py:38-58) as TWO launches around GRN's global-time norm:
//
//   cnx1_kernel   x -> [depthwise k7 dilated conv -> LayerNorm(C)] -> c2 (C -> 2C) -> GELU -> h, and sum_t h^2 of its column tile -> gp
//   cnx2_kernel   [|| h[b][c][:] ||_2 from the utterance's tile sums -> GRN factors] -> c3 (2C -> C) over h * factor -> + bias' + x -> x      (in place)
//   (utterances of more than CNX_GP_INLINE tiles: grn_tiles_kernel adds the tile sums up once, in the same order, between the two)
//
// Same arithmetic as the five launches they replace (dwconv_ln, gemm_s2 / EpiBias<GELU>, grn_norm, grn_finalize, gemm_s2 SCALED /
// EpiBias<RES>), except for the order in which GRN's sum over time is added up: the LayerNorm moments are summed in
// dwconv_ln_kernel's order (16 channel classes c % 16, each in ascending order, then the 16 partial sums in ascending order), the
// K16 steps of a contraction are walked in ascending order with the three part-products in conv3s.h's order, GRN's mean in
// grn_finalize_kernel's order, and the epilogues are gemm_epi.h's.  A column's arithmetic never looks at another column, and the
// time sum's order is a function of the utterance's own length (its equal-width tiles of <= 64 columns, ascending): an utterance
// comes out bit for bit the same whatever the batch, the launch geometry or the row split.
//
// Schedule: the B operand is STATIONARY.  A workgroup owns <= 64 columns of ONE utterance and ALL K input channels of them: the
// prologue leaves the whole split operand tile in LDS (C x 64 x 4 B = 96 KiB at C = 384) and no barrier follows it.  Every wave then
// walks its own 32-row m-tiles of the weight image: the A fragments come straight from global memory - the image is already in MFMA
// lane order, one 16-byte load per lane and piece, no LDS round trip, a 4-step register ring keeps them ahead of the MFMAs - and the B
// fragments are 16-byte LDS reads of the resident tile.  The gemm_s2 pipeline these launches ran on spends 1 650 of a slab's 3 800
// cycles at its barrier (S_TRACE stamps); here waves never meet after the prologue, one wave's GELU / store
// epilogue runs under the other wave's MFMAs on the same SIMD, and the LayerNorm output (cnx1) never exists in HBM.
// K = 768 (cnx2 at C = 384) does not fit the LDS at 64 columns: the operand tile is staged in two K halves with the accumulators
// kept across them (one m-tile per wave, twelve waves).
// Small launches (a streaming block, one utterance) split the rows of the weight image over blockIdx.z so that the chip is covered;
// the prologue is then recomputed per row block, which costs nothing on an idle chip.
#pragma once
#include "conv3s.h"
#include "gemm_epi.h"

namespace tvc {

struct CnxArgs {
    float* x;            // [C][..] layer input (cnx1) / residual and output (cnx2), utterance b at + b * C * T (equal lengths) or + pre[b] (ragged)
    float* h;            // [2C][..] hidden tensor
    int T, rs;           // utterance length and row stride (ragged: T unused, rs = the batch's frames)
    RagDev rg;
    const uint4* A6;     // split weight image of the 1x1 [K16 step][m-tile][part][lane][8 fp16]
    const float* wsc;    // its per-m-tile scales
    const float* bias;
    int MT, mt_per_wg;   // m-tiles of the image / per workgroup (blockIdx.z)
    // cnx1
    const float *dw_w, *dw_b, *ln_g, *ln_b;
    int dil;
    // cnx2
    float* gp;           // [B][gp_tiles][2C] GRN partials: sum of h^2 over one column tile's valid columns (cnx1 writes, cnx2 reads)
    int gp_tiles;        // tile slots per utterance (the longest utterance's tile count)
    int gp_sum;          // cnx2: 0 = add this utterance's own tiles up in ascending order; 1 = grn_tiles_kernel already did, into slot 0
    const float* grn_g;  // [2C]
    float* amax_y;       // optional per-utterance |max| slot of the output x (zeroed by the caller)
#ifdef CNX_TRACE
    unsigned long long* trace;   // diagnostic build (tools/micro/cnx_bench.hip): s_memtime stamps of two waves of workgroup CNX_TRACE
#endif
};
#ifdef CNX_TRACE
#define CNX_STAMP(id)                                                                                   \
    do {                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        if (a.trace && (int)(blockIdx.x + gridDim.x * blockIdx.y) == CNX_TRACE && (tid & 255) == 0 && tid < 512) {   \
            unsigned long long t_;                                                                      \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
            a.trace[(tid >> 8) * 32 + (id)] = t_;                                                       \
        }                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                              \
    } while (0)
#else
#define CNX_STAMP(id) do {} while (0)
#endif

constexpr int CNX_GP_INLINE = 16;   // tile sums a cnx2 workgroup adds up itself
constexpr int CNX_PD = 4;      // A-fragment ring depth (K16 steps in flight)
#ifndef CNX_ABL
#define CNX_ABL 0                // what-if builds of tools/micro/cnx_bench.hip (timing only, wrong results)
#endif
typedef float f32x4s_t __attribute__((ext_vector_type(4)));

// hi / lo += (A6 pieces of m-tile mt) x (the resident operand tile Ys), K16 steps [k_begin, k_begin + KSL) of the image against local
// steps [0, KSL) of Ys.  ring = the A fragments of the next CNX_PD steps (already requested); after the last step the ring holds the
// first steps of (mt_next, k_next) - the next item or the next K pass.
struct CnxNoPost {
    __device__ __forceinline__ void operator()(int) const {}
};
// post(k) runs behind step k's MFMAs, inside the step's scheduling region: the previous item's epilogue rides in the matrix pipe's shadow
template <int NT, int NC, int KSL, class Post = CnxNoPost>
__device__ __forceinline__ void cnx_mma(f32x16 (&hi)[NT], f32x16 (&lo)[NT], u32x4 (&ring)[CNX_PD][2], const uint4* __restrict__ A6, int MT, int mt, int k_begin,
                                        int mt_next, int k_next, const uint4* Ys, int lane, const Post& post = Post()) {
    static_assert(KSL % CNX_PD == 0, "ring depth divides the K walk");
    const int l31 = lane & 31, lh = lane >> 5;
    const uint4* yb = Ys + lh * NC + l31;
    u32x4 bq[2][NT][2];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p) bq[0][j][p] = *reinterpret_cast<const u32x4*>(yb + (p * 2) * NC + j * 32);
#pragma unroll
    for (int k0 = 0; k0 < KSL; k0 += CNX_PD) {
#pragma unroll
        for (int u = 0; u < CNX_PD; ++u) {
            const int k = k0 + u, fb = u & 1;
            const f16x8 a0 = __builtin_bit_cast(f16x8, ring[u][0]), a1 = __builtin_bit_cast(f16x8, ring[u][1]);
            // this slot's next occupant: CNX_PD steps ahead, rolling over into the next item / pass
            int kn = k + CNX_PD, mtl = mt, kg = k_begin;
            if (kn >= KSL) {
                kn -= KSL;
                mtl = mt_next;
                kg = k_next;
            }
            const uint4* ab = A6 + ((long)(kg + kn) * MT + mtl) * kPU4;
            if (!(CNX_ABL & 16)) {
                ring[u][0] = ldg_so4(ab, 16u * (unsigned)lane);
                ring[u][1] = ldg_so4(ab, 16u * (unsigned)(64 + lane));
            }
            // next step's B fragments (the walk's last step reads step 0 again: harmless)
            const int kb = k + 1 < KSL ? k + 1 : 0;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int p = 0; p < 2; ++p) bq[fb ^ 1][j][p] = *reinterpret_cast<const u32x4*>(yb + ((kb * 2 + p) * 2) * NC + j * 32);
#pragma unroll
            for (int j = 0; j < NT; ++j) lo[j] = TVC_MFMA16(a1, __builtin_bit_cast(f16x8, bq[fb][j][0]), lo[j]);
#pragma unroll
            for (int j = 0; j < NT; ++j) hi[j] = TVC_MFMA16(a0, __builtin_bit_cast(f16x8, bq[fb][j][0]), hi[j]);
#pragma unroll
            for (int j = 0; j < NT; ++j) lo[j] = TVC_MFMA16(a0, __builtin_bit_cast(f16x8, bq[fb][j][1]), lo[j]);
            post(k);
            __builtin_amdgcn_sched_barrier(0);      // the ring's loads and the next step's fragment reads stay in the step that issues them (the scheduler sank them to their uses)
        }
    }
}

template <int C, int NT>
struct Cnx1 {
    static constexpr int WAVES = 8, NTHR = WAVES * 64, NC = 32 * NT, KS = C / 16, CPT = C / 16, VW = NT;   // VW = channel classes (c % 16) per thread
    static constexpr int YS_U4 = KS * 4 * NC;
    static constexpr int LDS_BYTES = YS_U4 * 16 + (C * 8 + 2 * C + 16 * NC) * 4;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

template <int C, int NT>
__global__ __launch_bounds__((Cnx1<C, NT>::NTHR)) void cnx1_kernel(CnxArgs a) {
    using CF = Cnx1<C, NT>;
    constexpr int NC = CF::NC, KS = CF::KS, CPT = CF::CPT, VW = CF::VW, WAVES = CF::WAVES, NTHR = CF::NTHR;
    extern __shared__ __attribute__((aligned(16))) uint4 smem_c1[];
    uint4* Ys = smem_c1;                                            // [K16 step][part][8-channel half][column]; before that the conv outputs in fp32, [C][NC]
    float* scr = reinterpret_cast<float*>(smem_c1);
    float* Wl = reinterpret_cast<float*>(smem_c1 + CF::YS_U4);      // [C][8]: seven taps + bias
    float* Gl = Wl + C * 8;                                         // gamma [C], beta [C]
    float* red = Gl + 2 * C;                                        // [16][NC]
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    int Tb = a.T;
    long xoff = (long)b * C * a.T, hoff = (long)b * 2 * C * a.T;
    if (a.rg.tb) {
        Tb = a.rg.tb[b];
        xoff = hoff = a.rg.pre[b];
    }
    const int ntu = (Tb + NC - 1) / NC;
    if ((int)blockIdx.x >= ntu) return;
    const int TW = (Tb + ntu - 1) / ntu;                            // this utterance's tiles are equally wide
    const int t0 = blockIdx.x * TW;
    const int tw = Tb - t0 < TW ? Tb - t0 : TW;                     // valid columns (>= 1)
    const int rs = a.rs;
    const float* xb = a.x + xoff;

    for (int i = tid; i < C; i += NTHR) {
#pragma unroll
        for (int j = 0; j < 7; ++j) Wl[i * 8 + j] = a.dw_w[i * 7 + j];
        Wl[i * 8 + 7] = a.dw_b[i];
        Gl[i] = a.ln_g[i];
        Gl[C + i] = a.ln_b[i];
    }
    CNX_STAMP(0);
    __syncthreads();
    CNX_STAMP(1);

    // ---- depthwise conv + LayerNorm moments: thread = (column, channel class(es) v = c % 16), dwconv_ln_kernel's arithmetic ----
    const int col = NT == 2 ? lane : l31;
    const int tc = t0 + (col < tw ? col : tw - 1);                  // columns past the tile's end repeat its last one (never stored)
    unsigned tt[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        int q = tc + (j - 3) * a.dil;
        q = q < 0 ? 0 : (q >= Tb ? Tb - 1 : q);
        tt[j] = 4u * (unsigned)q;
    }
    float v[VW][CPT];
#pragma unroll
    for (int u = 0; u < VW; ++u) {
        const int vw = NT == 2 ? wave + 8 * u : wave + 8 * lh;
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = vw + i * 16;
            const f32x4s_t w0 = *reinterpret_cast<const f32x4s_t*>(Wl + c * 8), w1 = *reinterpret_cast<const f32x4s_t*>(Wl + c * 8 + 4);
            const float wj[7] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2]};
            float acc = w1[3];
            if (NT == 2) {
                const float* xr = xb + (long)c * rs;                // wave-uniform row
#pragma unroll
                for (int j = 0; j < 7; ++j) acc = fmaf(wj[j], (CNX_ABL & 1) ? (float)tt[j] : ldg_so(xr, tt[j]), acc);
            } else {
                const char* xr = reinterpret_cast<const char*>(xb + (long)c * rs);
#pragma unroll
                for (int j = 0; j < 7; ++j) acc = fmaf(wj[j], *reinterpret_cast<const float*>(xr + tt[j]), acc);
            }
            v[u][i] = acc;
            sum += acc;
            scr[c * NC + col] = acc;                                // for the second pass's thread mapping (8 consecutive channels per lane)
        }
        red[vw * NC + col] = sum;
    }
    CNX_STAMP(2);
    __syncthreads();
    CNX_STAMP(3);
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += red[w * NC + col];
    const float mean = tot / (float)C;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < VW; ++u) {
        const int vw = NT == 2 ? wave + 8 * u : wave + 8 * lh;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const float d = v[u][i] - mean;
            sq = fmaf(d, d, sq);
        }
        red[vw * NC + col] = sq;
    }
    __syncthreads();
    float tot2 = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot2 += red[w * NC + col];
    const float var = tot2 / (float)C;
    const float rstd = 1.f / sqrtf(var + 1e-5f);
    CNX_STAMP(4);

    // ---- normalise, split, operand tile: wave w rewrites the 16-channel blocks w, w + 8, ... in place (a block's fp32 values and
    // its four fragment rows per column are the same 64 * NC bytes; a wave's LDS accesses execute in order) -------------------
    for (int i = wave; i < KS; i += WAVES) {
        constexpr int NQ = NT == 2 ? 16 : 8;
        const int c0 = 16 * i + (NT == 2 ? 0 : 8 * lh);
        float y[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) y[q] = scr[(c0 + q) * NC + col];
#pragma unroll
        for (int q = 0; q < NQ; ++q) y[q] = fmaf((y[q] - mean) * rstd, Gl[c0 + q], Gl[C + c0 + q]);
#pragma unroll
        for (int hh = 0; hh < NQ / 8; ++hh) {
            const float y8[8] = {y[8 * hh], y[8 * hh + 1], y[8 * hh + 2], y[8 * hh + 3], y[8 * hh + 4], y[8 * hh + 5], y[8 * hh + 6], y[8 * hh + 7]};
            uint4 p1, p2;
            split8(y8, p1, p2);
            const int half = NT == 2 ? hh : lh;
            Ys[((i * 2 + 0) * 2 + half) * NC + col] = p1;
            Ys[((i * 2 + 1) * 2 + half) * NC + col] = p2;
        }
    }
    CNX_STAMP(5);
    __syncthreads();
    CNX_STAMP(6);

    // ---- c2: every wave walks its own m-tiles; no barrier from here on ----------------------------------------------------
    const int mt_lo = blockIdx.z * a.mt_per_wg;
    const int mt_hi = mt_lo + a.mt_per_wg < a.MT ? mt_lo + a.mt_per_wg : a.MT;
    int mt = mt_lo + wave;
    if (mt >= mt_hi) return;
    u32x4 ring[CNX_PD][2];
#pragma unroll
    for (int u = 0; u < CNX_PD; ++u) {
        const uint4* ab = a.A6 + ((long)u * a.MT + mt) * kPU4;
        ring[u][0] = ldg_so4(ab, 16u * (unsigned)lane);
        ring[u][1] = ldg_so4(ab, 16u * (unsigned)(64 + lane));
    }
    float* hb = a.h + hoff;
    // A wave's epilogue (GELU, 32 stores) is as long as its MFMA walk, and a wave that stores and then waits for its next A
    // fragments waits for the stores too (one in-order counter).  So the epilogue of item i is software-pipelined into the walk of
    // item i + 1: finished values wait in `pend` and leave a few per K16 step, behind that step's MFMAs.
    constexpr int NPEND = NT * 16, EPS = ((NPEND + KS - 1) / KS + 1) / 2 * 2;      // values leaving per K16 step (pairs)
    float pend[NPEND];
    int pmt = mt;
    // Columns past the tile's end hold copies of its last column (the prologue clamps), so their lanes compute that column's values bit
    // for bit and may store them to ITS address: every lane stores, no exec-masked block sits between the MFMAs and the epilogue
    // arithmetic, and the wait counters stay exact (a conditional store made every later fragment wait drain the stores).
    unsigned oo[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = j * 32 + l31;
        oo[j] = 4u * (unsigned)(4 * lh * rs + t0 + (n < tw ? n : tw - 1));
    }
    // GRN's row norms need sum_t h^2 over the whole utterance.  This tile's share - in a FIXED order, so that an utterance's result
    // never depends on the batch around it: a lane adds its own column of n-tile 0, then of n-tile 1 (columns past the tile's end
    // add nothing), five xor steps add the 32 columns of a half wave - leaves for gp[b][tile][row] right behind the
    // item's last store; cnx2 adds an utterance's tiles in ascending order.
    bool live[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) live[j] = j * 32 + l31 < tw;
    float sq[16];
    float* gpb = a.gp + ((long)b * a.gp_tiles + blockIdx.x) * (2 * C) + ((l31 >> 1) & 3) + 8 * (l31 >> 3) + 4 * lh;
    auto finish2 = [&](int e) __attribute__((always_inline)) {      // values e, e + 1 (same n-tile: e is even)
        const int j = e >> 4;
        f32x2e v = {pend[e], pend[e + 1]};
        asm volatile("" : "+v"(v));       // pins the arithmetic to the step that stores it (free-floating, all 32 GELUs were hoisted to the top of the walk and spilled)
        const f32x2e o = (CNX_ABL & 4) ? v : gelu_pair(v);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = (e + i) & 15;
            if (!((CNX_ABL & 8) && o[i] != 1.2345f)) stg_so(hb + (long)(pmt * 32 + (r & 3) + 8 * (r >> 2)) * rs, oo[j], o[i]);
            if (j == 0) sq[r] = live[0] ? o[i] * o[i] : 0.f;
            else sq[r] = live[j] ? fmaf(o[i], o[i], sq[r]) : sq[r];
        }
        if (e == NPEND - 2) {
            // reduce-scatter over the 32 columns of a half wave: each xor step halves the rows a lane still carries; lane l31 ends
            // up with row value l31 >> 1 (both lanes of a pair hold and store it: no exec-masked block)
            const bool b4 = l31 & 16, b3 = l31 & 8, b2 = l31 & 4, b1 = l31 & 2;
            float w8[8], w4[4], w2[2];
#pragma unroll
            for (int i = 0; i < 8; ++i) w8[i] = (b4 ? sq[i + 8] : sq[i]) + __shfl_xor(b4 ? sq[i] : sq[i + 8], 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) w4[i] = (b3 ? w8[i + 4] : w8[i]) + __shfl_xor(b3 ? w8[i] : w8[i + 4], 8);
#pragma unroll
            for (int i = 0; i < 2; ++i) w2[i] = (b2 ? w4[i + 2] : w4[i]) + __shfl_xor(b2 ? w4[i] : w4[i + 2], 4);
            float w1 = (b1 ? w2[1] : w2[0]) + __shfl_xor(b1 ? w2[0] : w2[1], 2);
            w1 += __shfl_xor(w1, 1);
            gpb[pmt * 32] = w1;
        }
    };
    auto post = [&](int k) __attribute__((always_inline)) {
#pragma unroll
        for (int e = EPS * k; e < EPS * k + EPS; e += 2)
            if (e < NPEND) finish2(e);
    };
    auto walk = [&](auto& poster) __attribute__((always_inline)) {
        const int mtn = mt + WAVES < mt_hi ? mt + WAVES : mt;       // (the last item's ring refills re-read its own first steps: harmless)
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = a.bias[mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
        f32x16 hi[NT], lo[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) hi[j][r] = lo[j][r] = 0.f;
        if (!(CNX_ABL & 2)) cnx_mma<NT, NC, KS>(hi, lo, ring, a.A6, a.MT, mt, 0, mtn, 0, Ys, lane, poster);
        const float cw = a.wsc[mt] * 1.f, cl = cw * kLoInv;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) pend[j * 16 + r] = comb(hi[j][r], lo[j][r], cw, cl) + bv[r];
        pmt = mt;
    };
    [[maybe_unused]] int stamp = 7;
    {
        CnxNoPost none;
        walk(none);
    }
    CNX_STAMP(stamp);
    while (mt + WAVES < mt_hi) {
        mt += WAVES;
        walk(post);
        ++stamp;
        CNX_STAMP(stamp);
    }
#pragma unroll
    for (int e = 0; e < NPEND; e += 2) finish2(e);
    CNX_STAMP(12);
}

template <int C, int KPASS>
struct Cnx2 {
    static constexpr int K = 2 * C, NT = 2, NC = 64, KSL = K / 16 / KPASS;       // K16 steps per pass
    static constexpr int WAVES = KPASS == 2 ? C / 32 : 8, NTHR = WAVES * 64;      // two passes: the accumulators live across them, one m-tile per wave
    static constexpr int YS_U4 = KSL * 4 * NC;
    static constexpr int GPP = K / 8 / KPASS;                                     // 8-channel groups per pass
    static constexpr int ITEMS = GPP * NC, XPER = (ITEMS + NTHR - 1) / NTHR;
    static constexpr int LDS_BYTES = YS_U4 * 16 + (K + 32) * 4;
    static_assert(ITEMS % NTHR == 0, "staging items divide");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

template <int C, int KPASS>
__global__ __launch_bounds__((Cnx2<C, KPASS>::NTHR)) void cnx2_kernel(CnxArgs a) {
    using CF = Cnx2<C, KPASS>;
    constexpr int K = CF::K, NT = CF::NT, NC = CF::NC, KSL = CF::KSL, WAVES = CF::WAVES, XPER = CF::XPER;
    extern __shared__ __attribute__((aligned(16))) uint4 smem_c2[];
    uint4* Ys = smem_c2;
    float* Fl = reinterpret_cast<float*>(smem_c2 + CF::YS_U4);      // GRN factors of this utterance [2C], then 32 floats of exchange
    float* red = Fl + K;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    int Tb = a.T;
    long xoff = (long)b * C * a.T, hoff = (long)b * K * a.T;
    if (a.rg.tb) {
        Tb = a.rg.tb[b];
        xoff = hoff = a.rg.pre[b];
    }
    const int ntu = (Tb + NC - 1) / NC;
    if ((int)blockIdx.x >= ntu) return;
    const int TW = (Tb + ntu - 1) / ntu;
    const int t0 = blockIdx.x * TW;
    const int tw = Tb - t0 < TW ? Tb - t0 : TW;
    const int rs = a.rs;

    // ---- GRN factors: grn_finalize_kernel's arithmetic and order (256 threads) --------------------------------------------
    // (their inputs are requested first and the first operand half right behind them, so that it flies under the reductions:
    // one in-order counter - waiting for a load waits for every older one)
    const float* g = a.gp + (long)b * a.gp_tiles * K;
    const int gtiles = a.gp_sum ? 1 : ntu;                          // (cnx1's tiles are these: 64 columns, or one tile of <= 32)
    constexpr int GPT = K / 256;
    float gv[GPT], gm[GPT];
    if (tid < 256) {
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            float ss = g[tid + 256 * i];
            for (int t = 1; t < gtiles; ++t) ss += g[(long)t * K + tid + 256 * i];
            gv[i] = ss;
            gm[i] = a.grn_g[tid + 256 * i];
        }
    }
    // ---- operand staging: an item = 8 channels of one column (gemm_s2's SCALED staging arithmetic) ---------------------------
    const float* hb = a.h + hoff;
    const int scol = lane;                                          // NC = 64: a wave = one 8-channel group x 64 columns
    const unsigned so = 4u * (unsigned)(t0 + (scol < tw ? scol : tw - 1));
    float xr[XPER][8];
    Bfp sx{1.f, 1.f};
    auto fetch = [&](int pass) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < XPER; ++i) {
            const int gg = pass * CF::GPP + wave + i * WAVES;
#pragma unroll
            for (int q = 0; q < 8; ++q) xr[i][q] = (CNX_ABL & 32) ? (float)(gg + q) : ldg_so(hb + (long)(8 * gg + q) * rs, so);
        }
    };
    auto deposit = [&](int pass) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < XPER; ++i) {
            const int gl = wave + i * WAVES, gg = pass * CF::GPP + gl;
            const f32x4s_t k0 = *reinterpret_cast<const f32x4s_t*>(Fl + 8 * gg), k1 = *reinterpret_cast<const f32x4s_t*>(Fl + 8 * gg + 4);
            xr[i][0] *= k0[0]; xr[i][1] *= k0[1]; xr[i][2] *= k0[2]; xr[i][3] *= k0[3];
            xr[i][4] *= k1[0]; xr[i][5] *= k1[1]; xr[i][6] *= k1[2]; xr[i][7] *= k1[3];
#pragma unroll
            for (int q = 0; q < 8; ++q) xr[i][q] *= sx.s;
            uint4 p1, p2;
            split8(xr[i], p1, p2);
            Ys[(((gl >> 1) * 2 + 0) * 2 + (gl & 1)) * NC + scol] = p1;
            Ys[(((gl >> 1) * 2 + 1) * 2 + (gl & 1)) * NC + scol] = p2;
        }
    };
    fetch(0);
    if (tid < 256) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < GPT; ++i) gv[i] = sqrtf(gv[i]);
#pragma unroll
        for (int i = 0; i < GPT; ++i) s += gv[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) red[wave] = s;
    }
    slab_barrier();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)K;
    const float den = mean + 1e-6f;
    if (tid < 256) {
        float mx = 0.f;
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const float f = fmaf(gm[i], gv[i] / den, 1.f);
            Fl[tid + 256 * i] = f;
            mx = fmaxf(mx, gv[i] * fabsf(f));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if (lane == 0) red[4 + wave] = mx;
    }
    slab_barrier();
    sx = bfp_from_amax(fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])));      // |h * factor| <= max_c gx[c] |f[c]|
    sx.s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sx.s)));       // (uniform: two scalar registers instead of two vector ones held across both K passes)
    sx.inv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sx.inv)));
    deposit(0);
    if (KPASS == 2) fetch(1);                                       // lands under the first pass's MFMAs
    slab_barrier();

    const int mt_lo = blockIdx.z * a.mt_per_wg;
    const int mt_hi = mt_lo + a.mt_per_wg < a.MT ? mt_lo + a.mt_per_wg : a.MT;
    int mt = mt_lo + wave;
    const bool has = mt < mt_hi;                                    // waves without an m-tile still stage (KPASS == 2) and meet the others at the |max| exchange
    if (!has) mt = mt_lo;
    u32x4 ring[CNX_PD][2];
#pragma unroll
    for (int u = 0; u < CNX_PD; ++u) {
        const uint4* ab = a.A6 + ((long)u * a.MT + mt) * kPU4;
        ring[u][0] = ldg_so4(ab, 16u * (unsigned)lane);
        ring[u][1] = ldg_so4(ab, 16u * (unsigned)(64 + lane));
    }
    float* xb = a.x + xoff;
    float mx_out = 0.f;
    bool work = has;
    while (KPASS == 2 || work) {
        const int mtn = (KPASS == 1 && mt + WAVES < mt_hi) ? mt + WAVES : mt;
        f32x16 hi[NT], lo[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) hi[j][r] = lo[j][r] = 0.f;
        if (KPASS == 1) {
            if (!(CNX_ABL & 2)) cnx_mma<NT, NC, KSL>(hi, lo, ring, a.A6, a.MT, mt, 0, mtn, 0, Ys, lane);
        } else {
            if (has && !(CNX_ABL & 2)) cnx_mma<NT, NC, KSL>(hi, lo, ring, a.A6, a.MT, mt, 0, mt, KSL, Ys, lane);
            slab_barrier();                                         // every wave is done with the first K half
            deposit(1);
            slab_barrier();
            if (!has) break;
            if (!(CNX_ABL & 2)) cnx_mma<NT, NC, KSL>(hi, lo, ring, a.A6, a.MT, mt, KSL, mt, KSL, Ys, lane);
        }
        // epilogue: EpiBias<ACT_NONE, true> (bias' = c3.bias + c3.weight . grn.beta), residual = the layer's input, in place
        const float cw = a.wsc[mt] * sx.inv, cl = cw * kLoInv;
        int el = tid;
        asm volatile("" : "+v"(el));      // (the epilogue's lane-derived indices are recomputed here instead of living - spilled, at K = 768 - across the MFMA walks)
        const int l31 = el & 31, lh = (el >> 5) & 1;
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = a.bias[mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = j * 32 + l31;
            const unsigned oo = 4u * (unsigned)(4 * lh * rs + t0 + (n < tw ? n : tw - 1));     // (columns past the end: copies of the last one, stored to its address)
            float res[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) res[r] = ldg_so(xb + (long)(mt * 32 + (r & 3) + 8 * (r >> 2)) * rs, oo);
            float mx = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float o = comb(hi[j][r], lo[j][r], cw, cl) + bv[r];
                o += res[r];
                res[r] = o;
                mx = fmaxf(mx, fabsf(o));
            }
            // (in place: only the column's own lane stores - a clamped lane of the second n-tile would re-read a column the first n-tile has already updated)
            if (n < tw && !((CNX_ABL & 8) && res[0] != 1.2345f)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) stg_so(xb + (long)(mt * 32 + (r & 3) + 8 * (r >> 2)) * rs, oo, res[r]);
                mx_out = fmaxf(mx_out, mx);
            }
        }
        if (KPASS == 2) break;
        work = mt + WAVES < mt_hi;
        mt += WAVES;
    }
    if (a.amax_y) amax_flush_wg(a.amax_y + b, mx_out, red + 8);      // the next contraction's |max| slot: one atomic per workgroup
}

}  // namespace tvc
