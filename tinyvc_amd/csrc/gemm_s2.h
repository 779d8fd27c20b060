// Plain GEMM on the split-precision path, pipelined:  out(m, n) = sum_k W[m][k] x[b][k][t],  n = b * T + t.
//
// Same arithmetic, weight image and epilogue functors as the TAPS = 1 launches of conv3s.h (two fp16 parts per fp32
// operand, three part-products per product into an accumulator pair), different schedule.  The cycle stamps
// of conv3s's slab loop (tools/micro/slab_trace.py) showed its two halves - stage a slab (wait for the loads, split,
// ds_write, request the next slab: ~1850 cycles) and multiply it (~1840 cycles, matrix-pipe bound) - strictly
// alternating between two barriers, so the matrix pipe idles half of the time.  Here
//   * the LDS staging area is double-buffered and a slab costs ONE barrier: while slab u multiplies out of buffer
//     u & 1, the same waves split slab u + 1 into the other buffer and request slab u + 2 - the staging
//     instructions sit between the two K16 steps' MFMAs of every wave instead of in a phase of their own.  The slab body
//     has no branch (the load cursor parks on the last slab of the walk and is advanced behind the MFMAs), so it is one
//     basic block and the compiler threads the ds_writes, splits and loads between the first step's MFMAs;
//   * the (tile, slab) pairs a persistent workgroup walks form ONE flat pipeline: the next tile's first slabs are
//     staged under the last slabs of this one, and a wave's epilogue (bias / GELU / residual, stores straight from the
//     accumulators) overlaps the other waves' next slab;
//   * one workgroup per CU (72-123 KB of LDS), 4-12 waves: no spills at any tile;
//   * the column-tile width is chosen per launch (64 ... 192 columns) so that the tile count fills whole rounds of
//     the 256 CUs (conv3s's 600 tiles on 512 slots were 2 rounds at 59 % fill) and, when a launch cannot fill the chip
//     (a streaming block is 896 columns), so that it spreads over the most CUs;
//   * GRN's per-(utterance, channel) factors travel with the slab (two 16-byte loads per staged item), so a flat
//     column tile may straddle any number of utterances.
#pragma once
#include "conv3s.h"
#include "gemm_epi.h"

namespace tvc {

constexpr int G2_NWV_MIN = 2;     // narrowest workgroup tile: 32 * G2_NWV_MIN columns

struct Gemm2Args {
    const uint4* A6;       // split weight image [K16 step][m-tile][part][lane][8 fp16]
    const float* wsc;      // its per-m-tile power-of-two scales
    const float* amax_x;   // per-utterance |max| of x (block-floating-point guard of the fp16 split), or nullptr: no scaling
    int MT;                // m-tiles (32 rows) in the image
    const float* x;        // [B][Cin][T], utterance b at x + b * xstride
    unsigned xstride;
    int Cin, T, ncols;     // ncols = B * T
    const float* kscale;   // SCALED: [B][Cin] factor applied to the input while it is staged
    int ncoltiles, mblocks, vtiles;
    const int* col2b;      // ragged batch (ragged.h; B = 1, T = all frames): frame -> utterance, for the per-utterance scalars (amax_x, kscale); else nullptr
};

template <int NWV_, int WN_>
struct G2Tile {
    static constexpr int NWV = NWV_, WN = WN_, MTB = 4, WM = 2;
    static constexpr int NW = 2 * NWV, NTHR = NW * 64, BM = 128, BN = NWV * WN * 32;
    static constexpr int A_U4 = 2 * MTB * kPU4;            // one buffer's weight pieces: [K16 step][m-tile][part][lane]
    static constexpr int X_U4 = 2 * kParts * 2 * BN;       // one buffer's input tile: [K16 step][part][8-channel half][column]
    static constexpr int BUF_U4 = A_U4 + X_U4;
    static constexpr int X_PER = WN;                       // staged items (8 channels x 1 column) per thread: 4 * BN / NTHR
    static constexpr int A_PIECES = 2 * MTB * kParts, A_PER = (A_PIECES + NW - 1) / NW;
    static constexpr int lds_bytes = 2 * BUF_U4 * 16;
};

template <int NWV, int WN, class Epi, bool SCALED>
__global__ __launch_bounds__(2 * NWV * 64) void gemm_s2_kernel(Gemm2Args a, Epi ep) {
    using TL = G2Tile<NWV, WN>;
    constexpr int BN = TL::BN, NW = TL::NW, A_PER = TL::A_PER, X_PER = TL::X_PER;
    extern __shared__ __attribute__((aligned(16))) uint4 smem_g2[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / NWV, wn = wave - wm * NWV;
    const int nslab = a.Cin / 32;
    const int stride = gridDim.x;

    // tile walk: virtual tile v runs on XCD v % 8 (workgroups are dealt round-robin, the grid is a multiple of 8); the
    // row blocks of one column tile share v % 8, i.e. one L2 serves their re-reads of the same input columns
    auto coords = [&](int v, int& mt0, int& n0) __attribute__((always_inline)) -> bool {
        const int r = v & 7, u = v >> 3;
        const int mb = u % a.mblocks;
        const int nt = (u / a.mblocks) * 8 + r;
        mt0 = mb * TL::MTB;
        n0 = nt * BN;
        return nt < a.ncoltiles;
    };
    auto next_valid = [&](int v) __attribute__((always_inline)) -> int {
        int m_, n_;
        while (v < a.vtiles && !coords(v, m_, n_)) v += stride;
        return v;
    };

    // staging registers of the one slab in flight, the load cursor and its per-tile lane offsets
    u32x4 ar[A_PER];
    float xr[X_PER][8];
    u32x4 kr[X_PER][2];
    unsigned xo[X_PER], ko[X_PER];
    float xsc[X_PER];           // block-floating-point scale of the item's utterance
    int xdst[X_PER];
#pragma unroll
    for (int i = 0; i < X_PER; ++i) {
        const int idx = tid + i * TL::NTHR;
        const int gk = idx / BN, c = idx - gk * BN;          // gk = 2 * (K16 step) + (8-channel half)
        xdst[i] = (gk >> 1) * (2 * kParts * BN) + (gk & 1) * BN + c;
    }
    auto tile_offsets = [&](int n0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < X_PER; ++i) {
            const int idx = tid + i * TL::NTHR;
            const int gk = idx / BN, c = idx - gk * BN;
            int p = n0 + c;
            p = p < a.ncols ? p : a.ncols - 1;               // columns past the end are computed, never stored
            const int b = p / a.T, t = p - b * a.T;
            const int bu = a.col2b ? a.col2b[p] : b;         // whose |max| slot and GRN factors
            xo[i] = (unsigned)b * a.xstride + (unsigned)(gk * 8 * a.T + t);
            ko[i] = (unsigned)(bu * a.Cin + gk * 8);
            xsc[i] = bfp_load(a.amax_x, bu).s;
        }
    };
    int lv = next_valid(blockIdx.x), ls = 0, lmt0 = 0, ln0 = 0;
    if (lv >= a.vtiles) return;
    coords(lv, lmt0, ln0);
    int cv = lv, cs = 0, cmt0 = lmt0, cn0 = ln0;
    tile_offsets(ln0);

    auto issue_load = [&]() __attribute__((always_inline)) {     // global -> registers only; the values are not touched here
        const uint4* abase = a.A6 + ((long)ls * 2 * a.MT + lmt0) * kPU4;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int q = wave + i * NW;
            q = q < TL::A_PIECES ? q : TL::A_PIECES - 1;
            const int kg = q / (TL::MTB * kParts), rem = q - kg * (TL::MTB * kParts);
            ar[i] = ldg_so4(abase, 16u * (unsigned)(kg * a.MT * kPU4 + rem * 64 + lane));
        }
        const float* xc = a.x + (long)ls * 32 * a.T;
#pragma unroll
        for (int i = 0; i < X_PER; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) xr[i][j] = ldg_so(xc + (long)j * a.T, 4u * xo[i]);
        if constexpr (SCALED) {
            const uint4* kc = reinterpret_cast<const uint4*>(a.kscale + ls * 32);
#pragma unroll
            for (int i = 0; i < X_PER; ++i) {
                kr[i][0] = ldg_so4(kc, 4u * ko[i]);
                kr[i][1] = ldg_so4(kc, 4u * ko[i] + 16u);
            }
        }
    };
    auto advance_load = [&]() __attribute__((always_inline)) {
        if (++ls == nslab) {      // next tile of this workgroup's walk; past the last one the cursor stays on the last slab
            const int v2 = next_valid(lv + stride);   // (the loads it repeats are harmless and keep the loop body free of branches)
            if (v2 < a.vtiles) {
                lv = v2;
                ls = 0;
                coords(lv, lmt0, ln0);
                tile_offsets(ln0);
            } else {
                ls = nslab - 1;
            }
        }
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {  // registers -> split -> LDS
        uint4* Ab = smem_g2 + buf * TL::BUF_U4;
        uint4* Xb = Ab + TL::A_U4;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int q = wave + i * NW;
            q = q < TL::A_PIECES ? q : TL::A_PIECES - 1;     // surplus slots of the last round rewrite the last piece with itself: no branch
            *reinterpret_cast<u32x4*>(Ab + q * 64 + lane) = ar[i];
        }
#pragma unroll
        for (int i = 0; i < X_PER; ++i) {
            if constexpr (SCALED) {
                // (element-wise on the vectors: indexing kr[i][0][j] in an unrolled loop was folded to element 0 by this compiler)
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                const f32x4 k0 = __builtin_bit_cast(f32x4, kr[i][0]), k1 = __builtin_bit_cast(f32x4, kr[i][1]);
                xr[i][0] *= k0.x; xr[i][1] *= k0.y; xr[i][2] *= k0.z; xr[i][3] *= k0.w;
                xr[i][4] *= k1.x; xr[i][5] *= k1.y; xr[i][6] *= k1.z; xr[i][7] *= k1.w;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) xr[i][j] *= xsc[i];                  // power of two: exact
            uint4 p1, p2;
            split8(xr[i], p1, p2);
            Xb[xdst[i]] = p1;
            Xb[2 * BN + xdst[i]] = p2;
        }
    };

    f32x16 hi[2][WN], lo[2][WN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) hi[i][j][r] = lo[i][j][r] = 0.f;
    auto multiply = [&](int buf, int kg) __attribute__((always_inline)) {
        const uint4* Ab = smem_g2 + buf * TL::BUF_U4 + (kg * TL::MTB * kParts + wm * 2 * kParts) * 64 + lane;
        const uint4* Xb = smem_g2 + buf * TL::BUF_U4 + TL::A_U4 + kg * (2 * kParts * BN) + lh * BN + wn * WN * 32 + l31;
        f16x8 af[2][kParts], bf[WN][kParts];
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int p = 0; p < kParts; ++p) bf[j][p] = __builtin_bit_cast(f16x8, Xb[p * 2 * BN + j * 32]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < kParts; ++p) af[i][p] = __builtin_bit_cast(f16x8, Ab[(i * kParts + p) * 64]);
        // three part-products per tile; the accumulators alternate (same order as conv3s.h)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) lo[i][j] = TVC_MFMA16(af[i][1], bf[j][0], lo[i][j]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) hi[i][j] = TVC_MFMA16(af[i][0], bf[j][0], hi[i][j]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) lo[i][j] = TVC_MFMA16(af[i][0], bf[j][1], lo[i][j]);
    };

#ifdef S_TRACE
    struct { unsigned long long* tr = nullptr; int trn = 0; } trs;
    __shared__ unsigned long long tr_lds[256];
    if (blockIdx.x == S_TRACE_WG && threadIdx.x == S_TRACE_TID) trs.tr = tr_lds;
#endif

    issue_load();                 // unit 0
    advance_load();
    lstore(0);
    issue_load();                 // unit 1
    advance_load();
    slab_barrier();
    int buf = 0;
    while (true) {
        TR_STAMP(trs, 0);
        multiply(buf, 0);
        TR_STAMP(trs, 1);
        lstore(buf ^ 1);          // unit u + 1 -> the other buffer
        TR_STAMP(trs, 2);
        issue_load();             // unit u + 2
        TR_STAMP(trs, 3);
        multiply(buf, 1);
        TR_STAMP(trs, 4);
        advance_load();           // (behind the MFMAs: the slab body above is one basic block)
        if (++cs == nslab) {
            // epilogue straight from the accumulators (gemm_epi.h functors finish the element)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float cw = a.wsc[cmt0 + wm * 2 + i];
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const int n = cn0 + (wn * WN + j) * 32 + l31;
                    const int nc = n < a.ncols ? n : a.ncols - 1;
                    const float c = cw * bfp_load(a.amax_x, a.col2b ? a.col2b[nc] : nc / a.T).inv, cl = c * kLoInv;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float v[4] = {comb(hi[i][j][4 * q], lo[i][j][4 * q], c, cl), comb(hi[i][j][4 * q + 1], lo[i][j][4 * q + 1], c, cl),
                                            comb(hi[i][j][4 * q + 2], lo[i][j][4 * q + 2], c, cl), comb(hi[i][j][4 * q + 3], lo[i][j][4 * q + 3], c, cl)};
                        ep.store(n, (cmt0 + wm * 2 + i) * 32 + 8 * q + 4 * lh, v);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) hi[i][j][r] = lo[i][j][r] = 0.f;
                }
            }
            cs = 0;
            cv = next_valid(cv + stride);
            if (cv >= a.vtiles) break;
            coords(cv, cmt0, cn0);
        }
        slab_barrier();           // buffer buf ^ 1 is complete, and nobody reads buffer buf any more
        TR_STAMP(trs, 5);
        buf ^= 1;
    }
#ifdef S_TRACE
    if (trs.tr) {
        const unsigned slot = atomicAdd(&g_trace_slot, 1u) & 63u;
        unsigned long long* g = g_trace + slot * 256;
        g[0] = 0x5452414345000000ull | (2ull << 20) | ((unsigned long long)NWV << 16) | ((unsigned long long)WN << 12) | ((unsigned long long)SCALED << 8);
        g[1] = ((unsigned long long)a.Cin << 32) | (unsigned)trs.trn;
        g[2] = ((unsigned long long)gridDim.x << 32) | (unsigned)(a.ncoltiles * a.mblocks);
        g[3] = 0;
        for (int i = 0; i < trs.trn; ++i) g[4 + i] = trs.tr[i];
    }
#endif
}

template <int NWV, int WN, class Epi, bool SCALED>
inline int gemm_s2_launch_t(tvc_ctx* ctx, hipStream_t s, const PackedW& w, const float* x, int B, int Cin, int T, long xstride, const Epi& ep,
                            const float* kscale, int ncu, const float* amax_x) {
    using TL = G2Tile<NWV, WN>;
    static bool ready_dev[64] = {};
    bool& ready = ready_dev[ctx->device & 63];
    if (!ready) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_s2_kernel<NWV, WN, Epi, SCALED>, hipFuncAttributeMaxDynamicSharedMemorySize, TL::lds_bytes);
        if (e != hipSuccess) return fail(ctx, TVC_ERR_HIP, "gemm_s2 setup: %s", hipGetErrorString(e));
        ready = true;
    }
    Gemm2Args a;
    a.A6 = reinterpret_cast<const uint4*>(w.A6);
    a.wsc = w.wscale;
    a.amax_x = amax_x;
    a.MT = w.MT6;
    a.x = x;
    a.xstride = (unsigned)(xstride ? xstride : (long)Cin * T);
    a.Cin = Cin;
    a.T = T;
    a.ncols = B * T;
    a.kscale = kscale;
    a.mblocks = w.MT6 / TL::MTB;
    a.ncoltiles = (a.ncols + TL::BN - 1) / TL::BN;
    a.vtiles = (a.ncoltiles + 7) / 8 * 8 * a.mblocks;
    a.col2b = nullptr;
    if (ctx->rag) {
        if (B != 1 || T != ctx->rag->Ttot) return fail(ctx, TVC_ERR_STATE, "gemm_s2: a ragged batch runs as one long utterance at the frame rate");
        a.col2b = ctx->rag->d_col2b;
    }
    const int slots = ncu / 8 * 8;       // one persistent workgroup per CU, a multiple of 8 (XCD walk)
    dim3 g((unsigned)(a.vtiles < slots ? a.vtiles : slots));
    hipLaunchKernelGGL((gemm_s2_kernel<NWV, WN, Epi, SCALED>), g, dim3(TL::NTHR), TL::lds_bytes, s, a, ep);
    return 0;
}

// true = launched (or failed: *rc); false = the shape is outside this kernel's preconditions, use gemm_s_launch
template <class Epi, bool SCALED = false>
inline bool gemm_s2_try(int* rc, tvc_ctx* ctx, hipStream_t s, const PackedW& w, const float* x, int B, int Cin, int T, long xstride, const Epi& ep,
                        const float* amax_x, const float* kscale = nullptr) {
    const long xs = xstride ? xstride : (long)Cin * T;
    if (Cin % 32 != 0 || Cin / 16 > w.S6 || w.MT6 % 4 != 0 || xs * B >= (1L << 29) || (long)B * T >= (1L << 29) || (SCALED && !kscale)) return false;
    static int ncu_dev[64] = {};
    int& ncu = ncu_dev[ctx->device & 63];
    if (!ncu) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, ctx->device) != hipSuccess) { *rc = fail(ctx, TVC_ERR_HIP, "gemm_s2: device properties"); return true; }
        ncu = prop.multiProcessorCount;
    }
    // column-tile width: whole rounds of the CUs, then the fewest staging round trips (wider tiles reuse a weight slab more)
    const long ncols = (long)B * T;
    const int mblocks = w.MT6 / 4, slots = ncu / 8 * 8;
    int best = 4;
    long best_cost = -1;
    for (int nwv = G2_NWV_MIN; nwv <= 6; ++nwv) {     // 64-column tiles only pay when the launch cannot fill the chip (a streaming block: 896 columns)
        const long tiles = (ncols + nwv * 32 - 1) / (nwv * 32) * mblocks;
        const long rounds = (tiles + slots - 1) / slots;
        const long cost = rounds * (nwv * 32 + 64);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = nwv; }
    }
    if (best == 2) *rc = gemm_s2_launch_t<2, 1, Epi, SCALED>(ctx, s, w, x, B, Cin, T, xstride, ep, kscale, ncu, amax_x);
    else if (best == 3) *rc = gemm_s2_launch_t<3, 1, Epi, SCALED>(ctx, s, w, x, B, Cin, T, xstride, ep, kscale, ncu, amax_x);
    else if (best == 4) *rc = gemm_s2_launch_t<4, 1, Epi, SCALED>(ctx, s, w, x, B, Cin, T, xstride, ep, kscale, ncu, amax_x);
    else if (best == 5) *rc = gemm_s2_launch_t<5, 1, Epi, SCALED>(ctx, s, w, x, B, Cin, T, xstride, ep, kscale, ncu, amax_x);
    else *rc = gemm_s2_launch_t<6, 1, Epi, SCALED>(ctx, s, w, x, B, Cin, T, xstride, ep, kscale, ncu, amax_x);
    return true;
}

}  // namespace tvc
