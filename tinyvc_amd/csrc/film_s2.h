// Upsample's second conv of a half with its FiLM and residual (decoder.py:176-188) for the 96 / 192 / 384-channel levels:
//     out = (W (.) lrelu(h) + b) * (W_sc cond + b_sc) + (W_sh cond + b_sh) + residual
// on conv_s2.h's flat pipeline (two LDS staging buffers, one barrier per step, no data-dependent branch in the loop body).
//
// A step is one 16-channel slab of BOTH inputs: the conv's three taps over the staged halo tile of h and the to_scale / to_shift
// columns of the same 16 channels of cond (the block has as many cond channels as conv channels), 30 MFMAs per wave between two
// barriers.  conv3s.h ran the FiLM products as separate phases behind the conv's; on this pipeline a phase of 1x1 steps costs as much
// per step as a conv step (measured: the step time is the latency of the global loads one step ahead, not the MFMA count), so the FiLM
// products ride inside the conv's steps instead: a tile has C / 16 steps, not 3 C / 16.
//
// Three results per element (conv, scale, shift) leave no registers for accumulator PAIRS (conv3s.h: 2 x 3 x 32 at three waves per
// SIMD), so this kernel adds the three part products of the fp16 split into ONE fp32 accumulator per result.  The low parts are then
// plain fp16 residuals (x - fp16(x), not scaled by 2^11), whose absolute resolution is fp16's 2^-24 (subnormals are kept by the MFMA);
// to make that negligible every operand is normalised first: weights per 32-row m-tile to |max| in [2^13, 2^14) at pack time (api.hip
// film_u), activations per utterance to |max| in [2^14, 2^15) by the power of two taken from the tensor's |max| slot - always, not only
// outside fp16's range.  A residual is then resolved to 2^-38 of its tensor's largest value; products stay below 2^29 and sums over
// K = 3 * 384 below 2^40.  Error against fp64 on N(0,1) operands, K = 768: 3.1e-7 rel rms at every input scale from 1e-7 to 1e5 (accumulator
// pairs: 1.9e-7, fp32 MFMA: 4.9e-7, bf16 x 3: 4.2e-7; tools/micro/f16split.hip, profiles/r03_f16split.txt).  The slots this kernel reads are exact maxima written by the producing kernels' epilogues.
#pragma once
#include "conv3s.h"
#include "conv_s2.h"

namespace tvc {


struct FilmS2Args {
    const uint4* img;      // FilmU image [96-row block][slab][30 pieces][lane][8 fp16]
    const float* tab;      // [6][C]: conv bias, conv row scale, b_sc, b_sh, to_scale row scale, to_shift row scale
    const float* x;        // [B][C][len]   (h: the half's first conv)
    const float* cond;     // [B][C][len]
    int C, len, dil;
    float* y;              // [B][C][len]
    const float* res;      // residual [B][C][len], or (res_lin > 0) the low-rate [B][C][res_lin] tensor it is interpolated from
    int res_lin;
    float res_scale;
    int tiles_per_utt, ntiles, mblocks;
    const float* amax_x;   // per-utterance |max| slots: h and cond (read), out (written, nullable)
    const float* amax_c;
    float* amax_y;
    // XPRE: x is conv_s2's pre-split output (two fp16 planes [B][part][C / 8][len][8 fp16] of lrelu(h) * 2^k, k from the bound
    // pre_w * |max of the producer's input| + pre_b): staged by copy, amax_x = that input's slot
    const uint4* xpre;
    float pre_w, pre_b;
    RagDev rag;            // RAG kernels (ragged.h): len / res_lin = row strides of the batch-wide tensors, tiles / extents from the table
};

// Workgroup tile 96 x 256: 8 waves side by side, each all 96 rows (WM = 3 m-tiles) of 32 columns - 144 accumulator registers of the 256 a
// wave has at two waves per SIMD.  The 12-wave layout of conv_s2.h (3 x 4 waves of 32 x 64: MW = 3, WN = 2) needs 96 accumulators + 28
// staging registers + fragments > the 168 registers of three waves per SIMD: it compiles (the code below is written for both) but spills
// its staging registers inside the step loop.
// NWV_ = 7: a 224-column tile for levels whose length fills 256-column tiles badly (400 samples = the 384-channel level of a 4 s utterance: 2 x 224 = 448
// columns instead of 512, -12 % of the launch); the columns are independent, so the tile width changes no value.
template <int NWV_>
struct FS2T {
    static constexpr int MTB = 3, MW = 1, WM = MTB / MW, NWV = NWV_, WN = 1, NW = MW * NWV, NTHR = NW * 64, BN = NWV * WN * 32, MAXD = 27, XROW = BN + 2 * MAXD;
    static constexpr int A_CONV = 3 * MTB * 2, A_PIECES = A_CONV + 2 * MTB * 2, A_PER = (A_PIECES + NW - 1) / NW;
    static constexpr int XS = (2 * XROW + NTHR - 1) / NTHR;       // conv staging items per thread
    static constexpr int A_U4 = A_PIECES * 64, X_U4 = 2 * 2 * XROW, C_U4 = 2 * 2 * BN, BUF_U4 = A_U4 + X_U4 + C_U4;
    static constexpr int lds_bytes = 2 * BUF_U4 * 16 + 6 * 384 * 4 + 64;
    static_assert(2 * BN <= NTHR, "one cond item per thread");
};
using FS2 = FS2T<8>;

// fp16(v) and the fp16 of what it left behind
__device__ __forceinline__ void split8u(const float (&v)[8], uint4& p1, uint4& p2) {
    unsigned o1[4], o2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split2<false>(v[2 * j], v[2 * j + 1], o1[j], o2[j]);
    p1 = make_uint4(o1[0], o1[1], o1[2], o1[3]);
    p2 = make_uint4(o2[0], o2[1], o2[2], o2[3]);
}

template <bool XPRE, bool RAG = false, int NWV_ = 8>
__global__ __launch_bounds__(FS2T<NWV_>::NTHR) __attribute__((amdgpu_waves_per_eu(2))) void film_s2_kernel(FilmS2Args a) {
    using TL = FS2T<NWV_>;
    constexpr int MTB = TL::MTB, WM = TL::WM, NWV = TL::NWV, WN = TL::WN, NW = TL::NW, BN = TL::BN, XROW = TL::XROW, A_PER = TL::A_PER, XS = TL::XS;
    extern __shared__ __attribute__((aligned(16))) uint4 smem_fs[];
    float* Tb = reinterpret_cast<float*>(smem_fs + 2 * TL::BUF_U4);      // six rows of 384
    float* red = Tb + 6 * 384;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / NWV, wn = wave - wm * NWV;
    const int C = a.C, nslab = C / 16, rs = a.len, dil = a.dil;      // rs = row stride of x / cond / y (= every utterance's length unless RAG)
    const int rf = a.res_lin > 0 ? rs / a.res_lin : 1;               // RAG: an utterance's low-rate length = its length / rf
    const int xw = BN + 2 * dil, nitems = 2 * xw;

    for (int i = tid; i < 6 * C; i += TL::NTHR) {
        const int r = i / C;
        Tb[r * 384 + (i - r * C)] = a.tab[i];
    }

    int tfirst, tlast;
    tile_range(a.ntiles, tfirst, tlast);
    if (tfirst >= tlast) return;
    // (b is also the hint of the ragged table walk; len / off = the utterance's length and first column)
    auto coords = [&](int v, int& mb, int& b, int& t0, int& len, int& off) __attribute__((always_inline)) {
        const int nt = v / a.mblocks;
        mb = v - nt * a.mblocks;
        const RagTile rt = rag_tile<RAG>(a.rag, nt, a.tiles_per_utt, rs, b);
        b = rt.b;
        t0 = rt.tin * BN;
        len = rt.len;
        off = rt.off;
    };

    // this thread's two staging items: (8-channel half, column) of the conv's halo tile and of the cond tile; threads beyond the
    // items repeat an earlier one (same value to the same LDS row: no branch in the loop body)
    u32x4 ar[A_PER];
    float xr[XPRE ? 1 : XS][8], cr[8];
    uint4 xq[XPRE ? XS : 1][2];          // XPRE: the two parts of an item, as stored
    // (an item's (half, column) pair is needed once per tile - tile_offsets recomputes it from a laundered thread index -; only its LDS row lives across the steps)
    auto item_of = [&](int t_, int k, int& g, int& c) __attribute__((always_inline)) {
        int item = t_ + k * TL::NTHR;
        item = item < nitems ? item : item - nitems;
        g = item / xw;
        c = item - g * xw;
    };
    int xdst[XS];
#pragma unroll
    for (int k = 0; k < XS; ++k) {
        int g, c;
        item_of(tid, k, g, c);
        xdst[k] = g * XROW + c;
    }
    const int cg = tid >= BN ? 1 : 0, cc = tid - cg * BN;      // (NTHR = 2 BN)
    const int cdst = cg * BN + cc;
    unsigned xo[XS], co = 0;
    float xs = 1.f, cs_ = 1.f;        // of the load cursor's tile
    float rxs = 1.f, rcs = 1.f;       // of the slab in flight
    int lv = tfirst, ls = 0, lmb, lb = 0, lt0, llen, loff;
    coords(lv, lmb, lb, lt0, llen, loff);
    auto tile_offsets = [&]() __attribute__((always_inline)) {
        const int ub = RAG ? loff : 0;             // RAG: the utterance's first column rides in the lane offsets
        int tl = tid;
        asm volatile("" : "+v"(tl));
#pragma unroll
        for (int k = 0; k < XS; ++k) {
            int g, c;
            item_of(tl, k, g, c);
            int p = lt0 - dil + c;
            p = p < 0 ? 0 : (p > llen - 1 ? llen - 1 : p);
            xo[k] = (unsigned)((XPRE ? 1 : 8) * g * rs + ub + p);      // XPRE: row g of the slab's two plane rows (16-byte elements); else channel 8 g of its sixteen
        }
        int pc = lt0 + cc;
        pc = pc > llen - 1 ? llen - 1 : pc;
        co = (unsigned)(8 * cg * rs + ub + pc);
        xs = XPRE ? 1.f : norm_from_amax(sload_f32(a.amax_x + lb)).s;
        cs_ = norm_from_amax(sload_f32(a.amax_c + lb)).s;
    };
    tile_offsets();
    auto issue_load = [&]() __attribute__((always_inline)) {
        const uint4* abase = a.img + ((long)lmb * nslab + ls) * (TL::A_PIECES * 64);
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int q = wave + i * NW;
            q = q < TL::A_PIECES ? q : TL::A_PIECES - 1;
            ar[i] = ldg_so4(abase, 16u * (unsigned)(q * 64 + lane));
        }
        const long ubase = RAG ? 0L : (long)lb * C;
        const float* xc = a.x + (ubase + (long)ls * 16) * rs;
        const float* cc_ = a.cond + (ubase + (long)ls * 16) * rs;
        if (XPRE) {      // item (8-channel half g, column p): one 16-byte load per part; xo = g rs + p -> plane row 2 ls + g, column p
            const uint4* xp = a.xpre + ((RAG ? 0L : (long)lb * 2 * (C >> 3)) + 2 * ls) * rs;
#pragma unroll
            for (int k = 0; k < XS; ++k) {
                const unsigned e = xo[k];
                xq[k][0] = __builtin_bit_cast(uint4, ldg_so4(xp, 16u * e));
                xq[k][1] = __builtin_bit_cast(uint4, ldg_so4(xp + (long)(C >> 3) * rs, 16u * e));
            }
        } else {
#pragma unroll
            for (int k = 0; k < XS; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) xr[k][j] = ldg_so(xc + (long)j * rs, 4u * xo[k]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) cr[j] = ldg_so(cc_ + (long)j * rs, 4u * co);
        rxs = xs;
        rcs = cs_;
    };
    auto advance_load = [&]() __attribute__((always_inline)) {
        if (++ls == nslab) {
            if (lv + 1 < tlast) {
                ++lv;
                ls = 0;
                coords(lv, lmb, lb, lt0, llen, loff);
                tile_offsets();
            } else {
                ls = nslab - 1;       // past the last step the cursor parks on it
            }
        }
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {
        uint4* Ab = smem_fs + buf * TL::BUF_U4;
        uint4* Xb = Ab + TL::A_U4;
        uint4* Cb = Xb + TL::X_U4;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int q = wave + i * NW;
            q = q < TL::A_PIECES ? q : TL::A_PIECES - 1;
            *reinterpret_cast<u32x4*>(Ab + q * 64 + lane) = ar[i];
        }
        float v[8];
        uint4 p1, p2;
#pragma unroll
        for (int k = 0; k < XS; ++k) {
            if (XPRE) {
                p1 = xq[k][0];
                p2 = xq[k][1];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaxf(xr[k][j], 0.1f * xr[k][j]) * rxs;      // leaky_relu(h, 0.1), normalised (power of two: exact)
                split8u(v, p1, p2);
            }
            Xb[xdst[k]] = p1;
            Xb[2 * XROW + xdst[k]] = p2;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = cr[j] * rcs;
        split8u(v, p1, p2);
        Cb[cdst] = p1;
        Cb[2 * BN + cdst] = p2;
    };

    f32x16 hi[WM][WN], sc[WM][WN], sh[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) hi[i][j][r] = sc[i][j][r] = sh[i][j][r] = 0.f;
    // acc += w2 x1 + w1 x2 + w1 x1
    auto tap_mul = [&](int buf, int tap) __attribute__((always_inline)) {
        const uint4* Ab = smem_fs + buf * TL::BUF_U4 + ((tap * MTB + wm * WM) * 2) * 64 + lane;
        const uint4* Xb = smem_fs + buf * TL::BUF_U4 + TL::A_U4 + lh * XROW + wn * WN * 32 + l31 + tap * dil;
        f16x8 bf[WN][2];
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p) bf[j][p] = __builtin_bit_cast(f16x8, Xb[p * 2 * XROW + j * 32]);
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            f16x8 af[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) af[p] = __builtin_bit_cast(f16x8, Ab[(i * 2 + p) * 64]);
#pragma unroll
            for (int j = 0; j < WN; ++j) hi[i][j] = TVC_MFMA16(af[1], bf[j][0], hi[i][j]);
#pragma unroll
            for (int j = 0; j < WN; ++j) hi[i][j] = TVC_MFMA16(af[0], bf[j][1], hi[i][j]);
#pragma unroll
            for (int j = 0; j < WN; ++j) hi[i][j] = TVC_MFMA16(af[0], bf[j][0], hi[i][j]);
        }
    };
    auto film_mul = [&](int buf) __attribute__((always_inline)) {
        const uint4* Ab = smem_fs + buf * TL::BUF_U4 + (TL::A_CONV + wm * WM * 2) * 64 + lane;
        const uint4* Cb = smem_fs + buf * TL::BUF_U4 + TL::A_U4 + TL::X_U4 + lh * BN + wn * WN * 32 + l31;
        f16x8 bf[WN][2];
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p) bf[j][p] = __builtin_bit_cast(f16x8, Cb[p * 2 * BN + j * 32]);
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            f16x8 af[2], ag[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) af[p] = __builtin_bit_cast(f16x8, Ab[(i * 2 + p) * 64]);
#pragma unroll
            for (int p = 0; p < 2; ++p) ag[p] = __builtin_bit_cast(f16x8, Ab[((MTB + i) * 2 + p) * 64]);
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                sc[i][j] = TVC_MFMA16(af[1], bf[j][0], sc[i][j]);
                sh[i][j] = TVC_MFMA16(ag[1], bf[j][0], sh[i][j]);
            }
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                sc[i][j] = TVC_MFMA16(af[0], bf[j][1], sc[i][j]);
                sh[i][j] = TVC_MFMA16(ag[0], bf[j][1], sh[i][j]);
            }
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                sc[i][j] = TVC_MFMA16(af[0], bf[j][0], sc[i][j]);
                sh[i][j] = TVC_MFMA16(ag[0], bf[j][0], sh[i][j]);
            }
        }
    };

    // consumer cursor
    int cv = tfirst, cs = 0, cmb, cb = 0, ct0, len, coff;
    coords(cv, cmb, cb, ct0, len, coff);
    float mx_run = 0.f;
    int flush_b = -1;

    issue_load();                 // step 0
    advance_load();
    lstore(0);
    issue_load();                 // step 1
    advance_load();
    slab_barrier();
    int buf = 0;
    while (true) {
        tap_mul(buf, 0);
        lstore(buf ^ 1);          // step u + 1 -> the other buffer
        issue_load();             // step u + 2
        tap_mul(buf, 1);
        tap_mul(buf, 2);
        film_mul(buf);
        advance_load();
        if (++cs == nslab) {
            cs = 0;
            // ---- tile end: the elements are finished and stored straight from the accumulators ---------------------------------
            // (the lane index is laundered through an empty asm so that the index math below is not hoisted out of the persistent loop
            // and kept in registers across the MFMA steps: conv3s.h tile_store)
            int el = lane;
            asm volatile("" : "+v"(el));
            const int e31 = el & 31, eh = el >> 5;
            const float ix = norm_from_amax(XPRE ? presplit_bound(a.pre_w, a.pre_b, a.amax_x, cb) : sload_f32(a.amax_x + cb)).inv, ic_ = norm_from_amax(sload_f32(a.amax_c + cb)).inv;
#pragma unroll
            for (int i = 0; i < WM; ++i) {
            const int row0 = (cmb * MTB + wm * WM + i) * 32;
            const float kc = Tb[384 + row0] * ix, ks = Tb[4 * 384 + row0] * ic_, kh = Tb[5 * 384 + row0] * ic_;      // one m-tile: one scale each
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int rr = row0 + 8 * g + 4 * eh;
                const float4 b0 = *reinterpret_cast<const float4*>(Tb + rr);
                const float4 b1 = *reinterpret_cast<const float4*>(Tb + 2 * 384 + rr);
                const float4 b2 = *reinterpret_cast<const float4*>(Tb + 3 * 384 + rr);
                const float bm[4] = {b0.x, b0.y, b0.z, b0.w}, bs[4] = {b1.x, b1.y, b1.z, b1.w}, bh[4] = {b2.x, b2.y, b2.z, b2.w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < WN; ++j) {
                        const int r = 4 * g + q;
                        const float h = fmaf(hi[i][j][r], kc, bm[q]), s = fmaf(sc[i][j][r], ks, bs[q]), t = fmaf(sh[i][j][r], kh, bh[q]);
                        hi[i][j][r] = __fadd_rn(__fmul_rn(h, s), t);         // h * scale + shift, rounded like the reference's two tensor ops
                        sc[i][j][r] = 0.f;
                        sh[i][j][r] = 0.f;
                    }
            }
            float* yb = RAG ? a.y + (long)row0 * rs + coff : a.y + ((long)cb * C + row0) * rs;                // uniform; a lane adds its 32-bit offset
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int t = ct0 + (wn * WN + j) * 32 + e31;
                const bool live = t < len;
                const int tc = live ? t : len - 1;
                const unsigned off = 4u * (unsigned)(4 * eh * rs + tc);
                if (a.res_lin > 0) {       // the residual is F.interpolate of the low-rate tensor, evaluated here
                    const Lerp lc = lerp_coord(tc, a.res_scale, RAG ? len / rf : a.res_lin);
                    const float* rb = RAG ? a.res + (long)row0 * a.res_lin + coff / rf : a.res + ((long)cb * C + row0) * a.res_lin;
                    const unsigned o0 = 4u * (unsigned)(4 * eh * a.res_lin + lc.i0), o1 = 4u * (unsigned)(4 * eh * a.res_lin + lc.i1);
#pragma unroll
                    for (int h8 = 0; h8 < 2; ++h8) {
                        float x0[8], x1[8];
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            const float* rr = rb + (long)(16 * h8 + (r & 3) + 8 * (r >> 2)) * a.res_lin;
                            x0[r] = ldg_so(rr, o0);
                            x1[r] = ldg_so(rr, o1);
                        }
#pragma unroll
                        for (int r = 0; r < 8; ++r) hi[i][j][8 * h8 + r] = __fadd_rn(hi[i][j][8 * h8 + r], lerp_eval(lc, x0[r], x1[r]));
                    }
                } else {
                    const float* rb = RAG ? a.res + (long)row0 * rs + coff : a.res + ((long)cb * C + row0) * rs;
                    float rv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[r] = ldg_so(rb + (long)((r & 3) + 8 * (r >> 2)) * rs, off);
#pragma unroll
                    for (int r = 0; r < 16; ++r) hi[i][j][r] = __fadd_rn(hi[i][j][r], rv[r]);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = hi[i][j][r];
                    if (live) {
                        stg_so(yb + (long)((r & 3) + 8 * (r >> 2)) * rs, off, e);
                        mx_run = fmaxf(mx_run, fabsf(e));
                    }
                    hi[i][j][r] = 0.f;
                }
            }
            }
            const int done_b = cb;
            ++cv;
            const bool last = cv >= tlast;
            if (!last) coords(cv, cmb, cb, ct0, len, coff);
            if (a.amax_y && (last || cb != done_b)) {          // the workgroup leaves utterance done_b: the waves' maxima meet in LDS,
                const float m = wave_max(mx_run);              // one thread publishes them behind the next barrier
                if (lane == 0) red[wave] = m;
                mx_run = 0.f;
                flush_b = done_b;
            }
            if (last) break;
        }
        slab_barrier();           // buffer buf ^ 1 is complete, and nobody reads buffer buf any more
        if (flush_b >= 0) {
            if (tid == 0) {
                float m = 0.f;
                for (int w = 0; w < NW; ++w) m = fmaxf(m, red[w]);
                if (m > 0.f) atomicMax(reinterpret_cast<unsigned*>(a.amax_y + flush_b), __builtin_bit_cast(unsigned, m));
            }
            flush_b = -1;
        }
        buf ^= 1;
    }
    if (flush_b >= 0) {
        slab_barrier();
        if (tid == 0) {
            float m = 0.f;
            for (int w = 0; w < NW; ++w) m = fmaxf(m, red[w]);
            if (m > 0.f) atomicMax(reinterpret_cast<unsigned*>(a.amax_y + flush_b), __builtin_bit_cast(unsigned, m));
        }
    }
}

// true = launched (or failed: *rc); false = outside this kernel's preconditions (use conv3s_launch)
// pre = true: h is conv_s2's pre-split output (conv_s2.h PRE) normalised by the bound pre_w * |max| + pre_b, bfp.x = the slot that bound is taken from
inline bool film_s2_try(int* rc, tvc_ctx* ctx, hipStream_t s, const FilmU& fu, const float* h, const float* cond, int B, int C, int len, int dil, float* out,
                        const float* res, int res_lin, float res_scale, const BfpSlots& bfp, bool pre = false, float pre_w = 0.f, float pre_b = 0.f) {
    if (!fu.img || fu.C != C || C % 96 != 0 || C > 384 || dil < 1 || dil > FS2::MAXD || !bfp.x || !bfp.c || !res) return false;
    if ((long)C * len * 4 >= (1L << 32) || (res_lin > 0 && (long)C * res_lin * 4 >= (1L << 32))) return false;
    static bool ready_dev[64] = {};
    static int ncu_dev[64] = {};
    bool& ready = ready_dev[ctx->device & 63];
    int& ncu = ncu_dev[ctx->device & 63];
    if (!ready) {
        hipDeviceProp_t prop;
        hipError_t e = hipGetDeviceProperties(&prop, ctx->device);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)film_s2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, FS2::lds_bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)film_s2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, FS2::lds_bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)film_s2_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FS2::lds_bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)film_s2_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FS2::lds_bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)film_s2_kernel<false, false, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, FS2T<7>::lds_bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)film_s2_kernel<true, false, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, FS2T<7>::lds_bytes);
        if (e != hipSuccess) {
            *rc = fail(ctx, TVC_ERR_HIP, "film_s2 setup: %s", hipGetErrorString(e));
            return true;
        }
        ncu = prop.multiProcessorCount;
        ready = true;
    }
    FilmS2Args a{};
    a.img = reinterpret_cast<const uint4*>(fu.img);
    a.tab = fu.tab;
    a.x = h;
    a.cond = cond;
    a.C = C;
    a.len = len;
    a.dil = dil;
    a.y = out;
    a.res = res;
    a.res_lin = res_lin;
    a.res_scale = res_scale;
    a.mblocks = C / 96;
    a.tiles_per_utt = (len + FS2::BN - 1) / FS2::BN;
    a.ntiles = a.tiles_per_utt * B * a.mblocks;
    a.amax_x = bfp.x;
    a.xpre = reinterpret_cast<const uint4*>(h);
    a.pre_w = pre_w;
    a.pre_b = pre_b;
    a.amax_c = bfp.c;
    a.amax_y = bfp.y;
    if (ctx->rag) {
        // ragged batch (ragged.h): the driver passed B = 1 and len = the batch's columns at this rate (= the row stride)
        if (B != 1 || len % ctx->rag->Ttot != 0) { *rc = fail(ctx, TVC_ERR_STATE, "film_s2: a ragged batch runs as one long utterance"); return true; }
        int ncol = 0;
        *rc = rag_view(ctx, s, len / ctx->rag->Ttot, FS2::BN, &a.rag, &ncol);
        if (*rc) return true;
        a.ntiles = ncol * a.mblocks;
    }
    const int grid = a.ntiles < ncu ? a.ntiles : ncu;
    if (ctx->rag) {
        if (pre) hipLaunchKernelGGL((film_s2_kernel<true, true>), dim3(grid), dim3(FS2::NTHR), FS2::lds_bytes, s, a);
        else hipLaunchKernelGGL((film_s2_kernel<false, true>), dim3(grid), dim3(FS2::NTHR), FS2::lds_bytes, s, a);
    } else if ((len + 223) / 224 * 224 < (len + 255) / 256 * 256) {
        // the narrower tile computes fewer padded columns at this length
        a.tiles_per_utt = (len + 223) / 224;
        a.ntiles = a.tiles_per_utt * B * a.mblocks;
        const int g7 = a.ntiles < ncu ? a.ntiles : ncu;
        if (pre) hipLaunchKernelGGL((film_s2_kernel<true, false, 7>), dim3(g7), dim3(FS2T<7>::NTHR), FS2T<7>::lds_bytes, s, a);
        else hipLaunchKernelGGL((film_s2_kernel<false, false, 7>), dim3(g7), dim3(FS2T<7>::NTHR), FS2T<7>::lds_bytes, s, a);
    } else if (pre) hipLaunchKernelGGL(film_s2_kernel<true>, dim3(grid), dim3(FS2::NTHR), FS2::lds_bytes, s, a);
    else hipLaunchKernelGGL(film_s2_kernel<false>, dim3(grid), dim3(FS2::NTHR), FS2::lds_bytes, s, a);
    *rc = launch_check(ctx, "film_s2");
    return true;
}

}  // namespace tvc
