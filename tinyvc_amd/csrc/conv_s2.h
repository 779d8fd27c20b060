// k3 dilated Conv1d (replicate padding) on the split-precision path, pipelined:  y = W (.) lrelu(x) + b  for the plain convs of
// FilterNet's 96/192/384-channel levels (Upsample c1 - with F.interpolate evaluated while staging - and c3, Downsample c1 / c2;
// decoder.py:143-146,166-171).
//
// Same arithmetic, weight image, block-floating-point guard and output values as conv3s.h's plain launches, different schedule -
// the one gemm_s2.h proved on the 1x1 GEMMs (there: -25 % under the fp16 split): the (tile, 16-channel slab) pairs a persistent
// workgroup walks are ONE flat pipeline with two LDS staging buffers and one barrier per slab.  While slab u multiplies out of
// buffer u & 1 (three K16 steps = the three taps of one staged halo tile), the same waves split slab u + 1 into the other buffer
// between the first and the second tap's MFMAs and request slab u + 2; the next tile's first slabs are staged under the last
// slabs of this one.  The loop body has no data-dependent branch (idle staging threads repeat another thread's item, surplus
// weight-piece slots rewrite the last piece, the load cursor parks on the last slab), so the compiler threads the staging
// instructions between the queued MFMAs.  The output tile leaves straight from the accumulators (no LDS park: the two staging
// buffers take its place), 128 contiguous bytes per row and store instruction.
#pragma once
#include "conv3s.h"

namespace tvc {

struct ConvS2Args {
    const uint4* A6;       // split weight image [slab * 3 + tap][m-tile][part][lane][8 fp16]
    const float* wsc;      // its per-m-tile power-of-two scales
    int MT;                // m-tiles in the image
    const float* x;        // [B][Cin][len], or (LERP) the low-rate tensor [B][Cin][lin]
    int Cin, len, dil, lin;
    float lscale;          // LERP: ATen's source-coordinate scale float(1 / scale_factor)
    float* y;              // [B][M][len]
    const float* bias;
    int M, tiles_per_utt, ntiles, mblocks;
    const float* amax_x;   // per-utterance |max| slots (conv3s.h): input (read, nullable), output (written, nullable)
    float* amax_y;
    // PRE: the output is written as the NEXT conv's ready operand (film_s2.h: split(lrelu(y) * 2^k) as two fp16 planes
    // [B][part][M / 8][len][8 fp16], k from the analytic bound pre_w * |x|max + pre_b >= |y|, which both kernels evaluate alike)
    uint4* ypre;
    float pre_w, pre_b;
    RagDev rag;            // RAG kernels (ragged.h): len / lin = row strides of the batch-wide tensors, tiles / extents from the table
};
// |conv + bias| <= (max_m sum_k |w|) |x|max + max |b|: the bound the producer normalises its pre-split output by and the consumer undoes
// (amax = the per-utterance |max| slot of the producer's INPUT; non-null for these launches)
__device__ __forceinline__ float presplit_bound(float pre_w, float pre_b, const float* amax, int b) { return fmaf(pre_w, sload_f32(amax + b), pre_b); }      // (b wave-uniform: scalar cache, conv3s.h)
// power of two that brings |max| into [2^14, 2^15) (1 for zero, Inf, NaN: they carry no information)
__device__ __forceinline__ Bfp norm_from_amax(float amax) {
    const unsigned u = __builtin_bit_cast(unsigned, amax);
    Bfp r{1.f, 1.f};
    if (u != 0u && u < 0x7f800000u) {
        int k = 14 - ((int)(u >> 23) - 127);
        k = k > 120 ? 120 : (k < -120 ? -120 : k);
        r.s = __builtin_bit_cast(float, (unsigned)(127 + k) << 23);
        r.inv = __builtin_bit_cast(float, (unsigned)(127 - k) << 23);
    }
    return r;
}

struct CS2 {
    static constexpr int MTB = 3, NWV = 4, WN = 2, NW = MTB * NWV, NTHR = NW * 64, BN = NWV * WN * 32, MAXD = 27, XROW = BN + 2 * MAXD;
    static constexpr int A_PIECES = 3 * MTB * kParts, A_PER = (A_PIECES + NW - 1) / NW;
    static constexpr int A_U4 = A_PIECES * 64, X_U4 = kParts * 2 * XROW, BUF_U4 = A_U4 + X_U4;
    static constexpr int TAB = 2 * 128 * 3;                  // bias and weight-scale rows of up to 384 output channels
    static constexpr int lds_bytes = 2 * BUF_U4 * 16 + TAB * 4 + 64;
};

template <bool LERP, bool PRE = false, bool RAG = false>
__global__ __launch_bounds__(CS2::NTHR) __attribute__((amdgpu_waves_per_eu(3))) void conv_s2_kernel(ConvS2Args a) {
    using TL = CS2;
    constexpr int MTB = TL::MTB, NWV = TL::NWV, WN = TL::WN, NW = TL::NW, BN = TL::BN, XROW = TL::XROW, A_PER = TL::A_PER;
    extern __shared__ __attribute__((aligned(16))) uint4 smem_c2[];
    float* Tb = reinterpret_cast<float*>(smem_c2 + 2 * TL::BUF_U4);      // [bias (M)][row scale (M)] then the |max| exchange
    float* red = Tb + TL::TAB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / NWV, wn = wave - wm * NWV;
    const int nslab = a.Cin / 16, rs = a.len, dil = a.dil;   // rs = row stride of y (= every utterance's length unless RAG)
    const int cs_ = LERP ? a.lin : rs;                       // channel stride of x
    const int xf = LERP ? rs / a.lin : 1;                    // RAG: an utterance's low-rate length = its length / xf
    const int xw = BN + 2 * dil, nitems = 2 * xw;

    for (int i = tid; i < a.M; i += TL::NTHR) {
        Tb[i] = a.bias[i];
        Tb[384 + i] = a.wsc[i >> 5];
    }

    int tfirst, tlast;
    tile_range(a.ntiles, tfirst, tlast);
    if (tfirst >= tlast) return;
    // (b is also the hint of the ragged table walk; len / off = the utterance's length and first column at the output rate)
    auto coords = [&](int v, int& mt0, int& b, int& t0, int& len, int& off) __attribute__((always_inline)) {
        const int nt = v / a.mblocks, mb = v - nt * a.mblocks;
        mt0 = mb * MTB;
        const RagTile rt = rag_tile<RAG>(a.rag, nt, a.tiles_per_utt, rs, b);
        b = rt.b;
        t0 = rt.tin * BN;
        len = rt.len;
        off = rt.off;
    };

    // staging registers of the one slab in flight and this thread's item: (8-channel half g, column c) of the halo tile;
    // threads beyond the tile's items repeat an earlier item (same value to the same LDS row: no branch in the loop body)
    u32x4 ar[A_PER];
    float xr[8], xr2[LERP ? 8 : 1];
    const int item = tid < nitems ? tid : tid - nitems;
    const int ig = item / xw, ic = item - ig * xw;
    const int xdst = ig * XROW + ic;
    unsigned xo = 0, xo1 = 0;
    float lw0 = 0.f, lw1 = 0.f, xsc = 1.f;       // of the load cursor's tile
    float rw0 = 0.f, rw1 = 0.f, rxs = 1.f;       // of the slab in flight in the registers (the cursor may already stand on the next tile when it is split)
    int lv = tfirst, ls = 0, lmt0, lb = 0, lt0, llen, loff;
    coords(lv, lmt0, lb, lt0, llen, loff);
    auto tile_offsets = [&]() __attribute__((always_inline)) {
        int p = lt0 - dil + ic;
        p = p < 0 ? 0 : (p > llen - 1 ? llen - 1 : p);
        const int cbase = 8 * ig * cs_ + (RAG ? (LERP ? loff / xf : loff) : 0);     // RAG: the utterance's first column rides in the lane offset
        if (LERP) {
            const Lerp lc = lerp_coord(p, a.lscale, RAG ? llen / xf : a.lin);
            xo = (unsigned)(cbase + lc.i0);
            xo1 = (unsigned)(cbase + lc.i1);
            lw0 = lc.w0;
            lw1 = lc.w1;
        } else {
            xo = (unsigned)(cbase + p);
        }
        xsc = bfp_load_u(a.amax_x, lb).s;
    };
    tile_offsets();
    auto issue_load = [&]() __attribute__((always_inline)) {     // global -> registers only; the values are not touched here
        const uint4* abase = a.A6 + ((long)ls * 3 * a.MT + lmt0) * kPU4;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int q = wave + i * NW;
            q = q < TL::A_PIECES ? q : TL::A_PIECES - 1;
            const int tap = q / (MTB * kParts), rem = q - tap * (MTB * kParts);
            ar[i] = ldg_so4(abase, 16u * (unsigned)(tap * a.MT * kPU4 + rem * 64 + lane));
        }
        const float* xc = a.x + ((RAG ? 0L : (long)lb * a.Cin) + (long)ls * 16) * cs_;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            xr[j] = ldg_so(xc + (long)j * cs_, 4u * xo);
            if (LERP) xr2[j] = ldg_so(xc + (long)j * cs_, 4u * xo1);
        }
        rw0 = lw0;
        rw1 = lw1;
        rxs = xsc;
    };
    auto advance_load = [&]() __attribute__((always_inline)) {
        if (++ls == nslab) {      // next tile of this workgroup's range; past the last one the cursor stays on the last slab
            if (lv + 1 < tlast) {
                ++lv;
                ls = 0;
                coords(lv, lmt0, lb, lt0, llen, loff);
                tile_offsets();
            } else {
                ls = nslab - 1;
            }
        }
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {  // registers -> lrelu, scale, split -> LDS
        uint4* Ab = smem_c2 + buf * TL::BUF_U4;
        uint4* Xb = Ab + TL::A_U4;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int q = wave + i * NW;
            q = q < TL::A_PIECES ? q : TL::A_PIECES - 1;
            *reinterpret_cast<u32x4*>(Ab + q * 64 + lane) = ar[i];
        }
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t = LERP ? fmaf(rw0, xr[j], __fmul_rn(rw1, xr2[j])) : xr[j];      // = lerp_eval
            v[j] = fmaxf(t, 0.1f * t) * rxs;                                               // = leaky_relu(x, 0.1), scaled (power of two)
        }
        uint4 p1, p2;
        split8(v, p1, p2);
        Xb[xdst] = p1;
        Xb[2 * XROW + xdst] = p2;
    };

    f32x16 hi[WN], lo[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) hi[j][r] = lo[j][r] = 0.f;
    auto multiply = [&](int buf, int tap) __attribute__((always_inline)) {
        const uint4* Ab = smem_c2 + buf * TL::BUF_U4 + ((tap * MTB + wm) * kParts) * 64 + lane;
        const uint4* Xb = smem_c2 + buf * TL::BUF_U4 + TL::A_U4 + lh * XROW + wn * WN * 32 + l31 + tap * dil;
        f16x8 af[kParts], bf[WN][kParts];
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int p = 0; p < kParts; ++p) bf[j][p] = __builtin_bit_cast(f16x8, Xb[p * 2 * XROW + j * 32]);
#pragma unroll
        for (int p = 0; p < kParts; ++p) af[p] = __builtin_bit_cast(f16x8, Ab[p * 64]);
#pragma unroll
        for (int j = 0; j < WN; ++j) lo[j] = TVC_MFMA16(af[1], bf[j][0], lo[j]);
#pragma unroll
        for (int j = 0; j < WN; ++j) hi[j] = TVC_MFMA16(af[0], bf[j][0], hi[j]);
#pragma unroll
        for (int j = 0; j < WN; ++j) lo[j] = TVC_MFMA16(af[0], bf[j][1], lo[j]);
    };

    // consumer cursor
    int cv = tfirst, cs = 0, cmt0, cb = 0, ct0, len, coff;
    coords(cv, cmt0, cb, ct0, len, coff);
    float mx_run = 0.f;
    int flush_b = -1;

    issue_load();                 // unit 0
    advance_load();
    lstore(0);
    issue_load();                 // unit 1
    advance_load();
    slab_barrier();
    int buf = 0;
    while (true) {
        multiply(buf, 0);
        lstore(buf ^ 1);          // unit u + 1 -> the other buffer
        issue_load();             // unit u + 2
        multiply(buf, 1);
        multiply(buf, 2);
        advance_load();           // (behind the MFMAs: the slab body above is one basic block)
        if (++cs == nslab) {
            // epilogue straight from the accumulators: bias, store, running |max|
            const Bfp sx = bfp_load_u(a.amax_x, cb);
            const int row0 = (cmt0 + wm) * 32;
            float* yb = RAG ? a.y + (long)row0 * rs + coff : a.y + ((long)cb * a.M + row0) * rs;
            const float ps = PRE ? norm_from_amax(presplit_bound(a.pre_w, a.pre_b, a.amax_x, cb)).s : 1.f;
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int t = ct0 + (wn * WN + j) * 32 + l31;
                const bool live = t < len;
                const unsigned off = 4u * (unsigned)(4 * lh * rs + (live ? t : len - 1));
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (row0 + 8 * g < a.M) {                      // uniform
                        const float4 bv = *reinterpret_cast<const float4*>(Tb + row0 + 8 * g + 4 * lh);
                        const float c = Tb[384 + row0] * sx.inv, cl = c * kLoInv;
                        const float e[4] = {comb(hi[j][4 * g], lo[j][4 * g], c, cl) + bv.x, comb(hi[j][4 * g + 1], lo[j][4 * g + 1], c, cl) + bv.y,
                                            comb(hi[j][4 * g + 2], lo[j][4 * g + 2], c, cl) + bv.z, comb(hi[j][4 * g + 3], lo[j][4 * g + 3], c, cl) + bv.w};
                        if (PRE) {
                            // lrelu, normalise, split; a position's 16-byte operand row = [lanes 0-31's four channels | lanes 32-63's four]:
                            // v_permlane32_swap hands the lower half of the wave both halves of the part-1 row, the upper half those of part 2
                            float v[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[q] = fmaxf(e[q], 0.1f * e[q]) * ps;
                            unsigned p1[2], p2[2];
                            split2<false>(v[0], v[1], p1[0], p2[0]);
                            split2<false>(v[2], v[3], p1[1], p2[1]);
                            const auto qx = __builtin_amdgcn_permlane32_swap(p1[0], p2[0], false, false);
                            const auto qy = __builtin_amdgcn_permlane32_swap(p1[1], p2[1], false, false);
                            if (live) a.ypre[(((RAG ? 0L : (long)cb * 2) + lh) * (a.M >> 3) + ((row0 >> 3) + g)) * rs + (RAG ? coff : 0) + t] = make_uint4(qx[0], qy[0], qx[1], qy[1]);
                        } else if (live) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                stg_so(yb + (long)(8 * g + q) * rs, off, e[q]);
                                mx_run = fmaxf(mx_run, fabsf(e[q]));
                            }
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) hi[j][r] = lo[j][r] = 0.f;
            }
            cs = 0;
            const int done_b = cb;
            ++cv;
            const bool last = cv >= tlast;
            if (!last) coords(cv, cmt0, cb, ct0, len, coff);
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
template <bool LERP, bool PRE = false>
inline bool conv_s2_try(int* rc, tvc_ctx* ctx, hipStream_t s, const PackedW& w, const float* x, int B, int Cin, int len, int dil, float* y, const BfpSlots& bfp,
                        int lin = 0, float lscale = 0.f, float pre_w = 0.f, float pre_b = 0.f) {
    if (PRE && (!bfp.x || w.M % 8 != 0)) return false;
    if (w.taps != 3 || w.MT6 % 3 != 0 || w.M > 384 || Cin % 16 != 0 || Cin / 16 > w.S6 || dil < 1 || dil > CS2::MAXD || bfp.c) return false;
    if ((long)B * w.M * len >= (1L << 31) / 4 * 4 && (long)w.M * len * 4 >= (1L << 32)) return false;
    if ((long)Cin * (LERP ? lin : len) * 4 >= (1L << 32) || (long)w.M * len * 4 >= (1L << 32)) return false;      // 32-bit byte offsets inside an utterance
    if (LERP != (lin > 0)) return false;
    static bool ready_dev[64] = {};
    static int ncu_dev[64] = {};
    bool& ready = ready_dev[ctx->device & 63];
    int& ncu = ncu_dev[ctx->device & 63];
    if (!ready) {
        hipDeviceProp_t prop;
        hipError_t e = hipGetDeviceProperties(&prop, ctx->device);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_s2_kernel<LERP, PRE>, hipFuncAttributeMaxDynamicSharedMemorySize, CS2::lds_bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_s2_kernel<LERP, PRE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, CS2::lds_bytes);
        if (e != hipSuccess) { *rc = fail(ctx, TVC_ERR_HIP, "conv_s2 setup: %s", hipGetErrorString(e)); return true; }
        ncu = prop.multiProcessorCount;
        ready = true;
    }
    ConvS2Args a;
    a.A6 = reinterpret_cast<const uint4*>(w.A6);
    a.wsc = w.wscale;
    a.MT = w.MT6;
    a.x = x;
    a.Cin = Cin;
    a.len = len;
    a.dil = dil;
    a.lin = lin;
    a.lscale = lscale;
    a.y = y;
    a.bias = w.bias;
    a.M = w.M;
    a.mblocks = w.MT6 / CS2::MTB;
    a.tiles_per_utt = (len + CS2::BN - 1) / CS2::BN;
    a.ntiles = a.tiles_per_utt * B * a.mblocks;
    a.amax_x = bfp.x;
    a.amax_y = PRE ? nullptr : bfp.y;
    a.ypre = reinterpret_cast<uint4*>(y);
    a.pre_w = pre_w;
    a.pre_b = pre_b;
    a.rag = RagDev{};
    if (ctx->rag) {
        // ragged batch (ragged.h): the driver passed B = 1 and len = the batch's columns at this rate (= the row stride)
        if (B != 1 || len % ctx->rag->Ttot != 0) { *rc = fail(ctx, TVC_ERR_STATE, "conv_s2: a ragged batch runs as one long utterance"); return true; }
        int ncol = 0;
        *rc = rag_view(ctx, s, len / ctx->rag->Ttot, CS2::BN, &a.rag, &ncol);
        if (*rc) return true;
        a.ntiles = ncol * a.mblocks;
    }
    const int grid = a.ntiles < ncu ? a.ntiles : ncu;
    if (ctx->rag) hipLaunchKernelGGL((conv_s2_kernel<LERP, PRE, true>), dim3(grid), dim3(CS2::NTHR), CS2::lds_bytes, s, a);
    else hipLaunchKernelGGL((conv_s2_kernel<LERP, PRE>), dim3(grid), dim3(CS2::NTHR), CS2::lds_bytes, s, a);
    *rc = launch_check(ctx, "conv_s2");
    return true;
}

}  // namespace tvc
