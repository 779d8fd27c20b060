// 48-channel k3 dilated convs of FilterNet (ups.3's c1..c4 with their FiLM, Downsample 2's c1 / c2; decoder.py:143-190) with
// the layer's weights RESIDENT in LDS.  At 48 channels a conv is only three 16-channel K slabs: the generic split kernel
// (conv3s.h) restages its weights and passes two barriers per slab for a handful of MFMAs per wave, and its matrix pipe sat idle
// two thirds of the time.  Here a persistent 8-wave workgroup loads the conv's 36 weight pieces (and FiLM's 24) once, stages
// the whole 48-channel halo tile of 128 samples in one go and then issues the tile's 27 (+18) MFMAs per wave back to back:
// one staging round trip and two barriers per tile instead of per slab.  Two-term fp16 split, accumulator pairs and the
// block-floating-point guard as in conv3s.h.
//   weights   the same pre-split images as conv3s (PackedW::A6: [K16 step = slab*3 + tap][m-tile][part][lane][8 fp16];
//             stacked FiLM image [slab][scale mt0, mt1, shift mt0, mt1][part]) - no new packing;
//   input     Xs[part][8-channel group (6)][position][8 fp16]: lrelu (and, for c1, F.interpolate) applied while depositing;
//   waves     wave w = m-tile (w >> 2) x 32-sample n-tile (w & 3); FiLM's cond fragments come from HBM straight into
//             B-fragment order and are split in registers; (conv, scale, shift) combine in registers;
//   epilogue  bias / FiLM / residual (direct, or F.interpolate of the low-rate tensor evaluated in place), 128-byte runs
//             per row and store instruction; residual and cond are requested before the MFMAs.
#include "conv3s.h"
#include "small_kernels.h"
#include "tvc_common.h"

namespace tvc {

namespace {

constexpr int kC48 = 48, kBN48 = 128, kXP48 = kBN48 + 2 * 27, kNT48 = 512;
typedef float f32x4s_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

// two fp16 parts of 4 fp32 values (8 bytes each)
__device__ __forceinline__ void split4_48(const float (&v)[4], u32x2_t& p1, u32x2_t& p2) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        unsigned q1, q2;
        split2(v[2 * j], v[2 * j + 1], q1, q2);
        p1[j] = q1;
        p2[j] = q2;
    }
}

// 8 consecutive channels of one column of a G8 tensor ([groups][columns][8 channels] fp32: 32 contiguous bytes) as two 16-byte loads.
// (element by element: an ext_vector indexed by the induction variable of an unrolled loop is folded wrongly by this compiler)
__device__ __forceinline__ void ld8_g8(float (&dst)[8], const float* base, unsigned byte_off) {
    const uint4* b4 = reinterpret_cast<const uint4*>(base);
    const f32x4s_t q0 = __builtin_bit_cast(f32x4s_t, ldg_so4(b4, byte_off)), q1 = __builtin_bit_cast(f32x4s_t, ldg_so4(b4, byte_off + 16u));
    dst[0] = q0.x; dst[1] = q0.y; dst[2] = q0.z; dst[3] = q0.w;
    dst[4] = q1.x; dst[5] = q1.y; dst[6] = q1.z; dst[7] = q1.w;
}
__device__ __forceinline__ void ld4_g8(float (&dst)[4], const float* base, unsigned byte_off) {
    const f32x4s_t q0 = __builtin_bit_cast(f32x4s_t, ldg_so4(reinterpret_cast<const uint4*>(base), byte_off));
    dst[0] = q0.x; dst[1] = q0.y; dst[2] = q0.z; dst[3] = q0.w;
}

struct Conv48Args {
    // The level's own tensors - x1, h, the skip tensor cond - travel in the G8 layout [B][6 groups][len][8 channels] (a staged item, a cond
    // fragment or a lane's four output channels are 32 / 32 / 16 contiguous bytes: 16-byte accesses instead of strided 4-byte ones)
    const float* x;        // G8 [B][6][len][8]; LERP: the low-rate tensor, planar [B][48][lin]
    const float* cond;     // FILM: G8
    const float* res;      // RES 1: G8; RES 2: low-rate planar [B][48][rlin], interpolated here
    float* out;            // G8
    const u32x4* A6;       // conv image, 36 pieces
    const u32x4* F6;       // stacked FiLM image, 24 pieces
    const u32x4* W5;       // C5: c5's image (1 m-tile, 3 K16 steps: 6 pieces) and bias [32]
    const float* b5;
    float* out5;           // C5: the 24-channel output in the G8 layout [B][3 groups][len][8 channels] (filter_up24s.hip's input)
    const float* bias;     // [>= 48]
    const float* bsc;      // FiLM to_scale / to_shift biases [48]
    const float* bsh;
    const float* wsc;      // per-m-tile power-of-two scales of the images: conv [2], FiLM [4] (scale mt0, mt1, shift mt0, mt1), c5 [1]
    const float* fsc;
    const float* w5sc;
    // block-floating-point guard (conv3s.h): per-utterance |max| slots of x / cond (read, nullable) and of the output (written, nullable)
    const float* amax_x;
    const float* amax_c;
    float* amax_y;
    int len, dil, lin, rlin, tiles_per_utt, ntiles;
    float lscale, rscale;
    RagDev rag;            // RAG kernels (ragged.h): len / lin / rlin = row strides of the batch-wide tensors, tiles / extents from the table
};

// RES: 0 none, 1 direct, 2 interpolated.  C5: Upsample.c5 (1x1, 48 -> 24, decoder.py:171,189) applied to the finished tile
// before it leaves the CU: the 48-channel block output is never written, only c5's 24 rows are.
template <bool FILM, bool LERP, int RES, bool C5 = false, bool RAG = false>
// (launches without FiLM - 73 KB of LDS, <= 128 registers - run TWO workgroups per CU: a plain 48-channel conv waits on its 0.47 GB of HBM traffic more
// than on its 27 MFMAs per tile, and the second workgroup's loads fly under the first one's arithmetic: Upsample 3 0.80 -> 0.765 ms same-box, round 4)
__global__ __launch_bounds__(kNT48) __attribute__((amdgpu_waves_per_eu(FILM ? 2 : 4))) void conv48s_kernel(Conv48Args a) {
    constexpr int C = kC48, BN = kBN48, XP = kXP48, NT = kNT48;
    extern __shared__ __attribute__((aligned(16))) uint4 smem_q[];
    u32x4* Xs = reinterpret_cast<u32x4*>(smem_q);              // [2 parts][6 groups][XP]
    u32x4* Wt = Xs + 12 * XP;                                  // 36 pieces
    u32x4* Ft = Wt + 36 * 64;                                  // 24 pieces (FILM)
    u32x4* W5t = Ft + (FILM ? 24 * 64 : 0);                   // 6 pieces (C5)
    float* Bi = reinterpret_cast<float*>(W5t + (C5 ? 6 * 64 : 0));    // bias, bsc, bsh [64 each], b5 [32], [224] = the C5 tile's |max| (LDS atomic), [228..235] = |max| exchange
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int mt = wave >> 2, nt = wave & 3;
    const int rs = a.len, dil = a.dil;                       // rs = row stride of cond / res / out (= every utterance's length unless RAG, ragged.h)
    const int XW = BN + 2 * dil;
    const int rsl = LERP ? a.lin : rs;                         // row stride of x
    const int xf = LERP ? rs / a.lin : 1, rf = RES == 2 ? rs / a.rlin : 1;     // RAG: an utterance's low-rate lengths = len / xf, len / rf
    int bh = 0;                                                // RAG: utterance hint of the table walk
    auto utt = [&](int tile) __attribute__((always_inline)) { return rag_tile<RAG>(a.rag, tile, a.tiles_per_utt, rs, bh).b; };

    for (int i = tid; i < 36 * 64; i += NT) Wt[i] = a.A6[i];
    if (FILM)
        for (int i = tid; i < 24 * 64; i += NT) Ft[i] = a.F6[i];
    if (C5) {
        for (int i = tid; i < 6 * 64; i += NT) W5t[i] = a.W5[i];
        if (tid < 32) Bi[192 + tid] = a.b5[tid];
        if (tid == 0) Bi[224] = 0.f;
    }
    if (tid < 64) {
        const int tc = tid < C ? tid : C - 1;                  // rows 48..63 do not exist
        Bi[tid] = a.bias[tc];
        if (FILM) {
            Bi[64 + tid] = a.bsc[tc];
            Bi[128 + tid] = a.bsh[tc];
        }
    }
    // this wave's m-tile: power-of-two scales of its weight rows
    const float cw = a.wsc[mt], cwsc = FILM ? a.fsc[mt] : 1.f, cwsh = FILM ? a.fsc[2 + mt] : 1.f, cw5 = C5 ? a.w5sc[0] : 1.f;

    // staging items (8-channel group, column): 6 * XW <= 1092 of them, three per thread
    constexpr int XPER = 3;
    float xr0[XPER][8], xr1[LERP ? XPER : 1][8], lam[LERP ? XPER : 1];
    int ig[XPER], ic[XPER];
#pragma unroll
    for (int i = 0; i < XPER; ++i) {
        const int idx = tid + i * NT;
        const int g = idx / XW;
        ic[i] = idx - g * XW;
        ig[i] = g;                                             // g >= 6: idle item (loads a valid address, never stores)
    }
    auto fetch = [&](int tile) __attribute__((always_inline)) {
        const RagTile rt = rag_tile<RAG>(a.rag, tile, a.tiles_per_utt, rs, bh);
        const int len = rt.len, lin = LERP ? len / xf : len;
        const int px0 = rt.tin * BN - dil;
        const float* xb = RAG ? a.x + (LERP ? rt.off / xf : 8L * rt.off) : a.x + (long)rt.b * C * rsl;
#pragma unroll
        for (int i = 0; i < XPER; ++i) {
            const int g = ig[i] > 5 ? 5 : ig[i];
            int p = px0 + ic[i];
            p = p < 0 ? 0 : (p > len - 1 ? len - 1 : p);
            if (LERP) {
                const Lerp lc = lerp_coord(p, a.lscale, lin);
                lam[i] = lc.w1;
                const unsigned o0 = 4u * (unsigned)(8 * g * rsl + lc.i0), o1 = 4u * (unsigned)(8 * g * rsl + lc.i1);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    xr0[i][j] = ldg_so(xb + (long)j * rsl, o0);
                    xr1[i][j] = ldg_so(xb + (long)j * rsl, o1);
                }
            } else {
                ld8_g8(xr0[i], xb, 32u * (unsigned)(g * rsl + p));
            }
        }
    };
    auto deposit = [&](float xs) __attribute__((always_inline)) {      // xs = the tile's block-floating-point input scale
#pragma unroll
        for (int i = 0; i < XPER; ++i) {
            if (ig[i] > 5) continue;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = LERP ? fmaf(1.f - lam[i], xr0[i][j], __fmul_rn(lam[i], xr1[i][j])) : xr0[i][j];   // = lerp_eval
                v[j] = fmaxf(t, 0.1f * t) * xs;                                                             // = leaky_relu(x, 0.1), scaled
            }
            uint4 p1, p2;
            split8(v, p1, p2);
            Xs[(0 + ig[i]) * XP + ic[i]] = __builtin_bit_cast(u32x4, p1);
            Xs[(6 + ig[i]) * XP + ic[i]] = __builtin_bit_cast(u32x4, p2);
        }
    };

    // persistent: a contiguous range of tiles per workgroup (one or two utterances: the output's |max| slot is published once
    // per utterance and workgroup, conv3s.h amax_flush_wg)
    int tile, tend;
    tile_range(a.ntiles, tile, tend);
    if (tile >= tend) return;
    bh = utt(tile);
    fetch(tile);
    deposit(bfp_load_u(a.amax_x, bh).s);
    slab_barrier();
    float mx_run = 0.f;
    int mx_b = bh;
    int sc_b = -1;      // the utterance whose scales sx / sc hold: computed when the walk enters it, not per tile
    Bfp sx{1.f, 1.f}, sc{1.f, 1.f};

    for (; tile < tend; ++tile) {
        const RagTile rt = rag_tile<RAG>(a.rag, tile, a.tiles_per_utt, rs, bh);
        const int b = rt.b, len = rt.len;
        bh = b;
        if (a.amax_y && b != mx_b) {
            amax_flush_wg(a.amax_y + mx_b, mx_run, Bi + 228);
            mx_run = 0.f;
            mx_b = b;
        }
        if (b != sc_b) {
            sc_b = b;
            sx = bfp_load_u(a.amax_x, b);
            sc = FILM ? bfp_load_u(a.amax_c, b) : Bfp{1.f, 1.f};
        }
        const int t0 = rt.tin * BN;
        const int next = tile + 1;
        const int n = nt * 32 + l31;
        const int t = t0 + n;
        const int tc = t < len ? t : len - 1;

        // cond fragments of this wave's columns: K16 step s -> channels 16 s + 8 lh + j
        float cr[FILM ? 3 : 1][8];
        if (FILM) {
            const float* cb = RAG ? a.cond + 8L * rt.off : a.cond + (long)b * C * rs;
            const unsigned oc = 32u * (unsigned)(lh * rs + tc);                     // group 2 s + lh, column tc
#pragma unroll
            for (int s = 0; s < 3; ++s) ld8_g8(cr[s], cb + (long)s * 16 * rs, oc);
        }
        // residual values of this lane's 16 rows
        float rv[RES ? 4 : 1][4];
        if (RES == 1) {
            const float* rb = RAG ? a.res + 8L * rt.off : a.res + (long)b * C * rs;
            const unsigned og = 32u * (unsigned)(4 * mt * rs + tc) + 16u * (unsigned)lh;      // group 4 mt + g, column tc, channels 4 lh ..
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (32 * mt + 8 * g < C) {
                    ld4_g8(rv[g], rb + (long)g * 8 * rs, og);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) rv[g][q] = 0.f;                      // rows past 48 do not exist
                }
            }
        } else if (RES == 2) {
            const float* rb = RAG ? a.res + rt.off / rf : a.res + (long)b * C * a.rlin;
            const Lerp lc = lerp_coord(tc, a.rscale, RAG ? len / rf : a.rlin);
            const unsigned o0 = 4u * (unsigned)((32 * mt + 4 * lh) * a.rlin + lc.i0), o1 = 4u * (unsigned)((32 * mt + 4 * lh) * a.rlin + lc.i1);
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float x0 = 0.f, x1 = 0.f;
                    if (32 * mt + 8 * g < C) {
                        x0 = ldg_so(rb + (long)(8 * g + q) * a.rlin, o0);
                        x1 = ldg_so(rb + (long)(8 * g + q) * a.rlin, o1);
                    }
                    rv[g][q] = fmaf(lc.w0, x0, __fmul_rn(lc.w1, x1));      // = lerp_eval
                }
        }

        // ---- conv: 9 K16 steps (slab, tap), this wave's m-tile x n-tile ------------------------------------
        f32x16 acc, alo;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = alo[r] = 0.f;
        {
            f16x8 af[2][2], bf[2][2];
            auto frags = [&](int s, int fb) __attribute__((always_inline)) {
                const int sl = s / 3, tap = s - sl * 3;
                const int row = (2 * sl + lh) * XP + n + tap * dil;
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    bf[fb][p] = __builtin_bit_cast(f16x8, Xs[p * 6 * XP + row]);
                    af[fb][p] = __builtin_bit_cast(f16x8, Wt[((s * 2 + mt) * 2 + p) * 64 + lane]);
                }
            };
            frags(0, 0);
#pragma unroll
            for (int s = 0; s < 9; ++s) {
                const int fb = s & 1;
                if (s + 1 < 9) frags(s + 1, fb ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                alo = TVC_MFMA16(af[fb][1], bf[fb][0], alo);
                acc = TVC_MFMA16(af[fb][0], bf[fb][0], acc);
                alo = TVC_MFMA16(af[fb][0], bf[fb][1], alo);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // The next tile's input is requested here, behind the conv's MFMAs (issuing the loads is a phase of the address path: between the
        // closing barriers it ran with the matrix pipe idle); unconditional - the last tile re-reads itself - so that the waits for the
        // older cond / residual loads stay exact counts.
        fetch(next < tend ? next : tile);
        {   // the conv result without its bias
            const float c = cw * sx.inv, cl = c * kLoInv;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = comb(acc[r], alo[r], c, cl);
        }
        // ---- FiLM scale / shift over cond: 3 K16 steps ----------------------------------------------------
        f32x16 asc, ash;
        if (FILM) {
            f32x16 lsc, lsh;
#pragma unroll
            for (int r = 0; r < 16; ++r) asc[r] = ash[r] = lsc[r] = lsh[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
#pragma unroll
                for (int j = 0; j < 8; ++j) cr[s][j] *= sc.s;
                uint4 p1, p2;
                split8(cr[s], p1, p2);
                const f16x8 cf[2] = {__builtin_bit_cast(f16x8, p1), __builtin_bit_cast(f16x8, p2)};
                f16x8 fa[2][2];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    fa[0][p] = __builtin_bit_cast(f16x8, Ft[((s * 4 + mt) * 2 + p) * 64 + lane]);
                    fa[1][p] = __builtin_bit_cast(f16x8, Ft[((s * 4 + 2 + mt) * 2 + p) * 64 + lane]);
                }
                lsc = TVC_MFMA16(fa[0][1], cf[0], lsc);
                lsh = TVC_MFMA16(fa[1][1], cf[0], lsh);
                asc = TVC_MFMA16(fa[0][0], cf[0], asc);
                ash = TVC_MFMA16(fa[1][0], cf[0], ash);
                lsc = TVC_MFMA16(fa[0][0], cf[1], lsc);
                lsh = TVC_MFMA16(fa[1][0], cf[1], lsh);
            }
            const float c1 = cwsc * sc.inv, c2 = cwsh * sc.inv;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                asc[r] = comb(asc[r], lsc[r], c1, c1 * kLoInv);
                ash[r] = comb(ash[r], lsh[r], c2, c2 * kLoInv);
            }
        }

        // ---- epilogue ---------------------------------------------------------------------------------------
        float xv[C5 ? 4 : 1][4];
        float mx = 0.f;                   // |max| of what this lane stores (C5: of its part of the finished tile)
        {
            float* ob = C5 ? nullptr : (RAG ? a.out + 8L * rt.off : a.out + (long)b * C * rs);
            const unsigned og = 32u * (unsigned)(4 * mt * rs + tc) + 16u * (unsigned)lh;      // G8: group 4 mt + g, column, this lane's four channels
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (32 * mt + 8 * g >= C) continue;                                // uniform per wave
                float v4[4];
                const f32x4s_t bv = *reinterpret_cast<const f32x4s_t*>(Bi + 32 * mt + 8 * g + 4 * lh);
                f32x4s_t bs, bh;
                if (FILM) {
                    bs = *reinterpret_cast<const f32x4s_t*>(Bi + 64 + 32 * mt + 8 * g + 4 * lh);
                    bh = *reinterpret_cast<const f32x4s_t*>(Bi + 128 + 32 * mt + 8 * g + 4 * lh);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v = acc[4 * g + q] + bv[q];
                    if (FILM) v = __fadd_rn(__fmul_rn(v, asc[4 * g + q] + bs[q]), ash[4 * g + q] + bh[q]);
                    if (RES) v = __fadd_rn(v, rv[g][q]);
                    v4[q] = v;
                    if (C5) {
                        xv[g][q] = v;
                        mx = fmaxf(mx, fabsf(v));
                    }
                }
                if (!C5 && t < len) {
                    stg_so4(ob + (long)g * 8 * rs, og, v4);
                    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v4[0]), fabsf(v4[1])), fmaxf(fabsf(v4[2]), fabsf(v4[3]))));
                }
            }
        }
        if (!C5) mx_run = fmaxf(mx_run, mx);
        if (C5) {
            // the finished 48 x 128 tile goes back into the (now idle) input tile as c5's B operand: split, rows
            // [part][group 4 mt + g][column], this lane's four channels are one 8-byte half of a row.
            // Per-tile power-of-two pre-scale: the tile's |max| meets in LDS (one ds_max per wave) behind the barrier that is
            // needed anyway, so the split never leaves fp16's range whatever the block produced.
            {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
                if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(Bi + 224), __builtin_bit_cast(unsigned, mx));
            }
            slab_barrier();                               // every wave is done reading Xs; the tile |max| is complete
            const Bfp s5 = bfp_from_amax(Bi[224]);
            // (a row = [lanes 0-31's four channels | lanes 32-63's four]: 8-byte stores from the two lane halves are 2-way bank
            // conflicts; v_permlane32_swap gives the lower half of the wave both halves of one row and the upper half both halves of
            // another - part 1 / part 2 of a group - so every store is 16 bytes)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (32 * mt + 8 * g >= C) continue;
                u32x2_t p1, p2;
#pragma unroll
                for (int q = 0; q < 4; ++q) xv[g][q] *= s5.s;
                split4_48(xv[g], p1, p2);
                const auto sx_ = __builtin_amdgcn_permlane32_swap(p1[0], p2[0], false, false);
                const auto sy_ = __builtin_amdgcn_permlane32_swap(p1[1], p2[1], false, false);
                const u32x4 row = {sx_[0], sy_[0], sx_[1], sy_[1]};
                *reinterpret_cast<u32x4*>(Xs + (6 * lh + 4 * mt + g) * XP + n) = row;
            }
            slab_barrier();
            if (tid == 0) Bi[224] = 0.f;                  // next use is behind the next tile's barriers
            if (mt == 0) {                                // 24 output rows = one m-tile: the first four waves, one n-tile each
                f32x16 a5, l5;
#pragma unroll
                for (int r = 0; r < 16; ++r) a5[r] = l5[r] = 0.f;
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    f16x8 fa[2], fb[2];
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        fb[p] = __builtin_bit_cast(f16x8, Xs[(p * 6 + 2 * s + lh) * XP + n]);
                        fa[p] = __builtin_bit_cast(f16x8, W5t[(s * 2 + p) * 64 + lane]);
                    }
                    l5 = TVC_MFMA16(fa[1], fb[0], l5);
                    a5 = TVC_MFMA16(fa[0], fb[0], a5);
                    l5 = TVC_MFMA16(fa[0], fb[1], l5);
                }
                float m5 = 0.f;
                if (t < len) {
                    float* o5 = RAG ? a.out5 + 8L * rt.off : a.out5 + (long)b * 24 * rs;
                    const unsigned o5o = 32u * (unsigned)t + 16u * (unsigned)lh;      // this lane's four channels of group g: 16 bytes
                    const float c = cw5 * s5.inv, cl = c * kLoInv;
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        const f32x4s_t b5v = *reinterpret_cast<const f32x4s_t*>(Bi + 192 + 8 * g + 4 * lh);
                        float v4[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            v4[q] = comb(a5[4 * g + q], l5[4 * g + q], c, cl) + b5v[q];
                            m5 = fmaxf(m5, fabsf(v4[q]));
                        }
                        stg_so4(o5 + (long)g * 8 * rs, o5o, v4);
                    }
                }
                mx_run = fmaxf(mx_run, m5);
            }
        }
        // ---- next tile's input: registers -> LDS ---------------------------------------------------------------
        slab_barrier();                                   // every wave is done reading Xs
        if (next < tend) {
            const int bn = utt(next);
            deposit(bn == b ? sx.s : bfp_load_u(a.amax_x, bn).s);
        }
        slab_barrier();
    }
    if (a.amax_y) amax_flush_wg(a.amax_y + mx_b, mx_run, Bi + 228);
}

template <bool FILM, bool LERP, int RES, bool C5 = false>
int launch48(tvc_ctx* ctx, hipStream_t s, Conv48Args a, int B) {
    static int ncu_dev[64] = {};
    int& ncu = ncu_dev[ctx->device & 63];
    constexpr size_t lds = (size_t)(12 * kXP48 + 36 * 64 + (FILM ? 24 * 64 : 0) + (C5 ? 6 * 64 : 0)) * 16 + 240 * 4;
    if (!ncu) {
        hipDeviceProp_t prop;
        hipError_t e = hipGetDeviceProperties(&prop, ctx->device);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv48s_kernel<FILM, LERP, RES, C5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv48s_kernel<FILM, LERP, RES, C5, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(ctx, TVC_ERR_HIP, "conv48s setup: %s", hipGetErrorString(e));
        ncu = prop.multiProcessorCount;
    }
    a.tiles_per_utt = (a.len + kBN48 - 1) / kBN48;
    a.ntiles = a.tiles_per_utt * B;
    a.rag = RagDev{};
    if (ctx->rag) {
        if (B != 1 || a.len % ctx->rag->Ttot != 0) return fail(ctx, TVC_ERR_STATE, "conv48s: a ragged batch runs as one long utterance");
        TVC_CHECK(rag_view(ctx, s, a.len / ctx->rag->Ttot, kBN48, &a.rag, &a.ntiles));
    }
    const int wpc = FILM ? 1 : 2;                     // persistent workgroups per CU
    const int grid = a.ntiles < wpc * ncu ? a.ntiles : wpc * ncu;
    if (ctx->rag) hipLaunchKernelGGL((conv48s_kernel<FILM, LERP, RES, C5, true>), dim3(grid), dim3(kNT48), lds, s, a);
    else hipLaunchKernelGGL((conv48s_kernel<FILM, LERP, RES, C5>), dim3(grid), dim3(kNT48), lds, s, a);
    return launch_check(ctx, "conv48s");
}

// ---------------------------------------------------------------------------------------------------------------------
// Two consecutive 48-channel convs in ONE kernel: out = conv_b(lrelu(conv_a(lrelu(x)) + b_a)) [FiLM(cond), + residual], the
// first conv's output never leaves the CU (Upsample 3: c1 -> c2 + FiLM1 + x_up, decoder.py:173-182; Downsample 2: c1 -> c2,
// decoder.py:150-155).  The second conv's halo is 2 * dil_b samples of a 128-column tile (5 % recompute at dil_b = 3; the
// c3 -> c4 pair, dil 27, would recompute 42 %: not fused), so a tile is 128 columns of the intermediate h and 128 - 2 dil_b
// output columns.  Both convs' weights (and FiLM's) are resident in LDS; h is split into the second conv's operand tile by
// the first conv's epilogue with an exact per-tile power-of-two pre-scale (the tile's |max| meets in LDS behind the barrier
// between the two convs); replicate padding of h at the utterance ends = a column clamp when the second conv reads it.
struct Conv48PArgs {
    const float* x;        // planar [B][48][len], or (LERP) the low-rate level output in the G8 layout [B][6][lin][8]
    const float* cond;     // FILM: G8 [B][6][len][8] (Conv48Args)
    const float* res;      // RES 2: the low-rate level output again (G8), interpolated here
    float* out;            // FILM: G8 (Upsample 3's x1); else planar [B][48][len]
    const u32x4* Aa;       // first / second conv image, 36 pieces each
    const u32x4* Ab;
    const u32x4* F6;       // stacked FiLM image, 24 pieces
    const float *bias_a, *bias_b, *bsc, *bsh;
    const float *wsc_a, *wsc_b, *fsc;
    const float* amax_x;
    const float* amax_c;
    float* amax_y;
    RagDev rag;            // RAG kernels (ragged.h): len / lin / rlin = row strides of the batch-wide tensors, tiles / extents from the table
    int len, da, db, lin, rlin, tiles_per_utt, ntiles;
    float lscale, rscale;
};
constexpr int kXPP = 136;      // input tile columns: 128 + 2 * dil_a (dil_a <= 4)

template <bool FILM, bool LERP, int RES, bool RAG = false>
__global__ __launch_bounds__(kNT48) __attribute__((amdgpu_waves_per_eu(2))) void conv48p_kernel(Conv48PArgs a) {
    constexpr int C = kC48, NT = kNT48, XP = kXPP, HP = 128;
    extern __shared__ __attribute__((aligned(16))) uint4 smem_p[];
    u32x4* Xs = reinterpret_cast<u32x4*>(smem_p);              // [2 parts][6 groups][XP]: lrelu(x), split
    u32x4* Hs = Xs + 12 * XP;                                  // [2 parts][6 groups][HP]: lrelu(conv_a + b_a), split
    u32x4* Wa = Hs + 12 * HP;                                  // 36 pieces
    u32x4* Wb = Wa + 36 * 64;                                  // 36 pieces
    u32x4* Ft = Wb + 36 * 64;                                  // 24 pieces (FILM)
    float* Bi = reinterpret_cast<float*>(Ft + (FILM ? 24 * 64 : 0));    // bias_a, bias_b, bsc, bsh [64 each], [256] = the h tile's |max|, [260..267] = |max| exchange
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int mt = wave >> 2, nt = wave & 3;
    const int rs = a.len, da = a.da, db = a.db;                 // rs = row stride of cond / out (= every utterance's length unless RAG, ragged.h)
    const int BNO = 128 - 2 * db;                              // output columns per tile
    const int XW = 128 + 2 * da;
    const int rsl = LERP ? a.lin : rs;                         // row stride of x
    const int xf = LERP ? rs / a.lin : 1, rf = RES == 2 ? rs / a.rlin : 1;     // RAG: an utterance's low-rate lengths = len / xf, len / rf
    int bh = 0;                                                // RAG: utterance hint of the table walk
    auto utt = [&](int tile) __attribute__((always_inline)) { return rag_tile<RAG>(a.rag, tile, a.tiles_per_utt, rs, bh).b; };

    for (int i = tid; i < 36 * 64; i += NT) {
        Wa[i] = a.Aa[i];
        Wb[i] = a.Ab[i];
    }
    if (FILM)
        for (int i = tid; i < 24 * 64; i += NT) Ft[i] = a.F6[i];
    if (tid < 64) {
        const int tc = tid < C ? tid : C - 1;
        Bi[tid] = a.bias_a[tc];
        Bi[64 + tid] = a.bias_b[tc];
        if (FILM) {
            Bi[128 + tid] = a.bsc[tc];
            Bi[192 + tid] = a.bsh[tc];
        }
    }
    if (tid == 0) Bi[256] = 0.f;
    // (wave-uniform: read through the scalar cache into scalar registers - as vector registers the four were spilled by the FiLM instantiation and
    // reloaded, s_waitcnt vmcnt(0), in the middle of every tile)
    const int mtu = __builtin_amdgcn_readfirstlane(mt);
    const float cwa = a.wsc_a[mtu], cwb = a.wsc_b[mtu], cwsc = FILM ? a.fsc[mtu] : 1.f, cwsh = FILM ? a.fsc[2 + mtu] : 1.f;

    // staging items (8-channel group, column): 6 * XW <= 816 of them, two per thread
    constexpr int XPER = 2;
    float xr0[XPER][8], xr1[LERP ? XPER : 1][8], lam[LERP ? XPER : 1];
    int ig[XPER], ic[XPER];
#pragma unroll
    for (int i = 0; i < XPER; ++i) {
        const int idx = tid + i * NT;
        const int g = idx / XW;
        ic[i] = idx - g * XW;
        ig[i] = g;                                             // g >= 6: idle item (loads a valid address, never stores)
    }
    auto fetch = [&](int tile) __attribute__((always_inline)) {
        const RagTile rt = rag_tile<RAG>(a.rag, tile, a.tiles_per_utt, rs, bh);
        const int len = rt.len, lin = LERP ? len / xf : len;
        const int px0 = rt.tin * BNO - db - da;
        const float* xb = RAG ? a.x + (LERP ? 8L * (rt.off / xf) : rt.off) : a.x + (long)rt.b * C * rsl;      // LERP: the low-rate level output, G8 (EpiBiasG8)
#pragma unroll
        for (int i = 0; i < XPER; ++i) {
            const int g = ig[i] > 5 ? 5 : ig[i];
            int p = px0 + ic[i];
            p = p < 0 ? 0 : (p > len - 1 ? len - 1 : p);
            if (LERP) {
                const Lerp lc = lerp_coord(p, a.lscale, lin);
                lam[i] = lc.w1;
                ld8_g8(xr0[i], xb, 32u * (unsigned)(g * rsl + lc.i0));
                ld8_g8(xr1[i], xb, 32u * (unsigned)(g * rsl + lc.i1));
            } else {
                const unsigned o = 4u * (unsigned)(8 * g * rsl + p);
#pragma unroll
                for (int j = 0; j < 8; ++j) xr0[i][j] = ldg_so(xb + (long)j * rsl, o);
            }
        }
    };
    auto deposit = [&](float xs) __attribute__((always_inline)) {
        // (the items' LDS rows are recomputed from a laundered thread index: as kernel-long invariants they were spilled, and their reload -
        // s_waitcnt vmcnt(0) - waited for the tile's stores)
        int td = tid;
        asm volatile("" : "+v"(td));
#pragma unroll
        for (int i = 0; i < XPER; ++i) {
            const int idx = td + i * NT;
            const int g = idx / XW, c = idx - g * XW;
            if (g > 5) continue;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = LERP ? fmaf(1.f - lam[i], xr0[i][j], __fmul_rn(lam[i], xr1[i][j])) : xr0[i][j];   // = lerp_eval
                v[j] = fmaxf(t, 0.1f * t) * xs;                                                             // = leaky_relu(x, 0.1), scaled
            }
            uint4 p1, p2;
            split8(v, p1, p2);
            Xs[(0 + g) * XP + c] = __builtin_bit_cast(u32x4, p1);
            Xs[(6 + g) * XP + c] = __builtin_bit_cast(u32x4, p2);
        }
    };

    int tile, tend;
    tile_range(a.ntiles, tile, tend);
    if (tile >= tend) return;
    bh = utt(tile);
    fetch(tile);
    deposit(bfp_load_u(a.amax_x, bh).s);
    slab_barrier();
    float mx_run = 0.f;
    int mx_b = bh;
    int sc_b = -1;      // the utterance whose scales sx / sc hold: computed when the walk enters it, not per tile
    Bfp sx{1.f, 1.f}, sc{1.f, 1.f};

    for (; tile < tend; ++tile) {
        const RagTile rt = rag_tile<RAG>(a.rag, tile, a.tiles_per_utt, rs, bh);
        const int b = rt.b, len = rt.len;
        bh = b;
        if (a.amax_y && b != mx_b) {
            amax_flush_wg(a.amax_y + mx_b, mx_run, Bi + 260);
            mx_run = 0.f;
            mx_b = b;
        }
        if (b != sc_b) {
            sc_b = b;
            sx = bfp_load_u(a.amax_x, b);
            sc = FILM ? bfp_load_u(a.amax_c, b) : Bfp{1.f, 1.f};
        }
        const int t0 = rt.tin * BNO;
        const int next = tile + 1;
        const int n = nt * 32 + l31;                       // this lane's column: of h in the first conv, of the output in the second
        const int t = t0 + n;
        const bool live = n < BNO && t < len;
        const int tc = t < len ? t : len - 1;
        const unsigned oo = 4u * (unsigned)((32 * mt + 4 * lh) * rs + tc);

        float cr[FILM ? 3 : 1][8];
        if (FILM) {
            const float* cb = RAG ? a.cond + 8L * rt.off : a.cond + (long)b * C * rs;      // G8 (Conv48Args)
            const unsigned oc = 32u * (unsigned)(lh * rs + tc);                     // group 2 s + lh, column tc
#pragma unroll
            for (int s = 0; s < 3; ++s) ld8_g8(cr[s], cb + (long)s * 16 * rs, oc);
        }
        float rv[RES ? 4 : 1][4];
        if (RES == 2) {      // the interpolated residual: the low-rate level output again (G8)
            const float* rb = RAG ? a.res + 8L * (rt.off / rf) : a.res + (long)b * C * a.rlin;
            const Lerp lc = lerp_coord(tc, a.rscale, RAG ? len / rf : a.rlin);
            const unsigned o0 = 32u * (unsigned)(4 * mt * a.rlin + lc.i0) + 16u * (unsigned)lh, o1 = 32u * (unsigned)(4 * mt * a.rlin + lc.i1) + 16u * (unsigned)lh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float x0[4] = {0.f, 0.f, 0.f, 0.f}, x1[4] = {0.f, 0.f, 0.f, 0.f};
                if (32 * mt + 8 * g < C) {
                    ld4_g8(x0, rb + (long)g * 8 * a.rlin, o0);
                    ld4_g8(x1, rb + (long)g * 8 * a.rlin, o1);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) rv[g][q] = fmaf(lc.w0, x0[q], __fmul_rn(lc.w1, x1[q]));      // = lerp_eval
            }
        }

        // ---- first conv: h column n (position t0 - db + n) from Xs columns n + tap * da -------------------------------------
        f32x16 acc, alo;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = alo[r] = 0.f;
        {
            f16x8 af[2][2], bf[2][2];
            auto frags = [&](int s, int fb) __attribute__((always_inline)) {
                const int sl = s / 3, tap = s - sl * 3;
                const int row = (2 * sl + lh) * XP + n + tap * da;
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    bf[fb][p] = __builtin_bit_cast(f16x8, Xs[p * 6 * XP + row]);
                    af[fb][p] = __builtin_bit_cast(f16x8, Wa[((s * 2 + mt) * 2 + p) * 64 + lane]);
                }
            };
            frags(0, 0);
#pragma unroll
            for (int s = 0; s < 9; ++s) {
                const int fb = s & 1;
                if (s + 1 < 9) frags(s + 1, fb ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                alo = TVC_MFMA16(af[fb][1], bf[fb][0], alo);
                acc = TVC_MFMA16(af[fb][0], bf[fb][0], acc);
                alo = TVC_MFMA16(af[fb][0], bf[fb][1], alo);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        fetch(next < tend ? next : tile);      // the next tile's input, requested behind the first conv's MFMAs (conv48s_kernel has the note)
        // h = lrelu(conv_a + b_a); its tile |max| -> LDS; after the barrier: pre-scale, split, rows [part][group 4 mt + g][column]
        float hv[4][4];
        float hmx = 0.f;
        {
            const float c = cwa * sx.inv, cl = c * kLoInv;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4s_t bv = *reinterpret_cast<const f32x4s_t*>(Bi + 32 * mt + 8 * g + 4 * lh);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v = comb(acc[4 * g + q], alo[4 * g + q], c, cl) + bv[q];
                    hv[g][q] = fmaxf(v, 0.1f * v);
                    if (32 * mt + 8 * g < C) hmx = fmaxf(hmx, fabsf(hv[g][q]));
                }
            }
            hmx = wave_max(hmx);
            if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(Bi + 256), __builtin_bit_cast(unsigned, hmx));
        }
        slab_barrier();                                   // the h tile's |max| is complete (Hs itself was last read before the previous tile's closing barriers)
        const Bfp sh_ = bfp_from_amax(Bi[256]);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (32 * mt + 8 * g >= C) continue;
            u32x2_t p1, p2;
#pragma unroll
            for (int q = 0; q < 4; ++q) hv[g][q] *= sh_.s;
            split4_48(hv[g], p1, p2);
            const auto sx_ = __builtin_amdgcn_permlane32_swap(p1[0], p2[0], false, false);
            const auto sy_ = __builtin_amdgcn_permlane32_swap(p1[1], p2[1], false, false);
            const u32x4 row = {sx_[0], sy_[0], sx_[1], sy_[1]};
            *reinterpret_cast<u32x4*>(Hs + (6 * lh + 4 * mt + g) * HP + n) = row;
        }
        slab_barrier();                                   // Hs is complete; every wave has read the tile |max|
        if (tid == 0) Bi[256] = 0.f;                      // (its next use is behind the next tile's barriers)

        // ---- second conv: output column n from Hs columns n + tap * db, clamped to the utterance (replicate padding of h) ----
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = alo[r] = 0.f;
        {
            const int lo = db - t0 > 0 ? db - t0 : 0;                                  // h column of position 0
            const int hi = len - 1 - t0 + db < HP - 1 ? len - 1 - t0 + db : HP - 1;    // ... of position len - 1
            int col[3];
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                int c = n + tap * db;
                c = c > HP - 1 ? HP - 1 : c;              // (lanes past the tile's output columns: any valid column)
                col[tap] = c < lo ? lo : (c > hi ? hi : c);
            }
            f16x8 af[2][2], bf[2][2];
            auto frags = [&](int s, int fb) __attribute__((always_inline)) {
                const int sl = s / 3, tap = s - sl * 3;
                const int row = (2 * sl + lh) * HP + col[tap];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    bf[fb][p] = __builtin_bit_cast(f16x8, Hs[p * 6 * HP + row]);
                    af[fb][p] = __builtin_bit_cast(f16x8, Wb[((s * 2 + mt) * 2 + p) * 64 + lane]);
                }
            };
            frags(0, 0);
#pragma unroll
            for (int s = 0; s < 9; ++s) {
                const int fb = s & 1;
                if (s + 1 < 9) frags(s + 1, fb ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                alo = TVC_MFMA16(af[fb][1], bf[fb][0], alo);
                acc = TVC_MFMA16(af[fb][0], bf[fb][0], acc);
                alo = TVC_MFMA16(af[fb][0], bf[fb][1], alo);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        {
            const float c = cwb * sh_.inv, cl = c * kLoInv;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = comb(acc[r], alo[r], c, cl);
        }
        f32x16 asc, ash;
        if (FILM) {
            f32x16 lsc, lsh;
#pragma unroll
            for (int r = 0; r < 16; ++r) asc[r] = ash[r] = lsc[r] = lsh[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
#pragma unroll
                for (int j = 0; j < 8; ++j) cr[s][j] *= sc.s;
                uint4 p1, p2;
                split8(cr[s], p1, p2);
                const f16x8 cf[2] = {__builtin_bit_cast(f16x8, p1), __builtin_bit_cast(f16x8, p2)};
                f16x8 fa[2][2];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    fa[0][p] = __builtin_bit_cast(f16x8, Ft[((s * 4 + mt) * 2 + p) * 64 + lane]);
                    fa[1][p] = __builtin_bit_cast(f16x8, Ft[((s * 4 + 2 + mt) * 2 + p) * 64 + lane]);
                }
                lsc = TVC_MFMA16(fa[0][1], cf[0], lsc);
                lsh = TVC_MFMA16(fa[1][1], cf[0], lsh);
                asc = TVC_MFMA16(fa[0][0], cf[0], asc);
                ash = TVC_MFMA16(fa[1][0], cf[0], ash);
                lsc = TVC_MFMA16(fa[0][0], cf[1], lsc);
                lsh = TVC_MFMA16(fa[1][0], cf[1], lsh);
            }
            const float c1 = cwsc * sc.inv, c2 = cwsh * sc.inv;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                asc[r] = comb(asc[r], lsc[r], c1, c1 * kLoInv);
                ash[r] = comb(ash[r], lsh[r], c2, c2 * kLoInv);
            }
        }
        // ---- epilogue ---------------------------------------------------------------------------------------------------
        {
            float mx = 0.f;
            // the FiLM launch's output is Upsample 3's x1: G8 (conv48s_kernel reads it); the plain launch's (Downsample 2's h2) stays planar for conv3s
            float* ob = RAG ? a.out + (FILM ? 8L : 1L) * rt.off : a.out + (long)b * C * rs;
            const unsigned og = 32u * (unsigned)(4 * mt * rs + tc) + 16u * (unsigned)lh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (32 * mt + 8 * g >= C) continue;                                // uniform per wave
                float v4[4];
                const f32x4s_t bv = *reinterpret_cast<const f32x4s_t*>(Bi + 64 + 32 * mt + 8 * g + 4 * lh);
                f32x4s_t bs, bh;
                if (FILM) {
                    bs = *reinterpret_cast<const f32x4s_t*>(Bi + 128 + 32 * mt + 8 * g + 4 * lh);
                    bh = *reinterpret_cast<const f32x4s_t*>(Bi + 192 + 32 * mt + 8 * g + 4 * lh);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v = acc[4 * g + q] + bv[q];
                    if (FILM) v = __fadd_rn(__fmul_rn(v, asc[4 * g + q] + bs[q]), ash[4 * g + q] + bh[q]);
                    if (RES) v = __fadd_rn(v, rv[g][q]);
                    v4[q] = v;
                    if (!FILM && live) stg_so(ob + (long)(8 * g + q) * rs, oo, v);
                }
                if (live) {
                    if (FILM) stg_so4(ob + (long)g * 8 * rs, og, v4);
                    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v4[0]), fabsf(v4[1])), fmaxf(fabsf(v4[2]), fabsf(v4[3]))));
                }
            }
            mx_run = fmaxf(mx_run, mx);
        }
        // ---- next tile's input: registers -> LDS -------------------------------------------------------------------------------
        slab_barrier();                                   // every wave is done reading Xs and Hs
        if (next < tend) {
            const int bn = utt(next);
            deposit(bn == b ? sx.s : bfp_load_u(a.amax_x, bn).s);
        }
        slab_barrier();
    }
    if (a.amax_y) amax_flush_wg(a.amax_y + mx_b, mx_run, Bi + 260);
}

template <bool FILM, bool LERP, int RES>
int launch48p(tvc_ctx* ctx, hipStream_t s, Conv48PArgs a, int B) {
    static int ncu_dev[64] = {};
    int& ncu = ncu_dev[ctx->device & 63];
    constexpr size_t lds = (size_t)(12 * kXPP + 12 * 128 + 72 * 64 + (FILM ? 24 * 64 : 0)) * 16 + 272 * 4;
    static_assert(lds <= 160 * 1024, "LDS");
    if (!ncu) {
        hipDeviceProp_t prop;
        hipError_t e = hipGetDeviceProperties(&prop, ctx->device);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv48p_kernel<FILM, LERP, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv48p_kernel<FILM, LERP, RES, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(ctx, TVC_ERR_HIP, "conv48p setup: %s", hipGetErrorString(e));
        ncu = prop.multiProcessorCount;
    }
    const int bno = 128 - 2 * a.db;
    a.tiles_per_utt = (a.len + bno - 1) / bno;
    a.ntiles = a.tiles_per_utt * B;
    a.rag = RagDev{};
    if (ctx->rag) {
        if (B != 1 || a.len % ctx->rag->Ttot != 0) return fail(ctx, TVC_ERR_STATE, "conv48p: a ragged batch runs as one long utterance");
        TVC_CHECK(rag_view(ctx, s, a.len / ctx->rag->Ttot, bno, &a.rag, &a.ntiles));
    }
    const int grid = a.ntiles < ncu ? a.ntiles : ncu;
    if (ctx->rag) hipLaunchKernelGGL((conv48p_kernel<FILM, LERP, RES, true>), dim3(grid), dim3(kNT48), lds, s, a);
    else hipLaunchKernelGGL((conv48p_kernel<FILM, LERP, RES>), dim3(grid), dim3(kNT48), lds, s, a);
    return launch_check(ctx, "conv48p");
}

}  // namespace

// mode bits: 1 = the input is the low-rate tensor [B][48][lin] (F.interpolate fused into the staging), 2 = FiLM over cond,
// residual: rlin == 0 and res != nullptr -> direct, rlin > 0 -> F.interpolate(res low-rate)
int run_conv48s(tvc_ctx* ctx, hipStream_t s, const PackedW& w, const float* x, int lin, float lscale, const PackedW* film, const float* bsc,
                const float* bsh, const float* cond, const float* res, int rlin, float rscale, float* out, int B, int len, int dil, const float* amax_x,
                const float* amax_c, float* amax_y, const PackedW* c5, float* out5) {
    if (w.cin != kC48 || w.M != kC48 || w.taps != 3 || w.MT6 != 2 || !w.A6) return fail(ctx, TVC_ERR_ARG, "conv48s: 48 -> 48 channel k3 convs only");
    if (dil < 1 || dil > 27) return fail(ctx, TVC_ERR_ARG, "conv48s: dilation must be 1..27");
    if ((long)len * kC48 * 4 >= (1L << 32)) return fail(ctx, TVC_ERR_ARG, "conv48s: utterance too long for 32-bit byte offsets");
    if (film && (film->MT6 != 4 || !film->A6 || !cond || !bsc || !bsh)) return fail(ctx, TVC_ERR_ARG, "conv48s: FiLM needs the stacked image, both biases and cond");
    Conv48Args a{};
    a.x = x; a.cond = cond; a.res = res; a.out = out;
    a.A6 = reinterpret_cast<const u32x4*>(w.A6);
    a.F6 = film ? reinterpret_cast<const u32x4*>(film->A6) : nullptr;
    a.bias = w.bias; a.bsc = bsc; a.bsh = bsh;
    a.wsc = w.wscale;
    a.fsc = film ? film->wscale : nullptr;
    a.amax_x = amax_x; a.amax_c = amax_c; a.amax_y = amax_y;
    a.len = len; a.dil = dil; a.lin = lin; a.rlin = rlin; a.lscale = lscale; a.rscale = rscale;
    // (the level's tensors are in the G8 layout; the interpolating input / residual variants of this kernel read planar low-rate tensors and
    // are not launched any more: the first half of Upsample 3 is run_conv48_pair)
    if (lin > 0 || rlin > 0) return fail(ctx, TVC_ERR_ARG, "conv48s: interpolated inputs / residuals go through run_conv48_pair");
    if (film) {
        if (!res) return fail(ctx, TVC_ERR_ARG, "conv48s: the FiLM variants carry a residual");
        if (c5) {   // FiLM2 + residual + c5: only c5's 24 rows leave the CU
            if (rlin > 0 || !out5 || c5->M != 24 || c5->cin != kC48 || c5->taps != 1 || c5->MT6 != 1 || !c5->A6)
                return fail(ctx, TVC_ERR_ARG, "conv48s: the fused c5 is the 48 -> 24 1x1 after the second FiLM");
            a.W5 = reinterpret_cast<const u32x4*>(c5->A6);
            a.b5 = c5->bias;
            a.w5sc = c5->wscale;
            a.out5 = out5;
            return launch48<true, false, 1, true>(ctx, s, a, B);
        }
        return launch48<true, false, 1>(ctx, s, a, B);
    }
    if (res) return fail(ctx, TVC_ERR_ARG, "conv48s: plain convs carry no residual");
    return launch48<false, false, 0>(ctx, s, a, B);
}

// The fused pair (conv48p_kernel): x (or, lin > 0, the low-rate tensor it is interpolated from) -> conv_a(dil da) -> lrelu -> conv_b(dil db)
// [-> FiLM(cond) + F.interpolate(res_low)] -> out.  film == nullptr: plain pair (Downsample), no residual.
int run_conv48_pair(tvc_ctx* ctx, hipStream_t s, const PackedW& wa, const PackedW& wb, const float* x, int lin, float lscale, const PackedW* film,
                    const float* bsc, const float* bsh, const float* cond, const float* res, int rlin, float rscale, float* out, int B, int len, int da,
                    int db, const float* amax_x, const float* amax_c, float* amax_y) {
    for (const PackedW* w : {&wa, &wb})
        if (w->cin != kC48 || w->M != kC48 || w->taps != 3 || w->MT6 != 2 || !w->A6) return fail(ctx, TVC_ERR_ARG, "conv48 pair: 48 -> 48 channel k3 convs only");
    if (da < 1 || da > 4 || db < 1 || db > 8) return fail(ctx, TVC_ERR_ARG, "conv48 pair: dilations (1..4, 1..8)");
    if ((long)len * kC48 * 4 >= (1L << 32)) return fail(ctx, TVC_ERR_ARG, "conv48 pair: utterance too long for 32-bit byte offsets");
    if (film && (film->MT6 != 4 || !film->A6 || !cond || !bsc || !bsh || !res || rlin <= 0))
        return fail(ctx, TVC_ERR_ARG, "conv48 pair: the FiLM variant needs the stacked image, both biases, cond and the low-rate residual");
    Conv48PArgs a{};
    a.x = x; a.cond = cond; a.res = res; a.out = out;
    a.Aa = reinterpret_cast<const u32x4*>(wa.A6);
    a.Ab = reinterpret_cast<const u32x4*>(wb.A6);
    a.F6 = film ? reinterpret_cast<const u32x4*>(film->A6) : nullptr;
    a.bias_a = wa.bias; a.bias_b = wb.bias; a.bsc = bsc; a.bsh = bsh;
    a.wsc_a = wa.wscale; a.wsc_b = wb.wscale; a.fsc = film ? film->wscale : nullptr;
    a.amax_x = amax_x; a.amax_c = amax_c; a.amax_y = amax_y;
    a.len = len; a.da = da; a.db = db; a.lin = lin; a.rlin = rlin; a.lscale = lscale; a.rscale = rscale;
    if (film) return lin > 0 ? launch48p<true, true, 2>(ctx, s, a, B) : fail(ctx, TVC_ERR_ARG, "conv48 pair: the FiLM pair starts from the low-rate tensor");
    if (lin > 0 || res) return fail(ctx, TVC_ERR_ARG, "conv48 pair: the plain pair has neither an interpolated input nor a residual");
    return launch48p<false, false, 0>(ctx, s, a, B);
}

}  // namespace tvc
