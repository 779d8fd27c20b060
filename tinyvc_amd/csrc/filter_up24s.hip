// The fused 24-channel full-rate block (FilterNet ups[4] + output_layer, decoder.py:173-190,220,233): every conv / FiLM 1x1
// runs on v_mfma_f32_32x32x16_f16 with both operands split into two fp16 parts (three part-products into an accumulator
// pair, fp32-equivalent accuracy, block-floating-point range guard: conv3s.h).
//
//   half A:  x_up = interp(x, x5) -> lrelu -> c1(d1) -> lrelu -> c2(d3) -> FiLM1(cond) -> + x_up          => x1
//   half B:  x1 -> lrelu -> c3(d9) -> lrelu -> c4(d27) -> FiLM2(cond) -> + x1 -> [c5 . output k7 folded]   => wave
//
// One persistent 8-wave workgroup per CU walks tiles of W output samples.
//   LDS     Xs[part][8-channel group][position][8 fp16]: lrelu(input tile), split while it is deposited;
//           Hs, same layout: lrelu(first conv + bias), split by the first conv's epilogue;
//           R[24][PS] fp32: the raw input tile (residual); half B overwrites it in place with x2 for the output conv;
//           Wt: the block's weights, pre-split on the host into MFMA A-lane order (api.hip up24s_half), resident.
//   K order a 24-channel k3 conv has 9 (tap, 8-channel group) units; a K16 step takes two of them, one per lane half
//           (5 steps, the 10th unit has zero weights).  A tap is a row offset in Xs / Hs, so every B fragment is one
//           ds_read_b128 of a contiguous 512-byte run per lane half.
//   tiles   a wave owns 32 output columns x all 24 (padded 32) rows: 15 MFMAs per conv, 12 for FiLM's scale and shift
//           (cond fragments are loaded from HBM straight into B-fragment order and split in registers).
//   HBM     per tile: input tile + halo and cond in, one tile out; the next tile's input is in flight in registers
//           across the whole tile (raw s_barrier: __syncthreads would wait for it).
#include "conv3s.h"
#include "conv_s2.h"
#include "small_kernels.h"
#include "tvc_common.h"

namespace tvc {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4s __attribute__((ext_vector_type(4)));

constexpr int U24S_WPE = 2;     // 8 waves per CU: 256 registers each

template <int W_, int D1_, int D2_, bool SECOND_, int E_, int NWAVES_ = 8>
struct U24S {
    static constexpr int C = 24, W = W_, D1 = D1_, D2 = D2_, E = E_, H = D1_ + D2_;
    static constexpr bool SECOND = SECOND_;
    static constexpr int NWAVES = NWAVES_, NT = 64 * NWAVES_;
    static constexpr int WGS_PER_CU = 8 / NWAVES_;             // two waves per SIMD (256 registers each)
    static constexpr int W2 = W + 2 * E;                       // columns the second conv must produce (position t0 - E + n)
    static constexpr int NT2 = (W2 + 31) / 32, W2r = NT2 * 32;
    static constexpr int NT1 = (W2 + 2 * D2 + 31) / 32;        // first-conv column tiles (position t0 - E - D2 + h)
    static constexpr int HP = NT1 * 32;                        // Hs rows per (part, group)
    static constexpr int XW = W2 + 2 * H;                      // input columns (position t0 - E - H + c)
    static constexpr int XP = XW;
    static constexpr int PS = W2r + 16;                        // fp32 residual tile row stride: 272 = the one stride <= 300 for which the output pass's 16-byte reads (8 lanes x 3 rows apart, 4 columns per lane) are conflict-free in all four ds_read_b128 lane groups (W2r + 4 was 3-way)
    static constexpr int ITEMS = 3 * XW, XPER = (ITEMS + NT - 1) / NT;
    static constexpr int PIECES = 28, FL = 320;      // floats behind the pieces: the blob's 304, then 8 for the |max| exchange
    static constexpr int LDS_BYTES = (6 * XP + 6 * HP + PIECES * 64) * 16 + (FL + C * PS) * 4;
    static_assert(NT2 <= NWAVES, "one second-conv tile per wave");
    static_assert(XW - H >= W2, "residual columns");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

struct Up24SArgs {
    const float* x;      // half A: low-rate input; half B: x1.  Both in the G8 layout [B][3 groups][columns][8 channels] fp32 (a staged item - 8 channels of one
                         // position - is 32 contiguous bytes: two 16-byte loads instead of eight strided 4-byte ones; the producers store 16 bytes per lane and group)
    const uint4* cond;   // skips[0] as down0s_kernel writes it: the FiLM 1x1s' READY B operand, two fp16 planes [B][part][3 groups][len][8 fp16] of
                         // cond * 2^k, k from the bound cbw |max of downs.0's input| + cbb >= |cond| (amax_c = that input's slot; both kernels evaluate it alike)
    float cbw, cbb;
    float* out;          // half A: x1 (G8 layout); half B: waveform [B][len]
    const u32x4* img;    // weight blob (api.hip up24s_half)
    int len, xf, tiles_per_utt, ntiles;
    float interp_scale;
    // block-floating-point guard (conv3s.h): per-utterance |max| slots of x / of the tensor cond was computed from (read, nullable) and of `out` (half A: written, nullable)
    const float* amax_x;
    const float* amax_c;
    float* amax_y;
    RagDev rag;          // RAG kernels (ragged.h): `len` = row stride of the batch-wide tensors, tiles / extents from the table; the second half
                         // writes utterance b's waveform to row rag.row[b] of the caller's padded [rows][Tmax * 480] output
};

// two fp16 parts of 4 fp32 values (8 bytes each)
__device__ __forceinline__ void split4(const float (&v)[4], u32x2& p1, u32x2& p2) {
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
    const f32x4s q0 = __builtin_bit_cast(f32x4s, ldg_so4(b4, byte_off)), q1 = __builtin_bit_cast(f32x4s, ldg_so4(b4, byte_off + 16u));
    dst[0] = q0.x; dst[1] = q0.y; dst[2] = q0.z; dst[3] = q0.w;
    dst[4] = q1.x; dst[5] = q1.y; dst[6] = q1.z; dst[7] = q1.w;
}

// (hi, lo) += W (.) src over the 9 (tap, group) units of a 24-channel k3 conv for this wave's 32 columns.
// src = Xs / Hs ([part][group][P rows]), wt = the conv's 10 pieces, col = this lane's column of tap 0, clamped to [lo_c, hi_c].
template <int P, int DIL>
__device__ __forceinline__ void conv24_phase(f32x16& hi, f32x16& lo, const u32x4* src, const u32x4* wt, int col, int lo_c, int hi_c, int lane) {
    const int lh = lane >> 5;
    // unit u = 2 s + lh -> tap u / 3, group u % 3 (u = 9: zero weights, any valid row)
    constexpr int TAP0[5] = {0, 0, 1, 2, 2}, GRP0[5] = {0, 2, 1, 0, 2};
    constexpr int TAP1[5] = {0, 1, 1, 2, 2}, GRP1[5] = {1, 0, 2, 1, 2};
    int row[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        int c = col + (lh ? TAP1[s] : TAP0[s]) * DIL;
        c = c < lo_c ? lo_c : (c > hi_c ? hi_c : c);
        row[s] = (lh ? GRP1[s] : GRP0[s]) * P + c;
    }
    f16x8 af[2][2], bf[2][2];
    auto frags = [&](int s, int fb) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            bf[fb][p] = __builtin_bit_cast(f16x8, src[p * 3 * P + row[s]]);
            af[fb][p] = __builtin_bit_cast(f16x8, wt[(s * 2 + p) * 64 + lane]);
        }
    };
    frags(0, 0);
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int fb = s & 1;
        if (s + 1 < 5) frags(s + 1, fb ^ 1);
        __builtin_amdgcn_sched_barrier(0);   // next step's LDS reads stay above this step's MFMAs
        lo = TVC_MFMA16(af[fb][1], bf[fb][0], lo);
        hi = TVC_MFMA16(af[fb][0], bf[fb][0], hi);
        lo = TVC_MFMA16(af[fb][0], bf[fb][1], lo);
        __builtin_amdgcn_sched_barrier(0);
    }
}

#ifdef U24_TRACE
// diagnostic build only (tools/micro/u24_trace.py): s_memtime stamps of waves 0, 1, 4, 7 of workgroup U24_TRACE over its first tiles, [half][wave slot][64]
static __device__ unsigned long long g_u24_trace[2 * 4 * 64];
#define U24_STAMP(id)                                                                                  \
    do {                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        if (tr_on && tr_n < 60) {                                                                      \
            unsigned long long t_;                                                                     \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
            g_u24_trace[((CF::SECOND ? 1 : 0) * 4 + tr_slot) * 64 + 1 + tr_n++] = (t_ << 8) | (unsigned)(id); \
        }                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    } while (0)
#else
#define U24_STAMP(id) do {} while (0)
#endif

template <class CF, bool RAG>
__global__ __launch_bounds__(CF::NT) __attribute__((amdgpu_waves_per_eu(U24S_WPE))) void up24s_kernel(Up24SArgs a) {
    constexpr int C = CF::C, W = CF::W, D1 = CF::D1, D2 = CF::D2, H = CF::H, E = CF::E, NT = CF::NT;
    constexpr int XP = CF::XP, HP = CF::HP, PS = CF::PS, XW = CF::XW, XPER = CF::XPER;
    extern __shared__ __attribute__((aligned(16))) uint4 smem_u[];
    u32x4* Xs = reinterpret_cast<u32x4*>(smem_u);
    u32x4* Hs = Xs + 6 * XP;
    u32x4* Wt = Hs + 6 * HP;
    float* Fl = reinterpret_cast<float*>(Wt + CF::PIECES * 64);   // ba, bb, bsc, bsh [32 each], w75 [24][7], b75, weight scales [297..300], the bound constants [301], [302]
    float* R = Fl + CF::FL;                                       // [24][PS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
#ifdef U24_TRACE
    const int tr_slot = wave == 0 ? 0 : (wave == 1 ? 1 : (wave == 4 ? 2 : 3));
    const bool tr_on = (int)blockIdx.x == U24_TRACE && lane == 0 && (wave == 0 || wave == 1 || wave == 4 || wave == 7);
    int tr_n = 0;
#endif
    const int rs = a.len;                                  // row stride of cond / x1 (= every utterance's length unless RAG)
    const int rsl = CF::SECOND ? rs : rs / a.xf;           // row stride of the input x
    int bh = 0;                                            // RAG: utterance hint of the table walk
    auto utt = [&](int tile) __attribute__((always_inline)) { return rag_tile<RAG>(a.rag, tile, a.tiles_per_utt, rs, bh).b; };

    // ---- once per workgroup: weights and biases -> LDS -------------------------------------------------
    for (int i = tid; i < CF::PIECES * 64 + 304 / 4; i += NT) Wt[i] = a.img[i];

    // ---- input tile staging: an item = 8 channels of one position ---------------------------------------
    // Per-thread item geometry is tile-invariant; global addresses are a uniform per-channel base (SGPRs) plus one 32-bit
    // lane offset per item, so a load costs no vector address arithmetic.
    float xr0[XPER][8], xr1[XPER][8], lam[XPER];
    int ig8[XPER], ic[XPER];
    bool ilive[XPER];
#pragma unroll
    for (int i = 0; i < XPER; ++i) {
        const int idx = tid + i * NT;
        int g = idx / XW;
        ic[i] = idx - g * XW;
        ilive[i] = idx < CF::ITEMS;      // idle items still load (g = 3 -> 2: a valid address), never store
        ig8[i] = 8 * (g > 2 ? 2 : g);
    }
    auto fetch = [&](int tile) __attribute__((always_inline)) {
        const RagTile rt = rag_tile<RAG>(a.rag, tile, a.tiles_per_utt, rs, bh);
        const int len = rt.len, lin = CF::SECOND ? len : len / a.xf;         // this utterance's extents (output rate / input rate)
        const int px0 = rt.tin * W - E - H;
        const uint4* xb = reinterpret_cast<const uint4*>(RAG ? a.x + 8L * (CF::SECOND ? rt.off : rt.off / a.xf) : a.x + (long)rt.b * C * rsl);      // G8: group g, column p at 32 (g rsl + p) bytes
        auto ld8 = [&](float (&dst)[8], unsigned o) __attribute__((always_inline)) {
            // (element by element: indexing an ext_vector with the induction variable of an unrolled loop is folded wrongly by this compiler -
            // elements 2 and 3 came out dead, their registers were handed to the next load; gemm_s2.h met the same)
            const f32x4s q0 = __builtin_bit_cast(f32x4s, ldg_so4(xb, o)), q1 = __builtin_bit_cast(f32x4s, ldg_so4(xb, o + 16u));
            dst[0] = q0.x; dst[1] = q0.y; dst[2] = q0.z; dst[3] = q0.w;
            dst[4] = q1.x; dst[5] = q1.y; dst[6] = q1.z; dst[7] = q1.w;
        };
#pragma unroll
        for (int i = 0; i < XPER; ++i) {
            int p = px0 + ic[i];
            p = p < 0 ? 0 : (p > len - 1 ? len - 1 : p);
            const int g = ig8[i] >> 3;
            if (CF::SECOND) {
#ifdef X1_PLANAR
                const float* xp = RAG ? a.x + rt.off : a.x + (long)rt.b * C * rsl;
                const unsigned o = 4u * (unsigned)(ig8[i] * rsl + p);
#pragma unroll
                for (int j = 0; j < 8; ++j) xr0[i][j] = ldg_so(xp + (long)j * rsl, o);
#else
                ld8(xr0[i], 32u * (unsigned)(g * rsl + p));
#endif
            } else {
                const Lerp lc = lerp_coord(p, a.interp_scale, lin);
                lam[i] = lc.w1;
                ld8(xr0[i], 32u * (unsigned)(g * rsl + lc.i0));
                ld8(xr1[i], 32u * (unsigned)(g * rsl + lc.i1));
            }
        }
    };
    auto deposit = [&](float xs) __attribute__((always_inline)) {      // xs = the tile's block-floating-point input scale
#pragma unroll
        for (int i = 0; i < XPER; ++i) {
            if (!ilive[i]) continue;
            const int g = ig8[i] >> 3, c = ic[i];
            float v[8];
            if (CF::SECOND) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = xr0[i][j];
            } else {
                const float w0 = 1.f - lam[i];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaf(w0, xr0[i][j], __fmul_rn(lam[i], xr1[i][j]));   // = lerp_eval
            }
            const int rc = c - H;                                  // residual column
            if (rc >= 0 && rc < CF::W2r) {
#pragma unroll
                for (int j = 0; j < 8; ++j) R[(ig8[i] + j) * PS + rc] = v[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.1f * v[j]) * xs;   // = leaky_relu(x, 0.1), scaled
            uint4 p1, p2;
            split8(v, p1, p2);
            Xs[(0 + g) * XP + c] = __builtin_bit_cast(u32x4, p1);
            Xs[(3 + g) * XP + c] = __builtin_bit_cast(u32x4, p2);
        }
    };

    // persistent: a contiguous range of tiles per workgroup (one or two utterances: x1's |max| slot is published once per
    // utterance and workgroup, conv3s.h amax_flush_wg)
    int tile, tend;
    tile_range(a.ntiles, tile, tend);
    if (tile < tend) {
        bh = utt(tile);
        fetch(tile);
        deposit(bfp_load_u(a.amax_x, bh).s);
    }
    slab_barrier();
    const float wsa = Fl[297], wsb = Fl[298], wssc = Fl[299], wssh = Fl[300], wl1 = Fl[301], bamax = Fl[302];
    float mx_run = 0.f;
    int mx_b = bh;
    int slot_b = -1;
    Bfp sx{1.f, 1.f}, sc{1.f, 1.f}, sh_{1.f, 1.f};

    for (; tile < tend; ++tile) {
        U24_STAMP(0);
        const RagTile rt = rag_tile<RAG>(a.rag, tile, a.tiles_per_utt, rs, bh);
        const int b = rt.b, len = rt.len;
        bh = b;
        if (!CF::SECOND && a.amax_y && b != mx_b) {
            amax_flush_wg(a.amax_y + mx_b, mx_run, Fl + 304);
            mx_run = 0.f;
            mx_b = b;
        }
        // block-floating-point scales: input (per-utterance |max| slot), cond, and the on-chip intermediate h = lrelu(conv_a + b_a),
        // bounded by sum|w_a| * amax_x + max|b_a| (never measured: it does not leave the CU)
        if (b != slot_b) {      // the utterance's slots and the scales they give: when the walk enters it (one or two utterances per workgroup), not per tile (600 cycles of 11 000)
            slot_b = b;
            const float slot_x = a.amax_x ? sload_f32(a.amax_x + b) : 0.f;
            const float slot_c = a.amax_c ? sload_f32(a.amax_c + b) : 0.f;
            sx = a.amax_x ? bfp_from_amax(slot_x) : Bfp{1.f, 1.f};
            sc = a.amax_c ? norm_from_amax(fmaf(a.cbw, slot_c, a.cbb)) : Bfp{1.f, 1.f};      // the scale down0s_kernel wrote the planes with
            sh_ = a.amax_x ? bfp_from_amax(fmaf(wl1, slot_x, bamax)) : Bfp{1.f, 1.f};
        }
        const int t0 = rt.tin * W;
        const int ph0 = t0 - E - D2;      // position of Hs column 0
        const int p20 = t0 - E;           // position of second-conv column 0
        const int next = tile + 1;

        // FiLM cond of this wave's second-conv tile: the producer left it split, scaled and in B-fragment order - K16 step 0 = channel
        // group lh, step 1 = group 2 for lh = 0 (the other half is the zero unit): four 16-byte loads, no arithmetic
        // The tile's global stores are issued BEHIND the next tile's deposit: the deposit waits for its fetched registers with vmcnt(0) (its stores
        // sit in exec-masked blocks, the counter cannot be exact), and stores issued before it made every tile wait for their acknowledgement.
        float keep[3][4];
        bool keep_live = false;
        unsigned keep_oo = 0;
        float wv_out = 0.f;
        bool wv_live = false;
        int wv_off = 0;
        u32x4 cq[2][2];
        if (wave < CF::NT2) {
            const uint4* cb = RAG ? a.cond + rt.off : a.cond + (long)b * 6 * rs;
            int t = p20 + wave * 32 + l31;
            t = t < 0 ? 0 : (t > len - 1 ? len - 1 : t);
            const unsigned o0 = 16u * (unsigned)(lh * rs + t), o1 = 16u * (unsigned)t;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                cq[0][p] = ldg_so4(cb + (long)(3 * p) * rs, o0);
                cq[1][p] = ldg_so4(cb + (long)(3 * p + 2) * rs, o1);
            }
        }
        U24_STAMP(1);

        // ---- S1: Hs = split(lrelu(conv_a(lrelu(x)) + ba)) ---------------------------------------------
        // The next tile's input is requested behind the wave's first group of MFMAs: the loads are independent of everything in this
        // tile and land in registers during it, but ISSUING them is 1 200 - 1 600 cycles of the address path (eight waves x eight 16-byte loads
        // behind the previous tile's stores), a phase of its own when it sat in front of S1 with the matrix pipe idle.
        bool fetched = false;
        for (int nt = wave; nt < CF::NT1; nt += CF::NWAVES) {
            f32x16 acc, alo;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = alo[r] = 0.f;
            const int h = nt * 32 + l31;
            conv24_phase<XP, D1>(acc, alo, Xs, Wt, h, 0, XW - 1, lane);      // Xs already holds the replicate-padded input
            if (!fetched) {
                fetched = true;
                if (next < tend) fetch(next);
            }
            const float c = wsa * sx.inv, cl = c * kLoInv;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const f32x4s bv = *reinterpret_cast<const f32x4s*>(Fl + 8 * g + 4 * lh);
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float t = comb(acc[4 * g + q], alo[4 * g + q], c, cl) + bv[q];
                    v[q] = fmaxf(t, 0.1f * t) * sh_.s;
                }
                u32x2 p1, p2;
                split4(v, p1, p2);
                // A position's 16-byte row = [lanes 0-31's four channels | lanes 32-63's four].  Two 8-byte stores per row from the two
                // lane halves are 2-way bank conflicts (16-byte stride inside a 16-lane store group); v_permlane32_swap hands the lower
                // half of the wave both halves of the part-1 row and the upper half both halves of the part-2 row: one 16-byte store
                // each, conflict-free.
                const auto qx = __builtin_amdgcn_permlane32_swap(p1[0], p2[0], false, false);
                const auto qy = __builtin_amdgcn_permlane32_swap(p1[1], p2[1], false, false);
                const u32x4 row = {qx[0], qy[0], qx[1], qy[1]};
                *reinterpret_cast<u32x4*>(Hs + (3 * lh + g) * HP + h) = row;
            }
        }
        if (!fetched && next < tend) fetch(next);
        U24_STAMP(2);
        slab_barrier();
        U24_STAMP(3);

        // ---- S2: (conv_b(Hs) + bb) * scale + shift + res ------------------------------------------------
        if (wave < CF::NT2) {
            const int n = wave * 32 + l31;
            const int t = p20 + n;
            f16x8 cf[2][2];
            {
                const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    cf[0][p] = __builtin_bit_cast(f16x8, cq[0][p]);
                    cf[1][p] = __builtin_bit_cast(f16x8, lh ? z : cq[1][p]);
                }
            }
            const float cb = wsb * sh_.inv, cbl = cb * kLoInv, c1 = wssc * sc.inv, c1l = c1 * kLoInv, c2 = wssh * sc.inv, c2l = c2 * kLoInv;
            float hv[3][4];                                                        // conv_b + b_b of this lane's 12 real rows
            {
                f32x16 acc, alo;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = alo[r] = 0.f;
                const int lo = -ph0 > 0 ? -ph0 : 0;                                // the layer's own replicate padding
                const int hi = (len - 1 - ph0) < (HP - 1) ? (len - 1 - ph0) : (HP - 1);
                conv24_phase<HP, D2>(acc, alo, Hs, Wt + 10 * 64, n, lo, hi, lane);
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const f32x4s bb = *reinterpret_cast<const f32x4s*>(Fl + 32 + 8 * g + 4 * lh);
#pragma unroll
                    for (int q = 0; q < 4; ++q) hv[g][q] = comb(acc[4 * g + q], alo[4 * g + q], cb, cbl) + bb[q];
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // the conv's accumulator pair is dead before FiLM's two pairs come alive
            f32x16 asc, lsc, ash, lsh;
#pragma unroll
            for (int r = 0; r < 16; ++r) asc[r] = lsc[r] = ash[r] = lsh[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f16x8 fa[2][2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int p = 0; p < 2; ++p) fa[mt][p] = __builtin_bit_cast(f16x8, Wt[(20 + (s * 2 + mt) * 2 + p) * 64 + lane]);
                lsc = TVC_MFMA16(fa[0][1], cf[s][0], lsc);
                lsh = TVC_MFMA16(fa[1][1], cf[s][0], lsh);
                asc = TVC_MFMA16(fa[0][0], cf[s][0], asc);
                ash = TVC_MFMA16(fa[1][0], cf[s][0], ash);
                lsc = TVC_MFMA16(fa[0][0], cf[s][1], lsc);
                lsh = TVC_MFMA16(fa[1][0], cf[s][1], lsh);
            }
            float mx = 0.f;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                float v4[4];
                const f32x4s bs = *reinterpret_cast<const f32x4s*>(Fl + 64 + 8 * g + 4 * lh);
                const f32x4s bh = *reinterpret_cast<const f32x4s*>(Fl + 96 + 8 * g + 4 * lh);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = 8 * g + 4 * lh + q;
                    const float hval = hv[g][q];
                    const float scv = comb(asc[4 * g + q], lsc[4 * g + q], c1, c1l) + bs[q];
                    const float shv = comb(ash[4 * g + q], lsh[4 * g + q], c2, c2l) + bh[q];
                    const float res = R[m * PS + n];
                    const float v = __fadd_rn(__fadd_rn(__fmul_rn(hval, scv), shv), res);
                    v4[q] = v;
                    if (CF::SECOND) R[m * PS + n] = v;                         // x2 stays on chip
                }
                if (!CF::SECOND) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) keep[g][q] = v4[q];           // x1 leaves behind the deposit (below)
                    if (n < W && t < len) mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v4[0]), fabsf(v4[1])), fmaxf(fabsf(v4[2]), fabsf(v4[3]))));
                }
            }
            mx_run = fmaxf(mx_run, mx);
            keep_live = !CF::SECOND && n < W && t < len;
            keep_oo = 32u * (unsigned)t + 16u * (unsigned)lh;
        }
        U24_STAMP(4);
        if (CF::SECOND) {
            slab_barrier();
            U24_STAMP(5);
            // ---- S4: c5 and output_layer folded into one Conv1d(24 -> 1, k7, replicate) on the parked x2 tile ----
            // 8 lanes per group of 4 consecutive outputs, 3 channels each: per channel 10 activations and 7
            // (broadcast) weights feed 28 FMAs; the 8 partial sums meet through three shuffles.
            const float* W7 = Fl + 128;
            const int part = tid & 7;
            const int lo = -p20 > 0 ? -p20 : 0;
            const int hi = (len - 1 - p20) < (CF::W2 - 1) ? (len - 1 - p20) : (CF::W2 - 1);
            const bool interior = lo == 0 && hi == CF::W2 - 1;      // no replicate padding inside this tile
            for (int g0 = 0; g0 < (W + 3) / 4; g0 += NT / 8) {      // uniform trip count: the shuffles need every lane
                const int g = g0 + (tid >> 3);
                float o4[4] = {0.f, 0.f, 0.f, 0.f};
                if (4 * g < W) {
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        const int c = part * 3 + cc;
                        float xv[10], wv[7];
                        if (E == 3 && interior) {   // columns 4 g .. 4 g + 11 of the row: three 16-byte reads instead of ten 4-byte ones
                            const f32x4s* rp = reinterpret_cast<const f32x4s*>(R + c * PS + 4 * g);
                            const f32x4s r0 = rp[0], r1 = rp[1], r2 = rp[2];
                            xv[0] = r0[0]; xv[1] = r0[1]; xv[2] = r0[2]; xv[3] = r0[3];
                            xv[4] = r1[0]; xv[5] = r1[1]; xv[6] = r1[2]; xv[7] = r1[3];
                            xv[8] = r2[0]; xv[9] = r2[1];
                        } else {             // a tile at an utterance's end: the layer's replicate padding = a column clamp (rare: not worth keeping the ten columns in registers)
#pragma unroll
                            for (int i = 0; i < 10; ++i) {
                                int col = 4 * g + E - 3 + i;
                                col = col < lo ? lo : (col > hi ? hi : col);
                                xv[i] = R[c * PS + col];
                            }
                        }
#pragma unroll
                        for (int j = 0; j < 7; ++j) wv[j] = W7[c * 7 + j];
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
                static_assert((W + 3) / 4 <= NT / 8, "one output per thread: it waits in a register until the deposit is through");
                wv_out = v + W7[168];
                wv_live = part < 4 && o < W && t0 + o < len;
                wv_off = t0 + o;
            }
        }
        // ---- next tile's input: registers -> LDS ----------------------------------------------------------
        U24_STAMP(6);
        slab_barrier();                                   // every wave is done with Xs, Hs and R
        U24_STAMP(7);
        if (next < tend) {
            const int bn = utt(next);
            deposit(bn == slot_b ? sx.s : bfp_load_u(a.amax_x, bn).s);
        }
        if (CF::SECOND) {
            float* wrow = RAG ? a.out + (long)a.rag.row[b] * a.rag.Tmax * kHop : a.out + (long)b * rs;
            if (wv_live) wrow[wv_off] = wv_out;
        } else if (keep_live) {
            float* ob = RAG ? a.out + 8L * rt.off : a.out + (long)b * C * rs;      // x1 in the G8 layout: this lane's four channels of group g = 16 bytes
#pragma unroll
            for (int g = 0; g < 3; ++g) stg_so4(ob + (long)g * 8 * rs, keep_oo, keep[g]);
        }
        U24_STAMP(8);
        slab_barrier();
    }
#ifdef U24_TRACE
    if (tr_on) g_u24_trace[((CF::SECOND ? 1 : 0) * 4 + tr_slot) * 64] = (unsigned long long)tr_n;
#endif
    if (!CF::SECOND && a.amax_y && tend > (int)((long)a.ntiles * blockIdx.x / gridDim.x)) amax_flush_wg(a.amax_y + mx_b, mx_run, Fl + 304);
}

template <class CF>
static int launch_up24s(tvc_ctx* ctx, hipStream_t s, Up24SArgs a, int B) {
    static int ncu_dev[64] = {};                    // per device of this process (the LDS attribute is per function and device)
    int& ncu = ncu_dev[ctx->device & 63];
    const size_t lds = (size_t)CF::LDS_BYTES;
    if (!ncu) {
        hipDeviceProp_t prop;
        hipError_t e = hipGetDeviceProperties(&prop, ctx->device);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)up24s_kernel<CF, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)up24s_kernel<CF, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(ctx, TVC_ERR_HIP, "up24s setup: %s", hipGetErrorString(e));
        ncu = prop.multiProcessorCount;
    }
    a.tiles_per_utt = (a.len + CF::W - 1) / CF::W;
    a.ntiles = a.tiles_per_utt * B;
    static_assert(CF::WGS_PER_CU * CF::LDS_BYTES <= 160 * 1024, "LDS for the workgroups that share a CU");
    const int slots = ncu * CF::WGS_PER_CU;
    if (ctx->rag) {
        if (B != 1 || a.len != ctx->rag->Ttot * kHop) return fail(ctx, TVC_ERR_STATE, "up24s: a ragged batch runs as one long utterance");
        TVC_CHECK(rag_view(ctx, s, kHop, CF::W, &a.rag, &a.ntiles));
    }
    int grid = a.ntiles < slots ? a.ntiles : slots;
    if (ctx->rag) hipLaunchKernelGGL((up24s_kernel<CF, true>), dim3(grid), dim3(CF::NT), lds, s, a);
    else hipLaunchKernelGGL((up24s_kernel<CF, false>), dim3(grid), dim3(CF::NT), lds, s, a);
    return launch_check(ctx, "up24s");
}

#ifdef U24_TRACE
}  // namespace tvc
extern "C" int tvc_debug_trace_u24(unsigned long long* host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(tvc::g_u24_trace), sizeof(tvc::g_u24_trace)) == hipSuccess ? 0 : -1;
}
namespace tvc {
#endif

constexpr int U24S_WA = 250, U24S_WB = 250;     // output samples per tile of the two halves (196 / 218 for the second half - one round of first-conv tiles instead of two - measured 8 % slower: the halo dominates)

// Upsample block with cin == 24 followed by FilterNet.output_layer:
// x [B][3][len/f][8] (G8 layout, see Up24SArgs), cond planes -> wave [B][len]; x1 is scratch of B * 24 * len floats (G8 layout).
// amax_x / amax_c: per-utterance |max| slots of x / cond; amax_x1: scratch slot [B] (zeroed) for the block's intermediate x1.
int run_up24_split(tvc_ctx* ctx, hipStream_t s, const UpW& u, const float* x, const float* cond_planes, float cbw, float cbb, float* x1, float* wave, int B, int len,
                   const float* amax_x, const float* amax_c, float* amax_x1) {
    if (!u.s24a || !u.s24b) return fail(ctx, TVC_ERR_STATE, "up24s: the split weight blobs of the 24-channel block are missing");
    if ((long)len * 24 * 4 >= (1L << 32)) return fail(ctx, TVC_ERR_ARG, "up24s: utterance too long for 32-bit element offsets");
    using CA = U24S<U24S_WA, 1, 3, false, 0>;       // (U24S<122, 1, 3, false, 0, 4>: two independent 4-wave workgroups of 122-sample tiles per CU, was measured at 602 vs 590 us)
    using CB = U24S<U24S_WB, 9, 27, true, 3>;
    Up24SArgs a{};
    a.len = len;
    a.xf = u.factor;
    a.interp_scale = (float)(1.0 / (double)u.factor);
    a.cond = reinterpret_cast<const uint4*>(cond_planes);
    a.cbw = cbw;
    a.cbb = cbb;
    a.amax_c = amax_c;
    a.x = x;
    a.amax_x = amax_x;
    a.out = x1;
    a.amax_y = amax_x1;
    a.img = reinterpret_cast<const u32x4*>(u.s24a);
    TVC_CHECK(launch_up24s<CA>(ctx, s, a, B));
    a.x = x1;
    a.amax_x = amax_x1;
    a.out = wave;
    a.amax_y = nullptr;
    a.img = reinterpret_cast<const u32x4*>(u.s24b);
    return launch_up24s<CB>(ctx, s, a, B);
}

// ---------------------------------------------------------------------------------------------------------------------
// FilterNet.downs[0] (decoder.py:206,224,227): Conv1d(17 -> 24, k3, replicate) over cat[source (16 ch), energy (1 ch)] at
// the full sample rate, on the same split-precision machinery: the 17 channels are three 8-channel groups (rows 17..23
// zero), 9 (tap, group) units = 5 K16 steps, 15 MFMAs per 32 samples.  HBM-bound (reads 17 rows, writes 24 + the
// 1/5-rate copy Downsample 1 starts from): persistent workgroups, two LDS input tiles, the input of tile i + 2 in flight in
// registers while tile i multiplies, one barrier per tile.
struct Down0SArgs {
    const float* source;   // [B][16][L]
    const float* energy;   // [B][1][L]
    float* out;            // optional (parity taps) [B][24][L] fp32
    uint4* planes;         // the output as the FiLM 1x1s' ready B operand (up24s_kernel): two fp16 planes [B][part][3 groups][L][8 fp16] of out * 2^k,
                           // k from the analytic bound Bi[29] |x|max + Bi[30] >= |out| (amax_x non-null)
    float* y2;             // optional, G8 layout [B][3][L / 5][8]: F.interpolate(out, scale_factor = 1/5) = the sample at 5 d + 2
    const u32x4* img;      // 10 weight pieces + 32 floats: bias, [31] = the image's scale (api.hip down0s)
    int len, tiles_per_utt, ntiles;
    const float* amax_x;   // per-utterance |max| of cat[source, energy] (block-floating-point guard, conv3s.h), nullable
    float* amax_y;         // ... of the output (written), nullable
    RagDev rag;            // RAG kernels (ragged.h): `len` = row stride of the batch-wide tensors, tiles / extents from the table
};

template <bool RAG>
// (two 60 KB workgroups per CU, four waves per SIMD at <= 128 registers: this kernel waits on HBM - 1.1 GB in and out per step - more than on its
// 15 MFMAs per tile, and the second workgroup's loads fly under the first one's arithmetic: 0.371 -> 0.339 ms same-box for the input stage, round 4)
static __global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4))) void down0s_kernel(Down0SArgs a) {
    constexpr int W = 254, XW = 256, XP = XW, NT = 512;
    extern __shared__ __attribute__((aligned(16))) uint4 smem_d[];
    u32x4* Xs = reinterpret_cast<u32x4*>(smem_d);              // [2 buffers][2 parts][3 groups][XP]
    u32x4* Wt = Xs + 2 * 6 * XP;                               // 10 pieces
    float* Bi = reinterpret_cast<float*>(Wt + 10 * 64);        // bias [32] ([31] = weight scale), [32..39] = |max| exchange
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int rs = a.len;                // row stride (= every utterance's length unless RAG)
    for (int i = tid; i < 10 * 64 + 8; i += NT) Wt[i] = a.img[i];

    // staging: thread -> (group tid >> 8 of the 16 source rows, column tid & 255); threads 0..255 also carry the energy row
    const int g = tid >> 8, c = tid & 255;
    float xa[8], xe = 0.f;
    int bh = 0;                          // RAG: utterance hint of the table walk
    auto fetch = [&](int tile) __attribute__((always_inline)) {
        const RagTile rt = rag_tile<RAG>(a.rag, tile, a.tiles_per_utt, rs, bh);
        const int b = rt.b, len = rt.len;
        int p = rt.tin * W - 1 + c;
        p = p < 0 ? 0 : (p > len - 1 ? len - 1 : p);
        const float* sb = RAG ? a.source + rt.off : a.source + (long)b * 16 * rs;
        const unsigned o = 4u * (unsigned)(8 * g * rs + p);
#pragma unroll
        for (int j = 0; j < 8; ++j) xa[j] = ldg_so(sb + (long)j * rs, o);
        xe = ldg_so(RAG ? a.energy + rt.off : a.energy + (long)b * rs, 4u * (unsigned)p);
    };
    auto utt = [&](int tile) __attribute__((always_inline)) { return rag_tile<RAG>(a.rag, tile, a.tiles_per_utt, rs, bh).b; };
    auto deposit = [&](int buf, float xs) __attribute__((always_inline)) {
        u32x4* X = Xs + buf * 6 * XP;
#pragma unroll
        for (int j = 0; j < 8; ++j) xa[j] *= xs;
        uint4 p1, p2;
        split8(xa, p1, p2);
        X[(0 + g) * XP + c] = __builtin_bit_cast(u32x4, p1);
        X[(3 + g) * XP + c] = __builtin_bit_cast(u32x4, p2);
        if (tid < 256) {                                       // group 2 = [energy, 0 x 7]
            const float ve[8] = {xe * xs, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            split8(ve, p1, p2);
            X[(0 + 2) * XP + c] = __builtin_bit_cast(u32x4, p1);
            X[(3 + 2) * XP + c] = __builtin_bit_cast(u32x4, p2);
        }
    };

    int tile, tend, cur = 0;                             // a contiguous range of tiles per workgroup (amax_flush_wg, conv3s.h)
    tile_range(a.ntiles, tile, tend);
    if (tile >= tend) return;
    bh = utt(tile);
    fetch(tile);
    deposit(0, bfp_load_u(a.amax_x, bh).s);
    if (tile + 1 < tend) fetch(tile + 1);
    slab_barrier();
    const int len2 = rs / 5;             // row stride of the 1/5-rate copy
    const float cw = Bi[31];
    float mx_run = 0.f;
    int mx_b = bh;
    for (; tile < tend; ++tile, cur ^= 1) {
        const RagTile rt = rag_tile<RAG>(a.rag, tile, a.tiles_per_utt, rs, bh);
        const int b = rt.b, len = rt.len;
        bh = b;
        if (a.amax_y && b != mx_b) {
            amax_flush_wg(a.amax_y + mx_b, mx_run, Bi + 32);
            mx_run = 0.f;
            mx_b = b;
        }
        const int t0 = rt.tin * W;
        const int next = tile + 1, next2 = next + 1;
        if (next < tend) deposit(cur ^ 1, bfp_load_u(a.amax_x, utt(next)).s);            // tile i + 1 (requested one tile ago) -> the other buffer
        if (next2 < tend) fetch(next2);               // tile i + 2 flies across this tile
        f32x16 acc, alo;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = alo[r] = 0.f;
        const int n = wave * 32 + l31;
        conv24_phase<XP, 1>(acc, alo, Xs + cur * 6 * XP, Wt, n, 0, XW - 1, lane);
        // the planes' scale 2^k (up24s_kernel undoes it) rides in the epilogue's constants: (acc c + lo cl + bias) 2^k is formed as
        // acc (c 2^k) + lo (cl 2^k) + bias 2^k - powers of two, the same bits - and the fp32 values the tap / the 1/5-rate copy / the |max| want are vs 2^-k
        const Bfp pn = a.amax_x ? norm_from_amax(fmaf(Bi[29], sload_f32(a.amax_x + b), Bi[30])) : Bfp{1.f, 1.f};
        const float ps = pn.s, pinv = pn.inv;
        const float cc = cw * bfp_load_u(a.amax_x, b).inv * ps, ccl = cc * kLoInv;
        const int t = t0 + n;
        float mx = 0.f;
        {
            const bool live = n < W && t < len;
            const int tc = t < len ? t : len - 1;
            float* ob = a.out ? (RAG ? a.out + rt.off : a.out + (long)b * 24 * rs) : nullptr;
            uint4* pb = RAG ? a.planes + rt.off : a.planes + (long)b * 6 * rs;
            float* y2b = a.y2 ? (RAG ? a.y2 + 8L * (rt.off / 5) : a.y2 + (long)b * 24 * len2) : nullptr;      // G8 layout [3][len2][8] (down24f_kernel's input)
            const unsigned oo = 4u * (unsigned)(4 * lh * rs + tc);
            const int q5 = tc / 5;
            const bool pick = live && a.y2 != nullptr && tc - 5 * q5 == 2;
#pragma unroll
            for (int gg = 0; gg < 3; ++gg) {
                const f32x4s bv = *reinterpret_cast<const f32x4s*>(Bi + 8 * gg + 4 * lh);
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[q] = comb(acc[4 * gg + q], alo[4 * gg + q], cc, ccl) + bv[q] * ps;
                    if (live) {
                        if (ob) stg_so(ob + (long)(8 * gg + q) * rs, oo, v[q] * pinv);
                        mx = fmaxf(mx, fabsf(v[q]));
                    }
                }
                if (pick) {
                    const float p4[4] = {v[0] * pinv, v[1] * pinv, v[2] * pinv, v[3] * pinv};
                    stg_so4(y2b + (long)gg * 8 * len2, 32u * (unsigned)q5 + 16u * (unsigned)lh, p4);
                }
                // a position's 16-byte operand row = [lanes 0-31's four channels | lanes 32-63's four]: v_permlane32_swap hands the lower half of
                // the wave both halves of the part-1 row and the upper half those of part 2 (every lane takes part: no branch around it)
                u32x2 p1, p2;
                split4(v, p1, p2);
                const auto qx = __builtin_amdgcn_permlane32_swap(p1[0], p2[0], false, false);
                const auto qy = __builtin_amdgcn_permlane32_swap(p1[1], p2[1], false, false);
                if (live) pb[(long)(3 * lh + gg) * rs + tc] = make_uint4(qx[0], qy[0], qx[1], qy[1]);
            }
        }
        mx_run = fmaxf(mx_run, mx * pinv);
        slab_barrier();
    }
    if (a.amax_y) amax_flush_wg(a.amax_y + mx_b, mx_run, Bi + 32);
}

int run_down0_split(tvc_ctx* ctx, hipStream_t s, const float* blob, const float* source, const float* energy, float* planes, float* out_fp32, float* y2, int B, int len,
                    const float* amax_x, float* amax_y) {
    if (!blob) return fail(ctx, TVC_ERR_STATE, "down0s: the split weight blob of downs.0 is missing");
    if ((long)len * 24 * 4 >= (1L << 32)) return fail(ctx, TVC_ERR_ARG, "down0s: utterance too long for 32-bit byte offsets");
    if (y2 && len % 5 != 0) return fail(ctx, TVC_ERR_ARG, "down0s: the 1/5-rate copy needs len % 5 == 0");
    static int ncu_dev[64] = {};
    int& ncu = ncu_dev[ctx->device & 63];
    constexpr size_t lds = (2 * 6 * 256 + 10 * 64 + 8 + 2) * 16;
    if (!ncu) {
        hipDeviceProp_t prop;
        hipError_t e = hipGetDeviceProperties(&prop, ctx->device);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)down0s_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)down0s_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(ctx, TVC_ERR_HIP, "down0s setup: %s", hipGetErrorString(e));
        ncu = prop.multiProcessorCount;
    }
    Down0SArgs a{source, energy, out_fp32, reinterpret_cast<uint4*>(planes), y2, reinterpret_cast<const u32x4*>(blob), len, (len + 253) / 254, 0, amax_x, amax_y, RagDev{}};
    a.ntiles = a.tiles_per_utt * B;
    if (ctx->rag) {
        if (B != 1 || len != ctx->rag->Ttot * kHop) return fail(ctx, TVC_ERR_STATE, "down0s: a ragged batch runs as one long utterance");
        TVC_CHECK(rag_view(ctx, s, kHop, 254, &a.rag, &a.ntiles));
    }
    const int grid = a.ntiles < 2 * ncu ? a.ntiles : 2 * ncu;      // two persistent workgroups per CU
    if (ctx->rag) hipLaunchKernelGGL(down0s_kernel<true>, dim3(grid), dim3(512), lds, s, a);
    else hipLaunchKernelGGL(down0s_kernel<false>, dim3(grid), dim3(512), lds, s, a);
    return launch_check(ctx, "down0s");
}

// ---------------------------------------------------------------------------------------------------------------------
// The whole 24-channel Downsample block in ONE kernel: xi -> c1 (dil 1) -> lrelu -> c2 (dil 2) -> lrelu -> c3 (dil 4, 24 -> 48) + down_res(xi)
// (decoder.py:143-158).  As three launches (one conv each, until round 3) the block was almost pure staging - without their MFMAs they took
// 75 of 85, 66 of 74 and 121 of 164 us (profiles/r03_whatif.txt) - and h1 / h2 crossed HBM between them (4 x 118 MB per step).  Here they stay on chip as the next conv's
// split operand tiles, like the intermediate of the fused ups.4 halves: a tile is 244 output samples; c1 computes h1 on the 256 columns
// c2's and c3's halos need (7 samples of xi beyond the tile on either side), c2 h2 on 252, c3 the 244 - one 32-column n-tile per wave and
// conv, two barriers per tile, the next tile's input split into the second input buffer and the one after it in flight in registers
// meanwhile.  Same K order, same part products, same epilogue arithmetic as the three launches had: for inputs inside fp16's window the
// result is bit-identical to theirs (measured before they were removed).  The intermediates have no |max| slot (they never leave the CU): h1 is scaled by the
// analytic bound (max_m sum|w1|) |xi|max + max|b1|, h2 by the bound of that bound, c3 and down_res share one scale as they share accumulators.
struct Down24FArgs {
    const float* x;        // xi, G8 layout [B][3][len][8] (down0s_kernel's 1/5-rate copy)
    float* out;            // [B][6][len][8]: the block's output in the G8 layout (conv48s.hip: 16-byte stores here, 16-byte fragment loads there)
    float* y2;             // optional [B][48][len / 4]: mean of samples 4 d + 1, 4 d + 2 (the next block's 1/4-rate input)
    const u32x4* img1;     // the three convs' blobs (api.hip Packer::conv24s): c1 (10 pieces + 64 floats), c2, c3 (20 pieces, bias = c3 + down_res, joint scales)
    const u32x4* img2;
    const u32x4* img3;
    const u32x4* rimg;     // down_res image (8 pieces)
    int len, tiles_per_utt, ntiles;
    float b1_w, b1_b, b2_w, b2_b;      // |h1| <= b1_w |xi|max + b1_b, |h2| <= b2_w |h1|bound + b2_b
    const float* amax_x;   // per-utterance |max| of xi (read, nullable), of out (written, nullable)
    float* amax_y;
    RagDev rag;            // RAG kernels (ragged.h): `len` = row stride of the batch-wide tensors, tiles / extents from the table
};
struct D24F {
    static constexpr int W = 244, XW = W + 14, XP = 264, HP = 256, NT = 512;
    static constexpr int LDS_BYTES = (2 * 6 * XP + 2 * 6 * HP + 48 * 64) * 16 + (3 * 64 + 8) * 4;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

template <bool RAG>
static __global__ __launch_bounds__(D24F::NT) __attribute__((amdgpu_waves_per_eu(2))) void down24f_kernel(Down24FArgs a) {
    constexpr int W = D24F::W, XW = D24F::XW, XP = D24F::XP, HP = D24F::HP, NT = D24F::NT;
    extern __shared__ __attribute__((aligned(16))) uint4 smem_df[];
    u32x4* Xs = reinterpret_cast<u32x4*>(smem_df);             // [2 buffers][2 parts][3 groups][XP]: lrelu(xi), position t0 - 7 + c
    u32x4* H1 = Xs + 2 * 6 * XP;                               // [2 parts][3 groups][HP]: lrelu(h1), position t0 - 6 + c
    u32x4* H2 = H1 + 6 * HP;                                   // lrelu(h2), position t0 - 4 + c
    u32x4* W1 = H2 + 6 * HP;                                   // 10 + 10 + 20 + 8 pieces
    u32x4* W2 = W1 + 10 * 64;
    u32x4* W3 = W2 + 10 * 64;
    u32x4* Wr = W3 + 20 * 64;
    float* Bi = reinterpret_cast<float*>(Wr + 8 * 64);         // the three blobs' 64 floats (bias, [62 + mt] = scales), then the |max| exchange
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int rs = a.len;                // row stride (= every utterance's length unless RAG)
    int bh = 0;                          // RAG: utterance hint of the table walk
    auto utt = [&](int tile) __attribute__((always_inline)) { return rag_tile<RAG>(a.rag, tile, a.tiles_per_utt, rs, bh).b; };
    for (int i = tid; i < 10 * 64; i += NT) {
        W1[i] = a.img1[i];
        W2[i] = a.img2[i];
    }
    for (int i = tid; i < 20 * 64; i += NT) W3[i] = a.img3[i];
    for (int i = tid; i < 8 * 64; i += NT) Wr[i] = a.rimg[i];
    if (tid < 64) {
        Bi[tid] = reinterpret_cast<const float*>(a.img1 + 10 * 64)[tid];
        Bi[64 + tid] = reinterpret_cast<const float*>(a.img2 + 10 * 64)[tid];
        Bi[128 + tid] = reinterpret_cast<const float*>(a.img3 + 20 * 64)[tid];
    }

    // staging items (8-channel group, column): 3 * XW = 774, two per thread
    float xa[2][8];
    auto fetch = [&](int tile) __attribute__((always_inline)) {
        const RagTile rt = rag_tile<RAG>(a.rag, tile, a.tiles_per_utt, rs, bh);
        const int len = rt.len;
        const int px0 = rt.tin * W - 7;
        const float* xb = RAG ? a.x + 8L * rt.off : a.x + (long)rt.b * 24 * rs;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * NT;
            int g = idx / XW;
            const int c = idx - g * XW;
            g = g > 2 ? 2 : g;                                 // idle items load a valid address
            int p = px0 + c;
            p = p < 0 ? 0 : (p > len - 1 ? len - 1 : p);
            ld8_g8(xa[i], xb, 32u * (unsigned)(g * rs + p));
        }
    };
    auto deposit = [&](int buf, float xs) __attribute__((always_inline)) {
        u32x4* X = Xs + buf * 6 * XP;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * NT;
            if (idx >= 3 * XW) continue;
            const int g = idx / XW, c = idx - g * XW;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(xa[i][j], 0.1f * xa[i][j]) * xs;
            uint4 p1, p2;
            split8(v, p1, p2);
            X[(0 + g) * XP + c] = __builtin_bit_cast(u32x4, p1);
            X[(3 + g) * XP + c] = __builtin_bit_cast(u32x4, p2);
        }
    };
    // block-floating-point scales of an utterance: xi (its slot), h1 and h2 (analytic bounds), and the one c3 and down_res share
    struct Sc {
        Bfp x, h1, hj;
    };
    auto scales = [&](int b) __attribute__((always_inline)) -> Sc {
        Sc r;
        r.x = bfp_load_u(a.amax_x, b);
        if (a.amax_x) {
            const float bound1 = fmaf(a.b1_w, sload_f32(a.amax_x + b), a.b1_b);
            r.h1 = bfp_from_amax(bound1);
            r.hj = bfp_min(bfp_from_amax(fmaf(a.b2_w, bound1, a.b2_b)), r.x);
        } else {
            r.h1 = r.hj = Bfp{1.f, 1.f};
        }
        return r;
    };
    // a 24-channel conv's epilogue into the next conv's operand tile: lrelu(acc + bias) * s, split, one 16-byte row per lane (up24s)
    auto to_tile = [&](const f32x16& acc, const f32x16& alo, float c, const float* bias, float s, u32x4* H, int col) __attribute__((always_inline)) {
        const float cl = c * kLoInv;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const f32x4s bv = *reinterpret_cast<const f32x4s*>(bias + 8 * g + 4 * lh);
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float t = comb(acc[4 * g + q], alo[4 * g + q], c, cl) + bv[q];
                v[q] = fmaxf(t, 0.1f * t) * s;
            }
            u32x2 p1, p2;
            split4(v, p1, p2);
            const auto qx = __builtin_amdgcn_permlane32_swap(p1[0], p2[0], false, false);
            const auto qy = __builtin_amdgcn_permlane32_swap(p1[1], p2[1], false, false);
            const u32x4 row = {qx[0], qy[0], qx[1], qy[1]};
            *reinterpret_cast<u32x4*>(H + (3 * lh + g) * HP + col) = row;
        }
    };

    int tile, tend, cur = 0;
    tile_range(a.ntiles, tile, tend);
    if (tile >= tend) return;
    bh = utt(tile);
    fetch(tile);
    deposit(0, bfp_load_u(a.amax_x, bh).s);
    if (tile + 1 < tend) fetch(tile + 1);
    slab_barrier();
    const int len2 = rs >> 2;            // row stride of the 1/4-rate copy
    float mx_run = 0.f;
    int mx_b = bh;
    int sc_b = -1;
    Sc sc{};
    for (; tile < tend; ++tile, cur ^= 1) {
        const RagTile rt = rag_tile<RAG>(a.rag, tile, a.tiles_per_utt, rs, bh);
        const int b = rt.b, len = rt.len;
        bh = b;
        if (a.amax_y && b != mx_b) {
            amax_flush_wg(a.amax_y + mx_b, mx_run, Bi + 192);
            mx_run = 0.f;
            mx_b = b;
        }
        const int t0 = rt.tin * W;
        const int next = tile + 1, next2 = next + 1;
        if (b != sc_b) {      // the utterance's scales: when the walk enters it, not per tile
            sc_b = b;
            sc = scales(b);
        }
        if (next < tend) {      // tile i + 1 (requested one tile ago) -> the other buffer
            const int bn = utt(next);
            deposit(cur ^ 1, bn == b ? sc.x.s : bfp_load_u(a.amax_x, bn).s);
        }
        const int n = wave * 32 + l31;                         // this lane's column of every conv's tile
        const int t = t0 + n;                                  // ... = the output position of c3's
        const bool live = n < W && t < len;
        const int tc = t < len ? t : len - 1;
        // down_res's B fragments (xi at the output position, raw), requested before the first multiply
        float xq0[8], xq1[8];
        {
            const float* xb2 = RAG ? a.x + 8L * rt.off : a.x + (long)b * 24 * rs;
            ld8_g8(xq0, xb2, 32u * (unsigned)(lh * rs + tc));          // K16 step 0: channels 8 lh + j = group lh
            ld8_g8(xq1, xb2, 32u * (unsigned)(2 * rs + tc));           // step 1: channels 16 + j = group 2 on lh = 0, the zero unit on lh = 1
        }
        // ---- c1: Xs -> H1 (position t0 - 6 + n needs xi at t0 - 7 + n + tap) ---------------------------------------------------
        {
            f32x16 acc, alo;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = alo[r] = 0.f;
            conv24_phase<XP, 1>(acc, alo, Xs + cur * 6 * XP, W1, n, 0, XW - 1, lane);       // Xs holds the replicate-padded input
            fetch(next2 < tend ? next2 : tile);      // tile i + 2 flies across this tile; requested behind c1's MFMAs (up24s_kernel has the note), the last tiles re-read themselves
            to_tile(acc, alo, Bi[62] * sc.x.inv, Bi, sc.h1.s, H1, n);
        }
        slab_barrier();
        // ---- c2: H1 -> H2 (position t0 - 4 + n needs h1 at positions ... + 2 (tap - 1) = H1 columns n + 2 tap) -------------------
        {
            f32x16 acc, alo;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = alo[r] = 0.f;
            const int lo = 6 - t0 > 0 ? 6 - t0 : 0;                                  // the layer's own replicate padding: h1 exists on [0, len)
            const int hi = len - 1 - t0 + 6 < HP - 1 ? len - 1 - t0 + 6 : HP - 1;
            conv24_phase<HP, 2>(acc, alo, H1, W2, n, lo, hi, lane);
            to_tile(acc, alo, Bi[64 + 62] * sc.h1.inv, Bi + 64, sc.hj.s, H2, n);
        }
        slab_barrier();
        // ---- c3 + down_res: H2 (columns n + 4 tap) and xi -> out, two m-tiles sharing every B fragment ---------------------------
        f32x16 acc[2], alo[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = alo[mt][r] = 0.f;
        {
            const int lo = 4 - t0 > 0 ? 4 - t0 : 0;
            const int hi = len - 1 - t0 + 4 < HP - 1 ? len - 1 - t0 + 4 : HP - 1;
            constexpr int TAP0[5] = {0, 0, 1, 2, 2}, GRP0[5] = {0, 2, 1, 0, 2};
            constexpr int TAP1[5] = {0, 1, 1, 2, 2}, GRP1[5] = {1, 0, 2, 1, 2};
            f16x8 af[2][2][2], bf[2][2];
            auto frags = [&](int s, int fb) __attribute__((always_inline)) {
                int c = n + (lh ? TAP1[s] : TAP0[s]) * 4;
                c = c < lo ? lo : (c > hi ? hi : c);
                const int row = (lh ? GRP1[s] : GRP0[s]) * HP + c;
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    bf[fb][p] = __builtin_bit_cast(f16x8, H2[p * 3 * HP + row]);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) af[fb][mt][p] = __builtin_bit_cast(f16x8, W3[((s * 2 + mt) * 2 + p) * 64 + lane]);
                }
            };
            frags(0, 0);
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                const int fb = s & 1;
                if (s + 1 < 5) frags(s + 1, fb ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) alo[mt] = TVC_MFMA16(af[fb][mt][1], bf[fb][0], alo[mt]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[mt] = TVC_MFMA16(af[fb][mt][0], bf[fb][0], acc[mt]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) alo[mt] = TVC_MFMA16(af[fb][mt][0], bf[fb][1], alo[mt]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {       // + down_res(xi): two more K16 steps on the same accumulators
            float xq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) xq[j] = (st ? xq1[j] : xq0[j]) * sc.hj.s;
            uint4 p1, p2;
            split8(xq, p1, p2);
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);
            const bool dead = st == 1 && lh;
            const f16x8 xf[2] = {__builtin_bit_cast(f16x8, dead ? z : p1), __builtin_bit_cast(f16x8, dead ? z : p2)};
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                f16x8 wf[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) wf[p] = __builtin_bit_cast(f16x8, Wr[((st * 2 + mt) * 2 + p) * 64 + lane]);
                alo[mt] = TVC_MFMA16(wf[1], xf[0], alo[mt]);
                acc[mt] = TVC_MFMA16(wf[0], xf[0], acc[mt]);
                alo[mt] = TVC_MFMA16(wf[0], xf[1], alo[mt]);
            }
        }
        float mx = 0.f;
        {
            float* ob = RAG ? a.out + 8L * rt.off : a.out + (long)b * 48 * rs;      // skips[1] in the G8 layout [6 groups][len][8] (conv48s.hip reads it as FiLM cond)
            float* y2b = a.y2 ? (RAG ? a.y2 + (rt.off >> 2) : a.y2 + (long)b * 48 * len2) : nullptr;
            const unsigned oo = 32u * (unsigned)tc + 16u * (unsigned)lh;
            const bool pair = a.y2 != nullptr && (t & 3) == 1 && t + 1 < len;      // 1/4-rate copy: mean of samples 4 d + 1, 4 d + 2
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float c = Bi[128 + 62 + mt] * sc.hj.inv, cl = c * kLoInv;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (32 * mt + 8 * g >= 48) continue;
                    const f32x4s bv = *reinterpret_cast<const f32x4s*>(Bi + 128 + 32 * mt + 8 * g + 4 * lh);
                    float v4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float v = comb(acc[mt][4 * g + q], alo[mt][4 * g + q], c, cl) + bv[q];
                        const float vn = __shfl_down(v, 1);                        // sample t + 1 (same tile: W % 4 == 0)
                        const int m = 32 * mt + 8 * g + 4 * lh + q;
                        v4[q] = v;
                        if (live) {
                            mx = fmaxf(mx, fabsf(v));
                            if (pair) y2b[(long)m * len2 + (t >> 2)] = fmaf(0.5f, v, __fmul_rn(0.5f, vn));
                        }
                    }
                    if (live) stg_so4(ob + (long)(4 * mt + g) * 8 * rs, oo, v4);
                }
            }
        }
        mx_run = fmaxf(mx_run, mx);
        // (no barrier here: the next tile's c1 writes H1, which nobody reads any more; its first barrier comes before anybody rewrites H2)
    }
    if (a.amax_y) amax_flush_wg(a.amax_y + mx_b, mx_run, Bi + 192);
}

int run_down24_fused(tvc_ctx* ctx, hipStream_t s, const DownW& d, const float* xi, float* out, float* y2, int B, int len, const float* amax_xi, float* amax_out) {
    if (!d.s24c1 || !d.s24c2 || !d.s24c3r || !d.res.A6 || d.res.MT6 != 2 || d.res.wjoint != d.c3.A6 || !(d.b1_w > 0.f))
        return fail(ctx, TVC_ERR_STATE, "down24f: the split weight blobs of the 24-channel Downsample block are missing");
    if (d.cin != 24 || d.cout != 48) return fail(ctx, TVC_ERR_ARG, "down24f: 24 -> 48 channels only");
    if ((long)len * 48 * 4 >= (1L << 32)) return fail(ctx, TVC_ERR_ARG, "down24f: utterance too long for 32-bit byte offsets");
    if (y2 && len % 4 != 0) return fail(ctx, TVC_ERR_ARG, "down24f: the 1/4-rate copy needs len % 4 == 0");
    static int ncu_dev[64] = {};
    int& ncu = ncu_dev[ctx->device & 63];
    if (!ncu) {
        hipDeviceProp_t prop;
        hipError_t e = hipGetDeviceProperties(&prop, ctx->device);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)down24f_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, D24F::LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)down24f_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, D24F::LDS_BYTES);
        if (e != hipSuccess) return fail(ctx, TVC_ERR_HIP, "down24f setup: %s", hipGetErrorString(e));
        ncu = prop.multiProcessorCount;
    }
    Down24FArgs a{};
    a.x = xi;
    a.out = out;
    a.y2 = y2;
    a.img1 = reinterpret_cast<const u32x4*>(d.s24c1);
    a.img2 = reinterpret_cast<const u32x4*>(d.s24c2);
    a.img3 = reinterpret_cast<const u32x4*>(d.s24c3r);
    a.rimg = reinterpret_cast<const u32x4*>(d.res.A6);
    a.len = len;
    a.tiles_per_utt = (len + D24F::W - 1) / D24F::W;
    a.ntiles = a.tiles_per_utt * B;
    a.b1_w = d.b1_w;
    a.b1_b = d.b1_b;
    a.b2_w = d.b2_w;
    a.b2_b = d.b2_b;
    a.amax_x = amax_xi;
    a.amax_y = amax_out;
    if (ctx->rag) {
        if (B != 1 || len != ctx->rag->Ttot * (kHop / 5)) return fail(ctx, TVC_ERR_STATE, "down24f: a ragged batch runs as one long utterance");
        TVC_CHECK(rag_view(ctx, s, kHop / 5, D24F::W, &a.rag, &a.ntiles));
    }
    const int grid = a.ntiles < ncu ? a.ntiles : ncu;
    if (ctx->rag) hipLaunchKernelGGL(down24f_kernel<true>, dim3(grid), dim3(D24F::NT), D24F::LDS_BYTES, s, a);
    else hipLaunchKernelGGL(down24f_kernel<false>, dim3(grid), dim3(D24F::NT), D24F::LDS_BYTES, s, a);
    return launch_check(ctx, "down24f");
}

}  // namespace tvc
