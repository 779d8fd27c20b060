// Conv1d (k = 1 or 3, dilated, replicate padding) as an implicit GEMM on the fp16 matrix pipe with
// fp32-equivalent accuracy.  Every fp32 operand is split into TWO fp16 parts, x = h1 + 2^-11 h2 with
// h1 = fp16(x), h2 = fp16((x - h1) * 2^11) (the residual is exact in fp32; the 2^11 keeps it out of fp16's subnormal range),
// 22 significand bits in all, and a product is accumulated in fp32 from THREE part-products: h1 w1 into one accumulator,
// h1 w2 + h2 w1 (the 2^-11-order terms, in units of 2^-11) into a second one; out = acc_hi + 2^-11 acc_lo.  The dropped
// h2 w2 term is <= 2^-24 relative.  Measured against fp64 on the part (tools/micro/f16split.hip, K = 768): 1.9e-7 rel rms,
// vs 4.9e-7 for the fp32 MFMA and 4.2e-7 for the bf16 x 3 / six-product split this replaces (which spent twice the
// matrix-pipe cycles, 1.5x the LDS bytes and 1.5x the split arithmetic).  v_mfma_f32_32x32x16_f16 keeps fp16 subnormals
// (measured), so the absolute error floor of an operand is 2^-36 of its scale unit.
//
// Range guard (block floating point).  fp16 tops out at 65504, so every operand travels with a power-of-two scale:
//   weights      normalised per 32-row m-tile at pack time (largest |w| of the tile in [1, 2)), the exponent comes back
//                in the epilogue (PackedW::wscale);
//   activations  every tensor that is read as a B operand has a per-utterance |max| slot, written by the kernel that
//                produces it (running max of the values it stores, one atomicMax per wave when it grows); the consuming
//                kernel multiplies by 2^-e while staging and by 2^e in its epilogue, e = floor(log2 amax), whenever amax
//                is outside [2^-10, 2^15) - inside that window e = 0 and nothing is scaled.  Intermediates that never leave
//                the CU (fused blocks) use the bound sum|w| * amax_in + max|b| instead of a measured maximum.
// FilterNet Downsample/Upsample convs (decoder.py:143-146, 166-171) and their FiLM (decoder.py:94-97).
//
// Layout.  K is walked in slabs of 16 input channels; a K16 step is (slab, tap).
//   weights   pre-split on the host (api.hip Packer::a6): image [step][m-tile][part][lane][8 fp16], one
//             1 KiB piece per (step, m-tile, part) already in MFMA lane order (row = lane & 31,
//             k = 8 * (lane >> 5) + j); a piece is one 16-byte load + one ds_write_b128 per lane.
//   input     the slab's halo tile is staged once: each thread loads 8 channels of one sample (coalesced
//             along time), applies the pre-activation, splits, and writes two 16-byte rows
//             Xs[part][channel-group][position][8 fp16]; a tap is a row offset, so every ds_read_b128 of
//             the MFMA loop is a contiguous 1 KiB wave access.
//   tile      a wave owns all MTB m-tiles of the workgroup for one 32-sample n-tile.
// Pipeline.  Persistent workgroups: conv launches run ONE 12-wave workgroup per CU (3 m-tiles x 4 column groups, 96 x 128
// or 96 x 256 output tile, LDS dominated by the parked output tile), GEMM launches two 8-wave workgroups per
// CU at a 128-register budget.  Per slab: barrier, registers -> LDS (weights copied, activations split), request the next
// slab (its loads fly across this slab's MFMAs; the first slab of the next phase / tile is requested behind the last
// one), barrier, MFMAs.  Barriers are raw s_barrier + lgkmcnt(0): __syncthreads() also drains vmcnt, i.e. it would wait
// for exactly that prefetch.  Measured alternatives (double-buffered LDS staging, two slabs in flight, transposed
// accumulators, producer / consumer waves, more workgroups per CU): DESIGN.md section 4.
#pragma once
#include <type_traits>
#include "conv_epi.h"
#include "small_kernels.h"
#include "tvc_common.h"

namespace tvc {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // register-friendly 16-byte value (HIP's uint4 struct defeats SROA in arrays)

constexpr int kParts = 2;                 // fp16 parts per fp32 operand
constexpr int kPU4 = kParts * 64;         // uint4 per (K16 step, m-tile) of a weight image
constexpr float kLoScale = 2048.f;        // h2 = fp16((x - h1) * 2^11)
constexpr float kLoInv = 1.f / 2048.f;

// schedule constants (each was swept on the part; the losing settings are described in DESIGN.md section 4)
constexpr int S_WPE = 3;      // waves per SIMD the register budget is sized for (plain / FiLM-fused kernels): 12-wave workgroups, no spills
constexpr int S_WPE_G = 4;    // 8-wave (GEMM) workgroups: two per CU need <= 128 registers
constexpr int S_WPE_F = 3;
constexpr int TVC_S_KG = 2;   // deepest K slab (in 16-channel groups) the GEMM launches use
constexpr int S_BPC = 1;      // persistent workgroups per CU
constexpr int S_FB = 2;       // fragment register sets: 2 = next tap's LDS reads under this tap's MFMAs, 1 = read, then multiply
constexpr int S_FB_G = 1;     // 8-wave (GEMM) workgroups: single fragment set (128-register budget)
constexpr int S_FB_F = 1;     // FiLM-fused kernels: more accumulator sets live, single fragment set keeps the slab loops spill-free
constexpr int S_FB_FW = 2;    // conv phase of the wide FiLM tile
constexpr int S_PF = 2;       // residual rows requested at a time by the lerp epilogue


// epilogues that ask for a second, 1x1 phase over another tensor accumulated into the SAME tile (Downsample: c3(h2) + down_res(xi))
template <class E, class = void>
struct wants_res_conv : std::false_type {};
template <class E>
struct wants_res_conv<E, std::void_t<decltype(E::kResConv)>> : std::bool_constant<E::kResConv> {};

template <int MTB_, int WM_, int NWV_, int WN_ = 1, int KG_ = 1, int MAXD_ = 27>
struct SplitTile {
    static constexpr int MTB = MTB_, WM = WM_, NWV = NWV_, WN = WN_;            // m-tiles per workgroup / per wave, waves along time, n-tiles per wave
    // KG = 16-channel groups staged per slab.  Convs use 1 (the three taps share one halo tile); the plain-GEMM launches
    // use 2-3: with one group a slab carries only 6 MFMAs per wave, far less than the load -> LDS -> barrier round trip.
    static constexpr int KG = KG_;
    static constexpr int MW = MTB / WM, NW = MW * NWV, NTHR = NW * 64;
    static_assert(MTB % WM == 0, "wave rows must tile the workgroup");
    static constexpr int BM = MTB * 32, BN = NWV * WN * 32;
    static constexpr int MAXD = MAXD_, XROW = BN + 2 * MAXD;                    // largest dilation the halo tile must hold (0 for plain GEMMs)
    static constexpr int XG_U4 = kParts * 2 * XROW;                             // one channel group: [part][8-channel half][position]
    static constexpr int X_U4 = KG * XG_U4;
    static constexpr int X_PER = (KG * 2 * XROW + NTHR - 1) / NTHR;             // staging items per thread
    static constexpr int a_u4(int taps) { return taps * KG * MTB * kPU4; }
    static constexpr int stage_u4(int taps) { return a_u4(taps) + X_U4; }
    static constexpr int KS_MAX = 2 * 768;                                      // SCALED launch: factors of <= 768 input channels for the <= 2 utterances a tile touches
    static constexpr int OS = BN + 4;                                           // row stride (floats) of the output tile parked in LDS
    static constexpr int lds_bytes(int taps) {
        const int stage = stage_u4(taps) * 16, out = BM * OS * 4;
        return (stage > out ? stage : out) + 2 * 6 * BM * 4 + KS_MAX * 4 + 64;  // + per-row tables of this workgroup (bias, FiLM biases, weight scales; two tiles' worth) + input-channel factors + the |max| exchange
    }
    static constexpr int bias_off(int taps) {                  // float offset of that area
        const int stage = stage_u4(taps) * 16, out = BM * OS * 4;
        return (stage > out ? stage : out) / 4;
    }
};

struct ConvSArgs {
    const uint4* A6;     // split weight image
    const float* wsc;    // its per-m-tile power-of-two scales (PackedW::wscale): out = acc * wsc[m-tile]
    int MT;              // m-tiles in the image
    const float* x;      // [B][Cin][len], utterance b at x + b * xstride
    long xstride;
    int Cin, len, dil, tiles_per_utt, ntiles;
    int lin = 0;         // LERP kernels: x is the low-rate tensor [B][Cin][lin]; the conv input is F.interpolate(x, scale_factor) = len samples
    float lscale = 0.f;  //               ATen's source-coordinate scale float(1 / scale_factor)
    int cmax = 0;        // > 0: input rows above cmax do not exist (their weights are zero): loads clamp the row index to cmax
    int flatT = 0;       // > 0: flat GEMM tiles over the B * flatT columns (len = B * flatT, B = 1 for the tile walk)
    const float* kscale = nullptr;   // optional per-(utterance, input channel) factor applied while staging (SCALED kernels), [B][Cin]
    const uint4* sc6 = nullptr;   // stacked FiLM [to_scale ; to_shift] image (1x1 over cond), FILM kernels only
    const float* fsc = nullptr;   // per-m-tile scales of sc6 (the residual 1x1's image shares A6's scales: packed jointly)
    const float* cond = nullptr;
    int Ccond = 0;
    // block-floating-point guard of the fp16 split: per-utterance |max| slots [B] of x / cond (read; nullptr = no scaling) and of
    // the output tensor (written; nullptr = nobody reads it as a B operand)
    const float* amax_x = nullptr;
    const float* amax_c = nullptr;
    float* amax_y = nullptr;
    RagDev rag;          // ragged batch (ragged.h, RAG kernels): `len` is the row stride of every tensor, the tile walk and the valid extents come from here
};

// power-of-two input scale from a tensor's per-utterance |max|: identity while amax is inside [2^-10, 2^15), else 2^-floor(log2 amax)
struct Bfp {
    float s, inv;
};
__device__ __forceinline__ Bfp bfp_from_amax(float amax) {
    const unsigned u = __builtin_bit_cast(unsigned, amax);
    int e = (int)(u >> 23) - 127;
    Bfp r{1.f, 1.f};
    if (u != 0u && u < 0x7f800000u && (e >= 15 || e < -10)) {     // zero, Inf and NaN carry no information: no scaling
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
        r.s = __builtin_bit_cast(float, (unsigned)(127 - e) << 23);
        r.inv = __builtin_bit_cast(float, (unsigned)(127 + e) << 23);
    }
    return r;
}
__device__ __forceinline__ Bfp bfp_load(const float* amax, int b) { return amax ? bfp_from_amax(amax[b]) : Bfp{1.f, 1.f}; }
// A slot read through the SCALAR cache (the address must be wave-uniform).  The persistent fused kernels read their utterance's slots at
// the top of every tile; as a vector load (the compiler cannot prove a global store does not alias them) each read was followed by
// `s_waitcnt vmcnt(0)` - i.e. every tile began by waiting for ALL of the previous tile's stores to be acknowledged (cycle stamps,
// tools/micro/u24_trace.py: 2-3 k of a 13 k-cycle tile).  Slots are written by EARLIER launches (caches are invalidated at launch boundaries),
// so the scalar path is coherent; it counts on lgkmcnt and leaves the store queue alone.
__device__ __forceinline__ float sload_f32(const float* p) {
    float v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}
__device__ __forceinline__ Bfp bfp_load_u(const float* amax, int b) { return amax ? bfp_from_amax(sload_f32(amax + b)) : Bfp{1.f, 1.f}; }
// the smaller of two scales (two tensors accumulated into one tile share it)
__device__ __forceinline__ Bfp bfp_min(const Bfp& a, const Bfp& b) { return a.s < b.s ? a : b; }
// Publishing a |max| slot.  Same-address device-scope atomics complete at ~3 per microsecond on this part (measured: one
// atomicMax per wave and tile - 200 k per launch on 64 slots - added 1 ms to a 0.3 ms kernel), so they are kept to a handful per
// slot and launch: persistent kernels walk CONTIGUOUS tile ranges (a workgroup meets one or two utterances), every wave keeps a
// running maximum in a register, and when the workgroup moves on to another utterance (and at its end) the waves' maxima meet in
// LDS and ONE thread issues ONE atomic, fire-and-forget (reading the slot first to skip it made the wave wait for that load and,
// with it, for the next tile's prefetch).  Non-negative floats order like their bit patterns; NaNs never enter a maximum (fmaxf).
__device__ __forceinline__ float wave_max(float mx) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    return mx;
}
// every thread of the workgroup calls it (it contains a barrier); red = LDS scratch of >= (workgroup waves) floats
__device__ __forceinline__ void amax_flush_wg(float* slot, float mx, float* red) {
    mx = wave_max(mx);
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));      // (the LDS address below is not worth a register held - or spilled - across the caller's tile loop)
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (tid == 0) {
        float m = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, red[w]);
        if (m > 0.f) atomicMax(reinterpret_cast<unsigned*>(slot), __builtin_bit_cast(unsigned, m));
    }
    // (red is rewritten at this workgroup's next flush, at least one tile - several barriers - later)
}
// contiguous tile range of persistent workgroup w of g: [first, last)
__device__ __forceinline__ void tile_range(int ntiles, int& first, int& last) {
    first = (int)((long)ntiles * blockIdx.x / gridDim.x);
    last = (int)((long)ntiles * (blockIdx.x + 1) / gridDim.x);
}

// Global accesses as uniform base (SGPR pair) + 32-bit byte offset per lane (the global_load saddr form): the row bases are
// pinned into SGPRs through an empty asm, otherwise the compiler re-associates base + row stride into chains of 64-bit
// vector adds (one v_lshl_add_u64 per load).
typedef const __attribute__((address_space(1))) float* gcf32;
typedef __attribute__((address_space(1))) float* gf32;
__device__ __forceinline__ float ldg_so(const float* base, unsigned byte_off) {
    gcf32 p = (gcf32)base;
    asm("" : "+s"(p));
    return *reinterpret_cast<gcf32>(reinterpret_cast<const __attribute__((address_space(1))) char*>(p) + byte_off);
}
__device__ __forceinline__ void stg_so(float* base, unsigned byte_off, float v) {
    gf32 p = (gf32)base;
    asm("" : "+s"(p));
    *reinterpret_cast<gf32>(reinterpret_cast<__attribute__((address_space(1))) char*>(p) + byte_off) = v;
}
__device__ __forceinline__ void stg_so4(float* base, unsigned byte_off, const float (&v)[4]) {
    typedef float f32x4g __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(1))) f32x4g* gf4;
    gf32 p = (gf32)base;
    asm("" : "+s"(p));
    *reinterpret_cast<gf4>(reinterpret_cast<__attribute__((address_space(1))) char*>(p) + byte_off) = f32x4g{v[0], v[1], v[2], v[3]};
}
__device__ __forceinline__ u32x4 ldg_so4(const uint4* base, unsigned byte_off) {
    typedef const __attribute__((address_space(1))) u32x4* gcu4;
    gcu4 p = (gcu4)base;
    asm("" : "+s"(p));
    return *reinterpret_cast<gcu4>(reinterpret_cast<const __attribute__((address_space(1))) char*>(p) + byte_off);
}


// The two parts of a pair of fp32 values: h1 = fp16(v) (packed), h2 = fp16((v - h1) * S), S = kLoScale (or 1: film_s2.h).
// The second part is ONE v_fma_mix per value - fp16(fma(h1, -S, v * S)), the fp16 operand read straight from h1's register half -
// instead of convert-back, subtract, scale, convert: four vector instructions per pair instead of six, on the path every staged
// activation takes.  The product h1 * S and the difference are exact in fp32, so the single rounding equals the four-step result bit for
// bit (tools/micro/split_mix.hip: 2 M pairs over the whole exponent range, none different).
template <bool SCALED = true>
__device__ __forceinline__ void split2(float v0, float v1, unsigned& p1, unsigned& p2) {
    const f32x2 a = {v0, v1};
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(a, f16x2v));
    const f32x2 b = SCALED ? a * kLoScale : a;
    const float ns = SCALED ? -kLoScale : -1.f;
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p1), "s"(ns), "v"(b[0]));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(p1), "s"(ns), "v"(b[1]));
    p2 = r;
}
// two fp16 parts of 8 fp32 values (v = h1 + 2^-11 h2), packed for one 16-byte LDS row each
__device__ __forceinline__ void split8(const float (&v)[8], uint4& p1, uint4& p2) {
    unsigned o1[4], o2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split2(v[2 * j], v[2 * j + 1], o1[j], o2[j]);
    p1 = make_uint4(o1[0], o1[1], o1[2], o1[3]);
    p2 = make_uint4(o2[0], o2[1], o2[2], o2[3]);
}
// acc_hi += w1 x1;  acc_lo += w2 x1 + w1 x2   (one K16 step of one 32 x 32 tile)
#define TVC_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
// out = acc_hi * c + acc_lo * (c / 2048)
__device__ __forceinline__ float comb(float hi, float lo, float c, float clo) { return fmaf(lo, clo, hi * c); }

// workgroup barrier that drains this wave's LDS traffic but not its global loads
__device__ __forceinline__ void slab_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifdef S_TRACE
// diagnostic build only: cycle stamps of one wave's walk through the slab loop (tools/micro/slab_trace.py)
static __device__ unsigned long long g_trace[64 * 256];
static __device__ unsigned g_trace_slot;
#ifndef S_TRACE_WG
#define S_TRACE_WG 0
#endif
#ifndef S_TRACE_TID
#define S_TRACE_TID 0
#endif
#define TR_STAMP(r, id)                                                                              \
    do {                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        if ((r).tr && (r).trn < 250) {                                                               \
            unsigned long long t_;                                                                   \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
            (r).tr[(r).trn++] = (t_ << 8) | (unsigned)(id);                                          \
        }                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                           \
    } while (0)
#else
#define TR_STAMP(r, id) do {} while (0)
#endif

// Staging registers of one thread (one slab in flight) and its share of the halo tile.
template <class TL>
struct SlabRegs {
    static constexpr int A_MAX = (3 * TL::KG * TL::MTB * kParts + TL::NW - 1) / TL::NW;   // up to 3 taps x KG channel groups of weight pieces
    u32x4 ar[A_MAX];
    float xr[TL::X_PER][8];
    float xr2[TL::X_PER][8];   // LERP staging: the second interpolation tap
#ifdef S_TRACE
    unsigned long long* tr = nullptr;
    int trn = 0;
#endif
};
template <class TL>
struct SlabMap {
    unsigned xo[TL::X_PER];    // utterance-relative element offset of (channel 8 g, position p)
    int xdst[TL::X_PER];       // LDS row of the item, -1 = idle
    int xg8[TL::X_PER];        // first channel of the item inside the slab (0 or 8)
    int xk[TL::X_PER];         // flat GEMM tiles: factor-row offset of the item's utterance inside the staged Ks
    unsigned xo1[TL::X_PER];   // LERP staging: offset of the second tap; w0 / w1 = the two weights (xo = first tap)
    float w0[TL::X_PER], w1[TL::X_PER];
    float xs[TL::X_PER];       // block-floating-point scale of the item's utterance (1 unless its |max| is outside the fp16 window)
};
// fT > 0 = flat GEMM tiles: the tile's columns are positions n = b * fT + t of the flattened [B * fT] axis (len = B * fT),
// utterance b starts at element b * fstride, channels are fT apart; a tile may straddle utterances.
template <class TL, bool LERP = false>
__device__ __forceinline__ void make_map(SlabMap<TL>& m, int len, int dil, int t0, int fT = 0, unsigned fstride = 0, int kcin = 0, int lin = 0,
                                         float lscale = 0.f, float xs = 1.f, const float* amax = nullptr, int rs_ = 0, const int* __restrict__ c2b = nullptr, int cmult = 1) {
    // ragged batches (ragged.h): rs_ = row stride of the tensor when it is not `len` (len = the tile's utterance, xb points at its first column);
    // c2b = frame -> utterance for GEMM tiles over the whole batch (the per-utterance scale is then the column's, as on flat tiles)
    const int rs = rs_ ? rs_ : len;
    const int xw = TL::BN + 2 * dil;
    // (the thread index is laundered: the map depends on it and on launch constants only, so the compiler would compute the items' (group, column)
    // once per kernel and keep them - then spill them - across every tile's MFMA phases; recomputing them per phase costs a few instructions)
    int tid0 = threadIdx.x;
    asm volatile("" : "+v"(tid0));
    __builtin_assume(tid0 >= 0 && tid0 < TL::NTHR);
#pragma unroll
    for (int i = 0; i < TL::X_PER; ++i) {
        int idx = tid0 + i * TL::NTHR;
        int gk = idx / xw, c = idx - gk * xw;                // gk = 2 * (channel group) + (8-channel half)
        int p = t0 - dil + c;
        p = p < 0 ? 0 : (p > len - 1 ? len - 1 : p);
        const bool live = gk < 2 * TL::KG;
        gk = live ? gk : 2 * TL::KG - 1;                     // idle items still load (valid address), never store
        const int ks = gk >> 1, g = gk & 1;
        m.xdst[i] = live ? ks * TL::XG_U4 + g * TL::XROW + c : -1;
        m.xg8[i] = ks * 16 + 8 * g;
        m.xs[i] = xs;
        if (fT > 0) {
            const int b = p / fT, t = p - b * fT;
            m.xo[i] = (unsigned)b * fstride + (unsigned)(m.xg8[i] * fT + t);
            m.xk[i] = (b - t0 / fT) * kcin;
            m.xs[i] = bfp_load(amax, b).s;                   // flat tiles straddle utterances: the scale is the column's
        } else if (LERP) {
            // the conv input at position p (already clamped = replicate padding of the interpolated signal) is
            // w0 * x[i0] + w1 * x[i1] of the low-rate row (ATen linear, align_corners = False: small_kernels.h)
            const Lerp lc = lerp_coord(p, lscale, lin);
            m.xo[i] = (unsigned)(m.xg8[i] * lin + lc.i0);
            m.xo1[i] = (unsigned)(m.xg8[i] * lin + lc.i1);
            m.w0[i] = lc.w0;
            m.w1[i] = lc.w1;
            m.xk[i] = 0;
        } else {
            m.xo[i] = (unsigned)(m.xg8[i] * rs + p);
            m.xk[i] = 0;
            if (c2b) m.xs[i] = bfp_load(amax, c2b[p / cmult]).s;
        }
    }
}
// global -> registers only (no use of the values here: the loads stay in flight behind the MFMAs)
template <class TL, int TAPS, bool LERP = false, bool CLAMP = false>
__device__ __forceinline__ void slab_load(SlabRegs<TL>& r, const SlabMap<TL>& m, const uint4* __restrict__ A6, int MT, int mt0,
                                          const float* __restrict__ xb, int Cin, int len, int s, int fT = 0, int cmax = 0, int lin = 0, int rs_ = 0, int tid_ = -1) {
    constexpr int MTB = TL::MTB, NW = TL::NW, X_PER = TL::X_PER, STEPS = TAPS * TL::KG;
    const int cs = LERP ? lin : (fT > 0 ? fT : (rs_ ? rs_ : len));     // channel stride
    constexpr int PIECES = STEPS * MTB * kParts, A_PER = (PIECES + NW - 1) / NW;
    static_assert(A_PER <= SlabRegs<TL>::A_MAX, "weight pieces must fit the staging registers");
    const int tid = tid_ >= 0 ? tid_ : (int)threadIdx.x;      // (split_phase passes its laundered copy: the piece offsets live for one phase, not across the tile loop)
    __builtin_assume(tid >= 0 && tid < TL::NTHR);
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (the laundered copy has lost its "wave-uniform" property: say it again, or `q < PIECES` becomes an exec-masked branch)
    __builtin_assume(wave >= 0 && wave < NW);                                        // (... and its range: the first pieces' `q < PIECES` are decided at compile time)
    const int ci0 = s * 16 * TL::KG;
    {
        const uint4* a_src = A6 + (long)mt0 * kPU4;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int q = wave + i * NW;
            q = q < PIECES ? q : PIECES - 1;
            const int tap = q / (MTB * kParts), rem = q - tap * (MTB * kParts);
            r.ar[i] = ldg_so4(a_src + (long)s * STEPS * MT * kPU4, 16u * (unsigned)(tap * MT * kPU4 + rem * 64 + lane));   // uniform slab base + this wave's piece
        }
    }
    const float* xc = xb + (long)ci0 * cs;               // uniform base, 32-bit lane offsets
    // Cin % 16 == 0 is a launch precondition (every level routed here has 48/96/192/384 channels): no ragged slab
    if (LERP) {
#pragma unroll
        for (int i = 0; i < X_PER; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                r.xr[i][j] = ldg_so(xc + (long)j * cs, 4u * m.xo[i]);
                r.xr2[i][j] = ldg_so(xc + (long)j * cs, 4u * m.xo1[i]);
            }
        return;
    }
    if constexpr (CLAMP) {   // input rows above cmax do not exist (their weights are zero): read row cmax instead; one code path, no branch
#pragma unroll
        for (int i = 0; i < X_PER; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int cj = ci0 + m.xg8[i] + j;
                cj = cj < cmax ? cj : cmax;
                r.xr[i][j] = xb[m.xo[i] + (unsigned)((cj - m.xg8[i]) * cs)];     // xo already carries the xg8 * cs channel term
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < X_PER; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) r.xr[i][j] = ldg_so(xc + (long)j * cs, 4u * m.xo[i]);   // uniform row base + one 32-bit lane offset per item
}
// slab 0 of a phase, issued by whoever runs before it (previous phase / previous tile / kernel entry)
template <class TL, int TAPS, bool LERP = false, bool CLAMP = false>
__device__ __forceinline__ void first_load(SlabRegs<TL>& r, const uint4* __restrict__ A6, int MT, int mt0, const float* __restrict__ xb,
                                           int Cin, int len, int dil, int t0, int fT = 0, unsigned fstride = 0, int cmax = 0, int lin = 0,
                                           float lscale = 0.f, int rs_ = 0) {
    SlabMap<TL> m;
    make_map<TL, LERP>(m, len, dil, t0, fT, fstride, 0, lin, lscale, 1.f, nullptr, rs_);
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    slab_load<TL, TAPS, LERP, CLAMP>(r, m, A6, MT, mt0, xb, Cin, len, 0, fT, cmax, lin, rs_, tid);
}

// (hi, lo) += W (.) x over all slabs of one input tensor (hi: the h1 w1 products, lo: h1 w2 + h2 w1 in units of 2^-11).
// Slab 0 is already in flight in `r` (first_load); `next()` is called in its place behind the last slab, so the following
// phase or tile starts without a cold load.  xs = the tile's block-floating-point input scale (flat GEMM tiles: per column,
// from `amax`), applied while staging; the caller's epilogue multiplies by its inverse.
struct NoMid { __device__ __forceinline__ void operator()() const {} };
// TWO: the same input tile is multiplied with two row blocks of the image one after the other (mt0, then mt0b) in ONE slab loop -
// FiLM's scale and shift on wide tiles, `mid()` is called between the two (it folds the first result away and clears the accumulators)
template <class TL, int TAPS, int A_U4, bool LRELU, int FB, bool SCALED = false, bool LERP = false, bool CLAMP = false, bool TWO = false, class Next, class Mid = NoMid>
__device__ __forceinline__ void split_phase(f32x16 (&hi)[TL::WM][TL::WN], f32x16 (&lo)[TL::WM][TL::WN], SlabRegs<TL>& r, const uint4* __restrict__ A6, int MT, int mt0,
                                            const float* __restrict__ xb, int Cin, int len, int dil, int t0, uint4* As, uint4* Xs, Next next, float xs,
                                            const float* Ks = nullptr, int fT = 0, unsigned fstride = 0, int cmax = 0, int lin = 0, float lscale = 0.f,
                                            int mt0b = 0, Mid mid = Mid(), const float* amax = nullptr, int rs_ = 0, const int* __restrict__ c2b = nullptr, int cmult = 1) {
    constexpr int MTB = TL::MTB, WM = TL::WM, WN = TL::WN, NWV = TL::NWV, NW = TL::NW, XROW = TL::XROW, X_PER = TL::X_PER;
    // (laundered thread index: everything derived from it - piece offsets, LDS fragment addresses - is recomputed per phase instead of being
    // hoisted out of the persistent tile loop, kept live across every other phase and spilled)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    __builtin_assume(tid >= 0 && tid < TL::NTHR);      // (... and its range, which decides `q < PIECES` for the first pieces at compile time)
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (uniform again: see slab_load)
    __builtin_assume(wave >= 0 && wave < NW);
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / NWV, wn = wave - wm * NWV;
    SlabMap<TL> m;
    make_map<TL, LERP>(m, len, dil, t0, fT, fstride, Cin, lin, lscale, xs, amax, rs_, c2b, cmult);
    constexpr int STEPS = TAPS * TL::KG;               // K16 steps per slab: (channel group, tap)
    constexpr int PIECES = STEPS * MTB * kParts, A_PER = (PIECES + NW - 1) / NW;
    auto lstore = [&](int sl) __attribute__((always_inline)) {
        uint4* Asb = As;
        uint4* Xsb = Xs;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int q = wave + i * NW;
            if (q < PIECES) *reinterpret_cast<u32x4*>(Asb + q * 64 + lane) = r.ar[i];
        }
#pragma unroll
        for (int i = 0; i < X_PER; ++i)
            if (m.xdst[i] >= 0) {
                if (LERP) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) r.xr[i][j] = fmaf(m.w0[i], r.xr[i][j], __fmul_rn(m.w1[i], r.xr2[i][j]));   // = lerp_eval
                }
                if (LRELU) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) r.xr[i][j] = fmaxf(r.xr[i][j], 0.1f * r.xr[i][j]);   // = leaky_relu(x, 0.1)
                }
                if (SCALED) {
                    const float4 k0 = *reinterpret_cast<const float4*>(Ks + m.xk[i] + sl * 16 * TL::KG + m.xg8[i]);
                    const float4 k1 = *reinterpret_cast<const float4*>(Ks + m.xk[i] + sl * 16 * TL::KG + m.xg8[i] + 4);
                    r.xr[i][0] *= k0.x; r.xr[i][1] *= k0.y; r.xr[i][2] *= k0.z; r.xr[i][3] *= k0.w;
                    r.xr[i][4] *= k1.x; r.xr[i][5] *= k1.y; r.xr[i][6] *= k1.z; r.xr[i][7] *= k1.w;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) r.xr[i][j] *= m.xs[i];                                     // power of two: exact
                uint4 p1, p2;
                split8(r.xr[i], p1, p2);
                Xsb[m.xdst[i]] = p1;
                Xsb[2 * XROW + m.xdst[i]] = p2;
            }
    };

    const int nslab1 = Cin / (16 * TL::KG);
    const int nslab = TWO ? 2 * nslab1 : nslab1;
    const uint4* as0 = As + wm * WM * kPU4 + lane;
    const uint4* xs0 = Xs + lh * XROW + wn * WN * 32 + l31;
    for (int s = 0; s < nslab; ++s) {
        TR_STAMP(r, 0);
        slab_barrier();                            // every wave is done reading the previous slab
        TR_STAMP(r, 1);
        if (TWO && s == nslab1) mid();
        lstore(TWO && s >= nslab1 ? s - nslab1 : s);   // slab s: registers -> LDS
        TR_STAMP(r, 2);
        if (s + 1 < nslab) {                       // flies across this slab's MFMAs
            if (TWO) slab_load<TL, TAPS, LERP, CLAMP>(r, m, A6, MT, s + 1 >= nslab1 ? mt0b : mt0, xb, Cin, len, s + 1 >= nslab1 ? s + 1 - nslab1 : s + 1, fT, cmax, lin, rs_, tid);
            else slab_load<TL, TAPS, LERP, CLAMP>(r, m, A6, MT, mt0, xb, Cin, len, s + 1, fT, cmax, lin, rs_, tid);
        } else next();
        TR_STAMP(r, 3);
        slab_barrier();
        TR_STAMP(r, 4);
        const uint4* as = as0;
        const uint4* xs_ = xs0;
        // fragments of tap t+1 are read while the MFMAs of tap t run
        f16x8 af[2][WM][kParts], bf[2][WN][kParts];
        auto frags = [&](int tap, int fb) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int p = 0; p < kParts; ++p)
                    bf[fb][j][p] = __builtin_bit_cast(f16x8, xs_[(tap / TAPS) * TL::XG_U4 + 2 * p * XROW + j * 32 + (tap % TAPS) * dil]);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int p = 0; p < kParts; ++p) af[fb][i][p] = __builtin_bit_cast(f16x8, as[(tap * MTB * kParts + i * kParts + p) * 64]);
        };
        if (FB == 2) frags(0, 0);
#pragma unroll
        for (int tap = 0; tap < STEPS; ++tap) {        // tap = step index (channel group * TAPS + tap)
            const int fb = FB == 2 ? tap & 1 : 0;
            if (FB == 2) {
                if (tap + 1 < STEPS) frags(tap + 1, fb ^ 1);
                __builtin_amdgcn_sched_barrier(0);   // keep the reads above the MFMAs (the scheduler sinks them to just-in-time otherwise)
            } else {
                frags(tap, 0);
            }
            // three part-products per tile, the two accumulators of a tile alternate (independent MFMAs back to back)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) lo[i][j] = TVC_MFMA16(af[fb][i][1], bf[fb][j][0], lo[i][j]);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) hi[i][j] = TVC_MFMA16(af[fb][i][0], bf[fb][j][0], hi[i][j]);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) lo[i][j] = TVC_MFMA16(af[fb][i][0], bf[fb][j][1], lo[i][j]);
            if (FB == 2) __builtin_amdgcn_sched_barrier(0);
        }
        TR_STAMP(r, 5);
    }
}

// FiLM scale and shift in ONE 1x1 phase over the cond tile: the stacked [to_scale ; to_shift] weight image supplies two
// m-tile groups (rows mt0.. and mtoff + mt0..) that share every staged input slab and every B fragment, so the cond
// tile is loaded, split and written once instead of twice and a slab carries 6 MFMAs per wave instead of 3.
template <class TL>
__device__ __forceinline__ void film_load(SlabRegs<TL>& r, const SlabMap<TL>& m, const uint4* __restrict__ F6, int MT, int mt0, int mtoff,
                                          const float* __restrict__ xb, int len, int s, int rs_ = 0) {
    const int rs = rs_ ? rs_ : len;                      // row stride of the cond tensor (ragged batches: not the utterance's length)
    constexpr int MTB = TL::MTB, NW = TL::NW, X_PER = TL::X_PER;
    constexpr int KG = TL::KG, GP = 2 * MTB * kParts, PIECES = KG * GP, A_PER = (PIECES + NW - 1) / NW;   // per 16-channel group: scale and shift pieces of MTB m-tiles
    static_assert(A_PER <= SlabRegs<TL>::A_MAX, "FiLM weight pieces must fit the staging registers");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        int q = wave + i * NW;
        q = q < PIECES ? q : PIECES - 1;
        const int kg = q / GP, q2 = q - kg * GP;
        const int grp = q2 / (MTB * kParts), rem = q2 - grp * (MTB * kParts);
        r.ar[i] = ldg_so4(F6 + ((long)s * KG * MT + mt0) * kPU4, 16u * (unsigned)((kg * MT + grp * mtoff) * kPU4 + rem * 64 + lane));
    }
    const float* xc = xb + (long)s * 16 * KG * rs;
#pragma unroll
    for (int i = 0; i < X_PER; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) r.xr[i][j] = ldg_so(xc + (long)j * rs, 4u * m.xo[i]);
}
template <class TL>
__device__ __forceinline__ void film_first_load(SlabRegs<TL>& r, const uint4* __restrict__ F6, int MT, int mt0, int mtoff,
                                                const float* __restrict__ xb, int len, int t0, int rs_ = 0) {
    SlabMap<TL> m;
    make_map<TL>(m, len, 0, t0, 0, 0u, 0, 0, 0.f, 1.f, nullptr, rs_);
    film_load<TL>(r, m, F6, MT, mt0, mtoff, xb, len, 0, rs_);
}
// (scale, shift) accumulator pairs: (asc, lsc), (ash, lsh)
template <class TL, class Next>
__device__ __forceinline__ void film_phase(f32x16 (&asc)[TL::WM][TL::WN], f32x16 (&lsc)[TL::WM][TL::WN], f32x16 (&ash)[TL::WM][TL::WN], f32x16 (&lsh)[TL::WM][TL::WN],
                                           SlabRegs<TL>& r, const uint4* __restrict__ F6,
                                           int MT, int mt0, int mtoff, const float* __restrict__ xb, int Cin, int len, int t0, uint4* As, uint4* Xs,
                                           Next next, float xs, int rs_ = 0) {
    constexpr int MTB = TL::MTB, WM = TL::WM, WN = TL::WN, NWV = TL::NWV, NW = TL::NW, XROW = TL::XROW, X_PER = TL::X_PER;
    constexpr int KG = TL::KG, GP = 2 * MTB * kParts, PIECES = KG * GP, A_PER = (PIECES + NW - 1) / NW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / NWV, wn = wave - wm * NWV;
    SlabMap<TL> m;
    make_map<TL>(m, len, 0, t0, 0, 0u, 0, 0, 0.f, 1.f, nullptr, rs_);
    const int nslab = Cin / (16 * KG);
    const uint4* as0 = As + wm * WM * kPU4 + lane;
    const uint4* xs0 = Xs + lh * XROW + wn * WN * 32 + l31;
    auto lstore = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int q = wave + i * NW;
            if (q < PIECES) *reinterpret_cast<u32x4*>(As + q * 64 + lane) = r.ar[i];
        }
#pragma unroll
        for (int i = 0; i < X_PER; ++i)
            if (m.xdst[i] >= 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) r.xr[i][j] *= xs;
                uint4 p1, p2;
                split8(r.xr[i], p1, p2);
                Xs[m.xdst[i]] = p1;
                Xs[2 * XROW + m.xdst[i]] = p2;
            }
    };
    for (int s = 0; s < nslab; ++s) {
        slab_barrier();
        lstore();
        if (s + 1 < nslab) film_load<TL>(r, m, F6, MT, mt0, mtoff, xb, len, s + 1, rs_);
        else next();
        slab_barrier();
        const uint4* as = as0;
        const uint4* xq = xs0;
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            f16x8 fc[WM][kParts], fh[WM][kParts], bf[WN][kParts];
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int p = 0; p < kParts; ++p) bf[j][p] = __builtin_bit_cast(f16x8, xq[kg * TL::XG_U4 + 2 * p * XROW + j * 32]);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int p = 0; p < kParts; ++p) {
                    fc[i][p] = __builtin_bit_cast(f16x8, as[(kg * GP + i * kParts + p) * 64]);
                    fh[i][p] = __builtin_bit_cast(f16x8, as[(kg * GP + MTB * kParts + i * kParts + p) * 64]);
                }
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    lsc[i][j] = TVC_MFMA16(fc[i][1], bf[j][0], lsc[i][j]);
                    lsh[i][j] = TVC_MFMA16(fh[i][1], bf[j][0], lsh[i][j]);
                    asc[i][j] = TVC_MFMA16(fc[i][0], bf[j][0], asc[i][j]);
                    ash[i][j] = TVC_MFMA16(fh[i][0], bf[j][0], ash[i][j]);
                    lsc[i][j] = TVC_MFMA16(fc[i][0], bf[j][1], lsc[i][j]);
                    lsh[i][j] = TVC_MFMA16(fh[i][0], bf[j][1], lsh[i][j]);
                }
        }
    }
}

// Output tile -> HBM through LDS: the accumulator layout gives a lane one sample of 16 different rows (4-byte
// stores, 128 B per row and instruction); parked as Ot[row][sample] the tile leaves as 16-byte stores (and the
// residual arrives as 16-byte loads) along time.  v already holds everything but the residual.
// mx: the wave's running |max| of the values it stores (the decimated copy y2 is a subset / two-sample mean of them), published
// by the kernel when the workgroup leaves the utterance.
template <class TL, bool RES>
__device__ __forceinline__ void tile_store(float* Ot, const f32x16 (&v)[TL::WM][TL::WN], float* __restrict__ y, const float* __restrict__ res,
                                           int b, int M, int len, int mt0, int t0, float& mx, float* __restrict__ y2 = nullptr, int f2 = 0, int rlin = 0, float rscale = 0.f,
                                           int rs_ = 0, int coloff = 0) {
    // ragged batches (ragged.h): rs_ = the tensors' row stride, coloff = first column of the tile's utterance (b = 0 then), len = its length;
    // an interpolated residual's low-rate tensor is laid out alike: rlin = ITS row stride, the utterance's low-rate length = len / (rs / rlin)
    const int rs = rs_ ? rs_ : len;
    // (uniform integer divisions run on the vector ALU: readfirstlane brings the quotients back where the uniform-base loads below need them)
    const int rvalid = (rs_ && rlin > 0) ? __builtin_amdgcn_readfirstlane(len / (rs / rlin)) : rlin;         // low-rate samples of this utterance
    const int rcol = (rs_ && rlin > 0) ? __builtin_amdgcn_readfirstlane(coloff / (rs / rlin)) : 0;           // its first low-rate column
    constexpr int WM = TL::WM, WN = TL::WN, BM = TL::BM, BN = TL::BN, OS = TL::OS, NTHR = TL::NTHR;
    // The thread index is laundered through an empty asm: everything below depends on it only, so the compiler would hoist
    // all of the store pass's index math (64-bit offsets included) out of the persistent tile loop and keep ~25 registers
    // live across the MFMA phases - enough to push the FiLM kernels into scratch.  Recomputing it per tile is free.
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int wm = wave / TL::NWV, wn = wave - wm * TL::NWV;
    auto park = [&]() __attribute__((always_inline)) {
        slab_barrier();                                // every wave is done with the staging buffers
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Ot[((wm * WM + i) * 32 + 4 * lh + (r & 3) + 8 * (r >> 2)) * OS + (wn * WN + j) * 32 + l31] = v[i][j][r];
        slab_barrier();
    };
    __builtin_amdgcn_sched_barrier(0);                         // nothing of the store pass moves up into the FiLM combine (three accumulator sets live there)
    float* yb = y + ((long)b * M + mt0 * 32) * rs + coloff + t0;      // offsets inside the tile's rows fit 32 bits
    // rlin > 0: the residual is F.interpolate(res_low) of a [B][M][rlin] tensor, evaluated here instead of read back
    const float* rb = RES ? (rlin > 0 ? res + ((long)b * M + mt0 * 32) * rlin + rcol : res + ((long)b * M + mt0 * 32) * rs + coloff + t0) : nullptr;
    const int rows = M - mt0 * 32 < BM ? M - mt0 * 32 : BM;
    const bool vec = ((rs | coloff) & 3) == 0;              // rows start 16-byte aligned (t0 is a multiple of 32)
    // optional 1/f2-rate copy for the next Downsample block (see C3EpiBias): pick for f2 = 3 / 5, two-sample mean for f2 = 4
    const int len2 = f2 > 0 ? rs / f2 : 0;                  // (row stride of the decimated copy)
    float* y2b = y2 ? y2 + ((long)b * M + mt0 * 32) * len2 + (f2 > 0 ? coloff / f2 : 0) : nullptr;
    if constexpr (RES) {
        if (rlin > 0) {
            // a thread's four columns are the same for every row it visits: the interpolation coordinates are computed once
            constexpr int G = BN / 4, RS = NTHR / G;
            static_assert(NTHR % G == 0, "column groups must tile the workgroup");
            const int c = (tid % G) * 4;
            unsigned o0[4], o1[4];
            float lam[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + c + u < len ? t0 + c + u : len - 1;
                const Lerp lc = lerp_coord(t, rscale, rvalid);
                o0[u] = 4u * (unsigned)lc.i0;
                o1[u] = 4u * (unsigned)lc.i1;
                lam[u] = lc.w1;
            }
            const bool full = vec && t0 + c + 3 < len;
            // the residual taps of every row this thread will store are requested before the tile is parked: their latency
            // runs under the two barriers and the LDS round trip instead of in front of every store
            // (two rows at a time: all four at once pushed the FiLM kernels into scratch)
            constexpr int NR = (BM + RS - 1) / RS, PF = S_PF;
            float x0[PF][4], x1[PF][4];
            const bool live = t0 + c < len;
            auto request = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
                for (int k = 0; k < PF; ++k) {
                    int row = tid / G + (k0 + k) * RS;
                    row = row < rows ? row : rows - 1;
                    const unsigned ro = 4u * (unsigned)(row * rlin);          // uniform tile base + 32-bit byte offsets (rows of one tile are < 2^30 B apart)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        x0[k][u] = ldg_so(rb, ro + o0[u]);
                        x1[k][u] = ldg_so(rb, ro + o1[u]);
                    }
                }
            };
            __builtin_amdgcn_sched_barrier(0);     // the requests stay below the FiLM combine: hoisted above it they overlap the three live accumulator sets
            request(0);
            park();
            if (live) {
#pragma unroll
            for (int k0 = 0; k0 < NR; k0 += PF) {
                float w[PF][4];
#pragma unroll
                for (int k = 0; k < PF; ++k)
#pragma unroll
                    for (int u = 0; u < 4; ++u) w[k][u] = fmaf(1.f - lam[u], x0[k][u], __fmul_rn(lam[u], x1[k][u]));          // = lerp_eval
                if (k0 + PF < NR) request(k0 + PF);                  // next rows fly while these are stored
#pragma unroll
                for (int k = 0; k < PF; ++k) {
                    const int row = tid / G + (k0 + k) * RS;
                    if (row >= rows) break;
                    const float4 o = *reinterpret_cast<const float4*>(Ot + row * OS + c);
                    const float e[4] = {o.x + w[k][0], o.y + w[k][1], o.z + w[k][2], o.w + w[k][3]};
                    const int off = row * rs + c;
                    if (full) {
                        *reinterpret_cast<float4*>(yb + off) = make_float4(e[0], e[1], e[2], e[3]);
                        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(e[0]), fabsf(e[1]))), fmaxf(fabsf(e[2]), fabsf(e[3])));
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (t0 + c + u < len) {
                                yb[off + u] = e[u];
                                mx = fmaxf(mx, fabsf(e[u]));
                            }
                    }
                }
            }
            }
            return;
        }
        if (vec && t0 + BN <= len) {
            // whole tile inside the utterance: the residual rows are requested before the tile is parked (same reason)
            constexpr int G = BN / 4, RS = NTHR / G, NR = (BM + RS - 1) / RS;
            const int c = (tid % G) * 4;
            float4 q[NR];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                int row = tid / G + k * RS;
                row = row < rows ? row : rows - 1;
                q[k] = *reinterpret_cast<const float4*>(rb + row * rs + c);
            }
            park();
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                const int row = tid / G + k * RS;
                if (row >= rows) break;
                float4 w = *reinterpret_cast<const float4*>(Ot + row * OS + c);
                w.x += q[k].x; w.y += q[k].y; w.z += q[k].z; w.w += q[k].w;
                *reinterpret_cast<float4*>(yb + row * rs + c) = w;
                mx = fmaxf(fmaxf(mx, fmaxf(fabsf(w.x), fabsf(w.y))), fmaxf(fabsf(w.z), fabsf(w.w)));
                if (y2b) {
                    const float e[4] = {w.x, w.y, w.z, w.w};
                    if (f2 == 4) {
                        y2b[row * len2 + ((t0 + c) >> 2)] = fmaf(0.5f, e[1], __fmul_rn(0.5f, e[2]));
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int t = t0 + c + u, qq = t / f2;
                            if (t - qq * f2 == (f2 >> 1)) y2b[row * len2 + qq] = e[u];
                        }
                    }
                }
            }
            return;
        }
    }
    park();
#pragma unroll 2
    for (int idx = tid; idx < BM * (BN / 4); idx += NTHR) {
        const int row = idx / (BN / 4), c = (idx - row * (BN / 4)) * 4;
        if (row >= rows || t0 + c >= len) continue;
        const float4 o = *reinterpret_cast<const float4*>(Ot + row * OS + c);
        const int off = row * rs + c;
        if (RES && rlin > 0) {
            const float e[4] = {o.x, o.y, o.z, o.w};
            float w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + c + u < len ? t0 + c + u : len - 1;
                const Lerp lc = lerp_coord(t, rscale, rvalid);
                w[u] = e[u] + lerp_eval(lc, rb[row * rlin + lc.i0], rb[row * rlin + lc.i1]);
            }
            if (vec && t0 + c + 3 < len) {
                *reinterpret_cast<float4*>(yb + off) = make_float4(w[0], w[1], w[2], w[3]);
                mx = fmaxf(fmaxf(mx, fmaxf(fabsf(w[0]), fabsf(w[1]))), fmaxf(fabsf(w[2]), fabsf(w[3])));
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (t0 + c + u < len) {
                        yb[off + u] = w[u];
                        mx = fmaxf(mx, fabsf(w[u]));
                    }
            }
        } else if (vec && t0 + c + 3 < len) {
            float4 w = o;
            if (RES) {
                const float4 q = *reinterpret_cast<const float4*>(rb + off);
                w.x += q.x; w.y += q.y; w.z += q.z; w.w += q.w;
            }
            *reinterpret_cast<float4*>(yb + off) = w;
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(w.x), fabsf(w.y))), fmaxf(fabsf(w.z), fabsf(w.w)));
            if (y2b) {
                const float e[4] = {w.x, w.y, w.z, w.w};
                if (f2 == 4) {
                    y2b[row * len2 + ((t0 + c) >> 2)] = fmaf(0.5f, e[1], __fmul_rn(0.5f, e[2]));
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int t = t0 + c + u, q = t / f2;
                        if (t - q * f2 == (f2 >> 1)) y2b[row * len2 + q] = e[u];
                    }
                }
            }
        } else {
            const float e[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (t0 + c + u < len) {
                    const float ov = RES ? e[u] + rb[off + u] : e[u];
                    yb[off + u] = ov;
                    mx = fmaxf(mx, fabsf(ov));
                    if (y2b && f2 != 4) {
                        const int t = t0 + c + u, q = t / f2;
                        if (t - q * f2 == (f2 >> 1)) y2b[row * len2 + q] = ov;
                    }
                }
        }
    }
}

// RAG: a ragged batch (ragged.h).  Time-tiled launches (plain convs, Downsample's c3 + down_res) walk the batch's column-tile table, `a.len` is
// the row stride and every tile takes its utterance's first column / length from the table; GEMM launches walk all columns of the batch and
// look the column's utterance up for the block-floating-point scale only.
template <class TL, int TAPS, bool LRELU, class Epi, bool FILM, bool SCALED = false, bool LERP = false, bool CLAMP = false, bool RAG = false>
__global__ __launch_bounds__(TL::NTHR) __attribute__((amdgpu_waves_per_eu(FILM ? S_WPE_F : (Epi::kIgemm ? (TL::NW <= 8 ? S_WPE_G : 5) : S_WPE)))) void conv3s_kernel(ConvSArgs a, Epi ep) {
    static_assert(!RAG || (!LERP && !SCALED && !(FILM && TL::WN > 1)), "ragged batches: plain / residual-conv / narrow FiLM / GEMM launches only");
    constexpr bool RAGT = RAG && !Epi::kIgemm;      // time-tiled ragged walk
    constexpr bool RAGG = RAG && Epi::kIgemm;       // GEMM over the whole ragged batch
    constexpr int MTB = TL::MTB, WM = TL::WM, WN = TL::WN, A_U4 = TL::a_u4(TAPS);
    extern __shared__ __attribute__((aligned(16))) uint4 smem_s[];
    uint4* As = smem_s;
    uint4* Xs = smem_s + A_U4;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lh = lane >> 5;
    const int wm = wave / TL::NWV;
    const int mblocks = a.MT / MTB;
    int len = a.len;                     // RAGT: the current tile's utterance (set by coords); otherwise the launch's
    const int rs = RAGT ? a.len : 0;     // RAGT: row stride of x / cond / y (0 = `len`, the helpers' default)
    int coloff = 0;                      // RAGT: first column of the current tile's utterance
    const int ntiles = a.ntiles;

    // persistent: a workgroup walks a CONTIGUOUS range of tiles; the first slab of every phase - including the
    // first phase of the NEXT tile - is loaded behind the last slab of the one before it, so only the very first
    // load of a workgroup is cold and the output stores of a tile overlap the next tile's input loads.
    // Tile order: column tiles in utterance order, the `mblocks` row blocks of one column tile next to each other - they re-read
    // the same input tile, now from the same workgroup's L2 a few microseconds apart - and a workgroup meets one or two
    // utterances (the |max| slot of the output is published once per utterance and workgroup, see amax_flush_wg).
    auto coords = [&](int v, int& mt0, int& b, int& t0) __attribute__((always_inline)) {
        const int nt_id = v / mblocks, mb = v - nt_id * mblocks;
        mt0 = mb * MTB;
        if constexpr (RAGT) {
            b = __builtin_amdgcn_readfirstlane(rag_find(a.rag.ts, a.rag.B, __builtin_amdgcn_readfirstlane(nt_id), b));      // (v / mblocks ran on the vector ALU)
            t0 = __builtin_amdgcn_readfirstlane((nt_id - a.rag.ts[b]) * TL::BN);
        } else {
            b = nt_id / a.tiles_per_utt;
            t0 = (nt_id - b * a.tiles_per_utt) * TL::BN;
        }
    };
    // RAGT: utterance b's length and first column at this launch's rate
    auto ulen = [&](int b_) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(a.rag.tb[b_] * a.rag.mult); };
    auto uoff = [&](int b_) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(a.rag.pre[b_] * a.rag.mult); };
    int tile, vtiles;
    tile_range(ntiles, tile, vtiles);
    SlabRegs<TL> regs;
#ifdef S_TRACE
    __shared__ unsigned long long tr_lds[256];
    unsigned long long t0_mem = 0, t0_real = 0;
    if (blockIdx.x == S_TRACE_WG && threadIdx.x == S_TRACE_TID) {
        regs.tr = tr_lds;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_mem), "=s"(t0_real)::"memory");
    }
#endif
    int mt0 = 0, b = 0, t0 = 0;
    if (tile >= vtiles) return;
    coords(tile, mt0, b, t0);
    float mx_run = 0.f;                  // this wave's |max| of what it stored for utterance mx_b
    int mx_b = b;
    float* const red = reinterpret_cast<float*>(smem_s) + TL::bias_off(TAPS) + 12 * TL::BM + TL::KS_MAX;
    const int fT = (TAPS == 1 && !FILM && !LERP && !RAGT) ? a.flatT : 0;      // flat column tiles are a property of the plain-GEMM launches (conv3s_launch_t checks): the convs compile without that path
    const unsigned fstride = (unsigned)a.xstride;
    // narrow FiLM tiles run their FiLM phase FIRST (scale and shift pairs: 4 accumulator sets, folded to 2 before the conv's pair comes
    // alive: 4 sets at the peak instead of 5, which did not fit the 168 registers of a 12-wave workgroup)
    constexpr bool FILM_FIRST = FILM && TL::WN == 1;
    auto load_first = [&](int mt0_, int b_, int t0_) __attribute__((always_inline)) {
        if constexpr (FILM_FIRST && RAGT) film_first_load<TL>(regs, a.sc6, 2 * a.MT, mt0_, a.MT, a.cond + uoff(b_), ulen(b_), t0_, rs);
        else if constexpr (FILM_FIRST) film_first_load<TL>(regs, a.sc6, 2 * a.MT, mt0_, a.MT, a.cond + (long)b_ * a.Ccond * len, len, t0_);
        else if constexpr (RAGT) first_load<TL, TAPS, LERP, CLAMP>(regs, a.A6, a.MT, mt0_, a.x + uoff(b_), a.Cin, ulen(b_), a.dil, t0_, 0, 0u, a.cmax, 0, 0.f, rs);
        else first_load<TL, TAPS, LERP, CLAMP>(regs, a.A6, a.MT, mt0_, fT ? a.x : a.x + (long)b_ * a.xstride, a.Cin, len, a.dil, t0_, fT, fstride, a.cmax, a.lin, a.lscale);
    };
    if constexpr (RAGT) {
        len = ulen(b);
        coloff = uoff(b);
    }
    load_first(mt0, b, t0);
    for (int tile_no = 0; tile < vtiles; ++tile_no) {
        const int nxt = tile + 1;
        if constexpr (!Epi::kIgemm) {
            if (a.amax_y && b != mx_b) {          // the workgroup moves on to another utterance (uniform)
                amax_flush_wg(a.amax_y + mx_b, mx_run, red);
                mx_run = 0.f;
                mx_b = b;
            }
        }
        auto load_next_tile = [&]() __attribute__((always_inline)) {
            if (nxt < vtiles) {
                int mt0n, bn = b, t0n;
                coords(nxt, mt0n, bn, t0n);
                load_first(mt0n, bn, t0n);
            }
        };
        const float* xb = RAGT ? a.x + coloff : (fT ? a.x : a.x + (long)b * a.xstride);
        // this tile's bias rows -> LDS: read back with ds_read (lgkmcnt), so the epilogue math never waits on vmcnt
        // while the next phase's prefetch is in flight (a global bias load would drag that whole prefetch with it)
        // (two copies by tile parity: without the LDS park no barrier separates a fast wave's next tile from a slow wave's epilogue)
        float* Bs = reinterpret_cast<float*>(smem_s) + TL::bias_off(TAPS) + (tile_no & 1) * 6 * TL::BM;
        if constexpr (!Epi::kIgemm) {
            for (int i = threadIdx.x; i < TL::BM; i += TL::NTHR) {
                int m = mt0 * 32 + i;
                m = m < ep.M ? m : ep.M - 1;
                Bs[i] = ep.bias[m];
                Bs[3 * TL::BM + i] = a.wsc[mt0 + (i >> 5)];               // the weight image's power-of-two scale of this row's m-tile
                if constexpr (FILM) {
                    Bs[TL::BM + i] = ep.bsc[m];
                    Bs[2 * TL::BM + i] = ep.bsh[m];
                    Bs[4 * TL::BM + i] = a.fsc[mt0 + (i >> 5)];
                    Bs[5 * TL::BM + i] = a.fsc[a.MT + mt0 + (i >> 5)];
                }
            }
        }
        float* Ks = reinterpret_cast<float*>(smem_s) + TL::bias_off(TAPS) + 12 * TL::BM;
        if constexpr (SCALED) {
            // first use is behind the first slab's barrier; the previous tile's last use is behind its last one
            const int b0 = fT ? t0 / fT : b;                      // flat tiles: the (at most two) utterances this tile touches
            const int nrow = fT ? ((t0 + TL::BN - 1 < len ? t0 + TL::BN - 1 : len - 1) / fT - b0 + 1) * a.Cin : a.Cin;
            for (int i = threadIdx.x; i < nrow; i += TL::NTHR) Ks[i] = a.kscale[(long)b0 * a.Cin + i];
        }
        const int rl = wm * WM * 32 + 4 * lh;                     // local row of accumulator register r of m-tile i: rl + 32 i + (r & 3) + 8 (r >> 2)
        // block-floating-point scales of this tile's inputs (identity unless an utterance's |max| leaves the fp16 window)
        const Bfp sx = (fT || RAGG) ? Bfp{1.f, 1.f} : bfp_load_u(a.amax_x, b);      // flat GEMM tiles: per column (make_map / the epilogue below)
        f32x16 hi[WM][WN], lo[WM][WN];
        auto clear = [&](f32x16 (&u)[WM][WN]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) u[i][j][r] = 0.f;
        };
        clear(hi);
        clear(lo);
        // hi <- (hi + 2^-11 lo) * wscale(row) * inv  (the conv result without its bias); wtab = the per-row weight-scale table
        // (the row tables are read four rows at a time behind a compiler fence: left alone, the scheduler hoists every table read of the
        // epilogue - 64 values per lane on the FiLM tiles - above the arithmetic and spills them)
        auto fold = [&](f32x16 (&h)[WM][WN], const f32x16 (&l)[WM][WN], const float* wtab, float inv) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w4 = *reinterpret_cast<const float4*>(wtab + rl + i * 32 + 8 * q);
                    const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float c = wv[u] * inv, cl = c * kLoInv;
#pragma unroll
                        for (int j = 0; j < WN; ++j) h[i][j][4 * q + u] = comb(h[i][j][4 * q + u], l[i][j][4 * q + u], c, cl);
                    }
#pragma unroll
                    for (int j = 0; j < WN; ++j) asm volatile("" : "+v"(h[i][j]) : : "memory");
                }
        };

        if constexpr (FILM && TL::WN > 1) {
            // Wide tiles (two n-tiles per wave): the accumulator sets of conv, scale and shift do not fit the register budget together, so
            // scale and shift are the two halves of ONE slab loop over the cond tile (split_phase<TWO>) sharing one extra pair:
            // out = ((h + b)(sc + b_sc)) + (sh + b_sh) in the same order as below, (h + b)(sc + b_sc) replaces h between the halves.  The cond
            // tile is staged twice; in exchange the conv phase (3/5 of the MFMAs) runs on the 96 x 256 tile of the plain convs.
            const float* cb = a.cond + (long)b * a.Ccond * len;
            const int mtoff = a.MT;
            const Bfp sc = bfp_load_u(a.amax_c, b);
            split_phase<TL, TAPS, A_U4, LRELU, S_FB_FW, false, LERP, false>(      // only one accumulator pair is live here: two fragment sets fit
                hi, lo, regs, a.A6, a.MT, mt0, xb, a.Cin, len, a.dil, t0, As, Xs,
                [&]() __attribute__((always_inline)) { first_load<TL, 1>(regs, a.sc6, 2 * a.MT, mt0, cb, a.Ccond, len, 0, t0); }, sx.s, nullptr, 0, 0u, 0, a.lin, a.lscale);
            fold(hi, lo, Bs + 3 * TL::BM, sx.inv);
            f32x16 fh[WM][WN];
            clear(fh);
            clear(lo);
            split_phase<TL, 1, A_U4, false, S_FB_F, false, false, false, true>(
                fh, lo, regs, a.sc6, 2 * a.MT, mt0, cb, a.Ccond, len, 0, t0, As, Xs, load_next_tile, sc.s, nullptr, 0, 0u, 0, 0, 0.f, mtoff + mt0,
                [&]() __attribute__((always_inline)) {
                    fold(fh, lo, Bs + 4 * TL::BM, sc.inv);
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = rl + i * 32 + (r & 3) + 8 * (r >> 2);
                            const float bm = Bs[row], bs = Bs[TL::BM + row];
#pragma unroll
                            for (int j = 0; j < WN; ++j) {
                                hi[i][j][r] = __fmul_rn(hi[i][j][r] + bm, fh[i][j][r] + bs);
                                fh[i][j][r] = 0.f;
                                lo[i][j][r] = 0.f;
                            }
                        }
                });
            fold(fh, lo, Bs + 5 * TL::BM, sc.inv);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float bh = Bs[2 * TL::BM + rl + i * 32 + (r & 3) + 8 * (r >> 2)];
#pragma unroll
                    for (int j = 0; j < WN; ++j) hi[i][j][r] = __fadd_rn(hi[i][j][r], fh[i][j][r] + bh);
                }
            tile_store<TL, true>(reinterpret_cast<float*>(smem_s), hi, ep.y, ep.res, b, ep.M, len, mt0, t0, mx_run, nullptr, 0, ep.res_lin, ep.res_scale);
        } else if constexpr (FILM) {
            // conv -> FiLM -> + residual (decoder.py:94-97,181-182): scale and shift come from one more 1x1 phase over
            // the cond tile on the same output tiles; (conv, scale, shift) combine in registers.
            const float* cb = RAGT ? a.cond + coloff : a.cond + (long)b * a.Ccond * len;
            const int mtoff = a.MT;                                   // the stacked FiLM image has 2 * a.MT m-tiles: scale rows, then shift rows
            const Bfp sc = bfp_load_u(a.amax_c, b);
            f32x16 asc[WM][WN], lsc[WM][WN], ash[WM][WN], lsh[WM][WN];
            clear(asc);
            clear(lsc);
            clear(ash);
            clear(lsh);
            film_phase<TL>(asc, lsc, ash, lsh, regs, a.sc6, 2 * a.MT, mt0, mtoff, cb, a.Ccond, len, t0, As, Xs,
                           [&]() __attribute__((always_inline)) { first_load<TL, TAPS, LERP, false>(regs, a.A6, a.MT, mt0, xb, a.Cin, len, a.dil, t0, 0, 0u, 0, a.lin, a.lscale, rs); }, sc.s, rs);
            fold(asc, lsc, Bs + 4 * TL::BM, sc.inv);            // scale and shift without their biases: frees two sets before the conv phase
            fold(ash, lsh, Bs + 5 * TL::BM, sc.inv);
            split_phase<TL, TAPS, A_U4, LRELU, S_FB_F, false, LERP, false>(hi, lo, regs, a.A6, a.MT, mt0, xb, a.Cin, len, a.dil, t0, As, Xs, load_next_tile, sx.s, nullptr, 0, 0u, 0,
                                                                            a.lin, a.lscale, 0, NoMid(), nullptr, rs);
            fold(hi, lo, Bs + 3 * TL::BM, sx.inv);
            // out = ((h + b)(sc + b_sc)) + (sh + b_sh), + residual in the store pass
            {
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = rl + i * 32 + 8 * q;
                        const float4 m4 = *reinterpret_cast<const float4*>(Bs + row), s4 = *reinterpret_cast<const float4*>(Bs + TL::BM + row),
                                     h4 = *reinterpret_cast<const float4*>(Bs + 2 * TL::BM + row);
                        const float bm[4] = {m4.x, m4.y, m4.z, m4.w}, bs[4] = {s4.x, s4.y, s4.z, s4.w}, bh[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int j = 0; j < WN; ++j)
                                hi[i][j][4 * q + u] = __fadd_rn(__fmul_rn(hi[i][j][4 * q + u] + bm[u], asc[i][j][4 * q + u] + bs[u]), ash[i][j][4 * q + u] + bh[u]);
#pragma unroll
                        for (int j = 0; j < WN; ++j) asm volatile("" : "+v"(hi[i][j]) : : "memory");
                    }
                tile_store<TL, true>(reinterpret_cast<float*>(smem_s), hi, ep.y, ep.res, RAGT ? 0 : b, ep.M, len, mt0, t0, mx_run, nullptr, 0, ep.res_lin, ep.res_scale, rs, coloff);
            }
        } else {
            float inv = sx.inv;
            if constexpr (wants_res_conv<Epi>::value) {
                // Downsample (decoder.py:148-157): out = c3(lrelu(h2)) + down_res(xi).  Both are linear into the same output tile:
                // the 1x1 runs as a second K phase over xi (a.cond, a.Ccond channels, image a.sc6) on the same accumulators, so
                // the residual tensor is never written or read back and its launch disappears (ep.bias = the two biases summed).
                // One accumulator pair, one unit: the two images are packed with JOINT per-m-tile scales (api.hip) and the two
                // inputs share the smaller of their block-floating-point scales.
                const float* cb = RAGT ? a.cond + coloff : a.cond + (long)b * a.Ccond * len;
                const Bfp s2 = bfp_min(sx, bfp_load_u(a.amax_c, b));
                inv = s2.inv;
                split_phase<TL, TAPS, A_U4, LRELU, S_FB, false, false, false>(
                    hi, lo, regs, a.A6, a.MT, mt0, xb, a.Cin, len, a.dil, t0, As, Xs,
                    [&]() __attribute__((always_inline)) { first_load<TL, 1>(regs, a.sc6, a.MT, mt0, cb, a.Ccond, len, 0, t0, 0, 0u, 0, 0, 0.f, rs); }, s2.s, nullptr, 0, 0u, 0, 0, 0.f,
                    0, NoMid(), nullptr, rs);
                split_phase<TL, 1, A_U4, false, S_FB, false, false, false>(hi, lo, regs, a.sc6, a.MT, mt0, cb, a.Ccond, len, 0, t0, As, Xs, load_next_tile, s2.s, nullptr, 0, 0u, 0, 0,
                                                                           0.f, 0, NoMid(), nullptr, rs);
            } else
            split_phase<TL, TAPS, A_U4, LRELU, Epi::kIgemm ? S_FB_G : S_FB, SCALED, LERP, CLAMP>(hi, lo, regs, a.A6, a.MT, mt0, xb, a.Cin, len, a.dil, t0, As, Xs,
                                                                                                  load_next_tile, sx.s, Ks, fT, fstride, a.cmax, a.lin, a.lscale, 0, NoMid(),
                                                                                                  a.amax_x, rs, RAGG ? a.rag.col2b : nullptr, RAGG ? a.rag.mult : 1);
            if constexpr (Epi::kIgemm) {
                // plain GEMM use (B = 1, len = all columns): the gemm_epi.h epilogue functors finish the element
                const int l31 = lane & 31, wn = wave - wm * TL::NWV;
                const int mtw = __builtin_amdgcn_readfirstlane(mt0 + wm * WM);
#pragma unroll
                for (int i = 0; i < WM; ++i) {
                    const float cw = a.wsc[mtw + i];
#pragma unroll
                    for (int j = 0; j < WN; ++j) {
                        const int col = t0 + (wn * WN + j) * 32 + l31;
                        const int n = b * len + col;
                        if (col < len) {
                            const float c = cw * (RAGG ? bfp_load(a.amax_x, a.rag.col2b[col / a.rag.mult]).inv : (fT ? bfp_load(a.amax_x, col / fT).inv : inv)), cl = c * kLoInv;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float v[4] = {comb(hi[i][j][4 * q], lo[i][j][4 * q], c, cl), comb(hi[i][j][4 * q + 1], lo[i][j][4 * q + 1], c, cl),
                                                    comb(hi[i][j][4 * q + 2], lo[i][j][4 * q + 2], c, cl), comb(hi[i][j][4 * q + 3], lo[i][j][4 * q + 3], c, cl)};
                                ep.store(n, (mt0 + wm * WM + i) * 32 + 8 * q + 4 * lh, v);
                            }
                        }
                    }
                }
                tile = nxt;
                if (tile < vtiles) coords(tile, mt0, b, t0);
                continue;
            }
            if constexpr (!Epi::kIgemm) {
                fold(hi, lo, Bs + 3 * TL::BM, inv);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float bm = Bs[rl + i * 32 + (r & 3) + 8 * (r >> 2)];
#pragma unroll
                        for (int j = 0; j < WN; ++j) hi[i][j][r] += bm;
                    }
                tile_store<TL, Epi::kRes>(reinterpret_cast<float*>(smem_s), hi, ep.y, ep.res, RAGT ? 0 : b, ep.M, len, mt0, t0, mx_run, ep.y2, ep.f2, 0, 0.f, rs, coloff);
            }
        }
        TR_STAMP(regs, 7);
        tile = nxt;
        if (tile < vtiles) {
            coords(tile, mt0, b, t0);
            if constexpr (RAGT) {
                len = ulen(b);
                coloff = uoff(b);
            }
        }
    }
    if constexpr (!Epi::kIgemm) {
        if (a.amax_y) amax_flush_wg(a.amax_y + mx_b, mx_run, red);
    }
#ifdef S_TRACE
    if (regs.tr) {
        const unsigned slot = atomicAdd(&g_trace_slot, 1u) & 63u;
        unsigned long long* g = g_trace + slot * 256;
        g[0] = 0x5452414345000000ull | ((unsigned long long)TL::MTB << 20) | ((unsigned long long)TL::KG << 16) | ((unsigned long long)TAPS << 12) | ((unsigned long long)SCALED << 8) | (unsigned long long)FILM;
        g[1] = ((unsigned long long)a.Cin << 32) | (unsigned)regs.trn;
        g[2] = ((unsigned long long)gridDim.x << 32) | (unsigned)ntiles;
        unsigned long long rt, mt;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(mt), "=s"(rt)::"memory");
        g[3] = ((mt - t0_mem) << 32) | ((rt - t0_real) & 0xffffffffull);     // kernel-long deltas: s_memtime ticks | 100 MHz ticks
        for (int i = 0; i < regs.trn; ++i) g[4 + i] = regs.tr[i];
    }
#endif
}

// per-utterance |max| slots of a launch's tensors (block-floating-point guard, see the top of this file): x / cond are read
// (nullptr = the tensor is known to sit inside the fp16 window: no scaling), y is written (nullptr = nobody needs it)
struct BfpSlots {
    const float* x = nullptr;
    const float* c = nullptr;
    float* y = nullptr;
};

template <class TL, int TAPS, bool LRELU, class Epi, bool FILM, bool SCALED = false, bool LERP = false, bool CLAMP = false>
inline int conv3s_launch_t(tvc_ctx* ctx, hipStream_t s, const PackedW& w, const float* x, int B, int Cin, int len, int dil, const Epi& ep,
                           const PackedW* wsc, const PackedW* wsh, const float* cond, int Ccond, long xstride = 0, int bpc = S_BPC,
                           const float* kscale = nullptr, bool flat = false, int cmax = 0, int lin = 0, float lscale = 0.f, const BfpSlots& bfp = {}) {
    if (Cin % (16 * TL::KG) != 0 || (FILM && Ccond % (16 * TL::KG) != 0)) return fail(ctx, TVC_ERR_ARG, "conv3s: channel counts must be multiples of the slab depth");
    if (Cin / 16 > w.S6) return fail(ctx, TVC_ERR_ARG, "conv3s: weight image has fewer K16 steps than the launch walks");
    static bool ready_dev[64] = {};                 // the attribute is per (function, device): one flag per device of this process
    bool& ready = ready_dev[ctx->device & 63];
    constexpr int lds = TL::lds_bytes(TAPS);
    if (!ready) {
        hipError_t e = hipFuncSetAttribute((const void*)conv3s_kernel<TL, TAPS, LRELU, Epi, FILM, SCALED, LERP, CLAMP>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return fail(ctx, TVC_ERR_HIP, "conv3s setup: %s", hipGetErrorString(e));
        ready = true;
    }
    ConvSArgs a;
    a.A6 = reinterpret_cast<const uint4*>(w.A6);
    a.wsc = w.wscale;
    a.MT = w.MT6;
    a.amax_x = bfp.x;
    a.amax_c = bfp.c;
    a.amax_y = bfp.y;
    if (Epi::kIgemm && bfp.y) return fail(ctx, TVC_ERR_ARG, "conv3s: GEMM launches finish their elements in the epilogue functor and cannot track the output's |max|");
    if (SCALED && (!kscale || Cin > 768)) return fail(ctx, TVC_ERR_ARG, "conv3s: scaled launch needs factors for <= 768 channels");
    a.kscale = kscale;
    a.cmax = cmax;
    if (CLAMP != (cmax > 0)) return fail(ctx, TVC_ERR_ARG, "conv3s: row clamp is a compile-time property of the launch");
    a.lin = lin;
    a.lscale = lscale;
    if (LERP && (lin <= 0 || flat)) return fail(ctx, TVC_ERR_ARG, "conv3s: interpolated input needs its low-rate length");
    a.x = x;
    a.xstride = xstride ? xstride : (long)Cin * (LERP ? lin : len);
    a.Cin = Cin;
    a.len = len;
    a.dil = dil;
    if (flat) {   // GEMM over the flattened B * len columns: no partial tile per utterance
        if (FILM || TAPS != 1 || dil != 0) return fail(ctx, TVC_ERR_ARG, "conv3s: flat tiles are for plain GEMMs");
        a.flatT = len;
        a.len = B * len;
        B = 1;
    }
    a.tiles_per_utt = (a.len + TL::BN - 1) / TL::BN;
    if (wants_res_conv<Epi>::value) {
        if (FILM || !wsc || wsc->MT6 != w.MT6 || wsc->taps != 1 || Ccond % 16 != 0 || Ccond / 16 > wsc->S6 || !cond)
            return fail(ctx, TVC_ERR_ARG, "conv3s: the residual 1x1 phase needs a matching 1x1 image and its input");
        if (wsc->wjoint != w.A6) return fail(ctx, TVC_ERR_STATE, "conv3s: the residual 1x1's image must be packed with the conv's per-m-tile scales");
        a.sc6 = reinterpret_cast<const uint4*>(wsc->A6);
        a.cond = cond;
        a.Ccond = Ccond;
    }
    if (FILM) {
        if (wsc->MT6 != 2 * w.MT6) return fail(ctx, TVC_ERR_ARG, "conv3s: FiLM image must stack scale and shift rows");
        a.sc6 = reinterpret_cast<const uint4*>(wsc->A6);   // stacked [to_scale ; to_shift] image (wsh unused)
        a.fsc = wsc->wscale;
        a.cond = cond;
        a.Ccond = Ccond;
    }
    a.ntiles = (a.MT / TL::MTB) * a.tiles_per_utt * B;
    static int ncu_dev[64] = {};
    int& ncu = ncu_dev[ctx->device & 63];
    if (!ncu) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, ctx->device) != hipSuccess) return fail(ctx, TVC_ERR_HIP, "conv3s: device properties");
        ncu = prop.multiProcessorCount;
    }
    const int slots = ncu * bpc;        // persistent: one resident workgroup per slot walks a contiguous range of the tiles
    if (ctx->rag) {
        // ragged batch (ragged.h): the driver passed B = 1 and len = the batch's columns at this rate (= the row stride)
        if constexpr (!LERP && !SCALED && !(FILM && TL::WN > 1)) {
            if (B != 1 || flat || a.len % ctx->rag->Ttot != 0) return fail(ctx, TVC_ERR_STATE, "conv3s: a ragged batch runs as one long utterance");
            static bool ready_rag[64] = {};
            bool& rr = ready_rag[ctx->device & 63];
            if (!rr) {
                hipError_t e = hipFuncSetAttribute((const void*)conv3s_kernel<TL, TAPS, LRELU, Epi, FILM, SCALED, LERP, CLAMP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                if (e != hipSuccess) return fail(ctx, TVC_ERR_HIP, "conv3s setup: %s", hipGetErrorString(e));
                rr = true;
            }
            int ncol = a.tiles_per_utt;
            TVC_CHECK(rag_view(ctx, s, a.len / ctx->rag->Ttot, Epi::kIgemm ? 0 : TL::BN, &a.rag, Epi::kIgemm ? nullptr : &ncol));
            a.ntiles = (a.MT / TL::MTB) * ncol;
            dim3 g((unsigned)(a.ntiles < slots ? a.ntiles : slots));
            hipLaunchKernelGGL((conv3s_kernel<TL, TAPS, LRELU, Epi, FILM, SCALED, LERP, CLAMP, true>), g, dim3(TL::NTHR), lds, s, a, ep);
            return 0;
        } else {
            return fail(ctx, TVC_ERR_STATE, "conv3s: this launch has no ragged-batch variant");
        }
    }
    dim3 g((unsigned)(a.ntiles < slots ? a.ntiles : slots));
    hipLaunchKernelGGL((conv3s_kernel<TL, TAPS, LRELU, Epi, FILM, SCALED, LERP, CLAMP>), g, dim3(TL::NTHR), lds, s, a, ep);
    return 0;
}

// k3 conv on the split path; Mpad = 64 (48 channels) or a multiple of 96 (FilterNet levels with C = 96, 192, 384).
// Tiles (each swept on the part): plain convs 96 x 256 (a wave owns two 32-sample n-tiles: every weight
// fragment and every slab's staging round trip serves twice the columns), 48-output-channel convs 64 x 192 (six waves per m-tile),
// FiLM-fused convs 96 x 128 or 96 x 256.
template <bool LRELU, class Epi, bool FILM = false, bool LERP = false>
inline int conv3s_launch(tvc_ctx* ctx, hipStream_t s, const PackedW& w, const float* x, int B, int Cin, int len, int dil, const Epi& ep,
                         const BfpSlots& bfp, const PackedW* wsc = nullptr, const PackedW* wsh = nullptr, const float* cond = nullptr, int Ccond = 0, int lin = 0,
                         float lscale = 0.f) {
    if (w.MT6 == 2)     // 48 output channels: two m-tiles, the second half empty
        return conv3s_launch_t<SplitTile<2, 1, 6, 1>, 3, LRELU, Epi, FILM, false, LERP>(ctx, s, w, x, B, Cin, len, dil, ep, wsc, wsh, cond, Ccond, 0, S_BPC, nullptr, false, 0,
                                                                                       lin, lscale, bfp);
    if constexpr (FILM) {
        // Two tiles for the FiLM-fused launches.  96 x 256 (the plain convs' tile): the conv phase stages a slab for twice the columns,
        // scale and shift run one after the other on one extra accumulator pair (split_phase<TWO>) and the cond tile is staged twice.
        // 96 x 128: FiLM's scale and shift in one phase (four accumulator sets, then the conv's pair); its register budget is tight
        // (measured slower per column with the accumulator pairs of the fp16 split), so it only serves launches that cannot fill
        // the chip with wide tiles (a streaming block).
        const long mb = w.MT6 / 3;
        const long tiles_w = mb * ((len + 255) / 256) * B;
        const long slots = 256 * S_BPC;
        const bool wide = tiles_w >= slots && !ctx->rag;      // (a ragged batch runs the narrow tile: its utterances are shorter than one wide tile)
        if (wide)
            return conv3s_launch_t<SplitTile<3, 1, 4, 2>, 3, LRELU, Epi, FILM, false, LERP>(ctx, s, w, x, B, Cin, len, dil, ep, wsc, wsh, cond, Ccond, 0, S_BPC, nullptr, false,
                                                                                           0, lin, lscale, bfp);
        // 32-channel slabs (KG = 2): the narrow tile serves launches that cannot fill the chip - a streaming block, one short utterance -, where a
        // workgroup's time is its chain of load -> LDS -> barrier -> MFMA round trips: half as many, same K order (bit-identical); 32-stream block
        // p50 1.66 -> 1.62 ms same-box, 138 registers, no scratch
        if (Cin % 32 == 0 && Ccond % 32 == 0)
            return conv3s_launch_t<SplitTile<3, 1, 4, 1, 2>, 3, LRELU, Epi, FILM, false, LERP>(ctx, s, w, x, B, Cin, len, dil, ep, wsc, wsh, cond, Ccond, 0, S_BPC, nullptr, false, 0,
                                                                                              lin, lscale, bfp);
        return conv3s_launch_t<SplitTile<3, 1, 4, 1>, 3, LRELU, Epi, FILM, false, LERP>(ctx, s, w, x, B, Cin, len, dil, ep, wsc, wsh, cond, Ccond, 0, S_BPC, nullptr, false, 0,
                                                                                       lin, lscale, bfp);
    } else      // (32-channel slabs on this tile too: measured neutral for the batch step and for the streaming block, and the interpolating instantiation spills at them: not kept)
        return conv3s_launch_t<SplitTile<3, 1, 4, 2>, 3, LRELU, Epi, FILM, false, LERP>(ctx, s, w, x, B, Cin, len, dil, ep, wsc, wsh, cond, Ccond, 0, S_BPC, nullptr, false, 0,
                                                                                       lin, lscale, bfp);
}

// Plain GEMM on the split path: out(m, n) = sum_k W[m][k] x[b][k][t], n = b * len + t, finished by a gemm_epi.h epilogue
// functor (store(n, m, v[4])).  Cin must be a multiple of 16 and rows [K, Cin) must be readable (weights there are 0).
template <int MTB, int NWV, int BPC, int KG, class Epi, bool SCALED, bool CLAMP>
inline int gemm_s_launch_k(tvc_ctx* ctx, hipStream_t s, const PackedW& w, const float* x, int B, int Cin, int len, long xstride, const Epi& ep,
                           const float* kscale, int cmax, const float* amax_x) {
    using TL = SplitTile<MTB, (MTB >= 4 ? 2 : 1), NWV, 1, KG, 0>;     // MTB >= 4: a wave owns two m-tiles (64 x 32), 8 waves cover 128 x 128
    // flat column tiles unless a tile could touch more than two utterances of a SCALED launch (factors of two are staged)
    // or the element offsets would not fit 32 bits
    const long xs = xstride ? xstride : (long)Cin * len;
    const bool flat = B > 1 && xs * B < (1L << 30) && (!SCALED || len >= TL::BN);
    return conv3s_launch_t<TL, 1, false, Epi, false, SCALED, false, CLAMP>(ctx, s, w, x, B, Cin, len, 0, ep, nullptr, nullptr, nullptr, 0, xstride, BPC, kscale, flat,
                                                                           cmax, 0, 0.f, BfpSlots{amax_x, nullptr, nullptr});
}
template <int MTB, int NWV, int BPC, class Epi, bool SCALED = false>
inline int gemm_s_launch(tvc_ctx* ctx, hipStream_t s, const PackedW& w, const float* x, int B, int Cin, int len, long xstride, const Epi& ep,
                         const float* amax_x, const float* kscale = nullptr) {
    if (w.MT6 % MTB != 0) return fail(ctx, TVC_ERR_ARG, "gemm_s: row tiles do not divide");
    // deepest slab the channel count allows: 32 or 16 input channels per load -> LDS -> barrier round trip
    if (TVC_S_KG >= 2 && Cin % 32 == 0) return gemm_s_launch_k<MTB, NWV, BPC, 2, Epi, SCALED, false>(ctx, s, w, x, B, Cin, len, xstride, ep, kscale, 0, amax_x);
    return gemm_s_launch_k<MTB, NWV, BPC, 1, Epi, SCALED, false>(ctx, s, w, x, B, Cin, len, xstride, ep, kscale, 0, amax_x);
}
// The input has only `krows` channel rows per utterance (not a multiple of the slab depth): K is rounded up to whole 32-channel
// slabs and the loads of the missing rows are clamped to the last real one (their weights are zero).  Its own instantiation:
// a second load path inside the shared kernels cost them 4 % (waitcnt placement), measured.
template <int MTB, int NWV, int BPC, class Epi>
inline int gemm_s_launch_ragged(tvc_ctx* ctx, hipStream_t s, const PackedW& w, const float* x, int B, int krows, int len, long xstride, const Epi& ep,
                                const float* amax_x) {
    if (w.MT6 % MTB != 0) return fail(ctx, TVC_ERR_ARG, "gemm_s: row tiles do not divide");
    return gemm_s_launch_k<MTB, NWV, BPC, 2, Epi, false, true>(ctx, s, w, x, B, (krows + 31) / 32 * 32, len, xstride, ep, nullptr, krows - 1, amax_x);
}

// Per-utterance |max| of a [B][C][len] tensor into slot[b] (zeroed first): the block-floating-point slot of a tensor whose
// producer does not track it (tensors that enter a stage through the C ABI, epilogue-functor outputs).  One pass over the tensor.
static __global__ __launch_bounds__(256) void amax_rows_kernel(const float* __restrict__ x, long n, float* __restrict__ slot) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const float* p = x + (long)b * n;
    float mx = 0.f;
    if ((n & 3) == 0) {                      // rows are whole float4s (and 16-byte aligned: workspace tensors are 256-byte aligned)
        for (long i = blockIdx.x * 256L + threadIdx.x; i < (n >> 2); i += (long)gridDim.x * 256) {
            const float4 v = reinterpret_cast<const float4*>(p)[i];
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    } else {
        for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) mx = fmaxf(mx, fabsf(p[i]));
    }
    amax_flush_wg(slot + b, mx, red);
}
// ragged batch (ragged.h): x is [C][rs] over the whole batch; utterance blockIdx.y owns columns [pre * mult, (pre + tb) * mult) of every row
static __global__ __launch_bounds__(256) void amax_rag_kernel(const float* __restrict__ x, int C, int rs, RagDev rg, float* __restrict__ slot) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const int n = rg.tb[b] * rg.mult;
    const float* p = x + (long)rg.pre[b] * rg.mult;
    float mx = 0.f;
    for (int c = blockIdx.x; c < C; c += gridDim.x) {
        const float* r = p + (long)c * rs;
        for (int i = threadIdx.x; i < n; i += 256) mx = fmaxf(mx, fabsf(r[i]));
    }
    amax_flush_wg(slot + b, mx, red);
}
// slot must have been zeroed (one memset per stage covers all of a stage's slots); rows of C * len floats must be 16-byte aligned
inline int run_amax_rows(tvc_ctx* ctx, hipStream_t s, const float* x, int B, int C, long len, float* slot) {
    if (ctx->rag) {     // (the driver passed B = 1 and len = the batch's columns at this tensor's rate)
        if (B != 1 || len % ctx->rag->Ttot != 0) return fail(ctx, TVC_ERR_STATE, "amax_rows: a ragged batch runs as one long utterance");
        RagDev rg;
        TVC_CHECK(rag_view(ctx, s, (int)(len / ctx->rag->Ttot), 0, &rg, nullptr));
        const int gx = C < 8 ? C : (ctx->rag->B >= 64 ? 8 : 16);
        hipLaunchKernelGGL(amax_rag_kernel, dim3((unsigned)gx, (unsigned)ctx->rag->B), dim3(256), 0, s, x, C, (int)len, rg, slot);
        return launch_check(ctx, "amax_rows (ragged)");
    }
    const long n = (long)C * len;
    if ((reinterpret_cast<uintptr_t>(x) & 15) != 0) return fail(ctx, TVC_ERR_ARG, "amax_rows: the tensor must be 16-byte aligned");
    // one atomic per workgroup: a few per utterance while the launch still fills the chip
    long gx = (n / 4 + 255) / 256 / 8;                     // >= 8 float4 per thread
    const long cap = B >= 64 ? 8 : 512 / (B > 0 ? B : 1);
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(amax_rows_kernel, dim3((unsigned)gx, (unsigned)B), dim3(256), 0, s, x, n, slot);
    return launch_check(ctx, "amax_rows");
}

}  // namespace tvc
