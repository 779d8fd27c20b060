// Source-filter decoder (decoder.py:24-266): SourceNet, additive harmonic oscillator, filtered-noise
// iSTFT, and the FilterNet U-Net.
#include "conv3s.h"
#include "conv_s2.h"
#include "film_s2.h"
#include "gemm_s2.h"
#include "small_kernels.h"
#include "tvc_common.h"

namespace tvc {

// =================================================================================================
// Harmonic oscillator (decoder.py:24-54).  The phase of harmonic m is the running sum over the whole
// utterance of fl32(fl32(fs*m)/24000), accumulated in fp64 and rounded to fp32 per sample — what
// ATen's CPU cumsum does for fp32 input.  Hierarchical scan: per-frame fp64 sums, an exclusive scan
// of those per (utterance, harmonic), then an in-frame scan fused with sin / voiced gate / amplitude.
// =================================================================================================

// fl32(x / 24000), the oscillator's per-sample phase increment: IEEE division costs a dozen vector instructions (v_div_scale x 2, v_rcp,
// four fmas, v_div_fmas, v_div_fixup) and there are 2 x 15 of them per output sample.  Multiply by the rounded reciprocal, take the exact
// remainder with one fma, correct with another: by Markstein's theorem the result is the correctly rounded quotient whenever the
// divisor's significand is not all ones; tools/micro/div24k.hip compares it with __fdiv_rn on every positive float from 2^-100 to 2^24
// (1 048 576 000 values): identical.
__device__ __forceinline__ float div24k(float x) {
    const float r = 1.0f / 24000.0f;
    const float q0 = __fmul_rn(x, r);
    return fmaf(fmaf(-q0, 24000.0f, x), r, q0);
}

__device__ __forceinline__ double wave_incl_scan(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        double u = __shfl_up(v, o);
        if (lane >= o) v += u;
    }
    return v;
}

// One wavefront per frame (grid (ceil(T / 4), B), four frames per workgroup): lane l < 60 owns the frame's samples 8 l ... 8 l + 7.
// csum[b][m][t] = sum over the frame's 480 samples of inc_m: a lane adds its eight increments, the wave reduces - no LDS, no barrier.
// Ragged batch (ragged.h) in the three oscillator kernels: f0 / amps / csum are [rows][T] over the whole batch (T = row stride), utterance
// b = blockIdx.y owns frames pre[b] ... + tb[b]; its phase starts at zero and its interpolations clamp at ITS last frame.
static __global__ __launch_bounds__(256) void harm_frame_sum_kernel(const float* __restrict__ f0, double* __restrict__ csum,
                                                                    int T, float scale, RagDev rg) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int t = 4 * blockIdx.x + (threadIdx.x >> 6);
    const int rs = T;
    long fbase = (long)b * T, cbase = (long)b * kHarm * T;
    if (rg.tb) {
        T = rg.tb[b];
        fbase = cbase = rg.pre[b];
        scale = __fdiv_rn((float)T, (float)(T * kHop));      // what the host computes for an utterance of its own
    }
    if (t >= T) return;
    const float* f = f0 + fbase;
    double acc[kHarm];
#pragma unroll
    for (int m = 0; m < kHarm; ++m) acc[m] = 0.0;
    if (lane < kHop / 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            Lerp c = lerp_coord(t * kHop + 8 * lane + k, scale, T);
            float fs = lerp_eval(c, f[c.i0], f[c.i1]);
#pragma unroll
            for (int m = 0; m < kHarm; ++m) acc[m] += (double)div24k(__fmul_rn(fs, (float)(m + 1)));
        }
    }
#pragma unroll
    for (int m = 0; m < kHarm; ++m) {
        double v = acc[m];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == m) csum[cbase + (long)m * rs + t] = v;
    }
}

// grid (15, B), one wavefront: exclusive prefix over frames, in place
static __global__ __launch_bounds__(64) void harm_frame_scan_kernel(double* __restrict__ csum, int T, RagDev rg) {
    double* p = csum + ((long)blockIdx.y * kHarm + blockIdx.x) * T;
    if (rg.tb) {
        p = csum + (long)blockIdx.x * T + rg.pre[blockIdx.y];
        T = rg.tb[blockIdx.y];
    }
    const int lane = threadIdx.x;
    double carry = 0.0;
    for (int base = 0; base < T; base += 64) {
        int i = base + lane;
        double v = i < T ? p[i] : 0.0;
        double inc = wave_incl_scan(v, lane);
        if (i < T) p[i] = carry + (inc - v);
        carry += __shfl(inc, 63);
    }
}

// One wavefront per frame (grid (ceil(T / 4), B)): lane l < 60 owns samples 8 l ... 8 l + 7.  Per harmonic: the lane's eight increments,
// one wave scan of their sums, the running prefix inside the lane - no LDS, no barrier (the 256-thread-per-frame version paid two
// barriers and a four-wave carry per harmonic).
static __global__ __launch_bounds__(256) void harm_synth_kernel(const float* __restrict__ f0, const float* __restrict__ amps,
                                                                const double* __restrict__ coff, float* __restrict__ source,
                                                                int T, float scale_size, float scale_amp, RagDev rg) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int t = 4 * blockIdx.x + (threadIdx.x >> 6);
    const int rs = T;
    const long Ls = (long)T * kHop;                      // row stride of `source`
    long fbase = (long)b * T, abase = (long)b * kHarm * T, sbase = (long)b * 16 * Ls;
    if (rg.tb) {
        T = rg.tb[b];
        fbase = abase = rg.pre[b];
        sbase = (long)rg.pre[b] * kHop;
        scale_size = __fdiv_rn((float)T, (float)(T * kHop));
    }
    if (t >= T) return;
    const bool act = lane < kHop / 8;
    const float* f = f0 + fbase;
    const int p0 = t * kHop + 8 * (act ? lane : 0);
    float fs[8], uv[8];
    Lerp ca[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const Lerp c = lerp_coord(p0 + e, scale_size, T);
        const float a0 = f[c.i0], a1 = f[c.i1];
        fs[e] = lerp_eval(c, a0, a1);
        uv[e] = lerp_eval(c, a0 > 20.f ? 1.f : 0.f, a1 > 20.f ? 1.f : 0.f);
        ca[e] = lerp_coord(p0 + e, scale_amp, T);
    }
    for (int m = 0; m < kHarm; ++m) {
        double d[8], sum = 0.0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            d[e] = act ? (double)div24k(__fmul_rn(fs[e], (float)(m + 1))) : 0.0;
            sum += d[e];
        }
        const double inc = wave_incl_scan(sum, lane);
        double run = coff[abase + (long)m * rs + t] + (inc - sum);       // phase in front of this lane's first sample
        if (act) {
            const float* am = amps + abase + (long)m * rs;
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                run += d[e];
                const float I = (float)run;                  // prefix rounded to fp32 (torch.cumsum output)
                const float frac = I - floorf(I);            // I % 1
                const float theta = __fmul_rn(6.2831854820251465f, frac);
                const float hsin = __fmul_rn(sincos_small(theta).x, uv[e]);
                const float amp = lerp_eval(ca[e], am[ca[e].i0], am[ca[e].i1]);
                o[e] = __fmul_rn(hsin, amp);
            }
            float4* dst = reinterpret_cast<float4*>(source + sbase + (long)m * Ls + p0);
            dst[0] = make_float4(o[0], o[1], o[2], o[3]);
            dst[1] = make_float4(o[4], o[5], o[6], o[7]);
        }
    }
}

// =================================================================================================
// Filtered noise (decoder.py:63-85): Y = exp(i angle) * kernel, zero frame prepended, rectangular
// window iSTFT (n_fft 1920, hop 480).  The per-frame c2r transform is a wave-level inverse FFT (fft.hip:
// run_noise_ifft); overlap-add divides by the frame-count envelope and trims 960 samples per side.
// =================================================================================================
// grid (x, B): blockIdx.y = utterance; smax[b] = per-utterance |max| slot of `source` (block-floating-point guard of
// FilterNet's first conv, conv3s.h): one atomic per workgroup
static __global__ __launch_bounds__(256) void noise_ola_kernel(const float* __restrict__ frames, float* __restrict__ source, int B, int T, float* __restrict__ smax, RagDev rg) {
    __shared__ float red[4];
    const long Ls = (long)T * kHop;                      // row stride of `source`
    const long b = blockIdx.y;
    long fbase = b * T, sbase = (b * 16 + 15) * Ls;
    if (rg.tb) {       // ragged batch (ragged.h): utterance b's frames / samples start behind those of the utterances before it; the envelope ends at ITS end
        T = rg.tb[b];
        fbase = rg.pre[b];
        sbase = 15 * Ls + (long)rg.pre[b] * kHop;
    }
    const long L = (long)T * kHop;
    float mx = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < L; i += (long)gridDim.x * blockDim.x) {
        int p = (int)i;
        int q = p + kNfft / 2;  // position in the untrimmed signal; frame f (0..T) covers [480 f, 480 f + 1920)
        int f_hi = q / kHop;
        if (f_hi > T) f_hi = T;
        int f_lo = (q - (kNfft - 1) + kHop - 1) / kHop;
        if (q - (kNfft - 1) <= 0) f_lo = 0;
        float s = 0.f;
        for (int f = f_lo > 1 ? f_lo : 1; f <= f_hi; ++f)  // frame 0 is the zero pad (decoder.py:81)
            s = __fadd_rn(s, frames[(fbase + (f - 1)) * kNfft + (q - f * kHop)]);
        const float v = s / (float)(f_hi - f_lo + 1);
        source[sbase + p] = v;
        mx = fmaxf(mx, fabsf(v));
    }
    amax_flush_wg(smax + b, mx, red);
}

// =================================================================================================
// SourceNet + dsp
// =================================================================================================
// cmax: per-utterance |max| slot of `content` (block-floating-point guard of the fp16 split, conv3s.h)
static int run_source_net(tvc_ctx* ctx, hipStream_t s, Ws& ws, bool dry, const float* content, const float* f0,
                          const float* energy, float* amps, float* kern, int B, int T, const float* cmax, float* xmax_zeroed, hipEvent_t amps_ready = nullptr) {
    const int ncols = B * T;
    float* ef = ws.get<float>((size_t)B * T);
    float* x = ws.get<float>((size_t)B * kSrcCh * T);
    const int NB = ctx->rag ? ctx->rag->B : B;      // utterances (a ragged batch runs as B = 1, T = all its frames: ragged.h)
    float* xmax = xmax_zeroed ? xmax_zeroed : ws.get<float>((size_t)NB);      // (the caller's, zeroed with its other slots, or an own one)
    if (!dry) {
        hipLaunchKernelGGL(window_max_kernel, dim3(grid_for((long)ncols * 64)), dim3(256), 0, s, energy, ef, (long)B, T, kHop);
        EpiSumCond ep{x, ctx->src_content_in.bias, ef, f0, ctx->src_e_w, ctx->src_e_b, ctx->src_f_w, ctx->src_f_b, kSrcCh, T, ncols};
        int rc = 0;
        if (!gemm_s2_try(&rc, ctx, s, ctx->src_content_in, content, B, kSslDim, T, 0, ep, cmax)) rc = gemm_s_launch<2, 4, 2>(ctx, s, ctx->src_content_in, content, B, kSslDim, T, 0, ep, cmax);
        TVC_CHECK(rc);
    }
    if (!dry && !xmax_zeroed) TVC_HIP(ctx, hipMemsetAsync(xmax, 0, (size_t)NB * sizeof(float), s));
    for (int i = 0; i < 3; ++i) TVC_CHECK(run_convnext(ctx, s, ws, dry, ctx->src_mid[i], x, B, T, i == 2 ? xmax : nullptr));     // the last layer publishes the |max| slot of its output
    if (dry) return 0;
    // to_amps (128 -> 15 rows: one 32-row m-tile)
    EpiBias<ACT_ELU1, false> ea{amps, ctx->src_to_amps.bias, nullptr, kHarm, T, ncols, (long)kHarm * T, 0};
    TVC_CHECK((gemm_s_launch<1, 4, 2>(ctx, s, ctx->src_to_amps, x, B, kSrcCh, T, 0, ea, xmax)));
    if (amps_ready) TVC_HIP(ctx, hipEventRecord(amps_ready, s));      // (run_decoder's fork: the oscillator on the side stream starts here, beside to_kernel)
    // to_kernel (128 -> 961 rows): the one sizeable contraction of the net, on the split-precision path
    EpiBias<ACT_ELU1, false> ek{kern, ctx->src_to_kernel.bias, nullptr, kBins, T, ncols, (long)kBins * T, 0};
    TVC_CHECK((gemm_s_launch<2, 4, 2>(ctx, s, ctx->src_to_kernel, x, B, kSrcCh, T, 0, ek, xmax)));
    return launch_check(ctx, "source_net");
}

// the oscillator's per-frame phase sums and their exclusive scan: needs f0 only
static int run_harm_sums(tvc_ctx* ctx, hipStream_t s, const float* f0, double* csum, int B, int T) {
    const long L = (long)T * kHop;
    const int NB = ctx->rag ? ctx->rag->B : B;
    const int Tg = ctx->rag ? ctx->rag->Tlong : T;
    RagDev rg;
    TVC_CHECK(rag_view(ctx, s, 1, 0, &rg, nullptr));
    const float scale_size = (float)T / (float)L;
    hipLaunchKernelGGL(harm_frame_sum_kernel, dim3((Tg + 3) / 4, NB), dim3(256), 0, s, f0, csum, T, scale_size, rg);
    hipLaunchKernelGGL(harm_frame_scan_kernel, dim3(kHarm, NB), dim3(64), 0, s, csum, T, rg);
    return launch_check(ctx, "harm_sums");
}

// Decoder.dsp (decoder.py:259-266): f0 [B,1,T], amps [B,15,T], kernel [B,961,T] -> source [B,16,L]
// smax (nullable): per-utterance |max| slot [B] of `source`, zeroed by the caller; the two kernels that write `source` publish into it
int run_dsp(tvc_ctx* ctx, hipStream_t s, Ws& ws, bool dry, const float* f0, const float* amps, const float* kern,
            const float* angle, uint64_t seed, float* source, int B, int T, float* smax, const DspFork* fk) {
    const long L = (long)T * kHop;
    double* csum = fk ? fk->csum : ws.get<double>((size_t)B * kHarm * T);
    float* frames = ws.get<float>((size_t)B * T * kNfft);
    const int NB = ctx->rag ? ctx->rag->B : B;
    const int Tg = ctx->rag ? ctx->rag->Tlong : T;      // frames of the longest utterance: the per-utterance grids' extent
    float* smax_own = smax ? nullptr : ws.get<float>((size_t)NB);
    if (dry) return 0;
    if (!smax) {
        smax = smax_own;
        TVC_HIP(ctx, hipMemsetAsync(smax, 0, (size_t)NB * sizeof(float), s));
    }
    RagDev rg;
    TVC_CHECK(rag_view(ctx, s, 1, 0, &rg, nullptr));
    // harmonics -> source[:, 0:15]
    const float scale_size = (float)T / (float)L;         // F.interpolate(f0, Lw): size given
    const float scale_amp = (float)(1.0 / (double)kHop);  // F.interpolate(amps, scale_factor=480)
    hipStream_t sh = s;      // the harmonic branch's stream
    if (fk) {                // forked (equal-length batches only): the frame sums are already scanned on the side stream; the synthesis waits for the amplitudes
        sh = fk->side;
        TVC_HIP(ctx, hipStreamWaitEvent(sh, fk->amps_ready, 0));
    } else {
        TVC_CHECK(run_harm_sums(ctx, s, f0, csum, B, T));
    }
    hipLaunchKernelGGL(harm_synth_kernel, dim3((Tg + 3) / 4, NB), dim3(256), 0, sh, f0, amps, csum, source, T, scale_size, scale_amp, rg);
    // the 15 harmonic rows are sin(.) * voiced gate * interpolated amps: bounded by the amplitudes' own |max| (3 000 values per utterance)
    TVC_CHECK(run_amax_rows(ctx, sh, amps, B, kHarm, T, smax));
    // noise -> source[:, 15]: kernel * exp(i angle) -> inverse 1920-point FFT per frame (fft.hip) -> overlap-add
    // (angle == nullptr: the kernel draws the phases itself while it stages the tile - small_kernels.h noise_phase_hash(seed, row, bin, frame) -,
    // no [B][961][T] phase tensor is written or read)
    TVC_CHECK(run_noise_ifft(ctx, s, kern, angle, seed, frames, B, T, ctx->rag && angle));
    hipLaunchKernelGGL(noise_ola_kernel, dim3(grid_for((long)Tg * kHop / 4, 256, NB >= 32 ? 16 : 512 / NB), NB), dim3(256), 0, s, frames, source, B, T, smax, rg);
    return launch_check(ctx, "dsp");
}

// =================================================================================================
// FilterNet (decoder.py:193-233).  Every Conv1d runs on the split-precision fp16 MFMA path (conv3s.h):
//   level (channels @ rate)     kernels
//   24 @ L        downs[0]      down0s_kernel (+ the 1/5-rate pick Downsample 1 starts from)
//   24 -> 48 @ L/5  Downsample 1  down24f_kernel: the whole block in one launch (h1 / h2 on chip; c3 also accumulates down_res(xi) and writes Downsample 2's 1/4-rate input)
//   48 -> 96 @ L/20 Downsample 2  conv48s_kernel (c1, c2: LDS-resident weights), conv3s (c3 + down_res as a second K phase)
//   96 -> 192, 192 -> 384       conv3s x3 per block, same c3 fusion
//   384, 192, 96  Upsample 0-2  conv3s: c1 (interpolating while it stages), c2 + FiLM1 + interpolated residual, c3, c4 + FiLM2 + residual;
//                               c5 as a split-precision GEMM launch
//   48            Upsample 3    conv48s_kernel x4, c5 inside the c4 + FiLM2 launch
//   24 @ L        Upsample 4    up24s_kernel x2 (second half = c3, c4, FiLM2, and c5 folded into the k7 output conv)
// F.interpolate is never materialised: Downsample's 1/f pick / two-sample mean is written by the producing conv's
// epilogue, Upsample's xf is evaluated while c1 stages its input and again by c2's epilogue for the residual.
// Every tensor a conv reads has a per-utterance |max| slot (block-floating-point guard of the fp16 split): the producing
// kernel's epilogue publishes it; tensors finished by an epilogue functor (the c5 GEMMs, content_in) get one pass of amax_rows.
// The architecture is fixed (the module mirror only accepts the reference's default channels / factors).
// =================================================================================================
// y = conv(lrelu(x)) + b for the plain k3 convs of the 96..384-channel levels (lin > 0: x is the low-rate tensor, F.interpolate is
// evaluated while staging): the pipelined kernel of conv_s2.h, or conv3s.h's two-barrier kernel outside its preconditions
static int plain_conv(tvc_ctx* ctx, hipStream_t s, const PackedW& w, const float* x, int B, int Cin, int len, int dil, float* y, const BfpSlots& bfp, int lin = 0,
                      float lscale = 0.f) {
    int rc = 0;
    if ((lin > 0 ? conv_s2_try<true>(&rc, ctx, s, w, x, B, Cin, len, dil, y, bfp, lin, lscale) : conv_s2_try<false>(&rc, ctx, s, w, x, B, Cin, len, dil, y, bfp)))
        return rc;
    if (lin > 0)
        return conv3s_launch<true, C3EpiBias<false>, false, true>(ctx, s, w, x, B, Cin, len, dil, C3EpiBias<false>{y, w.bias, nullptr, w.M, len}, bfp, nullptr, nullptr,
                                                                   nullptr, 0, lin, lscale);
    return conv3s_launch<true>(ctx, s, w, x, B, Cin, len, dil, C3EpiBias<false>{y, w.bias, nullptr, w.M, len}, bfp);
}

// out = FiLM(conv(lrelu(h)), cond) + residual (res_lin > 0: the residual is F.interpolate of the low-rate tensor `res`)
static int film_conv(tvc_ctx* ctx, hipStream_t s, const PackedW& w, const PackedW& fw, const FilmU& fu, const float* h, const float* cond, int B, int C, int len, int dil,
                     float* out, const float* bsc, const float* bsh, const float* res, int res_lin, float res_scale, const BfpSlots& bfp) {
    int rc = 0;
    // The two kernels round differently (film_s2.h: one accumulator per result), so the choice may depend on nothing but the utterance's
    // own shape: an utterance converts to the same samples in every batch.  Below one 256-column tile the narrow conv3s tile wastes less.
    if (rag_min_len(ctx, len) >= FS2::BN && film_s2_try(&rc, ctx, s, fu, h, cond, B, C, len, dil, out, res, res_lin, res_scale, bfp)) return rc;
    return conv3s_launch<true, C3EpiFilmFused, true>(ctx, s, w, h, B, C, len, dil, C3EpiFilmFused{out, w.bias, bsc, bsh, res, C, len, res_lin, res_scale}, bfp, &fw, &fw,
                                                     cond, C);
}

// One half of an Upsample block on the 96 / 192 / 384-channel levels: h = ca(lrelu(x)) (lin > 0: x = F.interpolate of the low-rate tensor,
// evaluated while staging), out = FiLM(cb(lrelu(h)), cond) + residual.  When both convs take their pipelined kernels - a property of the
// utterance's shape only - h crosses HBM as the second conv's READY operand: conv_s2's epilogue writes split(lrelu(h) * 2^k) as two fp16
// planes (the bytes of the fp32 tensor), k from the analytic bound hb_w |x|max + hb_b >= |h|, and film_s2 stages it by copy: the
// lrelu / scale / split of every staged element - repeated by each of the C / 96 row blocks that read it - is done once, by the producer.
static int conv_film_half(tvc_ctx* ctx, hipStream_t s, const PackedW& ca, const PackedW& cb, const PackedW& fw, const FilmU& fu, const float* x, int lin, float lscale,
                          int da, int db, float* h, const float* cond, int B, int C, int len, float* out, const float* bsc, const float* bsh, const float* res,
                          int res_lin, float res_scale, const float* ma_in, float* mh, const float* mcond, float* mout) {
    const bool pre = rag_min_len(ctx, len) >= FS2::BN && fu.img && fu.C == C && fu.hb_w > 0.f && C % 96 == 0 && C <= 384 && ma_in && mcond && res && db >= 1 && db <= FS2::MAXD &&
                     (long)C * len * 4 < (1L << 32) && (res_lin <= 0 || (long)C * res_lin * 4 < (1L << 32));
    if (pre) {
        int rc = 0;
        const BfpSlots in{ma_in, nullptr, nullptr};
        const bool ok = lin > 0 ? conv_s2_try<true, true>(&rc, ctx, s, ca, x, B, C, len, da, h, in, lin, lscale, fu.hb_w, fu.hb_b)
                                : conv_s2_try<false, true>(&rc, ctx, s, ca, x, B, C, len, da, h, in, 0, 0.f, fu.hb_w, fu.hb_b);
        if (ok) {
            TVC_CHECK(rc);
            if (!film_s2_try(&rc, ctx, s, fu, h, cond, B, C, len, db, out, res, res_lin, res_scale, BfpSlots{ma_in, mcond, mout}, true, fu.hb_w, fu.hb_b))
                return fail(ctx, TVC_ERR_STATE, "filter_net: the FiLM conv refused the pre-split operand its producer wrote");
            return rc;
        }
    }
    TVC_CHECK(plain_conv(ctx, s, ca, x, B, C, len, da, h, BfpSlots{ma_in, nullptr, mh}, lin, lscale));
    return film_conv(ctx, s, cb, fw, fu, h, cond, B, C, len, db, out, bsc, bsh, res, res_lin, res_scale, BfpSlots{mh, mcond, mout});
}

// [B][C / 8][l][8] (G8) -> [B][C][l]
static __global__ void g8_to_planar_kernel(const float* __restrict__ x, float* __restrict__ y, long B, long l, int C) {
    const long n = B * C * l;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long t = i % l, bc = i / l, c = bc % C, b = bc / C;
        y[i] = x[((b * (C / 8) + c / 8) * l + t) * 8 + (c & 7)];
    }
}

// cmax / smax (nullable): |max| slots of `content` / of cat[source, energy] if the caller already has them
// FilterNet's input contraction x0 = content_in(content) + f0 embedding (decoder.py:224-226): needs `content`, `f0` and content's |max| slot only
static int filter_input_gemm(tvc_ctx* ctx, hipStream_t s, const float* content, const float* f0, float* x, int B, int T, const float* cmax) {
    EpiSumCond ep{x, ctx->flt_content_in.bias, nullptr, f0, nullptr, nullptr, ctx->flt_f_w, ctx->flt_f_b, 384, T, B * T};
    int rc = 0;
    if (!gemm_s2_try(&rc, ctx, s, ctx->flt_content_in, content, B, kSslDim, T, 0, ep, cmax)) rc = gemm_s_launch<2, 4, 2>(ctx, s, ctx->flt_content_in, content, B, kSslDim, T, 0, ep, cmax);
    return rc;
}

int run_filter(tvc_ctx* ctx, hipStream_t s, Ws& ws, bool dry, const float* content, const float* f0,
               const float* energy, const float* source, float* wave, int B, int T, const FilterTaps* taps, const float* cmax, const float* smax, float* zeroed_slots, bool x_slot_set,
               float* x_pre, hipEvent_t x_ready) {
    const long L = (long)T * kHop;
    static const int ch[5] = {384, 192, 96, 48, 24};
    const long len_dn[5] = {L, L / 5, L / 20, L / 80, L / 240};   // skip i lives at len_dn[i]
    float* skip[5];
    for (int i = 0; i < 5; ++i) skip[i] = ws.get<float>((size_t)B * ch[4 - i] * len_dn[i]);
    float* x = x_pre ? x_pre : ws.get<float>((size_t)B * ch[0] * T);
    // Downsample i's input = interpolate(skip[i-1], 1/f), written by the conv that produces skip[i-1]
    float* xi_pre[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int i = 1; i <= 4; ++i) xi_pre[i] = ws.get<float>((size_t)B * ctx->downs[i - 1].cin * len_dn[i]);
    for (int i = 1; i < 4; ++i)
        if (len_dn[i] % ctx->downs[i].factor != 0 || (ctx->downs[i].factor == 4 && len_dn[i] % 4 != 0))
            return fail(ctx, TVC_ERR_ARG, "filter_net: level lengths must divide by the next Downsample factor");   // (the epilogues' 1/f-rate copies rely on it)

    // |max| slots, [B] floats each, one memset
    enum { S_CONTENT = 0, S_SRC, S_X, S_SKIP0, S_DH1 = S_SKIP0 + 5, S_DH2 = S_DH1 + 4, S_UHA = S_DH2 + 4, S_UX1 = S_UHA + 5, S_UHB = S_UX1 + 5, S_UXU = S_UHB + 5,
           S_LEV = S_UXU + 5, S_COUNT = S_LEV + 5 };
    const int NB = ctx->rag ? ctx->rag->B : B;      // utterances (a ragged batch runs as B = 1, T = all its frames: ragged.h)
    static_assert(S_COUNT == kFilterSlots && S_X == kFilterSlotX, "tvc_common.h kFilterSlots / kFilterSlotX");
    float* slots = zeroed_slots ? zeroed_slots : ws.get<float>((size_t)S_COUNT * NB);
    auto slot = [&](int i) { return slots + (size_t)i * NB; };

    if (!dry) {
        ProfScope ps(ctx, s, dry, "filter.in+down0");
        if (!zeroed_slots) TVC_HIP(ctx, hipMemsetAsync(slots, 0, (size_t)S_COUNT * NB * sizeof(float), s));
        if (!cmax) {
            TVC_CHECK(run_amax_rows(ctx, s, content, B, kSslDim, T, slot(S_CONTENT)));
            cmax = slot(S_CONTENT);
        }
        if (!smax) {
            TVC_CHECK(run_amax_rows(ctx, s, source, B, 16, L, slot(S_SRC)));
            TVC_CHECK(run_amax_rows(ctx, s, energy, B, 1, L, slot(S_SRC)));
            smax = slot(S_SRC);
        }
        if (!x_ready) TVC_CHECK(filter_input_gemm(ctx, s, content, f0, x, B, T, cmax));      // (x_ready: the caller launched it on its side stream)
        // x0's |max| slot: the functor finishes the elements, so the slot is the bound bw |content|max + bb instead of a pass over x0
        if (!(zeroed_slots && x_slot_set)) TVC_CHECK(run_slot_affine(ctx, s, slot(S_X), cmax, 1, ctx->flt_in_bw, ctx->flt_in_bb, NB));
        // skips[0] is read as FiLM cond only (ups[4]): it is written as the two halves' ready operand; the fp32 tensor exists for the parity tap alone
        TVC_CHECK(run_down0_split(ctx, s, ctx->flt_down0s, source, energy, skip[0], taps ? taps->skips[0] : nullptr, xi_pre[1], B, (int)L, smax, slot(S_SKIP0)));
    }
    // down path (xi = the 1/f-rate pick / two-sample mean of skip[i-1]: bounded by skip[i-1]'s |max|, same slot)
    for (int i = 1; i <= 4; ++i) {
        const DownW& d = ctx->downs[i - 1];
        const int len = (int)len_dn[i];
        size_t mk = ws.mark();
        float* xi = xi_pre[i];
        float* h1 = d.cin == 24 ? nullptr : ws.get<float>((size_t)B * d.cin * len);      // (the 24-channel block keeps its intermediates on chip)
        float* h2 = d.cin == 24 ? nullptr : ws.get<float>((size_t)B * d.cin * len);
        if (!dry) {
            static const char* names[4] = {"filter.down1", "filter.down2", "filter.down3", "filter.down4"};
            ProfScope ps(ctx, s, dry, names[i - 1]);
            float* y2 = i < 4 ? xi_pre[i + 1] : nullptr;              // the next block's 1/f-rate input
            const int f2 = i < 4 ? ctx->downs[i].factor : 0;
            const float* mxi = slot(S_SKIP0 + i - 1);
            float *mh1 = slot(S_DH1 + i - 1), *mh2 = slot(S_DH2 + i - 1), *mout = slot(S_SKIP0 + i);
            if (d.cin == 24) {   // the whole 24-channel block in one launch; c3 folds down_res(xi) in and writes y2
                TVC_CHECK(run_down24_fused(ctx, s, d, xi, skip[i], y2, B, len, mxi, mout));
            } else {
                if (d.cin == 48) {   // c1 -> c2 in one launch, weights resident in LDS, c1's output never leaves the CU (conv48s.hip)
                    TVC_CHECK(run_conv48_pair(ctx, s, d.c1, d.c2, xi, 0, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.f, h2, B, len, 1, 2, mxi, nullptr, mh2));
                } else {
                    TVC_CHECK(plain_conv(ctx, s, d.c1, xi, B, d.cin, len, 1, h1, BfpSlots{mxi, nullptr, mh1}));
                    TVC_CHECK(plain_conv(ctx, s, d.c2, h1, B, d.cin, len, 2, h2, BfpSlots{mh1, nullptr, mh2}));
                }
                // c3 + down_res(xi) as a second K phase on the same accumulators: no residual tensor, no 1x1 launch
                TVC_CHECK((conv3s_launch<true, C3EpiBiasResConv>(ctx, s, d.c3, h2, B, d.cin, len, 4,
                                                                 C3EpiBiasResConv{{skip[i], d.c3res_bias, nullptr, d.cout, len, y2, f2}},
                                                                 BfpSlots{mh2, mxi, mout}, &d.res, nullptr, xi, d.cin)));
            }
        }
        ws.release(mk);
    }
    if (!dry && x_ready) TVC_HIP(ctx, hipStreamWaitEvent(s, x_ready, 0));      // the up path is x0's first reader: join here, behind the whole down path
    // up path: level outputs are persistent, block temporaries are released per level
    float* xlev[5];
    {
        long l = T;
        for (int i = 0; i < 5; ++i) {
            l *= ctx->ups[i].factor;
            xlev[i] = ws.get<float>((size_t)B * ctx->ups[i].cout * l);
        }
    }
    long len = T;
    const float* mx_in = slot(S_X);                                   // |max| slot of the level's input
    for (int i = 0; i < 5; ++i) {
        const UpW& u = ctx->ups[i];
        const int lin = (int)len;
        len *= u.factor;
        const int lo = (int)len, C = u.cin, nc = B * lo;
        const float* cond = skip[4 - i];
        const float* mcond = slot(S_SKIP0 + 4 - i);
        size_t mk = ws.mark();
        float* xu = ws.get<float>((size_t)B * C * lo);
        float* h = ws.get<float>((size_t)B * C * lo);
        float* x1 = ws.get<float>((size_t)B * C * lo);
        if (!dry && C == 24) {
            // last level: Upsample block + output_layer in two launches, waveform written directly
            ProfScope ps(ctx, s, dry, "filter.up4+out");
            TVC_CHECK(run_up24_split(ctx, s, u, x, cond, ctx->down0_bw, ctx->down0_bb, x1, wave, B, lo, mx_in, smax, slot(S_UX1 + i)));
        } else if (!dry) {
            static const char* names[4] = {"filter.up0", "filter.up1", "filter.up2", "filter.up3"};
            ProfScope ps(ctx, s, dry, names[i]);
            const float lscale = (float)(1.0 / (double)u.factor);   // F.interpolate(scale_factor=f): ATen uses float(1/f)
            for (int half = 0; half < 2; ++half) {
                const PackedW& ca = half ? u.c3 : u.c1;
                const PackedW& cb = half ? u.c4 : u.c2;
                const PackedW& fw = half ? u.film2 : u.film1;      // stacked [to_scale ; to_shift] rows, each group padded to whole 32-row tiles
                const float* bsc = fw.bias;                        // its bias row: [b_scale (C) ; b_shift (C)]
                const float* bsh = fw.bias + C;
                const int da = half ? 9 : 1, db = half ? 27 : 3;
                float* xout = half ? xu : x1;   // first half: x1 = FiLM1(c2(c1(xf))) + xf; second half: xu = FiLM2(c4(c3(x1))) + x1
                const float* ma_in = half ? slot(S_UX1 + i) : mx_in;             // input of the half's first conv
                float* mh = slot((half ? S_UHB : S_UHA) + i);                    // its output h
                float* mout = slot((half ? S_UXU : S_UX1) + i);                  // the half's output
                if (C == 48) {   // LDS-resident weights (conv48s.hip)
                    if (half == 0) {   // c1 (interpolating) -> c2 + FiLM1 + interpolated residual in one launch
                        TVC_CHECK(run_conv48_pair(ctx, s, ca, cb, x, lin, lscale, &fw, bsc, bsh, cond, x, lin, lscale, xout, B, lo, da, db, ma_in, mcond, mout));
                    } else {
                        TVC_CHECK(run_conv48s(ctx, s, ca, x1, 0, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.f, h, B, lo, da, ma_in, nullptr, mh));
                        // c4 + FiLM2 + residual + c5 (48 -> 24) in one launch: the level's output (and its |max|) is written directly
                        TVC_CHECK(run_conv48s(ctx, s, cb, h, 0, 0.f, &fw, bsc, bsh, cond, x1, 0, 0.f, nullptr, B, lo, db, mh, mcond, slot(S_LEV + i), &u.c5, xlev[i]));
                    }
                } else if (half == 0) {
                    TVC_CHECK(conv_film_half(ctx, s, ca, cb, fw, u.fu1, x, lin, lscale, da, db, h, cond, B, C, lo, xout, bsc, bsh, x, lin, lscale, ma_in, mh, mcond, mout));
                } else {
                    TVC_CHECK(conv_film_half(ctx, s, ca, cb, fw, u.fu2, x1, 0, 0.f, da, db, h, cond, B, C, lo, xout, bsc, bsh, x1, 0, 0.f, ma_in, mh, mcond, mout));
                }
            }
            if (C != 48) {   // c5 (1x1, C -> C/2) as a split-precision GEMM launch; its output's |max| slot = the bound c5_bw |xu|max + c5_bb (the functor finishes the elements)
                // (the launch also writes the level output's slot: the bound c5_bw |xu|max + c5_bb from its input's slot)
                EpiBias<ACT_NONE, false> ep{xlev[i], u.c5.bias, nullptr, u.cout, lo, nc, (long)u.cout * lo, 0, slot(S_LEV + i), slot(S_UXU + i), u.c5_bw, u.c5_bb, NB};
                if (u.cout == 48) {   // the 48-channel level output: G8 layout (conv48s.hip's interpolating launch reads 16-byte rows)
                    EpiBiasG8 eg{xlev[i], u.c5.bias, u.cout, lo, nc, slot(S_LEV + i), slot(S_UXU + i), u.c5_bw, u.c5_bb, NB};
                    TVC_CHECK((gemm_s_launch<2, 4, 2>(ctx, s, u.c5, xu, B, C, lo, 0, eg, slot(S_UXU + i))));
                } else if (u.c5.MT6 % 3 == 0) TVC_CHECK((gemm_s_launch<3, 4, 1>(ctx, s, u.c5, xu, B, C, lo, 0, ep, slot(S_UXU + i))));
                else TVC_CHECK((gemm_s_launch<2, 4, 2>(ctx, s, u.c5, xu, B, C, lo, 0, ep, slot(S_UXU + i))));
            }
        }
        ws.release(mk);
        x = xlev[i];
        mx_in = slot(S_LEV + i);
    }
    if (!dry && taps) {   // parity taps: the block outputs are still live in the workspace
        for (int i = 1; i < 5; ++i)      // (skips[0] was written into the tap by its producer)
            if (taps->skips[i] && i == 1)      // the 48-channel skip travels in the G8 layout (down24f_kernel -> conv48s.hip's FiLM cond)
                hipLaunchKernelGGL(g8_to_planar_kernel, dim3(grid_for((long)B * 48 * len_dn[1])), dim3(256), 0, s, skip[1], taps->skips[1], (long)B, len_dn[1], 48);
            else if (taps->skips[i])
                TVC_HIP(ctx, hipMemcpyAsync(taps->skips[i], skip[i], (size_t)B * ch[4 - i] * len_dn[i] * sizeof(float), hipMemcpyDeviceToDevice, s));
        long l = T;
        for (int i = 0; i < 4; ++i) {
            l *= ctx->ups[i].factor;
            if (taps->ups[i] && ctx->ups[i].cout <= 48)      // the 48- and 24-channel levels travel in the fused kernels' G8 layout [B][C / 8][l][8]: back to [B][C][l] for the tap
                hipLaunchKernelGGL(g8_to_planar_kernel, dim3(grid_for((long)B * ctx->ups[i].cout * l)), dim3(256), 0, s, xlev[i], taps->ups[i], (long)B, l, ctx->ups[i].cout);
            else if (taps->ups[i])
                TVC_HIP(ctx, hipMemcpyAsync(taps->ups[i], xlev[i], (size_t)B * ctx->ups[i].cout * l * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
    }
    return dry ? 0 : launch_check(ctx, "filter_net");
}

int run_decoder(tvc_ctx* ctx, hipStream_t s, Ws& ws, bool dry, const float* content, const float* f0,
                const float* energy, const float* angle, uint64_t seed, float* wave, float* amps_out,
                float* kernel_out, float* source_out, int B, int T, const float* content_bound, const float* energy_bound) {
    const long L = (long)T * kHop;
    float* amps = amps_out ? amps_out : ws.get<float>((size_t)B * kHarm * T);
    float* kern = kernel_out ? kernel_out : ws.get<float>((size_t)B * kBins * T);
    float* source = source_out ? source_out : ws.get<float>((size_t)B * 16 * L);
    // |max| slots shared by the stages: content (SourceNet's and FilterNet's input contraction), cat[source, energy] (FilterNet's first conv)
    const int NB = ctx->rag ? ctx->rag->B : B;
    // (one block, zeroed by one launch together with the two bounds: [cmax | smax | SourceNet's output slot | FilterNet's slots])
    float* cmax = ws.get<float>((size_t)(3 + kFilterSlots) * NB);
    float* smax = cmax + NB;
    float* xmax = smax + NB;
    float* fslots = xmax + NB;
    if (!dry) {
        TVC_CHECK(run_slot_prep(ctx, s, cmax, (3 + kFilterSlots) * NB, content_bound ? cmax : nullptr, content_bound, 0, 1.f, 0.f, energy_bound ? smax : nullptr,
                                energy_bound, 1, 1.f, 0.f, content_bound ? fslots + (size_t)kFilterSlotX * NB : nullptr, ctx->flt_in_bw, ctx->flt_in_bb,
                                NB));      // (the dsp kernels raise smax to cat[source, energy]'s; the third: FilterNet's x0 slot from |content|max)
        if (!content_bound) TVC_CHECK(run_amax_rows(ctx, s, content, B, kSslDim, T, cmax));
        if (!energy_bound) TVC_CHECK(run_amax_rows(ctx, s, energy, B, 1, L, smax));
    }
    // FilterNet's input contraction (768 -> 384 at the frame rate: 48 us at the bench shape) reads nothing SourceNet or the DSP stage produce and
    // its output is first read by Upsample 0, behind the whole down path: outside a stream capture (a fork inside a replayed graph costs more than
    // it hides) it runs on the
    // context's side stream - free again since the encoder joined its pitch chain - beside SourceNet's small launches and the vector-ALU-bound
    // oscillator, and run_filter waits for it where the up path begins.
    float* x0 = nullptr;
    double* csum = nullptr;
    bool fork = false;
    if (wave || dry) {      // (a dry run sizes the workspace for the full decoder whatever pointers it was handed)
        x0 = ws.get<float>((size_t)B * 384 * T);
        csum = ws.get<double>((size_t)B * kHarm * T);
        fork = !dry && ctx->side && ctx->ev_fork2 && ctx->ev_join2 && ctx->ev_amps;      // (a ragged batch too: the side stream's kernels read its base tables only - rag_setup built them on s in front of the fork -, no column-tile table)
        if (fork) {
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) fork = false;
        }
        if (fork) {
            // The side stream's chain: the oscillator's frame sums (f0 only) -> FilterNet's input contraction -> [amplitudes ready] -> the harmonic
            // synthesis (vector-ALU-bound, no LDS) beside SourceNet's to_kernel GEMM and the noise branch's FFTs on the launch stream.
            TVC_HIP(ctx, hipEventRecord(ctx->ev_fork2, s));            // content, f0 and cmax are complete on s
            TVC_HIP(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_fork2, 0));
            TVC_CHECK(run_harm_sums(ctx, ctx->side, f0, csum, B, T));
            {
                ProfScope ps(ctx, ctx->side, dry, "filter_net.input@side");      // FilterNet's work outside its region on the launch stream: bench.py adds it to the roofline's duration
                TVC_CHECK(filter_input_gemm(ctx, ctx->side, content, f0, x0, B, T, cmax));
            }
        }
    }
    size_t mk = ws.mark();
    {
        ProfScope ps(ctx, s, dry, "source_net");
        TVC_CHECK(run_source_net(ctx, s, ws, dry, content, f0, energy, amps, kern, B, T, cmax, xmax, fork ? ctx->ev_amps : nullptr));
    }
    ws.release(mk);
    if (!dry && !wave && !source_out) return 0;  // SourceNet.forward alone (decoder.py:126-134): the caller asked for amps / kernel only
    {
        ProfScope ps(ctx, s, dry, "dsp");
        const DspFork fk{ctx->side, csum, ctx->ev_amps};
        TVC_CHECK(run_dsp(ctx, s, ws, dry, f0, amps, kern, angle, seed, source, B, T, smax, fork ? &fk : nullptr));
        if (fork) {      // join: the side stream's whole chain (harmonics, FilterNet's input contraction) is behind this event
            TVC_HIP(ctx, hipEventRecord(ctx->ev_join2, ctx->side));
            TVC_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_join2, 0));
        }
    }
    ws.release(mk);
    if (!dry && !wave) return 0;                 // ... or for Decoder.dsp's output (decoder.py:259-266) without the FilterNet pass
    ProfScope ps(ctx, s, dry, "filter_net");
    TVC_CHECK(run_filter(ctx, s, ws, dry, content, f0, energy, source, wave, B, T, nullptr, cmax, smax, fslots, content_bound != nullptr, x0, fork ? ctx->ev_join2 : nullptr));
    ws.release(mk);
    return 0;
}

}  // namespace tvc

#ifdef S_TRACE
extern "C" int tvc_debug_trace_dec(unsigned long long* host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(tvc::g_trace), sizeof(tvc::g_trace)) == hipSuccess ? 0 : -1;
}
#endif
