// Source-filter decoder (decoder.py:24-266): SourceNet, additive harmonic oscillator, filtered-noise
// iSTFT, and the FilterNet U-Net.
#include "conv3.h"
#include "conv3m48.h"
#include "conv3s.h"
#include "igemm.h"
#include "small_kernels.h"
#include "tvc_common.h"

#ifndef TVC_C48_C5
#define TVC_C48_C5 1   // ups.3: c5 (48 -> 24) applied inside the c4 + FiLM2 launch (conv48s.hip)
#endif
#ifndef TVC_RESCONV
#define TVC_RESCONV 1   // Downsample 2-4: down_res(xi) accumulated by c3's launch as a second K phase (no residual tensor, no 1x1 launch)
#endif
#ifndef TVC_C48R
#define TVC_C48R 1   // 48-channel k3 convs (ups.3, Downsample 2's c1 / c2) with LDS-resident weights (conv48s.hip); 0 = the generic split kernel
#endif
#ifndef TVC_SPLIT_SRC
#define TVC_SPLIT_SRC 1   // SourceNet's to_kernel 1x1 (128 -> 961) on the split-precision GEMM path
#endif
#ifndef TVC_FFT
#define TVC_FFT 1   // filtered-noise iSTFT as wave-level FFTs (fft.hip); 0 = real-DFT GEMMs
#endif

namespace tvc {

#ifndef TVC_FUSE_LERP
#define TVC_FUSE_LERP 1     // Upsample's interpolate evaluated inside c1's staging and c2's residual epilogue (split-path levels)
#endif
#ifndef TVC_FUSE_DECIM
#define TVC_FUSE_DECIM 1    // Downsample's interpolate(1/f) written by the producing conv's epilogue (pick / two-sample mean)
#endif
#ifndef TVC_DSP_FORK
#define TVC_DSP_FORK 0      // 1: harmonic oscillator on the side stream beside the filtered-noise branch (measured: no gain, both fill the GPU)
#endif
#ifndef TVC_SPLIT_1X1
#define TVC_SPLIT_1X1 1    // FilterNet's remaining 1x1 convs (Upsample.c5 of ups.0-2) on the split-precision GEMM path (-0.05 ms)
#endif
#ifndef TVC_SPLIT_IDFT
#define TVC_SPLIT_IDFT 1   // inverse DFT GEMMs of the noise branch on the split-precision path
#endif
#ifndef IDFT_MTB
#define IDFT_MTB 2
#define IDFT_NWV 4
#define IDFT_BPC 2
#endif

#ifndef TVC_USE_C48
#define TVC_USE_C48 1
#endif

// =================================================================================================
// Harmonic oscillator (decoder.py:24-54).  The phase of harmonic m is the running sum over the whole
// utterance of fl32(fl32(fs*m)/24000), accumulated in fp64 and rounded to fp32 per sample — what
// ATen's CPU cumsum does for fp32 input.  Hierarchical scan: per-frame fp64 sums, an exclusive scan
// of those per (utterance, harmonic), then an in-frame scan fused with sin / voiced gate / amplitude.
// =================================================================================================

__device__ __forceinline__ double wave_incl_scan(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        double u = __shfl_up(v, o);
        if (lane >= o) v += u;
    }
    return v;
}

// grid (T, B): csum[b][m][t] = sum over the frame's 480 samples of inc_m
static __global__ __launch_bounds__(256) void harm_frame_sum_kernel(const float* __restrict__ f0, double* __restrict__ csum,
                                                                    int T, float scale) {
    __shared__ double red[4][kHarm];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const float* f = f0 + (long)b * T;
    double acc[kHarm];
#pragma unroll
    for (int m = 0; m < kHarm; ++m) acc[m] = 0.0;
    for (int i = tid; i < kHop; i += 256) {
        Lerp c = lerp_coord(t * kHop + i, scale, T);
        float fs = lerp_eval(c, f[c.i0], f[c.i1]);
#pragma unroll
        for (int m = 0; m < kHarm; ++m) acc[m] += (double)__fdiv_rn(__fmul_rn(fs, (float)(m + 1)), 24000.f);
    }
#pragma unroll
    for (int m = 0; m < kHarm; ++m) {
        double v = acc[m];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave][m] = v;
    }
    __syncthreads();
    if (tid < kHarm) csum[((long)b * kHarm + tid) * T + t] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

// grid (15, B), one wavefront: exclusive prefix over frames, in place
static __global__ __launch_bounds__(64) void harm_frame_scan_kernel(double* __restrict__ csum, int T) {
    double* p = csum + ((long)blockIdx.y * kHarm + blockIdx.x) * T;
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

// grid (T, B): thread i < 240 owns samples 2i, 2i+1 of the frame
static __global__ __launch_bounds__(256) void harm_synth_kernel(const float* __restrict__ f0, const float* __restrict__ amps,
                                                                const double* __restrict__ coff, float* __restrict__ source,
                                                                int T, float scale_size, float scale_amp) {
    __shared__ double wtot[4];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const bool act = tid < kHop / 2;
    const long L = (long)T * kHop;
    const float* f = f0 + (long)b * T;
    const int p0 = t * kHop + 2 * tid;
    float fs[2], uv[2];
    Lerp ca[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        int p = act ? p0 + e : t * kHop;
        Lerp c = lerp_coord(p, scale_size, T);
        float a0 = f[c.i0], a1 = f[c.i1];
        fs[e] = lerp_eval(c, a0, a1);
        uv[e] = lerp_eval(c, a0 > 20.f ? 1.f : 0.f, a1 > 20.f ? 1.f : 0.f);
        ca[e] = lerp_coord(p, scale_amp, T);
    }
    for (int m = 0; m < kHarm; ++m) {
        double d0 = act ? (double)__fdiv_rn(__fmul_rn(fs[0], (float)(m + 1)), 24000.f) : 0.0;
        double d1 = act ? (double)__fdiv_rn(__fmul_rn(fs[1], (float)(m + 1)), 24000.f) : 0.0;
        double pair = d0 + d1;
        double inc = wave_incl_scan(pair, lane);
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        double base = coff[((long)b * kHarm + m) * T + t];
        for (int w = 0; w < wave; ++w) base += wtot[w];
        __syncthreads();
        if (act) {
            double e0 = base + (inc - pair) + d0;
            double e1 = e0 + d1;
            const float* am = amps + ((long)b * kHarm + m) * T;
            float o[2];
            double cyc[2] = {e0, e1};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float I = (float)cyc[e];                     // prefix rounded to fp32 (torch.cumsum output)
                float frac = I - floorf(I);                  // I % 1
                float theta = __fmul_rn(6.2831854820251465f, frac);
                float hsin = __fmul_rn(sinf(theta), uv[e]);
                float amp = lerp_eval(ca[e], am[ca[e].i0], am[ca[e].i1]);
                o[e] = __fmul_rn(hsin, amp);
            }
            *reinterpret_cast<float2*>(source + ((long)b * 16 + m) * L + p0) = make_float2(o[0], o[1]);
        }
    }
}

// =================================================================================================
// Filtered noise (decoder.py:63-85): Y = exp(i angle) * kernel, zero frame prepended, rectangular
// window iSTFT (n_fft 1920, hop 480).  The per-frame c2r transform is a dense [1920 x 1922]
// contraction; overlap-add divides by the frame-count envelope and trims 960 samples per side.
// =================================================================================================
static __global__ void noise_spec_kernel(const float* __restrict__ kern, const float* __restrict__ angle,
                                         float* __restrict__ yri, long n_per_b, int B) {
    // yri[b][0..960][t] = cos(angle)*kernel, yri[b][961..1921][t] = sin(angle)*kernel
    long total = n_per_b * B;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long b = i / n_per_b, r = i - b * n_per_b;
        float a = angle[i], k = kern[i], sn, cs;
        sincosf(a, &sn, &cs);
        yri[b * 2 * n_per_b + r] = __fmul_rn(cs, k);
        yri[b * 2 * n_per_b + n_per_b + r] = __fmul_rn(sn, k);
    }
}

// counter-based uniform phases when the caller gives no `noise_angle`
static __global__ void angle_fill_kernel(float* __restrict__ angle, long n, uint64_t seed) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        float u = (float)(z >> 40) * (1.0f / 16777216.0f);  // [0, 1)
        angle[i] = u * 6.2831854820251465f - 3.1415927410125732f;
    }
}

static __global__ void noise_ola_kernel(const float* __restrict__ frames, float* __restrict__ source, int B, int T) {
    const long L = (long)T * kHop;
    long total = (long)B * L;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long b = i / L;
        int p = (int)(i - b * L);
        int q = p + kNfft / 2;  // position in the untrimmed signal; frame f (0..T) covers [480 f, 480 f + 1920)
        int f_hi = q / kHop;
        if (f_hi > T) f_hi = T;
        int f_lo = (q - (kNfft - 1) + kHop - 1) / kHop;
        if (q - (kNfft - 1) <= 0) f_lo = 0;
        float s = 0.f;
        for (int f = f_lo > 1 ? f_lo : 1; f <= f_hi; ++f)  // frame 0 is the zero pad (decoder.py:81)
            s = __fadd_rn(s, frames[((long)b * T + (f - 1)) * kNfft + (q - f * kHop)]);
        source[(b * 16 + 15) * L + p] = s / (float)(f_hi - f_lo + 1);
    }
}

// =================================================================================================
// SourceNet + dsp
// =================================================================================================
static int run_source_net(tvc_ctx* ctx, hipStream_t s, Ws& ws, bool dry, const float* content, const float* f0,
                          const float* energy, float* amps, float* kern, int B, int T) {
    const int ncols = B * T;
    float* ef = ws.get<float>((size_t)B * T);
    float* x = ws.get<float>((size_t)B * kSrcCh * T);
    if (!dry) {
        hipLaunchKernelGGL(window_max_kernel, dim3(grid_for((long)ncols * 64)), dim3(256), 0, s, energy, ef, (long)B, T, kHop);
        LoadPlain ld{content, kSslDim, T, (long)kSslDim * T};
        EpiSumCond ep{x, ctx->src_content_in.bias, ef, f0, ctx->src_e_w, ctx->src_e_b, ctx->src_f_w, ctx->src_f_b, kSrcCh, T, ncols};
        if (TVC_SPLIT_SRC && ctx->src_content_in.MT6 % 2 == 0)
            TVC_CHECK((gemm_s_launch<2, 4, 2>(ctx, s, ctx->src_content_in, content, B, kSslDim, T, 0, ep)));
        else
            igemm_launch(s, ctx->src_content_in.At, ctx->src_content_in.Mpad, ctx->src_content_in.Kpad, ncols, T, ld, ep);
    }
    for (int i = 0; i < 3; ++i) TVC_CHECK(run_convnext(ctx, s, ws, dry, ctx->src_mid[i], x, B, T));
    if (dry) return 0;
    LoadPlain ld{x, kSrcCh, T, (long)kSrcCh * T};
    EpiBias<ACT_ELU1, false> ea{amps, ctx->src_to_amps.bias, nullptr, kHarm, T, ncols, (long)kHarm * T, 0};
    igemm_launch(s, ctx->src_to_amps.At, ctx->src_to_amps.Mpad, ctx->src_to_amps.Kpad, ncols, T, ld, ea);
    EpiBias<ACT_ELU1, false> ek{kern, ctx->src_to_kernel.bias, nullptr, kBins, T, ncols, (long)kBins * T, 0};
    if (TVC_SPLIT_SRC && ctx->src_to_kernel.MT6 % 2 == 0)   // 128 -> 961 rows: the one sizeable contraction of the net, on the split path
        TVC_CHECK((gemm_s_launch<2, 4, 2>(ctx, s, ctx->src_to_kernel, x, B, kSrcCh, T, 0, ek)));
    else
        igemm_launch(s, ctx->src_to_kernel.At, ctx->src_to_kernel.Mpad, ctx->src_to_kernel.Kpad, ncols, T, ld, ek);
    return launch_check(ctx, "source_net");
}

// Decoder.dsp (decoder.py:259-266): f0 [B,1,T], amps [B,15,T], kernel [B,961,T] -> source [B,16,L]
int run_dsp(tvc_ctx* ctx, hipStream_t s, Ws& ws, bool dry, const float* f0, const float* amps, const float* kern,
            const float* angle, uint64_t seed, float* source, int B, int T) {
    const int ncols = B * T;
    const long L = (long)T * kHop;
    double* csum = ws.get<double>((size_t)B * kHarm * T);
    float* yri = ws.get<float>((size_t)B * 2 * kBins * T);
    float* frames = ws.get<float>((size_t)B * T * kNfft);
    float* ang = angle ? nullptr : ws.get<float>((size_t)B * kBins * T);
    if (dry) return 0;
    // harmonics -> source[:, 0:15]
    const float scale_size = (float)T / (float)L;         // F.interpolate(f0, Lw): size given
    const float scale_amp = (float)(1.0 / (double)kHop);  // F.interpolate(amps, scale_factor=480)
    // (independent of the noise branch below: different inputs, different rows of `source`) -> side stream, joined at the end
    hipStream_t sh = s;
    const bool fork = TVC_DSP_FORK && ctx->side;
    if (fork) {
        TVC_HIP(ctx, hipEventRecord(ctx->ev_fork, s));
        TVC_HIP(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
        sh = ctx->side;
    }
    hipLaunchKernelGGL(harm_frame_sum_kernel, dim3(T, B), dim3(256), 0, sh, f0, csum, T, scale_size);
    hipLaunchKernelGGL(harm_frame_scan_kernel, dim3(kHarm, B), dim3(64), 0, sh, csum, T);
    hipLaunchKernelGGL(harm_synth_kernel, dim3(T, B), dim3(256), 0, sh, f0, amps, csum, source, T, scale_size, scale_amp);
    if (fork) TVC_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->side));
    // noise -> source[:, 15]
    if (!angle) {
        hipLaunchKernelGGL(angle_fill_kernel, dim3(grid_for((long)B * kBins * T)), dim3(256), 0, s, ang, (long)B * kBins * T, seed);
        angle = ang;
    }
    if (TVC_FFT) {
        TVC_CHECK(run_noise_ifft(ctx, s, kern, angle, frames, B, T));
    } else {
    hipLaunchKernelGGL(noise_spec_kernel, dim3(grid_for((long)B * kBins * T)), dim3(256), 0, s, kern, angle, yri, (long)kBins * T, B);
    {   // even part from the real halves (rows 0..960 of yri), then odd part from the imaginary halves of bins 1..959
#if TVC_SPLIT_IDFT
        // split-precision path: K rows beyond 961 / 959 (up to 992 / 960, whole 32-channel slabs) meet zero weights and
        // stay inside yri (its imaginary half follows the real one)
        TVC_CHECK((gemm_s_launch<IDFT_MTB, IDFT_NWV, IDFT_BPC>(ctx, s, ctx->istft_e, yri, B, 992, T, (long)2 * kBins * T, EpiFramesPart<false>{frames, ncols})));
        TVC_CHECK((gemm_s_launch<IDFT_MTB, IDFT_NWV, IDFT_BPC>(ctx, s, ctx->istft_o, yri + (long)(kBins + 1) * T, B, 960, T, (long)2 * kBins * T,
                                                               EpiFramesPart<true>{frames, ncols})));
#else
        LoadPlain le{yri, kBins, T, (long)2 * kBins * T};
        EpiFramesPart<false> ee{frames, ncols};
        igemm_launch(s, ctx->istft_e.At, ctx->istft_e.Mpad, ctx->istft_e.Kpad, ncols, T, le, ee);
        LoadPlain lo{yri + (long)(kBins + 1) * T, kBins - 2, T, (long)2 * kBins * T};
        EpiFramesPart<true> eo{frames, ncols};
        igemm_launch(s, ctx->istft_o.At, ctx->istft_o.Mpad, ctx->istft_o.Kpad, ncols, T, lo, eo);
#endif
    }
    }
    hipLaunchKernelGGL(noise_ola_kernel, dim3(grid_for((long)B * L)), dim3(256), 0, s, frames, source, B, T);
    if (fork) TVC_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_join, 0));
    return launch_check(ctx, "dsp");
}

// =================================================================================================
// FilterNet (decoder.py:193-233) — round-1 form: one implicit-GEMM launch per Conv1d, with
// leaky_relu / replicate padding / FiLM / residuals fused into the loaders and epilogues, so HBM
// traffic is the layer-boundary model of SURVEY.md §8d (each conv reads its input, writes its output).
// =================================================================================================
template <int TAPS, bool LRELU, class Epi>
static void conv_launch(hipStream_t s, const PackedW& w, const float* x, int cin, int len, int dil, int B, const Epi& ep) {
    static_assert(TAPS == 3, "k3 convs only");
    LoadConv3<LRELU> ld{x, cin, len, dil, (long)cin * len};
    igemm_launch(s, w.At, w.Mpad, w.Kpad, B * len, len, ld, ep);
}

#ifndef TVC_SPLIT
#define TVC_SPLIT 1
#endif

#ifndef TVC_DOWN24_SPLIT
#define TVC_DOWN24_SPLIT 1   // Downsample 1 (24 -> 48 channels at 1/5 rate): c1, c2, c3 on the split-precision path (conv24s_kernel)
#endif
#ifndef TVC_DOWN0_SPLIT
#define TVC_DOWN0_SPLIT 1   // downs.0 (17 -> 24 channels at the full rate) on the split-precision path (filter_up24s.hip)
#endif
#ifndef TVC_UP24_SPLIT
#define TVC_UP24_SPLIT 1   // fused ups.4 + output layer on the split-precision bf16 MFMA path (filter_up24s.hip); 0 = fp32 16x16x4 tiles (filter_up24.hip)
#endif
#ifndef TVC_SPLIT48
#define TVC_SPLIT48 1   // 48-channel levels on the split path too (rows padded 48 -> 64; with the stacked FiLM phase ups.3 1.49 -> 1.25 ms)
#endif

int run_filter(tvc_ctx* ctx, hipStream_t s, Ws& ws, bool dry, const float* content, const float* f0,
               const float* energy, const float* source, float* wave, int B, int T, const FilterTaps* taps) {
    const long L = (long)T * kHop;
    static const int ch[5] = {384, 192, 96, 48, 24};
    // level lengths: skip i lives at len_dn[i]
    long len_dn[5] = {L, L / 5, L / 20, L / 80, L / 240};
    float* skip[5];
    for (int i = 0; i < 5; ++i) skip[i] = ws.get<float>((size_t)B * ch[4 - i] * len_dn[i]);
    float* x = ws.get<float>((size_t)B * ch[0] * T);
    // Downsample inputs produced by the previous block's last conv (its epilogue also writes the 1/f-rate copy),
    // instead of a separate interpolate pass that re-reads the full-rate skip tensor
    float* xi_pre[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    bool xi_fused[5] = {false, false, false, false, false};
    for (int i = 1; i <= 4; ++i) {
        const DownW& d = ctx->downs[i - 1];
        const bool producer_ok = i == 1 ? (TVC_USE_C48 != 0 || TVC_DOWN0_SPLIT != 0)                                     // downs.0 conv (16x16x4 kernel), factor 5: pick
                                        : (((TVC_SPLIT && ctx->downs[i - 2].cin % 16 == 0 && ctx->downs[i - 2].cout % 96 == 0) ||   // conv3s c3 of the block before
                                            (TVC_DOWN24_SPLIT && ctx->downs[i - 2].cin == 24 && ctx->downs[i - 2].cout == 48 && d.factor == 4)) &&   // conv24s c3
                                           len_dn[i - 1] % 4 == 0);
        if (TVC_FUSE_DECIM && producer_ok && ((d.factor == 5 && i == 1) || ((d.factor == 3 || d.factor == 4) && i > 1)) && len_dn[i - 1] % d.factor == 0) {
            xi_pre[i] = ws.get<float>((size_t)B * d.cin * len_dn[i]);
            xi_fused[i] = true;
        }
    }

    if (!dry) {
        ProfScope ps(ctx, s, dry, "filter.in+down0");
        LoadPlain ld{content, kSslDim, T, (long)kSslDim * T};
        EpiSumCond ep{x, ctx->flt_content_in.bias, nullptr, f0, nullptr, nullptr, ctx->flt_f_w, ctx->flt_f_b, ch[0], T, B * T};
        if (TVC_SPLIT_SRC && ctx->flt_content_in.MT6 % 2 == 0)
            TVC_CHECK((gemm_s_launch<2, 4, 2>(ctx, s, ctx->flt_content_in, content, B, kSslDim, T, 0, ep)));
        else
            igemm_launch(s, ctx->flt_content_in.At, ctx->flt_content_in.Mpad, ctx->flt_content_in.Kpad, B * T, T, ld, ep);
        if (TVC_DOWN0_SPLIT && (!xi_fused[1] || L % 5 == 0))   // downs[0] on the split-precision path; its epilogue also writes Downsample 1's 1/5-rate input
            TVC_CHECK(run_down0_split(ctx, s, ctx->flt_down0s, source, energy, skip[0], xi_fused[1] ? xi_pre[1] : nullptr, B, (int)L));
        else if (TVC_USE_C48)   // downs[0]: k3 conv over cat[source (16 ch), energy (1 ch)], read from the two tensors in place
            conv3mt_launch<2, false>(s, ctx->flt_down0, source, B, 17, (int)L, 1,
                                     C3EpiBias<false, 5>{skip[0], ctx->flt_down0.bias, nullptr, 24, (int)L, xi_fused[1] ? xi_pre[1] : nullptr,
                                                         xi_fused[1] ? 5 : 0},
                                     FilmOps(), energy, 16);
        else
            TVC_CHECK(run_down0(ctx, s, ctx->flt_down0, source, energy, skip[0], B, (int)L));
    }
    // down path
    for (int i = 1; i <= 4; ++i) {
        const DownW& d = ctx->downs[i - 1];
        const int lin = (int)len_dn[i - 1], len = (int)len_dn[i];
        size_t mk = ws.mark();
        float* xi = xi_fused[i] ? xi_pre[i] : ws.get<float>((size_t)B * d.cin * len);
        float* res = ws.get<float>((size_t)B * d.cout * len);
        float* h1 = ws.get<float>((size_t)B * d.cin * len);
        float* h2 = ws.get<float>((size_t)B * d.cin * len);
        if (!dry) {
            static const char* names[4] = {"filter.down1", "filter.down2", "filter.down3", "filter.down4"};
            ProfScope ps(ctx, s, dry, names[i - 1]);
            // F.interpolate(scale_factor=1/f): ATen uses scale = 1/(1/f) = f
            if (!xi_fused[i]) {
                const LerpLaunch ll = lerp_launch((long)B * d.cin, len);
                hipLaunchKernelGGL(lerp_resize_kernel, ll.grid, dim3(256), 0, s, skip[i - 1], xi, (long)B * d.cin, lin, len, (float)d.factor, ll.tx);
            }
            const int nc = B * len;
            // c3 on the generic split kernel folds down_res(xi) in as a second K phase: no residual tensor, no launch for it
            const bool d24s = TVC_DOWN24_SPLIT && d.cin == 24 && d.cout == 48;
            const bool resconv = TVC_RESCONV && ((TVC_SPLIT && d.cin % 16 == 0 && d.cout % 96 == 0 && d.res.MT6 == d.c3.MT6 && d.c3res_bias != nullptr) ||
                                                 (d24s && d.s24c3r != nullptr && d.res.MT6 == 2));
            if (!resconv) {
                EpiBias<ACT_NONE, false> ep{res, d.res.bias, nullptr, d.cout, len, nc, (long)d.cout * len, 0};
                if (TVC_SPLIT_1X1 && d.cin % 16 == 0 && d.res.MT6 % 3 == 0) {
                    TVC_CHECK((gemm_s_launch<3, 4, 1>(ctx, s, d.res, xi, B, d.cin, len, 0, ep)));
                } else {
                    LoadPlain ld{xi, d.cin, len, (long)d.cin * len};
                    igemm_launch(s, d.res.At, d.res.Mpad, d.res.Kpad, nc, len, ld, ep);
                }
            }
            if (d24s) {   // the whole 24-channel block on the split-precision path; c3's epilogue adds res and writes the next block's 1/4-rate input
                TVC_CHECK(run_down24_split(ctx, s, d, xi, resconv ? nullptr : res, h1, h2, skip[i], (i < 4 && xi_fused[i + 1]) ? xi_pre[i + 1] : nullptr, B, len));
            } else if (d.cin == 24 && TVC_USE_C48) {   // 24 output channels = two 16-row tiles, many small waves
                conv3mt_launch<2, true>(s, d.c1, xi, B, d.cin, len, 1, C3EpiBias<false>{h1, d.c1.bias, nullptr, d.cin, len});
                conv3mt_launch<2, true>(s, d.c2, h1, B, d.cin, len, 2, C3EpiBias<false>{h2, d.c2.bias, nullptr, d.cin, len});
            } else if (d.cin == 48 && TVC_C48R) {   // weights resident in LDS, one staging round trip per tile (conv48s.hip)
                TVC_CHECK(run_conv48s(ctx, s, d.c1, xi, 0, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.f, h1, B, len, 1));
                TVC_CHECK(run_conv48s(ctx, s, d.c2, h1, 0, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.f, h2, B, len, 2));
            } else if (d.cin == 48 && TVC_SPLIT48) {
                TVC_CHECK(conv3s_launch<true>(ctx, s, d.c1, xi, B, d.cin, len, 1, C3EpiBias<false>{h1, d.c1.bias, nullptr, d.cin, len}));
                TVC_CHECK(conv3s_launch<true>(ctx, s, d.c2, h1, B, d.cin, len, 2, C3EpiBias<false>{h2, d.c2.bias, nullptr, d.cin, len}));
            } else if (d.cin == 48 && TVC_USE_C48) {   // 48 = 3 x 16: the 16x16x4 kernel has no row padding
                conv3m48_launch<true>(s, d.c1, xi, B, d.cin, len, 1, C3EpiBias<false>{h1, d.c1.bias, nullptr, d.cin, len});
                conv3m48_launch<true>(s, d.c2, h1, B, d.cin, len, 2, C3EpiBias<false>{h2, d.c2.bias, nullptr, d.cin, len});
            } else if (TVC_SPLIT && d.cin % 96 == 0) {   // bf16x3 split path, 16x the fp32 MFMA rate per part-product
                TVC_CHECK(conv3s_launch<true>(ctx, s, d.c1, xi, B, d.cin, len, 1, C3EpiBias<false>{h1, d.c1.bias, nullptr, d.cin, len}));
                TVC_CHECK(conv3s_launch<true>(ctx, s, d.c2, h1, B, d.cin, len, 2, C3EpiBias<false>{h2, d.c2.bias, nullptr, d.cin, len}));
            } else {
                conv3_launch<true>(s, d.c1.At, d.c1.Mpad, xi, B, d.cin, len, 1, C3EpiBias<false>{h1, d.c1.bias, nullptr, d.cin, len});
                conv3_launch<true>(s, d.c2.At, d.c2.Mpad, h1, B, d.cin, len, 2, C3EpiBias<false>{h2, d.c2.bias, nullptr, d.cin, len});
            }
            if (d24s) {
            } else if (d.cout == 48 && d.cin % 16 == 0 && TVC_SPLIT48 >= 2)
                TVC_CHECK(conv3s_launch<true>(ctx, s, d.c3, h2, B, d.cin, len, 4, C3EpiBias<true>{skip[i], d.c3.bias, res, d.cout, len}));
            else if (d.cout == 48 && TVC_USE_C48)
                conv3m48_launch<true>(s, d.c3, h2, B, d.cin, len, 4, C3EpiBias<true>{skip[i], d.c3.bias, res, d.cout, len});
            else if (resconv)
                TVC_CHECK((conv3s_launch<true, C3EpiBiasResConv>(ctx, s, d.c3, h2, B, d.cin, len, 4,
                                                                 C3EpiBiasResConv{{skip[i], d.c3res_bias, nullptr, d.cout, len,
                                                                                   (i < 4 && xi_fused[i + 1]) ? xi_pre[i + 1] : nullptr,
                                                                                   (i < 4 && xi_fused[i + 1]) ? ctx->downs[i].factor : 0}},
                                                                 &d.res, nullptr, xi, d.cin)));
            else if (TVC_SPLIT && d.cin % 16 == 0 && d.cout % 96 == 0)
                TVC_CHECK(conv3s_launch<true>(ctx, s, d.c3, h2, B, d.cin, len, 4,
                                              C3EpiBias<true>{skip[i], d.c3.bias, res, d.cout, len, (i < 4 && xi_fused[i + 1]) ? xi_pre[i + 1] : nullptr,
                                                              (i < 4 && xi_fused[i + 1]) ? ctx->downs[i].factor : 0}));
            else
                conv3_launch<true>(s, d.c3.At, d.c3.Mpad, h2, B, d.cin, len, 4, C3EpiBias<true>{skip[i], d.c3.bias, res, d.cout, len});
        }
        ws.release(mk);
    }
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
    bool fused_out = false;
    for (int i = 0; i < 5; ++i) {
        const UpW& u = ctx->ups[i];
        const int lin = (int)len;
        len *= u.factor;
        const int lo = (int)len, C = u.cin, nc = B * lo;
        const float* cond = skip[4 - i];
        size_t mk = ws.mark();
        float* xu = ws.get<float>((size_t)B * C * lo);
        float* film = (C < 96 && C != 24 && !(C == 48 && (TVC_USE_C48 + TVC_SPLIT48 > 0))) ? ws.get<float>((size_t)B * 2 * C * lo) : nullptr;
        float* h = ws.get<float>((size_t)B * C * lo);
        float* x1 = ws.get<float>((size_t)B * C * lo);
        if (!dry && C == 24) {
            // last level: Upsample block + output_layer in two launches, waveform written directly
            ProfScope ps(ctx, s, dry, "filter.up4+out");
            if (TVC_UP24_SPLIT) TVC_CHECK(run_up24_split(ctx, s, u, x, cond, x1, wave, B, lo));
            else TVC_CHECK(run_up24_fused(ctx, s, u, x, cond, x1, wave, B, lo, ctx->flt_out_w, ctx->flt_out_b));
            fused_out = true;
        } else if (!dry) {
            static const char* names[5] = {"filter.up0", "filter.up1", "filter.up2", "filter.up3", "filter.up4"};
            ProfScope ps(ctx, s, dry, names[i]);
            // F.interpolate(scale_factor=f): ATen uses scale = float(1/f).  On the split path the interpolated tensor is never
            // written: c1 interpolates while it stages its input and c2's epilogue interpolates the residual (x_up).
            const float lscale = (float)(1.0 / (double)u.factor);
            const bool split_level = (C == 48 && TVC_SPLIT48) || (TVC_SPLIT && C % 96 == 0);
            const bool lerp_fused = TVC_FUSE_LERP && split_level;
            if (!lerp_fused) {
                const LerpLaunch ll = lerp_launch((long)B * C, lo);
                hipLaunchKernelGGL(lerp_resize_kernel, ll.grid, dim3(256), 0, s, x, xu, (long)B * C, lin, lo, lscale, ll.tx);
            }
            const bool c5_fused = TVC_C48R && TVC_C48_C5 && split_level && C == 48 && u.cout == 24 && u.c5.MT6 == 1;
            for (int half = 0; half < 2; ++half) {
                const PackedW& ca = half ? u.c3 : u.c1;
                const PackedW& cb = half ? u.c4 : u.c2;
                const int da = half ? 9 : 1, db = half ? 27 : 3;
                const float* xin = half ? x1 : xu;
                float* xout = half ? xu : x1;  // 2nd half writes over xu (its input and residual are x1)
                const PackedW& wsc = half ? u.sc2 : u.sc1;
                const PackedW& wsh = half ? u.sh2 : u.sh1;
                if (split_level && C == 48 && TVC_C48R) {   // the 48-channel level with its weights resident in LDS (conv48s.hip)
                    const PackedW& fw = half ? u.film2 : u.film1;
                    if (half == 0 && lerp_fused) {
                        TVC_CHECK(run_conv48s(ctx, s, ca, x, lin, lscale, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.f, h, B, lo, da));
                        TVC_CHECK(run_conv48s(ctx, s, cb, h, 0, 0.f, &fw, wsc.bias, wsh.bias, cond, x, lin, lscale, xout, B, lo, db));
                    } else {
                        TVC_CHECK(run_conv48s(ctx, s, ca, xin, 0, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.f, h, B, lo, da));
                        if (half == 1 && c5_fused)   // c4 + FiLM2 + residual + c5 in one launch: the level's output is written directly
                            TVC_CHECK(run_conv48s(ctx, s, cb, h, 0, 0.f, &fw, wsc.bias, wsh.bias, cond, xin, 0, 0.f, nullptr, B, lo, db, &u.c5, xlev[i]));
                        else
                            TVC_CHECK(run_conv48s(ctx, s, cb, h, 0, 0.f, &fw, wsc.bias, wsh.bias, cond, xin, 0, 0.f, xout, B, lo, db));
                    }
                    continue;
                }
                if (split_level) {
                    const PackedW& fw = half ? u.film2 : u.film1;      // stacked [to_scale ; to_shift] rows, each group padded to whole 32-row tiles
                    if (half == 0 && lerp_fused) {
                        TVC_CHECK((conv3s_launch<true, C3EpiBias<false>, false, true>(ctx, s, ca, x, B, C, lo, da, C3EpiBias<false>{h, ca.bias, nullptr, C, lo},
                                                                                      nullptr, nullptr, nullptr, 0, lin, lscale)));
                        TVC_CHECK((conv3s_launch<true, C3EpiFilmFused, true>(ctx, s, cb, h, B, C, lo, db,
                                                                              C3EpiFilmFused{xout, cb.bias, wsc.bias, wsh.bias, x, C, lo, lin, lscale},
                                                                              &fw, &fw, cond, C)));
                    } else {
                        TVC_CHECK(conv3s_launch<true>(ctx, s, ca, xin, B, C, lo, da, C3EpiBias<false>{h, ca.bias, nullptr, C, lo}));
                        TVC_CHECK((conv3s_launch<true, C3EpiFilmFused, true>(ctx, s, cb, h, B, C, lo, db,
                                                                              C3EpiFilmFused{xout, cb.bias, wsc.bias, wsh.bias, xin, C, lo},
                                                                              &fw, &fw, cond, C)));
                    }
                    continue;
                }
                if (C == 48 && TVC_USE_C48) {
                    // 48-channel level on 16x16x4 tiles (no row padding), FiLM and residual fused
                    conv3m48_launch<true>(s, ca, xin, B, C, lo, da, C3EpiBias<false>{h, ca.bias, nullptr, C, lo});
                    conv3m48_launch<true, C3EpiFilmFused, true>(s, cb, h, B, C, lo, db,
                                                                C3EpiFilmFused{xout, cb.bias, wsc.bias, wsh.bias, xin, C, lo},
                                                                FilmOps{wsc.At, wsh.At, cond, C});
                    continue;
                }
                conv3_launch<true>(s, ca.At, ca.Mpad, xin, B, C, lo, da, C3EpiBias<false>{h, ca.bias, nullptr, C, lo});
                if (C >= 96) {
                    // second conv with FiLM(cond) and the residual fused: scale/shift never touch HBM
                    // (measured time-neutral at C >= 96, and it removes the [B][2C][len] film tensor)
                    conv3_launch<true, C3EpiFilmFused, true>(s, cb.At, cb.Mpad, h, B, C, lo, db,
                                                             C3EpiFilmFused{xout, cb.bias, wsc.bias, wsh.bias, xin, C, lo},
                                                             FilmOps{wsc.At, wsh.At, cond, C});
                } else {
                    // C = 48: three live accumulator sets cost more occupancy than the film round-trip (measured)
                    const PackedW& fw = half ? u.film2 : u.film1;
                    LoadPlain ld{cond, C, lo, (long)C * lo};
                    EpiBias<ACT_NONE, false> ep{film, fw.bias, nullptr, 2 * C, lo, nc, (long)2 * C * lo, 0};
                    igemm_launch(s, fw.At, fw.Mpad, fw.Kpad, nc, lo, ld, ep);
                    conv3_launch<true>(s, cb.At, cb.Mpad, h, B, C, lo, db, C3EpiFilm{xout, cb.bias, film, xin, C, lo});
                }
            }
            EpiBias<ACT_NONE, false> ep{xlev[i], u.c5.bias, nullptr, u.cout, lo, nc, (long)u.cout * lo, 0};
            if (c5_fused) {
            } else if (TVC_SPLIT_1X1 && C % 16 == 0 && u.c5.MT6 % 3 == 0) {
                TVC_CHECK((gemm_s_launch<3, 4, 1>(ctx, s, u.c5, xu, B, C, lo, 0, ep)));
            } else if (TVC_SPLIT_1X1 && C % 16 == 0 && u.c5.MT6 % 2 == 0) {
                TVC_CHECK((gemm_s_launch<2, 4, 2>(ctx, s, u.c5, xu, B, C, lo, 0, ep)));
            } else {
                LoadPlain ld{xu, C, lo, (long)C * lo};
                igemm_launch(s, u.c5.At, u.c5.Mpad, u.c5.Kpad, nc, lo, ld, ep);
            }
        }
        ws.release(mk);
        x = xlev[i];
    }
    if (!dry && !fused_out) {
        ProfScope ps(ctx, s, dry, "filter.out");
        TVC_CHECK(run_out_conv7(ctx, s, x, ctx->flt_out_w, ctx->flt_out_b, wave, B, 24, (int)L));
    }
    if (!dry && taps) {   // parity taps: the block outputs are still live in the workspace
        for (int i = 0; i < 5; ++i)
            if (taps->skips[i])
                TVC_HIP(ctx, hipMemcpyAsync(taps->skips[i], skip[i], (size_t)B * ch[4 - i] * len_dn[i] * sizeof(float), hipMemcpyDeviceToDevice, s));
        long l = T;
        for (int i = 0; i < 4; ++i) {
            l *= ctx->ups[i].factor;
            if (taps->ups[i])
                TVC_HIP(ctx, hipMemcpyAsync(taps->ups[i], xlev[i], (size_t)B * ctx->ups[i].cout * l * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
    }
    return dry ? 0 : launch_check(ctx, "filter_net");
}

int run_decoder(tvc_ctx* ctx, hipStream_t s, Ws& ws, bool dry, const float* content, const float* f0,
                const float* energy, const float* angle, uint64_t seed, float* wave, float* amps_out,
                float* kernel_out, float* source_out, int B, int T) {
    const long L = (long)T * kHop;
    float* amps = amps_out ? amps_out : ws.get<float>((size_t)B * kHarm * T);
    float* kern = kernel_out ? kernel_out : ws.get<float>((size_t)B * kBins * T);
    float* source = source_out ? source_out : ws.get<float>((size_t)B * 16 * L);
    size_t mk = ws.mark();
    {
        ProfScope ps(ctx, s, dry, "source_net");
        TVC_CHECK(run_source_net(ctx, s, ws, dry, content, f0, energy, amps, kern, B, T));
    }
    ws.release(mk);
    {
        ProfScope ps(ctx, s, dry, "dsp");
        TVC_CHECK(run_dsp(ctx, s, ws, dry, f0, amps, kern, angle, seed, source, B, T));
    }
    ws.release(mk);
    ProfScope ps(ctx, s, dry, "filter_net");
    TVC_CHECK(run_filter(ctx, s, ws, dry, content, f0, energy, source, wave, B, T));
    ws.release(mk);
    return 0;
}

}  // namespace tvc
