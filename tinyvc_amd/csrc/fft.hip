// 1920-point real FFTs for the two transforms of the path: |STFT| of the input (spectrogram.py:8-15) and the inverse
// transform of the filtered-noise synthesis (decoder.py:63-85).  A frame is 1920 real samples = one complex FFT of 960
// points on z[m] = x[2m] + i x[2m+1] plus the usual even/odd untangling; 960 = 64 x 15:
//   * a wavefront owns a frame; lane n1 holds z[64 n2 + n1] for n2 = 0..14 (coalesced loads);
//   * 15-point DFT over n2 in registers (3 x 5 Cooley-Tukey), twiddle W_960^(n1 k2);
//   * 64-point radix-2 DIF FFT across the lanes for each of the 15 values (cross-lane exchanges, the six stage
//     twiddles are lane constants) -> lane l holds bin 15 * bitrev6(l) + k2;
//   * untangling / magnitude (forward) or linearisation (inverse) through the wave's LDS row.
// 8 waves = 8 consecutive frames of one utterance per workgroup, so the [961][T] spectrogram is written / read in
// 32-byte runs along time through an LDS transpose.  ~0.1 MFLOP per frame instead of the 3.7 MFLOP of the half-size
// real-DFT GEMMs this replaces (frontend.hip, decoder.hip keep those behind -DTVC_FFT=0).
// Twiddles come from fp64-computed tables (api.hip build_dft_tables): tw960[j] = (cos, sin)(2 pi j / 960),
// tw1920[k] = (cos, sin)(2 pi k / 1920), hann[n] = 0.5 - 0.5 cos(2 pi n / 1920).
#include "small_kernels.h"
#include "tvc_common.h"

namespace tvc {

namespace {

constexpr int kM = 960;          // complex FFT size
constexpr int kFW = 8;           // frames (waves) per workgroup

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// a * (c + i SIGN s)
template <int SIGN>
__device__ __forceinline__ float2 cmul_tw(float2 a, float2 w) {
    const float s = SIGN < 0 ? -w.y : w.y;
    return make_float2(fmaf(a.x, w.x, -a.y * s), fmaf(a.x, s, a.y * w.x));
}
// multiply by SIGN * i
template <int SIGN>
__device__ __forceinline__ float2 mul_i(float2 a) { return SIGN < 0 ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x); }

// 3-point DFT in place: y[k] = sum_n x[n] e^(SIGN 2 pi i n k / 3)
template <int SIGN>
__device__ __forceinline__ void dft3(float2& a, float2& b, float2& c) {
    const float s = 0.86602540378443864676f;
    const float2 t = cadd(b, c), d = csub(b, c);
    const float2 m = make_float2(fmaf(-0.5f, t.x, a.x), fmaf(-0.5f, t.y, a.y));
    const float2 js = mul_i<SIGN>(make_float2(s * d.x, s * d.y));     // SIGN i s (b - c)
    a = cadd(a, t);
    b = cadd(m, js);
    c = csub(m, js);
}
// 5-point DFT in place
template <int SIGN>
__device__ __forceinline__ void dft5(float2& x0, float2& x1, float2& x2, float2& x3, float2& x4) {
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;    // cos(2 pi / 5), cos(4 pi / 5)
    const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;     // sin(2 pi / 5), sin(4 pi / 5)
    const float2 t1 = cadd(x1, x4), t2 = cadd(x2, x3), t3 = csub(x1, x4), t4 = csub(x2, x3);
    const float2 m1 = make_float2(fmaf(c1, t1.x, fmaf(c2, t2.x, x0.x)), fmaf(c1, t1.y, fmaf(c2, t2.y, x0.y)));
    const float2 m2 = make_float2(fmaf(c2, t1.x, fmaf(c1, t2.x, x0.x)), fmaf(c2, t1.y, fmaf(c1, t2.y, x0.y)));
    const float2 u1 = mul_i<SIGN>(make_float2(fmaf(s1, t3.x, s2 * t4.x), fmaf(s1, t3.y, s2 * t4.y)));   // SIGN i (s1 t3 + s2 t4)
    const float2 u2 = mul_i<SIGN>(make_float2(fmaf(s2, t3.x, -s1 * t4.x), fmaf(s2, t3.y, -s1 * t4.y)));  // SIGN i (s2 t3 - s1 t4)
    x0 = cadd(x0, cadd(t1, t2));
    x1 = cadd(m1, u1);
    x4 = csub(m1, u1);
    x2 = cadd(m2, u2);
    x3 = csub(m2, u2);
}

// register slot of output index k2 of the 15-point DFT below
__device__ __forceinline__ constexpr int slot15(int k2) { return 5 * (k2 % 3) + k2 / 3; }

// Complex FFT of 960 points held by one wavefront.  In: v[n2] = z[64 n2 + lane].  Out: v[slot15(k2)] = Z[15 bitrev6(lane) + k2]
// with Z[k] = sum_m z[m] e^(SIGN 2 pi i m k / 960) (unnormalised).
template <int SIGN>
__device__ __forceinline__ void fft960_wave(float2 (&v)[15], int lane, const float2* __restrict__ tw960) {
    // 15-point DFT over n2 = 5 na + nb -> k2 = ka + 3 kb
#pragma unroll
    for (int nb = 0; nb < 5; ++nb) dft3<SIGN>(v[nb], v[5 + nb], v[10 + nb]);          // v[5 ka + nb]
#pragma unroll
    for (int ka = 1; ka < 3; ++ka)
#pragma unroll
        for (int nb = 1; nb < 5; ++nb) v[5 * ka + nb] = cmul_tw<SIGN>(v[5 * ka + nb], tw960[64 * nb * ka]);   // W_15^(nb ka)
#pragma unroll
    for (int ka = 0; ka < 3; ++ka) dft5<SIGN>(v[5 * ka], v[5 * ka + 1], v[5 * ka + 2], v[5 * ka + 3], v[5 * ka + 4]);   // v[5 ka + kb]
    // twiddle W_960^(n1 k2), n1 = lane
#pragma unroll
    for (int k2 = 1; k2 < 15; ++k2) v[slot15(k2)] = cmul_tw<SIGN>(v[slot15(k2)], tw960[lane * k2]);
    // 64-point DIF FFT across the lanes
#pragma unroll
    for (int h = 32; h >= 1; h >>= 1) {
        const bool lower = (lane & h) != 0;
        const float2 w = tw960[(480 / h) * (lane & (h - 1))];                          // W_(2h)^(lane mod h)
#pragma unroll
        for (int i = 0; i < 15; ++i) {
            const float2 p = make_float2(__shfl_xor(v[i].x, h), __shfl_xor(v[i].y, h));
            const float2 sum = cadd(v[i], p), dif = csub(p, v[i]);                      // lower lanes: partner is the upper element
            v[i] = lower ? (h > 1 ? cmul_tw<SIGN>(dif, w) : dif) : sum;
        }
    }
}

__device__ __forceinline__ int bitrev6(int l) { return (int)(__brev((unsigned)l) >> 26); }

// ---------------------------------------------------------------------------------------------------------------------
// spec[b][k][t] = | sum_n hann[n] x_b[(t + 1) * 480 - 960 + n (reflected)] e^(-2 pi i k n / 1920) |, k = 0..960, t = 0..T-1
__global__ __launch_bounds__(kFW * 64) void stft_fft_kernel(const float* __restrict__ wav, float* __restrict__ spec, const float2* __restrict__ tw960,
                                                            const float2* __restrict__ tw1920, const float* __restrict__ hann, int L, int T, RagDev rg) {
    extern __shared__ __attribute__((aligned(16))) float2 smem_f[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float2* Zb = smem_f + wave * kM;                                   // this wave's 960 bins
    float* Os = reinterpret_cast<float*>(smem_f);                      // [961][kFW] magnitudes, over the bins once every wave has read its own (61 KB: two workgroups per CU)
    // ragged batch (ragged.h): utterance b reads its own row of the caller's padded [rows][Tmax * 480] tensor over its own length (the
    // reflection is at ITS end) and writes columns pre[b] ... of the [961][T] spectrogram of the whole batch (T = row stride)
    const int rs = T;
    const int groups = rg.tb ? (int)gridDim.x / rg.B : (T + kFW - 1) / kFW;
    const int b = blockIdx.x / groups, t0 = (blockIdx.x - b * groups) * kFW;
    long wbase = (long)b * L, sbase = (long)b * kBins * T;
    if (rg.tb) {
        T = rg.tb[b];
        L = T * kHop;
        wbase = (long)rg.row[b] * rg.Tmax * kHop;
        sbase = rg.pre[b];
        if (t0 >= T) return;
    }
    const int t = t0 + wave;
    if (t < T) {
        const float* wb = wav + wbase;
        const int start = (t + 1) * kHop - kNfft / 2;                  // frame t + 1 of the centred STFT (frame 0 is dropped)
        float2 v[15];
#pragma unroll
        for (int n2 = 0; n2 < 15; ++n2) {
            const int n = 2 * (64 * n2 + lane);
            int p0 = start + n, p1 = p0 + 1;
            if (p0 < 0) p0 = -p0;
            if (p0 >= L) p0 = 2 * (L - 1) - p0;
            if (p1 < 0) p1 = -p1;
            if (p1 >= L) p1 = 2 * (L - 1) - p1;
            const float2 w = *reinterpret_cast<const float2*>(hann + n);
            v[n2] = make_float2(__fmul_rn(wb[p0], w.x), __fmul_rn(wb[p1], w.y));
        }
        fft960_wave<-1>(v, lane, tw960);
        const int k1 = bitrev6(lane);
#pragma unroll
        for (int k2 = 0; k2 < 15; ++k2) Zb[15 * k1 + k2] = v[slot15(k2)];
    }
    __syncthreads();
    constexpr int NK = (kM + 64) / 64;                                 // bins per lane: k = lane + 64 i <= 960
    float mag[NK];
    if (t < T) {
        // X[k] = (Z[k] + conj Z[M-k]) / 2 - i W_1920^k (Z[k] - conj Z[M-k]) / 2
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            const int k = lane + 64 * i;
            mag[i] = 0.f;
            if (k <= kM) {
                const float2 a = Zb[k == kM ? 0 : k], c = Zb[k == 0 ? 0 : kM - k];
                const float2 e = make_float2(0.5f * (a.x + c.x), 0.5f * (a.y - c.y));
                const float2 o = make_float2(0.5f * (a.x - c.x), 0.5f * (a.y + c.y));
                const float2 w = tw1920[k];                                // (cos, sin): W = cos - i sin
                const float re = e.x + fmaf(w.x, o.y, -w.y * o.x);         // -i W o = (c o.y - s o.x, -(c o.x + s o.y))
                const float im = e.y - fmaf(w.x, o.x, w.y * o.y);
                mag[i] = sqrtf(fmaf(re, re, im * im));
            }
        }
    }
    __syncthreads();                                                   // every wave has read its bins: the tile may be overwritten
    if (t < T) {
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            const int k = lane + 64 * i;
            if (k <= kM) Os[k * kFW + wave] = mag[i];
        }
    }
    __syncthreads();
    const int nf = T - t0 < kFW ? T - t0 : kFW;
    float* sb = spec + sbase + t0;
    if (nf == kFW && (rs & 3) == 0 && ((sbase + t0) & 3) == 0) {
        for (int i = tid; i < kBins * (kFW / 4); i += kFW * 64) {
            const int k = i / (kFW / 4), q = (i - k * (kFW / 4)) * 4;
            *reinterpret_cast<float4*>(sb + (long)k * rs + q) = *reinterpret_cast<const float4*>(Os + k * kFW + q);
        }
    } else {
        for (int i = tid; i < kBins * kFW; i += kFW * 64) {
            const int k = i / kFW, f = i - k * kFW;
            if (f < nf) sb[(long)k * rs + f] = Os[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// frames[(b T + t)][n] = irfft_1920(kernel[b][:, t] * exp(i angle[b][:, t]))[n]  (torch.fft.irfft semantics: 1/N, the
// imaginary parts of bins 0 and 960 do not enter)
// angle == nullptr: the phases are the library's own draw, noise_phase_hash(seed, utterance row, bin, frame) (small_kernels.h), evaluated here
template <bool DRAW>
__global__ __launch_bounds__(kFW * 64) void noise_ifft_kernel(const float* __restrict__ kern, const float* __restrict__ angle, float* __restrict__ frames,
                                                              const float2* __restrict__ tw960, const float2* __restrict__ tw1920, int T, RagDev rg,
                                                              unsigned long long seed, const int* __restrict__ rowmap) {
    extern __shared__ __attribute__((aligned(16))) float2 smem_f[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int YS = kBins + 1;                                      // row stride (962 float2: rows stay 16-byte aligned)
    // ragged batch (ragged.h): `kern` is [961][T] over the whole batch (utterance b at columns pre[b] ...), `angle` the caller's padded
    // [rows][961][Tmax] tensor (as = its row stride), the frames of utterance b land behind those of the utterances before it
    const int rs = T;
    int as = T;
    const int groups = rg.tb ? (int)gridDim.x / rg.B : (T + kFW - 1) / kFW;
    const int b = blockIdx.x / groups, t0 = (blockIdx.x - b * groups) * kFW;
    long kbase = (long)b * kBins * T, abase = kbase, fbase = (long)b * T;
    if (rg.tb) {
        T = rg.tb[b];
        kbase = fbase = rg.pre[b];
        if (rg.row) {
            as = rg.Tmax;
            abase = (long)rg.row[b] * kBins * rg.Tmax;
        } else {
            abase = kbase;           // the library's own draw: laid out like `kern`
        }
        if (t0 >= T) return;
    }
    const int nf = T - t0 < kFW ? T - t0 : kFW;
    {   // Y[f][k] = kernel * (cos, sin)(angle): the [961][T] tensors are read in runs of kFW frames
        const float* kb = kern + kbase + t0;
        const float* ab = angle + abase + t0;
        // a thread's 16 elements are requested together (clamped addresses, no branch in front of the loads), then converted
        constexpr int PER = (kBins * kFW + kFW * 64 - 1) / (kFW * 64);
        float av[PER], kv[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            int i = tid + j * kFW * 64;
            i = i < kBins * kFW ? i : kBins * kFW - 1;
            const int k = i / kFW;
            int f = i - k * kFW;
            f = f < nf ? f : nf - 1;
            av[j] = DRAW ? noise_phase_hash(seed, rowmap ? rowmap[b] : b, k, t0 + f) : ab[(long)k * as + f];
            kv[j] = kb[(long)k * rs + f];
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int i = tid + j * kFW * 64;
            const int k = i / kFW, f = i - k * kFW;
            float2 sc = sincos_small(av[j]);             // (sin, cos); the phases are drawn in [-pi, pi)
            if (!(fabsf(av[j]) <= 6.2831855f)) sincosf(av[j], &sc.x, &sc.y);      // a caller's own phases may be anything (NaN included): libm's full range
            if (i < kBins * kFW && f < nf) smem_f[f * YS + k] = make_float2(__fmul_rn(sc.y, kv[j]), __fmul_rn(sc.x, kv[j]));
        }
    }
    __syncthreads();
    float2* Y = smem_f + wave * YS;
    const int t = t0 + wave;
    float2 v[15];
    if (t < T) {
        // Z[k] = E[k] + i O[k],  E = Y[k] + conj Y[M-k],  O = (Y[k] - conj Y[M-k]) e^(+2 pi i k / 1920),  k = 0..959
#pragma unroll
        for (int n2 = 0; n2 < 15; ++n2) {
            const int k = 64 * n2 + lane;
            float2 a = Y[k], c = Y[kM - k];
            if (k == 0) a.y = 0.f, c.y = 0.f;                           // DC and Nyquist enter with their real parts only
            const float2 e = make_float2(a.x + c.x, a.y - c.y), d = make_float2(a.x - c.x, a.y + c.y);
            const float2 w = tw1920[k];
            const float2 o = make_float2(fmaf(d.x, w.x, -d.y * w.y), fmaf(d.x, w.y, d.y * w.x));
            v[n2] = make_float2(e.x - o.y, e.y + o.x);
        }
    }
    __syncthreads();                                                   // every wave has read its row before it is overwritten
    if (t < T) {
        fft960_wave<+1>(v, lane, tw960);
        const int k1 = bitrev6(lane);
        const float sc = 1.f / (float)kNfft;
#pragma unroll
        for (int k2 = 0; k2 < 15; ++k2) {
            const float2 z = v[slot15(k2)];
            Y[15 * k1 + k2] = make_float2(z.x * sc, z.y * sc);         // (x[2m], x[2m+1]), m = 15 k1 + k2
        }
    }
    __syncthreads();
    if (t < T) {
        float* fr = frames + (fbase + t) * kNfft;
        for (int i = lane; i < kM / 2; i += 64)                        // 4 consecutive samples per lane and store
            *reinterpret_cast<float4*>(fr + 4 * i) = *reinterpret_cast<const float4*>(Y + 2 * i);
    }
}

}  // namespace

int run_stft_fft(tvc_ctx* ctx, hipStream_t s, const float* wav, float* spec, int B, int64_t L) {
    if (!ctx->fft_tw960 || !ctx->fft_tw1920 || !ctx->fft_hann) return fail(ctx, TVC_ERR_STATE, "fft tables missing");
    const int T = (int)(L / kHop);
    static bool ready_dev[64] = {};
    bool& ready = ready_dev[ctx->device & 63];
    constexpr int lds = kFW * kM * 8;
    static_assert(kBins * kFW * 4 <= lds, "the magnitude tile overlays the bins");
    if (!ready) {
        hipError_t e = hipFuncSetAttribute((const void*)stft_fft_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return fail(ctx, TVC_ERR_HIP, "stft_fft setup: %s", hipGetErrorString(e));
        ready = true;
    }
    RagDev rg;
    TVC_CHECK(rag_view(ctx, s, 1, 0, &rg, nullptr));
    const int groups = ((ctx->rag ? ctx->rag->Tlong : T) + kFW - 1) / kFW;
    if (ctx->rag) B = ctx->rag->B;
    hipLaunchKernelGGL(stft_fft_kernel, dim3((unsigned)(B * groups)), dim3(kFW * 64), lds, s, wav, spec, reinterpret_cast<const float2*>(ctx->fft_tw960),
                       reinterpret_cast<const float2*>(ctx->fft_tw1920), ctx->fft_hann, (int)L, T, rg);
    return launch_check(ctx, "stft_fft");
}

// angle_padded (ragged batches only): `angle` is the caller's padded [rows][961][Tmax] tensor, not the batch-wide [961][T] layout
int run_noise_ifft(tvc_ctx* ctx, hipStream_t s, const float* kern, const float* angle, uint64_t seed, float* frames, int B, int T, bool angle_padded) {
    if (!ctx->fft_tw960 || !ctx->fft_tw1920) return fail(ctx, TVC_ERR_STATE, "fft tables missing");
    static bool ready_dev[64] = {};
    bool& ready = ready_dev[ctx->device & 63];
    constexpr int lds = kFW * (kBins + 1) * 8;
    if (!ready) {
        hipError_t e = hipFuncSetAttribute((const void*)noise_ifft_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)noise_ifft_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return fail(ctx, TVC_ERR_HIP, "noise_ifft setup: %s", hipGetErrorString(e));
        ready = true;
    }
    RagDev rg;
    TVC_CHECK(rag_view(ctx, s, 1, 0, &rg, nullptr));
    const int* rowmap = rg.row;                    // ragged batch: utterance -> row of the call (the hash's row index)
    if (!angle_padded) rg.row = nullptr;
    const int groups = ((ctx->rag ? ctx->rag->Tlong : T) + kFW - 1) / kFW;
    if (ctx->rag) B = ctx->rag->B;
    const dim3 grid((unsigned)(B * groups)), blk(kFW * 64);
    const float2* t960 = reinterpret_cast<const float2*>(ctx->fft_tw960);
    const float2* t1920 = reinterpret_cast<const float2*>(ctx->fft_tw1920);
    if (angle) hipLaunchKernelGGL(noise_ifft_kernel<false>, grid, blk, lds, s, kern, angle, frames, t960, t1920, T, rg, 0ull, rowmap);
    else hipLaunchKernelGGL(noise_ifft_kernel<true>, grid, blk, lds, s, kern, kern, frames, t960, t1920, T, rg, (unsigned long long)seed, rowmap);
    return launch_check(ctx, "noise_ifft");
}

}  // namespace tvc
