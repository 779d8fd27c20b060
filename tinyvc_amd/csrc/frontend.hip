// Front end: |STFT| (spectrogram.py:8-15), energy envelope (energy_estimation.py:9-14),
// semitone shift (pitch_shift.py:5-15).
#include "conv3s.h"
#include "igemm.h"
#include "small_kernels.h"
#include "tvc_common.h"

namespace tvc {

#ifndef DFT_MTB   // workgroup of the DFT GEMMs: DFT_MTB x DFT_NWV waves of 32 x 32, DFT_BPC persistent workgroups per CU
#define DFT_MTB 2
#define DFT_NWV 4
#define DFT_BPC 2   // measured best of {4x4x1, 2x4x2, 2x4x1, 4x2x2, 2x8x1, 4x3x1}
#endif
#ifndef TVC_FFT
#define TVC_FFT 1   // |STFT| and the noise iSTFT as wave-level 1920-point FFTs (fft.hip); 0 = the half-size real-DFT GEMMs below
#endif
#ifndef TVC_SPLIT_DFT
#define TVC_SPLIT_DFT 1   // forward / inverse DFT GEMMs on the split-precision bf16 path
#endif

// ---- |STFT| as two half-size windowed real-DFT contractions on the fp32 matrix pipe ---------------
// spec[b][f][t] = | sum_n hann[n] x_b[(t+1)*480 + n - 960 (reflected)] e^{-2 pi i f n / 1920} |
// Each 1920-sample frame is folded about its centre once (fold kernel: e[n] = x[n] + x[N-n],
// o[n] = x[n] - x[N-n], stored k-major [960][B*T] so the GEMM's lanes read contiguous columns);
// Re and Im are then [961 x 960] real GEMMs over columns (b, t): 3.7 MFLOP per frame instead of 7.4.
// grid (15, ceil(ncols/32)): a workgroup folds 64 consecutive n of 32 frames.  Reads run along n
// (contiguous in the waveform), the k-major stores run along the frame index: LDS transposes.
static __global__ __launch_bounds__(256) void stft_fold_kernel(const float* __restrict__ wav, float* __restrict__ fe,
                                                               float* __restrict__ fo, int L, int T, int ncols) {
    __shared__ float te[64][33], to[64][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = blockIdx.x * 64, col0 = blockIdx.y * 32;
    for (int c = wave; c < 32; c += 4) {
        int col = col0 + c;
        float e = 0.f, o = 0.f;
        if (col < ncols) {
            int b = col / T, t = col - b * T;
            const float* wb = wav + (long)b * L;
            int n = k0 + lane + 1;
            int start = (t + 1) * 480 - 960;               // frame t+1 of the centred STFT (frame 0 is dropped)
            int p0 = start + n, p1 = start + 1920 - n;
            if (p0 < 0) p0 = -p0;
            if (p0 >= L) p0 = 2 * (L - 1) - p0;
            if (p1 < 0) p1 = -p1;
            if (p1 >= L) p1 = 2 * (L - 1) - p1;
            float a = wb[p0], d = wb[p1];
            e = n == 960 ? a : a + d;
            o = n == 960 ? 0.f : a - d;
        }
        te[lane][c] = e;
        to[lane][c] = o;
    }
    __syncthreads();
    const int cl = tid & 31;
    if (col0 + cl < ncols)
        for (int kk = tid >> 5; kk < 64; kk += 8) {
            long i = (long)(k0 + kk) * ncols + col0 + cl;
            fe[i] = te[kk][cl];
            fo[i] = to[kk][cl];
        }
}

int run_stft(tvc_ctx* ctx, hipStream_t s, Ws& ws, bool dry, const float* wav, float* spec, int B, int64_t L) {
    const int T = (int)(L / kHop);
    const int ncols = B * T;
    if (TVC_FFT) return dry ? 0 : run_stft_fft(ctx, s, wav, spec, B, L);
    float* fe = ws.get<float>((size_t)960 * ncols);
    float* fo = ws.get<float>((size_t)960 * ncols);
    if (dry) return 0;
    hipLaunchKernelGGL(stft_fold_kernel, dim3(15, (ncols + 31) / 32), dim3(256), 0, s, wav, fe, fo, (int)L, T, ncols);
#if TVC_SPLIT_DFT
    // both half-size DFT GEMMs on the split-precision path: fe/fo [960][ncols] are one "utterance" of ncols samples
    TVC_CHECK((gemm_s_launch<DFT_MTB, DFT_NWV, DFT_BPC>(ctx, s, ctx->stft_re, fe, 1, 960, ncols, 0, EpiStftPart<false>{spec, T, ncols})));
    TVC_CHECK((gemm_s_launch<DFT_MTB, DFT_NWV, DFT_BPC>(ctx, s, ctx->stft_im, fo, 1, 960, ncols, 0, EpiStftPart<true>{spec, T, ncols})));
#else
    {
        LoadMatrix ld{fe, 960, ncols};
        EpiStftPart<false> ep{spec, T, ncols};
        igemm_launch(s, ctx->stft_re.At, ctx->stft_re.Mpad, ctx->stft_re.Kpad, ncols, T, ld, ep);
    }
    {
        LoadMatrix ld{fo, 959, ncols};
        EpiStftPart<true> ep{spec, T, ncols};
        igemm_launch(s, ctx->stft_im.At, ctx->stft_im.Mpad, ctx->stft_im.Kpad, ncols, T, ld, ep);
    }
#endif
    return launch_check(ctx, "stft");
}

// ---- energy -------------------------------------------------------------------------------------
// e[j] = max |x[64 j - 32 .. 64 j + 95]| (max_pool1d(|x|, 128, 64, 32): out-of-range = -inf)
static __global__ void energy_pool_kernel(const float* __restrict__ wav, float* __restrict__ e, int B, int L, int ne) {
    long total = (long)B * ne;
    const int lane = threadIdx.x & 63;
    long wave = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    for (long i = wave; i < total; i += nwaves) {
        int b = (int)(i / ne), j = (int)(i - (long)b * ne);
        const float* x = wav + (long)b * L;
        int lo = 64 * j - 32;
        float m = -INFINITY;
        for (int k = lane; k < 128; k += 64) {
            int p = lo + k;
            if (p >= 0 && p < L) m = fmaxf(m, fabsf(x[p]));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) e[i] = m;
    }
}

int run_energy(tvc_ctx* ctx, hipStream_t s, Ws& ws, bool dry, const float* wav, float* energy, int B, int64_t L) {
    const int ne = (int)((L + 2 * 32 - 128) / 64 + 1);
    float* e = ws.get<float>((size_t)B * ne);
    if (dry) return 0;
    hipLaunchKernelGGL(energy_pool_kernel, dim3(grid_for((long)B * ne * 64)), dim3(256), 0, s, wav, e, B, (int)L, ne);
    // F.interpolate(e, L): `size` given -> scale = float(in) / float(out)
    float scale = (float)ne / (float)L;
    {
        const LerpLaunch ll = lerp_launch((long)B, (int)L);
        hipLaunchKernelGGL(lerp_resize_kernel, ll.grid, dim3(256), 0, s, e, energy, (long)B, ne, (int)L, scale, ll.tx);
    }
    return launch_check(ctx, "energy");
}

// ---- shift_frequency ----------------------------------------------------------------------------
// midi = log2(relu(f/440) + 1e-6) * 12 + 69 + shift;  f' = 440 * 2^((midi - 69) / 12)
// Every step is the fp32 op ATen runs, in ATen's order.  The two transcendentals are evaluated in fp64 and rounded
// once: ATen's CPU log2 / pow (Sleef u10) return the correctly rounded fp32 value on ~99 % of inputs and differ by one
// ulp otherwise, whereas two different <= 1 ulp approximations disagree on a large fraction - and one ulp of `midi`
// is 2.2e-7 of f0, which the harmonic oscillator integrates over the whole utterance (measured: 4.5e-5 of the 7.1e-5
// end-to-end rms difference at 4 s came from this function alone before the change).
static __global__ void shift_kernel(const float* __restrict__ f0, float* __restrict__ out, long n, float shift) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float r = __fdiv_rn(f0[i], 440.f);
        r = r < 0.f ? 0.f : r;                       // relu; NaN stays NaN as in F.relu
        const float lg = (float)log2((double)__fadd_rn(r, 1e-6f));
        float midi = __fadd_rn(__fmul_rn(lg, 12.f), 69.f);
        midi = __fadd_rn(midi, shift);
        float e = __fdiv_rn(__fsub_rn(midi, 69.f), 12.f);
        out[i] = __fmul_rn(440.f, (float)exp2((double)e));
    }
}

int run_shift(tvc_ctx* ctx, hipStream_t s, const float* f0, float* out, int64_t n, float semitones) {
    hipLaunchKernelGGL(shift_kernel, dim3(grid_for(n)), dim3(256), 0, s, f0, out, (long)n, semitones);
    return launch_check(ctx, "shift_frequency");
}

}  // namespace tvc
