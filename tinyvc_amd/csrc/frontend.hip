// Front end: |STFT| (spectrogram.py:8-15), energy envelope (energy_estimation.py:9-14),
// semitone shift (pitch_shift.py:5-15).
#include "small_kernels.h"
#include "tvc_common.h"

namespace tvc {

// |STFT| (spectrogram.py:8-15): one wavefront per 1920-sample frame, a 960-point complex FFT on the packed real frame
// (fft.hip).  Needs no scratch.
int run_stft(tvc_ctx* ctx, hipStream_t s, Ws& ws, bool dry, const float* wav, float* spec, int B, int64_t L) {
    (void)ws;
    return dry ? 0 : run_stft_fft(ctx, s, wav, spec, B, L);
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

// ragged batch (ragged.h): utterance b = blockIdx.y pools its own row of the caller's padded tensor over its own length into
// e[eoff(b) ...] (eoff(b) = floor(7.5 pre[b]): the pooled lengths floor(7.5 tb) of the utterances before it fit in front), then
// F.interpolate(e_b, L_b) lands at samples pre[b] * 480 ... of the batch-wide energy row
static __global__ __launch_bounds__(256) void energy_rag_kernel(const float* __restrict__ wav, float* __restrict__ e, float* __restrict__ energy, RagDev rg, int phase) {
    const int b = blockIdx.y;
    const int T = rg.tb[b], L = T * kHop, ne = (L + 2 * 32 - 128) / 64 + 1;
    float* eb = e + (15L * rg.pre[b]) / 2;
    if (phase == 0) {
        const float* x = wav + (long)rg.row[b] * rg.Tmax * kHop;
        const int lane = threadIdx.x & 63;
        const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), nwaves = (int)((gridDim.x * blockDim.x) >> 6);
        for (int j = wave; j < ne; j += nwaves) {
            const int lo = 64 * j - 32;
            float m = -INFINITY;
            for (int k = lane; k < 128; k += 64) {
                const int p = lo + k;
                if (p >= 0 && p < L) m = fmaxf(m, fabsf(x[p]));
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            if (lane == 0) eb[j] = m;
        }
    } else {
        const float scale = __fdiv_rn((float)ne, (float)L);      // F.interpolate(e, L): `size` given -> scale = float(in) / float(out)
        float* y = energy + (long)rg.pre[b] * kHop;
        for (int d = blockIdx.x * blockDim.x + threadIdx.x; d < L; d += gridDim.x * blockDim.x) {
            const Lerp c = lerp_coord(d, scale, ne);
            y[d] = lerp_eval(c, eb[c.i0], eb[c.i1]);
        }
    }
}

// out[b] = a * in[b * in_stride] + c: the |max| slot of a tensor from the slot of the tensor it is a bounded function of (in_stride = 0:
// one value for every utterance).  One 64-thread workgroup; NaN / Inf carry through (bfp_from_amax ignores them).
static __global__ void slot_affine_kernel(float* __restrict__ out, const float* __restrict__ in, int in_stride, float a, float c, int n) {
    for (int b = threadIdx.x; b < n; b += blockDim.x) out[b] = fmaf(a, in[(long)b * in_stride], c);
}
int run_slot_affine(tvc_ctx* ctx, hipStream_t s, float* out, const float* in, int in_stride, float a, float c, int n) {
    hipLaunchKernelGGL(slot_affine_kernel, dim3(1), dim3(64), 0, s, out, in, in_stride, a, c, n);
    return launch_check(ctx, "slot_affine");
}
// The decoder's slots in one launch: zero[0 .. nz) = 0 (the slots its kernels raise with atomicMax), then the two bounds it is handed:
// o1[b] = a1 in1[b * s1] + c1, o2[b] = a2 in2[b * s2] + c2 (either pair may be null), and o3[b] = a3 o1[b] + c3 (a bound of a bounded function
// of the first tensor; null without o1).  One workgroup.
static __global__ void slot_prep_kernel(float* __restrict__ zero, int nz, float* __restrict__ o1, const float* __restrict__ in1, int s1, float a1, float c1,
                                        float* __restrict__ o2, const float* __restrict__ in2, int s2, float a2, float c2, float* __restrict__ o3, float a3,
                                        float c3, int n) {
    for (int i = threadIdx.x; i < nz; i += blockDim.x) zero[i] = 0.f;
    __syncthreads();      // (the bounds' slots lie inside the zeroed block)
    for (int b = threadIdx.x; b < n; b += blockDim.x) {
        if (o1) {
            const float v1 = fmaf(a1, in1[(long)b * s1], c1);
            o1[b] = v1;
            if (o3) o3[b] = fmaf(a3, v1, c3);
        }
        if (o2) o2[b] = fmaf(a2, in2[(long)b * s2], c2);
    }
}
int run_slot_prep(tvc_ctx* ctx, hipStream_t s, float* zero, int nz, float* o1, const float* in1, int s1, float a1, float c1, float* o2, const float* in2, int s2,
                  float a2, float c2, float* o3, float a3, float c3, int n) {
    hipLaunchKernelGGL(slot_prep_kernel, dim3(1), dim3(256), 0, s, zero, nz, o1, in1, s1, a1, c1, o2, in2, s2, a2, c2, o3, a3, c3, n);
    return launch_check(ctx, "slot_prep");
}
// emax[b] = max_j e[b][j] (one workgroup per utterance: 1 500 values); spec_bound[b] = 960.5 emax[b] >= every |STFT| bin (the Hann window's sum)
// zero[0 .. nz) = 0: the encoder's atomicMax slots, zeroed here instead of by a memset of their own
static __global__ __launch_bounds__(256) void pooled_max_kernel(const float* __restrict__ e, int ne, float* __restrict__ emax, float* __restrict__ spec_bound, float* __restrict__ zero, int nz) {
    __shared__ float red[4];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nz; i += gridDim.x * 256) zero[i] = 0.f;
    const float* p = e + (long)blockIdx.x * ne;
    float m = 0.f;
    for (int j = threadIdx.x; j < ne; j += 256) m = fmaxf(m, p[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        emax[blockIdx.x] = m;
        if (spec_bound) spec_bound[blockIdx.x] = fmaf(960.5f, m, 0.f);
    }
}

// the same for a ragged batch: utterance b = blockIdx.x pools ITS floor(7.5 tb) + 1 values (energy_rag_kernel's layout) - the bound its own B = 1 call computes
static __global__ __launch_bounds__(256) void pooled_max_rag_kernel(const float* __restrict__ e, RagDev rg, float* __restrict__ emax, float* __restrict__ spec_bound, float* __restrict__ zero, int nz) {
    __shared__ float red[4];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nz; i += gridDim.x * 256) zero[i] = 0.f;
    const int b = blockIdx.x;
    const int L = rg.tb[b] * kHop, ne = (L + 2 * 32 - 128) / 64 + 1;
    const float* p = e + (15L * rg.pre[b]) / 2;
    float m = 0.f;
    for (int j = threadIdx.x; j < ne; j += 256) m = fmaxf(m, p[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        emax[b] = m;
        if (spec_bound) spec_bound[b] = fmaf(960.5f, m, 0.f);
    }
}

int run_energy(tvc_ctx* ctx, hipStream_t s, Ws& ws, bool dry, const float* wav, float* energy, int B, int64_t L, float* emax, float* spec_bound, float* zero, int nz) {
    const int ne = (int)((L + 2 * 32 - 128) / 64 + 1);
    float* e = ws.get<float>((size_t)B * ne + 8);
    if (dry) return 0;
    if (ctx->rag) {
        RagDev rg;
        TVC_CHECK(rag_view(ctx, s, kHop, 0, &rg, nullptr));
        const int gx = ctx->rag->B >= 32 ? 8 : 64;
        hipLaunchKernelGGL(energy_rag_kernel, dim3(gx, ctx->rag->B), dim3(256), 0, s, wav, e, energy, rg, 0);
        if (emax) hipLaunchKernelGGL(pooled_max_rag_kernel, dim3(ctx->rag->B), dim3(256), 0, s, e, rg, emax, spec_bound, zero, nz);
        hipLaunchKernelGGL(energy_rag_kernel, dim3(gx * 4, ctx->rag->B), dim3(256), 0, s, wav, e, energy, rg, 1);
        return launch_check(ctx, "energy (ragged)");
    }
    hipLaunchKernelGGL(energy_pool_kernel, dim3(grid_for((long)B * ne * 64)), dim3(256), 0, s, wav, e, B, (int)L, ne);
    if (emax) hipLaunchKernelGGL(pooled_max_kernel, dim3(B), dim3(256), 0, s, e, ne, emax, spec_bound, zero, nz);
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
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = shift_frequency_one(f0[i], shift);
}

// u in [0, 1) -> u * 2 * pi - pi with the reference expression's three fp32 roundings (decoder.py:78: `torch.rand(...) * 2 * math.pi - math.pi`
// is three tensor ops; u * 2 is exact), in place, four values per thread
static __global__ __launch_bounds__(256) void uniform_to_angle_kernel(float* __restrict__ u, int64_t n) {
    const int64_t i = 4 * (blockIdx.x * (int64_t)blockDim.x + threadIdx.x);
    const float pi = 3.1415927410125732f;          // float(math.pi)
    if (i + 3 < n && (reinterpret_cast<uintptr_t>(u) & 15) == 0) {
        float4 v = *reinterpret_cast<float4*>(u + i);
        v.x = __fsub_rn(__fmul_rn(__fmul_rn(v.x, 2.f), pi), pi);
        v.y = __fsub_rn(__fmul_rn(__fmul_rn(v.y, 2.f), pi), pi);
        v.z = __fsub_rn(__fmul_rn(__fmul_rn(v.z, 2.f), pi), pi);
        v.w = __fsub_rn(__fmul_rn(__fmul_rn(v.w, 2.f), pi), pi);
        *reinterpret_cast<float4*>(u + i) = v;
    } else {
        for (int64_t k = i; k < n && k < i + 4; ++k) u[k] = __fsub_rn(__fmul_rn(__fmul_rn(u[k], 2.f), pi), pi);
    }
}
int run_uniform_to_angle(tvc_ctx* ctx, hipStream_t s, float* u, int64_t n) {
    const int64_t groups = (n + 3) / 4;
    hipLaunchKernelGGL(uniform_to_angle_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, u, n);
    return launch_check(ctx, "uniform_to_angle");
}

int run_shift(tvc_ctx* ctx, hipStream_t s, const float* f0, float* out, int64_t n, float semitones) {
    hipLaunchKernelGGL(shift_kernel, dim3(grid_for(n)), dim3(256), 0, s, f0, out, (long)n, semitones);
    return launch_check(ctx, "shift_frequency");
}

}  // namespace tvc
