// Epilogue functors of the plain-GEMM launches of conv3s.h (gemm_s_launch): store(n, m, v[4]) finishes four consecutive
// output rows m..m+3 of column n = b * T + t (bias, activation, residual, the conditioning sums of the two input layers).
#pragma once
#include <hip/hip_runtime.h>

namespace tvc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { ACT_NONE = 0, ACT_GELU = 1, ACT_ELU1 = 2 };

// erf(a): two fma chains and one expf, both evaluated and selected (no branch): |a| <= 0.9277: a + a P(a^2), above: 1 - exp(Q(|a|)) (N. Juffa's
// single-precision kernels).  Against the correctly rounded erf on every float of +-[2^-30, 16): within 1 ulp (libm's erff: 2), and GELU's
// largest absolute error against fp64 is the same 4.5e-7 with either (tools/micro/erf_fast.hip).  Kept for the accuracy (the reference's CPU erf
// is a 1-ulp one); the time is libm's: the GELU epilogue is 12 % of the encoder's first 1x1 launches with either.
__device__ __forceinline__ float erf_fast(float a) {
    const float t = fabsf(a), s = a * a;
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    r = copysignf(1.0f - expf(r), a);
    float p = -5.96761703e-4f;
    p = fmaf(p, s, 4.99119423e-3f);
    p = fmaf(p, s, -2.67681349e-2f);
    p = fmaf(p, s, 1.12819925e-1f);
    p = fmaf(p, s, -3.76125336e-1f);
    p = fmaf(p, s, 1.28379166e-1f);
    p = fmaf(p, a, a);
    return t > 0.927734375f ? r : p;
}

// GELU of two values with the fma chains on the packed fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two IEEE operations per
// instruction, the same roundings as the scalar ones): bit for bit act_apply(., ACT_GELU) of either value, ~27 instead of ~42 vector
// instructions per value (the exp and the selects stay scalar).  The epilogue of the encoder's first 1x1 is bound by exactly these.
typedef float f32x2e __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2e gelu_pair(f32x2e o) {
    auto sp = [](float x) __attribute__((always_inline)) { return f32x2e{x, x}; };
    auto fm = [](f32x2e x, f32x2e y, f32x2e z) __attribute__((always_inline)) { return __builtin_elementwise_fma(x, y, z); };
    const f32x2e a = o * sp(0.70710678118654752f);
    const f32x2e t = {fabsf(a[0]), fabsf(a[1])}, s = a * a;
    f32x2e r = fm(sp(-1.72853470e-5f), t, sp(3.83197126e-4f));
    const f32x2e u = fm(sp(-3.88396438e-3f), t, sp(2.42546219e-2f));
    r = fm(r, s, u);
    r = fm(r, t, sp(-1.06777877e-1f));
    r = fm(r, t, sp(-6.34846687e-1f));
    r = fm(r, t, sp(-1.28717512e-1f));
    r = fm(r, t, -t);
    const f32x2e ex = {expf(r[0]), expf(r[1])};
    const f32x2e one_m = sp(1.0f) - ex;
    r = f32x2e{copysignf(one_m[0], a[0]), copysignf(one_m[1], a[1])};
    f32x2e p = sp(-5.96761703e-4f);
    p = fm(p, s, sp(4.99119423e-3f));
    p = fm(p, s, sp(-2.67681349e-2f));
    p = fm(p, s, sp(1.12819925e-1f));
    p = fm(p, s, sp(-3.76125336e-1f));
    p = fm(p, s, sp(1.28379166e-1f));
    p = fm(p, a, a);
    const f32x2e erf = {t[0] > 0.927734375f ? r[0] : p[0], t[1] > 0.927734375f ? r[1] : p[1]};
    return (sp(0.5f) * o) * (sp(1.f) + erf);
}

__device__ __forceinline__ float act_apply(float o, int act) {
    if (act == ACT_GELU) return 0.5f * o * (1.f + erf_fast(o * 0.70710678118654752f));
    if (act == ACT_ELU1) return (o > 0.f ? o : (expf(o) - 1.f)) + 1.f;
    return o;
}

// y[b][m][t] = act(acc + bias[m]) (+ res[b][m][t])
template <int ACT, bool RES>
struct EpiBias {
    static constexpr bool kIgemm = true;   // store(n, m, v[4]) interface (also usable from the split-precision kernel)
    float* y;
    const float* bias;
    const float* res;
    int M, T, ncols;
    long y_bs, res_bs;  // batch strides
    // optional: the |max| slot of y as a bound from the slot of the GEMM's input, bound_out[u] = bw bound_in[u] + bb for every utterance u
    // (the functor finishes the elements, so no maximum is tracked; the one thread that stores element (0, 0) writes all nb of them -
    // a 64-thread launch of its own otherwise)
    float* bound_out = nullptr;
    const float* bound_in = nullptr;
    float bw = 0.f, bb = 0.f;
    int nb = 0;
    __device__ __forceinline__ void store(int n, int m, const float v[4]) const {
        if (n >= ncols) return;
        if (bound_out != nullptr && n == 0 && m == 0)
            for (int u = 0; u < nb; ++u) bound_out[u] = fmaf(bw, bound_in[u], bb);
        int b = n / T, t = n - b * T;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (m + r < M) {
                float o = act_apply(v[r] + bias[m + r], ACT);
                if (RES) o += res[b * res_bs + (long)(m + r) * T + t];
                y[b * y_bs + (long)(m + r) * T + t] = o;
            }
        }
    }
};

// y = acc + bias in the G8 layout [B][M / 8][T][8 channels] (the 48-channel level output conv48s.hip interpolates from: four consecutive
// rows of one column are 16 contiguous bytes, one store).  m is a multiple of 4; M a multiple of 8.
struct EpiBiasG8 {
    static constexpr bool kIgemm = true;
    float* y;
    const float* bias;
    int M, T, ncols;
    float* bound_out = nullptr;      // (as in EpiBias)
    const float* bound_in = nullptr;
    float bw = 0.f, bb = 0.f;
    int nb = 0;
    __device__ __forceinline__ void store(int n, int m, const float v[4]) const {
        if (n >= ncols || m >= M) return;
        if (bound_out != nullptr && n == 0 && m == 0)
            for (int u = 0; u < nb; ++u) bound_out[u] = fmaf(bw, bound_in[u], bb);
        const int b = n / T, t = n - b * T;
        typedef float f32x4e __attribute__((ext_vector_type(4)));
        const f32x4e o = {v[0] + bias[m], v[1] + bias[m + 1], v[2] + bias[m + 2], v[3] + bias[m + 3]};
        *reinterpret_cast<f32x4e*>(y + (((long)b * (M >> 3) + (m >> 3)) * T + t) * 8 + (m & 7)) = o;
    }
};

// content_in(content) + energy_in(e) + f0_in(log(relu(f0)+1e-6))   (decoder.py:128, :223)
// e / lf0 are per-(b,t) scalars feeding 1->M 1x1 convs; `e` may be null (FilterNet has no energy).
struct EpiSumCond {
    static constexpr bool kIgemm = true;   // store(n, m, v[4]) interface (also usable from the split-precision kernel)
    float* y;
    const float* bias;
    const float* e;     // [B][T] or null
    const float* f0;    // [B][T]
    const float* we;
    const float* be;
    const float* wf;
    const float* bf;
    int M, T, ncols;
    __device__ __forceinline__ void store(int n, int m, const float v[4]) const {
        if (n >= ncols) return;
        int b = n / T, t = n - b * T;
        float lf = logf(fmaxf(f0[n], 0.f) + 1e-6f);
        float ev = e ? e[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (m + r < M) {
                float o = v[r] + bias[m + r];
                if (e) o = __fadd_rn(o, __fadd_rn(__fmul_rn(we[m + r], ev), be[m + r]));
                o = __fadd_rn(o, __fadd_rn(__fmul_rn(wf[m + r], lf), bf[m + r]));
                y[((long)b * M + m + r) * T + t] = o;
            }
        }
    }
};

// |STFT| in two passes: pass 1 parks Re X in `spec`; pass 2 holds Im X in its accumulators and
// overwrites spec with sqrt(re^2 + im^2).  Rows are bins.
}  // namespace tvc
