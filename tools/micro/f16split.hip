// Microtest for the two-term fp16 split (x = h1 + 2^-11 h2) on v_mfma_f32_32x32x16_f16:
//  (1) does the f16 MFMA flush subnormal inputs?  (2) f16 vs bf16 issue rate  (3) GEMM error vs fp64 of
//  fp32 MFMA / bf16x3 six products / fp16x2 three products (two accumulators), at several activation magnitudes.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=on f16split.hip -o f16split && ./f16split
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void subnormal_probe(float* out) {
    const int lane = threadIdx.x;
    f16x8 a, b;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // A[row][k] = 2^-20 (fp16 subnormal) for k = 0 only, B[k][col] = 1 for k = 0
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)0.f; b[j] = (_Float16)0.f; }
    if (lane < 32) { a[0] = __builtin_bit_cast(_Float16, (unsigned short)0x0010); b[0] = (_Float16)1.f; }   // 0x0010 = 2^-24 * 16 = 2^-20
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    out[lane] = acc[0];
    // subnormal x subnormal-free big: 2^-20 * 1024
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (lane < 32) b[0] = (_Float16)1024.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    out[64 + lane] = acc[0];
    // smallest subnormal 2^-24 times 1
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (lane < 32) { a[0] = __builtin_bit_cast(_Float16, (unsigned short)0x0001); b[0] = (_Float16)1.f; }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    out[128 + lane] = acc[0];
    // f32 -> f16 conversion of a subnormal-range value (does v_cvt flush?)
    float v = 3.0e-6f * (1 + lane);
    _Float16 h = (_Float16)v;
    out[192 + lane] = (float)h;
}

template <int MODE>   // 0 f16, 1 bf16
__global__ __launch_bounds__(256) void rate_k(float* out, int iters) {
    f32x16 acc[3];
    for (int i = 0; i < 3; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 ah[4], bh[4];
    bf16x8 ab[4], bb[4];
    unsigned seed = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int q = 0; q < 4; ++q)
        for (int j = 0; j < 8; ++j) {
            seed = seed * 1664525u + 1013904223u;
            float x = ((int)(seed >> 8) % 2001 - 1000) * 1e-3f;
            seed = seed * 1664525u + 1013904223u;
            float y = ((int)(seed >> 8) % 2001 - 1000) * 1e-3f;
            ah[q][j] = (_Float16)x; bh[q][j] = (_Float16)y; ab[q][j] = (__bf16)x; bb[q][j] = (__bf16)y;
        }
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[(i + u) & 3], bh[(i * 2 + u) & 3], acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[(i + u) & 3], bb[(i * 2 + u) & 3], acc[i], 0, 0, 0);
            }
    }
    float s = 0;
    for (int i = 0; i < 3; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// one wave per 32 x 32 output tile; W [M][K], X [K][N] fp32 row-major
// MODE 0: fp32 mfma 32x32x2; 1: bf16x3 six products; 2: fp16x2 three products, two accumulators; 3: fp16x2, ONE accumulator (h2 unscaled);
// 4: as 3 with both operands normalised first (W to |max| in [2^13, 2^14), X to [2^14, 2^15): film_s2.h), ws / xn = the powers of two
template <int MODE>
__global__ __launch_bounds__(64) void gemm_k(const float* __restrict__ W, const float* __restrict__ X, float* __restrict__ Y, int M, int K, int N, float xs, float ws = 1.f, float xn = 1.f) {
    const int lane = threadIdx.x, l31 = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    f32x16 acc, acc2;
    for (int r = 0; r < 16; ++r) acc[r] = acc2[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        float wv[8], xv[8];
        for (int j = 0; j < 8; ++j) {
            wv[j] = W[(long)(m0 + l31) * K + k0 + 8 * lh + j] * (MODE == 4 ? ws : 1.f);
            xv[j] = X[(long)(k0 + 8 * lh + j) * N + n0 + l31] * xs * (MODE == 4 ? xn : 1.f);
        }
        if (MODE == 0) {
            for (int kk = 0; kk < 16; kk += 2) {
                // 32x32x2 f32: A lane holds row l31, k = lh; B lane holds col l31, k = lh
                float a = W[(long)(m0 + l31) * K + k0 + kk + lh], b = X[(long)(k0 + kk + lh) * N + n0 + l31] * xs;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
        } else if (MODE == 1) {
            bf16x8 w[3], x[3];
            for (int j = 0; j < 8; ++j) {
                __bf16 h1 = (__bf16)wv[j]; float r = wv[j] - (float)h1; __bf16 h2 = (__bf16)r; float r2 = r - (float)h2; __bf16 h3 = (__bf16)r2;
                w[0][j] = h1; w[1][j] = h2; w[2][j] = h3;
                h1 = (__bf16)xv[j]; r = xv[j] - (float)h1; h2 = (__bf16)r; r2 = r - (float)h2; h3 = (__bf16)r2;
                x[0][j] = h1; x[1][j] = h2; x[2][j] = h3;
            }
            const int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
            for (int q = 0; q < 6; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[PA[q]], x[PB[q]], acc, 0, 0, 0);
        } else {
            const float S = MODE == 2 ? 2048.f : 1.f;
            f16x8 w1, w2, x1, x2;
            for (int j = 0; j < 8; ++j) {
                _Float16 h1 = (_Float16)wv[j]; _Float16 h2 = (_Float16)((wv[j] - (float)h1) * S);
                w1[j] = h1; w2[j] = h2;
                h1 = (_Float16)xv[j]; h2 = (_Float16)((xv[j] - (float)h1) * S);
                x1[j] = h1; x2[j] = h2;
            }
            if (MODE == 2) {
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, x1, acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, x2, acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, x1, acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, x1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, x2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, x1, acc, 0, 0, 0);
            }
        }
    }
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + 4 * lh + (r & 3) + 8 * (r >> 2);
        float v = MODE == 2 ? fmaf(acc2[r], 1.f / 2048.f, acc[r]) : acc[r];
        Y[(long)row * N + n0 + l31] = MODE == 4 ? v / ws / xn / xs : v / xs;
    }
}

static double relrms(const std::vector<float>& y, const std::vector<double>& ref) {
    double a = 0, b = 0;
    for (size_t i = 0; i < y.size(); ++i) { double d = y[i] - ref[i]; a += d * d; b += ref[i] * ref[i]; }
    return std::sqrt(a / b);
}

int main() {
    {
        float* d; hipMalloc(&d, 256 * 4);
        subnormal_probe<<<1, 64>>>(d);
        float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("subnormal A (2^-20) x 1        -> %.6g (expect 9.53674e-07 if NOT flushed)\n", h[0]);
        printf("subnormal A (2^-20) x 1024     -> %.6g (expect 0.000976562)\n", h[64]);
        printf("smallest subnormal (2^-24) x 1 -> %.6g (expect 5.96046e-08)\n", h[128]);
        printf("cvt f32->f16 of 3e-6, 6e-6     -> %.6g %.6g (expect ~2.98e-06, 5.96e-06 if cvt keeps subnormals)\n", h[192], h[193]);
        hipFree(d);
    }
    {
        float* out; hipMalloc(&out, 1024 * 256 * 4);
        for (int mode = 0; mode < 2; ++mode)
            for (int rep = 0; rep < 2; ++rep) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                const int iters = 20000, blocks = 1024;
                hipEventRecord(e0);
                if (mode == 0) rate_k<0><<<blocks, 256>>>(out, iters); else rate_k<1><<<blocks, 256>>>(out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                double flops = (double)blocks * 4 * iters * 3 * 2.0 * 32 * 32 * 16;
                printf("%s MFMA 32x32x16: %.3f ms -> %.1f TFLOP/s\n", mode ? "bf16" : "f16 ", ms, flops / ms * 1e-9);
            }
        hipFree(out);
    }
    const int M = 128, K = 768, N = 256;
    std::vector<float> W(M * K), X(K * N), Y(M * N);
    srand(1);
    auto rnd = []() { double u = 0; for (int i = 0; i < 12; ++i) u += rand() / (double)RAND_MAX; return u - 6.0; };
    for (auto& w : W) w = (float)(rnd() * 0.05);
    for (auto& x : X) x = (float)rnd();
    float *dW, *dX, *dY;
    hipMalloc(&dW, W.size() * 4); hipMalloc(&dX, X.size() * 4); hipMalloc(&dY, Y.size() * 4);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
    std::vector<double> ref(M * N, 0.0);
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < K; ++k) {
            double w = W[m * K + k];
            for (int n = 0; n < N; ++n) ref[m * N + n] += w * (double)X[k * N + n];
        }
    const float scales[] = {1.f, 1e-3f, 1e-5f, 1e-7f, 1e3f, 1.5e4f, 1e5f};
    for (float xs : scales) {
        printf("activation scale %-8g:", xs);
        float wmax = 0.f, xmax = 0.f;
        for (auto w : W) wmax = std::max(wmax, std::fabs(w));
        for (auto x : X) xmax = std::max(xmax, std::fabs(x * xs));
        int ew, ex;
        frexpf(wmax, &ew);
        frexpf(xmax, &ex);
        const float ws = ldexpf(1.f, 14 - ew), xn = ldexpf(1.f, 15 - ex);
        for (int mode = 0; mode < 5; ++mode) {
            dim3 g(N / 32, M / 32);
            if (mode == 4) gemm_k<4><<<g, 64>>>(dW, dX, dY, M, K, N, xs, ws, xn);
            if (mode == 0) gemm_k<0><<<g, 64>>>(dW, dX, dY, M, K, N, xs);
            if (mode == 1) gemm_k<1><<<g, 64>>>(dW, dX, dY, M, K, N, xs);
            if (mode == 2) gemm_k<2><<<g, 64>>>(dW, dX, dY, M, K, N, xs);
            if (mode == 3) gemm_k<3><<<g, 64>>>(dW, dX, dY, M, K, N, xs);
            hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost);
            const char* nm[5] = {"fp32", "bf16x3", "f16x2(2acc)", "f16x2(1acc)", "f16x2(1acc,norm)"};
            printf("  %s %.3e", nm[mode], relrms(Y, ref));
        }
        printf("\n");
    }
    return 0;
}
