// Epilogue functors of the k3 / 1x1 conv launches of conv3s.h, conv48s.hip and filter_up24s.hip: what happens to an
// accumulator tile on its way to HBM (bias, residual, the next Downsample's 1/f-rate copy, FiLM combine).
#pragma once
#include "gemm_epi.h"

namespace tvc {

// Epilogue interface: store(b, t, m, v[4]) for 4 consecutive channels m..m+3 at (b, t); t < len guaranteed.
// F2 = 3 / 5: compile-time decimation factor of the scalar store() path (a runtime division per element costs more than
// the interpolate pass it replaces); the tile_store path of conv3s.h uses the runtime fields y2 / f2.
template <bool RES, int F2 = 0>
struct C3EpiBias {
    static constexpr bool kRes = RES;
    static constexpr bool kIgemm = false;
    float* y;
    const float* bias;
    const float* res;
    int M, len;
    // optional second output: the F.interpolate(scale_factor = 1/f2) copy the next Downsample block starts from
    // (decoder.py:148).  For f2 = 3 / 5 ATen's source coordinate f2 * (d + 0.5) - 0.5 is the integer f2*d + f2/2 (weight
    // exactly 1), for f2 = 4 it is 4d + 1.5 (weights exactly 0.5 / 0.5): the copy is a pick / a two-sample mean of y.
    float* y2 = nullptr;
    int f2 = 0;
    __device__ __forceinline__ void store(int b, int t, int m, const float v[4]) const {
        const int q = F2 > 0 ? t / F2 : 0;
        const bool pick = F2 > 0 && y2 != nullptr && t - q * F2 == (F2 >> 1);   // f2 = 4 needs two samples: tile_store only
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (m + r < M) {
                long i = ((long)b * M + m + r) * len + t;
                float o = v[r] + bias[m + r];
                if (RES) o += res[i];
                y[i] = o;
                if (pick) y2[((long)b * M + m + r) * (len / F2) + q] = o;
            }
    }
};

// C3EpiBias whose launch also accumulates a 1x1 conv of a second tensor into the tile (conv3s.h wants_res_conv):
// Downsample's c3 + down_res; `bias` = the two biases summed at pack time.
struct C3EpiBiasResConv : C3EpiBias<false, 0> {
    static constexpr bool kResConv = true;
};

// conv -> FiLM -> + residual with scale/shift computed in-kernel: store(b, t, m, h[4], sc[4], sh[4])
struct C3EpiFilmFused {
    static constexpr bool kIgemm = false;
    float* y;
    const float* bias;
    const float* bsc;
    const float* bsh;
    const float* res;
    int M, len;
    // conv3s.h only: res_lin > 0 -> `res` is the low-rate [B][M][res_lin] tensor and the residual is its
    // F.interpolate(scale_factor) (ATen scale float(1/scale_factor) in res_scale), evaluated in the epilogue
    int res_lin = 0;
    float res_scale = 0.f;
    __device__ __forceinline__ void store(int b, int t, int m, const float h[4], const float sc[4], const float sh[4]) const {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (m + r < M) {
                long i = ((long)b * M + m + r) * len + t;
                float hv = h[r] + bias[m + r];
                y[i] = __fadd_rn(__fadd_rn(__fmul_rn(hv, sc[r] + bsc[m + r]), sh[r] + bsh[m + r]), res[i]);
            }
    }
};

}  // namespace tvc
