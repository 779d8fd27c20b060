// Prototype timing: a k3 dilated conv C -> C on the operand-stationary schedule of cnx_s3.h (whole K of a 64-column tile resident in LDS,
// A fragments straight from the packed image, no barrier after staging), against conv_s2's 97-103 us at C = 384 x 25 600 columns.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -Itinyvc_amd/csrc tools/micro/convb_proto.hip -o tools/micro/convb_proto_bin
#include "cnx_s3.h"
#ifndef PD
#define PD 4
#endif
#include <cstdio>
#include <vector>
using namespace tvc;

template <int C, int D>
struct CB {
    static constexpr int NT = 2, NC = 64, P = NC + 2 * D, KS = C / 16 * 3, MT = C / 32;
    static constexpr int WAVES = MT <= 12 ? MT : 12, NTHR = WAVES * 64;
    static constexpr int XS_U4 = (C / 16) * 4 * P;
    static constexpr int ITEMS = (C / 8) * P, XPER = (ITEMS + NTHR - 1) / NTHR;
    static constexpr int LDS = XS_U4 * 16;
};

template <int C, int D>
__global__ __launch_bounds__((CB<C, D>::NTHR)) void convb_kernel(const float* x, float* y, const uint4* A6, const float* wsc, const float* bias, int len) {
    using CF = CB<C, D>;
    constexpr int NT = CF::NT, NC = CF::NC, P = CF::P, KS = CF::KS, MT = CF::MT, WAVES = CF::WAVES, NTHR = CF::NTHR, XPER = CF::XPER;
    extern __shared__ __attribute__((aligned(16))) uint4 Xs[];      // [slab][part][half][P]
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    const int ntu = (len + NC - 1) / NC, TW = (len + ntu - 1) / ntu;
    const int t0 = blockIdx.x * TW, tw = len - t0 < TW ? len - t0 : TW;
    const float* xb = x + (long)b * C * len;
    float* yb = y + (long)b * C * len;
    // staging: item = (8-channel group g, position p): lrelu, split
#pragma unroll
    for (int i = 0; i < XPER; ++i) {
        int idx = tid + i * NTHR;
        const bool ok = idx < CF::ITEMS;
        idx = ok ? idx : CF::ITEMS - 1;
        const int g = idx / P, p = idx - g * P;
        int t = t0 - D + p;
        t = t < 0 ? 0 : (t > len - 1 ? len - 1 : t);
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = xb[(long)(8 * g + q) * len + t];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.1f * v[q]);
        uint4 p1, p2;
        split8(v, p1, p2);
        if (ok) {
            Xs[(((g >> 1) * 2 + 0) * 2 + (g & 1)) * P + p] = p1;
            Xs[(((g >> 1) * 2 + 1) * 2 + (g & 1)) * P + p] = p2;
        }
    }
    __syncthreads();
    for (int mt = wave; mt < MT; mt += WAVES) {
        u32x4 ring[PD][2];
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            const uint4* ab = A6 + ((long)u * MT + mt) * kPU4;
            ring[u][0] = ldg_so4(ab, 16u * (unsigned)lane);
            ring[u][1] = ldg_so4(ab, 16u * (unsigned)(64 + lane));
        }
        f32x16 hi[NT], lo[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) hi[j][r] = lo[j][r] = 0.f;
        const uint4* yb4 = Xs + lh * P + l31;
        u32x4 bq[2][NT][2];
        auto bread = [&](int k, int fb) __attribute__((always_inline)) {
            const int slab = k / 3, tap = k - slab * 3;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int p = 0; p < 2; ++p) bq[fb][j][p] = *reinterpret_cast<const u32x4*>(yb4 + ((slab * 2 + p) * 2) * P + tap * D + j * 32);
        };
        bread(0, 0);
#pragma unroll 1
        for (int k0 = 0; k0 < KS; k0 += 24) {
#pragma unroll
            for (int u = 0; u < 24; ++u) {
                const int k = k0 + u, fb = u & 1, ru = u % PD;
                const f16x8 a0 = __builtin_bit_cast(f16x8, ring[ru][0]), a1 = __builtin_bit_cast(f16x8, ring[ru][1]);
                int kn = k + PD;
                kn = kn < KS ? kn : KS - 1;
                const uint4* ab = A6 + ((long)kn * MT + mt) * kPU4;
                ring[ru][0] = ldg_so4(ab, 16u * (unsigned)lane);
                ring[ru][1] = ldg_so4(ab, 16u * (unsigned)(64 + lane));
                {
                    const int kb = k + 1 < KS ? k + 1 : 0;
                    // (k0 is a multiple of 12 = 4 slabs: slab / tap of step k + 1 from u alone)
                    const int slab = kb / 3, tap = kb - slab * 3;
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int p = 0; p < 2; ++p) bq[fb ^ 1][j][p] = *reinterpret_cast<const u32x4*>(yb4 + ((slab * 2 + p) * 2) * P + tap * D + j * 32);
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) lo[j] = TVC_MFMA16(a1, __builtin_bit_cast(f16x8, bq[fb][j][0]), lo[j]);
#pragma unroll
                for (int j = 0; j < NT; ++j) hi[j] = TVC_MFMA16(a0, __builtin_bit_cast(f16x8, bq[fb][j][0]), hi[j]);
#pragma unroll
                for (int j = 0; j < NT; ++j) lo[j] = TVC_MFMA16(a0, __builtin_bit_cast(f16x8, bq[fb][j][1]), lo[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const float cw = wsc[mt], cl = cw * kLoInv;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = j * 32 + l31;
            const unsigned oo = 4u * (unsigned)(4 * lh * len + t0 + (n < tw ? n : tw - 1));
#pragma unroll
            for (int r = 0; r < 16; ++r)
                stg_so(yb + (long)(mt * 32 + (r & 3) + 8 * (r >> 2)) * len, oo, comb(hi[j][r], lo[j][r], cw, cl) + bias[mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh]);
        }
    }
}

template <int C, int D>
static void run(int B, int len) {
    using CF = CB<C, D>;
    const size_t n = (size_t)B * C * len;
    float *x, *y, *misc;
    hipMalloc(&x, n * 4);
    hipMalloc(&y, n * 4);
    hipMalloc(&misc, 8192 * 4);
    std::vector<float> hx(n), hm(8192, 0.01f);
    for (auto& v : hx) v = (float)(rand() & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(misc, hm.data(), 8192 * 4, hipMemcpyHostToDevice);
    const size_t img = (size_t)CF::KS * CF::MT * 2 * 64 * 16;
    uint16_t* A;
    hipMalloc(&A, img);
    std::vector<uint16_t> w(img / 2);
    for (auto& v : w) v = 0x2000 + (rand() & 0x3ff);
    hipMemcpy(A, w.data(), img, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)convb_kernel<C, D>, hipFuncAttributeMaxDynamicSharedMemorySize, CF::LDS);
    const int tx = (len + 63) / 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((convb_kernel<C, D>), dim3(tx, B), dim3(CF::NTHR), CF::LDS, 0, x, y, (const uint4*)A, misc, misc + 1024, len);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((convb_kernel<C, D>), dim3(tx, B), dim3(CF::NTHR), CF::LDS, 0, x, y, (const uint4*)A, misc, misc + 1024, len);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("C=%d d=%d B=%d len=%d  tiles %d x %d, LDS %d KB, %d waves: %.1f us   (%s)\n", C, D, B, len, tx, B, CF::LDS / 1024, CF::WAVES, ms * 1000.f / 20, hipGetErrorString(hipGetLastError()));
    hipFree(x); hipFree(y); hipFree(misc); hipFree(A);
}

int main() {
    run<384, 1>(64, 400);
    run<384, 9>(64, 400);
    run<192, 1>(64, 1200);
    run<192, 9>(64, 1200);
    run<192, 27>(64, 1200);
    run<96, 9>(64, 4800);
    return 0;
}
