// Stand-alone timing of the fused ConvNeXt launches (cnx_s3.h) at the bench shape, with -DCNX_ABL=<bits> what-if builds:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -Itinyvc_amd/csrc [-DCNX_ABL=n] tools/micro/cnx_bench.hip -o tools/micro/cnx_bench_bin
// bits: 1 = no depthwise-conv loads, 2 = no MFMA walk, 4 = no GELU, 8 = no stores, 16 = no A-fragment loads (ring stays constant), 32 = no operand staging loads (cnx2)
#include "cnx_s3.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace tvc;

template <class F>
static float time_it(F f, int reps = 20) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / reps;
}

template <int C>
static void run(int B, int T) {
    constexpr int KP = C == 384 ? 2 : 1;
    const size_t nx = (size_t)B * C * T, nh = 2 * nx;
    float *x, *h, *gx, *misc;
    hipMalloc(&x, nx * 4);
    hipMalloc(&h, nh * 4);
    hipMalloc(&gx, (size_t)B * 2 * C * 4 * ((T + 63) / 64));
    hipMalloc(&misc, 65536 * 4);
    std::vector<float> hx(nx), hm(65536);
    for (auto& v : hx) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto& v : hm) v = (float)rand() / RAND_MAX * 0.1f + 0.01f;
    hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice);
    hipMemcpy(misc, hm.data(), 65536 * 4, hipMemcpyHostToDevice);
    hipMemcpy(gx, hm.data(), (size_t)B * 2 * C * 4 > 65536 * 4 ? 65536 * 4 : (size_t)B * 2 * C * 4, hipMemcpyHostToDevice);
    // weight images: small fp16 values (0x2c00 = 0.0625) keep everything finite
    const size_t img1 = (size_t)(C / 16) * (2 * C / 32) * 2 * 64 * 16, img2 = (size_t)(2 * C / 16) * (C / 32) * 2 * 64 * 16;
    uint16_t *A1, *A2;
    hipMalloc(&A1, img1);
    hipMalloc(&A2, img2);
    std::vector<uint16_t> w(std::max(img1, img2) / 2);
    for (auto& v : w) v = 0x2000 + (rand() & 0x3ff);
    hipMemcpy(A1, w.data(), img1, hipMemcpyHostToDevice);
    hipMemcpy(A2, w.data(), img2, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)cnx1_kernel<C, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, Cnx1<C, 2>::LDS_BYTES);
    hipFuncSetAttribute((const void*)cnx2_kernel<C, KP>, hipFuncAttributeMaxDynamicSharedMemorySize, Cnx2<C, KP>::LDS_BYTES);
    CnxArgs a{};
    a.x = x;
    a.h = h;
    a.T = T;
    a.rs = T;
    a.wsc = misc;
    a.bias = misc + 1024;
    a.dw_w = misc + 4096;
    a.dw_b = misc + 8192;
    a.ln_g = misc + 9216;
    a.ln_b = misc + 10240;
    a.dil = 3;
    a.gp = gx;
    a.gp_tiles = (T + 63) / 64;
    a.gp_sum = 0;
    a.grn_g = misc + 12288;
    const int tx = (T + 63) / 64;
    CnxArgs a1 = a, a2 = a;
    a1.A6 = (const uint4*)A1;
    a1.MT = 2 * C / 32;
    a1.mt_per_wg = a1.MT;
    a2.A6 = (const uint4*)A2;
    a2.MT = C / 32;
    a2.mt_per_wg = a2.MT;
    const float t1 = time_it([&] { hipLaunchKernelGGL((cnx1_kernel<C, 2>), dim3(tx, B, 1), dim3(512), (Cnx1<C, 2>::LDS_BYTES), 0, a1); });
    const float t2 = time_it([&] { hipLaunchKernelGGL((cnx2_kernel<C, KP>), dim3(tx, B, 1), dim3((Cnx2<C, KP>::NTHR)), (Cnx2<C, KP>::LDS_BYTES), 0, a2); });
    printf("C=%d B=%d T=%d  cnx1 %.1f us   cnx2 %.1f us   (%s)\n", C, B, T, t1, t2, hipGetErrorString(hipGetLastError()));
#ifdef CNX_TRACE
    {
        unsigned long long* tr;
        hipMalloc(&tr, 64 * 8);
        hipMemset(tr, 0, 64 * 8);
        a1.trace = tr;
        hipLaunchKernelGGL((cnx1_kernel<C, 2>), dim3(tx, B, 1), dim3(512), (Cnx1<C, 2>::LDS_BYTES), 0, a1);
        hipDeviceSynchronize();
        unsigned long long ht[64];
        hipMemcpy(ht, tr, 64 * 8, hipMemcpyDeviceToHost);
        for (int w = 0; w < 2; ++w) {
            printf("  cnx1 wave %d:", w * 4);
            for (int i = 1; i < 13; ++i)
                if (ht[w * 32 + i]) printf(" [%d]+%lld", i, (long long)(ht[w * 32 + i] - ht[w * 32 + 0]));
            printf("\n");
        }
        hipFree(tr);
    }
#endif
    hipFree(x); hipFree(h); hipFree(gx); hipFree(misc); hipFree(A1); hipFree(A2);
}

int main() {
    run<384>(64, 200);
    run<128>(64, 200);
    run<384>(1, 200);
    return 0;
}
