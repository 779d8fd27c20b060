#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
__device__ __forceinline__ float erf_fast(float a) {
    const float t = fabsf(a), s = a * a;
    // |a| > 0.927734375: 1 - exp(poly)
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, s, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    r = 1.0f - expf(r);
    r = copysignf(r, a);
    float p = -5.96761703e-4f;
    p = fmaf(p, s, 4.99119423e-3f);
    p = fmaf(p, s, -2.67681349e-2f);
    p = fmaf(p, s, 1.12819925e-1f);
    p = fmaf(p, s, -3.76125336e-1f);
    p = fmaf(p, s, 1.28379166e-1f);
    p = fmaf(p, a, a);
    return t > 0.927734375f ? r : p;
}
__device__ int ulpdiff(float a, float b) {
    int ia, ib; memcpy(&ia, &a, 4); memcpy(&ib, &b, 4);
    if (ia < 0) ia = 0x80000000 - ia;
    if (ib < 0) ib = 0x80000000 - ib;
    int d = ia - ib; return d < 0 ? -d : d;
}
__global__ void k(unsigned lo, unsigned n, unsigned* stats) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned u = lo + i;
    float x; memcpy(&x, &u, 4);
    for (int sg = 0; sg < 2; ++sg) {
        const float a = sg ? -x : x;
        const float ref = (float)erf((double)a);
        atomicMax(&stats[0], (unsigned)ulpdiff(erff(a), ref));
        atomicMax(&stats[1], (unsigned)ulpdiff(erf_fast(a), ref));
        // what GELU needs: 0.5 a (1 + erf(a / sqrt 2)) against double
        const double gd = 0.5 * (double)a * (1.0 + erf((double)a * 0.70710678118654752440));
        const float g0 = 0.5f * a * (1.f + erff(a * 0.70710678118654752f)), g1 = 0.5f * a * (1.f + erf_fast(a * 0.70710678118654752f));
        const double sc = fabs(gd) > 1e-30 ? fabs(gd) : 1e-30;
        atomicMax(&stats[2], (unsigned)(fmin(fabs(g0 - gd) / sc, 1.0) * 1e9));
        atomicMax(&stats[3], (unsigned)(fmin(fabs(g1 - gd) / sc, 1.0) * 1e9));
        atomicMax(&stats[4], (unsigned)(fabs(g0 - gd) * 1e12));
        atomicMax(&stats[5], (unsigned)(fabs(g1 - gd) * 1e12));
    }
}
int main() {
    unsigned* st; hipMalloc(&st, 32); hipMemset(st, 0, 32);
    // |a| from 2^-30 to 16
    const unsigned lo = (127 - 30) << 23, hi = (127 + 4) << 23;
    unsigned long long total = (unsigned long long)hi - lo;
    for (unsigned long long s = 0; s < total; s += (1ull << 28)) {
        unsigned n = (unsigned)((total - s) < (1ull << 28) ? (total - s) : (1ull << 28));
        k<<<(n + 255) / 256, 256>>>((unsigned)(lo + s), n, st);
    }
    hipDeviceSynchronize();
    unsigned h[8]; hipMemcpy(h, st, 32, hipMemcpyDeviceToHost);
    printf("|a| in [2^-30, 16), %llu values per sign: erff max ulp %u, erf_fast max ulp %u; GELU max rel err %.3g / %.3g, max abs err %.3g / %.3g\n", total, h[0], h[1], h[2] * 1e-9, h[3] * 1e-9, h[4] * 1e-12, h[5] * 1e-12);
    return 0;
}
