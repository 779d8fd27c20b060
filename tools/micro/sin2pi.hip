#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
// sin(x) for x in [0, 2 pi]: two-term Cody-Waite reduction by pi/2 with fma, degree-7 / degree-8 minimax kernels (Cephes sinf / cosf)
__device__ __forceinline__ float sin_0_2pi(float x) {
    const float qf = rintf(x * 0.63661977236758134f);
    float r = fmaf(-qf, 1.57079637050628662109375f, x);
    r = fmaf(-qf, -4.37113900018624283e-8f, r);
    const int q = (int)qf;
    const float z = r * r;
    float s = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    s = fmaf(s * z, r, r);
    float c = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    c = fmaf(z, fmaf(z, c, -0.5f), 1.0f);
    float v = (q & 1) ? c : s;
    return (q & 2) ? -v : v;
}
// (sin, cos)(a) for |a| <= 2 pi (the noise phases are drawn in [-pi, pi)): the same reduction and kernels
__device__ __forceinline__ void sincos_small(float a, float* sn, float* cs) {
    const float qf = rintf(a * 0.63661977236758134f);
    float r = fmaf(-qf, 1.57079637050628662109375f, a);
    r = fmaf(-qf, -4.37113900018624283e-8f, r);
    const int q = (int)qf;
    const float z = r * r;
    float s = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    s = fmaf(s * z, r, r);
    float c = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    c = fmaf(z, fmaf(z, c, -0.5f), 1.0f);
    const float vs = (q & 1) ? c : s, vc = (q & 1) ? s : c;
    *sn = (q & 2) ? -vs : vs;
    *cs = ((q + 1) & 2) ? -vc : vc;
}
__global__ void k2(unsigned lo, unsigned n, unsigned* stats) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned u = lo + i;
    float x; memcpy(&x, &u, 4);
    for (int sign = 0; sign < 2; ++sign) {
        const float a = sign ? -x : x;
        float s0, c0, s1, c1;
        sincosf(a, &s0, &c0);
        sincos_small(a, &s1, &c1);
        const float rs = (float)sin((double)a), rc = (float)cos((double)a);
        atomicMax(&stats[0], (unsigned)(fmaxf(fabsf(s0 - rs), fabsf(c0 - rc)) * 16777216.f * 1000.f));
        atomicMax(&stats[1], (unsigned)(fmaxf(fabsf(s1 - rs), fabsf(c1 - rc)) * 16777216.f * 1000.f));
    }
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
    const float ref = (float)sin((double)x);
    const int da = ulpdiff(sinf(x), ref), db = ulpdiff(sin_0_2pi(x), ref);
    // absolute error matters more than ulps near the zeros of sin: track |err| in units of 2^-24 (1 ulp at 0.5..1)
    const float ea = fabsf(sinf(x) - ref) * 16777216.f, eb = fabsf(sin_0_2pi(x) - ref) * 16777216.f;
    atomicMax(&stats[0], (unsigned)da); atomicMax(&stats[1], (unsigned)db);
    if (da) atomicAdd(&stats[2], 1u);
    if (db) atomicAdd(&stats[3], 1u);
    atomicMax(&stats[4], (unsigned)(ea * 1000.f)); atomicMax(&stats[5], (unsigned)(eb * 1000.f));
}
int main() {
    unsigned* st; hipMalloc(&st, 32); hipMemset(st, 0, 32);
    const float twopi = 6.2831854820251465f; unsigned hi; memcpy(&hi, &twopi, 4);
    // all floats from 2^-20 to 2 pi
    const unsigned lo = (127 - 20) << 23;
    unsigned long long total = (unsigned long long)hi - lo + 1;
    for (unsigned long long s = 0; s < total; s += (1ull << 28)) {
        unsigned n = (unsigned)((total - s) < (1ull << 28) ? (total - s) : (1ull << 28));
        k<<<(n + 255) / 256, 256>>>((unsigned)(lo + s), n, st);
    }
    hipDeviceSynchronize();
    unsigned h[8]; hipMemcpy(h, st, 32, hipMemcpyDeviceToHost);
    printf("values %llu: sinf max ulp %u (not correctly rounded: %u, max abs err %.3f x 2^-24); sin_0_2pi max ulp %u (not correctly rounded: %u, max abs err %.3f x 2^-24)\n",
           total, h[0], h[2], h[4] / 1000.0, h[1], h[3], h[5] / 1000.0);
    // sincos on [-pi, pi]
    hipMemset(st, 0, 32);
    const float pi = 3.1415927410125732f; unsigned hp; memcpy(&hp, &pi, 4);
    total = (unsigned long long)hp - lo + 1;
    for (unsigned long long s = 0; s < total; s += (1ull << 28)) {
        unsigned n = (unsigned)((total - s) < (1ull << 28) ? (total - s) : (1ull << 28));
        k2<<<(n + 255) / 256, 256>>>((unsigned)(lo + s), n, st);
    }
    hipDeviceSynchronize();
    hipMemcpy(h, st, 32, hipMemcpyDeviceToHost);
    printf("sincos on +-[2^-20, pi], %llu values per sign: largest absolute error of sincosf %.3f x 2^-24, of sincos_small %.3f x 2^-24\n", total, h[0] / 1000.0, h[1] / 1000.0);
    return 0;
}
