#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <cstring>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ref_pair(float v0, float v1, unsigned& p1, unsigned& p2) {
    f32x2 a = {v0, v1};
    f16x2v h1 = __builtin_convertvector(a, f16x2v);
    f32x2 r = (a - __builtin_convertvector(h1, f32x2)) * 2048.f;
    f16x2v h2 = __builtin_convertvector(r, f16x2v);
    p1 = __builtin_bit_cast(unsigned, h1);
    p2 = __builtin_bit_cast(unsigned, h2);
}
__device__ __forceinline__ void mix_pair(float v0, float v1, unsigned& p1, unsigned& p2) {
    f32x2 a = {v0, v1};
    f16x2v h1 = __builtin_convertvector(a, f16x2v);
    p1 = __builtin_bit_cast(unsigned, h1);
    f32x2 b = a * 2048.f;
    unsigned r;
    const float ns = -2048.f;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p1), "s"(ns), "v"(b[0]));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(p1), "s"(ns), "v"(b[1]));
    p2 = r;
}
__global__ void k(const float* x, unsigned* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    unsigned a1, a2, b1, b2;
    ref_pair(x[2 * i], x[2 * i + 1], a1, a2);
    mix_pair(x[2 * i], x[2 * i + 1], b1, b2);
    out[4 * i] = a1; out[4 * i + 1] = a2; out[4 * i + 2] = b1; out[4 * i + 3] = b2;
}
int main() {
    const int n = 1 << 22;
    std::vector<float> h(n);
    srand(3);
    for (int i = 0; i < n; ++i) {
        unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand();
        float f; memcpy(&f, &u, 4);
        int cls = i % 4;
        if (cls == 0) f = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
        else if (cls == 1) f = ldexpf(rand() / (float)RAND_MAX - 0.5f, rand() % 40 - 24);
        else if (cls == 2 && !(std::isfinite(f) && std::fabs(f) < 60000.f)) f = 1.f / (1 + rand() % 1000);
        else if (cls == 3) f = (i & 4) ? 65504.f - (rand() % 64) : -6.1e-5f * (rand() / (float)RAND_MAX);
        h[i] = f;
    }
    float* dx; unsigned* dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 8);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 2 / 256, 256>>>(dx, dout, n);
    std::vector<unsigned> o(2 * n);
    hipMemcpy(o.data(), dout, n * 8, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < n / 2; ++i) if (o[4 * i] != o[4 * i + 2] || o[4 * i + 1] != o[4 * i + 3]) { if (bad < 5) printf("diff at %d: x=%g,%g ref %08x %08x mix %08x %08x\n", i, h[2*i], h[2*i+1], o[4*i], o[4*i+1], o[4*i+2], o[4*i+3]); ++bad; }
    printf("pairs %d, mismatches %ld\n", n / 2, bad);
    return 0;
}
