// Are (float)log2((double)x), (float)exp2((double)x), (float)exp((double)x) on the device the correctly rounded fp32
// values?  Writes the three result arrays for a fixed input grid to stdout-named file; tools/micro/f64round.py compares
// them with numpy's fp64 results rounded once.  (Diagnostic for the shift_frequency / pitch-softmax bit-identity rate.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k(const float* x, float* l2, float* e2, float* e, float* dv, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    l2[i] = (float)log2((double)x[i]);
    e2[i] = (float)exp2((double)(x[i] - 2.f));
    e[i] = (float)exp((double)(-x[i]));
    dv[i] = x[i] / 440.f;
}

int main(int argc, char** argv) {
    const int n = 1 << 20;
    std::vector<float> x(n);
    for (int i = 0; i < n; ++i) x[i] = 0.05f + 3.9f * (float)i / (float)n;
    float *dx, *d[4];
    hipMalloc(&dx, n * 4);
    for (auto& p : d) hipMalloc(&p, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, d[0], d[1], d[2], d[3], n);
    FILE* f = fopen(argc > 1 ? argv[1] : "f64round.bin", "wb");
    fwrite(x.data(), 4, n, f);
    std::vector<float> o(n);
    for (auto p : d) {
        hipMemcpy(o.data(), p, n * 4, hipMemcpyDeviceToHost);
        fwrite(o.data(), 4, n, f);
    }
    fclose(f);
    return 0;
}
