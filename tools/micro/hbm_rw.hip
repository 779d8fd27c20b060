// HBM read / write / copy rates of plain streaming kernels (16-byte accesses, grid-stride), 1 GiB buffers: what "the HBM roof" is for a
// kernel that mostly reads, mostly writes, or does both.   hipcc --offload-arch=gfx950 -O3 tools/micro/hbm_rw.hip -o tools/micro/hbm_rw_bin && tools/micro/hbm_rw_bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_write(float4* p, long n, float v) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = make_float4(v, v, v, v);
}
__global__ void k_read(const float4* p, long n, float* out) {
    float s = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 1.2345f) *out = s;
}
__global__ void k_write_nt(float4* p, long n, float v) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 x = {v, v, v, v};
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) __builtin_nontemporal_store(x, reinterpret_cast<f4*>(p) + i);
}
__global__ void k_copy_nt(const float4* a, float4* b, long n) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const f4*>(a) + i), reinterpret_cast<f4*>(b) + i);
}
__global__ void k_copy(const float4* a, float4* b, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) b[i] = a[i];
}
template <class F>
static float time_ms(F f, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}
int main() {
    const long bytes = 1L << 30, n = bytes / 16;
    float4 *a, *b;
    float* out;
    hipMalloc(&a, bytes);
    hipMalloc(&b, bytes);
    hipMalloc(&out, 4);
    hipMemset(a, 0, bytes);
    hipMemset(b, 0, bytes);
    for (int grid : {2048, 8192, 32768}) {
        const float tw = time_ms([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, a, n, 1.f); }, 10);
        const float tr = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, out); }, 10);
        const float tc = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); }, 10);
        const float twn = time_ms([&] { hipLaunchKernelGGL(k_write_nt, dim3(grid), dim3(256), 0, 0, a, n, 1.f); }, 10);
        const float tcn = time_ms([&] { hipLaunchKernelGGL(k_copy_nt, dim3(grid), dim3(256), 0, 0, a, b, n); }, 10);
        printf("grid %6d: nontemporal write %.2f TB/s   copy %.2f TB/s\n", grid, bytes / twn * 1e-9, 2.0 * bytes / tcn * 1e-9);
        printf("grid %6d: write %.2f TB/s   read %.2f TB/s   copy %.2f TB/s (read + write bytes)\n", grid, bytes / tw * 1e-9, bytes / tr * 1e-9, 2.0 * bytes / tc * 1e-9);
    }
    return 0;
}
