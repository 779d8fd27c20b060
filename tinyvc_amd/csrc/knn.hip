// kNN feature match (feature_retrieval.py:15-33, metrics='cos', k=4, alpha=0).
//
// The reference materialises sims[T][N] and calls torch.topk; here the index streams through the matrix pipe in
// 128-vector tiles and every lane keeps a running top-4 for the query column it owns in the MFMA accumulator layout
// (index vectors are the M axis, queries the N axis, so the 16 accumulator registers of a lane are 16 index vectors
// against ONE query: the reduction is lane-local).  The index is split across workgroups for occupancy; a second kernel
// merges the per-split candidates, emits int64 indices and gathers + averages the 4 raw index vectors.
//
// Similarities run on the split-precision path (conv3s.h): bf16 part-products, fp32 accumulation, error below an fp32
// FMA chain's.  Two index storages share the kernels (the prepared blob is self-describing, so every entry point - and a
// captured HIP graph - works with either):
//   kind 0, fp32 storage: raw rows [N][768] fp32 (the final gather) + the cosine-normalised vectors as three bf16 parts
//           per value in MFMA lane order (10 B per element); six part-products per product; bit-exact indices on the
//           gap-checked fixtures.
//   kind 1, fp16 storage (BASELINE configs[4]: a 1 M-vector index): the raw vectors as fp16 in MFMA lane
//           order + one fp32 inverse norm per vector (2 B per element: 1.5 GB at N = 1 M).  An fp16 value is exactly two
//           bf16 parts, split while a tile is staged into LDS; five part-products per product; the similarity is
//           dot(q_hat, r) * inv_norm; the gather reads the same image.
// Tie-break: equal similarities -> lower index first (torch.topk leaves it unspecified).
#include <hip/hip_fp16.h>

#include "conv3s.h"
#include "small_kernels.h"
#include "tvc_common.h"

namespace tvc {

constexpr int KD = kSslDim;        // 768
constexpr int STEPS = KD / 16;     // 48 K16 steps per index tile
constexpr int HDR = 64;            // blob header, floats: [0] magic, [1] kind, [2] N (low 32 bits), [3] N (high), [4] |max| of the raw vectors (a float), [5] format version (tvc_common.h)
constexpr int KIND_F32 = 0, KIND_F16 = 1;
constexpr int BLOB_MAGIC = kBlobMagic;
// fp32 kind:  header | rows fp32 [N][768] | bf16x3 image of v / den [Npad*768*3 bf16] | inv = 1 / den [Npad] | fp16 image of v / den [Npad*768]
// fp16 kind:  header | inv [Npad] | fp16 image of the raw vectors [Npad*768] | largest inv of every 128-vector tile [Npad/128]
// (both fp16 images in the 128-vector-tiled MFMA lane order, one part)
__host__ __device__ inline const float* blob_inv(const float* blob, int kind, long N, long Npad) {
    return kind == KIND_F16 ? blob + HDR : blob + HDR + (size_t)N * KD + (size_t)Npad * KD * 3 / 2;
}
__host__ __device__ inline const uint4* blob_img16(const float* blob, int kind, long N, long Npad) {
    return reinterpret_cast<const uint4*>(blob_inv(blob, kind, N, Npad) + Npad);
}
__host__ __device__ inline const float* blob_invmax(const float* blob, long Npad) {      // fp16 kind only
    return blob + HDR + Npad + (size_t)Npad * KD / 2;
}
#ifndef KNN_BLOCKS
#define KNN_BLOCKS 1024   // target workgroup count (query tiles x index splits)
#endif

static inline int64_t npad128(int64_t N) { return (N + 127) / 128 * 128; }

// element (vector n, channel k) of a 128-vector-tiled MFMA-ordered image with P parts per (m-tile, step): the index of
// part 0's 8-value piece row; row = lane & 31, k = 16 step + 8 (lane >> 5) + j
__device__ __forceinline__ long img_elem(long n, int k, int parts) {
    const long tile = n >> 7;
    const int mt = (int)(n & 127) >> 5, l31 = (int)(n & 31);
    const int step = k >> 4, lh = (k >> 3) & 1, j = k & 7;
    return ((((tile * STEPS + step) * 4 + mt) * parts) * 64 + (lh * 32 + l31)) * 8 + j;
}

// raw vector value (n, k) of either blob kind (the gathers)
__device__ __forceinline__ float blob_row_value(const float* __restrict__ blob, int kind, long N, long Npad, long n, int k) {
    if (kind == KIND_F16) return __half2float(reinterpret_cast<const __half*>(blob + HDR + Npad)[img_elem(n, k, 1)]);
    return blob[HDR + n * KD + k];
}

static __global__ void blob_header_kernel(float* blob, int kind, long N) {
    int* h = reinterpret_cast<int*>(blob);
    if (threadIdx.x < HDR) h[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        h[0] = BLOB_MAGIC;
        h[1] = kind;
        h[2] = (int)(N & 0xffffffffL);
        h[3] = (int)(N >> 32);
        h[5] = kBlobVersion;
    }
}

// fp32 storage.  index [768][N] (the [1,768,N] tensor of index.pt) -> raw rows + the bf16x3 image of v / (||v|| + 1e-6)
// (feature_retrieval.py:25 recomputes that normalisation on every call).  One thread per vector; reads run along n.
static __global__ void index_prepare_kernel(const float* __restrict__ index, float* __restrict__ rows,
                                            unsigned short* __restrict__ img, float* __restrict__ inv, __half* __restrict__ img16,
                                            long N, long Npad) {
    long n = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (n >= Npad) return;
    float den = 1.f;
    if (n < N) {
        float s = 0.f;
        for (int k = 0; k < KD; ++k) {
            float v = index[(long)k * N + n];
            s = fmaf(v, v, s);
        }
        den = sqrtf(s) + 1e-6f;
    }
    inv[n] = n < N ? 1.f / den : 0.f;
    for (int k = 0; k < KD; ++k) {
        float raw = n < N ? index[(long)k * N + n] : 0.f;
        if (n < N) rows[n * KD + k] = raw;
        float v = raw / den;
        img16[img_elem(n, k, 1)] = __float2half(v);      // the coarse pass's operand (knn_coarse_kernel)
        __bf16 h1 = (__bf16)v;
        float r = v - (float)h1;
        __bf16 h2 = (__bf16)r;
        float r2 = r - (float)h2;
        __bf16 h3 = (__bf16)r2;
        const long base = img_elem(n, k, 3);
        img[base] = __builtin_bit_cast(unsigned short, h1);
        img[base + 512] = __builtin_bit_cast(unsigned short, h2);
        img[base + 1024] = __builtin_bit_cast(unsigned short, h3);
    }
}

// fp16 storage.  rows16 [N][768] IEEE half (row-major: one vector per row) -> inverse norms + the fp16 image.
// One wavefront per vector: lanes run along k (coalesced 128-byte reads), the norm is a fixed-order wave reduction.
static __global__ __launch_bounds__(256) void index_prepare_f16_kernel(const __half* __restrict__ rows16, float* __restrict__ inv,
                                                                       __half* __restrict__ img, long N, long Npad) {
    const int lane = threadIdx.x & 63;
    const long n = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    if (n >= Npad) return;
    float s = 0.f;
    for (int k = lane; k < KD; k += 64) {
        const __half h = n < N ? rows16[n * KD + k] : __float2half(0.f);
        const float v = __half2float(h);
        s = fmaf(v, v, s);
        img[img_elem(n, k, 1)] = h;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) inv[n] = n < N ? 1.f / (sqrtf(s) + 1e-6f) : 0.f;
}

// header[4] = the largest |value| of the index: `matched` (the mean of four of its rows) is bounded by it, so the conversion takes the
// |max| slot of the decoder's content input from here instead of a pass over the tensor (block-floating-point guard, conv3s.h)
template <class T>
static __global__ __launch_bounds__(256) void index_amax_kernel(const T* __restrict__ p, long n, float* __restrict__ slot) {
    __shared__ float red[4];
    float mx = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) mx = fmaxf(mx, fabsf((float)p[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (m > 0.f) atomicMax(reinterpret_cast<unsigned*>(slot), __builtin_bit_cast(unsigned, m));      // non-negative floats order like their bits; NaN never enters
    }
}
const float* knn_index_amax(const float* prepared) { return prepared + 4; }

int run_prepare_index(tvc_ctx* ctx, hipStream_t s, const float* index, float* prepared, int64_t N) {
    const long Npad = npad128(N);
    hipLaunchKernelGGL(blob_header_kernel, dim3(1), dim3(64), 0, s, prepared, KIND_F32, (long)N);
    float* rows = prepared + HDR;
    unsigned short* img = reinterpret_cast<unsigned short*>(rows + (size_t)N * KD);
    float* inv = const_cast<float*>(blob_inv(prepared, KIND_F32, N, Npad));
    __half* img16 = reinterpret_cast<__half*>(const_cast<uint4*>(blob_img16(prepared, KIND_F32, N, Npad)));
    hipLaunchKernelGGL(index_prepare_kernel, dim3((unsigned)((Npad + 255) / 256)), dim3(256), 0, s, index, rows, img, inv, img16, (long)N, Npad);
    hipLaunchKernelGGL(index_amax_kernel<float>, dim3(64), dim3(256), 0, s, index, (long)N * KD, prepared + 4);
    return launch_check(ctx, "knn_prepare_index");
}

static __global__ void index_invmax_kernel(const float* __restrict__ inv, float* __restrict__ invmax, long ntiles) {
    const long t = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    float m = 0.f;
    for (int i = 0; i < 128; ++i) m = fmaxf(m, inv[t * 128 + i]);
    invmax[t] = m;
}

int run_prepare_index_f16(tvc_ctx* ctx, hipStream_t s, const void* rows16, float* prepared, int64_t N) {
    const long Npad = npad128(N);
    hipLaunchKernelGGL(blob_header_kernel, dim3(1), dim3(64), 0, s, prepared, KIND_F16, (long)N);
    float* inv = prepared + HDR;
    __half* img = reinterpret_cast<__half*>(inv + Npad);
    hipLaunchKernelGGL(index_prepare_f16_kernel, dim3((unsigned)((Npad * 64 + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const __half*>(rows16), inv, img, (long)N, Npad);
    hipLaunchKernelGGL(index_invmax_kernel, dim3((unsigned)((Npad / 128 + 255) / 256)), dim3(256), 0, s, inv, const_cast<float*>(blob_invmax(prepared, Npad)), Npad / 128);
    hipLaunchKernelGGL(index_amax_kernel<__half>, dim3(64), dim3(256), 0, s, reinterpret_cast<const __half*>(rows16), (long)N * KD, prepared + 4);
    return launch_check(ctx, "knn_prepare_index_f16");
}

// qn[b][k][t] = src[b][k][t] / (||src[b][:][t]|| + 1e-6).  One workgroup = 64 consecutive columns;
// its 4 waves each sum a quarter of the 768 channels (lanes along time, coalesced), partial sums of
// squares meet in LDS in a fixed order, then every wave rescales its quarter.
// qh (optional) = the same values in fp16, in the coarse pass's B-fragment order [256-query tile][K16 step][8-channel half][query][8]
// (columns beyond ncols of the last tile are zero); cnt / flag (optional) = the two-stage search's per-query candidate
// counters and its overflow flag, zeroed here.
constexpr int QN_WAVES = 16;     // waves per 64-column group: each squares / scales KD / 16 = 48 channels (strided rows: the loop is a latency chain)
static __global__ __launch_bounds__(QN_WAVES * 64) void query_normalize_kernel(const float* __restrict__ src, float* __restrict__ qn, int B, int T,
                                                                     uint4* __restrict__ qh, int* __restrict__ cnt, int* __restrict__ flag) {
    __shared__ float part[QN_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long ncols = (long)B * T;
    const long n = blockIdx.x * 64L + lane;
    const bool ok = n < ncols;
    const long nn = ok ? n : ncols - 1;
    const int b = (int)(nn / T), t = (int)(nn - (long)b * T);
    const float* p = src + (long)b * KD * T + t;
    float* q = qn + (long)b * KD * T + t;
    constexpr int KW = KD / QN_WAVES;
    const int k0 = wave * KW;
    float x[KW];       // the wave's 48 channels stay in registers: requested together (the summing loop was a chain of load latencies), read once
#pragma unroll
    for (int i = 0; i < KW; ++i) x[i] = p[(long)(k0 + i) * T];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < KW; ++i) s = fmaf(x[i], x[i], s);
    part[wave][lane] = s;
    __syncthreads();
    float ss = part[0][lane];
#pragma unroll
    for (int w = 1; w < QN_WAVES; ++w) ss += part[w][lane];      // fixed order
    const float den = sqrtf(ss) + 1e-6f;
    if (blockIdx.x == 0 && threadIdx.x == 0 && flag) *flag = 0;
    if (wave == 0 && ok && cnt) cnt[n] = 0;
    uint4* qhp = qh ? qh + (n >> 8) * (long)(STEPS * 2 * 256) + (n & 255) : nullptr;
#pragma unroll
    for (int i = 0; i < KW; i += 8) {
        const int k = k0 + i;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ok ? x[i + j] / den : 0.f;
        if (ok) {
#pragma unroll
            for (int j = 0; j < 8; ++j) q[(long)(k + j) * T] = v[j];
        }
        if (qh) {
            unsigned o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const __half2 h = __floats2half2_rn(v[2 * j], v[2 * j + 1]);
                o[j] = (unsigned)__half_as_ushort(__low2half(h)) | ((unsigned)__half_as_ushort(__high2half(h)) << 16);
            }
            qhp[(long)(k >> 3) * 256] = make_uint4(o[0], o[1], o[2], o[3]);     // (K16 step, half) = k / 8
        }
    }
}

// torch.topk orders NaN above every number; a query column with NaN / Inf samples upstream makes every similarity NaN.
// Mapping NaN to +inf keeps that order (ties -> lowest index, so such a column selects rows 0..3 like any all-equal
// column) and, more to the point, keeps the 0x7fffffff list sentinel from ever reaching the row gather.
__device__ __forceinline__ float nan_max(float x) { return x != x ? INFINITY : x; }

struct Top4 {
    float v[4];
    int i[4];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = -INFINITY;
            i[j] = 0x7fffffff;
        }
    }
    // strict ordering: higher value first, then lower index
    __device__ __forceinline__ static bool better(float a, int ia, float b, int ib) { return a > b || (a == b && ia < ib); }
    __device__ __forceinline__ void insert(float x, int ix) {
        if (!better(x, ix, v[3], i[3])) return;
        if (better(x, ix, v[0], i[0])) { v[3] = v[2]; i[3] = i[2]; v[2] = v[1]; i[2] = i[1]; v[1] = v[0]; i[1] = i[0]; v[0] = x; i[0] = ix; }
        else if (better(x, ix, v[1], i[1])) { v[3] = v[2]; i[3] = i[2]; v[2] = v[1]; i[2] = i[1]; v[1] = x; i[1] = ix; }
        else if (better(x, ix, v[2], i[2])) { v[3] = v[2]; i[3] = i[2]; v[2] = x; i[2] = ix; }
        else { v[3] = x; i[3] = ix; }
    }
};

// The exact kernel's own split: three bf16 parts per fp32 operand (x = x1 + x2 + x3, residuals exact), six part-products per
// product on v_mfma_f32_32x32x16_bf16.  (The conv / GEMM kernels moved to the two-term fp16 split of conv3s.h; this kernel only runs
// for N < 4096 and as the in-call fallback of the two-stage search, whose coarse passes are single fp16 products.)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split8_bf3(const float (&v)[8], uint4& p1, uint4& p2, uint4& p3) {
    unsigned o1[4], o2[4], o3[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x2 a = {v[2 * j], v[2 * j + 1]};
        bf16x2 h1 = __builtin_convertvector(a, bf16x2);
        f32x2 r = a - __builtin_convertvector(h1, f32x2);
        bf16x2 h2 = __builtin_convertvector(r, bf16x2);
        f32x2 r2 = r - __builtin_convertvector(h2, f32x2);
        bf16x2 h3 = __builtin_convertvector(r2, bf16x2);
        o1[j] = __builtin_bit_cast(unsigned, h1);
        o2[j] = __builtin_bit_cast(unsigned, h2);
        o3[j] = __builtin_bit_cast(unsigned, h3);
    }
    p1 = make_uint4(o1[0], o1[1], o1[2], o1[3]);
    p2 = make_uint4(o2[0], o2[1], o2[2], o2[3]);
    p3 = make_uint4(o3[0], o3[1], o3[2], o3[3]);
}
// two bf16 parts of 8 fp16 values (exact: 11 significant bits fit in 8 + 8), packed for one 16-byte LDS row each
__device__ __forceinline__ void split8_half(const u32x4 h8, uint4& p1, uint4& p2) {
    const f16x8 hv = __builtin_bit_cast(f16x8, h8);
    unsigned o1[4], o2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x2 a = {(float)hv[2 * j], (float)hv[2 * j + 1]};
        bf16x2 h1 = __builtin_convertvector(a, bf16x2);
        f32x2 r = a - __builtin_convertvector(h1, f32x2);
        bf16x2 h2 = __builtin_convertvector(r, bf16x2);
        o1[j] = __builtin_bit_cast(unsigned, h1);
        o2[j] = __builtin_bit_cast(unsigned, h2);
    }
    p1 = make_uint4(o1[0], o1[1], o1[2], o1[3]);
    p2 = make_uint4(o2[0], o2[1], o2[2], o2[3]);
}

// grid = qtiles * nsplit; workgroup = KNN_QT queries x (tiles_per_split index tiles of 128).
// 8 waves as 2 (index halves) x 4 (query quarters); a wave owns 64 index vectors x 64 queries = 2 x 2 MFMA tiles, so every
// A and B fragment read from LDS feeds two MFMAs (0.5 KiB of LDS reads per MFMA; the 64 x 32 wave tile of round 1 read
// 0.75) and a (tile, step) carries 24 MFMAs per wave between barriers.  The (tile, step) sequence is one flat pipeline:
// registers hold step g+1, LDS is double-buffered, one raw barrier per step.
// NP = bf16 parts per index value in LDS (3: fp32 storage, 2: fp16 storage).
constexpr int KNN_QT = 256;                                      // queries per workgroup
constexpr int KNN_A_U4 = 12 * 64, KNN_X_U4 = 3 * 2 * KNN_QT;    // one LDS buffer each: 12 KiB + 24 KiB
template <bool F16>
__device__ __forceinline__ void knn_topk_body(const float* __restrict__ blob, long Npad, int N, const float* __restrict__ qn, int ncols, int T,
                                              int nsplit, int tiles_per_split, float* __restrict__ cand_v, int* __restrict__ cand_i,
                                              uint4* smem) {
    constexpr int NP = F16 ? 2 : 3;                    // parts per index value in LDS
    constexpr int GP = F16 ? 4 : 12;                   // 1 KiB pieces per (tile, step) in the global image
    uint4* As = smem;
    uint4* Xs = smem + 2 * KNN_A_U4;
    const uint4* img = F16 ? reinterpret_cast<const uint4*>(blob + HDR + Npad) : reinterpret_cast<const uint4*>(blob + HDR + (size_t)N * KD);
    const float* inv = blob + HDR;                     // fp16 storage only

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, lh = lane >> 5;
    const int split = blockIdx.x % nsplit;
    const int qtile = blockIdx.x / nsplit;
    const int n0 = qtile * KNN_QT;
    const int mtiles = (int)(Npad >> 7);
    const int mt_lo = split * tiles_per_split;
    const int mt_hi = min(mtiles, mt_lo + tiles_per_split);
    const int G = (mt_hi - mt_lo) * STEPS;               // flat (tile, step) sequence

    // staging roles: every thread owns one query item (8 channels of one column: 2 halves x 256 columns); the index pieces of
    // a (tile, step) go to waves p % 8 (fp32 storage: 12 pieces, waves 0..3 take two) or to waves 0..3 (fp16 storage: one
    // fp16 piece each, split into its two bf16 parts on the way into LDS)
    const float* qp;
    int xdst;
    {
        const int g = tid >> 8, pos = tid & 255;
        int n = n0 + pos;
        n = n < ncols ? n : ncols - 1;
        const int b = n / T, t = n - b * T;
        qp = qn + ((long)b * KD + 8 * g) * T + t;
        xdst = g * KNN_QT + pos;
    }
    const bool a2 = !F16 && wave < 4;                  // this wave stages a second piece (wave + 8)
    const bool a1 = !F16 || wave < 4;
    float xr[8];
    u32x4 ar[2];
    auto gload = [&](int g) __attribute__((always_inline)) {
        const int mt = mt_lo + g / STEPS, st = g - (g / STEPS) * STEPS;
        const uint4* src = img + ((long)mt * STEPS + st) * (GP * 64) + lane;
        if (a1) ar[0] = *reinterpret_cast<const u32x4*>(src + wave * 64);
        if (a2) ar[1] = *reinterpret_cast<const u32x4*>(src + (wave + 8) * 64);
#pragma unroll
        for (int j = 0; j < 8; ++j) xr[j] = qp[(long)(st * 16 + j) * T];
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {
        uint4* ab = As + buf * KNN_A_U4;
        if (F16) {
            if (a1) {
                uint4 p1, p2;
                split8_half(ar[0], p1, p2);
                ab[(wave * 2) * 64 + lane] = p1;
                ab[(wave * 2 + 1) * 64 + lane] = p2;
            }
        } else {
            *reinterpret_cast<u32x4*>(ab + wave * 64 + lane) = ar[0];
            if (a2) *reinterpret_cast<u32x4*>(ab + (wave + 8) * 64 + lane) = ar[1];
        }
        uint4 p1, p2, p3;
        split8_bf3(xr, p1, p2, p3);
        uint4* xb = Xs + buf * KNN_X_U4;
        xb[xdst] = p1;
        xb[2 * KNN_QT + xdst] = p2;
        xb[4 * KNN_QT + xdst] = p3;
    };

    Top4 top[2];
    top[0].init();
    top[1].init();
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (G > 0) {
        gload(0);
        lstore(0);
        if (G > 1) gload(1);
        slab_barrier();
    }
    for (int g = 0; g < G; ++g) {
        const int cur = g & 1;
        if (g + 1 < G) {
            lstore(cur ^ 1);                        // step g+1 (its buffer was last read in step g-1, behind the barrier)
            if (g + 2 < G) gload(g + 2);            // flies across this step's MFMAs and the next barrier
        }
        const uint4* as = As + cur * KNN_A_U4 + wm * (2 * NP * 64) + lane;
        const uint4* xs = Xs + cur * KNN_X_U4 + lh * KNN_QT + wn * 64 + l31;
        bf16x8 af[2][NP], bf[2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) bf[j][p] = __builtin_bit_cast(bf16x8, xs[p * 2 * KNN_QT + j * 32]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < NP; ++p) af[i][p] = __builtin_bit_cast(bf16x8, as[(i * NP + p) * 64]);
        // part-products, least significant first: (index part, query part)
#ifdef KNN_NQ_TEST
        constexpr int NQ = KNN_NQ_TEST;      // timing experiment only (wrong results): fewer part-products
#else
        constexpr int NQ = F16 ? 5 : 6;
#endif
        constexpr int PA3[6] = {2, 1, 0, 1, 0, 0}, PB3[6] = {0, 1, 2, 0, 1, 0};
        constexpr int PA2[5] = {1, 0, 1, 0, 0}, PB2[5] = {1, 2, 0, 1, 0};
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][F16 ? PA2[q < 5 ? q : 0] : PA3[q]], bf[j][F16 ? PB2[q < 5 ? q : 0] : PB3[q]],
                                                                        acc[i][j], 0, 0, 0);
        const int st = g - (g / STEPS) * STEPS;
        if (st == STEPS - 1) {
            // running top-4: a lane's 16 registers of one MFMA tile are 16 index vectors against ONE of its two queries
            const int m0 = (mt_lo + g / STEPS) * 128;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row < N) {
                        const float sc = F16 ? inv[row] : 1.f;
#pragma unroll
                        for (int j = 0; j < 2; ++j) top[j].insert(nan_max(F16 ? acc[i][j][r] * sc : acc[i][j][r]), row);
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j][r] = 0.f;
                }
        }
        slab_barrier();
    }

    // merge the 4 partial lists (wm in {0,1} x lh in {0,1}) of every query through LDS (the staging buffers are free now:
    // the loop's last barrier is behind every wave's last fragment read)
    float (*mv)[16] = reinterpret_cast<float (*)[16]>(smem);
    int (*mi)[16] = reinterpret_cast<int (*)[16]>(reinterpret_cast<float*>(smem) + KNN_QT * 16);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = wn * 64 + j * 32 + l31;
        const int slot = (wm * 2 + lh) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mv[q][slot + e] = top[j].v[e];
            mi[q][slot + e] = top[j].i[e];
        }
    }
    __syncthreads();
    if (tid < KNN_QT) {
        Top4 t4;
        t4.init();
        for (int e = 0; e < 16; ++e) t4.insert(mv[tid][e], mi[tid][e]);
        int n = n0 + tid;
        if (n < ncols) {
            long o = ((long)split * ncols + n) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                cand_v[o + e] = t4.v[e];
                cand_i[o + e] = t4.i[e];
            }
        }
    }
}

static __global__ __launch_bounds__(512) void knn_topk_split_kernel(const float* __restrict__ blob, long Npad, int N,
                                                                    const float* __restrict__ qn, int ncols, int T,
                                                                    int nsplit, int tiles_per_split,
                                                                    float* __restrict__ cand_v, int* __restrict__ cand_i,
                                                                    const int* __restrict__ run_flag) {
    __shared__ __attribute__((aligned(16))) uint4 smem[2 * (KNN_A_U4 + KNN_X_U4)];      // 72 KiB; the final merge reuses 32 KiB of it
    if (run_flag && *run_flag == 0) return;                      // two-stage search succeeded: nothing to do (uniform)
    const int kind = reinterpret_cast<const int*>(blob)[1];      // uniform: which storage this prepared index uses
    if (kind == KIND_F16) knn_topk_body<true>(blob, Npad, N, qn, ncols, T, nsplit, tiles_per_split, cand_v, cand_i, smem);
    else knn_topk_body<false>(blob, Npad, N, qn, ncols, T, nsplit, tiles_per_split, cand_v, cand_i, smem);
}

// One workgroup = 32 consecutive query columns: merge split candidates -> top-4, write indices,
// gather the 4 raw rows per query (coalesced along the feature axis), average, and write
// out[b][k][t] through an LDS transpose so stores run along t.
static __global__ __launch_bounds__(256) void knn_merge_gather_kernel(const float* __restrict__ cand_v, const int* __restrict__ cand_i,
                                                                      int nsplit, int ncols, int T, int N, long Npad,
                                                                      const float* __restrict__ blob,
                                                                      float* __restrict__ out, int64_t* __restrict__ idx_out,
                                                                      const float* __restrict__ rv, const int* __restrict__ ri,
                                                                      const int* __restrict__ flag) {
    __shared__ int sel[32][4];
    __shared__ float tile[32][193];
    if (flag && *flag == 0) {      // the two-stage search's rescored lists (one "split") are the result
        cand_v = rv;
        cand_i = ri;
        nsplit = 1;
    }
    const int tid = threadIdx.x;
    const int n0 = blockIdx.x * 32;
    const int kind = reinterpret_cast<const int*>(blob)[1];
    if (tid < 32) {
        int n = n0 + tid;
        Top4 t4;
        t4.init();
        if (n < ncols) {
            for (int s = 0; s < nsplit; ++s) {
                long o = ((long)s * ncols + n) * 4;
                for (int e = 0; e < 4; ++e) t4.insert(cand_v[o + e], cand_i[o + e]);
            }
            for (int e = 0; e < 4; ++e) t4.i[e] = (unsigned)t4.i[e] < (unsigned)N ? t4.i[e] : 0;   // never gather through a sentinel
            if (idx_out)
                for (int e = 0; e < 4; ++e) idx_out[(long)n * 4 + e] = (int64_t)t4.i[e];
        }
        for (int e = 0; e < 4; ++e) sel[tid][e] = n < ncols ? t4.i[e] : 0;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    for (int kc = 0; kc < KD; kc += 192) {
        // gather: wave handles queries wave, wave+4, ...; lanes run along k (3 x 64 = 192)
        // (four queries' 48 loads in flight per lane: one query at a time was eight serial round trips per wave and chunk - 48 us per
        // launch whatever the batch, a chain of latencies)
        for (int q0 = wave; q0 < 32; q0 += 16) {
            float r[4][3][4];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                for (int u = 0; u < 3; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[qq][u][e] = blob_row_value(blob, kind, N, Npad, sel[q0 + 4 * qq][e], kc + lane + 64 * u);
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const float sum = __fadd_rn(__fadd_rn(__fadd_rn(r[qq][u][0], r[qq][u][1]), r[qq][u][2]), r[qq][u][3]);
                    tile[q0 + 4 * qq][lane + 64 * u] = sum * 0.25f;
                }
        }
        __syncthreads();
        // scatter: lanes run along the 32 queries (time), 8 k-rows per pass
        for (int kk = tid >> 5; kk < 192; kk += 8) {
            int q = tid & 31;
            int n = n0 + q;
            if (n < ncols) {
                int b = n / T, t = n - b * T;
                out[((long)b * KD + kc + kk) * T + t] = tile[q][kk];
            }
        }
        __syncthreads();
    }
}

// ---- index-sharded search (one index shard per GPU): local top-4 with similarities, slot gather, finish ----
// merge the split candidates of every query -> this shard's top-4 (similarity, local index)
static __global__ __launch_bounds__(256) void knn_merge_kernel(const float* __restrict__ cand_v, const int* __restrict__ cand_i, int nsplit, int ncols,
                                                               float* __restrict__ sims_out, int64_t* __restrict__ idx_out,
                                                               const float* __restrict__ rv, const int* __restrict__ ri, const int* __restrict__ flag) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= ncols) return;
    if (flag && *flag == 0) {
        cand_v = rv;
        cand_i = ri;
        nsplit = 1;
    }
    Top4 t4;
    t4.init();
    for (int sp = 0; sp < nsplit; ++sp) {
        long o = ((long)sp * ncols + n) * 4;
        for (int e = 0; e < 4; ++e) t4.insert(cand_v[o + e], cand_i[o + e]);
    }
    for (int e = 0; e < 4; ++e) {
        sims_out[(long)n * 4 + e] = t4.v[e];
        idx_out[(long)n * 4 + e] = (int64_t)t4.i[e];
    }
}
// slots[n][e][:] = raw row idx[n][e] of this shard, or zeros where idx < 0 (the row lives on another rank)
static __global__ __launch_bounds__(192) void knn_slot_gather_kernel(const float* __restrict__ blob, const int64_t* __restrict__ idx, long nslots,
                                                                     long N, long Npad, float* __restrict__ slots) {
    const long sl = blockIdx.x;
    if (sl >= nslots) return;
    const int kind = reinterpret_cast<const int*>(blob)[1];
    const int64_t i = idx[sl];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i >= 0 && i < N) {
        const int k = 4 * threadIdx.x;
        v = make_float4(blob_row_value(blob, kind, N, Npad, i, k), blob_row_value(blob, kind, N, Npad, i, k + 1),
                        blob_row_value(blob, kind, N, Npad, i, k + 2), blob_row_value(blob, kind, N, Npad, i, k + 3));
    }
    reinterpret_cast<float4*>(slots + sl * KD)[threadIdx.x] = v;
}
// out[b][k][t] = (((s0 + s1) + s2) + s3) * 0.25 from slots [B*T][4][768] (the same order as the single-GPU gather),
// transposed through LDS so reads run along k and stores along t
static __global__ __launch_bounds__(256) void knn_finish_kernel(const float* __restrict__ slots, int ncols, int T, float* __restrict__ out) {
    __shared__ float tile[32][193];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 32;
    for (int kc = 0; kc < KD; kc += 192) {
        for (int q = wave; q < 32; q += 4) {
            const int n = n0 + q < ncols ? n0 + q : ncols - 1;
            const float* r0 = slots + ((long)n * 4) * KD + kc;
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                int k = lane + 64 * u;
                float sum = __fadd_rn(__fadd_rn(__fadd_rn(r0[k], r0[KD + k]), r0[2 * KD + k]), r0[3 * KD + k]);
                tile[q][k] = sum * 0.25f;
            }
        }
        __syncthreads();
        for (int kk = tid >> 5; kk < 192; kk += 8) {
            int q = tid & 31;
            int n = n0 + q;
            if (n < ncols) {
                int b = n / T, t = n - b * T;
                out[((long)b * KD + kc + kk) * T + t] = tile[q][kk];
            }
        }
        __syncthreads();
    }
}

// =================================================================================================
// Two-stage search (N >= KNN_COARSE_MIN).  The exact kernel above spends six bf16 part-products per similarity on every
// (query, index vector) pair although only a handful of index vectors per query can be in the top 4.  Stage 1 computes a
// COARSE similarity c with ONE fp16 product per element (fp16 has 11 significant bits: |c - s| <= eps for unit vectors, see
// C_EPS) and keeps, per query, every index vector that could still be in the exact top 4:
//   pass A  coarse top-4 values over a sample of the index (any subset's 4th best is a lower bound of the global 4th
//           best) -> theta = c4 - 2 eps;
//   pass B  coarse similarities of the whole index; rows with c >= theta are appended to the query's candidate list
//           (C_CAP entries; a handful for random data);
//   rescore exact fp32 similarity of every candidate (fp32 FMA chain over the raw vectors, x 1 / norm), top-4 with the
//           library's order (similarity descending, lower index first).
// Why this is exact: let R be the final top-4 (by the rescored similarity s') and T4 the coarse top-4 of the sample.
// For r in R: s'_r >= 4th largest s' overall >= min_{t in T4} s'_t >= c4 - eps, hence c_r >= s'_r - eps >= c4 - 2 eps = theta,
// so r is a candidate.  If any query collects more than C_CAP candidates (dense neighbourhoods), a flag makes the exact
// kernel run instead (it is always launched and exits at once when the flag is clear), so results never depend on the
// data distribution - only the speed does.
// =================================================================================================
constexpr int C_QT = 256, C_MT = 256, C_K = 64, C_STEPS = KD / C_K;            // 256 queries x 256 index vectors x K = 64 per step
constexpr int C_A_U4 = 4 * 8 * 64, C_X_U4 = 4 * 2 * C_QT;                      // one LDS buffer each: 32 KiB + 32 KiB
constexpr int C_LDS = 2 * (C_A_U4 + C_X_U4) * 16;                              // 128 KiB, double-buffered
constexpr int C_CAP = 256;                                                     // candidates per query before the exact fallback (theta comes from a sample, so lists run to a few dozen)
// |coarse - exact| for unit vectors: both operands rounded to fp16 (2^-11 relative each, 2^-25 absolute below the normal
// range), fp32 accumulation of 768 exact products (<= 768 * 2^-24), the rescoring's own rounding (< 1e-6):
// 2^-10 * 1.01 + 768 * 2^-25 + 768 * 2^-24 + 1e-6 < 1.06e-3; C_EPS leaves a margin.
constexpr float C_EPS = 1.25e-3f;
#ifndef KNN_COARSE_MIN
#define KNN_COARSE_MIN 4096
#endif

struct Top4V {   // four largest values
    float v[4];
    __device__ __forceinline__ void init() { v[0] = v[1] = v[2] = v[3] = -INFINITY; }
    __device__ __forceinline__ void insert(float x) {
        if (!(x > v[3])) return;
        if (x > v[0]) { v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = x; }
        else if (x > v[1]) { v[3] = v[2]; v[2] = v[1]; v[1] = x; }
        else if (x > v[2]) { v[3] = v[2]; v[2] = x; }
        else v[3] = x;
    }
};

// grid = qtiles * nsplit; a workgroup owns 256 queries and walks `tiles_per_split` 256-vector index tiles (of the first
// `t2_cover` tiles).  8 waves as 2 (index halves) x 4 (query quarters), a wave owns 128 index vectors x 64 queries = 4 x 2
// MFMA tiles (v_mfma_f32_32x32x16_f16); both operands arrive as ready 16-byte fragments rows (the index's fp16 image,
// the queries' fp16 image), so staging is copies only: 8 loads + 8 ds_write_b128 per thread and K = 64 step, 32 MFMAs per
// wave and step.  MODE 0: coarse top-4 values per (split, query) -> c4v.  MODE 1: rows with c >= theta -> candidate lists.
template <bool F16, int MODE>
static __device__ __forceinline__ void knn_coarse_body(const float* __restrict__ blob, long Npad, int N, const uint4* __restrict__ qh,
                                                       int ncols, int nsplit, int tiles_per_split, int t2_cover,
                                                       float* __restrict__ c4v, int ns_sample,
                                                       int* __restrict__ cnt, int* __restrict__ cand, float* __restrict__ candv,
                                                       int* __restrict__ overflow) {
    extern __shared__ __attribute__((aligned(16))) uint4 csmem[];
    uint4* As = csmem;
    uint4* Xs = csmem + 2 * C_A_U4;
    const int kind = F16 ? KIND_F16 : KIND_F32;
    const uint4* img = blob_img16(blob, kind, N, Npad);
    const float* inv = blob_inv(blob, kind, N, Npad);          // fp16 storage: the image holds the raw vectors

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, lh = lane >> 5;
    const int split = blockIdx.x % nsplit;
    const int qtile = blockIdx.x / nsplit;
    const int n0 = qtile * C_QT;
    const int mtiles = (int)(Npad >> 7);                         // 128-vector tiles of the image
    const int t_lo = split * tiles_per_split;
    const int t_hi = min(t2_cover, t_lo + tiles_per_split);
    const int G = (t_hi - t_lo) * C_STEPS;

    // staging: thread tid copies uint4 number tid + 512 i (i < 4) of the step's A block [k16][m-tile (8)][lane] and of its
    // X block [k16][8-channel half][query] - the LDS images are linear in exactly that order
    const uint4* xsrc = qh + (long)qtile * (STEPS * 2 * 256) + (tid & 511);
    u32x4 ar[4], xr[4];
    auto gload = [&](int g) __attribute__((always_inline)) {
        const int t2 = t_lo + g / C_STEPS, st = g - (g / C_STEPS) * C_STEPS;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = wave + 8 * i, k16 = piece >> 3, mt8 = piece & 7;
            int t128 = 2 * t2 + (mt8 >> 2);
            t128 = t128 < mtiles ? t128 : mtiles - 1;                 // odd tile count: the missing half re-reads the last tile (its rows are >= N: ignored)
            ar[i] = *reinterpret_cast<const u32x4*>(img + (((long)t128 * STEPS + 4 * st + k16) * 4 + (mt8 & 3)) * 64 + lane);
            xr[i] = *reinterpret_cast<const u32x4*>(xsrc + (long)(4 * st + i) * 512);
        }
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(As + buf * C_A_U4 + tid + 512 * i) = ar[i];
            *reinterpret_cast<u32x4*>(Xs + buf * C_X_U4 + tid + 512 * i) = xr[i];
        }
    };

    Top4V top[2];
    top[0].init();
    top[1].init();
    float th[2] = {INFINITY, INFINITY};
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            // theta = (fourth largest coarse similarity of pass A's sample: its splits' top-4 lists, c4v) - 2 eps
            const int n = n0 + wn * 64 + j * 32 + l31;
            if (n < ncols) {
                Top4V t4;
                t4.init();
                for (int sp = 0; sp < ns_sample; ++sp) {
                    const float4 v4 = *reinterpret_cast<const float4*>(c4v + ((long)sp * ncols + n) * 4);
                    t4.insert(v4.x);
                    t4.insert(v4.y);
                    t4.insert(v4.z);
                    t4.insert(v4.w);
                }
                th[j] = t4.v[3] - 2.f * C_EPS;
            }
        }
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#ifdef S_TRACE
    struct { unsigned long long* tr = nullptr; int trn = 0; } trs;
    __shared__ unsigned long long tr_lds[256];
    if (MODE == 1 && blockIdx.x == S_TRACE_WG && threadIdx.x == S_TRACE_TID) trs.tr = tr_lds;
#endif
    if (G > 0) {
        gload(0);
        lstore(0);
        gload(G > 1 ? 1 : 0);
        slab_barrier();
    }
    for (int g = 0; g < G; ++g) {
        const int cur = g & 1;
        TR_STAMP(trs, 0);
        // no branch around the staging: past the end it re-stages the last step into the idle buffer (harmless), and the
        // compiler is free to thread the stores and loads between this step's MFMAs
        lstore(cur ^ 1);
        gload(g + 2 < G ? g + 2 : G - 1);
        TR_STAMP(trs, 1);
        const uint4* as = As + cur * C_A_U4 + wm * (4 * 64) + lane;
        const uint4* xs = Xs + cur * C_X_U4 + lh * C_QT + wn * 64 + l31;
#pragma unroll
        for (int k16 = 0; k16 < 4; ++k16) {
            f16x8 af[4], bf[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = __builtin_bit_cast(f16x8, as[(k16 * 8 + i) * 64]);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = __builtin_bit_cast(f16x8, xs[k16 * 2 * C_QT + j * 32]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        TR_STAMP(trs, 2);
        const int st = g - (g / C_STEPS) * C_STEPS;
        if (st == C_STEPS - 1) {
            const int t2 = t_lo + g / C_STEPS;
            const int m0 = t2 * C_MT;
            // fp16 storage: c = acc * inv[row].  max(acc, 0) * (largest inv of the tile) bounds c from above, so the
            // per-row inverse norm is only fetched for the few values that survive that bound.
            float imx = 1.f;
            if (F16) {
                const float* im = blob_invmax(blob, Npad);
                const int ta = 2 * t2 < mtiles ? 2 * t2 : mtiles - 1, tb = 2 * t2 + 1 < mtiles ? 2 * t2 + 1 : mtiles - 1;
                imx = fmaxf(im[ta], im[tb]);
            }
            // Per 32 x 32 accumulator tile: the largest of a lane's 16 values decides, for the whole wave at once (ballot), whether
            // any of them can matter; only then are the 16 walked one by one.  (The element-wise walk with its per-element
            // branches cost 11 000 cycles per 256 x 256 tile - 18 % of the pass - although a lane sees a hit every few dozen tiles.)
            const bool whole = m0 + C_MT <= N;                      // uniform: no row of this tile lies beyond the index
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float mx = acc[i][j][0];
                    bool anynan = false;
#pragma unroll
                    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[i][j][r]);
#pragma unroll
                    for (int r = 0; r < 16; ++r) anynan |= acc[i][j][r] != acc[i][j][r];      // NaN orders as the maximum (nan_max, torch.topk): never skipped, in either pass
                    bool maybe;
                    if (MODE == 0) maybe = anynan || !((F16 ? fmaxf(mx, 0.f) * imx : mx) <= top[j].v[3]);
                    else maybe = anynan || (F16 ? fmaxf(mx, 0.f) * imx : mx) >= th[j];
                    if (__builtin_amdgcn_ballot_w64(maybe) != 0ull) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = m0 + (wm * 4 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                            if (whole || row < N) {
                                const float a = acc[i][j][r];
                                if (MODE == 0) {
                                    if (!F16) {
                                        top[j].insert(nan_max(a));
                                    } else if (!(fmaxf(a, 0.f) * imx <= top[j].v[3])) {      // could enter the list (NaN passes)
                                        top[j].insert(nan_max(a * inv[row]));
                                    }
                                } else {
                                    bool hit;                                                  // a NaN similarity is a candidate (the exact kernel ranks it first)
                                    float cv = a;                                              // the coarse similarity the list keeps beside the row (rescore's second threshold)
                                    if (!F16) hit = !(a < th[j]);
                                    else {
                                        hit = a != a || (fmaxf(a, 0.f) * imx >= th[j]);
                                        if (hit) {
                                            cv = a * inv[row];
                                            hit = a != a || cv >= th[j];
                                        }
                                    }
                                    hit = hit && th[j] != INFINITY;                            // a padding column of the 256-query tile has no list (its threshold is +inf)
                                    if (hit) {
                                        const int n = n0 + wn * 64 + j * 32 + l31;
                                        const int pos = atomicAdd(&cnt[n], 1);
                                        if (pos < C_CAP) {
                                            cand[(long)n * C_CAP + pos] = row;
                                            candv[(long)n * C_CAP + pos] = cv;
                                        } else *overflow = 1;
                                    }
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                }
        }
        TR_STAMP(trs, 3);
        slab_barrier();
        TR_STAMP(trs, 4);
    }
#ifdef S_TRACE
    if (trs.tr) {
        const unsigned slot = atomicAdd(&g_trace_slot, 1u) & 63u;
        unsigned long long* gt = g_trace + slot * 256;
        gt[0] = 0x5452414345000000ull | (9ull << 20) | ((unsigned long long)F16 << 8);
        gt[1] = ((unsigned long long)G << 32) | (unsigned)trs.trn;
        gt[2] = ((unsigned long long)gridDim.x << 32) | (unsigned)nsplit;
        gt[3] = 0;
        for (int i = 0; i < trs.trn; ++i) gt[4 + i] = trs.tr[i];
    }
#endif
    if (MODE == 1) return;

    // merge the 4 partial lists (wm x lh) of every query through LDS, write this split's four largest coarse values
    float (*mv)[16] = reinterpret_cast<float (*)[16]>(csmem);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = wn * 64 + j * 32 + l31;
        const int slot = (wm * 2 + lh) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) mv[q][slot + e] = top[j].v[e];
    }
    __syncthreads();
    if (tid < C_QT) {
        Top4V t4;
        t4.init();
        for (int e = 0; e < 16; ++e) t4.insert(mv[tid][e]);
        const int n = n0 + tid;
        if (n < ncols) {
            const long o = ((long)split * ncols + n) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) c4v[o + e] = t4.v[e];
        }
    }
}

// The blob's kind lives in device memory (its header): one launch, the workgroup branches to the storage's instantiation of the body
// (both fit the 256 registers a wave has at one 128 KiB workgroup per CU; two launches cost an empty 4.6 us one per pass).
template <int MODE>
static __global__ __launch_bounds__(512) void knn_coarse_kernel(const float* __restrict__ blob, long Npad, int N, const uint4* __restrict__ qh,
                                                                int ncols, int nsplit, int tiles_per_split, int t2_cover,
                                                                float* __restrict__ c4v, int ns_sample,
                                                                int* __restrict__ cnt, int* __restrict__ cand, float* __restrict__ candv,
                                                                int* __restrict__ overflow) {
    if (reinterpret_cast<const int*>(blob)[1] == KIND_F16)
        knn_coarse_body<true, MODE>(blob, Npad, N, qh, ncols, nsplit, tiles_per_split, t2_cover, c4v, ns_sample, cnt, cand, candv, overflow);
    else
        knn_coarse_body<false, MODE>(blob, Npad, N, qh, ncols, nsplit, tiles_per_split, t2_cover, c4v, ns_sample, cnt, cand, candv, overflow);
}

// One wavefront per query: exact similarity of every candidate = (fp32 FMA chain of q_hat against the raw vector, lanes
// along k, fixed-order shuffle reduction) * (1 / norm); top-4 in the library's order -> rv / ri [ncols][4].
// A query with fewer than four candidates had non-finite coarse similarities (NaN / Inf samples upstream): rows 0..3 with
// similarity +inf, what the exact kernel's NaN-as-maximum rule selects.
static __global__ __launch_bounds__(256) void knn_rescore_kernel(const float* __restrict__ blob, long Npad, int N, const float* __restrict__ qn,
                                                                 int ncols, int T, const int* __restrict__ cnt, const int* __restrict__ cand,
                                                                 const float* __restrict__ candv, float* __restrict__ rv, int* __restrict__ ri) {
    constexpr int RG = 4;                           // candidates scored together
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= ncols) return;
    const int kind = reinterpret_cast<const int*>(blob)[1];
    const float* inv = blob_inv(blob, kind, N, Npad);
    const int b = n / T, t = n - b * T;
    const float* qp = qn + (long)b * KD * T + t;
    // the list's rows and coarse values are requested together with its length (all C_CAP entries exist; those past the end are masked
    // below): one round trip instead of three in a kernel that is a chain of them (WAIT 0.64, ACTIVE 0.09)
    int rowl[C_CAP / 64];
    float cvl[C_CAP / 64];
#pragma unroll
    for (int c = 0; c < C_CAP / 64; ++c) {
        rowl[c] = cand[(long)n * C_CAP + 64 * c + lane];
        cvl[c] = candv[(long)n * C_CAP + 64 * c + lane];
    }
    const int nc = min(cnt[n], C_CAP);
    Top4 t4;
    t4.init();
    // Second threshold.  Pass B's was cut from a SAMPLE of the index, so a list holds the few dozen rows above the sample's fourth best.
    // The list itself knows better: it contains every row with coarse >= theta, hence the four largest coarse values of the WHOLE index;
    // with c4 the fourth of them, four rows have exact >= c4 - eps, so a row of the exact top four has exact >= c4 - eps and coarse
    // >= c4 - 2 eps.  Only those are scored (a handful): the 3 KB gather per candidate was the kernel's whole cost.  NaN coarse values are
    // left out of the ranking (a lower c4: more rows kept) and always scored.
    float th2;
    {
        float rk[C_CAP / 64];
#pragma unroll
        for (int c = 0; c < C_CAP / 64; ++c) {
            if (!(64 * c + lane < nc)) {
                cvl[c] = -INFINITY;
                rowl[c] = 0;
            }
            rk[c] = cvl[c] == cvl[c] ? cvl[c] : -INFINITY;
        }
        float wm = -INFINITY;
#pragma unroll
        for (int round = 0; round < 4; ++round) {
            float lm = rk[0];
#pragma unroll
            for (int c = 1; c < C_CAP / 64; ++c) lm = fmaxf(lm, rk[c]);
            wm = lm;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) wm = fmaxf(wm, __shfl_xor(wm, o));
            const unsigned long long who = __builtin_amdgcn_ballot_w64(lm == wm);
            if (lane == __builtin_ctzll(who)) {       // one instance leaves the ranking (equal values count once each)
                bool done = false;
#pragma unroll
                for (int c = 0; c < C_CAP / 64; ++c)
                    if (!done && rk[c] == wm) {
                        rk[c] = -INFINITY;
                        done = true;
                    }
            }
        }
        th2 = wm - 2.f * C_EPS;                       // (fewer than four finite values: -inf, everything is scored)
    }
    static_assert(C_CAP == 256, "four 64-entry chunks");
    if (kind == KIND_F16) {
        // the fp16 image keeps a vector as 96 16-byte segments (8 consecutive channels each, 4 KiB apart): lane l reads
        // segments l and 64 + l (l < 32) with one 16-byte load each instead of twelve 2-byte loads
        const uint4* img = blob_img16(blob, kind, N, Npad);
        float qs[2][8];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int seg = lane + 64 * h;
                qs[h][j] = seg < 96 ? qp[(long)(8 * seg + j) * T] : 0.f;
            }
        // The candidate list is read 64 entries at a time (one coalesced load, the row numbers then come out of a register by lane
        // broadcast) and the candidates are scored four at a time: their loads are in flight together instead of one list entry -> one
        // vector -> one reduction after the other (the chain was 2 - 3 us per candidate with nothing else to issue: ACTIVE 0.06).
        for (int base = 0, ch = 0; base < nc; base += 64, ++ch) {
            const int mine = ch == 0 ? rowl[0] : (ch == 1 ? rowl[1] : (ch == 2 ? rowl[2] : rowl[3]));
            const float cvv = ch == 0 ? cvl[0] : (ch == 1 ? cvl[1] : (ch == 2 ? cvl[2] : cvl[3]));
            unsigned long long mask = __builtin_amdgcn_ballot_w64(base + lane < nc && !(cvv < th2));      // the entries that pass (NaN does)
            while (mask) {
                int row[RG];
                bool live[RG];
                float d[RG], iv[RG];
                u32x4 w[RG][2];
                int pos = 0;
#pragma unroll
                for (int g = 0; g < RG; ++g) {
                    live[g] = mask != 0ull;
                    if (live[g]) {
                        pos = __builtin_ctzll(mask);
                        mask &= mask - 1;
                    }
                    row[g] = __shfl(mine, pos);      // (past the end: the last one again, not inserted)
                    const long rbase = ((long)(row[g] >> 7) * STEPS * 4 + ((row[g] & 127) >> 5)) * 64 + (row[g] & 31);     // + (step * 4) * 64 + half * 32
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int seg = lane + 64 * h < 96 ? lane + 64 * h : 0;
                        w[g][h] = *reinterpret_cast<const u32x4*>(img + rbase + (long)(seg >> 1) * 256 + (seg & 1) * 32);
                    }
                    iv[g] = inv[row[g]];
                }
#pragma unroll
                for (int g = 0; g < RG; ++g) {
                    d[g] = 0.f;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        if (lane + 64 * h < 96) {
                            const f16x8 hv = __builtin_bit_cast(f16x8, w[g][h]);
#pragma unroll
                            for (int j = 0; j < 8; ++j) d[g] = fmaf(qs[h][j], (float)hv[j], d[g]);
                        }
                    }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1)
#pragma unroll
                    for (int g = 0; g < RG; ++g) d[g] += __shfl_xor(d[g], o);
#pragma unroll
                for (int g = 0; g < RG; ++g)
                    if (live[g]) t4.insert(nan_max(d[g] * iv[g]), row[g]);
            }
        }
    } else {
        float qv[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) qv[u] = qp[(long)(lane + 64 * u) * T];
        const float* rows = blob + HDR;
        for (int base = 0, ch = 0; base < nc; base += 64, ++ch) {
            const int mine = ch == 0 ? rowl[0] : (ch == 1 ? rowl[1] : (ch == 2 ? rowl[2] : rowl[3]));
            const float cvv = ch == 0 ? cvl[0] : (ch == 1 ? cvl[1] : (ch == 2 ? cvl[2] : cvl[3]));
            unsigned long long mask = __builtin_amdgcn_ballot_w64(base + lane < nc && !(cvv < th2));      // the entries that pass (NaN does)
            while (mask) {
                int row[RG];
                bool live[RG];
                float d[RG], iv[RG], xv[RG][12];
                int pos = 0;
#pragma unroll
                for (int g = 0; g < RG; ++g) {
                    live[g] = mask != 0ull;
                    if (live[g]) {
                        pos = __builtin_ctzll(mask);
                        mask &= mask - 1;
                    }
                    row[g] = __shfl(mine, pos);      // (past the end: the last one again, not inserted)
                    const float* rp = rows + (long)row[g] * KD + lane;
#pragma unroll
                    for (int u = 0; u < 12; ++u) xv[g][u] = rp[64 * u];
                    iv[g] = inv[row[g]];
                }
#pragma unroll
                for (int g = 0; g < RG; ++g) {
                    d[g] = 0.f;
#pragma unroll
                    for (int u = 0; u < 12; ++u) d[g] = fmaf(qv[u], xv[g][u], d[g]);
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1)
#pragma unroll
                    for (int g = 0; g < RG; ++g) d[g] += __shfl_xor(d[g], o);
#pragma unroll
                for (int g = 0; g < RG; ++g)
                    if (live[g]) t4.insert(nan_max(d[g] * iv[g]), row[g]);
            }
        }
    }
    if (nc < 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            t4.v[e] = INFINITY;
            t4.i[e] = e;
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            rv[(long)n * 4 + e] = t4.v[e];
            ri[(long)n * 4 + e] = t4.i[e];
        }
    }
}

struct KnnPlan {
    int ncols, qtiles, nsplit, tps;
    long Npad;
};
static KnnPlan knn_plan(int B, int T, int64_t N) {
    KnnPlan p;
    p.ncols = B * T;
    p.Npad = npad128(N);
    p.qtiles = (p.ncols + KNN_QT - 1) / KNN_QT;
    const int mtiles = (int)(p.Npad / 128);
    int nsplit = (KNN_BLOCKS + p.qtiles - 1) / p.qtiles;
    if (nsplit > mtiles) nsplit = mtiles;
    if (nsplit < 1) nsplit = 1;
    p.tps = (mtiles + nsplit - 1) / nsplit;
    p.nsplit = (mtiles + p.tps - 1) / p.tps;
    return p;
}

struct KnnLists {        // where the merge kernels find the per-query top-4 lists
    float* cv = nullptr;    // exact kernel: [nsplit][ncols][4]
    int* ci = nullptr;
    float* rv = nullptr;    // two-stage search: [ncols][4]
    int* ri = nullptr;
    int* flag = nullptr;    // 0 = the two-stage lists are valid; nullptr = exact kernel only
};

template <int MODE>
static int coarse_launch(tvc_ctx* ctx, hipStream_t s, const float* prepared, long Npad, int N, const uint4* qh, int ncols, int qtiles, int t2_cover,
                         float* c4v, int ns_sample, int* cnt, int* cand, float* candv, int* flag, int* nsplit_out) {
    static bool ready_dev[64] = {};
    bool& ready = ready_dev[ctx->device & 63];
    if (!ready) {
        hipError_t e = hipFuncSetAttribute((const void*)knn_coarse_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, C_LDS);
        if (e != hipSuccess) return fail(ctx, TVC_ERR_HIP, "knn coarse setup: %s", hipGetErrorString(e));
        ready = true;
    }
    int nsplit = (768 + qtiles - 1) / qtiles;           // ~3 workgroups per CU's worth of work units (one resident workgroup per CU: 128 KiB of LDS)
    if (nsplit > t2_cover) nsplit = t2_cover;
    if (nsplit < 1) nsplit = 1;
    const int tps = (t2_cover + nsplit - 1) / nsplit;
    nsplit = (t2_cover + tps - 1) / tps;
    if (nsplit_out) *nsplit_out = nsplit;
    hipLaunchKernelGGL((knn_coarse_kernel<MODE>), dim3((unsigned)(qtiles * nsplit)), dim3(512), C_LDS, s, prepared, Npad, N, qh, ncols, nsplit, tps, t2_cover,
                       c4v, ns_sample, cnt, cand, candv, flag);
    return 0;
}

// workspace of the two-stage search for `qtiles` 256-query tiles (also the worst case over index sizes)
static int coarse_max_nsplit(int qtiles) { return (768 + qtiles - 1) / qtiles; }

// query normalisation + the per-query top-4 lists; shared by the whole-index match and the index-sharded variant.
// (The blob's kind lives in device memory - its header -, so the host cannot pick an instantiation: the kernels branch on it.)
static int knn_candidates(tvc_ctx* ctx, hipStream_t s, Ws& ws, bool dry, const float* src, const float* prepared, int64_t N, int B, int T,
                          const KnnPlan& p, KnnLists* L) {
    float* qn = ws.get<float>((size_t)B * KD * T);
    L->cv = ws.get<float>((size_t)p.nsplit * p.ncols * 4);
    L->ci = ws.get<int>((size_t)p.nsplit * p.ncols * 4);
    const bool two_stage = N >= KNN_COARSE_MIN;
    uint4* qh = nullptr;
    float* c4v = nullptr;
    int *cnt = nullptr, *cand = nullptr;
    float* candv = nullptr;
    const int cq = (p.ncols + C_QT - 1) / C_QT;
    if (two_stage) {
        qh = ws.get<uint4>((size_t)cq * STEPS * 2 * 256);
        c4v = ws.get<float>((size_t)coarse_max_nsplit(cq) * p.ncols * 4);
        cnt = ws.get<int>((size_t)p.ncols);
        cand = ws.get<int>((size_t)p.ncols * C_CAP);
        candv = ws.get<float>((size_t)p.ncols * C_CAP);
        L->rv = ws.get<float>((size_t)p.ncols * 4);
        L->ri = ws.get<int>((size_t)p.ncols * 4);
        L->flag = ws.get<int>(64);
    }
    if (dry) return 0;
    if (N > 0x7fffff00L) return fail(ctx, TVC_ERR_ARG, "index too large");
    const int nblk = two_stage ? cq * 4 : (p.ncols + 63) / 64;      // the fp16 query image is written for whole 256-query tiles
    hipLaunchKernelGGL(query_normalize_kernel, dim3(nblk), dim3(QN_WAVES * 64), 0, s, src, qn, B, T, qh, cnt, L->flag);
    if (two_stage) {
        ProfScope ps(ctx, s, dry, "knn.coarse+rescore");
        const int t2 = (int)((p.Npad + C_MT - 1) / C_MT);          // 256-vector tiles
        // pass A: an eighth of the index, at least 1280 vectors.  (A smaller sample lowers the threshold: longer lists - whose rows the
        // rescoring wave's own threshold then drops unscored - and, past C_CAP entries, the exact fallback: a sixteenth sends the
        // 100 000-vector index there.  Five tiles x 50 query tiles of the bench batch are one round of the chip.)
        int sample = t2 / 8;
        if (sample < 5) sample = 5;
        if (sample > t2) sample = t2;
        int nsA = 1;
        TVC_CHECK((coarse_launch<0>(ctx, s, prepared, p.Npad, (int)N, qh, p.ncols, cq, sample, c4v, 0, cnt, cand, candv, L->flag, &nsA)));
        TVC_CHECK((coarse_launch<1>(ctx, s, prepared, p.Npad, (int)N, qh, p.ncols, cq, t2, c4v, nsA, cnt, cand, candv, L->flag, nullptr)));
        hipLaunchKernelGGL(knn_rescore_kernel, dim3((p.ncols + 3) / 4), dim3(256), 0, s, prepared, p.Npad, (int)N, qn, p.ncols, T, cnt, cand, candv, L->rv, L->ri);
    }
    ProfScope ps(ctx, s, dry, "knn.exact");       // ~0 when the two-stage search succeeded (the kernel exits on the flag)
    hipLaunchKernelGGL(knn_topk_split_kernel, dim3((unsigned)(p.qtiles * p.nsplit)), dim3(512), 0, s, prepared, p.Npad, (int)N, qn, p.ncols, T,
                       p.nsplit, p.tps, L->cv, L->ci, (const int*)L->flag);
    return 0;
}

int run_knn_topk(tvc_ctx* ctx, hipStream_t s, Ws& ws, bool dry, const float* src, const float* prepared, int64_t N,
                 float* sims_out, int64_t* idx_out, int B, int T) {
    const KnnPlan p = knn_plan(B, T, N);
    KnnLists L;
    TVC_CHECK(knn_candidates(ctx, s, ws, dry, src, prepared, N, B, T, p, &L));
    if (dry) return 0;
    hipLaunchKernelGGL(knn_merge_kernel, dim3((p.ncols + 255) / 256), dim3(256), 0, s, L.cv, L.ci, p.nsplit, p.ncols, sims_out, idx_out, L.rv, L.ri, L.flag);
    return launch_check(ctx, "knn_topk");
}

int run_knn_slots(tvc_ctx* ctx, hipStream_t s, const float* prepared, int64_t N, const int64_t* idx, float* slots, int64_t nslots) {
    hipLaunchKernelGGL(knn_slot_gather_kernel, dim3((unsigned)nslots), dim3(192), 0, s, prepared, idx, (long)nslots, (long)N, (long)npad128(N), slots);
    return launch_check(ctx, "knn_slots");
}

int run_knn_finish(tvc_ctx* ctx, hipStream_t s, const float* slots, float* out, int B, int T) {
    const int ncols = B * T;
    hipLaunchKernelGGL(knn_finish_kernel, dim3((ncols + 31) / 32), dim3(256), 0, s, slots, ncols, T, out);
    return launch_check(ctx, "knn_finish");
}

int run_knn(tvc_ctx* ctx, hipStream_t s, Ws& ws, bool dry, const float* src, const float* prepared, int64_t N,
            float* out, int64_t* idx_out, int B, int T) {
    const KnnPlan p = knn_plan(B, T, N);
    KnnLists L;
    TVC_CHECK(knn_candidates(ctx, s, ws, dry, src, prepared, N, B, T, p, &L));
    if (dry) return 0;
    hipLaunchKernelGGL(knn_merge_gather_kernel, dim3((p.ncols + 31) / 32), dim3(256), 0, s, L.cv, L.ci, p.nsplit, p.ncols, T, (int)N, p.Npad,
                       prepared, out, idx_out, L.rv, L.ri, L.flag);
    return launch_check(ctx, "knn_match");
}

}  // namespace tvc

#ifdef S_TRACE
extern "C" int tvc_debug_trace_knn(unsigned long long* host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(tvc::g_trace), sizeof(tvc::g_trace)) == hipSuccess ? 0 : -1;
}
#endif
