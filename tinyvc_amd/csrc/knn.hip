// kNN feature match (feature_retrieval.py:15-33, metrics='cos', k=4, alpha=0).
//
// The reference materialises sims[T][N] and calls torch.topk; here the index streams through the
// fp32 matrix pipe in 128-vector tiles and every lane keeps a running top-4 for the query column it
// owns in the MFMA accumulator layout (index vectors are the M axis, queries the N axis, so the 16
// accumulator registers of a lane are 16 index vectors against ONE query: the reduction is
// lane-local).  The index is split across workgroups for occupancy; a second kernel merges the
// per-split candidates, emits int64 indices and gathers + averages the 4 raw index vectors.
//
// Tie-break: equal similarities -> lower index first (torch.topk leaves it unspecified).
#include "conv3s.h"
#include "igemm.h"
#include "small_kernels.h"
#include "tvc_common.h"

namespace tvc {

constexpr int KD = kSslDim;  // 768
#ifndef KNN_BK
#define KNN_BK 16      // K-slab depth of the similarity GEMM (deeper slabs cost occupancy: measured slower)
#endif
#ifndef KNN_SPLIT
#define KNN_SPLIT 1    // similarity GEMM on the split-precision bf16 path (0: exact-fp32 MFMA kernel)
#endif
#ifndef KNN_BLOCKS
#define KNN_BLOCKS 1024   // target workgroup count (query tiles x index splits)
#endif
#ifndef KNN_PIN
#define KNN_PIN 0
#endif
#ifndef KNN_WAVES
#define KNN_WAVES 8    // waves per workgroup: 8 -> each wave owns 64 x 32 (32 accumulator registers)
#endif

static inline int64_t npad128(int64_t N) { return (N + 127) / 128 * 128; }

// prepared index blob: [768][Npad] cosine-normalised columns, then [N][768] raw rows
static __global__ void index_prepare_kernel(const float* __restrict__ index, float* __restrict__ normT,
                                            float* __restrict__ rows, long N, long Npad) {
    long n = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (n >= Npad) return;
    if (n >= N) {
        for (int k = 0; k < KD; ++k) normT[(long)k * Npad + n] = 0.f;
        return;
    }
    float s = 0.f;
    for (int k = 0; k < KD; ++k) {
        float v = index[(long)k * N + n];
        s = fmaf(v, v, s);
    }
    float den = sqrtf(s) + 1e-6f;
    for (int k = 0; k < KD; ++k) {
        float v = index[(long)k * N + n];
        normT[(long)k * Npad + n] = v / den;
        rows[n * KD + k] = v;
    }
}

// Split-precision image of the normalised index for knn_topk_split_kernel: every value as three bf16 parts
// (v = p1 + p2 + p3, residuals exact), laid out [128-vector tile][K16 step][m-tile][part][lane][8] so that a
// (tile, step) is 12 contiguous 1 KiB pieces already in MFMA lane order (row = lane & 31, k = 8 (lane >> 5) + j).
static __global__ void index_split_kernel(const float* __restrict__ normT, unsigned short* __restrict__ img, long Npad) {
    long n = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (n >= Npad) return;
    const long tile = n >> 7;
    const int mt = (int)(n & 127) >> 5, l31 = (int)(n & 31);
    for (int k = 0; k < KD; ++k) {
        float v = normT[(long)k * Npad + n];
        __bf16 h1 = (__bf16)v;
        float r = v - (float)h1;
        __bf16 h2 = (__bf16)r;
        float r2 = r - (float)h2;
        __bf16 h3 = (__bf16)r2;
        const int step = k >> 4, lh = (k >> 3) & 1, j = k & 7;
        long base = ((((tile * (KD / 16) + step) * 4 + mt) * 3) * 64 + (lh * 32 + l31)) * 8 + j;
        img[base] = __builtin_bit_cast(unsigned short, h1);
        img[base + 512] = __builtin_bit_cast(unsigned short, h2);
        img[base + 1024] = __builtin_bit_cast(unsigned short, h3);
    }
}

// prepared blob: [768][Npad] normalised columns | [N][768] raw rows | split image (Npad * 768 * 3 bf16)
int run_prepare_index(tvc_ctx* ctx, hipStream_t s, const float* index, float* prepared, int64_t N) {
    long Npad = npad128(N);
    hipLaunchKernelGGL(index_prepare_kernel, dim3((unsigned)((Npad + 255) / 256)), dim3(256), 0, s, index, prepared,
                       prepared + (size_t)KD * Npad, (long)N, Npad);
    unsigned short* img = reinterpret_cast<unsigned short*>(prepared + (size_t)KD * Npad + (size_t)N * KD);
    hipLaunchKernelGGL(index_split_kernel, dim3((unsigned)((Npad + 255) / 256)), dim3(256), 0, s, prepared, img, Npad);
    return launch_check(ctx, "knn_prepare_index");
}

// qn[b][k][t] = src[b][k][t] / (||src[b][:][t]|| + 1e-6).  One workgroup = 64 consecutive columns;
// its 4 waves each sum a quarter of the 768 channels (lanes along time, coalesced), partial sums of
// squares meet in LDS in a fixed order, then every wave rescales its quarter.
static __global__ __launch_bounds__(256) void query_normalize_kernel(const float* __restrict__ src, float* __restrict__ qn, int B, int T) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long ncols = (long)B * T;
    const long n = blockIdx.x * 64L + lane;
    const bool ok = n < ncols;
    const long nn = ok ? n : ncols - 1;
    const int b = (int)(nn / T), t = (int)(nn - (long)b * T);
    const float* p = src + (long)b * KD * T + t;
    float* q = qn + (long)b * KD * T + t;
    const int k0 = wave * (KD / 4), k1 = k0 + KD / 4;
    float s = 0.f;
    for (int k = k0; k < k1; ++k) s = fmaf(p[(long)k * T], p[(long)k * T], s);
    part[wave][lane] = s;
    __syncthreads();
    const float den = sqrtf(((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane]) + 1e-6f;
    if (!ok) return;
    for (int k = k0; k < k1; ++k) q[(long)k * T] = p[(long)k * T] / den;
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

// grid = qtiles * nsplit ; workgroup = 128 queries x (tiles_per_split index tiles of 128)
static __global__ __launch_bounds__(KNN_WAVES * 64) void knn_topk_kernel(const float* __restrict__ normT, long Npad, int N,
                                                              const float* __restrict__ qn, int ncols, int T,
                                                              int nsplit, int tiles_per_split,
                                                              float* __restrict__ cand_v, int* __restrict__ cand_i) {
    constexpr int BM = 128, BN = 128, BK = KNN_BK, TM = 2, TN = KNN_WAVES == 8 ? 1 : 2;
    constexpr int NTHR = KNN_WAVES * 64, BRS = NTHR / 128;   // B staging: thread owns column tid % 128, rows tid / 128 + BRS * j
    __shared__ __attribute__((aligned(16))) float smem[BK * BM + BK * BN];
    float* As = smem;
    float* Bs = smem + BK * BM;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = KNN_WAVES == 8 ? wave >> 2 : wave >> 1, wn = KNN_WAVES == 8 ? wave & 3 : wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int split = blockIdx.x % nsplit;
    const int qtile = blockIdx.x / nsplit;
    const int n0 = qtile * BN;
    const int mtiles = (int)(Npad / BM);
    const int mt_lo = split * tiles_per_split;
    const int mt_hi = min(mtiles, mt_lo + tiles_per_split);

    LoadPlain ld{qn, KD, T, (long)KD * T};
    const LoadPlain::Ctx col = ld.ctx(n0 + (tid & 127), ncols, T);
    const int brow0 = tid >> 7;

    Top4 top[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) top[j].init();

    for (int mt = mt_lo; mt < mt_hi; ++mt) {
        const int m0 = mt * BM;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        constexpr int AP = BK * BM / 4 / NTHR, BP = BK * BN / NTHR;   // per-thread float4 / float staging counts
        float4 areg[AP];
        float breg[BP];
        auto load_slab = [&](int k0) {
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                int idx = tid + i * NTHR;
                int kk = idx >> 5, c4 = idx & 31;
                areg[i] = *reinterpret_cast<const float4*>(normT + (long)(k0 + kk) * Npad + m0 + c4 * 4);
            }
#pragma unroll
            for (int j = 0; j < BP; ++j) breg[j] = ld.get(col, k0 + brow0 + BRS * j);
        };
        load_slab(0);
        for (int kt = 0; kt < KD / BK; ++kt) {
#pragma unroll
            for (int i = 0; i < AP; ++i) *reinterpret_cast<float4*>(As + (tid + i * NTHR) * 4) = areg[i];
#pragma unroll
            for (int j = 0; j < BP; ++j) Bs[(brow0 + BRS * j) * BN + (tid & 127)] = breg[j];
            __syncthreads();
            if (kt + 1 < KD / BK) load_slab((kt + 1) * BK);
#pragma unroll
            for (int ks = 0; ks < BK / 2; ++ks) {
                const int k = 2 * ks + lh;
                float a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = As[k * BM + (wm * TM + i) * 32 + l31];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = Bs[k * BN + (wn * TN + j) * 32 + l31];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
        }
        // running top-4: this lane's 16 registers are 16 index vectors against its own query
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row < N) top[j].insert(nan_max(acc[i][j][r]), row);
                }
    }

    // merge the 4 partial lists (wm in {0,1} x lh in {0,1}) of every query through LDS
    __shared__ float mv[128][16];
    __shared__ int mi[128][16];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int q = (wn * TN + j) * 32 + l31;
        int slot = (wm * 2 + lh) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mv[q][slot + e] = top[j].v[e];
            mi[q][slot + e] = top[j].i[e];
        }
    }
    __syncthreads();
    if (tid < 128) {
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

// Same search on the split-precision path (conv3s.h): sims from six bf16 part-products per K16 block, fp32
// accumulation, error below the fp32 MFMA chain's (so the top-4 decisions are at least as faithful).
// 8 waves of 64 (index) x 32 (queries); the (tile, step) sequence is one flat pipeline: registers hold step g+1,
// LDS is double-buffered, one raw barrier per step.
static __global__ __launch_bounds__(512) void knn_topk_split_kernel(const uint4* __restrict__ img, long Npad, int N,
                                                                    const float* __restrict__ qn, int ncols, int T,
                                                                    int nsplit, int tiles_per_split,
                                                                    float* __restrict__ cand_v, int* __restrict__ cand_i) {
    constexpr int STEPS = KD / 16;                       // 48
    constexpr int A_U4 = 12 * 64, X_U4 = 3 * 2 * 128;    // one buffer each: 12 KiB + 12 KiB
    __shared__ __attribute__((aligned(16))) uint4 smem[2 * (A_U4 + X_U4)];
    __shared__ float mv[128][16];
    __shared__ int mi[128][16];
    uint4* As = smem;
    uint4* Xs = smem + 2 * A_U4;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, lh = lane >> 5;
    const int split = blockIdx.x % nsplit;
    const int qtile = blockIdx.x / nsplit;
    const int n0 = qtile * 128;
    const int mtiles = (int)(Npad >> 7);
    const int mt_lo = split * tiles_per_split;
    const int mt_hi = min(mtiles, mt_lo + tiles_per_split);
    const int G = (mt_hi - mt_lo) * STEPS;               // flat (tile, step) sequence

    // staging roles: threads 0..255 own one query item (8 channels of one column) + weight piece tid / 64;
    // threads 256..511 own weight pieces 4.. (two each)
    const bool xrole = tid < 256;
    const float* qp = qn;
    int xdst = 0;
    if (xrole) {
        const int g = tid >> 7, pos = tid & 127;
        int n = n0 + pos;
        n = n < ncols ? n : ncols - 1;
        const int b = n / T, t = n - b * T;
        qp = qn + ((long)b * KD + 8 * g) * T + t;
        xdst = g * 128 + pos;
    }
    const int pa = xrole ? wave : 4 + 2 * (wave - 4);    // first weight piece of this thread
    float xr[8];
    u32x4 ar[2];
    auto gload = [&](int g) __attribute__((always_inline)) {
        const int mt = mt_lo + g / STEPS, st = g - (g / STEPS) * STEPS;
        const uint4* src = img + ((long)mt * STEPS + st) * (12 * 64) + lane;
        ar[0] = *reinterpret_cast<const u32x4*>(src + pa * 64);
        if (!xrole) ar[1] = *reinterpret_cast<const u32x4*>(src + (pa + 1) * 64);
        if (xrole) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xr[j] = qp[(long)(st * 16 + j) * T];
        }
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {
        uint4* ab = As + buf * A_U4;
        *reinterpret_cast<u32x4*>(ab + pa * 64 + lane) = ar[0];
        if (!xrole) *reinterpret_cast<u32x4*>(ab + (pa + 1) * 64 + lane) = ar[1];
        if (xrole) {
            uint4 p1, p2, p3;
            split8(xr, p1, p2, p3);
            uint4* xb = Xs + buf * X_U4;
            xb[xdst] = p1;
            xb[256 + xdst] = p2;
            xb[512 + xdst] = p3;
        }
    };

    Top4 top;
    top.init();
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

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
        const uint4* as = As + cur * A_U4 + wm * (6 * 64) + lane;
        const uint4* xs = Xs + cur * X_U4 + lh * 128 + wn * 32 + l31;
        bf16x8 af[2][3], bf[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) bf[p] = __builtin_bit_cast(bf16x8, xs[p * 256]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) af[i][p] = __builtin_bit_cast(bf16x8, as[(i * 3 + p) * 64]);
#if KNN_PIN
        __builtin_amdgcn_sched_barrier(0);          // all nine reads, then the twelve MFMAs
#endif
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA[q]], bf[PB[q]], acc[i], 0, 0, 0);
        const int st = g - (g / STEPS) * STEPS;
        if (st == STEPS - 1) {
            // running top-4: this lane's 16 registers are 16 index vectors against its own query
            const int m0 = (mt_lo + g / STEPS) * 128;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int row = m0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (row < N) top.insert(nan_max(acc[i][r]), row);
                    acc[i][r] = 0.f;
                }
        }
        slab_barrier();
    }

    // merge the 4 partial lists (wm in {0,1} x lh in {0,1}) of every query through LDS
    {
        int q = wn * 32 + l31;
        int slot = (wm * 2 + lh) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mv[q][slot + e] = top.v[e];
            mi[q][slot + e] = top.i[e];
        }
    }
    __syncthreads();
    if (tid < 128) {
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

// One workgroup = 32 consecutive query columns: merge split candidates -> top-4, write indices,
// gather the 4 raw rows per query (coalesced along the feature axis), average, and write
// out[b][k][t] through an LDS transpose so stores run along t.
static __global__ __launch_bounds__(256) void knn_merge_gather_kernel(const float* __restrict__ cand_v, const int* __restrict__ cand_i,
                                                                      int nsplit, int ncols, int T, int N,
                                                                      const float* __restrict__ rows,
                                                                      float* __restrict__ out, int64_t* __restrict__ idx_out) {
    __shared__ int sel[32][4];
    __shared__ float tile[32][193];
    const int tid = threadIdx.x;
    const int n0 = blockIdx.x * 32;
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
        for (int q = wave; q < 32; q += 4) {
            const float* r0 = rows + (long)sel[q][0] * KD + kc;
            const float* r1 = rows + (long)sel[q][1] * KD + kc;
            const float* r2 = rows + (long)sel[q][2] * KD + kc;
            const float* r3 = rows + (long)sel[q][3] * KD + kc;
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                int k = lane + 64 * u;
                float sum = __fadd_rn(__fadd_rn(__fadd_rn(r0[k], r1[k]), r2[k]), r3[k]);
                tile[q][k] = sum * 0.25f;
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
                                                               float* __restrict__ sims_out, int64_t* __restrict__ idx_out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= ncols) return;
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
static __global__ __launch_bounds__(192) void knn_slot_gather_kernel(const float* __restrict__ rows, const int64_t* __restrict__ idx, long nslots,
                                                                     long N, float* __restrict__ slots) {
    const long sl = blockIdx.x;
    if (sl >= nslots) return;
    const int64_t i = idx[sl];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i >= 0 && i < N) v = reinterpret_cast<const float4*>(rows + i * KD)[threadIdx.x];
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

int run_knn_topk(tvc_ctx* ctx, hipStream_t s, Ws& ws, bool dry, const float* src, const float* prepared, int64_t N,
                 float* sims_out, int64_t* idx_out, int B, int T) {
    const int ncols = B * T;
    const long Npad = npad128(N);
    const int qtiles = (ncols + 127) / 128;
    const int mtiles = (int)(Npad / 128);
    int nsplit = (KNN_BLOCKS + qtiles - 1) / qtiles;
    if (nsplit > mtiles) nsplit = mtiles;
    if (nsplit < 1) nsplit = 1;
    const int tps = (mtiles + nsplit - 1) / nsplit;
    nsplit = (mtiles + tps - 1) / tps;
    float* qn = ws.get<float>((size_t)B * KD * T);
    float* cv = ws.get<float>((size_t)nsplit * ncols * 4);
    int* ci = ws.get<int>((size_t)nsplit * ncols * 4);
    if (dry) return 0;
    if (N > 0x7fffff00L) return fail(ctx, TVC_ERR_ARG, "index too large");
    hipLaunchKernelGGL(query_normalize_kernel, dim3((ncols + 63) / 64), dim3(256), 0, s, src, qn, B, T);
    const uint4* img = reinterpret_cast<const uint4*>(prepared + (size_t)KD * Npad + (size_t)N * KD);
    hipLaunchKernelGGL(knn_topk_split_kernel, dim3((unsigned)(qtiles * nsplit)), dim3(512), 0, s, img, Npad, (int)N, qn, ncols, T, nsplit, tps, cv, ci);
    hipLaunchKernelGGL(knn_merge_kernel, dim3((ncols + 255) / 256), dim3(256), 0, s, cv, ci, nsplit, ncols, sims_out, idx_out);
    return launch_check(ctx, "knn_topk");
}

int run_knn_slots(tvc_ctx* ctx, hipStream_t s, const float* prepared, int64_t N, const int64_t* idx, float* slots, int64_t nslots) {
    const float* rows = prepared + (size_t)KD * npad128(N);
    hipLaunchKernelGGL(knn_slot_gather_kernel, dim3((unsigned)nslots), dim3(192), 0, s, rows, idx, (long)nslots, (long)N, slots);
    return launch_check(ctx, "knn_slots");
}

int run_knn_finish(tvc_ctx* ctx, hipStream_t s, const float* slots, float* out, int B, int T) {
    const int ncols = B * T;
    hipLaunchKernelGGL(knn_finish_kernel, dim3((ncols + 31) / 32), dim3(256), 0, s, slots, ncols, T, out);
    return launch_check(ctx, "knn_finish");
}

int run_knn(tvc_ctx* ctx, hipStream_t s, Ws& ws, bool dry, const float* src, const float* prepared, int64_t N,
            float* out, int64_t* idx_out, int B, int T) {
    const int ncols = B * T;
    const long Npad = npad128(N);
    const int qtiles = (ncols + 127) / 128;
    const int mtiles = (int)(Npad / 128);
    int nsplit = (KNN_BLOCKS + qtiles - 1) / qtiles;
    if (nsplit > mtiles) nsplit = mtiles;
    if (nsplit < 1) nsplit = 1;
    const int tps = (mtiles + nsplit - 1) / nsplit;
    nsplit = (mtiles + tps - 1) / tps;
    float* qn = ws.get<float>((size_t)B * KD * T);
    float* cv = ws.get<float>((size_t)nsplit * ncols * 4);
    int* ci = ws.get<int>((size_t)nsplit * ncols * 4);
    if (dry) return 0;
    if (N > 0x7fffff00L) return fail(ctx, TVC_ERR_ARG, "index too large");
    hipLaunchKernelGGL(query_normalize_kernel, dim3((ncols + 63) / 64), dim3(256), 0, s, src, qn, B, T);
    if (KNN_SPLIT) {
        const uint4* img = reinterpret_cast<const uint4*>(prepared + (size_t)KD * Npad + (size_t)N * KD);
        hipLaunchKernelGGL(knn_topk_split_kernel, dim3((unsigned)(qtiles * nsplit)), dim3(512), 0, s, img, Npad, (int)N, qn, ncols, T,
                           nsplit, tps, cv, ci);
    } else {
        hipLaunchKernelGGL(knn_topk_kernel, dim3((unsigned)(qtiles * nsplit)), dim3(KNN_WAVES * 64), 0, s, prepared, Npad, (int)N, qn,
                           ncols, T, nsplit, tps, cv, ci);
    }
    hipLaunchKernelGGL(knn_merge_gather_kernel, dim3((ncols + 31) / 32), dim3(256), 0, s, cv, ci, nsplit, ncols, T, (int)N,
                       prepared + (size_t)KD * Npad, out, idx_out);
    return launch_check(ctx, "knn_match");
}

}  // namespace tvc
