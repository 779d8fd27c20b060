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
#include "igemm.h"
#include "small_kernels.h"
#include "tvc_common.h"

namespace tvc {

constexpr int KD = kSslDim;  // 768
#ifndef KNN_BK
#define KNN_BK 16      // K-slab depth of the similarity GEMM (deeper slabs cost occupancy: measured slower)
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

int run_prepare_index(tvc_ctx* ctx, hipStream_t s, const float* index, float* prepared, int64_t N) {
    long Npad = npad128(N);
    hipLaunchKernelGGL(index_prepare_kernel, dim3((unsigned)((Npad + 255) / 256)), dim3(256), 0, s, index, prepared,
                       prepared + (size_t)KD * Npad, (long)N, Npad);
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
                    if (row < N) top[j].insert(acc[i][j][r], row);
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

// One workgroup = 32 consecutive query columns: merge split candidates -> top-4, write indices,
// gather the 4 raw rows per query (coalesced along the feature axis), average, and write
// out[b][k][t] through an LDS transpose so stores run along t.
static __global__ __launch_bounds__(256) void knn_merge_gather_kernel(const float* __restrict__ cand_v, const int* __restrict__ cand_i,
                                                                      int nsplit, int ncols, int T,
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

int run_knn(tvc_ctx* ctx, hipStream_t s, Ws& ws, bool dry, const float* src, const float* prepared, int64_t N,
            float* out, int64_t* idx_out, int B, int T) {
    const int ncols = B * T;
    const long Npad = npad128(N);
    const int qtiles = (ncols + 127) / 128;
    const int mtiles = (int)(Npad / 128);
    int nsplit = (1024 + qtiles - 1) / qtiles;
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
    hipLaunchKernelGGL(knn_topk_kernel, dim3((unsigned)(qtiles * nsplit)), dim3(KNN_WAVES * 64), 0, s, prepared, Npad, (int)N, qn,
                       ncols, T, nsplit, tps, cv, ci);
    hipLaunchKernelGGL(knn_merge_gather_kernel, dim3((ncols + 31) / 32), dim3(256), 0, s, cv, ci, nsplit, ncols, T,
                       prepared + (size_t)KD * Npad, out, idx_out);
    return launch_check(ctx, "knn_match");
}

}  // namespace tvc
