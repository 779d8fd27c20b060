// match_features with the reference's FULL signature (module/tinyvc/feature_retrieval.py:15-33): k = 1 ... 8 nearest index vectors under
// metrics 'cos' / 'IP' / 'L2', the mean of the k raw vectors.  The inference path only ever asks for k = 4, 'cos' - that is knn.hip's
// prepared-index search on the matrix pipe; this file serves every other argument on the RAW index [768][N] in plain fp32: no prepared
// blob, no split precision, no assumption about the vectors' range (inner products and distances of un-normalised vectors are not bounded
// the way cosines are).  It is not a hot path - HBM- and fma-bound, a few milliseconds for the headline shapes - and is written for
// clarity: exact fp32 similarities in one fixed summation order, a lane-local top-k, one merge per query.
//
//   similarity of query q and index vector r (what torch.topk ranks in the reference):
//     cos   (q / (||q|| + 1e-6)) . (r / (||r|| + 1e-6))     evaluated as (q . r) / ((||q|| + 1e-6) (||r|| + 1e-6))
//     IP    q . r
//     L2    -||q - r||                                        the direct sum of squared differences (torch.cdist's matmul form loses
//                                                             digits to cancellation; on inputs whose neighbours are decidable in fp32
//                                                             both give the same ranks, and this one is the more accurate of the two)
//   ties: lower index first (torch.topk leaves it unspecified); a NaN similarity orders as the maximum, as in torch.topk.
#include "tvc_common.h"

namespace tvc {

namespace {

constexpr int GQ = 4;          // queries per workgroup
constexpr int GK = 8;          // largest k
constexpr int GT = 256;        // threads per workgroup
constexpr int KDG = kSslDim;   // 768

// is (a, ia) ranked before (b, ib)?  NaN first (torch.topk's order), then larger value, then lower index
__device__ __forceinline__ bool ranks_before(float a, int ia, float b, int ib) {
    const bool na = a != a, nb = b != b;
    if (na || nb) return na && (!nb || ia < ib);
    return a > b || (a == b && ia < ib);
}

// one workgroup: GQ consecutive query columns n = b * T + t of src [B][768][T] against the whole index [768][N]
__global__ __launch_bounds__(GT) void knn_general_topk_kernel(const float* __restrict__ src, const float* __restrict__ index, int N, int ncols, int T, int k,
                                                             int metric, int64_t* __restrict__ idx_out, float* __restrict__ val_out) {
    extern __shared__ __attribute__((aligned(16))) float smem_g[];
    float (*qs)[GQ] = reinterpret_cast<float (*)[GQ]>(smem_g);                               // [768][GQ] the queries, channel-major: one 16-byte broadcast read per channel
    float* qn = smem_g + KDG * GQ;                                                           // [GQ (+ pad)] ||q|| + 1e-6
    float (*cv)[GT * GK] = reinterpret_cast<float (*)[GT * GK]>(qn + 16);                    // [GQ][GT * GK] the lanes' candidates: value ...
    int (*ci)[GT * GK] = reinterpret_cast<int (*)[GT * GK]>(qn + 16 + GQ * GT * GK);         // ... and index
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * GQ;
    for (int i = tid; i < KDG * GQ; i += GT) {
        const int c = i / GQ, j = i - c * GQ;
        const int n = n0 + j < ncols ? n0 + j : ncols - 1;
        const int b = n / T, t = n - b * T;
        qs[c][j] = src[((long)b * KDG + c) * T + t];
    }
    __syncthreads();
    if (tid < GQ) {      // torch.norm(p = 2): fp32 sum of squares in ascending channel order, sqrt
        float s = 0.f;
        for (int c = 0; c < KDG; ++c) s = fmaf(qs[c][tid], qs[c][tid], s);
        qn[tid] = sqrtf(s) + 1e-6f;
    }
    __syncthreads();
    float tv[GQ][GK];
    int ti[GQ][GK];
#pragma unroll
    for (int j = 0; j < GQ; ++j)
#pragma unroll
        for (int u = 0; u < GK; ++u) {
            tv[j][u] = -INFINITY;
            ti[j][u] = 0x7fffffff;
        }
    for (int r = tid; r < N; r += GT) {
        float acc[GQ] = {0.f, 0.f, 0.f, 0.f}, rr = 0.f;
        const float* col = index + r;
        if (metric == 2) {
            for (int c = 0; c < KDG; ++c) {
                const float v = col[(long)c * N];
                const float4 q = *reinterpret_cast<const float4*>(&qs[c][0]);
                const float d0 = q.x - v, d1 = q.y - v, d2 = q.z - v, d3 = q.w - v;
                acc[0] = fmaf(d0, d0, acc[0]);
                acc[1] = fmaf(d1, d1, acc[1]);
                acc[2] = fmaf(d2, d2, acc[2]);
                acc[3] = fmaf(d3, d3, acc[3]);
            }
        } else {
            for (int c = 0; c < KDG; ++c) {
                const float v = col[(long)c * N];
                const float4 q = *reinterpret_cast<const float4*>(&qs[c][0]);
                acc[0] = fmaf(q.x, v, acc[0]);
                acc[1] = fmaf(q.y, v, acc[1]);
                acc[2] = fmaf(q.z, v, acc[2]);
                acc[3] = fmaf(q.w, v, acc[3]);
                rr = fmaf(v, v, rr);
            }
        }
        const float rn = sqrtf(rr) + 1e-6f;
#pragma unroll
        for (int j = 0; j < GQ; ++j) {
            float sim = metric == 2 ? -sqrtf(acc[j]) : (metric == 1 ? acc[j] : acc[j] / (qn[j] * rn));
            int id = r;
            // insertion into the lane's sorted list (compile-time register indices: a compare-and-swap chain)
#pragma unroll
            for (int u = 0; u < GK; ++u) {
                const bool before = ranks_before(sim, id, tv[j][u], ti[j][u]);
                const float ov = tv[j][u];
                const int oi = ti[j][u];
                tv[j][u] = before ? sim : ov;
                ti[j][u] = before ? id : oi;
                sim = before ? ov : sim;
                id = before ? oi : id;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < GQ; ++j)
#pragma unroll
        for (int u = 0; u < GK; ++u) {
            cv[j][tid * GK + u] = tv[j][u];
            ci[j][tid * GK + u] = ti[j][u];
        }
    __syncthreads();
    // wave j merges query j: k rounds of "best remaining candidate" over the workgroup's GT * GK candidates
    const int j = wave;      // (GT / 64 == GQ)
    const int n = n0 + j;
    for (int round = 0; round < k; ++round) {
        float bv = -INFINITY;
        int bi = 0x7fffffff, bp = -1;
        for (int p = lane; p < GT * GK; p += 64) {
            const float v = cv[j][p];
            const int i = ci[j][p];
            if (i != 0x7fffffff && ranks_before(v, i, bv, bi)) {
                bv = v;
                bi = i;
                bp = p;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o);
            const int oi = __shfl_xor(bi, o), op = __shfl_xor(bp, o);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || ranks_before(ov, oi, bv, bi))) {
                bv = ov;
                bi = oi;
                bp = op;
            }
        }
        if (lane == 0 && n < ncols) {
            idx_out[(long)n * k + round] = bi == 0x7fffffff ? 0 : bi;
            if (val_out) val_out[(long)n * k + round] = bv;
        }
        if (bp >= 0 && (bp & 63) == lane) ci[j][bp] = 0x7fffffff;      // taken (the owner lane of position bp scans it again next round)
        __builtin_amdgcn_wave_barrier();
    }
}

// out[b][c][t] = mean_j index[c][idx[n][j]]: the k rows added in rank order, divided by k (torch's mean over the stacked neighbours)
__global__ __launch_bounds__(256) void knn_general_gather_kernel(const float* __restrict__ index, int N, const int64_t* __restrict__ idx, int k, int ncols, int T,
                                                                 float* __restrict__ out) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63);
    const int cq = threadIdx.x >> 6;      // four channel classes per workgroup
    if (n >= ncols) return;
    const int b = n / T, t = n - b * T;
    int id[GK];
#pragma unroll
    for (int u = 0; u < GK; ++u) id[u] = u < k ? (int)idx[(long)n * k + u] : 0;
    const float fk = (float)k;
    for (int c = cq + 4 * blockIdx.y; c < KDG; c += 4 * gridDim.y) {
        const float* row = index + (long)c * N;
        float s = row[id[0]];
#pragma unroll
        for (int u = 1; u < GK; ++u)
            if (u < k) s = __fadd_rn(s, row[id[u]]);
        out[((long)b * KDG + c) * T + t] = __fdiv_rn(s, fk);
    }
}

}  // namespace

int run_knn_general(tvc_ctx* ctx, hipStream_t s, Ws& ws, bool dry, const float* src, const float* index, int64_t N, int k, int metric, float* out, int64_t* idx_out,
                    float* val_out, int B, int T) {
    const long ncols = (long)B * T;
    int64_t* idx = idx_out ? idx_out : ws.get<int64_t>((size_t)ncols * k);
    if (dry) return 0;
    static_assert(GT / 64 == GQ, "one merging wave per query");
    constexpr int lds = (KDG * GQ + 16 + 2 * GQ * GT * GK) * 4;
    static bool ready_dev[64] = {};
    bool& ready = ready_dev[ctx->device & 63];
    if (!ready) {
        hipError_t e = hipFuncSetAttribute((const void*)knn_general_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return fail(ctx, TVC_ERR_HIP, "knn_match_general setup: %s", hipGetErrorString(e));
        ready = true;
    }
    hipLaunchKernelGGL(knn_general_topk_kernel, dim3((unsigned)((ncols + GQ - 1) / GQ)), dim3(GT), lds, s, src, index, (int)N, (int)ncols, T, k, metric, idx, val_out);
    hipLaunchKernelGGL(knn_general_gather_kernel, dim3((unsigned)((ncols + 63) / 64), 16), dim3(256), 0, s, index, (int)N, idx, k, (int)ncols, T, out);
    return launch_check(ctx, "knn_match_general");
}

}  // namespace tvc
