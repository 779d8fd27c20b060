// Host side of ragged batches (ragged.h): the per-call tables (frames per utterance, their prefix, frame -> utterance, per-launch
// column-tile prefixes) are built on the device from lengths that travel as kernel ARGUMENTS - asynchronous on the caller's stream,
// no host staging buffer to keep alive, capturable.
#include "ragged.h"
#include "tvc_common.h"

namespace tvc {

namespace {
struct IntChunk {
    int v[960];
};
__global__ void upload_ints_kernel(IntChunk c, int* __restrict__ dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = c.v[i];
}
// one workgroup: pre = exclusive prefix of tb * mult_num / bn (bn = 0: of tb itself), pre[B] = total; col2b (optional) = utterance of every frame
__global__ __launch_bounds__(1024) void rag_prefix_kernel(const int* __restrict__ tb, int* pre, int* __restrict__ col2b, int B, int mult, int bn) {
    __shared__ int part[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < B; base += 1024) {
        const int b = base + tid;
        int v = 0;
        if (b < B) v = bn > 0 ? (tb[b] * mult + bn - 1) / bn : tb[b];
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(inc, o);
            if (lane >= o) inc += u;
        }
        if (lane == 63) part[wave] = inc;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += part[w];
        const int carry = carry_s;
        const int excl = carry + woff + inc - v;
        if (b < B) pre[b] = excl;
        __syncthreads();
        if (tid == 1023) carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (tid == 0) pre[B] = carry_s;
    if (col2b) {      // frame -> utterance: every thread looks its frames' utterance up in the finished prefix (a 26-minute utterance must not be one thread's loop)
        __syncthreads();
        __threadfence_block();
        const int total = carry_s;
        for (int f = tid; f < total; f += 1024) {
            int lo = 0, hi = B - 1;           // last b with pre[b] <= f
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (pre[mid] <= f) lo = mid;
                else hi = mid - 1;
            }
            col2b[f] = lo;
        }
    }
}
int upload_ints(tvc_ctx* ctx, hipStream_t s, const std::vector<int>& src, int* dst) {
    for (size_t o = 0; o < src.size(); o += 960) {
        IntChunk c;
        const int n = (int)(src.size() - o < 960 ? src.size() - o : 960);
        std::memcpy(c.v, src.data() + o, (size_t)n * sizeof(int));
        hipLaunchKernelGGL(upload_ints_kernel, dim3((n + 255) / 256), dim3(256), 0, s, c, dst + o, n);
    }
    return launch_check(ctx, "rag upload");
}
}  // namespace

int rag_setup(tvc_ctx* ctx, hipStream_t s, bool dry, RagHost& h, const std::vector<int>& frames, const std::vector<int>& rows, int Tmax, int* scratch) {
    h.B = (int)frames.size();
    h.tb = frames;
    h.row = rows;
    h.Tmax = Tmax;
    h.pre.assign(h.B + 1, 0);
    h.Tlong = 0;
    h.Tshort = h.B ? frames[0] : 0;
    for (int b = 0; b < h.B; ++b) {
        h.pre[b + 1] = h.pre[b] + frames[b];
        if (frames[b] > h.Tlong) h.Tlong = frames[b];
        if (frames[b] < h.Tshort) h.Tshort = frames[b];
    }
    h.Ttot = h.pre[h.B];
    int* p = scratch;
    int* d_tb = p;
    p += h.B + 1;
    int* d_pre = p;
    p += h.B + 1;
    int* d_row = p;
    p += h.B + 1;
    int* d_col2b = p;
    p += h.Ttot;
    h.d_pool = p;
    h.pool_slots = kRagTabSlots;
    h.tabs.clear();
    h.d_tb = d_tb;
    h.d_pre = d_pre;
    h.d_row = d_row;
    h.d_col2b = d_col2b;
    if (dry) return 0;
    TVC_CHECK(upload_ints(ctx, s, frames, d_tb));
    TVC_CHECK(upload_ints(ctx, s, rows, d_row));
    hipLaunchKernelGGL(rag_prefix_kernel, dim3(1), dim3(1024), 0, s, d_tb, d_pre, d_col2b, h.B, 1, 0);
    return launch_check(ctx, "rag tables");
}

int rag_min_len(const tvc_ctx* ctx, int len) { return ctx->rag ? ctx->rag->Tshort * (len / ctx->rag->Ttot) : len; }

int rag_view(tvc_ctx* ctx, hipStream_t s, int mult, int bn, RagDev* out, int* ntiles) {
    *out = RagDev{};
    RagHost* h = ctx->rag;
    if (!h) return 0;
    out->tb = h->d_tb;
    out->pre = h->d_pre;
    out->col2b = h->d_col2b;
    out->row = h->d_row;
    out->B = h->B;
    out->mult = mult;
    out->Tmax = h->Tmax;
    if (bn <= 0) return 0;
    for (auto& t : h->tabs)
        if (t.mult == mult && t.bn == bn) {
            out->ts = t.d;
            if (ntiles) *ntiles = t.total;
            return 0;
        }
    if ((int)h->tabs.size() >= h->pool_slots) return fail(ctx, TVC_ERR_STATE, "ragged batch: out of column-tile tables");
    int* d = h->d_pool + (size_t)h->tabs.size() * (h->B + 1);
    int total = 0;
    for (int b = 0; b < h->B; ++b) total += (h->tb[b] * mult + bn - 1) / bn;
    hipLaunchKernelGGL(rag_prefix_kernel, dim3(1), dim3(1024), 0, s, h->d_tb, d, (int*)nullptr, h->B, mult, bn);
    TVC_CHECK(launch_check(ctx, "rag tile table"));
    h->tabs.push_back({mult, bn, total, d});
    out->ts = d;
    if (ntiles) *ntiles = total;
    return 0;
}

}  // namespace tvc
