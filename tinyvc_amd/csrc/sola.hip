// Streaming tail: SOLA alignment + cross-fade (stream.py:74-95) and the optional phase-vocoder
// cross-fade (stream.py:9-26), batched over streams, with the arg-max kept on the device (the
// reference's `.item()` host sync at stream.py:79 is what serialises concurrent streams).
#include "tvc_common.h"

namespace tvc {

constexpr int CROSS = kSolaCross, SEARCH = kSolaSearch, DELAY = kSolaDelay;

// phase_vocoder(a = sola buffer, b = new head, fade_out, fade_in) for n = 1920, written into `res`.
// Direct DFT of the two windowed segments (961 bins x 1920 samples each), then the 961-term cosine
// bank per output sample — O(n^2) like the reference's [1920, 961] broadcast.
static __device__ void phase_vocoder_block(const float* a, const float* b, const float* fin, float* res,
                                           float* re_a, float* im_a, float* re_b, float* im_b) {
    const int n = CROSS, nb = n / 2 + 1;
    const float two_pi = 6.283185307179586f;
    for (int k = threadIdx.x; k < nb; k += blockDim.x) {
        float ra = 0.f, ia = 0.f, rb = 0.f, ib = 0.f;
        for (int j = 0; j < n; ++j) {
            float w = sqrtf((1.f - fin[j]) * fin[j]);
            float sn, cs;
            int r = (int)(((long)k * j) % n);
            sincosf(two_pi * (float)r / (float)n, &sn, &cs);
            float av = a[j] * w, bv = b[j] * w;
            ra = fmaf(av, cs, ra);
            ia = fmaf(-av, sn, ia);
            rb = fmaf(bv, cs, rb);
            ib = fmaf(-bv, sn, ib);
        }
        re_a[k] = ra; im_a[k] = ia; re_b[k] = rb; im_b[k] = ib;
    }
    __syncthreads();
    // per-bin magnitude sum, phase of a, wrapped phase advance
    for (int k = threadIdx.x; k < nb; k += blockDim.x) {
        float mag = hypotf(re_a[k], im_a[k]) + hypotf(re_b[k], im_b[k]);
        if (k >= 1 && k < nb - 1) mag *= 2.f;
        float pa = atan2f(im_a[k], re_a[k]);
        float pb = atan2f(im_b[k], re_b[k]);
        float dp = pb - pa;
        dp = dp - two_pi * floorf(dp / 2.f / 3.14159265358979f + 0.5f);
        re_a[k] = mag;                       // reuse: magnitude
        im_a[k] = pa;                        // phase of a
        re_b[k] = two_pi * (float)k + dp;    // w
    }
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        float t = (float)j / (float)n;
        float acc = 0.f;
        for (int k = 0; k < nb; ++k) acc += re_a[k] * cosf(re_b[k] * t + im_a[k]);
        float fi = fin[j], fo = 1.f - fi;
        float w = sqrtf(fo * fi);
        res[j] = a[j] * (fo * fo) + b[j] * (fi * fi) + acc * w / (float)n;
    }
    __syncthreads();
}

// Normalised cross-correlation over the SEARCH + 1 lags (the two F.conv1d of stream.py:77-78), split over kSolaGroups workgroups
// per stream: one lag per thread, the same sums in the same order as the one-workgroup kernel below (bit-identical values), so
// the search of 32 streams fills the chip instead of 32 CUs (173 us -> see DESIGN.md).  part[(st * G + g) * 2] = best value of
// the group, [.. + 1] = its lag (as bits); ties keep the lowest lag.
static __global__ __launch_bounds__(256) void sola_corr_kernel(const float* __restrict__ y, const float* __restrict__ sola_buf,
                                                               float* __restrict__ part, long Ly, int block) {
    constexpr int LPG = (SEARCH + 1 + kSolaGroups - 1) / kSolaGroups;     // 241 lags per group
    __shared__ float ci[LPG + CROSS];
    __shared__ __attribute__((aligned(16))) float sb[CROSS];
    __shared__ float bestv[4];
    __shared__ int besti[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int st = blockIdx.x / kSolaGroups, g = blockIdx.x - st * kSolaGroups;
    const int lag0 = g * LPG;
    const int tw_len = block + CROSS + SEARCH;
    const float* tw = y + (long)st * Ly + (Ly - tw_len - DELAY) + lag0;
    const int nload = (lag0 + LPG + CROSS <= CROSS + SEARCH ? LPG + CROSS : CROSS + SEARCH - lag0);
    for (int i = tid; i < LPG + CROSS; i += 256) {
        ci[i] = i < nload ? tw[i] : 0.f;
    }
    for (int i = tid; i < CROSS; i += 256) sb[i] = sola_buf[(long)st * CROSS + i];
    __syncthreads();
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    const int lag = lag0 + tid;
    if (tid < LPG && lag <= SEARCH) {
        float nom = 0.f, den = 0.f;
        // (the loop is LDS-bound - three 4-byte reads per term for four waves -: the buffer's four values come as one 16-byte broadcast read
        // and the square is taken in a register; same products and sums in the same order)
        const float* c = ci + tid;
        static_assert(CROSS % 4 == 0, "four terms per broadcast read");
#pragma unroll 2
        for (int j = 0; j < CROSS; j += 4) {
            const float4 b4 = *reinterpret_cast<const float4*>(sb + j);
            const float c0 = c[j], c1 = c[j + 1], c2 = c[j + 2], c3 = c[j + 3];
            nom = fmaf(c0, b4.x, nom);
            den = __fadd_rn(den, __fmul_rn(c0, c0));
            nom = fmaf(c1, b4.y, nom);
            den = __fadd_rn(den, __fmul_rn(c1, c1));
            nom = fmaf(c2, b4.z, nom);
            den = __fadd_rn(den, __fmul_rn(c2, c2));
            nom = fmaf(c3, b4.w, nom);
            den = __fadd_rn(den, __fmul_rn(c3, c3));
        }
        bv = nom / sqrtf(den + 1e-8f);
        bi = lag;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(bv, o);
        int oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { bestv[wave] = bv; besti[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        float v = bestv[0];
        int i = besti[0];
        for (int w = 1; w < 4; ++w)
            if (bestv[w] > v || (bestv[w] == v && besti[w] < i)) { v = bestv[w]; i = besti[w]; }
        part[(long)blockIdx.x * 2] = v;
        part[(long)blockIdx.x * 2 + 1] = __int_as_float(i);
    }
}

// one workgroup per stream; `part` != nullptr: the lag search was done by sola_corr_kernel, only its kSolaGroups candidates are compared here
static __global__ __launch_bounds__(256) void sola_kernel(const float* __restrict__ y, float* __restrict__ sola_buf,
                                                          const float* __restrict__ fade_in, float* __restrict__ out,
                                                          int32_t* __restrict__ shift_out, long Ly, int block, int use_pv,
                                                          const float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* ci = sm;                       // [CROSS + SEARCH] head of temp_wav
    float* sb = sm + CROSS + SEARCH;      // [CROSS] sola buffer
    float* sq = sb + CROSS;               // [CROSS + SEARCH] squares of ci
    __shared__ float bestv[4];
    __shared__ int besti[4];
    __shared__ int s_shift;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int st = blockIdx.x;
    const int tw_len = block + CROSS + SEARCH;  // temp_wav = y[-tw_len-DELAY : -DELAY]
    const float* tw = y + (long)st * Ly + (Ly - tw_len - DELAY);
    float* sbuf = sola_buf + (long)st * CROSS;

    if (!part) {
        for (int i = tid; i < CROSS + SEARCH; i += 256) {
            float v = tw[i];
            ci[i] = v;
            sq[i] = v * v;
        }
    }
    for (int i = tid; i < CROSS; i += 256) sb[i] = sbuf[i];
    __syncthreads();

    // normalised cross-correlation over SEARCH+1 lags (two F.conv1d in stream.py:77-78)
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    if (part) {
        if (tid < kSolaGroups) {
            bv = part[((long)st * kSolaGroups + tid) * 2];
            bi = __float_as_int(part[((long)st * kSolaGroups + tid) * 2 + 1]);
        }
    } else {
        for (int lag = tid; lag <= SEARCH; lag += 256) {
            float nom = 0.f, den = 0.f;
            for (int j = 0; j < CROSS; ++j) {
                nom = fmaf(ci[lag + j], sb[j], nom);
                den += sq[lag + j];
            }
            float v = nom / sqrtf(den + 1e-8f);
            if (v > bv) { bv = v; bi = lag; }   // ascending lags per thread: first maximum kept
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(bv, o);
        int oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { bestv[wave] = bv; besti[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        float v = bestv[0];
        int i = besti[0];
        for (int w = 1; w < 4; ++w)
            if (bestv[w] > v || (bestv[w] == v && besti[w] < i)) { v = bestv[w]; i = besti[w]; }
        s_shift = i;
        if (shift_out) shift_out[st] = i;
    }
    __syncthreads();
    const int shift = s_shift;
    const float* seg = tw + shift;  // temp_wav[shift : shift + block + CROSS]

    // cross-faded head (first CROSS samples) into ci[0:CROSS]
    if (use_pv) {
        float* head = sq;            // b = seg[:CROSS]
        for (int i = tid; i < CROSS; i += 256) head[i] = seg[i];
        __syncthreads();
        float* scratch = sm + 2 * (CROSS + SEARCH) + CROSS + 8;
        phase_vocoder_block(sb, head, fade_in, ci, scratch, scratch + 968, scratch + 2 * 968, scratch + 3 * 968);
    } else {
        for (int i = tid; i < CROSS; i += 256) {
            float fi = fade_in[i];
            ci[i] = __fadd_rn(__fmul_rn(seg[i], fi), __fmul_rn(sb[i], 1.f - fi));
        }
        __syncthreads();
    }
    // temp = [head (CROSS) | seg[CROSS : block + CROSS]];  out = temp[:block]; sola = temp[block:]
    float* o = out + (long)st * block;
    for (int i = tid; i < block; i += 256) o[i] = i < CROSS ? ci[i] : seg[i];
    for (int i = tid; i < CROSS; i += 256) {
        int src = block + i;
        sbuf[i] = src < CROSS ? ci[src] : seg[src];
    }
}

int run_sola(tvc_ctx* ctx, hipStream_t s, const float* y, float* sola_buf, const float* fade_in, float* out,
             int32_t* shift_out, int S, int64_t Ly, int block, int use_pv) {
    size_t lds = (size_t)(2 * (CROSS + SEARCH) + CROSS + 8 + 4 * 968) * sizeof(float) + 64;
    // the lag search on S * kSolaGroups workgroups when its scratch (context-owned, one call at a time like every other scratch of the
    // context) holds their candidates; the arg-max, cross-fade and buffer update stay one workgroup per stream
    float* part = nullptr;
    if (ctx->sola_part && (long)S * kSolaGroups * 2 <= kSolaPartFloats) {
        part = const_cast<float*>(ctx->sola_part);
        hipLaunchKernelGGL(sola_corr_kernel, dim3(S * kSolaGroups), dim3(256), 0, s, y, sola_buf, part, (long)Ly, block);
    }
    hipLaunchKernelGGL(sola_kernel, dim3(S), dim3(256), lds, s, y, sola_buf, fade_in, out, shift_out, (long)Ly, block, use_pv, part);
    return launch_check(ctx, "sola");
}

// The rolling input buffer of a stream (stream.py:69-70: `input_wav = roll(input_wav, -block)`, `input_wav[-block:] = block`), in place and in
// ONE launch instead of three ATen kernels: buf[st][i] = buf[st][i + m] for i < n - m, buf[st][n - m + j] = blocks[st][j].  One workgroup per
// stream reads the whole row into registers (PER x 1024 >= n), meets at a barrier, writes it back shifted.
template <int PER>
static __global__ __launch_bounds__(1024) void stream_push_kernel(float* __restrict__ buf, const float* __restrict__ blocks, int n, int m) {
    float* b = buf + (long)blockIdx.x * n;
    const float* nb = blocks + (long)blockIdx.x * m;
    float v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = threadIdx.x + j * 1024;
        v[j] = i < n - m ? b[i + m] : (i < n ? nb[i - (n - m)] : 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = threadIdx.x + j * 1024;
        if (i < n) b[i] = v[j];
    }
}
int run_stream_push(tvc_ctx* ctx, hipStream_t s, float* buf, const float* blocks, int S, int n, int m) {
    const int per = (n + 1023) / 1024;
    if (per <= 16) hipLaunchKernelGGL(stream_push_kernel<16>, dim3((unsigned)S), dim3(1024), 0, s, buf, blocks, n, m);
    else if (per <= 32) hipLaunchKernelGGL(stream_push_kernel<32>, dim3((unsigned)S), dim3(1024), 0, s, buf, blocks, n, m);
    else return fail(ctx, TVC_ERR_ARG, "stream_push: a stream buffer holds at most 32 768 samples");
    return launch_check(ctx, "stream_push");
}

}  // namespace tvc
