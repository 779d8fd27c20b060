// Front door of the entry scripts, on the device: sample-rate conversion to the model's 24 kHz
// (reference infer.py:46,63 / infer_streaming.py:70 call torchaudio.functional.resample), int16 PCM <-> fp32 and dB gain
// (infer_streaming.py:85-94).  torchaudio is third-party arithmetic that is absent from the build image, so the resampler
// restates its documented default algorithm (sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99) and its parity is
// pinned only against this repository's own host restatement (tinyvc_amd/resample.py).
#include <cmath>
#include <map>
#include <utility>

#include "small_kernels.h"
#include <mutex>

#include "tvc_common.h"

namespace tvc {

// y[r][t] = sum_k kern[t % new][k] * x[r][(t / new) * orig + k - width]   (zero outside [0, n))
// = torchaudio's conv1d(pad(x, (width, width + orig)), kern, stride = orig), output interleaved over the `new` phases.
static __global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, long n, long n_out,
                                                              const float* __restrict__ kern, int orig, int newf, int width, int taps) {
    const long total = (long)rows * n_out;
    for (long o = blockIdx.x * (long)blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        const long r = o / n_out, t = o - r * n_out;
        const long j = t / newf;
        const int i = (int)(t - j * newf);
        const float* kr = kern + (long)i * taps;
        const float* xr = x + r * n;
        const long base = j * orig - width;
        float acc = 0.f;
        for (int k = 0; k < taps; ++k) {
            const long p = base + k;
            const float v = (p >= 0 && p < n) ? xr[p] : 0.f;
            acc = fmaf(kr[k], v, acc);
        }
        y[o] = acc;
    }
}

// chunk = int16 / 32768, then torchaudio.functional.gain (x * 10^(dB/20); skipped for 0 dB)   (infer_streaming.py:85-89)
static __global__ void pcm16_to_f32_kernel(const int16_t* __restrict__ pcm, float* __restrict__ y, long n, float ratio, int apply) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = (float)pcm[i] / 32768.f;
        if (apply) v = v * ratio;
        y[i] = v;
    }
}
// gain, * 32768, numpy's float32 -> int16 cast: truncation toward zero, out-of-range values wrap through int32
// (infer_streaming.py:91-94)
static __global__ void f32_to_pcm16_kernel(const float* __restrict__ x, int16_t* __restrict__ pcm, long n, float ratio, int apply) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = x[i];
        if (apply) v = v * ratio;
        v = v * 32768.f;
        const int q = (v != v || fabsf(v) >= 2147483648.f) ? (int)0x80000000 : (int)v;   // x86 cvttss2si's "indefinite" outside int32
        pcm[i] = (int16_t)(unsigned short)(q & 0xffff);
    }
}

static long gcd_l(long a, long b) { return b ? gcd_l(b, a % b) : a; }

struct ResampleTable {
    float* kern = nullptr;
    int orig = 0, newf = 0, width = 0, taps = 0;
};
static std::map<std::pair<tvc_ctx*, std::pair<int, int>>, ResampleTable>& tables() {
    static std::map<std::pair<tvc_ctx*, std::pair<int, int>>, ResampleTable> t;
    return t;
}
static std::mutex& tables_mu() {      // one host thread per ctx is the contract, but the map is shared by the ctxs of a process
    static std::mutex m;
    return m;
}

// Hann-windowed sinc filter bank, computed in fp64 and rounded once (the formula of tinyvc_amd/resample.py:_kernel)
// (the first use of a rate pair allocates and uploads its table synchronously: not inside a stream capture)
static int get_table(tvc_ctx* ctx, int orig_freq, int new_freq, ResampleTable* out) {
    std::lock_guard<std::mutex> lk(tables_mu());
    auto key = std::make_pair(ctx, std::make_pair(orig_freq, new_freq));
    auto it = tables().find(key);
    if (it != tables().end()) {
        *out = it->second;
        return 0;
    }
    const long g = gcd_l(orig_freq, new_freq);
    const int orig = (int)(orig_freq / g), newf = (int)(new_freq / g);
    const double lpw = 6.0, rolloff = 0.99, pi = 3.14159265358979323846;
    const double base = (double)(orig < newf ? orig : newf) * rolloff;
    const int width = (int)std::ceil(lpw * orig / base);
    const int taps = 2 * width + orig;
    std::vector<float> k((size_t)newf * taps);
    for (int i = 0; i < newf; ++i)
        for (int c = 0; c < taps; ++c) {
            double t = ((double)(-i) / newf + (double)(c - width) / orig) * base;
            t = t < -lpw ? -lpw : (t > lpw ? lpw : t);
            const double w = std::cos(t * pi / lpw / 2.0);
            t *= pi;
            const double sinc = t == 0.0 ? 1.0 : std::sin(t) / t;
            k[(size_t)i * taps + c] = (float)(sinc * (w * w) * (base / orig));
        }
    ResampleTable tb;
    tb.orig = orig;
    tb.newf = newf;
    tb.width = width;
    tb.taps = taps;
    TVC_HIP(ctx, hipMalloc((void**)&tb.kern, k.size() * sizeof(float)));      // one-off per (ctx, rate pair), outside any hot path
    TVC_HIP(ctx, hipMemcpy(tb.kern, k.data(), k.size() * sizeof(float), hipMemcpyHostToDevice));
    tables()[key] = tb;
    *out = tb;
    return 0;
}

void frontdoor_release(tvc_ctx* ctx) {
    std::lock_guard<std::mutex> lk(tables_mu());
    for (auto it = tables().begin(); it != tables().end();) {
        if (it->first.first == ctx) {
            if (it->second.kern) (void)hipFree(it->second.kern);
            it = tables().erase(it);
        } else {
            ++it;
        }
    }
}

int64_t resample_out_len(int64_t n, int orig_freq, int new_freq) {
    if (n <= 0 || orig_freq <= 0 || new_freq <= 0) return 0;
    const long g = gcd_l(orig_freq, new_freq);
    const long orig = orig_freq / g, newf = new_freq / g;
    return (newf * n + orig - 1) / orig;          // ceil(new * n / orig), torchaudio's target length
}

int run_resample(tvc_ctx* ctx, hipStream_t s, const float* x, float* y, int rows, int64_t n, int orig_freq, int new_freq) {
    ResampleTable tb;
    TVC_CHECK(get_table(ctx, orig_freq, new_freq, &tb));
    const long n_out = resample_out_len(n, orig_freq, new_freq);
    hipLaunchKernelGGL(resample_kernel, dim3(grid_for((long)rows * n_out)), dim3(256), 0, s, x, y, rows, (long)n, n_out, tb.kern, tb.orig, tb.newf,
                       tb.width, tb.taps);
    return launch_check(ctx, "resample");
}

static float db_ratio(float gain_db) { return (float)std::pow(10.0, (double)gain_db / 20.0); }

int run_pcm16_to_f32(tvc_ctx* ctx, hipStream_t s, const int16_t* pcm, float* y, int64_t n, float gain_db) {
    hipLaunchKernelGGL(pcm16_to_f32_kernel, dim3(grid_for(n)), dim3(256), 0, s, pcm, y, (long)n, db_ratio(gain_db), gain_db != 0.f);
    return launch_check(ctx, "pcm16_to_f32");
}
int run_f32_to_pcm16(tvc_ctx* ctx, hipStream_t s, const float* x, int16_t* pcm, int64_t n, float gain_db) {
    hipLaunchKernelGGL(f32_to_pcm16_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, pcm, (long)n, db_ratio(gain_db), gain_db != 0.f);
    return launch_check(ctx, "f32_to_pcm16");
}

}  // namespace tvc
