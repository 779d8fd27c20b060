// Internal declarations shared by the translation units of libtinyvc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/tinyvc_hip.h"
#include "ragged.h"

namespace tvc {

constexpr int kSampleRate = 24000;
constexpr int kNfft = 1920;
constexpr int kHop = 480;
constexpr int kBins = 961;
constexpr int kSslDim = 768;
constexpr int kSslCh = 384;
constexpr int kPitchCh = 128;
constexpr int kPitchClasses = 512;
constexpr int kSrcCh = 128;
constexpr int kHarm = 15;  // num_harmonics + 1 sinusoids
constexpr int kSolaCross = 1920, kSolaSearch = 1920, kSolaDelay = 3840;
constexpr int kSolaGroups = 8, kSolaPartFloats = 16384;   // lag groups (workgroups) per stream of the correlation search; its scratch in the context's constant arena

// A conv / 1x1 weight packed for the split-precision MFMA kernels (conv3s.h): the two-part fp16 image A6 (K16 steps x MT6 = Mpad / 32
// m-tiles x 2 parts, 1 KiB pieces in MFMA lane order; every m-tile normalised by a power of two, wscale[MT6] = what the epilogue
// multiplies back) plus the bias row [Mpad].  M / K = real rows / k = cin * taps.
struct PackedW {
    const float* bias = nullptr;
    const float* A6 = nullptr;
    const float* wscale = nullptr;   // [MT6] power-of-two scale of every m-tile: W = image * wscale
    const float* wjoint = nullptr;   // the A6 of the image this one shares its m-tile scales with (two convs accumulated into one tile), else nullptr
    int M = 0, K = 0, Mpad = 0, Kpad = 0, cin = 0, taps = 1, MT6 = 0, S6 = 0;   // S6 = 16-channel slabs in A6 (zero-padded to a multiple of 6)
};

struct ConvNeXtW {
    const float* dw_w = nullptr;  // [C][7]
    const float* dw_b = nullptr;  // [C]
    const float* ln_g = nullptr;
    const float* ln_b = nullptr;
    PackedW c2, c3;
    const float* grn_g = nullptr;  // [2C]
    const float* grn_b = nullptr;
    const float* c3_bias_grn = nullptr;  // c3.bias + c3.weight . grn.beta  [Mpad] (GRN's beta folded through the 1x1)
    float ln_bound = 0.f;                // sqrt(C) max|gamma| + max|beta| >= |LayerNorm output|: the first 1x1's input needs no |max| slot while this is < 2^15
    int C = 0, dilation = 1;
};

struct DownW {
    PackedW res, c1, c2, c3;             // c3 and res share their per-m-tile scales (accumulated into one tile)
    const float* c3res_bias = nullptr;   // c3.bias + down_res.bias [c3.Mpad]: c3 launches that fold the residual 1x1 in as a second K phase
    const float* s24c1 = nullptr;   // cin == 24: weight blobs of down24f_kernel (filter_up24s.hip)
    const float* s24c2 = nullptr;
    const float* s24c3r = nullptr;   // c3's blob with c3.bias + down_res.bias and the joint scales (the launch folds the residual 1x1 in)
    float b1_w = 0.f, b1_b = 0.f, b2_w = 0.f, b2_b = 0.f;   // cin == 24: |c1 out| <= b1_w |xi|max + b1_b, |c2 out| <= b2_w |c1 out| + b2_b (down24f_kernel's on-chip intermediates)
    int cin = 0, cout = 0, factor = 1;
};
// conv + FiLM packed for film_s2.h (cin >= 96): one weight image whose 30 KiB units hold everything a (96-row block, 16-channel slab)
// step multiplies - the conv's three taps and the to_scale / to_shift columns of the same 16 channels - and one table of per-row
// constants [6][C]: conv bias, conv row scale, b_scale, b_shift, to_scale row scale, to_shift row scale.
struct FilmU {
    const float* img = nullptr;
    const float* tab = nullptr;
    int C = 0;
    float hb_w = 0.f, hb_b = 0.f;      // bound of the half's FIRST conv (the producer of this kernel's h): |c(x) + b| <= hb_w |x|max + hb_b
};
struct UpW {
    PackedW c1, c2, c3, c4, c5, film1, film2;  // film = [to_scale ; to_shift] stacked on M (2C); film.bias = [b_scale (C) ; b_shift (C)]
    FilmU fu1, fu2;                            // (c2, film1) and (c4, film2) for the single-accumulator pipelined kernel
    const float* s24a = nullptr;               // cin == 24: weight blobs of the two halves of the split-precision fused block (filter_up24s.hip)
    const float* s24b = nullptr;
    float c5_bw = 0.f, c5_bb = 0.f;            // |c5 output| <= c5_bw |its input|max + c5_bb (max_m sum_k |w|, max |b|): the level output's |max| slot without a pass over it
    int cin = 0, cout = 0, factor = 1;
};

struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
};

}  // namespace tvc

struct tvc_prof_region {
    std::string name;
    hipEvent_t a = nullptr, b = nullptr;
};

struct tvc_ctx {
    int device = 0;
    int profiling = 0;                        // tvc_profile_enable: 0 = off, 1 = hipEvent pairs around every named region, 2 = around `filter_net` only
    std::vector<tvc_prof_region> regions;
    std::vector<hipEvent_t> event_pool;       // recycled hipEvents: no hipEventCreate on the hot path
    hipStream_t side = nullptr;               // fork/join stream: the pitch estimator runs beside the SSL chain
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_fork2 = nullptr, ev_join2 = nullptr, ev_amps = nullptr;      // the decoder's fork: FilterNet's input contraction beside SourceNet / dsp (decoder.hip run_decoder)
    tvc::RagHost* rag = nullptr;              // the ragged batch the drivers are currently running for (ragged.h); nullptr = equal lengths
    int rag_batch_frames = 0;                 // tvc_ctx_set_ragged_batch_frames: frames per in-kernel batch of THIS context's ragged calls (0 = the default)
    bool enc_ready = false, dec_ready = false;  // which checkpoint groups tvc_finalize_weights packed
    char enc_missing[160] = {0}, dec_missing[160] = {0};
    std::map<std::string, tvc::HostTensor> host;  // staged checkpoint tensors
    std::vector<float> pitch_table;
    float* arena = nullptr;        // packed checkpoint weights, one allocation per finalize
    float* const_arena = nullptr;  // FFT tables, built at ctx_create
    size_t arena_floats = 0;
    char err[512] = {0};

    // constant tables
    const float* fft_tw960 = nullptr;    // fft.hip tables: (cos, sin)(2 pi j / 960) [960], (cos, sin)(2 pi k / 1920) [961], periodic Hann [1920]
    const float* fft_tw1920 = nullptr;
    const float* fft_hann = nullptr;
    const float* pitch_freq = nullptr;  // [512]
    const float* sola_part = nullptr;   // scratch of the split SOLA search: (best value, best lag) per (stream, lag group); kSolaPartFloats floats

    // encoder
    tvc::PackedW enc_in;  // ssl(384) and pitch(128) input 1x1 stacked: M = 512
    const float* ssl_ln_g = nullptr;
    const float* ssl_ln_b = nullptr;
    const float* pit_ln_g = nullptr;
    const float* pit_ln_b = nullptr;
    tvc::ConvNeXtW ssl_mid[6], pit_mid[4], src_mid[3];
    tvc::PackedW ssl_out, pit_out;
    // source net
    tvc::PackedW src_content_in, src_to_amps, src_to_kernel;
    const float* src_e_w = nullptr;
    const float* src_e_b = nullptr;
    const float* src_f_w = nullptr;
    const float* src_f_b = nullptr;
    // filter net
    tvc::PackedW flt_content_in;
    const float* flt_down0s = nullptr;   // downs.0 weight blob of the split-precision kernel (filter_up24s.hip)
    float down0_bw = 0.f, down0_bb = 0.f; // |downs.0 output| <= down0_bw |input|max + down0_bb: the scale skips[0] is written / read with
    float flt_in_bw = 0.f, flt_in_bb = 0.f;   // |content_in(content) + f0_in(log f0)| <= flt_in_bw |content|max + flt_in_bb: FilterNet's x0 slot without a pass over x0
    const float* flt_f_w = nullptr;
    const float* flt_f_b = nullptr;
    tvc::DownW downs[4];
    tvc::UpW ups[5];
};

namespace tvc {

inline int fail(tvc_ctx* ctx, int code, const char* fmt, ...) {
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

#define TVC_HIP(ctx, call)                                                                       \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return tvc::fail(ctx, TVC_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                             __FILE__, __LINE__);                                                \
    } while (0)

#define TVC_CHECK(expr)        \
    do {                       \
        int rc_ = (expr);      \
        if (rc_ != 0) return rc_; \
    } while (0)

// RAII region timer: records a hipEvent pair on the launch stream when ctx->profiling is on.
struct ProfScope {
    tvc_ctx* ctx;
    hipStream_t s;
    int idx = -1;
    ProfScope(tvc_ctx* c, hipStream_t st, bool dry, const char* name) : ctx(c), s(st) {
        if (!c || !c->profiling || dry) return;
        if (c->profiling == 2 && std::strncmp(name, "filter_net", 10) != 0) return;     // the roofline's regions alone (filter_net, and filter_net.input@side when the input contraction is forked): 2-4 event records per step instead of 38
        tvc_prof_region r;
        r.name = name;
        auto take = [&](hipEvent_t* e) {
            if (!c->event_pool.empty()) {
                *e = c->event_pool.back();
                c->event_pool.pop_back();
                return true;
            }
            return hipEventCreate(e) == hipSuccess;
        };
        if (!take(&r.a) || !take(&r.b)) return;
        (void)hipEventRecord(r.a, s);
        c->regions.push_back(r);
        idx = (int)c->regions.size() - 1;
    }
    ~ProfScope() {
        if (idx >= 0) (void)hipEventRecord(ctx->regions[idx].b, s);
    }
};

// Bump allocator over the caller's workspace.  In dry mode it only measures.
struct Ws {
    char* base;
    size_t cap, off = 0, peak = 0;
    bool dry;
    Ws(void* p, size_t c, bool d) : base((char*)p), cap(c), dry(d) {}
    template <class T>
    T* get(size_t n) {
        size_t bytes = (n * sizeof(T) + 255) & ~size_t(255);
        size_t o = off;
        off += bytes;
        if (off > peak) peak = off;
        if (dry) return (T*)(uintptr_t)256;  // never dereferenced
        return (T*)(base + o);
    }
    size_t mark() const { return off; }
    void release(size_t m) { off = m; }
    bool ok() const { return dry || peak <= cap; }
};

inline int launch_check(tvc_ctx* ctx, const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, TVC_ERR_HIP, "launch %s: %s", what, hipGetErrorString(e));
    return 0;
}

// ---- stage drivers (each enqueues kernels on `s`; `dry` = measure workspace only) ----------
int run_stft(tvc_ctx*, hipStream_t, Ws&, bool dry, const float* wav, float* spec, int B, int64_t L);
int run_stft_fft(tvc_ctx*, hipStream_t, const float* wav, float* spec, int B, int64_t L);
int run_noise_ifft(tvc_ctx*, hipStream_t, const float* kern, const float* angle, uint64_t seed, float* frames, int B, int T, bool angle_padded = false);      // angle == nullptr: phases drawn in the kernel from `seed`
// emax (optional, equal-length batches only): per-utterance max of the pooled |x| = max |wav| of the utterance, written (not accumulated)
int run_energy(tvc_ctx*, hipStream_t, Ws&, bool dry, const float* wav, float* energy, int B, int64_t L, float* emax = nullptr, float* spec_bound = nullptr,
               float* zero = nullptr, int nz = 0);      // (emax given: the pooled-maximum launch also zeroes zero[0 .. nz))
const float* knn_index_amax(const float* prepared);
// out[b] = a * in[b * in_stride] + c for b < n: a |max| slot from the slot of the tensor it is a bounded function of (frontend.hip)
int run_slot_affine(tvc_ctx*, hipStream_t, float* out, const float* in, int in_stride, float a, float c, int n);
int run_slot_prep(tvc_ctx*, hipStream_t, float* zero, int nz, float* o1, const float* in1, int s1, float a1, float c1, float* o2, const float* in2, int s2, float a2,
                  float c2, float* o3, float a3, float c3, int n);
// prepared kNN blob (knn.hip): header word 0 = magic, word 5 = format version.  Version 2 (round 5): word 4 holds the raw vectors' |max| (a float), which
// the decoder takes as the bound of `matched` - a blob of another version has 0 there (= "no scaling": the fp16 range guard silently off) and is refused.
constexpr int kBlobMagic = 0x54564B4E, kBlobVersion = 2;
constexpr int kFilterSlotX = 2;       // ... and the one of its input contraction's output (S_X)
constexpr int kFilterSlots = 41;      // run_filter's |max| slots per utterance (decoder.hip S_COUNT)
      // device pointer to the prepared index's |max| (one float)
// spec_bound (optional): per-utterance upper bounds of |spec| (the slot of the input contraction); nullptr = one pass over spec measures it
int run_encoder(tvc_ctx*, hipStream_t, Ws&, bool dry, const float* spec, float* ssl, float* f0,
                float* logits, int B, int T, const float* spec_bound = nullptr, float* zeroed_slots = nullptr,      // zeroed_slots: 3 x utterances floats already zeroed on this stream
                float* f0_shifted = nullptr, float shift = 0.f);      // f0_shifted: also shift_frequency(f0, shift)
int run_pitch_decode(tvc_ctx*, hipStream_t, const float* logits, float* f0, int B, int T);
int run_knn(tvc_ctx*, hipStream_t, Ws&, bool dry, const float* src, const float* prepared, int64_t N,
            float* out, int64_t* idx_out, int B, int T);
int run_knn_topk(tvc_ctx*, hipStream_t, Ws&, bool dry, const float* src, const float* prepared, int64_t N,
                 float* sims_out, int64_t* idx_out, int B, int T);
// match_features for any k <= 8 and metric (0 cos, 1 IP, 2 L2) on the RAW index [768][N] in plain fp32 (knn_general.hip)
int run_knn_general(tvc_ctx*, hipStream_t, Ws&, bool dry, const float* src, const float* index, int64_t N, int k, int metric, float* out, int64_t* idx_out,
                    float* val_out, int B, int T);
int run_knn_slots(tvc_ctx*, hipStream_t, const float* prepared, int64_t N, const int64_t* idx, float* slots, int64_t nslots);
int run_knn_finish(tvc_ctx*, hipStream_t, const float* slots, float* out, int B, int T);
int run_shift(tvc_ctx*, hipStream_t, const float* f0, float* out, int64_t n, float semitones);
int run_uniform_to_angle(tvc_ctx*, hipStream_t, float* u, int64_t n);
// content_bound (optional): ONE float, an upper bound of |content| (the prepared index's |max| when content came out of the kNN match);
// energy_bound (optional): per-utterance upper bounds of |energy|.  nullptr = measured by a pass over the tensor.
int run_decoder(tvc_ctx*, hipStream_t, Ws&, bool dry, const float* content, const float* f0,
                const float* energy, const float* angle, uint64_t seed, float* wave, float* amps_out,
                float* kernel_out, float* source_out, int B, int T, const float* content_bound = nullptr, const float* energy_bound = nullptr);
struct FilterTaps {   // optional copies of FilterNet's block outputs (tvc_filter_net_f32)
    float* skips[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    float* ups[4] = {nullptr, nullptr, nullptr, nullptr};
};
// cmax / smax / the trailing float* of run_dsp: per-utterance |max| slots of content / cat[source, energy] (block-floating-point
// guard of the fp16 split, conv3s.h); nullptr = the stage computes (or keeps) its own
int run_filter(tvc_ctx*, hipStream_t, Ws&, bool dry, const float* content, const float* f0, const float* energy,
               const float* source, float* wave, int B, int T, const FilterTaps* taps = nullptr, const float* cmax = nullptr, const float* smax = nullptr,
               float* zeroed_slots = nullptr, bool x_slot_set = false, float* x_pre = nullptr, hipEvent_t x_ready = nullptr);      // zeroed_slots: kFilterSlots x utterances floats the caller has already zeroed on this
                                                                             // stream; x_slot_set: ... and has set slot kFilterSlotX to flt_in_bw |content|max + flt_in_bb; x_pre / x_ready: the input contraction's output, already
                                                                             // launched by the caller on another stream, and the event that says it is there (run_decoder's fork)
// run_decoder's fork (decoder.hip): the harmonic oscillator on the context's side stream beside SourceNet's output GEMMs and the noise branch.
// csum = the frame sums' buffer, already holding the scanned sums (launched on `side` by the caller); amps_ready = recorded on the launch stream
// behind the amplitudes' GEMM; the caller joins `side` itself.
struct DspFork {
    hipStream_t side;
    double* csum;
    hipEvent_t amps_ready;
};
int run_dsp(tvc_ctx*, hipStream_t, Ws&, bool dry, const float* f0, const float* amps, const float* kern,
            const float* angle, uint64_t seed, float* source, int B, int T, float* smax = nullptr, const DspFork* fk = nullptr);
int run_sola(tvc_ctx*, hipStream_t, const float* y, float* sola_buf, const float* fade_in, float* out, int32_t* shift_out,
             int S, int64_t Ly, int block, int use_pv);
int run_stream_push(tvc_ctx* ctx, hipStream_t s, float* buf, const float* blocks, int S, int n, int m);
int64_t resample_out_len(int64_t n, int orig_freq, int new_freq);
int run_resample(tvc_ctx*, hipStream_t, const float* x, float* y, int rows, int64_t n, int orig_freq, int new_freq);
int run_pcm16_to_f32(tvc_ctx*, hipStream_t, const int16_t* pcm, float* y, int64_t n, float gain_db);
int run_f32_to_pcm16(tvc_ctx*, hipStream_t, const float* x, int16_t* pcm, int64_t n, float gain_db);
void frontdoor_release(tvc_ctx*);
int run_prepare_index(tvc_ctx*, hipStream_t, const float* index, float* prepared, int64_t N);
int run_prepare_index_f16(tvc_ctx*, hipStream_t, const void* rows_f16, float* prepared, int64_t N);

// fused FilterNet kernels (filter_up24s.hip, conv48s.hip)
// (the amax_* arguments are the per-utterance |max| slots of the block-floating-point guard, conv3s.h)
// skips[0] travels as the FiLM 1x1s' ready operand (two fp16 planes, scaled by the bound cbw |max of downs.0's input| + cbb): run_down0_split writes
// it (out_fp32: optional fp32 copy for the parity taps), run_up24_split reads it (amax_c = the slot of downs.0's INPUT)
int run_up24_split(tvc_ctx*, hipStream_t, const UpW& u, const float* x, const float* cond_planes, float cbw, float cbb, float* x1, float* out, int B, int len,
                   const float* amax_x, const float* amax_c, float* amax_x1);
int run_down0_split(tvc_ctx*, hipStream_t, const float* blob, const float* source, const float* energy, float* planes, float* out_fp32, float* y2, int B, int len,
                    const float* amax_x, float* amax_y);
int run_down24_fused(tvc_ctx*, hipStream_t, const DownW& d, const float* xi, float* out, float* y2, int B, int len, const float* amax_xi, float* amax_out);
int run_conv48s(tvc_ctx*, hipStream_t, const PackedW& w, const float* x, int lin, float lscale, const PackedW* film, const float* bsc, const float* bsh,
                const float* cond, const float* res, int rlin, float rscale, float* out, int B, int len, int dil, const float* amax_x, const float* amax_c,
                float* amax_y, const PackedW* c5 = nullptr, float* out5 = nullptr);

int run_conv48_pair(tvc_ctx*, hipStream_t, const PackedW& wa, const PackedW& wb, const float* x, int lin, float lscale, const PackedW* film, const float* bsc,
                    const float* bsh, const float* cond, const float* res, int rlin, float rscale, float* out, int B, int len, int da, int db,
                    const float* amax_x, const float* amax_c, float* amax_y);

// ConvNeXt-v2 layer on x [B, C, T] in place (convnext.py:49-58); tmp buffers from ws.
int run_convnext(tvc_ctx*, hipStream_t, Ws&, bool dry, const ConvNeXtW& w, float* x, int B, int T, float* amax_out = nullptr);

}  // namespace tvc
