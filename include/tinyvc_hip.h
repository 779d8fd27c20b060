/* tinyvc_hip.h — C ABI of libtinyvc_hip.so: the MI355X (gfx950) implementation of the tinyvc
 * voice-conversion inference path (encoder -> kNN match -> source-filter decoder -> SOLA).
 *
 * The reference (uthree/tinyvc) has no FFI: its boundary is the Python module surface.  Each entry
 * point below names the reference function it stands in for; the Python host package
 * (`tinyvc_amd.module.*`) re-exposes those functions with the reference's names and binds them to
 * these symbols through ctypes (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - every `const float*` / `float*` tensor argument is a DEVICE pointer owned by the caller
 *     (a torch tensor's data_ptr), fp32, contiguous, channels-first [B, C, T] like the reference;
 *     16-byte aligned;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); all
 *     work is enqueued asynchronously on it; nothing here synchronises the device (one exception: the first use of a prepared
 *     index this process did not prepare itself reads its 24-byte header, see tvc_knn_prepare_index_f32).  Outside a stream
 *     capture the encoder forks its pitch estimator, and the decoder its harmonic oscillator and FilterNet's input contraction,
 *     onto a context-owned side stream and joins them back onto `stream` with events before the call returns, so `stream` order
 *     is all a caller ever sees; while `stream` is being captured (and for ragged batches, in the decoder) everything stays on
 *     `stream` (one chain replays faster than a fork inside a graph: DESIGN.md section 4).  Results do not depend on it;
 *   - stream capture: every call may be captured into a HIP graph, with two rules.  Kernel arguments are baked into the graph,
 *     so a call that draws its own noise phases (noise_angle = NULL: the seed is an argument) is refused with TVC_ERR_STATE while
 *     capturing - pass noise_angle and refill that buffer between replays.  And a prepared index must have been used (or
 *     prepared) by this process once before the capture, so that its header check does not have to synchronise;
 *   - `ws` is a caller-allocated device scratch buffer of at least tvc_workspace_bytes() bytes;
 *     the library never allocates device memory after tvc_finalize_weights();
 *   - arithmetic: fp32 in, fp32 out.  Channel contractions run on the fp16 matrix pipe with every fp32 operand split into two
 *     fp16 parts (22 significand bits, fp32-GEMM-level error); a per-utterance power-of-two scale keeps the parts inside fp16's
 *     range, so results do not depend on the loudness of the input, and one utterance's range never affects another utterance of
 *     the batch.  The scale comes from an UPPER BOUND of the tensor's largest magnitude: measured by the producing kernel's epilogue
 *     where that is free, analytic elsewhere (max |wav| bounds the energy envelope and, times the Hann window's sum, every |STFT|
 *     bin; the index's |max|, stored in the prepared blob, bounds the matched content; an l1 norm of the weights times the input's
 *     bound + the largest bias bounds a 1x1's output); inside [2^-10, 2^15) nothing is scaled, so a bound and the exact maximum give
 *     the same bits.  The whole-path calls (tvc_convert_*) and the stage calls derive them per utterance in the same way for equal
 *     and ragged batches;
 *   - return value: 0 = ok, negative = tvc_status; tvc_last_error() has the message;
 *   - one ctx per device, one host thread per ctx at a time.
 */
#ifndef TINYVC_HIP_H
#define TINYVC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tvc_ctx tvc_ctx;

enum tvc_status {
    TVC_OK = 0,
    TVC_ERR_ARG = -1,        /* bad shape / null pointer / unsupported size */
    TVC_ERR_HIP = -2,        /* a HIP runtime call failed (message has the hipError string) */
    TVC_ERR_STATE = -3,      /* weights missing / not finalised */
    TVC_ERR_WORKSPACE = -4   /* workspace too small */
};

#define TVC_ABI_VERSION 1
int tvc_version(void);

/* lifecycle ------------------------------------------------------------------------------- */
int tvc_ctx_create(int hip_device, tvc_ctx** out);
void tvc_ctx_destroy(tvc_ctx* ctx);
const char* tvc_last_error(const tvc_ctx* ctx);

/* Checkpoint loading: one call per state_dict entry of encoder.pt / decoder.pt
 * (reference infer.py:34-37 `load_state_dict`; key names and shapes: SURVEY.md §2.4).
 * `host_data` is a HOST pointer to contiguous fp32; the library copies it.  After the last tensor,
 * tvc_finalize_weights() packs everything into the kernels' layouts and uploads once. */
int tvc_load_tensor(tvc_ctx* ctx, const char* state_dict_key, const float* host_data,
                    const int64_t* shape, int ndim);
/* 512-entry class->Hz table of PitchEstimator.id2freq (reference encoder.py:48-54), host fp32. */
int tvc_set_pitch_table(tvc_ctx* ctx, const float* host_freqs, int n);
int tvc_finalize_weights(tvc_ctx* ctx);

/* Scratch needed by any entry point below for a batch of B utterances of L samples (L % 480 == 0)
 * matched against an index of N vectors. */
int tvc_workspace_bytes(tvc_ctx* ctx, int B, int64_t L, int64_t N, size_t* out_bytes);

/* front end ------------------------------------------------------------------------------- */
/* module.utils.spectrogram (reference module/utils/spectrogram.py:8-15):
 * wav [B, L] -> spec [B, 961, T], T = L/480. */
int tvc_stft_mag_f32(tvc_ctx* ctx, void* stream, const float* wav, float* spec, int B, int64_t L,
                     void* ws, size_t ws_bytes);
/* module.utils.estimate_energy (reference module/utils/energy_estimation.py:9-14):
 * wav [B, L] -> energy [B, 1, L]. */
int tvc_energy_f32(tvc_ctx* ctx, void* stream, const float* wav, float* energy, int B, int64_t L,
                   void* ws, size_t ws_bytes);

/* front door ------------------------------------------------------------------------------- */
/* torchaudio.functional.resample(waveform, orig_freq, new_freq) as the entry scripts call it (reference infer.py:46,63;
 * infer_streaming.py:70): x [rows, n] -> y [rows, tvc_resample_out_len(n, ...)] with torchaudio's default Hann-windowed
 * sinc polyphase filter (lowpass_filter_width 6, rolloff 0.99).  torchaudio itself is not available to pin this against
 * (SURVEY.md 8c): the parity reference is this repository's host restatement tinyvc_amd/resample.py. */
int64_t tvc_resample_out_len(int64_t n, int orig_freq, int new_freq);
int tvc_resample_f32(tvc_ctx* ctx, void* stream, const float* x, float* y, int rows, int64_t n, int orig_freq,
                     int new_freq);
/* The streaming loop's sample conversions (reference infer_streaming.py:85-94): int16 PCM -> float / 32768 ->
 * torchaudio.functional.gain(gain_db) on the way in; gain(gain_db) -> * 32768 -> numpy's float32 -> int16 cast
 * (truncation toward zero, wrap-around outside int16) on the way out.  gain_db == 0 skips the gain multiply, as
 * torchaudio does. */
int tvc_pcm16_to_f32(tvc_ctx* ctx, void* stream, const int16_t* pcm, float* y, int64_t n, float gain_db);
int tvc_f32_to_pcm16(tvc_ctx* ctx, void* stream, const float* x, int16_t* pcm, int64_t n, float gain_db);

/* encoder --------------------------------------------------------------------------------- */
/* Encoder.infer (reference module/tinyvc/encoder.py:113-116): spec [B,961,T] ->
 * ssl [B,768,T], f0 [B,1,T]; `logits` [B,512,T] is optional (NULL to skip): Encoder.forward's
 * second output (encoder.py:108-111). */
int tvc_encoder_f32(tvc_ctx* ctx, void* stream, const float* spec, float* ssl, float* f0,
                    float* logits, int B, int T, void* ws, size_t ws_bytes);

/* PitchEstimator.decode (reference module/tinyvc/encoder.py:61-67, with id2freq :48-54 as the uploaded table):
 * logits [B,512,T] -> f0 [B,1,T] (top-4 classes, softmax over their logits, expectation of the class frequencies,
 * <= 20 Hz -> 0).  tvc_encoder_f32 runs the same kernel on its own logits. */
int tvc_pitch_decode_f32(tvc_ctx* ctx, void* stream, const float* logits, float* f0, int B, int T);

/* kNN match ------------------------------------------------------------------------------- */
/* Prepare an index for matching, once per index: index [768, N] (the [1,768,N] tensor of index.pt,
 * reference extract_index.py:58 / infer.py:49, or Generator.encode's output) -> `prepared`, a blob of
 * tvc_knn_prepared_elems(N) floats: a 256-byte header, the raw vectors row-major (the final gather) and the vectors
 * scaled by 1/(||r||+1e-6) (feature_retrieval.py:25 recomputes that on every call) split into three bf16 parts per
 * value in MFMA lane order (the similarity GEMM's operand), plus the same vectors in fp16 and their inverse norms (the
 * coarse pass of the two-stage search): 12 bytes per index element.  The layout is private to the
 * library; the blob is self-describing (magic, kind, N, the raw vectors' |max| - the decoder's bound of the matched content -,
 * format version), so every entry point below takes either kind of blob.  A blob at an address this process did not prepare
 * (a copy, a blob read back from a file) has its header read once, on its first use outside a stream capture - the one place the
 * library waits for `stream` -, and is refused (TVC_ERR_ARG) if the magic, the format version or N do not match: blobs of an
 * earlier format must be prepared again. */
int64_t tvc_knn_prepared_elems(int64_t N);
int tvc_knn_prepare_index_f32(tvc_ctx* ctx, void* stream, const float* index, float* prepared,
                              int64_t N);
/* fp16 index storage for very large indices (BASELINE.json configs[4]: 1 M vectors): `rows_f16` =
 * [N, 768] IEEE binary16, one vector per row (index.pt's tensor transposed and cast to half) -> a blob of
 * tvc_knn_prepared_elems_f16(N) floats holding the fp16 vectors in MFMA lane order and one fp32 inverse norm per
 * vector: 2 bytes per index element (1.5 GB at N = 1 M).  Matching against it computes cos = dot(q_hat, r) / (||r||+1e-6)
 * with r the fp16 values (exactly two bf16 parts each), fp32 accumulation; the k = 4 rows averaged are the fp16 values.
 * Results equal the fp32-storage path run on the same (fp16-rounded) vectors up to fp32 rounding of the similarities. */
int64_t tvc_knn_prepared_elems_f16(int64_t N);
int tvc_knn_prepare_index_f16(tvc_ctx* ctx, void* stream, const void* rows_f16, float* prepared,
                              int64_t N);

/* The library remembers the N every blob was prepared with (by device address) and refuses calls that pass another N.  Call this
 * before the memory of a prepared blob is reused for anything else than a fresh tvc_knn_prepare_index_* (for instance a COPY of
 * another blob): an address that is recycled must not inherit the record.  (No reference counterpart: feature_retrieval.py:25-27
 * re-normalises the index on every call and keeps no prepared state.) */
int tvc_knn_forget(tvc_ctx* ctx, const float* prepared);
/* match_features(source, reference, k=4, alpha=0, metrics='cos')
 * (reference module/tinyvc/feature_retrieval.py:15-33): src [B,768,T] against one shared prepared
 * index -> out [B,768,T]; idx_out [B,T,4] int64 (nullable) = topk indices, ties -> lowest index. */
int tvc_knn_match_f32(tvc_ctx* ctx, void* stream, const float* src, const float* prepared, int64_t N,
                      float* out, int64_t* idx_out, int B, int T, void* ws, size_t ws_bytes);

/* match_features with the reference's full signature (module/tinyvc/feature_retrieval.py:15-33): the k = 1 ... 8 nearest index vectors of
 * every source frame under metric 0 = 'cos', 1 = 'IP' (inner product), 2 = 'L2' (negative Euclidean distance), out = the mean of the k RAW
 * vectors (alpha blending is the caller's one-liner).  src [B,768,T]; index [768,N] = the raw [1,768,N] tensor of index.pt (no prepared
 * blob: this is not the inference path - that asks for k = 4, 'cos' and runs tvc_knn_match_f32 - but every other argument the reference
 * accepts, in plain fp32 on the raw vectors); out [B,768,T]; idx_out [B,T,k] int64 and sim_out [B,T,k] (nullable) = torch.topk's indices
 * and values in rank order, equal similarities by lower index.  N >= k.  Workspace: B * T * k * 8 bytes + 4096 when idx_out is NULL. */
int tvc_knn_match_general_f32(tvc_ctx* ctx, void* stream, const float* src, const float* index, int64_t N, int k, int metric,
                              float* out, int64_t* idx_out, float* sim_out, int B, int T, void* ws, size_t ws_bytes);

/* pitch shift ----------------------------------------------------------------------------- */
/* Index-sharded variant of the match (a very large speaker index split over the GPUs of a node):
 * every rank holds a prepared shard and
 *   1. tvc_knn_topk_f32: this shard's top-4 per query: sims_out [B,T,4] (cosine similarity, descending, ties ->
 *      lower index first) and idx_out [B,T,4] (LOCAL indices into the shard);
 *   2. the host all-gathers (sim, global index) and keeps the global top-4 per query (tinyvc_amd/parallel.py);
 *   3. tvc_knn_gather_slots_f32: slots [nslots,768] <- this shard's raw rows for idx[nslots] (local index, or a
 *      negative value where the row lives on another rank -> zeros); summed over ranks every slot has exactly one
 *      contributor, so the all-reduce is exact;
 *   4. tvc_knn_finish_f32: out [B,768,T] = mean of the 4 slots of each query in the single-GPU order. */
int tvc_knn_topk_f32(tvc_ctx* ctx, void* stream, const float* src, const float* prepared, int64_t N,
                     float* sims_out, int64_t* idx_out, int B, int T, void* ws, size_t ws_bytes);
int tvc_knn_gather_slots_f32(tvc_ctx* ctx, void* stream, const float* prepared, int64_t N,
                             const int64_t* idx, float* slots, int64_t nslots);
int tvc_knn_finish_f32(tvc_ctx* ctx, void* stream, const float* slots, float* out, int B, int T);

/* module.utils.shift_frequency (reference module/utils/pitch_shift.py:5-15), n elements. */
int tvc_shift_frequency_f32(tvc_ctx* ctx, void* stream, const float* f0, float* out, int64_t n,
                            float semitones);

/* The affine map of the noise phases (reference module/tinyvc/decoder.py:78: `torch.rand(N, fft_bin, Lf) * 2 * math.pi - math.pi`,
 * three tensor ops): u [n] uniform draws in [0, 1) -> u * 2 * pi - pi with the same three fp32 roundings, in place, one launch.
 * The draw itself stays the caller's (torch's generator on the device, as in the reference). */
int tvc_noise_angle_from_uniform_f32(tvc_ctx* ctx, void* stream, float* u, int64_t n);

/* decoder --------------------------------------------------------------------------------- */
/* Decoder.infer (reference module/tinyvc/decoder.py:253-257): content [B,768,T], f0 [B,1,T],
 * energy [B,1,L], noise_angle [B,961,T] = the uniform phases of decoder.py:78 in [-pi,pi)
 * (NULL -> drawn on device from `seed`; not bit-comparable with torch's CPU generator)
 * -> wave [B, L]. */
int tvc_decoder_f32(tvc_ctx* ctx, void* stream, const float* content, const float* f0,
                    const float* energy, const float* noise_angle, uint64_t seed, float* wave,
                    int B, int T, void* ws, size_t ws_bytes);
/* Stage outputs of the decoder for parity tests (any pointer may be NULL):
 * amps [B,15,T], kernel [B,961,T] (SourceNet.forward, decoder.py:126-134),
 * source [B,16,L] (Decoder.dsp, decoder.py:259-266). Same arguments as tvc_decoder_f32.
 * `wave` may be NULL too: the call then stops behind the last stage asked for - amps / kernel only = SourceNet.forward
 * alone (no DSP, no FilterNet pass), + source = up to Decoder.dsp. */
int tvc_decoder_stages_f32(tvc_ctx* ctx, void* stream, const float* content, const float* f0,
                           const float* energy, const float* noise_angle, uint64_t seed,
                           float* wave, float* amps, float* kernel, float* source, int B, int T,
                           void* ws, size_t ws_bytes);

/* FilterNet.forward (reference module/tinyvc/decoder.py:222-233): content [B,768,T], f0 [B,1,T], energy [B,1,L],
 * source [B,16,L] (Decoder.dsp's output) -> wave [B, L].  For parity tests the block outputs can be copied out:
 * skips[i] (i = 0..4, any entry or the array itself may be NULL) = the outputs of `downs[i]` (decoder.py:227-229):
 * [B,24,L], [B,48,L/5], [B,96,L/20], [B,192,L/80], [B,384,L/240]; ups[i] (i = 0..3) = the outputs of `ups[i]`
 * (decoder.py:231-232): [B,192,2T], [B,96,6T], [B,48,24T], [B,24,96T].  ups[4] does not exist in this implementation:
 * its 1x1 (c5) is folded into output_layer's k7 conv when the weights are packed, and `wave` is what checks it. */
int tvc_filter_net_f32(tvc_ctx* ctx, void* stream, const float* content, const float* f0, const float* energy,
                       const float* source, float* wave, float* const* skips, float* const* ups, int B, int T,
                       void* ws, size_t ws_bytes);

/* Decoder.dsp (reference module/tinyvc/decoder.py:259-266): f0 [B,1,T], amps [B,15,T],
 * kernel [B,961,T], noise_angle as above -> source [B,16,L] (15 harmonics + filtered noise). */
int tvc_dsp_f32(tvc_ctx* ctx, void* stream, const float* f0, const float* amps, const float* kernel,
                const float* noise_angle, uint64_t seed, float* source, int B, int T, void* ws,
                size_t ws_bytes);

/* whole path ------------------------------------------------------------------------------ */
/* Generator.convert (reference module/infer/generator.py:26-34): wav [B, L] (already padded to
 * L % 480 == 0, autopad_waveform is a host-side zero pad) -> wave [B, L]. */
int tvc_convert_f32(tvc_ctx* ctx, void* stream, const float* wav, const float* prepared_index,
                    int64_t N, float pitch_shift, const float* noise_angle, uint64_t seed,
                    float* wave, int B, int64_t L, void* ws, size_t ws_bytes);

/* Generator.convert over a RAGGED batch (the reference converts the files of a directory one by one: infer.py:60-66; padding
 * them to a common length would change results - GRN and the phase scan run over the whole time axis, convnext.py:31-34).
 * wav / wave [B, Lmax] row-major, utterance b occupies its first lens[b] samples (lens[b] % 480 == 0, 960 < lens[b] <= Lmax; the
 * tail of a wave row is zero-filled); noise_angle [B, 961, Lmax/480] (utterance b uses its first lens[b]/480 frames) or NULL.
 * `lens` is a HOST array (lengths are launch geometry; they travel to the device as kernel arguments, asynchronously on `stream`).
 * The utterances share every kernel launch - the kernels take per-utterance lengths (csrc/ragged.h) -, in at most four batches per
 * call by length class (frames < 11, < 43, < 128, the rest: the kernels a FilterNet level runs depend on the utterance's length there),
 * one after the other on `stream`.  Every utterance gets exactly the samples a B = 1 tvc_convert_f32 call gives it for the same noise
 * phases.  With noise_angle = NULL the library draws them itself: the phase of (row b, bin, frame) is a hash of `seed` and of those three
 * numbers alone - independent of the other utterances, of their lengths and of the split into batches -, so row b of a ragged call equals row
 * b of an equal-length call with the same seed over its own frames, and row 0 the B = 1 call.  A single
 * utterance may be up to 80 000 frames.  Workspace: tvc_workspace_bytes_ragged. */
int tvc_workspace_bytes_ragged(tvc_ctx* ctx, int B, int64_t Lmax, const int64_t* lens, int64_t N, size_t* out_bytes);
/* Which utterances of a ragged call share their kernel launches: batch_of_row[b] = the in-kernel batch (0 .. *n_batches - 1, the order they
 * run in) that utterance b is converted in.  Pure host logic, no context, no device: the split tvc_convert_ragged_f32 makes (length classes
 * at 11 / 43 / 128 frames, at most `max_frames` frames per batch; 0 or anything above 80 000 = the default, 80 000 - pass what the context
 * that will convert was given by tvc_ctx_set_ragged_batch_frames).  Returns TVC_ERR_ARG for a length the ragged call would refuse. */
int tvc_ragged_plan(int B, int64_t Lmax, const int64_t* lens, int max_frames, int32_t* batch_of_row, int* n_batches);
/* Frames per in-kernel batch of THIS context's ragged calls (0 or anything above 80 000 = the default, 80 000): a smaller cap cuts a length
 * class into several batches, one after the other.  Results do not depend on it (every utterance equals its B = 1 conversion);
 * tvc_workspace_bytes_ragged and tvc_convert_ragged_f32 of the context follow it, so set it between calls (one host thread per ctx at a time).
 * (The tests use it to reach the several-batches-per-class path with small inputs; there is no environment variable and no process-wide state behind it.) */
int tvc_ctx_set_ragged_batch_frames(tvc_ctx* ctx, int max_frames);
int tvc_convert_ragged_f32(tvc_ctx* ctx, void* stream, const float* wav, int64_t Lmax, const int64_t* lens, const float* prepared_index,
                           int64_t N, float pitch_shift, const float* noise_angle, uint64_t seed, float* wave, int B, void* ws,
                           size_t ws_bytes);

/* streaming tail -------------------------------------------------------------------------- */
/* StreamInfer.audio_callback after convert (reference module/infer/stream.py:74-95), batched over
 * S streams: y [S, Ly] converted buffers; sola_buf [S,1920] in/out; fade_in [1920] = the sin^2
 * window init_buffer builds (stream.py:61; fade_out = 1 - fade_in); out [S, block];
 * shift_out [S] int32 (nullable) = chosen SOLA lag.  block = 1920 in the reference;
 * crossfade = search = 1920, delay = 3840 (stream.py:48-50).  use_phase_vocoder selects
 * phase_vocoder() (stream.py:9-26) instead of the sin^2 cross-fade. */
int tvc_sola_f32(tvc_ctx* ctx, void* stream, const float* y, float* sola_buf, const float* fade_in,
                 float* out, int32_t* shift_out, int S, int64_t Ly, int block,
                 int use_phase_vocoder);

/* StreamInfer.audio_callback before convert (reference module/infer/stream.py:69-70: `input_wav = torch.roll(input_wav, -block)`,
 * `input_wav[-block:] = block`), batched over S streams, in place and in one launch: buf [S, n] rolling input buffers,
 * blocks [S, block] the new blocks; afterwards buf[s] = (old buf[s][block:], blocks[s]).  n <= 32 768. */
int tvc_stream_push_f32(tvc_ctx* ctx, void* stream, float* buf, const float* blocks, int S, int64_t n, int block);

/* measurement ----------------------------------------------------------------------------- */
/* on = 1: every stage (and every FilterNet block) is bracketed by a hipEvent pair on the launch stream (19 pairs per
 * convert, ~0.1 ms of a 8.4 ms step); on = 2: only the `filter_net` region (the roofline's kernel group); on = 0: off.
 * tvc_profile_read() synchronises those events and writes "region=milliseconds;..." (summed per region name since the
 * previous read) into buf. */
int tvc_profile_enable(tvc_ctx* ctx, int on);
int tvc_profile_read(tvc_ctx* ctx, char* buf, size_t buf_bytes);

#ifdef __cplusplus
}
#endif
#endif /* TINYVC_HIP_H */
