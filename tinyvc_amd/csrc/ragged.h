// Ragged batches inside the kernels (infer.py:60-66 converts a directory's files one by one; here they are ONE batch whose utterances
// keep their own lengths): GRN norms (convnext.py:31-34), replicate / reflect padding, the oscillator's phase scan and the iSTFT envelope
// all end at every utterance's true end, so each utterance gets exactly the samples its own B = 1 conversion produces.
//
// Layout of a ragged batch: every tensor is ONE long "utterance" - [C][S] with S = the sum of the utterances' lengths at that tensor's
// rate, utterance b occupying columns [pre[b] * mult, (pre[b] + tb[b]) * mult) (tb = its frames, mult = samples per frame at that rate:
// 480, 96, 24, 6, 2, 1).  The drivers (run_encoder, run_filter, ...) are called with B = 1 and T = sum(tb): their allocations, strides
// and the column-independent kernels (1x1 GEMMs, kNN, pitch decode) are then right as they are; kernels whose arithmetic couples columns
// get this view and treat `len` as the ROW STRIDE while the valid extent of a tile's utterance comes from tb[].
//   * time-tiled persistent kernels walk a per-launch prefix table of column tiles (ts[b] = tiles of the utterances before b), so no
//     tile is dead and a workgroup still meets one or two utterances (the |max| slots stay per utterance);
//   * flat GEMMs look the utterance of a column up in col2b[] for the per-utterance scalars only (|max| slots, GRN factors).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

struct tvc_ctx;

namespace tvc {

struct RagDev {                   // device view handed to a kernel (tb == nullptr: equal lengths, the kernel's ordinary path)
    const int* tb = nullptr;      // [B] frames of utterance b
    const int* pre = nullptr;     // [B + 1] exclusive prefix of tb
    const int* col2b = nullptr;   // [pre[B]] frame column -> utterance
    const int* ts = nullptr;      // [B + 1] this launch's column-tile prefix (time-tiled kernels only)
    const int* row = nullptr;     // [B] row of utterance b in the caller's padded [rows][Lmax] tensors (boundary kernels only)
    int B = 0, mult = 0;          // mult: samples per frame at this launch's rate
    int Tmax = 0;                 // frames of the caller's padded rows
};

struct RagHost {                  // one ragged (sub-)batch of a call; lives for the duration of the call
    int B = 0, Ttot = 0, Tmax = 0, Tlong = 0, Tshort = 0;     // Tlong / Tshort = the longest / shortest utterance's frames
    std::vector<int> tb, pre, row;
    const int* d_tb = nullptr;
    const int* d_pre = nullptr;
    const int* d_col2b = nullptr;
    const int* d_row = nullptr;
    int* d_pool = nullptr;        // tile tables, (B + 1) ints each
    int pool_slots = 0;
    struct Tab {
        int mult, bn, total;
        const int* d;
    };
    std::vector<Tab> tabs;
};

constexpr int kRagTabSlots = 24;      // distinct (rate, tile width) pairs a conversion launches with (13 today)

// ints of device scratch a ragged batch of B utterances and Ttot frames needs (tb, pre, row, col2b, the tile tables)
inline size_t rag_scratch_ints(int B, int Ttot) { return (size_t)3 * (B + 1) + (size_t)Ttot + (size_t)kRagTabSlots * (B + 1) + 64; }

// fills `h` from the host lengths (frames), uploads the tables into `scratch` (rag_scratch_ints ints) on stream s
int rag_setup(tvc_ctx* ctx, hipStream_t s, bool dry, RagHost& h, const std::vector<int>& frames, const std::vector<int>& rows, int Tmax, int* scratch);
// the view of the context's current ragged batch for a launch at `mult` samples per frame; bn > 0: with the column-tile table of
// bn-wide tiles (built on first use), *ntiles = its total.  Equal-length calls (no current batch) get the empty view.
int rag_view(tvc_ctx* ctx, hipStream_t s, int mult, int bn, RagDev* out, int* ntiles);

// Shape-dependent kernel choices (decoder.hip: film_s2 needs one 256-column tile per utterance) look at the SHORTEST utterance of a ragged batch:
// a batch only holds utterances that make the same choices (api.hip ragged_split), so this is every member's own decision.
int rag_min_len(const tvc_ctx* ctx, int len);

// ---- device side -------------------------------------------------------------------------------------------------------------
// utterance of column tile ct: ts[b] <= ct < ts[b + 1]; `hint` = the previous tile's utterance (a persistent walk only moves forward)
__device__ __forceinline__ int rag_find(const int* __restrict__ ts, int B, int ct, int hint) {
    int b = hint;
    if (ct >= ts[b]) {
        if (ct < ts[b + 1]) return b;
        if (b + 2 <= B && ct < ts[b + 2]) return b + 1;
    }
    int lo = 0, hi = B - 1;           // last b with ts[b] <= ct
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (ts[mid] <= ct) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// (utterance, tile index inside it, its length and first column at the launch's rate) of column tile `tile`; equal-length launches:
// tiles_per_utt tiles per utterance of `len` columns (off = 0: those kernels address utterance b as b * C * len)
struct RagTile {
    int b, tin, len, off;
};
template <bool RAG>
__device__ __forceinline__ RagTile rag_tile(const RagDev& r, int tile, int tiles_per_utt, int len, int hint) {
    RagTile t;
    if constexpr (RAG) {
        t.b = rag_find(r.ts, r.B, tile, hint);
        t.tin = tile - r.ts[t.b];
        t.len = r.tb[t.b] * r.mult;
        t.off = r.pre[t.b] * r.mult;
    } else {
        t.b = tile / tiles_per_utt;
        t.tin = tile - t.b * tiles_per_utt;
        t.len = len;
        t.off = 0;
    }
    return t;
}

}  // namespace tvc
