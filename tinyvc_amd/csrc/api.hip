// libtinyvc_hip.so — context, checkpoint packing, workspace sizing and the extern "C" surface.
#include <atomic>
#include <cmath>
#include <functional>
#include <mutex>

#include <cstdlib>

#include "tvc_common.h"

using namespace tvc;

#ifndef TVC_SIDE_PRIO_EXPR
#define TVC_SIDE_PRIO_EXPR prio_least
#endif
namespace {

struct ArenaBuilder {
    std::vector<float> buf;
    size_t put(const std::vector<float>& v) {
        size_t off = (buf.size() + 63) & ~size_t(63);  // 256-byte aligned
        buf.resize(off + v.size(), 0.f);
        std::copy(v.begin(), v.end(), buf.begin() + off);
        return off;
    }
};

// Offsets are resolved to device pointers after the single upload.
struct Fixup {
    const float** slot;
    size_t off;
};

int pad_m(int M) {
    if (M <= 32) return 32;
    if (M <= 64) return 64;
    if (M <= 96) return 96;
    if (M % 96 == 0 && M % 128 != 0) return M;
    return (M + 127) / 128 * 128;
}

struct Packer {
    tvc_ctx* ctx;
    ArenaBuilder ab;
    std::vector<Fixup> fix;
    std::string missing;

    const HostTensor* find(const std::string& key) {
        auto it = ctx->host.find(key);
        if (it == ctx->host.end()) {
            if (missing.empty()) missing = key;
            return nullptr;
        }
        return &it->second;
    }
    void raw(const std::string& key, const float** slot, size_t expect) {
        const HostTensor* t = find(key);
        if (!t) return;
        if (t->data.size() != expect) {
            if (missing.empty()) missing = key + " (wrong size)";
            return;
        }
        fix.push_back({slot, ab.put(t->data)});
    }
    // ---- two-part fp16 split of the packed weights (conv3s.h): w = (h1 + 2^-11 h2) * 2^e, e per 32-row m-tile -------------------
    static uint16_t f16_bits(float f) {
        const _Float16 h = (_Float16)f;          // round to nearest even, subnormals kept
        uint16_t u;
        std::memcpy(&u, &h, 2);
        return u;
    }
    static float f16_value(uint16_t u) {
        _Float16 h;
        std::memcpy(&h, &u, 2);
        return (float)h;
    }
    // the two parts of w / scale (scale = a power of two: the division is exact)
    static void split2(float w, float scale, uint16_t* h1, uint16_t* h2) {
        const float x = w / scale;
        *h1 = f16_bits(x);
        *h2 = f16_bits((x - f16_value(*h1)) * 2048.f);
    }
    // power of two that brings `amax` into [1, 2) (1 for an all-zero tile)
    static float pow2_scale(float amax) {
        if (!(amax > 0.f) || !std::isfinite(amax)) return 1.f;
        int e;
        std::frexp(amax, &e);                    // amax = m * 2^e, m in [0.5, 1)
        return std::ldexp(1.f, e - 1);
    }
    std::map<const PackedW*, std::vector<float>> host_wscale;      // the scales chosen for every packed image (joint packing, fused-block blobs)

    // Stack one or more conv weights [cout_i][cin][taps] along cout into the host staging layout At[k][m], k = ci*taps + tap
    // (zero-padded to Kpad x Mpad) and the bias row [Mpad].
    bool stage(const std::vector<std::string>& names, PackedW* pw, int cin, int taps, std::vector<float>* At, std::vector<float>* bias, int* group_rows) {
        int M = 0;
        std::vector<const HostTensor*> ws, bs;
        for (auto& n : names) {
            const HostTensor* w = find(n + ".weight");
            const HostTensor* b = find(n + ".bias");
            if (!w || !b) return false;
            if (w->shape.size() != 3 || w->shape[1] != cin || w->shape[2] != taps ||
                (int64_t)b->data.size() != w->shape[0]) {
                if (missing.empty()) missing = n + " (unexpected shape)";
                return false;
            }
            ws.push_back(w);
            bs.push_back(b);
            M += (int)w->shape[0];
        }
        pw->M = M;
        pw->K = cin * taps;
        pw->cin = cin;
        pw->taps = taps;
        pw->Mpad = pad_m(M);
        pw->Kpad = (pw->K + 15) / 16 * 16;
        At->assign((size_t)pw->Kpad * pw->Mpad, 0.f);
        bias->assign(pw->Mpad, 0.f);
        int m0 = 0;
        for (size_t i = 0; i < ws.size(); ++i) {
            int cout = (int)ws[i]->shape[0];
            for (int m = 0; m < cout; ++m) {
                (*bias)[m0 + m] = bs[i]->data[m];
                for (int k = 0; k < pw->K; ++k) (*At)[(size_t)k * pw->Mpad + m0 + m] = ws[i]->data[(size_t)m * pw->K + k];
            }
            m0 += cout;
        }
        bool equal_groups = ws.size() > 1;
        for (auto* w : ws) equal_groups = equal_groups && w->shape[0] == ws[0]->shape[0];
        *group_rows = equal_groups ? (int)ws[0]->shape[0] : 0;
        return true;
    }
    // image geometry: group_rows > 0 = the M rows are `M / group_rows` stacked groups (FiLM scale ; shift), each padded to whole 32-row tiles
    static int image_mt(const PackedW* pw, int group_rows) {
        const int gp = group_rows > 0 ? (group_rows + 31) / 32 * 32 : 0;
        return group_rows > 0 ? (pw->M / group_rows) * gp / 32 : pw->Mpad / 32;
    }
    static int image_row(const PackedW* pw, int group_rows, int m) {      // staged row of image row m (pw->M = a padding row)
        if (group_rows <= 0) return m < pw->M ? m : pw->M;
        const int gp = (group_rows + 31) / 32 * 32, g = m / gp, mi = m - g * gp;
        return mi < group_rows ? g * group_rows + mi : pw->M;
    }
    std::vector<float> mt_amax(const PackedW* pw, const std::vector<float>& At, int group_rows) {
        const int MT = image_mt(pw, group_rows);
        std::vector<float> amax(MT, 0.f);
        for (int mt = 0; mt < MT; ++mt)
            for (int r = 0; r < 32; ++r) {
                const int m = image_row(pw, group_rows, mt * 32 + r);
                if (m >= pw->M) continue;
                for (int k = 0; k < pw->K; ++k) amax[mt] = std::max(amax[mt], std::fabs(At[(size_t)k * pw->Mpad + m]));
            }
        return amax;
    }
    void conv(const std::vector<std::string>& names, PackedW* pw, int cin, int taps) {
        std::vector<float> At, bias;
        int group_rows = 0;
        if (!stage(names, pw, cin, taps, &At, &bias, &group_rows)) return;
        // only the split image goes to the device: `At` is the host-side staging layout it is built from
        fix.push_back({&pw->bias, ab.put(bias)});
        std::vector<float> sc = mt_amax(pw, At, group_rows);
        for (auto& v : sc) v = pow2_scale(v);
        a6(pw, At, group_rows, sc);
    }
    // two convs whose results are accumulated into ONE tile (Downsample: c3(h2) + down_res(xi)): the same per-m-tile scales for both
    void conv_joint(const std::string& na, PackedW* pa, int cin_a, int taps_a, const std::string& nb, PackedW* pb, int cin_b, int taps_b) {
        std::vector<float> Aa, ba, Ab, bb;
        int ga = 0, gb = 0;
        if (!stage({na}, pa, cin_a, taps_a, &Aa, &ba, &ga) || !stage({nb}, pb, cin_b, taps_b, &Ab, &bb, &gb)) return;
        if (pa->Mpad != pb->Mpad) {
            if (missing.empty()) missing = na + " / " + nb + " (row counts differ)";
            return;
        }
        fix.push_back({&pa->bias, ab.put(ba)});
        fix.push_back({&pb->bias, ab.put(bb)});
        std::vector<float> sa = mt_amax(pa, Aa, 0), sb = mt_amax(pb, Ab, 0);
        for (size_t i = 0; i < sa.size(); ++i) sa[i] = pow2_scale(std::max(sa[i], sb[i]));
        const size_t off_a = a6(pa, Aa, 0, sa);
        a6(pb, Ab, 0, sa);
        fix.push_back({&pb->wjoint, off_a});      // resolves to pa->A6: the launch checks that the pair was packed together
    }
    // two-part fp16 image of At for conv3s.h: [step = slab*taps + tap][m-tile][part][lane][8 fp16],
    // lane -> row m = 32*mt + (lane & 31), channel ci = 16*slab + 8*(lane >> 5) + j.  Returns the image's arena offset.
    size_t a6(PackedW* pw, const std::vector<float>& At, int group_rows, const std::vector<float>& scale) {
        const int taps = pw->taps, cin = pw->cin, nslab = ((cin + 15) / 16 + 5) / 6 * 6;   // zero slabs up to a multiple of 6: any slab depth divides
        const int MT = image_mt(pw, group_rows);
        std::vector<float> img((size_t)nslab * taps * MT * 2 * 256, 0.f);
        uint16_t* o = reinterpret_cast<uint16_t*>(img.data());
        for (int s = 0; s < nslab; ++s)
            for (int tap = 0; tap < taps; ++tap)
                for (int mt = 0; mt < MT; ++mt)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int ci = s * 16 + 8 * (lane >> 5) + j, m = image_row(pw, group_rows, mt * 32 + (lane & 31));
                            const float w = (ci < cin && m < pw->M) ? At[(size_t)(ci * taps + tap) * pw->Mpad + m] : 0.f;
                            const size_t base = (((size_t)(s * taps + tap) * MT + mt) * 2 * 64 + lane) * 8 + j;
                            split2(w, scale[mt], &o[base], &o[base + 512]);
                        }
        pw->MT6 = MT;
        pw->S6 = nslab;
        const size_t off = ab.put(img);
        fix.push_back({&pw->A6, off});
        fix.push_back({&pw->wscale, ab.put(scale)});
        host_wscale[pw] = scale;
        return off;
    }
    // conv (k3) + FiLM (two 1x1s) of one Upsample half for film_s2.h.  Image: [96-row block][16-channel slab][30 pieces][lane][8 fp16],
    // piece q < 18: conv tap q / 6, m-tile (q % 6) / 2 of the block, part q % 2; q >= 18: to_scale (q < 24) / to_shift, m-tile, part.
    // Lane order as in a6 (row = lane & 31, channel = 16 slab + 8 (lane >> 5) + j).  This kernel adds all three part products into ONE
    // accumulator, so the second part is the UNSCALED fp16 residual and every m-tile is normalised to |max| in [2^13, 2^14): the
    // residual's absolute fp16 resolution (2^-24, subnormals kept) is then 2^-37 of the tile's largest weight.
    void film_u(FilmU* fu, const std::string& conv_name, const std::string& film, int C, const std::string& first_conv) {
        const HostTensor* w = find(conv_name + ".weight");
        const HostTensor* b = find(conv_name + ".bias");
        const HostTensor* wsc = find(film + ".to_scale.weight");
        const HostTensor* bsc = find(film + ".to_scale.bias");
        const HostTensor* wsh = find(film + ".to_shift.weight");
        const HostTensor* bsh = find(film + ".to_shift.bias");
        if (!w || !b || !wsc || !bsc || !wsh || !bsh) return;
        if (C % 96 != 0 || w->data.size() != (size_t)C * C * 3 || wsc->data.size() != (size_t)C * C || wsh->data.size() != (size_t)C * C ||
            b->data.size() != (size_t)C || bsc->data.size() != (size_t)C || bsh->data.size() != (size_t)C) {
            if (missing.empty()) missing = conv_name + " / " + film + " (unexpected shape)";
            return;
        }
        const int MT = C / 32, nslab = C / 16, mblocks = C / 96;
        auto tile_scale = [&](const std::vector<float>& wt, int per_row, int mt) {
            float amax = 0.f;
            for (int r = 0; r < 32; ++r)
                for (int k = 0; k < per_row; ++k) amax = std::max(amax, std::fabs(wt[(size_t)(mt * 32 + r) * per_row + k]));
            return pow2_scale(amax) * (1.f / 8192.f);
        };
        std::vector<float> tab((size_t)6 * C);
        std::vector<float> s_conv(MT), s_sc(MT), s_sh(MT);
        for (int mt = 0; mt < MT; ++mt) {
            s_conv[mt] = tile_scale(w->data, 3 * C, mt);
            s_sc[mt] = tile_scale(wsc->data, C, mt);
            s_sh[mt] = tile_scale(wsh->data, C, mt);
        }
        for (int m = 0; m < C; ++m) {
            tab[m] = b->data[m];
            tab[(size_t)C + m] = s_conv[m / 32];
            tab[(size_t)2 * C + m] = bsc->data[m];
            tab[(size_t)3 * C + m] = bsh->data[m];
            tab[(size_t)4 * C + m] = s_sc[m / 32];
            tab[(size_t)5 * C + m] = s_sh[m / 32];
        }
        std::vector<float> img((size_t)mblocks * nslab * 30 * 64 * 4, 0.f);
        uint16_t* o = reinterpret_cast<uint16_t*>(img.data());
        for (int mb = 0; mb < mblocks; ++mb)
            for (int s = 0; s < nslab; ++s)
                for (int q = 0; q < 30; ++q) {
                    const bool conv = q < 18;
                    const int qq = conv ? q : q - 18, grp = qq / 6, mi = (qq % 6) / 2, part = qq % 2, mt = mb * 3 + mi;
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int m = mt * 32 + (lane & 31), ci = s * 16 + 8 * (lane >> 5) + j;
                            float x;
                            if (conv) x = w->data[((size_t)m * C + ci) * 3 + grp] / s_conv[mt];
                            else if (grp == 0) x = wsc->data[(size_t)m * C + ci] / s_sc[mt];
                            else x = wsh->data[(size_t)m * C + ci] / s_sh[mt];
                            const uint16_t h1 = f16_bits(x);
                            o[((((size_t)mb * nslab + s) * 30 + q) * 64 + lane) * 8 + j] = part == 0 ? h1 : f16_bits(x - f16_value(h1));
                        }
                }
        // the bound the half's first conv normalises its pre-split output by (conv_s2.h PRE): max_m sum_{k, tap} |w[m][k][tap]| and max |b|, a
        // hair above in float so that rounding cannot undercut them
        if (const HostTensor* w1 = find(first_conv + ".weight")) {
            const HostTensor* b1 = find(first_conv + ".bias");
            if (b1 && w1->data.size() == (size_t)C * C * 3 && b1->data.size() == (size_t)C) {
                double wl1 = 0.0, bm = 0.0;
                for (int m = 0; m < C; ++m) {
                    double sum = 0.0;
                    for (int k = 0; k < 3 * C; ++k) sum += std::fabs((double)w1->data[(size_t)m * 3 * C + k]);
                    wl1 = std::max(wl1, sum);
                    bm = std::max(bm, std::fabs((double)b1->data[m]));
                }
                fu->hb_w = (float)(wl1 * 1.0001);
                fu->hb_b = (float)(bm * 1.0001);
            }
        }
        fu->C = C;
        fix.push_back({&fu->img, ab.put(img)});
        fix.push_back({&fu->tab, ab.put(tab)});
    }
    // Weight blob of one half of the fused ups.4 kernel (filter_up24s.hip): 28 pieces of 1 KiB in
    // v_mfma_f32_32x32x16_f16 A-lane order (row m = lane & 31, k = 8 * (lane >> 5) + j), two fp16 parts each:
    //   [conv a: 5 steps][2 parts] [conv b: 5 steps][2 parts] [FiLM: 2 steps][to_scale, to_shift][2 parts]
    // followed by 304 floats: biases a, b, scale, shift (32 each), the folded output taps [24][7] and their bias [296], then
    // [297..300] the power-of-two scales of conv a, conv b, to_scale, to_shift, [301] max_m sum_k |w_a[m][k]| and [302] max |b_a|
    // (the bound of the block's on-chip intermediate, see the kernel).
    // K runs in units of (tap, 8-channel group): unit u = 2 * step + (lane >> 5), tap = u / 3, group = u % 3 (a 24-channel
    // conv has 9 units, the 10th is zero; FiLM's 1x1 has 3).
    // Second half: Upsample.c5 (1x1, decoder.py:171,189) and FilterNet.output_layer (k7, decoder.py:220,233) have nothing
    // between them, so they are one k7 conv 24 -> 1: w75[c][j] = sum_m w7[m][j] w5[m][c], b75 = b7 + sum_{m,j} w7[m][j] b5[m]
    // (replicate padding commutes with the 1x1), accumulated in double.
    void up24s_half(const float** slot, const std::string& ca, const std::string& cb, const std::string& film, const std::string& c5,
                    const std::string& out7) {
        const HostTensor* wa = find(ca + ".weight");
        const HostTensor* ba = find(ca + ".bias");
        const HostTensor* wb = find(cb + ".weight");
        const HostTensor* bb = find(cb + ".bias");
        const HostTensor* wsc = find(film + ".to_scale.weight");
        const HostTensor* bsc = find(film + ".to_scale.bias");
        const HostTensor* wsh = find(film + ".to_shift.weight");
        const HostTensor* bsh = find(film + ".to_shift.bias");
        if (!wa || !ba || !wb || !bb || !wsc || !bsc || !wsh || !bsh) return;
        const int C = 24;
        if (wa->data.size() != (size_t)C * C * 3 || wb->data.size() != (size_t)C * C * 3 || wsc->data.size() != (size_t)C * C ||
            wsh->data.size() != (size_t)C * C) {
            if (missing.empty()) missing = ca + " (unexpected shape for the 24-channel block)";
            return;
        }
        std::vector<float> img(28 * 256 + 304, 0.f);
        uint16_t* o = reinterpret_cast<uint16_t*>(img.data());
        auto amax_of = [](const HostTensor* t) {
            float a = 0.f;
            for (float v : t->data) a = std::max(a, std::fabs(v));
            return a;
        };
        const float sa = pow2_scale(amax_of(wa)), sb = pow2_scale(amax_of(wb)), ssc = pow2_scale(amax_of(wsc)), ssh = pow2_scale(amax_of(wsh));
        auto put2 = [&](int piece0, int lane, int j, float w, float scale) {   // parts of one value into pieces piece0, +1
            const size_t base = ((size_t)piece0 * 64 + lane) * 8 + j;
            split2(w, scale, &o[base], &o[base + 512]);
        };
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int m = lane & 31, lh = lane >> 5;
                for (int s = 0; s < 5; ++s) {
                    const int u = 2 * s + lh, tap = u / 3, ci = 8 * (u % 3) + j;
                    const bool real = u < 9 && m < C;
                    put2(s * 2, lane, j, real ? wa->data[((size_t)m * C + ci) * 3 + tap] : 0.f, sa);
                    put2(10 + s * 2, lane, j, real ? wb->data[((size_t)m * C + ci) * 3 + tap] : 0.f, sb);
                }
                for (int s = 0; s < 2; ++s) {
                    const int u = 2 * s + lh, ci = 8 * u + j;
                    const bool real = u < 3 && m < C;
                    put2(20 + (s * 2 + 0) * 2, lane, j, real ? wsc->data[(size_t)m * C + ci] : 0.f, ssc);
                    put2(20 + (s * 2 + 1) * 2, lane, j, real ? wsh->data[(size_t)m * C + ci] : 0.f, ssh);
                }
            }
        float* fl = img.data() + 28 * 256;
        double l1max = 0.0;
        float bamax = 0.f;
        for (int m = 0; m < C; ++m) {
            fl[m] = ba->data[m];
            fl[32 + m] = bb->data[m];
            fl[64 + m] = bsc->data[m];
            fl[96 + m] = bsh->data[m];
            double l1 = 0.0;
            for (int k = 0; k < C * 3; ++k) l1 += std::fabs((double)wa->data[(size_t)m * C * 3 + k]);
            l1max = std::max(l1max, l1);
            bamax = std::max(bamax, std::fabs(ba->data[m]));
        }
        fl[297] = sa;
        fl[298] = sb;
        fl[299] = ssc;
        fl[300] = ssh;
        fl[301] = (float)(l1max * 1.0000002);      // rounded up: it is a bound
        fl[302] = bamax;
        if (!c5.empty()) {
            const HostTensor* w5 = find(c5 + ".weight");
            const HostTensor* b5 = find(c5 + ".bias");
            const HostTensor* w7 = find(out7 + ".weight");
            const HostTensor* b7 = find(out7 + ".bias");
            if (!w5 || !b5 || !w7 || !b7) return;
            if (w5->data.size() != (size_t)C * C || w7->data.size() != (size_t)C * 7 || b7->data.size() != 1) {
                if (missing.empty()) missing = c5 + " (unexpected shape for the folded output conv)";
                return;
            }
            double bias = b7->data[0];
            for (int c = 0; c < C; ++c)
                for (int j = 0; j < 7; ++j) {
                    double acc = 0.0;
                    for (int m = 0; m < C; ++m) acc += (double)w7->data[(size_t)m * 7 + j] * (double)w5->data[(size_t)m * C + c];
                    fl[128 + c * 7 + j] = (float)acc;
                }
            for (int m = 0; m < C; ++m)
                for (int j = 0; j < 7; ++j) bias += (double)w7->data[(size_t)m * 7 + j] * (double)b5->data[m];
            fl[128 + 168] = (float)bias;
        }
        fix.push_back({slot, ab.put(img)});
    }
    // Weight blob of the downs.0 kernel (filter_up24s.hip): the 17 -> 24 k3 conv in the same 10-piece layout
    // as a 24-channel conv (input rows 17..23 zero), then 32 floats: bias [24], [31] = the image's power-of-two scale.
    void down0s(const float** slot, const std::string& name, float* bound_w, float* bound_b) {
        const HostTensor* w = find(name + ".weight");
        const HostTensor* b = find(name + ".bias");
        if (!w || !b) return;
        const int C = 24, CI = 17;
        if (w->data.size() != (size_t)C * CI * 3 || b->data.size() != (size_t)C) {
            if (missing.empty()) missing = name + " (unexpected shape for the split-precision downs.0 blob)";
            return;
        }
        std::vector<float> img(10 * 256 + 32, 0.f);
        uint16_t* o = reinterpret_cast<uint16_t*>(img.data());
        float amax = 0.f;
        for (float v : w->data) amax = std::max(amax, std::fabs(v));
        const float sc = pow2_scale(amax);
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j)
                for (int s = 0; s < 5; ++s) {
                    const int m = lane & 31, u = 2 * s + (lane >> 5), tap = u / 3, ci = 8 * (u % 3) + j;
                    const float v = (u < 9 && m < C && ci < CI) ? w->data[((size_t)m * CI + ci) * 3 + tap] : 0.f;
                    const size_t base = ((size_t)(s * 2) * 64 + lane) * 8 + j;
                    split2(v, sc, &o[base], &o[base + 512]);
                }
        double l1max = 0.0;
        float bmax = 0.f;
        for (int m = 0; m < C; ++m) {
            img[10 * 256 + m] = b->data[m];
            double l1 = 0.0;
            for (int k = 0; k < CI * 3; ++k) l1 += std::fabs((double)w->data[(size_t)m * CI * 3 + k]);
            l1max = std::max(l1max, l1);
            bmax = std::max(bmax, std::fabs(b->data[m]));
        }
        // |out| <= l1max |x|max + bmax (rounded up a little: the bound must hold for the fp32-rounded sums too): the scale of the pre-split planes
        img[10 * 256 + 29] = *bound_w = (float)(l1max * 1.0001);
        img[10 * 256 + 30] = *bound_b = bmax * 1.0001f;
        img[10 * 256 + 31] = sc;
        fix.push_back({slot, ab.put(img)});
    }
    // Weight blob of one 24-input-channel k3 conv for down24f_kernel (filter_up24s.hip): pieces [step][m-tile][part]
    // (same (tap, group) K order as up24s_half), then 64 floats: bias [M <= 48], [62], [63] = the power-of-two scales of the (at
    // most two) m-tiles.  M = 24 (one m-tile) or 48 (two).  `joint`: take the scales of this already packed image instead of the
    // weight's own (c3 of the 24-channel Downsample block is accumulated with down_res into one tile: conv_joint).
    void conv24s(const float** slot, const std::string& name, int M, const std::string& extra_bias = "", const PackedW* joint = nullptr) {
        const HostTensor* w = find(name + ".weight");
        const HostTensor* b = find(name + ".bias");
        const HostTensor* eb = extra_bias.empty() ? nullptr : find(extra_bias);
        if (!extra_bias.empty() && (!eb || eb->data.size() != (size_t)M)) {
            if (missing.empty()) missing = extra_bias + " (wrong size)";
            return;
        }
        if (!w || !b) return;
        const int CI = 24, MT = (M + 31) / 32;
        if (w->data.size() != (size_t)M * CI * 3 || b->data.size() != (size_t)M) {
            if (missing.empty()) missing = name + " (unexpected shape for the 24-channel split-precision blob)";
            return;
        }
        std::vector<float> img((size_t)10 * MT * 256 + 64, 0.f);
        uint16_t* o = reinterpret_cast<uint16_t*>(img.data());
        std::vector<float> sc(MT, 1.f);
        if (joint) {
            auto it = host_wscale.find(joint);
            if (it == host_wscale.end() || (int)it->second.size() != MT) {
                if (missing.empty()) missing = name + " (its joint image is not packed yet)";
                return;
            }
            sc = it->second;
        } else {
            for (int mt = 0; mt < MT; ++mt) {
                float amax = 0.f;
                for (int m = 32 * mt; m < std::min(M, 32 * mt + 32); ++m)
                    for (int k = 0; k < CI * 3; ++k) amax = std::max(amax, std::fabs(w->data[(size_t)m * CI * 3 + k]));
                sc[mt] = pow2_scale(amax);
            }
        }
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j)
                for (int s = 0; s < 5; ++s)
                    for (int mt = 0; mt < MT; ++mt) {
                        const int m = 32 * mt + (lane & 31), u = 2 * s + (lane >> 5), tap = u / 3, ci = 8 * (u % 3) + j;
                        const float v = (u < 9 && m < M) ? w->data[((size_t)m * CI + ci) * 3 + tap] : 0.f;
                        const size_t base = ((size_t)((s * MT + mt) * 2) * 64 + lane) * 8 + j;
                        split2(v, sc[mt], &o[base], &o[base + 512]);
                    }
        for (int m = 0; m < M; ++m) img[(size_t)10 * MT * 256 + m] = b->data[m] + (eb ? eb->data[m] : 0.f);
        for (int mt = 0; mt < MT; ++mt) img[(size_t)10 * MT * 256 + 62 + mt] = sc[mt];
        fix.push_back({slot, ab.put(img)});
    }
    void convnext(const std::string& p, ConvNeXtW* w, int C, int dil) {
        w->C = C;
        w->dilation = dil;
        raw(p + ".c1.weight", &w->dw_w, (size_t)C * 7);
        raw(p + ".c1.bias", &w->dw_b, C);
        raw(p + ".norm.gamma", &w->ln_g, C);
        raw(p + ".norm.beta", &w->ln_b, C);
        {   // |LayerNorm output| <= sqrt(C - 1) max|gamma| + max|beta| whatever the data (a normalised column has |x_hat| <= sqrt(C - 1))
            const HostTensor* g = find(p + ".norm.gamma");
            const HostTensor* bt = find(p + ".norm.beta");
            float gm = 0.f, bm = 0.f;
            if (g) for (float v : g->data) gm = std::max(gm, std::fabs(v));
            if (bt) for (float v : bt->data) bm = std::max(bm, std::fabs(v));
            w->ln_bound = std::sqrt((float)C) * gm + bm;
        }
        conv({p + ".c2"}, &w->c2, C, 1);
        raw(p + ".grn.gamma", &w->grn_g, 2 * C);
        raw(p + ".grn.beta", &w->grn_b, 2 * C);
        conv({p + ".c3"}, &w->c3, 2 * C, 1);
        const HostTensor* w3 = find(p + ".c3.weight");
        const HostTensor* b3 = find(p + ".c3.bias");
        const HostTensor* gb = find(p + ".grn.beta");
        if (w3 && b3 && gb && gb->data.size() == (size_t)2 * C && w3->data.size() == (size_t)C * 2 * C) {
            std::vector<float> fb(w->c3.Mpad, 0.f);
            for (int m = 0; m < C; ++m) {
                double acc = b3->data[m];
                for (int k = 0; k < 2 * C; ++k) acc += (double)w3->data[(size_t)m * 2 * C + k] * (double)gb->data[k];
                fb[m] = (float)acc;
            }
            fix.push_back({&w->c3_bias_grn, ab.put(fb)});
        }
    }
};

// Tables of the wave-level 1920-point FFTs (fft.hip), computed in fp64: (cos, sin)(2 pi j / 960), (cos, sin)(2 pi k / 1920),
// periodic Hann window.
void build_fft_tables(Packer& pk, tvc_ctx* ctx) {
    const int N = kNfft;
    const double two_pi = 6.283185307179586476925286766559;
    std::vector<float> t960(2 * 960), t1920(2 * 961 + 2), hann(N);
    for (int j = 0; j < 960; ++j) {
        t960[2 * j] = (float)std::cos(two_pi * j / 960.0);
        t960[2 * j + 1] = (float)std::sin(two_pi * j / 960.0);
    }
    for (int k = 0; k <= 960; ++k) {
        t1920[2 * k] = (float)std::cos(two_pi * k / 1920.0);
        t1920[2 * k + 1] = (float)std::sin(two_pi * k / 1920.0);
    }
    for (int n = 0; n < N; ++n) hann[n] = (float)(0.5 - 0.5 * std::cos(two_pi * n / N));
    pk.fix.push_back({&ctx->fft_tw960, pk.ab.put(t960)});
    pk.fix.push_back({&ctx->fft_tw1920, pk.ab.put(t1920)});
    pk.fix.push_back({&ctx->fft_hann, pk.ab.put(hann)});
    pk.fix.push_back({&ctx->sola_part, pk.ab.put(std::vector<float>(tvc::kSolaPartFloats, 0.f))});   // device scratch, not a table (sola.hip)
}

}  // namespace

// Prepared kNN blobs of this process: device pointer -> N it was prepared for.  The kernels take the blob's geometry (offsets of the
// inverse norms and the fp16 image) from the caller's N, so a call whose N differs from the one the blob was prepared with would read
// out of bounds: such a call is refused.  (A blob this process did not prepare - e.g. a copy - is unknown here and trusted.)
// The record is made only after the prepare launches succeeded; tvc_knn_forget drops it when the memory is handed to something else
// (a device address is recycled: a blob copied to where a blob of another size once lived must not inherit that record), and the
// registry is bounded: past kMaxBlobRecords it starts over (a forgotten record only loses this check).
static std::mutex g_blob_mu;
static std::map<const void*, int64_t> g_blobs;
constexpr size_t kMaxBlobRecords = 4096;
static void blob_record(const void* p, int64_t N) {
    std::lock_guard<std::mutex> lk(g_blob_mu);
    if (g_blobs.size() >= kMaxBlobRecords && !g_blobs.count(p)) g_blobs.clear();
    g_blobs[p] = N;
}
static void blob_forget(const void* p) {
    std::lock_guard<std::mutex> lk(g_blob_mu);
    g_blobs.erase(p);
}
// A blob this process did not prepare (a copy, a blob loaded from a file) is read ONCE - its 24-byte header, after the caller's stream has drained -
// and must carry this build's magic, format version and the caller's N; then it is recorded like a prepared one.  Inside a stream capture nothing
// may synchronise: an unknown blob is trusted there (capture after one eager call, as the module path does).
static int blob_check(tvc_ctx* ctx, hipStream_t s, const void* p, int64_t N, const char* what) {
    {
        std::lock_guard<std::mutex> lk(g_blob_mu);
        auto it = g_blobs.find(p);
        if (it != g_blobs.end()) {
            if (it->second != N)
                return fail(ctx, TVC_ERR_ARG, "%s: this blob was prepared for N = %lld index vectors, the call says N = %lld", what, (long long)it->second, (long long)N);
            return 0;
        }
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return 0;
    int h[6] = {0, 0, 0, 0, 0, 0};
    if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(s) != hipSuccess || hipMemcpy(h, p, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess)
        return fail(ctx, TVC_ERR_HIP, "%s: cannot read the prepared index's header", what);
    const int64_t n = ((int64_t)(unsigned)h[3] << 32) | (unsigned)h[2];
    if (h[0] != tvc::kBlobMagic || (h[1] != 0 && h[1] != 1))
        return fail(ctx, TVC_ERR_ARG, "%s: `prepared` is not a blob of tvc_knn_prepare_index_f32 / _f16", what);
    if (h[5] != tvc::kBlobVersion)
        return fail(ctx, TVC_ERR_ARG, "%s: the prepared index has format version %d, this library writes and reads version %d: prepare it again", what, h[5], tvc::kBlobVersion);
    if (n != N) return fail(ctx, TVC_ERR_ARG, "%s: this blob was prepared for N = %lld index vectors, the call says N = %lld", what, (long long)n, (long long)N);
    blob_record(p, N);
    return 0;
}

// noise_angle = NULL makes the library draw the phases from `seed` - a kernel ARGUMENT, which a stream capture bakes into the graph: every replay
// would synthesise the same noise.  A capturing caller must pass the phases (a buffer it refills between replays, as the module path does).
static int draw_under_capture(tvc_ctx* ctx, hipStream_t s, const float* noise_angle, const char* what) {
    if (noise_angle) return 0;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        return fail(ctx, TVC_ERR_STATE, "%s: noise_angle = NULL inside a stream capture would replay ONE seed's phases on every graph launch; pass noise_angle", what);
    return 0;
}

extern "C" {

int tvc_version(void) { return TVC_ABI_VERSION; }

int tvc_ctx_create(int hip_device, tvc_ctx** out) {
    if (!out) return TVC_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || hip_device < 0 || hip_device >= count) return TVC_ERR_HIP;
    tvc_ctx* c = new tvc_ctx();
    c->device = hip_device;
    {   // constant tables (FFT twiddles, Hann window): independent of any checkpoint
        Packer pk{c};
        build_fft_tables(pk, c);
        if (hipSetDevice(hip_device) != hipSuccess ||
            hipMalloc((void**)&c->const_arena, pk.ab.buf.size() * sizeof(float)) != hipSuccess ||
            hipMemcpy(c->const_arena, pk.ab.buf.data(), pk.ab.buf.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
            if (c->const_arena) (void)hipFree(c->const_arena);
            delete c;
            return TVC_ERR_HIP;
        }
        for (auto& f : pk.fix) *f.slot = c->const_arena + f.off;
    }
    // the side stream carries the pitch estimator beside the SSL trunk (encoder.hip): lowest priority, so that its workgroups take the
    // slots the trunk's launches leave free instead of competing with them (the pitch chain has ~150 us of slack)
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    if (hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, TVC_SIDE_PRIO_EXPR) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork2, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join2, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_amps, hipEventDisableTiming) != hipSuccess) {
        tvc_ctx_destroy(c);
        return TVC_ERR_HIP;
    }
    *out = c;
    return TVC_OK;
}

void tvc_ctx_destroy(tvc_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (auto& r : ctx->regions) {
        if (r.a) (void)hipEventDestroy(r.a);
        if (r.b) (void)hipEventDestroy(r.b);
    }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->ev_fork2) (void)hipEventDestroy(ctx->ev_fork2);
    if (ctx->ev_join2) (void)hipEventDestroy(ctx->ev_join2);
    if (ctx->ev_amps) (void)hipEventDestroy(ctx->ev_amps);
    if (ctx->side) (void)hipStreamDestroy(ctx->side);
    frontdoor_release(ctx);
    if (ctx->arena) (void)hipFree(ctx->arena);
    if (ctx->const_arena) (void)hipFree(ctx->const_arena);
    delete ctx;
}

const char* tvc_last_error(const tvc_ctx* ctx) { return ctx ? ctx->err : "null ctx"; }

int tvc_load_tensor(tvc_ctx* ctx, const char* key, const float* host_data, const int64_t* shape, int ndim) {
    if (!ctx || !key || !host_data || !shape || ndim < 1 || ndim > 4) return fail(ctx, TVC_ERR_ARG, "tvc_load_tensor: bad argument");
    size_t n = 1;
    HostTensor t;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] <= 0) return fail(ctx, TVC_ERR_ARG, "tvc_load_tensor(%s): bad shape", key);
        n *= (size_t)shape[i];
        t.shape.push_back(shape[i]);
    }
    t.data.assign(host_data, host_data + n);
    ctx->host[key] = std::move(t);
    ctx->enc_ready = ctx->dec_ready = false;
    return TVC_OK;
}

int tvc_set_pitch_table(tvc_ctx* ctx, const float* host_freqs, int n) {
    if (!ctx || !host_freqs || n != kPitchClasses) return fail(ctx, TVC_ERR_ARG, "pitch table must have %d entries", kPitchClasses);
    ctx->pitch_table.assign(host_freqs, host_freqs + n);
    ctx->enc_ready = ctx->dec_ready = false;
    return TVC_OK;
}

int tvc_finalize_weights(tvc_ctx* ctx) {
    if (!ctx) return TVC_ERR_ARG;
    ctx->enc_ready = ctx->dec_ready = false;
    Packer pk{ctx};
    if (ctx->pitch_table.size() == (size_t)kPitchClasses)
        pk.fix.push_back({&ctx->pitch_freq, pk.ab.put(ctx->pitch_table)});
    else
        pk.missing = "pitch table (tvc_set_pitch_table)";

    // encoder (encoder.py:75-116): both estimators read the same spectrogram -> stacked input 1x1
    pk.conv({"ssl_feature_estimator.input_layer", "pitch_estimator.input_layer"}, &ctx->enc_in, kBins, 1);
    pk.raw("ssl_feature_estimator.norm.gamma", &ctx->ssl_ln_g, kSslCh);
    pk.raw("ssl_feature_estimator.norm.beta", &ctx->ssl_ln_b, kSslCh);
    pk.raw("pitch_estimator.norm.gamma", &ctx->pit_ln_g, kPitchCh);
    pk.raw("pitch_estimator.norm.beta", &ctx->pit_ln_b, kPitchCh);
    static const int ssl_dil[6] = {1, 3, 9, 1, 1, 1};
    for (int i = 0; i < 6; ++i)
        pk.convnext("ssl_feature_estimator.mid_layers." + std::to_string(i), &ctx->ssl_mid[i], kSslCh, ssl_dil[i]);
    for (int i = 0; i < 4; ++i)
        pk.convnext("pitch_estimator.mid_layers." + std::to_string(i), &ctx->pit_mid[i], kPitchCh, 1);
    pk.conv({"ssl_feature_estimator.output_layer"}, &ctx->ssl_out, kSslCh, 1);
    pk.conv({"pitch_estimator.output_layer"}, &ctx->pit_out, kPitchCh, 1);
    const std::string missing_enc = pk.missing;
    pk.missing.clear();

    // source net (decoder.py:102-134)
    pk.conv({"source_net.content_in"}, &ctx->src_content_in, kSslDim, 1);
    pk.raw("source_net.energy_in.weight", &ctx->src_e_w, kSrcCh);
    pk.raw("source_net.energy_in.bias", &ctx->src_e_b, kSrcCh);
    pk.raw("source_net.f0_in.weight", &ctx->src_f_w, kSrcCh);
    pk.raw("source_net.f0_in.bias", &ctx->src_f_b, kSrcCh);
    for (int i = 0; i < 3; ++i)
        pk.convnext("source_net.mid_layers." + std::to_string(i), &ctx->src_mid[i], kSrcCh, 1);
    pk.conv({"source_net.to_amps"}, &ctx->src_to_amps, kSrcCh, 1);
    pk.conv({"source_net.to_kernel"}, &ctx->src_to_kernel, kSrcCh, 1);

    // filter net (decoder.py:193-233)
    static const int ch[5] = {384, 192, 96, 48, 24};
    static const int fac[5] = {2, 3, 4, 4, 5};
    pk.conv({"filter_net.content_in"}, &ctx->flt_content_in, kSslDim, 1);
    pk.raw("filter_net.f0_in.weight", &ctx->flt_f_w, ch[0]);
    pk.raw("filter_net.f0_in.bias", &ctx->flt_f_b, ch[0]);
    // analytic |max| bounds of 1x1 outputs (a hair above max_m sum_k |w_mk| and max |b|): the slot of a tensor an epilogue functor finishes
    // comes from its input's slot instead of a pass over the tensor
    auto bound_1x1 = [&](const std::string& name, int cout, int cin, float* bw, float* bb) {
        const HostTensor* w = pk.find(name + ".weight");
        const HostTensor* b = pk.find(name + ".bias");
        if (!w || !b || w->data.size() != (size_t)cout * cin || b->data.size() != (size_t)cout) return;
        double wl1 = 0.0, bm = 0.0;
        for (int m = 0; m < cout; ++m) {
            double sum = 0.0;
            for (int k = 0; k < cin; ++k) sum += std::fabs((double)w->data[(size_t)m * cin + k]);
            wl1 = std::max(wl1, sum);
            bm = std::max(bm, std::fabs((double)b->data[m]));
        }
        *bw = (float)(wl1 * 1.0001);
        *bb = (float)(bm * 1.0001);
    };
    {
        bound_1x1("filter_net.content_in", ch[0], kSslDim, &ctx->flt_in_bw, &ctx->flt_in_bb);
        const HostTensor* fw = pk.find("filter_net.f0_in.weight");
        const HostTensor* fb = pk.find("filter_net.f0_in.bias");
        if (fw && fb) {      // + f0_in(log(relu(f0) + 1e-6)): |log| < 89 for every finite fp32 f0
            double wm = 0.0, bm = 0.0;
            for (float v : fw->data) wm = std::max(wm, std::fabs((double)v));
            for (float v : fb->data) bm = std::max(bm, std::fabs((double)v));
            ctx->flt_in_bb += (float)((wm * 89.0 + bm) * 1.0001);
        }
    }
    pk.down0s(&ctx->flt_down0s, "filter_net.downs.0", &ctx->down0_bw, &ctx->down0_bb);
    for (int i = 1; i <= 4; ++i) {
        DownW& d = ctx->downs[i - 1];
        d.cin = ch[5 - i];
        d.cout = ch[4 - i];
        d.factor = fac[5 - i];
        std::string p = "filter_net.downs." + std::to_string(i);
        pk.conv({p + ".c1"}, &d.c1, d.cin, 3);
        pk.conv({p + ".c2"}, &d.c2, d.cin, 3);
        pk.conv_joint(p + ".c3", &d.c3, d.cin, 3, p + ".down_res", &d.res, d.cin, 1);      // c3(h2) + down_res(xi) land in one tile: joint scales
        {   // c3.bias + down_res.bias for the launches that accumulate both convs into one tile
            const HostTensor* b3 = pk.find(p + ".c3.bias");
            const HostTensor* br = pk.find(p + ".down_res.bias");
            if (b3 && br && b3->data.size() == (size_t)d.cout && br->data.size() == (size_t)d.cout) {
                std::vector<float> sum(d.c3.Mpad, 0.f);
                for (int m = 0; m < d.cout; ++m) sum[m] = b3->data[m] + br->data[m];
                pk.fix.push_back({&d.c3res_bias, pk.ab.put(sum)});
            }
        }
        if (d.cin == 24 && d.cout == 48) {
            pk.conv24s(&d.s24c1, p + ".c1", 24);
            pk.conv24s(&d.s24c2, p + ".c2", 24);
            pk.conv24s(&d.s24c3r, p + ".c3", 48, p + ".down_res.bias", &d.c3);
            // bounds of the fused block's on-chip intermediates (down24f_kernel): max_m sum_k |w| and max |b| of c1 and c2, a hair above
            auto bound = [&](const std::string& name, float* bw, float* bb) {
                const HostTensor* w = pk.find(name + ".weight");
                const HostTensor* b = pk.find(name + ".bias");
                if (!w || !b || w->data.size() != (size_t)24 * 24 * 3 || b->data.size() != 24) return;
                double wl1 = 0.0, bm = 0.0;
                for (int m = 0; m < 24; ++m) {
                    double sum = 0.0;
                    for (int k = 0; k < 72; ++k) sum += std::fabs((double)w->data[(size_t)m * 72 + k]);
                    wl1 = std::max(wl1, sum);
                    bm = std::max(bm, std::fabs((double)b->data[m]));
                }
                *bw = (float)(wl1 * 1.0001);
                *bb = (float)(bm * 1.0001);
            };
            bound(p + ".c1", &d.b1_w, &d.b1_b);
            bound(p + ".c2", &d.b2_w, &d.b2_b);
        }
    }
    for (int i = 0; i < 5; ++i) {
        UpW& u = ctx->ups[i];
        u.cin = ch[i];
        u.cout = i < 4 ? ch[i + 1] : ch[4];
        u.factor = fac[i];
        std::string p = "filter_net.ups." + std::to_string(i);
        pk.conv({p + ".c1"}, &u.c1, u.cin, 3);
        pk.conv({p + ".c2"}, &u.c2, u.cin, 3);
        pk.conv({p + ".c3"}, &u.c3, u.cin, 3);
        pk.conv({p + ".c4"}, &u.c4, u.cin, 3);
        pk.conv({p + ".c5"}, &u.c5, u.cin, 1);
        bound_1x1(p + ".c5", u.cout, u.cin, &u.c5_bw, &u.c5_bb);
        pk.conv({p + ".film1.to_scale", p + ".film1.to_shift"}, &u.film1, u.cin, 1);
        pk.conv({p + ".film2.to_scale", p + ".film2.to_shift"}, &u.film2, u.cin, 1);
        if (u.cin >= 96) {
            pk.film_u(&u.fu1, p + ".c2", p + ".film1", u.cin, p + ".c1");
            pk.film_u(&u.fu2, p + ".c4", p + ".film2", u.cin, p + ".c3");
        }
        if (u.cin == 24) {
            pk.up24s_half(&u.s24a, p + ".c1", p + ".c2", p + ".film1", "", "");
            pk.up24s_half(&u.s24b, p + ".c3", p + ".c4", p + ".film2", p + ".c5", "filter_net.output_layer");
        }
    }

    const std::string missing_dec = pk.missing;
    snprintf(ctx->enc_missing, sizeof(ctx->enc_missing), "%s", missing_enc.c_str());
    snprintf(ctx->dec_missing, sizeof(ctx->dec_missing), "%s", missing_dec.c_str());
    if (!missing_enc.empty() && !missing_dec.empty())
        return fail(ctx, TVC_ERR_STATE, "no complete checkpoint: encoder lacks %s; decoder lacks %s", missing_enc.c_str(), missing_dec.c_str());

    TVC_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->arena) {
        TVC_HIP(ctx, hipFree(ctx->arena));
        ctx->arena = nullptr;
    }
    ctx->arena_floats = pk.ab.buf.size();
    TVC_HIP(ctx, hipMalloc((void**)&ctx->arena, ctx->arena_floats * sizeof(float)));
    TVC_HIP(ctx, hipMemcpy(ctx->arena, pk.ab.buf.data(), ctx->arena_floats * sizeof(float), hipMemcpyHostToDevice));
    for (auto& f : pk.fix) *f.slot = ctx->arena + f.off;
    ctx->enc_ready = missing_enc.empty();
    ctx->dec_ready = missing_dec.empty();
    ctx->host.clear();   // staged copies are no longer needed
    return TVC_OK;
}

enum { NEED_NONE = 0, NEED_ENC = 1, NEED_DEC = 2 };
static int need_ready(tvc_ctx* ctx, int need) {
    if (!ctx) return TVC_ERR_ARG;
    if ((need & NEED_ENC) && !ctx->enc_ready)
        return fail(ctx, TVC_ERR_STATE, "encoder weights not loaded (%s)", ctx->enc_missing[0] ? ctx->enc_missing : "tvc_finalize_weights not called");
    if ((need & NEED_DEC) && !ctx->dec_ready)
        return fail(ctx, TVC_ERR_STATE, "decoder weights not loaded (%s)", ctx->dec_missing[0] ? ctx->dec_missing : "tvc_finalize_weights not called");
    return 0;
}

static int convert_impl(tvc_ctx* ctx, hipStream_t s, Ws& ws, bool dry, const float* wav, const float* prepared,
                        int64_t N, float pitch_shift, const float* angle,
                        uint64_t seed, float* wave, int B, int64_t L) {
    const int T = (int)(L / kHop);
    float* spec = ws.get<float>((size_t)B * kBins * T);
    float* energy = ws.get<float>((size_t)B * L);
    float* ssl = ws.get<float>((size_t)B * kSslDim * T);
    float* matched = ws.get<float>((size_t)B * kSslDim * T);
    float* f0 = ws.get<float>((size_t)B * T);
    float* f0s = ws.get<float>((size_t)B * T);
    // |max| slots the path can bound without a pass over the tensors (block-floating-point guard of the fp16 split, conv3s.h; equal-length
    // batches): emax = max |wav| per utterance (the energy stage's pooled maxima, 1 500 values each) bounds the energy envelope - a linear
    // interpolation of them - and, times the Hann window's sum (960), every |STFT| bin; `matched` is a mean of index rows.
    // A ragged batch (the driver passed B = 1, T = all frames: ragged.h) derives them per utterance in the same way - an utterance's scales,
    // and with them its bits, are those of its own B = 1 call at every input amplitude (test_gpu_ragged.py scales the input by 1e4 and 1e-7).
    const int NB = ctx->rag ? ctx->rag->B : B;
    float* emax = ws.get<float>((size_t)5 * NB);
    float* spec_bound = emax + NB;
    float* enc_slots = spec_bound + NB;      // the encoder's three atomicMax slots: zeroed by the energy stage's pooled-maximum launch
    const bool bounds = true;
    size_t m = ws.mark();
    {
        ProfScope ps(ctx, s, dry, "stft");
        TVC_CHECK(run_stft(ctx, s, ws, dry, wav, spec, B, L));
    }
    ws.release(m);
    {
        ProfScope ps(ctx, s, dry, "energy");
        TVC_CHECK(run_energy(ctx, s, ws, dry, wav, energy, B, L, bounds ? emax : nullptr, bounds ? spec_bound : nullptr, bounds ? enc_slots : nullptr, 3 * NB));
    }
    ws.release(m);
    {
        ProfScope ps(ctx, s, dry, "encoder");
        TVC_CHECK(run_encoder(ctx, s, ws, dry, spec, ssl, f0, nullptr, B, T, bounds ? spec_bound : nullptr, bounds ? enc_slots : nullptr, f0s, pitch_shift));
    }
    ws.release(m);
    {
        ProfScope ps(ctx, s, dry, "knn");
        TVC_CHECK(run_knn(ctx, s, ws, dry, ssl, prepared, N, matched, nullptr, B, T));
    }
    ws.release(m);
    TVC_CHECK(run_decoder(ctx, s, ws, dry, matched, f0s, energy, angle, seed, wave, nullptr, nullptr, nullptr, B, T, dry ? nullptr : knn_index_amax(prepared),
                          bounds ? emax : nullptr));
    ws.release(m);
    return 0;
}

int tvc_workspace_bytes(tvc_ctx* ctx, int B, int64_t L, int64_t N, size_t* out_bytes) {
    TVC_CHECK(need_ready(ctx, NEED_NONE));
    if (!out_bytes || B <= 0 || L <= 0 || L % kHop != 0 || N < 4) return fail(ctx, TVC_ERR_ARG, "tvc_workspace_bytes: need B>0, L%%480==0, N>=4");
    Ws ws(nullptr, 0, true);
    TVC_CHECK(convert_impl(ctx, nullptr, ws, true, nullptr, nullptr, N, 0.f, nullptr, 0, nullptr, B, L));
    *out_bytes = ws.peak + 4096;
    return TVC_OK;
}

// Workspace is validated *before* launching: every entry runs its driver once in dry mode.
#define TVC_RUN(call_dry, call_real)                                                              \
    {                                                                                             \
        Ws dryws(nullptr, 0, true);                                                               \
        {                                                                                         \
            Ws& ws = dryws;                                                                       \
            int rc0 = (call_dry);                                                                 \
            if (rc0) return rc0;                                                                  \
        }                                                                                         \
        if (dryws.peak > ws_bytes)                                                                \
            return fail(ctx, TVC_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", dryws.peak, ws_bytes); \
        Ws ws(wsp, ws_bytes, false);                                                              \
        return (call_real);                                                                       \
    }

int tvc_stft_mag_f32(tvc_ctx* ctx, void* stream, const float* wav, float* spec, int B, int64_t L, void* wsp, size_t ws_bytes) {
    TVC_CHECK(need_ready(ctx, NEED_NONE));
    if (!wav || !spec || B <= 0 || L <= 0 || L % kHop) return fail(ctx, TVC_ERR_ARG, "tvc_stft_mag_f32: bad argument (L must be a multiple of 480)");
    if (L < kNfft / 2 + 1) return fail(ctx, TVC_ERR_ARG, "tvc_stft_mag_f32: L must exceed 960 (reflect padding)");
    TVC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    TVC_RUN(run_stft(ctx, s, ws, true, wav, spec, B, L), run_stft(ctx, s, ws, false, wav, spec, B, L));
}

int64_t tvc_resample_out_len(int64_t n, int orig_freq, int new_freq) { return resample_out_len(n, orig_freq, new_freq); }

int tvc_resample_f32(tvc_ctx* ctx, void* stream, const float* x, float* y, int rows, int64_t n, int orig_freq, int new_freq) {
    if (!ctx) return TVC_ERR_ARG;
    if (!x || !y || rows <= 0 || n <= 0 || orig_freq <= 0 || new_freq <= 0) return fail(ctx, TVC_ERR_ARG, "tvc_resample_f32: bad argument");
    TVC_HIP(ctx, hipSetDevice(ctx->device));
    return run_resample(ctx, (hipStream_t)stream, x, y, rows, n, orig_freq, new_freq);
}

int tvc_pcm16_to_f32(tvc_ctx* ctx, void* stream, const int16_t* pcm, float* y, int64_t n, float gain_db) {
    if (!ctx) return TVC_ERR_ARG;
    if (!pcm || !y || n <= 0) return fail(ctx, TVC_ERR_ARG, "tvc_pcm16_to_f32: bad argument");
    TVC_HIP(ctx, hipSetDevice(ctx->device));
    return run_pcm16_to_f32(ctx, (hipStream_t)stream, pcm, y, n, gain_db);
}

int tvc_f32_to_pcm16(tvc_ctx* ctx, void* stream, const float* x, int16_t* pcm, int64_t n, float gain_db) {
    if (!ctx) return TVC_ERR_ARG;
    if (!x || !pcm || n <= 0) return fail(ctx, TVC_ERR_ARG, "tvc_f32_to_pcm16: bad argument");
    TVC_HIP(ctx, hipSetDevice(ctx->device));
    return run_f32_to_pcm16(ctx, (hipStream_t)stream, x, pcm, n, gain_db);
}

int tvc_energy_f32(tvc_ctx* ctx, void* stream, const float* wav, float* energy, int B, int64_t L, void* wsp, size_t ws_bytes) {
    TVC_CHECK(need_ready(ctx, NEED_NONE));
    if (!wav || !energy || B <= 0 || L < 128) return fail(ctx, TVC_ERR_ARG, "tvc_energy_f32: bad argument");
    TVC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    TVC_RUN(run_energy(ctx, s, ws, true, wav, energy, B, L), run_energy(ctx, s, ws, false, wav, energy, B, L));
}

int tvc_encoder_f32(tvc_ctx* ctx, void* stream, const float* spec, float* ssl, float* f0, float* logits, int B, int T, void* wsp, size_t ws_bytes) {
    TVC_CHECK(need_ready(ctx, NEED_ENC));
    if (!spec || !ssl || !f0 || B <= 0 || T <= 0) return fail(ctx, TVC_ERR_ARG, "tvc_encoder_f32: bad argument");
    TVC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    TVC_RUN(run_encoder(ctx, s, ws, true, spec, ssl, f0, logits, B, T), run_encoder(ctx, s, ws, false, spec, ssl, f0, logits, B, T));
}

int tvc_pitch_decode_f32(tvc_ctx* ctx, void* stream, const float* logits, float* f0, int B, int T) {
    TVC_CHECK(need_ready(ctx, NEED_ENC));
    if (!logits || !f0 || B <= 0 || T <= 0) return fail(ctx, TVC_ERR_ARG, "tvc_pitch_decode_f32: bad argument");
    TVC_HIP(ctx, hipSetDevice(ctx->device));
    return run_pitch_decode(ctx, (hipStream_t)stream, logits, f0, B, T);
}

int64_t tvc_knn_prepared_elems(int64_t N) {
    if (N <= 0) return 0;
    int64_t npad = (N + 127) / 128 * 128;
    // header + raw rows [N][768] + split image of the normalised vectors (3 bf16 per value = 1.5 floats)
    // + inverse norms [Npad] + fp16 image of the normalised vectors (the coarse pass's operand, half a float per value)
    return 64 + N * (int64_t)kSslDim + (int64_t)kSslDim * npad * 3 / 2 + npad + (int64_t)kSslDim * npad / 2;
}

int64_t tvc_knn_prepared_elems_f16(int64_t N) {
    if (N <= 0) return 0;
    int64_t npad = (N + 127) / 128 * 128;
    // header + inverse norms [Npad] + fp16 image (half a float per value) + the largest inverse norm of every 128-vector tile
    return 64 + npad + (int64_t)kSslDim * npad / 2 + npad / 128;
}

int tvc_knn_prepare_index_f32(tvc_ctx* ctx, void* stream, const float* index, float* prepared, int64_t N) {
    if (!ctx) return TVC_ERR_ARG;
    if (!index || !prepared || N <= 0) return fail(ctx, TVC_ERR_ARG, "tvc_knn_prepare_index_f32: bad argument");
    TVC_HIP(ctx, hipSetDevice(ctx->device));
    blob_forget(prepared);
    TVC_CHECK(run_prepare_index(ctx, (hipStream_t)stream, index, prepared, N));
    blob_record(prepared, N);
    return TVC_OK;
}

int tvc_knn_forget(tvc_ctx* ctx, const float* prepared) {
    if (!ctx) return TVC_ERR_ARG;
    blob_forget(prepared);
    return TVC_OK;
}

int tvc_knn_prepare_index_f16(tvc_ctx* ctx, void* stream, const void* rows_f16, float* prepared, int64_t N) {
    if (!ctx) return TVC_ERR_ARG;
    if (!rows_f16 || !prepared || N <= 0) return fail(ctx, TVC_ERR_ARG, "tvc_knn_prepare_index_f16: bad argument");
    TVC_HIP(ctx, hipSetDevice(ctx->device));
    blob_forget(prepared);
    TVC_CHECK(run_prepare_index_f16(ctx, (hipStream_t)stream, rows_f16, prepared, N));
    blob_record(prepared, N);
    return TVC_OK;
}

int tvc_knn_match_f32(tvc_ctx* ctx, void* stream, const float* src, const float* prepared, int64_t N, float* out,
                      int64_t* idx_out, int B, int T, void* wsp, size_t ws_bytes) {
    if (!ctx) return TVC_ERR_ARG;
    if (!src || !prepared || !out || B <= 0 || T <= 0) return fail(ctx, TVC_ERR_ARG, "tvc_knn_match_f32: bad argument");
    if (N < 4) return fail(ctx, TVC_ERR_ARG, "tvc_knn_match_f32: index needs at least k=4 vectors (torch.topk raises too)");
    TVC_CHECK(blob_check(ctx, (hipStream_t)stream, prepared, N, "tvc_knn_match_f32"));
    TVC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    TVC_RUN(run_knn(ctx, s, ws, true, src, prepared, N, out, idx_out, B, T),
            run_knn(ctx, s, ws, false, src, prepared, N, out, idx_out, B, T));
}

int tvc_knn_match_general_f32(tvc_ctx* ctx, void* stream, const float* src, const float* index, int64_t N, int k, int metric, float* out,
                              int64_t* idx_out, float* sim_out, int B, int T, void* wsp, size_t ws_bytes) {
    if (!ctx) return TVC_ERR_ARG;
    if (!src || !index || !out || B <= 0 || T <= 0) return fail(ctx, TVC_ERR_ARG, "tvc_knn_match_general_f32: bad argument");
    if (k < 1 || k > 8) return fail(ctx, TVC_ERR_ARG, "tvc_knn_match_general_f32: k must be 1 ... 8");
    if (metric < 0 || metric > 2) return fail(ctx, TVC_ERR_ARG, "tvc_knn_match_general_f32: metric is 0 ('cos'), 1 ('IP') or 2 ('L2')");
    if (N < k) return fail(ctx, TVC_ERR_ARG, "tvc_knn_match_general_f32: selected index k out of range (the index has fewer than k vectors; torch.topk raises too)");
    if (N > 0x7ffffffe || (long)B * T > 0x7fffffff) return fail(ctx, TVC_ERR_ARG, "tvc_knn_match_general_f32: sizes beyond 32-bit indexing");
    TVC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    TVC_RUN(run_knn_general(ctx, s, ws, true, src, index, N, k, metric, out, idx_out, sim_out, B, T),
            run_knn_general(ctx, s, ws, false, src, index, N, k, metric, out, idx_out, sim_out, B, T));
}

int tvc_knn_topk_f32(tvc_ctx* ctx, void* stream, const float* src, const float* prepared, int64_t N, float* sims_out,
                     int64_t* idx_out, int B, int T, void* wsp, size_t ws_bytes) {
    if (!ctx) return TVC_ERR_ARG;
    if (!src || !prepared || !sims_out || !idx_out || B <= 0 || T <= 0) return fail(ctx, TVC_ERR_ARG, "tvc_knn_topk_f32: bad argument");
    if (N < 4) return fail(ctx, TVC_ERR_ARG, "tvc_knn_topk_f32: an index shard needs at least k=4 vectors");
    TVC_CHECK(blob_check(ctx, (hipStream_t)stream, prepared, N, "tvc_knn_topk_f32"));
    TVC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    TVC_RUN(run_knn_topk(ctx, s, ws, true, src, prepared, N, sims_out, idx_out, B, T),
            run_knn_topk(ctx, s, ws, false, src, prepared, N, sims_out, idx_out, B, T));
}

int tvc_knn_gather_slots_f32(tvc_ctx* ctx, void* stream, const float* prepared, int64_t N, const int64_t* idx, float* slots,
                             int64_t nslots) {
    if (!ctx) return TVC_ERR_ARG;
    if (!prepared || !idx || !slots || N <= 0 || nslots <= 0) return fail(ctx, TVC_ERR_ARG, "tvc_knn_gather_slots_f32: bad argument");
    TVC_CHECK(blob_check(ctx, (hipStream_t)stream, prepared, N, "tvc_knn_gather_slots_f32"));
    TVC_HIP(ctx, hipSetDevice(ctx->device));
    return run_knn_slots(ctx, (hipStream_t)stream, prepared, N, idx, slots, nslots);
}

int tvc_knn_finish_f32(tvc_ctx* ctx, void* stream, const float* slots, float* out, int B, int T) {
    if (!ctx) return TVC_ERR_ARG;
    if (!slots || !out || B <= 0 || T <= 0) return fail(ctx, TVC_ERR_ARG, "tvc_knn_finish_f32: bad argument");
    TVC_HIP(ctx, hipSetDevice(ctx->device));
    return run_knn_finish(ctx, (hipStream_t)stream, slots, out, B, T);
}

int tvc_shift_frequency_f32(tvc_ctx* ctx, void* stream, const float* f0, float* out, int64_t n, float semitones) {
    if (!ctx) return TVC_ERR_ARG;
    if (!f0 || !out || n <= 0) return fail(ctx, TVC_ERR_ARG, "tvc_shift_frequency_f32: bad argument");
    TVC_HIP(ctx, hipSetDevice(ctx->device));
    return run_shift(ctx, (hipStream_t)stream, f0, out, n, semitones);
}

int tvc_noise_angle_from_uniform_f32(tvc_ctx* ctx, void* stream, float* u, int64_t n) {
    if (!ctx) return TVC_ERR_ARG;
    if (!u || n <= 0) return fail(ctx, TVC_ERR_ARG, "tvc_noise_angle_from_uniform_f32: bad argument");
    TVC_HIP(ctx, hipSetDevice(ctx->device));
    return run_uniform_to_angle(ctx, (hipStream_t)stream, u, n);
}

int tvc_decoder_stages_f32(tvc_ctx* ctx, void* stream, const float* content, const float* f0, const float* energy,
                           const float* noise_angle, uint64_t seed, float* wave, float* amps, float* kernel,
                           float* source, int B, int T, void* wsp, size_t ws_bytes) {
    TVC_CHECK(need_ready(ctx, NEED_DEC));
    if (!content || !f0 || !energy || B <= 0 || T <= 0 || (!wave && !amps && !kernel && !source)) return fail(ctx, TVC_ERR_ARG, "tvc_decoder_f32: bad argument");
    if (wave || source) TVC_CHECK(draw_under_capture(ctx, (hipStream_t)stream, noise_angle, "tvc_decoder_f32"));
    TVC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    TVC_RUN(run_decoder(ctx, s, ws, true, content, f0, energy, noise_angle, seed, wave, amps, kernel, source, B, T),
            run_decoder(ctx, s, ws, false, content, f0, energy, noise_angle, seed, wave, amps, kernel, source, B, T));
}

int tvc_filter_net_f32(tvc_ctx* ctx, void* stream, const float* content, const float* f0, const float* energy, const float* source,
                       float* wave, float* const* skips, float* const* ups, int B, int T, void* wsp, size_t ws_bytes) {
    TVC_CHECK(need_ready(ctx, NEED_DEC));
    if (!content || !f0 || !energy || !source || !wave || B <= 0 || T <= 0) return fail(ctx, TVC_ERR_ARG, "tvc_filter_net_f32: bad argument");
    TVC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    FilterTaps taps;
    for (int i = 0; i < 5 && skips; ++i) taps.skips[i] = skips[i];
    for (int i = 0; i < 4 && ups; ++i) taps.ups[i] = ups[i];
    TVC_RUN(run_filter(ctx, s, ws, true, content, f0, energy, source, wave, B, T, nullptr),
            run_filter(ctx, s, ws, false, content, f0, energy, source, wave, B, T, &taps));
}

int tvc_dsp_f32(tvc_ctx* ctx, void* stream, const float* f0, const float* amps, const float* kernel, const float* noise_angle,
                uint64_t seed, float* source, int B, int T, void* wsp, size_t ws_bytes) {
    TVC_CHECK(need_ready(ctx, NEED_NONE));
    if (!f0 || !amps || !kernel || !source || B <= 0 || T <= 0) return fail(ctx, TVC_ERR_ARG, "tvc_dsp_f32: bad argument");
    TVC_CHECK(draw_under_capture(ctx, (hipStream_t)stream, noise_angle, "tvc_dsp_f32"));
    TVC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    TVC_RUN(run_dsp(ctx, s, ws, true, f0, amps, kernel, noise_angle, seed, source, B, T),
            run_dsp(ctx, s, ws, false, f0, amps, kernel, noise_angle, seed, source, B, T));
}

int tvc_decoder_f32(tvc_ctx* ctx, void* stream, const float* content, const float* f0, const float* energy,
                    const float* noise_angle, uint64_t seed, float* wave, int B, int T, void* wsp, size_t ws_bytes) {
    return tvc_decoder_stages_f32(ctx, stream, content, f0, energy, noise_angle, seed, wave, nullptr, nullptr, nullptr, B, T, wsp, ws_bytes);
}

int tvc_convert_f32(tvc_ctx* ctx, void* stream, const float* wav, const float* prepared, int64_t N,
                    float pitch_shift, const float* noise_angle, uint64_t seed, float* wave, int B, int64_t L,
                    void* wsp, size_t ws_bytes) {
    TVC_CHECK(need_ready(ctx, NEED_ENC | NEED_DEC));
    if (!wav || !prepared || !wave || B <= 0 || L <= 0 || L % kHop) return fail(ctx, TVC_ERR_ARG, "tvc_convert_f32: bad argument (L must be a positive multiple of 480)");
    if (L < kNfft / 2 + 1) return fail(ctx, TVC_ERR_ARG, "tvc_convert_f32: L must exceed 960 samples (STFT reflect padding, as torch.stft requires)");
    if (N < 4) return fail(ctx, TVC_ERR_ARG, "tvc_convert_f32: index needs at least k=4 vectors");
    TVC_CHECK(blob_check(ctx, (hipStream_t)stream, prepared, N, "tvc_convert_f32"));
    TVC_CHECK(draw_under_capture(ctx, (hipStream_t)stream, noise_angle, "tvc_convert_f32"));
    TVC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    TVC_RUN(convert_impl(ctx, s, ws, true, wav, prepared, N, pitch_shift, noise_angle, seed, wave, B, L),
            convert_impl(ctx, s, ws, false, wav, prepared, N, pitch_shift, noise_angle, seed, wave, B, L));
}

// ---- ragged batches ---------------------------------------------------------------------------------------------------------
namespace {
// Every utterance of a ragged call is converted inside the kernels (ragged.h), in batches of utterances that select the SAME kernels: which
// FiLM kernel a FilterNet level runs depends on the utterance's own length there (film_s2 / the pre-split hand-over need one 256-column tile:
// 2 T, 6 T, 24 T >= 256, decoder.hip film_conv), so the frame counts split into four classes at 11, 43 and 128 frames; inside a class every
// utterance takes exactly the path its own B = 1 call takes and the result is bit-identical to it.
constexpr int kRagClassBounds[3] = {11, 43, 128};
constexpr int kRagMaxFrames = 80000;       // frames per in-kernel batch: 24 rows x 480 x 4 B x frames stays below the 32-bit byte offsets of the 24-channel kernels
struct RagBatchPlan {
    std::vector<int> rows, frames;
    int Ttot = 0;
};
int ragged_split(tvc_ctx* ctx, int cap, int B, int64_t Lmax, const int64_t* lens, std::vector<RagBatchPlan>* batches) {
    std::vector<RagBatchPlan> open(4);          // the batch being filled, per class (cap: tvc_ctx_set_ragged_batch_frames / tvc_ragged_plan's argument, 0 = the default)
    const int max_frames = cap > 0 && cap < kRagMaxFrames ? cap : kRagMaxFrames;
    for (int b = 0; b < B; ++b) {
        if (lens[b] <= 0 || lens[b] % kHop || lens[b] > Lmax || lens[b] < kNfft / 2 + 1)
            return fail(ctx, TVC_ERR_ARG, "ragged batch: lens[%d] = %lld must be a multiple of 480 in (960, Lmax]", b, (long long)lens[b]);
        const int T = (int)(lens[b] / kHop);
        if (T > kRagMaxFrames) return fail(ctx, TVC_ERR_ARG, "ragged batch: lens[%d] = %lld is longer than a batch may be; convert it with tvc_convert_f32", b, (long long)lens[b]);
        const int cls = (T >= kRagClassBounds[0]) + (T >= kRagClassBounds[1]) + (T >= kRagClassBounds[2]);
        RagBatchPlan& p = open[cls];
        if (p.Ttot + T > max_frames && !p.rows.empty()) {
            batches->push_back(p);
            p = RagBatchPlan();
        }
        p.rows.push_back(b);
        p.frames.push_back(T);
        p.Ttot += T;
    }
    for (int c = 3; c >= 0; --c)
        if (!open[c].rows.empty()) batches->push_back(open[c]);
    return 0;
}
// one in-kernel ragged batch: [tables][convert workspace]; the drivers run it as ONE utterance of Ttot frames (B = 1) with ctx->rag set
int ragged_batch(tvc_ctx* ctx, hipStream_t s, Ws& ws, bool dry, const RagBatchPlan& p, const float* wav, int64_t Lmax, const float* prepared, int64_t N,
                 float pitch_shift, const float* angle, uint64_t seed, float* wave) {
    int* scratch = ws.get<int>(rag_scratch_ints((int)p.rows.size(), p.Ttot));
    RagHost h;
    TVC_CHECK(rag_setup(ctx, s, dry, h, p.frames, p.rows, (int)(Lmax / kHop), scratch));
    ctx->rag = &h;
    const int rc = convert_impl(ctx, s, ws, dry, wav, prepared, N, pitch_shift, angle, seed, wave, 1, (int64_t)p.Ttot * kHop);
    ctx->rag = nullptr;
    return rc;
}
// the batches of a call run one after the other on the caller's stream and share one workspace region.  Sized for a call with AND without
// caller-supplied noise phases (the library's own draw needs a buffer the injected phases do not).
// angle_mode: 0 / 1 = a call without / with caller-supplied phases (what a conversion needs), -1 = the larger of the two (what
// tvc_workspace_bytes_ragged promises: it does not know which call will follow)
int ragged_batch_bytes(tvc_ctx* ctx, const std::vector<RagBatchPlan>& batches, int64_t Lmax, int64_t N, size_t* bytes, int angle_mode) {
    *bytes = 0;
    for (auto& p : batches)
        for (int with_angle = angle_mode < 0 ? 0 : angle_mode; with_angle <= (angle_mode < 0 ? 1 : angle_mode); ++with_angle) {
            Ws ws(nullptr, 0, true);
            TVC_CHECK(ragged_batch(ctx, nullptr, ws, true, p, nullptr, Lmax, nullptr, N, 0.f, with_angle ? (const float*)256 : nullptr, 0, nullptr));
            const size_t need = (ws.peak + 4095) & ~size_t(4095);
            if (need > *bytes) *bytes = need;
        }
    return 0;
}
}  // namespace

int tvc_ctx_set_ragged_batch_frames(tvc_ctx* ctx, int max_frames) {
    if (!ctx) return TVC_ERR_ARG;
    if (max_frames < 0) return fail(ctx, TVC_ERR_ARG, "tvc_ctx_set_ragged_batch_frames: the cap is a frame count (0 = the default)");
    ctx->rag_batch_frames = max_frames;
    return TVC_OK;
}
int tvc_ragged_plan(int B, int64_t Lmax, const int64_t* lens, int max_frames, int32_t* batch_of_row, int* n_batches) {
    if (!lens || !batch_of_row || !n_batches || B <= 0 || Lmax <= 0 || Lmax % kHop != 0 || max_frames < 0) return TVC_ERR_ARG;
    std::vector<RagBatchPlan> batches;
    const int rc = ragged_split(nullptr, max_frames, B, Lmax, lens, &batches);
    if (rc) return rc;
    for (size_t i = 0; i < batches.size(); ++i)
        for (int b : batches[i].rows) batch_of_row[b] = (int32_t)i;
    *n_batches = (int)batches.size();
    return TVC_OK;
}

int tvc_workspace_bytes_ragged(tvc_ctx* ctx, int B, int64_t Lmax, const int64_t* lens, int64_t N, size_t* out_bytes) {
    TVC_CHECK(need_ready(ctx, NEED_NONE));
    if (!out_bytes || !lens || B <= 0 || Lmax <= 0 || Lmax % kHop != 0 || N < 4) return fail(ctx, TVC_ERR_ARG, "tvc_workspace_bytes_ragged: need B>0, Lmax%%480==0, N>=4");
    std::vector<RagBatchPlan> batches;
    TVC_CHECK(ragged_split(ctx, ctx->rag_batch_frames, B, Lmax, lens, &batches));
    size_t bytes = 0;
    TVC_CHECK(ragged_batch_bytes(ctx, batches, Lmax, N, &bytes, -1));
    *out_bytes = bytes + 4096;
    return TVC_OK;
}

int tvc_convert_ragged_f32(tvc_ctx* ctx, void* stream, const float* wav, int64_t Lmax, const int64_t* lens, const float* prepared, int64_t N,
                           float pitch_shift, const float* noise_angle, uint64_t seed, float* wave, int B, void* wsp, size_t ws_bytes) {
    TVC_CHECK(need_ready(ctx, NEED_ENC | NEED_DEC));
    if (!wav || !lens || !prepared || !wave || B <= 0 || Lmax <= 0 || Lmax % kHop) return fail(ctx, TVC_ERR_ARG, "tvc_convert_ragged_f32: bad argument (Lmax must be a positive multiple of 480)");
    if (N < 4) return fail(ctx, TVC_ERR_ARG, "tvc_convert_ragged_f32: index needs at least k=4 vectors");
    TVC_CHECK(blob_check(ctx, (hipStream_t)stream, prepared, N, "tvc_convert_ragged_f32"));
    TVC_CHECK(draw_under_capture(ctx, (hipStream_t)stream, noise_angle, "tvc_convert_ragged_f32"));
    TVC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    std::vector<RagBatchPlan> batches;
    TVC_CHECK(ragged_split(ctx, ctx->rag_batch_frames, B, Lmax, lens, &batches));
    size_t bytes = 0;
    TVC_CHECK(ragged_batch_bytes(ctx, batches, Lmax, N, &bytes, noise_angle ? 1 : 0));      // one dry walk per batch: host time on the launch path
    if (bytes > ws_bytes) return fail(ctx, TVC_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", bytes, ws_bytes);
    // the whole padded output is cleared once: every kernel writes its utterance's own samples only
    TVC_HIP(ctx, hipMemsetAsync(wave, 0, (size_t)B * Lmax * sizeof(float), s));
    for (auto& p : batches) {
        Ws ws(wsp, bytes, false);
        TVC_CHECK(ragged_batch(ctx, s, ws, false, p, wav, Lmax, prepared, N, pitch_shift, noise_angle, seed, wave));
    }
    return TVC_OK;
}

int tvc_profile_enable(tvc_ctx* ctx, int on) {
    if (!ctx) return TVC_ERR_ARG;
    ctx->profiling = on < 0 ? 0 : (on > 2 ? 1 : on);
    return TVC_OK;
}

// Synchronises the recorded events and writes "name=ms;name=ms;..." (durations summed per region
// name since the last read) into buf.
int tvc_profile_read(tvc_ctx* ctx, char* buf, size_t buf_bytes) {
    if (!ctx || !buf || buf_bytes < 2) return TVC_ERR_ARG;
    std::vector<std::pair<std::string, double>> agg;
    for (auto& r : ctx->regions) {
        float ms = 0.f;
        if (r.a && r.b && hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            bool found = false;
            for (auto& a : agg)
                if (a.first == r.name) {
                    a.second += ms;
                    found = true;
                }
            if (!found) agg.push_back({r.name, ms});
        }
        if (r.a) ctx->event_pool.push_back(r.a);
        if (r.b) ctx->event_pool.push_back(r.b);
    }
    ctx->regions.clear();
    std::string out;
    for (auto& a : agg) {
        char tmp[160];
        snprintf(tmp, sizeof(tmp), "%s=%.6f;", a.first.c_str(), a.second);
        out += tmp;
    }
    snprintf(buf, buf_bytes, "%s", out.c_str());
    return TVC_OK;
}

int tvc_sola_f32(tvc_ctx* ctx, void* stream, const float* y, float* sola_buf, const float* fade_in, float* out,
                 int32_t* shift_out, int S, int64_t Ly, int block, int use_phase_vocoder) {
    if (!ctx) return TVC_ERR_ARG;
    if (!y || !sola_buf || !fade_in || !out || S <= 0 || block <= 0 || Ly < block + kSolaCross + kSolaSearch + kSolaDelay)
        return fail(ctx, TVC_ERR_ARG, "tvc_sola_f32: bad argument (Ly must cover block+1920+1920+3840)");
    TVC_HIP(ctx, hipSetDevice(ctx->device));
    return run_sola(ctx, (hipStream_t)stream, y, sola_buf, fade_in, out, shift_out, S, Ly, block, use_phase_vocoder);
}

int tvc_stream_push_f32(tvc_ctx* ctx, void* stream, float* buf, const float* blocks, int S, int64_t n, int block) {
    if (!ctx) return TVC_ERR_ARG;
    if (!buf || !blocks || S <= 0 || block <= 0 || n < block || n > 32768) return fail(ctx, TVC_ERR_ARG, "tvc_stream_push_f32: bad argument (block <= n <= 32768)");
    TVC_HIP(ctx, hipSetDevice(ctx->device));
    return run_stream_push(ctx, (hipStream_t)stream, buf, blocks, S, (int)n, block);
}

}  // extern "C"
