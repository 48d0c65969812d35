// eval_engine.hip -- host side of the evaluation kernel: run slots, band partition, table upload, launches.
#include "build_knobs.h"
#include "engine.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

int AmtGpuContext::prof_id(const char* name)
{
    for (size_t i = 0; i < prof_names.size(); ++i) if (prof_names[i] == name) return (int)i;
    prof_names.push_back(name);
    prof_ms.push_back(0.0);
    prof_calls.push_back(0);
    return (int)prof_names.size() - 1;
}
hipEvent_t AmtGpuContext::prof_event()
{
    if (!prof_pool.empty()) { hipEvent_t e = prof_pool.back(); prof_pool.pop_back(); return e; }
    hipEvent_t e;
    AMT_HIP(hipEventCreate(&e));
    return e;
}
int AmtGpuContext::prof_begin(const char* name)
{
    if (!profiling) return -1;
    if (prof_spans.size() >= 4096) prof_resolve();
    ProfSpan sp{prof_id(name), prof_event(), prof_event()};
    AMT_HIP(hipEventRecord(sp.a, stream));
    prof_spans.push_back(sp);
    return (int)prof_spans.size() - 1;
}
void AmtGpuContext::prof_end(int span)
{
    if (span < 0) return;
    AMT_HIP(hipEventRecord(prof_spans[span].b, stream));
}
void AmtGpuContext::prof_resolve()
{
    for (auto& sp : prof_spans) {
        AMT_HIP(hipEventSynchronize(sp.b));
        float ms = 0;
        AMT_HIP(hipEventElapsedTime(&ms, sp.a, sp.b));
        prof_ms[sp.name] += ms;
        prof_calls[sp.name] += 1;
        prof_pool.push_back(sp.a);
        prof_pool.push_back(sp.b);
    }
    prof_spans.clear();
}

namespace amt {

namespace {
int lds_pitch(int w) { return ((w + 31) & ~31) + 8; }

// The composite of a flat level c with the logo, (8 c - 255 b) / a (or 8 c where a <= 0: AddLogo, LogoScan.hpp:320-333), is affine in c
// at every pixel, and CalcCorrelation5x5 is linear in its window: a mask pixel's response on level c is P + Q c in real arithmetic, and
// the reference's table holds |that| with the rounding of its fp32 evaluation.  Least-squares line through the 32 signed responses.
void fit_response(const float* resp32, float& P, float& Q)
{
    double sy = 0, sxy = 0;
    for (int c = 0; c < 32; ++c) { sy += resp32[c]; sxy += (c - 15.5) * resp32[c]; }
    const double q = sxy / 2728.0;                      // sum (c - 15.5)^2, c = 0..31
    Q = (float)q;
    P = (float)(sy / 32.0 - q * 15.5);
}
// |P + Q c| as the linear kernel forms it: one fp32 FMA of the bin number
float formula_response(float P, float Q, int c) { return std::fabs(std::fmaf(Q, (float)c, P)); }
}

EvalEngine::EvalEngine(AmtGpuContext* ctx, std::vector<EvalLogoSpec> specs, std::vector<float> fades, bool take_abs,
                       int out_frame_stride, const char* prof_name)
    : ctx_(ctx), specs_(std::move(specs)), fades_(std::move(fades)), take_abs_(take_abs), out_frame_stride_(out_frame_stride),
      prof_name_(prof_name)
{
    ctx_->bind();
#ifdef AMT_EXPERIMENT
    if (const char* e = std::getenv("AMTGPU_G")) group_frames_ = std::atoi(e);       // instrumented builds only
#endif
    constexpr int kPlaneCapMax = kEvalThreads * kEvalStage;                            // floats per LDS plane
    const int nl = (int)specs_.size();
    std::vector<EvalLogoDev> hl(nl);
    d_a_.resize(nl); d_b_.resize(nl); d_scales_.resize(nl); d_kslot_.resize(nl); d_slot2_.resize(nl);
    plane_cap_ = 0;
    for (int i = 0; i < nl; ++i) {
        EvalLogoSpec& S = specs_[i];
        const MaskTables& T = S.tables;
        const int w = S.planes.w, h = S.planes.h;
        const int lp = lds_pitch(w);
        if (w > 0xFFFF || h > 0xFFFF || T.count >= (1 << 24) - kTablePad) throw std::runtime_error("logo too large");   // 24-bit table index math
        const int cpad = std::max(kTablePad, (T.count + kTablePad - 1) / kTablePad * kTablePad);

        // run slots: horizontally adjacent mask pixels (same row, x+1: consecutive in raster order) pair up;
        // one thread evaluates one slot.  slot = (n << 28) | first mask pixel
        std::vector<uint32_t> slots;
        for (int m = 0; m < T.count;) {
            const int n = (m + 1 < T.count && T.pos[m + 1] == T.pos[m] + 1) ? 2 : 1;
            slots.push_back(((uint32_t)n << 28) | (uint32_t)m);
            m += n;
        }
        auto slot_m0 = [&](int s) { return (int)(slots[s] & 0x0FFFFFFFu); };
        auto slot_n = [&](int s) { return (int)(slots[s] >> 28); };
        auto slot_y = [&](int s) { return (int)(T.pos[slot_m0(s)] >> 16); };
        auto slot_x = [&](int s) { return (int)(T.pos[slot_m0(s)] & 0xFFFF); };
        // bands: up to kEvalThreads consecutive slots whose 5x5 windows fit the LDS plane.  While five rows of the WHOLE logo width fit
        // (w <= 576) a band stages whole rows; a wider logo's bands stage only the columns their windows touch -- a band then ends
        // where the raster order wraps to the next row (its bounding box would span the whole width), so wide logos get more and
        // smaller bands but no width limit (the reference takes any even w x h, LogoScan.hpp:69)
        const bool whole_rows = 5 * lp <= kPlaneCapMax;
        const int band0 = (int)bands_.size();
        const int ns = (int)slots.size();
        for (int s = 0; s < ns;) {
            EvalBand B;
            B.logo = i; B.s0 = s;
            const int ytop = slot_y(s) - 2;
            int minx = slot_x(s), maxx = slot_x(s) + slot_n(s) - 1;
            int e = s;
            B.x0 = 0; B.bw = w; B.lp = lp;
            while (e < ns && e - s < kEvalThreads) {
                const int mnx = std::min(minx, slot_x(e)), mxx = std::max(maxx, slot_x(e) + slot_n(e) - 1);
                const int x0 = whole_rows ? 0 : std::max(0, (mnx - 2) & ~3), x1 = whole_rows ? w : std::min(w, mxx + 4);
                const int lpb = whole_rows ? lp : lds_pitch(x1 - x0);
                if ((slot_y(e) + 2 - ytop + 1) * lpb > kPlaneCapMax) break;
                minx = mnx; maxx = mxx;
                B.x0 = x0; B.bw = x1 - x0; B.lp = lpb;
                ++e;
            }
            if (e == s) throw std::runtime_error("logo too wide for the evaluation kernel");      // (one slot's 5 x 6 window always fits)
            B.nslots = e - s;
            B.y0 = ytop;
            B.nrows = slot_y(e - 1) + 2 - ytop + 1;
            B.m0 = slot_m0(s);
            B.npix = slot_m0(e - 1) + (int)(slots[e - 1] >> 28) - B.m0;
            plane_cap_ = std::max(plane_cap_, B.nrows * B.lp);
            bands_.push_back(B);
            s = e;
        }

        // tables, re-laid for the kernel: bins major / mask pixels minor; tap pairs major / slots minor
        std::vector<float2> scales((size_t)kNumBins * cpad, float2{0.0f, 0.0f});
        for (int m = 0; m < T.count; ++m)
            for (int c = 0; c < kNumBins; ++c)
                scales[(size_t)c * cpad + m] = float2{T.scales[((size_t)m * 32 + c) * 2], T.scales[((size_t)m * 32 + c) * 2 + 1]};
        const int spad = std::max(64, (ns + 63) / 64 * 64);
        std::vector<float2> kslot((size_t)25 * spad, float2{0.0f, 0.0f});
        std::vector<uint2> slot2(spad, uint2{0u, 0u});
        for (int bi = band0; bi < (int)bands_.size(); ++bi) {
            const EvalBand& B = bands_[bi];
            for (int sidx = B.s0; sidx < B.s0 + B.nslots; ++sidx) {
                const int m0 = slot_m0(sidx);
                const int n = (int)(slots[sidx] >> 28);
                const int x = (int)(T.pos[m0] & 0xFFFF), y = (int)(T.pos[m0] >> 16);
                slot2[sidx] = uint2{slots[sidx], (uint32_t)((y - 2 - B.y0) * B.lp + (x - 2 - B.x0))};
                const float* k0 = &T.kernels[(size_t)m0 * 25];
                const float* k1 = n > 1 ? &T.kernels[(size_t)(m0 + 1) * 25] : nullptr;
                for (int c = 0; c < 5; ++c)
                    for (int r = 0; r < 5; ++r) {
                        const float a0 = c < 4 ? k0[r * 5 + c + 1] : k0[r * 5];
                        const float a1 = k1 ? (c < 4 ? k1[r * 5 + c] : k1[r * 5 + 4]) : 0.0f;
                        kslot[(size_t)(c * 5 + r) * spad + sidx] = float2{a0, a1};
                    }
            }
        }
        d_a_[i].upload(S.planes.A(0), (size_t)w * h, ctx_->stream);
        d_b_[i].upload(S.planes.B(0), (size_t)w * h, ctx_->stream);
        d_scales_[i].upload(scales, ctx_->stream);
        d_kslot_[i].upload(kslot, ctx_->stream);
        d_slot2_[i].upload(slot2, ctx_->stream);

        EvalLogoDev& D = hl[i];
        D.a = d_a_[i].get(); D.b = d_b_[i].get(); D.scales = d_scales_[i].get();
        D.kslot = d_kslot_[i].get(); D.slot2 = d_slot2_[i].get(); D.nslots_pad = spad;
        D.band0 = band0; D.nbands = (int)bands_.size() - band0;
        D.w = w; D.h = h; D.count = T.count; D.count_pad = cpad;
        D.imgx = S.imgx; D.imgy = S.imgy; D.row0 = S.row0; D.row_step = S.row_step; D.deint = S.deint;
        D.blackScore = T.blackScore;
        D.out_off = S.out_off;
        D.lp = lp;
    }
#ifdef AMT_EXPERIMENT
    if (std::getenv("AMTGPU_VERBOSE")) {
        for (int i = 0; i < nl; ++i)
            fprintf(stderr, "[amtgpu] eval logo %d: %dx%d count=%d bands=%d lp=%d\n", i, hl[i].w, hl[i].h, hl[i].count, hl[i].nbands, hl[i].lp);
        for (const EvalBand& B : bands_)
            fprintf(stderr, "[amtgpu]   band logo=%d slots=%d npix=%d y0=%d nrows=%d x0=%d bw=%d lp=%d\n", B.logo, B.nslots, B.npix, B.y0, B.nrows, B.x0, B.bw, B.lp);
    }
#endif
    d_logos_.upload(hl, ctx_->stream);
    d_bands_.upload(bands_, ctx_->stream);
    d_fades_.upload(fades_, ctx_->stream);
}

double EvalEngine::mask_pixel_evals_per_frame() const
{
    double s = 0;
    for (const auto& sp : specs_) s += (double)sp.tables.count * fades_.size();
    return s;
}

void EvalEngine::run(const void* dY, int64_t frame_stride_bytes, int pitch, int bits, int nframes, float* dout,
                     const int* dframe_map)
{
    if (nframes <= 0 || specs_.empty()) return;
    ctx_->bind();
    const int es = bits <= 8 ? 1 : 2;
    if (frame_stride_bytes % es) throw std::runtime_error("frame stride not a multiple of the sample size");
    // frames per workgroup: amortises the per-band tap loads; keep >= ~2k workgroups per launch
    const int nl = (int)specs_.size();
    int G = group_frames_ > 0 ? group_frames_ : (int)std::max(1LL, std::min(8LL, (long long)nframes * nl / 2048));
    const int nf_all = (int)fades_.size();
    if (pair_eligible() && pair_addressable(pitch * es)) {
        // fades {0, 1}: the window of s and the window of bg are the two blends themselves
        const size_t dot = prof_name_.find('.');
        const int sp = ctx_->prof_begin(("logo_eval_pair_kernel" + (dot == std::string::npos ? std::string() : prof_name_.substr(dot))).c_str());
        const int Gp = group_frames_ > 0 ? group_frames_ : (int)std::max(1LL, std::min((long long)kTileMaxFrames, (long long)nframes * nl / 2048));
        AMT_HIP(launch_logo_eval_pair(ctx_->stream, bits, d_logos_.get(), d_tls_.get(), nl, dY, dframe_map,
                                      frame_stride_bytes / es, pitch, nframes, Gp, dout, out_frame_stride_, take_abs_ ? 1 : 0));
        ctx_->prof_end(sp);
        return;
    }
    for (int f0 = 0; f0 < nf_all; f0 += kEvalMaxFades) {
        const int nf = std::min(kEvalMaxFades, nf_all - f0);
        G = std::max(1, std::min(G, kEvalThreads / nf));
        const int sp = ctx_->prof_begin(prof_name_.c_str());
        AMT_HIP(launch_logo_eval_fused(ctx_->stream, bits, d_logos_.get(), nl, d_bands_.get(), d_fades_.get(), nf, f0, dY, dframe_map,
                                       frame_stride_bytes / es, pitch, nframes, G, dout, out_frame_stride_, take_abs_ ? 1 : 0,
                                       plane_cap_));
        ctx_->prof_end(sp);
    }
}

// The pair kernel evaluates fade 0 on s and fade 1 on bg = a*s + b*maxv directly.  That equals the reference's blend
// fade*bg + (1-fade)*s bit for bit as long as 0*bg == 0, i.e. bg is finite; it also takes the score bin from a window mean that
// it assumes to be below 2^31 (CorrelationScore's (int)avg, LogoScan.hpp:304, is INT_MIN beyond -- exact_math.h score_bin).  Both
// hold when |a| + |b| < 8192 at every pixel (|bg| < 2^30 for samples up to 65535); logos outside that, field logos and logos
// without mask pixels keep the generic kernel.
bool EvalEngine::pair_eligible()
{
    if (pair_state_ >= 0) return pair_state_ == 1;
    pair_state_ = 0;
#ifdef AMT_EXPERIMENT
    if (std::getenv("AMTGPU_NO_PAIR")) return false;       // instrumented builds: time the generic kernel on the scan
#endif
    if (fades_.size() != 2 || fades_[0] != 0.0f || fades_[1] != 1.0f || specs_.empty()) return false;
    for (const EvalLogoSpec& S : specs_) {
        const int w = S.planes.w, h = S.planes.h;
        // (tile units are four columns wide and keep 8-byte alignment: even widths >= 6; 24-bit slot byte offsets)
        if (w < 6 || (w & 1) || h < 5 || S.tables.count <= 0 || S.tables.count >= (1 << 20)) return false;
        if (!S.deint) return false;                             // LogoFrame's logos are deinterlaced ones; field logos keep the generic kernel
        const float* a = S.planes.A(0);
        const float* b = S.planes.B(0);
        for (int p = 0; p < w * h; ++p)
            if (!(std::fabs(a[p]) + std::fabs(b[p]) < 8192.0f)) return false;       // |bg| <= (|a| + |b|) * 65535 < 2^30; NaN fails the comparison
    }
    ensure_tiles();
    pair_state_ = 1;
    return true;
}

// the pair kernel addresses a frame's samples with 32-bit byte offsets
bool EvalEngine::pair_addressable(int pitch_bytes) const
{
    for (const EvalLogoSpec& S : specs_) {
        const long long last = (long long)(S.imgy + S.row0 + (long long)S.planes.h * S.row_step + 1) * pitch_bytes + (long long)(S.imgx + S.planes.w) * 2;
        if (last >= (1LL << 31) || pitch_bytes <= 0) return false;
    }
    return true;
}

// tile plans of the pair kernel (eval_tiles.hpp): per logo the bands, the eight wave tiles of every band, and per slot
// (= lane of a tile) the mask pixel's window offset, taps and scales
void EvalEngine::ensure_tiles()
{
    if (tiles_ready_) return;
    ctx_->bind();
    const int nl = (int)specs_.size();
    std::vector<TileLogoDev> hl(nl);
    d_tkp_.resize(nl); d_tsc_.resize(nl); d_tpq_.resize(nl); d_tinfo_.resize(nl); d_tpos_.resize(nl); d_tlin_.resize(nl); d_tiles_.resize(nl); d_tbands_.resize(nl); d_tlist_.resize(nl);
    for (int i = 0; i < nl; ++i) {
        const EvalLogoSpec& S = specs_[i];
        const MaskTables& T = S.tables;
        const TilePlan P = build_tile_plan(T.pos, T.count, S.planes.w, S.planes.h);
        const size_t ns = (size_t)P.nslots();
        std::vector<float2> kp(13 * ns, float2{0.0f, 0.0f}), sc((size_t)kNumBins * ns, float2{0.0f, 0.0f}), pq(ns, float2{0.0f, 0.0f});
        std::vector<uint32_t> spos(ns, 0u);
        for (size_t s = 0; s < ns; ++s) {
            const int m = P.slot_pixel[s];
            if (m < 0) continue;                               // (an idle lane: zero taps and a zero response -> its terms are 0 / floorResp)
            fit_response(&T.resp[(size_t)m * 32], pq[s].x, pq[s].y);
            spos[s] = T.pos[m];
            const float* k = &T.kernels[(size_t)m * 25];
            float ksum = 0.0f;
            for (int t = 0; t < 25; ++t) ksum += k[t];
            for (int j = 0; j < 13; ++j) kp[(size_t)j * ns + s] = float2{k[2 * j], 2 * j + 1 < 25 ? k[2 * j + 1] : ksum};    // [12].y = sum of the taps (linear kernel)
            for (int c = 0; c < kNumBins; ++c)
                sc[(size_t)c * ns + s] = float2{T.scales[((size_t)m * 32 + c) * 2], T.scales[((size_t)m * 32 + c) * 2 + 1]};
        }
        d_tkp_[i].upload(kp, ctx_->stream);
        d_tsc_[i].upload(sc, ctx_->stream);
        d_tpq_[i].upload(pq, ctx_->stream);
        d_tpos_[i].upload(spos, ctx_->stream);
        // the linear kernel's blob
        const size_t npx = (size_t)S.planes.w * S.planes.h;
        const size_t o_pq = kp.size() * sizeof(float2), o_si = o_pq + pq.size() * sizeof(float2), o_a = o_si + P.sinfo.size() * sizeof(uint32_t),
                     o_b = o_a + npx * sizeof(float), total = o_b + npx * sizeof(float);
        if (total >= ((size_t)1 << 31)) throw std::runtime_error("logo too large");
        std::vector<char> blob(total);
        std::memcpy(blob.data(), kp.data(), o_pq);
        std::memcpy(blob.data() + o_pq, pq.data(), o_si - o_pq);
        std::memcpy(blob.data() + o_si, P.sinfo.data(), o_a - o_si);
        std::memcpy(blob.data() + o_a, S.planes.A(0), npx * sizeof(float));
        std::memcpy(blob.data() + o_b, S.planes.B(0), npx * sizeof(float));
        d_tlin_[i].upload(blob, ctx_->stream);
        d_tinfo_[i].upload(P.sinfo, ctx_->stream);
        d_tiles_[i].upload(P.tiles, ctx_->stream);
        d_tbands_[i].upload(P.bands, ctx_->stream);
        std::vector<int> tlist;
        for (size_t t = 0; t < P.tiles.size(); ++t) if (P.tiles[t].npix > 0) tlist.push_back((int)t);
        d_tlist_[i].upload(tlist, ctx_->stream);
        hl[i] = TileLogoDev{d_tkp_[i].get(), d_tsc_[i].get(), d_tpq_[i].get(), d_tinfo_[i].get(), d_tpos_[i].get(), d_tiles_[i].get(), d_tbands_[i].get(), d_tlist_[i].get(),
                            (int)P.bands.size(), (int)ns, (int)tlist.size(), T.floorResp};
        hl[i].lin = d_tlin_[i].get();
        hl[i].lin_pq = (unsigned)o_pq; hl[i].lin_sinfo = (unsigned)o_si; hl[i].lin_a = (unsigned)o_a; hl[i].lin_b = (unsigned)o_b;
    }
    d_tls_.upload(hl, ctx_->stream);
    tiles_ready_ = true;
}

// ------------------------------------------------------------------------------------------------------------------------
// linear (decision-guarded) mode
// ------------------------------------------------------------------------------------------------------------------------
void EvalEngine::ensure_linear()
{
    if (linear_ready_) return;
    const int nl = (int)specs_.size();
    lin_err_corr_.assign(nl, 0.0); lin_err_sum_.assign(nl, 0.0); lin_err_formula_.assign(nl, 0.0);
    lin_formula_ok_ = true;
    double vunit = 1.0;
    for (int i = 0; i < nl; ++i) {
        const EvalLogoSpec& S = specs_[i];
        const MaskTables& T = S.tables;
        const int w = S.planes.w, h = S.planes.h;
        // ---- error bound of the linear evaluation against the reference's evaluation order (u = 2^-24, v = bound on window values) ----
        // every window value (s, bg, any blend with fade in [0,1] ... fades up to 2 are covered by the factor below) is <= v = vunit*maxv;
        // exact path:  W_i carries 3 roundings, the mean 6 + 1, (W_i - m), the product, the 6-deep sum:   |d corr| <= 27 u v sum|k|
        // linear path: corr = sum k_i w_i - mean * sum k_i as two 13-deep FMA chains (<= 14 u sum|k_i w_i| + 2u for the add and the
        //              mean term, whose own error 7 u v is multiplied by |sum k_i| <= sum|k_i|): 23 u v sum|k|, for s and for bg with
        //              weights 1-f and f; 3 roundings to combine:  26
        //  => |corr_lin - corr_exact| <= 53 u v sum|k_i| (58 below, + 24 for the tap sum formed on the host); the clamp is 1-Lipschitz, so a term moves by <= scale*scale2 times that (+ 2u|t|);
        // two summation orders of the count terms (the reference's sequential one, this kernel's lanes / tiles / waves) differ by
        // <= (count + 32) u sum|t|, and |t| <= scale2.
        const float* a = S.planes.A(0);
        const float* b = S.planes.B(0);
        for (int p = 0; p < w * h; ++p) {
            const double ab = (double)std::fabs(a[p]) + std::fabs(b[p]);
            if (!(ab < 1e30)) throw std::runtime_error("logo coefficients are not finite: the linear mode has no error bound for them");   // (NaN fails the comparison)
            vunit = std::max(vunit, ab);
        }
        // ---- the scale of a term.  The reference looks {scale, scale2} = {1 / r, min(1, r / L)} up by bin, r = |response on that flat level|,
        // L = floorResp, and forms  t = clamp(x * scale, -1, 1) * scale2  (LogoScan.hpp:302-308), which is  clamp(x, -r, r) / max(r, L).
        // The linear kernel forms r' = |fma(Q, bin, P)| from the pixel's fitted line (fit_response) and  t' = med3(x, -r', r') * rcp(max(r', L)):
        // no gather.  Both are odd, monotone and piecewise linear in x with breakpoints at r and r': sup_x |t - t'| is reached at a
        // breakpoint or at infinity, and is evaluated HERE for every (pixel, bin) from the reference's own table floats and the kernel's
        // own fp32 r' (std::fmaf is the device's v_fma_f32) -- no estimate.  v_rcp_f32 is good to 1 ulp and the product rounds once: 4 u t'.
        const double L = T.floorResp;
        if (!(L > 1e-20 && L < 1e30) || T.resp.size() != (size_t)T.count * 32) lin_formula_ok_ = false;
        double ecorr = 0, tsum = 0, eformula = 0;
        for (int m = 0; m < T.count; ++m) {
            double sk = 0;
            for (int t = 0; t < 25; ++t) sk += std::fabs(T.kernels[(size_t)m * 25 + t]);
            double smax = 0, s2max = 0, devmax = 0;
            float Pf = 0.0f, Qf = 0.0f;
            if (lin_formula_ok_) fit_response(&T.resp[(size_t)m * 32], Pf, Qf);
            if (!(std::fabs(Pf) < 1e30f && std::fabs(Qf) < 1e30f)) lin_formula_ok_ = false;
            for (int c = 0; c < kNumBins; ++c) {
                const double sc = std::fabs(T.scales[((size_t)m * 32 + c) * 2]), s2 = std::fabs(T.scales[((size_t)m * 32 + c) * 2 + 1]);
                smax = std::max(smax, sc * s2);
                s2max = std::max(s2max, s2);
                if (!lin_formula_ok_) continue;
                const double rk = formula_response(Pf, Qf, c), dk = std::max(rk, L);
                const auto tref = [&](double x) { return std::min(1.0, x * sc) * s2; };      // x >= 0
                const auto tker = [&](double x) { return std::min(x, rk) / dk; };
                double dev = std::fabs((sc > 0 ? s2 : 0.0) - rk / dk);                        // x -> infinity
                dev = std::max(dev, std::fabs(tref(rk) - tker(rk)));
                if (sc > 0) dev = std::max(dev, std::fabs(tref(1.0 / sc) - tker(1.0 / sc)));
                devmax = std::max(devmax, dev);
                smax = std::max(smax, 1.0 / dk);                                              // the kernel's own slope in x
                s2max = std::max(s2max, rk / dk);
            }
            eformula += devmax;
            // (+ 24 u v sum|k|: the host's fp32 sum of the 25 taps that multiplies the mean, kp[12].y in ensure_tiles, carries up to
            // 24 roundings of partial sums <= sum|k|, times a mean <= v)
            ecorr += smax * sk * (58.0 + 24.0);
            tsum += s2max;
        }
        const double black = std::max(1e-30, (double)std::fabs(T.blackScore));
        lin_err_corr_[i] = ecorr / black;                                       // x u x v
        lin_err_sum_[i] = ((double)T.count + 52.0) * tsum / black + 8.0;        // x u  (+ the final division / abs, results are O(1); + 4 u |t'| for rcp and its product; + 8 u for a listed pair's correction t(exact bin) - t(tentative bin))
        lin_err_formula_[i] = eformula / black;
    }
    // v, the bound on window values in units of maxv: a blend with fades in [0, 1] is convex, so max(1, |a| + |b|) holds; fades
    // outside that range (ReMakeLogo's reach 1.9: |f| + |1 - f| <= 3) get a factor 2
    bool unit_fades = true;
    for (float f : fades_) unit_fades = unit_fades && f >= 0.0f && f <= 1.0f;
    vmax_unit_ = (float)(vunit * (unit_fades ? 1.0 : 2.0));
    linear_ready_ = true;
}

// the tile kernels address a frame's samples with 32-bit byte offsets and the slot tables with 24-bit ones
bool EvalEngine::tiles_usable(int pitch_bytes) const
{
    for (const EvalLogoSpec& S : specs_) {
        const int w = S.planes.w, h = S.planes.h;
        // (24-bit slot byte offsets; the linear kernel's list entries carry the slot number -- at most 1.1 x count -- in 20 bits)
        if (w < 6 || (w & 1) || h < 5 || S.tables.count <= 0 || S.tables.count >= 900000) return false;
    }
    return pair_addressable(pitch_bytes);
}

float EvalEngine::linear_error_bound(int logo, int bits) const
{
    const_cast<EvalEngine*>(this)->ensure_linear();
    const double u = 1.0 / 16777216.0, maxv = (double)((1 << bits) - 1);
    return (float)(1.25 * (u * (lin_err_corr_[logo] * vmax_unit_ * maxv + lin_err_sum_[logo]) + lin_err_formula_[logo]));
}

void EvalEngine::run_linear(const void* dY, int64_t frame_stride_bytes, int pitch, int bits, int nframes, float* dout, const int* dframe_map,
                            uint8_t* dforce)
{
    if (nframes <= 0 || specs_.empty()) return;
    const int es = bits <= 8 ? 1 : 2;
    // The linear kernel evaluates AMTAnalyzeLogo's 11 fades on tile plans.  A logo without mask pixels (maskratio 0: results are
    // 0 / blackScore = 0 / 0, as the reference's), odd or tiny shapes and any other fade list keep the exact kernel.
    // the linear kernel is written for AMTAnalyzeLogo's fades: eleven, from exactly 0 to exactly 1
    if ((int)fades_.size() != 11 || fades_.front() != 0.0f || fades_.back() != 1.0f || !tiles_usable(pitch * es)) { run(dY, frame_stride_bytes, pitch, bits, nframes, dout, dframe_map); return; }
    ensure_linear();
    if (!lin_formula_ok_) { run(dY, frame_stride_bytes, pitch, bits, nframes, dout, dframe_map); return; }      // (a degenerate logo: the exact kernel)
    ensure_tiles();
    ctx_->bind();
    if (frame_stride_bytes % es) throw std::runtime_error("frame stride not a multiple of the sample size");
    const int nl = (int)specs_.size();
    // The interpolated mean is within 19 u v of the exactly evaluated one (7 + 7 roundings of the two means, 3 to combine them, 9 on
    // the exact side... see ensure_linear), v = a bound on the window values: fades in [0, 1] blend convexly, so v = max(1, |a| + |b|) * maxv
    // (vmax_unit_, ensure_linear).  The kernel compares in fixed point -- mean * 2^qlog2 with v * 2^qlog2 < 2^30 -- through three more
    // fp32 roundings of values below v * 2^qlog2: 22 u v in all.
    const double vwin = (double)vmax_unit_ * (double)((1 << bits) - 1);
    const float bin_eps = (float)(22.0 / 16777216.0 * vwin * 1.0001);
    int qlog2 = 24;
    while (qlog2 > 4 && std::ldexp(vwin, qlog2) >= 1073741824.0) --qlog2;
    if (!(std::ldexp(vwin, qlog2) < 1073741824.0)) { run(dY, frame_stride_bytes, pitch, bits, nframes, dout, dframe_map); return; }   // (absurd coefficients: the exact kernel)
    // frames per workgroup: four workgroups of four waves share a CU's 160 KB of LDS -- 16 KB of tile planes, 4 x 8 B x the list's entries,
    // 3 KB of running sums per frame
    const int qcap = std::max(16, std::min(640, lin_queue_));
    const int g_fit = (int)((160 * 1024 / 4 - 16 * 1024 - 4 * 8 * qcap - 16) / (4 * 768));
    if (g_fit < 1) { run(dY, frame_stride_bytes, pitch, bits, nframes, dout, dframe_map); return; }
    const int gmax = std::min(g_fit, bits > 8 ? kLinMaxFrames16 : kLinMaxFrames);
    const long long wgs_min = bits > 8 ? AMT_LIN_WGS_MIN16 : 2048;
    const int G = std::min(gmax, group_frames_ > 0 ? group_frames_ : (int)std::max(1LL, std::min(16LL, (long long)nframes * nl / wgs_min)));
    const size_t dot = prof_name_.find('.');
    const int sp = ctx_->prof_begin(("logo_eval_linear_kernel" + (dot == std::string::npos ? std::string() : prof_name_.substr(dot))).c_str());
    AMT_HIP(launch_logo_eval_linear(ctx_->stream, bits, d_logos_.get(), d_tls_.get(), nl, d_fades_.get(), 11, 0, dY, dframe_map,
                                    frame_stride_bytes / es, pitch, nframes, G, dout, out_frame_stride_, take_abs_ ? 1 : 0, bin_eps, qlog2,
                                    qcap, dforce));
    ctx_->prof_end(sp);
}

#ifndef AMT_LISTED_FADE_CHUNK
#define AMT_LISTED_FADE_CHUNK 1      /* fades per workgroup of the listed re-evaluation (0: all of them, round 3's form) */
#endif
void EvalEngine::run_listed(const void* dY, int64_t frame_stride_bytes, int pitch, int bits, int max_frames, const int* dlist,
                            const int* dcount, float* dout)
{
    if (max_frames <= 0 || specs_.empty()) return;
    ctx_->bind();
    const int es = bits <= 8 ? 1 : 2;
    if (frame_stride_bytes % es) throw std::runtime_error("frame stride not a multiple of the sample size");
    const int nl = (int)specs_.size();
    const int nf_all = (int)fades_.size();
    for (int f0 = 0; f0 < nf_all; f0 += kEvalMaxFades) {
        const int nf = std::min(kEvalMaxFades, nf_all - f0);
        const int G = 1;      // a handful of frames is expected: one per workgroup, so that the pass lasts one frame's walk over the bands
        const int sp = ctx_->prof_begin((prof_name_ + "_refine").c_str());
        AMT_HIP(launch_logo_eval_fused(ctx_->stream, bits, d_logos_.get(), nl, d_bands_.get(), d_fades_.get(), nf, f0, dY, dlist,
                                       frame_stride_bytes / es, pitch, max_frames, G, dout, out_frame_stride_, take_abs_ ? 1 : 0,
                                       plane_cap_, dcount, 1, AMT_LISTED_FADE_CHUNK));
        ctx_->prof_end(sp);
    }
}

} // namespace amt
