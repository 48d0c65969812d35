// eval_engine.hip -- host side of the evaluation kernel: run slots, band partition, table upload, launches.
#include "engine.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

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
        if (5 * lp > kPlaneCapMax) throw std::runtime_error("logo too wide for the evaluation kernel");
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
        auto slot_y = [&](int s) { return (int)(T.pos[slot_m0(s)] >> 16); };
        // bands: up to kEvalThreads consecutive slots whose 5x5 windows fit the LDS plane
        const int band0 = (int)bands_.size();
        const int ns = (int)slots.size();
        for (int s = 0; s < ns;) {
            EvalBand B;
            B.logo = i; B.s0 = s;
            const int ytop = slot_y(s) - 2;
            int e = s;
            while (e < ns && e - s < kEvalThreads && (slot_y(e) + 2 - ytop + 1) * lp <= kPlaneCapMax) ++e;
            B.nslots = e - s;
            B.y0 = ytop;
            B.nrows = slot_y(e - 1) + 2 - ytop + 1;
            B.m0 = slot_m0(s);
            B.npix = slot_m0(e - 1) + (int)(slots[e - 1] >> 28) - B.m0;
            plane_cap_ = std::max(plane_cap_, B.nrows * lp);
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
                slot2[sidx] = uint2{slots[sidx], (uint32_t)((y - 2 - B.y0) * lp + (x - 2))};
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
            fprintf(stderr, "[amtgpu]   band logo=%d slots=%d npix=%d y0=%d nrows=%d\n", B.logo, B.nslots, B.npix, B.y0, B.nrows);
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

} // namespace amt
