// stats_decisions.cpp -- see stats_decisions.hpp
#include "build_knobs.h"
#include "stats_decisions.hpp"

#include <algorithm>

#include "host_parallel.hpp"

namespace amt {

namespace {
enum { W_DIFF_TOP = 0, W_DIFF_BOT = 1, W_VERT = 2, W_COMB = 3, W_COMB_PREV = 4, W_SUM = 5, W_VERT_PREV = 6, WORDS = 8 };
inline uint64_t m(const uint64_t* metrics, int n, int w) { return metrics[(size_t)n * WORDS + w]; }
}

// frame n starts a new scene when its field-difference energy is at least 4 per pixel on average and more
// than three times the median of the previous 15 frames' energies (frames 1 .. n-1 near the clip's start; the
// median of s values = the one at index s / 2 in ascending order).  The window is kept sorted as it slides -- the
// decisions are replicated on every rank over the WHOLE clip, so they cost O(1) per frame.
std::vector<int> scene_changes(const uint64_t* metrics, int nframes, int width, int height)
{
    const uint64_t floor_energy = (uint64_t)width * height * 4;
    auto energy = [&](int k) { return m(metrics, k, W_DIFF_TOP) + m(metrics, k, W_DIFF_BOT); };
    // frames [lo, hi) of the clip: the window is primed from the (up to) 15 frames before lo, so a range needs nothing from its neighbours
    auto range = [&](int lo, int hi, std::vector<int>& out) {
        uint64_t win[16];
        int s = 0;                               // win[0..s) = the energies of frames max(1, n-15) .. n-1, ascending
        for (int k = std::max(1, lo - 15); k < lo; ++k) win[s++] = energy(k);
        std::sort(win, win + s);
        for (int n = std::max(1, lo); n < hi; ++n) {
            const uint64_t e = energy(n);
            const uint64_t med = s ? win[s / 2] : 0;
            if (e >= floor_energy && e > 3 * med) out.push_back(n);
            if (s == 15) {                       // frame n-15 leaves
                const uint64_t old = energy(n - 15);
                int i = 0;
                while (i + 1 < s && win[i] != old) ++i;
                for (; i + 1 < s; ++i) win[i] = win[i + 1];
                --s;
            }
            int i = s++;                         // frame n enters
            for (; i > 0 && win[i - 1] > e; --i) win[i] = win[i - 1];
            win[i] = e;
        }
    };
    const int parts = parallel_parts(nframes);
    std::vector<std::vector<int>> part(parts);
    parallel_ranges(nframes, parts, [&](int lo, int hi, int p) { range(lo, hi, part[p]); });
    std::vector<int> out;
    for (auto& v : part) out.insert(out.end(), v.begin(), v.end());
    return out;
}

// Field matching per frame from the two weaves' combing energy (minus nothing: both share the picture's
// own vertical detail): 'C' the frame's own fields belong together (COMB*1.5 < COMB_PREV), 'P' its top
// field belongs with the previous bottom field (COMB_PREV*1.5 < COMB), 'B' undecided.  3:2 pulldown gives
// C C P P x every five frames, progressive 30p gives all C, interlaced video all B (while moving).
// Frame n is classified from the 10-frame window [n-4, n+6) cut to the clip.  The window's counts slide with it: the 3:2 hits
// are kept per cycle offset q = (position of frame 0 in the cycle), which does not move with the window -- the phase ph of the
// window's first frame a is q = (ph - a) mod 5.
void classify_cadence(const uint64_t* metrics, int nframes, int width, int height, uint8_t* cadence, uint8_t* phase)
{
    std::vector<char> code(nframes, 'B');
    std::vector<uint64_t> motion_of(nframes);
    const uint64_t still = (uint64_t)width * height / 2;      // < 0.5 per pixel of field difference: nothing moves
    constexpr uint8_t kStill = 0xFF;                           // first pass: "nothing moves in this frame's window"
    static_assert(kStill != kCadence60i && kStill != kCadence24p && kStill != kCadence30p, "the sentinel must not be a cadence");
    // the first pass's classes live in buffers of this function: the caller's arrays only ever receive finished decisions (a worker
    // that throws -- an allocation, a thread that cannot be started -- leaves them untouched, never holding the sentinel)
    std::vector<uint8_t> cls1(nframes), ph1(nframes);
    // First pass, frame ranges in parallel: everything that depends on the frame's window only.  Second pass, in order: a still window
    // keeps the cadence of the frame before it (and advances its 3:2 phase).
    parallel_ranges(nframes, parallel_parts(nframes), [&](int lo, int hi, int) {
        for (int n = lo; n < hi; ++n) {
            const uint64_t c0 = m(metrics, n, W_COMB), c1 = m(metrics, n, W_COMB_PREV);
            if (c0 * 3 < c1 * 2) code[n] = 'C';
            else if (c1 * 3 < c0 * 2) code[n] = 'P';
            motion_of[n] = m(metrics, n, W_DIFF_TOP) + m(metrics, n, W_DIFF_BOT);
        }
    });
    parallel_ranges(nframes, parallel_parts(nframes), [&](int lo, int hi, int) {
        int hitq[5] = {0, 0, 0, 0, 0}, nC = 0, nDecided = 0;
        auto slide = [&](int k, int sign) {                        // frame k enters (+1) or leaves (-1) the window
            nC += sign * (code[k] == 'C');
            nDecided += sign * (code[k] != 'B');
            if (code[k] == 'B') return;
            // C counts where (k + q) % 5 is 0 or 1, P where it is 2 or 3: two offsets q each
            const int r = k % 5, q0 = code[k] == 'C' ? 5 - r : 7 - r;
            hitq[q0 % 5] += sign;
            hitq[(q0 + 1) % 5] += sign;
        };
        int wa = std::max(0, lo - 4), wb = wa;                     // the counts cover [wa, wb)
        uint64_t motion = 0;
        int motion_from = -1;                                      // the latest frame of the window that holds `motion`
        for (int n = lo; n < hi; ++n) {
            const int a = std::max(0, n - 4), b = std::min(nframes, n + 6);      // 10-frame window
            while (wb < b) slide(wb++, +1);
            while (wa < a) slide(wa++, -1);
            // the window's largest motion: recomputed only when the frame that left held it or the window is still growing
            if (n == lo || motion_from < a) {
                motion = 0;
                for (int k = a; k < b; ++k) if (motion_of[k] >= motion) { motion = motion_of[k]; motion_from = k; }
            } else if (b > 0 && motion_of[b - 1] >= motion) { motion = motion_of[b - 1]; motion_from = b - 1; }
            int best = -1, bestPhase = 0;
            const int a5 = a % 5;
            for (int ph = 0; ph < 5; ++ph) {          // ph = position of frame `a` in the cycle
                const int q = ph - a5;
                const int hit = hitq[q < 0 ? q + 5 : q];
                if (hit > best) { best = hit; bestPhase = ph; }
            }
            const int span = b - a;
            uint8_t cls, ph = 0;
            if (motion < still) { cls = kStill; }                        // nothing moves: decided in the second pass
            else if (nDecided * 2 < span) { cls = kCadence60i; }         // moving, but neither weave is clean: true interlaced
            else if (best * 10 >= span * 7) { cls = kCadence24p; ph = (uint8_t)((n - a + bestPhase) % 5); }
            else if (nC * 10 >= span * 7) { cls = kCadence30p; }
            else if (nDecided * 10 >= span * 7 && best * 10 >= span * 5) { cls = kCadence24p; ph = (uint8_t)((n - a + bestPhase) % 5); }
            else { cls = kCadence60i; }
            cls1[n] = cls;
            ph1[n] = ph;
        }
    });
    uint8_t last = kCadence60i, lastPhase = 0;
    for (int n = 0; n < nframes; ++n) {
        const bool keep = cls1[n] == kStill;
        cadence[n] = last = keep ? last : cls1[n];
        phase[n] = lastPhase = keep ? (last == kCadence24p ? (uint8_t)((lastPhase + 1) % 5) : (uint8_t)0) : ph1[n];
    }
}

std::vector<int> cadence_durations(const uint8_t* cadence, const uint8_t* phase, int nframes)
{
    std::vector<int> d;
    for (int n = 0; n < nframes;) {
        if (cadence[n] == kCadence24p && phase[n] == 0 && n + 5 <= nframes) {
            bool whole = true;
            for (int k = 1; k < 5; ++k) whole = whole && cadence[n + k] == kCadence24p && phase[n + k] == k;
            if (whole) { d.insert(d.end(), {2, 3, 2, 3}); n += 5; continue; }
        }
        if (cadence[n] == kCadence60i) { d.push_back(1); d.push_back(1); }
        else d.push_back(2);
        ++n;
    }
    return d;
}

} // namespace amt
