// stats_decisions.cpp -- see stats_decisions.hpp
#include "stats_decisions.hpp"

#include <algorithm>

namespace amt {

namespace {
enum { W_DIFF_TOP = 0, W_DIFF_BOT = 1, W_VERT = 2, W_COMB = 3, W_COMB_PREV = 4, W_SUM = 5, W_VERT_PREV = 6, WORDS = 8 };
inline uint64_t m(const uint64_t* metrics, int n, int w) { return metrics[(size_t)n * WORDS + w]; }
}

// frame n starts a new scene when its field-difference energy is at least 4 per pixel on average and more
// than three times the median of the previous 15 frames' energies
std::vector<int> scene_changes(const uint64_t* metrics, int nframes, int width, int height)
{
    std::vector<int> out;
    const uint64_t floor_energy = (uint64_t)width * height * 4;
    std::vector<uint64_t> hist;
    for (int n = 1; n < nframes; ++n) {
        const uint64_t e = m(metrics, n, W_DIFF_TOP) + m(metrics, n, W_DIFF_BOT);
        const int k0 = std::max(1, n - 15);
        hist.clear();
        for (int k = k0; k < n; ++k) hist.push_back(m(metrics, k, W_DIFF_TOP) + m(metrics, k, W_DIFF_BOT));
        uint64_t med = 0;
        if (!hist.empty()) {
            std::nth_element(hist.begin(), hist.begin() + hist.size() / 2, hist.end());
            med = hist[hist.size() / 2];
        }
        if (e >= floor_energy && e > 3 * med) out.push_back(n);
    }
    return out;
}

// Field matching per frame from the two weaves' combing energy (minus nothing: both share the picture's
// own vertical detail): 'C' the frame's own fields belong together (COMB*1.5 < COMB_PREV), 'P' its top
// field belongs with the previous bottom field (COMB_PREV*1.5 < COMB), 'B' undecided.  3:2 pulldown gives
// C C P P x every five frames, progressive 30p gives all C, interlaced video all B (while moving).
void classify_cadence(const uint64_t* metrics, int nframes, int width, int height, uint8_t* cadence, uint8_t* phase)
{
    std::vector<char> code(nframes, 'B');
    for (int n = 0; n < nframes; ++n) {
        const uint64_t c0 = m(metrics, n, W_COMB), c1 = m(metrics, n, W_COMB_PREV);
        if (c0 * 3 < c1 * 2) code[n] = 'C';
        else if (c1 * 3 < c0 * 2) code[n] = 'P';
    }
    const uint64_t still = (uint64_t)width * height / 2;      // < 0.5 per pixel of field difference: nothing moves
    uint8_t last = kCadence60i, lastPhase = 0;
    for (int n = 0; n < nframes; ++n) {
        const int a = std::max(0, n - 4), b = std::min(nframes, n + 6);      // 10-frame window
        int best = -1, bestPhase = 0, nC = 0, nDecided = 0;
        uint64_t motion = 0;
        for (int k = a; k < b; ++k) {
            nC += code[k] == 'C';
            nDecided += code[k] != 'B';
            motion = std::max(motion, m(metrics, k, W_DIFF_TOP) + m(metrics, k, W_DIFF_BOT));
        }
        for (int ph = 0; ph < 5; ++ph) {          // ph = position of frame `a` in the cycle
            int hit = 0;
            for (int k = a; k < b; ++k) {
                const int pos = (k - a + ph) % 5;
                if (pos <= 1) hit += code[k] == 'C';
                else if (pos <= 3) hit += code[k] == 'P';
            }
            if (hit > best) { best = hit; bestPhase = ph; }
        }
        const int span = b - a;
        uint8_t cls, ph = 0;
        if (motion < still) { cls = last; ph = last == kCadence24p ? (uint8_t)((lastPhase + 1) % 5) : 0; }   // nothing moves: keep
        else if (nDecided * 2 < span) { cls = kCadence60i; }         // moving, but neither weave is clean: true interlaced
        else if (best * 10 >= span * 7) { cls = kCadence24p; ph = (uint8_t)((n - a + bestPhase) % 5); }
        else if (nC * 10 >= span * 7) { cls = kCadence30p; }
        else if (nDecided * 10 >= span * 7 && best * 10 >= span * 5) { cls = kCadence24p; ph = (uint8_t)((n - a + bestPhase) % 5); }
        else { cls = kCadence60i; }
        cadence[n] = cls;
        phase[n] = ph;
        last = cls;
        lastPhase = ph;
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
