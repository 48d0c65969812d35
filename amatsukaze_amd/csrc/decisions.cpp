// decisions.cpp -- see decisions.hpp.  fp32 sums keep the reference's left-to-right order.
#include "build_knobs.h"
#include "decisions.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <limits>
#include <stdexcept>

#include "host_parallel.hpp"

namespace amt {

namespace {
constexpr float kUnknownBelow = 0.2f;      // |score| below this is "unknown" (LogoScan.hpp:1538)
}

LogoSelection select_logo(const float* evals, int numFrames, int numLogos, int numCandidates)
{
    if (numCandidates < 0) numCandidates = numLogos;
    LogoSelection sel;
    if (numCandidates <= 0) return sel;
    // per candidate: a frame counts for logo i when the logo is seen (corr0) and erasing it leaves little (corr1); the residue is an
    // fp32 sum in frame order (the reference's).  Candidates are independent: one thread each on long clips.
    std::vector<int> hitsOf(numCandidates);
    std::vector<float> scoreOf(numCandidates);
    auto one = [&](int i) {
        int hits = 0;
        float residue = 0.0f;
        for (int n = 0; n < numFrames; ++n) {
            const float* r = evals + ((size_t)n * numLogos + i) * 2;
            if (r[0] > kUnknownBelow && std::abs(r[1]) < kUnknownBelow) { ++hits; residue += std::abs(r[1]); }
        }
        hitsOf[i] = hits;
        scoreOf[i] = hits == 0 ? std::numeric_limits<float>::infinity() : (residue / hits) * (numFrames / (float)hits);
    };
    const int parts = std::min(numCandidates, parallel_parts(numFrames));   // never more threads than amtgpu_host_set_parallelism allows
    if (parts > 1) {
        // (ranges of candidates: parallel_ranges joins its threads and forwards exceptions)
        parallel_ranges(numCandidates, parts, [&](int lo, int hi, int) { for (int i = lo; i < hi; ++i) one(i); });
    } else {
        for (int i = 0; i < numCandidates; ++i) one(i);
    }
    float bestScore = 0;
    int bestHits = 0;
    for (int i = 0; i < numCandidates; ++i)
        if (i == 0 || scoreOf[i] < bestScore) { bestScore = scoreOf[i]; sel.bestLogo = i; bestHits = hitsOf[i]; }
    if (sel.bestLogo >= 0) sel.logoRatio = (float)bestHits / numFrames;
    return sel;
}

std::string logoframe_text(const float* evals, int numFrames, int numLogos, int logoIndex, int fpsNum, int fpsDen)
{
    const int fps = (int)std::round((float)fpsNum / fpsDen);
    const int halfAvg = int(fps * 1.0f / 2 + 0.5f);       // +-0.5 s moving average / min-max window
    const int avgLen = 2 * halfAvg + 1;
    const int halfMed = int(fps * 0.5f / 2 + 0.5f);       // 0.5 s median
    const int N = numFrames;
    if (N <= 0) return std::string();

    // per-frame signed evidence; frames outside the clip repeat the end values (a padded copy: no clamps in the loops below)
    const int pad = std::max(halfAvg, halfMed) + 1;
    std::vector<float> evp((size_t)N + 2 * pad);
    for (int j = 0; j < N + 2 * pad; ++j) {
        const int n = std::max(0, std::min(N - 1, j - pad));
        const float* r = evals + ((size_t)n * numLogos + logoIndex) * 2;
        evp[j] = std::max(0.0f, r[0]) + std::min(0.0f, r[1]);
    }
    const float* const ev = evp.data() + pad;                  // ev[i], -pad <= i < N + pad

    // The selection and the text are replicated on every rank over the WHOLE clip.  The window passes are local
    // -- a frame's state and median depend on +-halfAvg frames of evidence -- so frame ranges run on a few threads; inside a range they
    // run window-offset outermost, frames innermost (vector loops over frames; every frame still sees its window's values in the
    // reference's order -- the fp32 mean is a left-to-right sum), and the median slides a sorted window primed at the range's start.
    std::vector<int> state(N);
    std::vector<float> smooth(N);
    parallel_ranges(N, parallel_parts(N), [&](int lo, int hi, int) {
        const int M = hi - lo;
        std::vector<float> before(M), after(M), sum(M, 0.0f);
        const float* const e = ev + lo;                               // e[i] = ev[lo + i]
        for (int i = 0; i < M; ++i) { before[i] = e[i - halfAvg]; after[i] = e[i + 1]; }
        for (int d = 1; d < halfAvg; ++d)
            for (int i = 0; i < M; ++i) {
                before[i] = std::max(before[i], e[i - halfAvg + d]);
                after[i] = std::max(after[i], e[i + 1 + d]);
            }
        for (int d = -halfAvg; d <= halfAvg; ++d)
            for (int i = 0; i < M; ++i) sum[i] += e[i + d];
        for (int i = 0; i < M; ++i) {
            const float mm = std::min(before[i], after[i]);
            const int byMinMax = (std::abs(mm) < 0.5f) ? 1 : (mm < 0.0f) ? 0 : 2;
            const float mean = sum[i] / avgLen;
            const int byMean = (std::abs(mean) < kUnknownBelow) ? 1 : (mean < 0.0f) ? 0 : 2;
            state[lo + i] = (byMinMax == byMean) ? byMinMax : 1;
        }
        // The median slides a window kept in ascending order.  NaN evidence (corr0 = +inf with corr1 = -inf: a logo whose blackScore
        // is 0) has no place in `<`: the reference's std::sort over it is undefined behaviour (LogoScan.hpp:1745-1747), so the
        // order is fixed here -- NaN after every number, like the checker -- and every search below is bounded.
        const int K = 2 * halfMed + 1;
        auto lt = [](float a, float b) { return a < b || (b != b && a == a); };
        auto same = [](float a, float b) { return a == b || (a != a && b != b); };
        std::vector<float> win(e - halfMed, e - halfMed + K);         // the window of the range's first frame
        std::sort(win.begin(), win.end(), lt);
        for (int i = 0; i < M; ++i) {
            smooth[lo + i] = win[halfMed];
            if (i + 1 == M) break;
            const float out = e[i - halfMed], in = e[i + 1 + halfMed];
            int k = 0;
            while (k + 1 < K && !same(win[k], out)) ++k;
            for (; k + 1 < K && lt(win[k + 1], in); ++k) win[k] = win[k + 1];    // the gap moves up ...
            for (; k > 0 && lt(in, win[k - 1]); --k) win[k] = win[k - 1];        // ... or down to where `in` belongs
            win[k] = in;
        }
    });

    // unknown runs adopt their neighbours' state when both sides agree (outside the clip counts as off)
    for (int i = 0; i < N;) {
        if (state[i] != 1) { ++i; continue; }
        int e = i;
        while (e < N && state[e] == 1) ++e;
        const int left = i == 0 ? 0 : state[i - 1];
        const int right = e == N ? 0 : state[e];
        if (left == right) std::fill(state.begin() + i, state.begin() + e, left);
        i = e;
    }

    auto firstFrom = [&](int from, auto pred) { while (from < N && !pred(from)) ++from; return from; };
    auto lastBefore = [&](int from, int lo, auto pred) { while (from > lo && !pred(from - 1)) --from; return from; };
    auto above = [&](int i) { return smooth[i] >= kUnknownBelow; };
    auto below = [&](int i) { return smooth[i] <= -kUnknownBelow; };

    std::string out;
    char line[96];
    for (int cur = 0; cur != N;) {
        const int onAt = firstFrom(cur, [&](int i) { return state[i] == 2; });
        const int offAt = firstFrom(onAt, [&](int i) { return state[i] == 0; });
        int sEnd = onAt, eEnd = offAt;
        // (`< T` and `> -T` as the reference writes them, not `!above` / `!below`: they differ on a NaN median)
        if (sEnd != N) sEnd = above(sEnd) ? lastBefore(sEnd, 0, [&](int i) { return smooth[i] < kUnknownBelow; }) : firstFrom(sEnd, above);
        if (eEnd != N) eEnd = below(eEnd) ? lastBefore(eEnd, sEnd, [&](int i) { return smooth[i] > -kUnknownBelow; }) : firstFrom(eEnd, below);
        const int sStart = lastBefore(sEnd, cur, below);
        const int eStart = lastBefore(eEnd, sEnd, above);
        int sBest = sStart;
        while (sBest < sEnd && !(smooth[sBest] > 0)) ++sBest;
        const int eBest = lastBefore(eEnd, eStart, [&](int i) { return smooth[i] > 0; });
        if (sEnd != eEnd) {
            std::snprintf(line, sizeof line, "%6d S 0 ALL %6d %6d\n", sBest, sStart, sEnd);
            out += line;
            std::snprintf(line, sizeof line, "%6d E 0 ALL %6d %6d\n", eBest - 1, eStart - 1, eEnd - 1);
            out += line;
        }
        cur = offAt;
    }
    return out;
}

std::vector<int> parse_logoframe(const std::string& text, int numFrames)
{
    struct Mark { bool start; int from, to; };
    std::vector<Mark> marks;
    auto isws = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; };
    size_t p = 0;
    while (p < text.size()) {
        size_t eol = text.find('\n', p);
        if (eol == std::string::npos) eol = text.size();
        // fields: <digits> <one char> <digits> <token> <digits> <digits...>
        std::string tok[6];
        size_t q = p;
        int nt = 0;
        while (nt < 6) {
            while (q < eol && isws(text[q])) ++q;
            if (q >= eol) break;
            size_t b = q;
            while (q < eol && !isws(text[q])) ++q;
            tok[nt++] = text.substr(b, q - b);
        }
        p = eol + 1;
        if (nt < 6) continue;
        auto digits = [](const std::string& s) { return !s.empty() && std::all_of(s.begin(), s.end(), [](char c) { return c >= '0' && c <= '9'; }); };
        size_t lead = 0;
        while (lead < tok[5].size() && tok[5][lead] >= '0' && tok[5][lead] <= '9') ++lead;
        if (!digits(tok[0]) || tok[1].size() != 1 || !digits(tok[2]) || !digits(tok[4]) || lead == 0) continue;
        marks.push_back(Mark{(tok[1][0] | 0x20) == 's', std::stoi(tok[4]), std::stoi(tok[5].substr(0, lead))});
    }
    std::vector<int> st(numFrames, 0);
    auto paint = [&](int a, int b, int v) {
        a = std::min(numFrames, a);
        b = std::min(numFrames, b);
        for (int i = a; i < b; ++i) st[i] = v;
    };
    if (marks.size() % 2) throw std::runtime_error("Invalid logoframe file. Start and End must be cyclic.");
    for (size_t i = 0; i < marks.size(); i += 2) {
        const Mark &s = marks[i], &e = marks[i + 1];
        if (!s.start || e.start) throw std::runtime_error("Invalid logoframe file. Start and End must be cyclic.");
        paint(s.from, s.to + 1, 1);        // fade-in span
        paint(s.to, e.from + 1, 2);        // logo on
        paint(e.from + 1, e.to + 1, 1);    // fade-out span
    }
    return st;
}

namespace {
inline int argmin11(const float* v)
{
    int best = 0;
    for (int i = 1; i < 11; ++i) if (v[i] < v[best]) best = i;
    return best;
}
} // namespace

FadePair fade_from_analysis(const float* analysis, int numFrames, int n)
{
    // nine samples around n.  The reference indexes clamp(n+i)+i through the analyze clip (frame k>>3,
    // slot k&7); the clip's frame number is clamped by the AviSynth cache, slot s of analyze frame q
    // describes source frame clamp(8q+s).  Net effect away from the clip ends: frames n-8,n-6,..,n+8.
    const int nAnalyze = (numFrames + 7) / 8;
    int best[9];
    const float* centre = nullptr;
    for (int i = -4; i <= 4; ++i) {
        const int k = std::max(0, std::min(numFrames - 1, n + i)) + i;
        const int q = std::max(0, std::min(nAnalyze - 1, k >> 3));
        const int src = std::max(0, std::min(numFrames - 1, q * 8 + (k & 7)));
        const float* rec = analysis + (size_t)src * 33;
        best[i + 4] = argmin11(rec);
        if (i == 0) centre = rec;
    }
    float before = 0, after = 0;
    for (int i = 1; i <= 4; ++i) { before += best[4 - i]; after += best[4 + i]; }
    before /= 40;
    after /= 40;
    const bool abrupt = (before < 0.3 && after > 0.7) || (before > 0.7 && after < 0.3);
    if (abrupt) return FadePair{argmin11(centre + 11) / 10.0f, argmin11(centre + 22) / 10.0f};
    const float f = best[4] / 10.0f;
    return FadePair{f, f};
}

FadePair fade_for_frame(const std::vector<int>& frameState, int maxFadeLength, const float* analysis, int numFrames, int n)
{
    if (frameState.empty()) return fade_from_analysis(analysis, numFrames, n);
    const int half = maxFadeLength >> 1;
    const int first = frameState[std::max(0, std::min(numFrames - 1, n - half))];
    bool uniform = true;
    for (int i = -half + 1; i <= half && uniform; ++i)
        uniform = frameState[std::max(0, std::min(numFrames - 1, n + i))] == first;
    if (uniform) {
        const float f = frameState[std::max(0, std::min(numFrames - 1, n))] == 2 ? 1.0f : 0.0f;
        return FadePair{f, f};
    }
    return fade_from_analysis(analysis, numFrames, n);
}

} // namespace amt
