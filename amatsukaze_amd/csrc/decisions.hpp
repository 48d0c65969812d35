// decisions.hpp -- host-side decision logic that consumes the GPU-produced per-frame numbers.
// O(frames x small window), integer/threshold logic: stays on the host like SURVEY.md section 8 rows
// a10/a12 prescribe.  Replaces LogoFrame::selectLogo / writeResult (LogoScan.hpp:1647-1827) and
// AMTEraseLogo::ReadLogoFrameFile / CalcFade / CalcFade2 (:1263-1341, :1421-1461).
#pragma once

#include <string>
#include <vector>

namespace amt {

struct LogoSelection { int bestLogo = -1; float logoRatio = 0; };

// evals: numFrames*numLogos*{corr0,corr1}
LogoSelection select_logo(const float* evals, int numFrames, int numLogos, int numCandidates);

// logoframe text ("%6d S 0 ALL %6d %6d\n%6d E 0 ALL %6d %6d\n" per logo section)
std::string logoframe_text(const float* evals, int numFrames, int numLogos, int logoIndex, int fpsNum, int fpsDen);

// per-frame state 0 = off, 1 = transition/unknown, 2 = on; throws std::runtime_error on S/E misordering
std::vector<int> parse_logoframe(const std::string& text, int numFrames);

struct FadePair { float top, bottom; };
// analysis: numFrames*33 floats (p[11],t[11],b[11] per source frame)
FadePair fade_from_analysis(const float* analysis, int numFrames, int n);
FadePair fade_for_frame(const std::vector<int>& frameState, int maxFadeLength, const float* analysis, int numFrames, int n);

} // namespace amt
