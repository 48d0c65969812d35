// stats_decisions.hpp -- host decisions on the whole-frame metrics (self-specified):
// scene changes (what the reference obtains from chapter_exe's "SCPos:" lines, CMAnalyze.hpp:411-439) and
// the per-frame cadence class that the KFM analysis passes would leave in *.duration.txt
// (FilteredSource.hpp:265-269,637-676).  Integer arithmetic only.
#pragma once
#include <cstdint>
#include <vector>

namespace amt {

enum Cadence : uint8_t { kCadence60i = 0, kCadence24p = 1, kCadence30p = 2 };

std::vector<int> scene_changes(const uint64_t* metrics, int nframes, int width, int height);
void classify_cadence(const uint64_t* metrics, int nframes, int width, int height, uint8_t* cadence, uint8_t* phase);
// durations in 60p ticks of the clip AMTDecimate wraps: 60i frame -> 1,1; 30p -> 2; a full 3:2 cycle -> 2,3,2,3
std::vector<int> cadence_durations(const uint8_t* cadence, const uint8_t* phase, int nframes);

} // namespace amt
