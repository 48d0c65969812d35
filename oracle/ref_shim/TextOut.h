// oracle/ref_shim/TextOut.h -- TEST INFRASTRUCTURE ONLY: debug text overlay (AMTEraseLogo mode!=0) is out of scope.
#pragma once
inline void DrawText(const PVideoFrame&, bool, int, int, const char*) { }
