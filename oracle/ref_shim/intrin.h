// oracle/ref_shim/intrin.h -- TEST INFRASTRUCTURE ONLY: MSVC <intrin.h> surface used by
// Amatsukaze/ComputeKernel.cpp:10,23,33 (MSVC-signature __cpuid; _xgetbv comes from <immintrin.h> with -mxsave).
#pragma once
static inline void __cpuid(int info[4], int leaf)
{
    __asm__ __volatile__("cpuid" : "=a"(info[0]), "=b"(info[1]), "=c"(info[2]), "=d"(info[3]) : "a"(leaf), "c"(0));
}
