// build_knobs.h -- every compile-time knob of the library, fenced.
//
// The kernels carry ablations (AMT_*_NO_*: parts of a kernel switched off to time the rest -- WRONG RESULTS BY DESIGN), phase timers
// (AMT_*_TIMING), shape parameters of the sweeps quoted in profiles/ (AMT_LIN_G, AMT_TILE_WAVES ...) and host-side tracing
// (AMT_TRACE_CALLS, AMT_EXPERIMENT).  The release library is built with NONE of them (amatsukaze_amd/build.py FLAGS / EXTRA_FLAGS,
// tests/test_abi_and_host.py); a stray -DAMT_LIN_NO_FIXUP in a packager's CXXFLAGS must not produce a library that is silently wrong
// or silently different.  Any of them without AMT_INSTRUMENTED_BUILD -- which only build.py's build_variant() sets, and which
// names the product libamt_gpu_<variant>.so -- is a compile error.  Included first by every source of the library.
#pragma once

#if !defined(AMT_INSTRUMENTED_BUILD)
#if defined(AMT_LIN_NO_EVAL) || defined(AMT_LIN_NO_FIXUP) || defined(AMT_LIN_NO_FLUSH) || defined(AMT_LIN_NO_CONVERT) || defined(AMT_LIN_RAW_SAMEFRAME) || \
    defined(AMT_PAIR_NO_SUM) || defined(AMT_PAIR_NO_EVAL) || defined(AMT_PAIR_NO_CONVERT) || defined(AMT_PAIR_NO_FLUSH) || defined(AMT_PAIR_NO_GATHER) || \
    defined(AMT_PAIR_NO_RAW) || defined(AMT_PAIR_RAW_SAMEFRAME)
#error "an ablation macro (AMT_*_NO_* / *_RAW_SAMEFRAME) is defined: these builds compute wrong results by design and exist only as instrumented variants (amatsukaze_amd/build.py build_variant)"
#endif
#if defined(AMT_LIN_TIMING) || defined(AMT_PAIR_TIMING) || defined(AMT_FUSED_TIMING) || defined(AMT_TRACE_CALLS) || defined(AMT_EXPERIMENT) || \
    defined(AMT_TRACE_NO_INGEST) || defined(AMT_TRACE_NO_EVENTS) || defined(AMT_TRACE_NO_ANALYSIS_KERNEL) || defined(AMT_SAME_STREAM)
#error "an instrumentation macro (AMT_*_TIMING / AMT_TRACE_* / AMT_EXPERIMENT / AMT_SAME_STREAM) is defined outside an instrumented build (amatsukaze_amd/build.py build_variant)"
#endif
#if defined(AMT_LIN_G) || defined(AMT_LIN_G16) || defined(AMT_LIN_OCC) || defined(AMT_LIN_OCC16) || defined(AMT_LIN_WAVES) || defined(AMT_LIN_WGS_MIN16) || \
    defined(AMT_TILE_WAVES) || defined(AMT_TILE_G) || defined(AMT_PAIR_OCC) || defined(AMT_FUSED_OCC) || defined(AMT_FUSED_BG_LDS) || defined(AMT_LISTED_FADE_CHUNK) || \
    defined(AMT_STATS_ROWS) || defined(AMT_STATS_ROWS8) || defined(AMT_STATS_RUN) || defined(AMT_STATS_COLB) || defined(AMT_STATS_LEAN) || defined(AMT_STATS_PINGPONG) || \
    defined(AMT_STATS_DEAL) || defined(AMT_STATS_NT) || defined(AMT_STATS_OCC) || defined(AMT_STATS_WAVES) || defined(AMT_STATS_LDS_BYTES) || \
    defined(AMT_DELOGO_ROWS) || defined(AMT_DELOGO_FRAMES) || defined(AMT_SCAN_ACC_FIXED32) || defined(AMT_WG_PLAIN_MAP)
#error "a shape / tuning macro of the kernels is defined on the command line: the release library is built with the defaults in the sources (instrumented variants: amatsukaze_amd/build.py build_variant)"
#endif
#endif
