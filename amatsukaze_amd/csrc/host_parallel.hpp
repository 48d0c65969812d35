// host_parallel.hpp -- frame ranges of the host-side decision passes dealt over a few threads.
//
// The decisions of the sharded passes run REPLICATED on every rank over the whole clip: at eight ranks they are the
// serial term of the end-to-end pass.  Their window filters are local (a frame looks at most 15 frames back and 15 ahead), so a frame
// range is independent of every other once its window is primed from the frames before it; only the two state machines (unknown-run
// fill + text of writeResult, "nothing moves: keep the last cadence") walk the clip in order, and they touch a byte or two per frame.
#pragma once

#include <algorithm>
#include <atomic>
#include <exception>
#include <thread>
#include <vector>

namespace amt {

// limits set by amtgpu_host_set_parallelism (0 = the defaults: half the host's cores, at most 32 threads; 32 768 frames per thread)
struct HostParallelism { static std::atomic<int>& max_threads() { static std::atomic<int> v{0}; return v; }
                         static std::atomic<int>& min_frames() { static std::atomic<int> v{0}; return v; } };

// number of contiguous ranges [0, n) is cut into.  The decisions never depend on it (tests sweep it).
inline int parallel_parts(int n)
{
    if (n <= 0) return 0;
    const unsigned hc = std::max(1u, std::thread::hardware_concurrency());
    const int mt = HostParallelism::max_threads().load(std::memory_order_relaxed), mf = HostParallelism::min_frames().load(std::memory_order_relaxed);
    const int maxThreads = mt > 0 ? mt : (int)std::min(32u, std::max(1u, hc / 2));
    const int grain = mf > 0 ? mf : 32768;
    return std::max(1, std::min(maxThreads, n / grain));
}

// fn(lo, hi, part) over [0, n) cut into `parts` contiguous ranges (parts from parallel_parts(n), passed in so that a caller can size its
// per-part outputs first)
template <class F> void parallel_ranges(int n, int parts, F&& fn)
{
    if (n <= 0 || parts <= 0) return;
    if (parts == 1) { fn(0, n, 0); return; }
    std::vector<std::thread> th;
    th.reserve(parts - 1);
    auto bound = [&](int p) { return (int)((long long)n * p / parts); };
    // an exception on a worker (an allocation, say) travels to the caller: the first one is rethrown once every thread has been joined
    std::vector<std::exception_ptr> err(parts);
    auto guarded = [&](int p) { try { fn(bound(p), bound(p + 1), p); } catch (...) { err[p] = std::current_exception(); } };
    int started = 1;
    try {
        for (int p = 1; p < parts; ++p, ++started) th.emplace_back(guarded, p);
    } catch (...) {                                   // (no more threads to be had: the caller takes the remaining ranges itself)
        for (int p = started; p < parts; ++p) guarded(p);
    }
    guarded(0);
    for (auto& t : th) t.join();
    for (auto& e : err) if (e) std::rethrow_exception(e);
}

} // namespace amt
