// amt_gpu_upload.hip -- C ABI part 5: frames between host and HBM (the step AMTSource::GetFrame feeds, AMTSource.hpp:721-780,
// 428-442): device allocation, the pinned staging ring with its worker threads, registered (page-locked in place) host frames,
// downloads, stream markers.
//
// Ingest path.  A decoder's frames normally sit in pageable memory, which the DMA engines cannot read: they are staged through
// a ring of pinned slots on the way -- host memcpy into a slot, hipMemcpyAsync out of it on the side stream.  One core's memcpy
// (~10-25 GB/s) is below what PCIe Gen5 x16 carries (~55 GB/s), so the staging copy of a large upload is shared out over a few
// worker threads, and the ring has four slots so that the CPU fills slots k+1.. while the copy engine drains slot k.  A host that
// can keep its frame buffers in place (a decoder's frame pool) registers them once (amtgpu_frames_register = hipHostRegister):
// uploads from inside a registered range skip the ring altogether.
#include "build_knobs.h"
#include "../../include/amt_gpu.h"

#if defined(__x86_64__)
#include <emmintrin.h>
#endif

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>

#include "api_common.hpp"

namespace amt {
hipError_t launch_ingest_rows(hipStream_t st, const void* src_host_mapped, long long src_stride, void* dst, long long dst_stride,
                              unsigned long long chunk, long long nchunks);
}
using namespace amt;

namespace {
inline void cpu_relax()
{
#if defined(__x86_64__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}
} // namespace

// ---------------------------------------------------------------------------------------------
// staging workers: a fixed set of threads that run slices of one job at a time (the caller takes a slice itself)
// ---------------------------------------------------------------------------------------------
struct AmtGpuContext::UploadPool {
    // Jobs arrive in bursts -- one per ring slot of a large upload, a fraction of a millisecond apart -- and last ~0.1 ms each, so a
    // worker that went to sleep on a condition variable after every job would spend the next one waking up (measured: the calling
    // thread then does most slices itself and the upload stays at one core's memcpy rate).  Workers therefore keep watching the job
    // counter for kSpinUs after a job before they block.
    static constexpr int kSpinUs = 1000;
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_work;
    std::atomic<uint64_t> generation{0};
    std::atomic<int> sleepers{0};
    const std::function<void(int)>* job = nullptr;      // job(slice); published by the release store of `claim`
    // [job number : 24][slices of the job : 20][next slice to hand out : 20] in ONE word, so that a claim is self-describing: a worker
    // that comes late cannot take a stale index for a slice of the job that has started meanwhile
    std::atomic<uint64_t> claim{0};
    std::atomic<int> pending{0};
    std::atomic<bool> stop{false};

    explicit UploadPool(int nworkers)
    {
        for (int i = 0; i < nworkers; ++i) workers.emplace_back([this] { loop(); });
    }
    ~UploadPool()
    {
        stop.store(true);
        { std::lock_guard<std::mutex> lk(m); }
        cv_work.notify_all();
        for (auto& t : workers) t.join();
    }
    void work()
    {
        for (;;) {
            const uint64_t v = claim.fetch_add(1, std::memory_order_acq_rel);
            const int s = (int)(v & 0xFFFFF), n = (int)((v >> 20) & 0xFFFFF);
            if (s >= n) return;
            (*job)(s);                       // a valid claim keeps its job alive: run() returns only when pending reaches 0
            pending.fetch_sub(1, std::memory_order_release);
        }
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            // wait for a job newer than `seen`: spin for a while, then block
            const auto t0 = std::chrono::steady_clock::now();
            uint64_t g;
            int polls = 0;
            while ((g = generation.load(std::memory_order_acquire)) == seen && !stop.load(std::memory_order_relaxed)) {
                if ((++polls & 63) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(kSpinUs)) {
                    std::unique_lock<std::mutex> lk(m);
                    sleepers.fetch_add(1, std::memory_order_seq_cst);
                    cv_work.wait(lk, [&] { return generation.load(std::memory_order_seq_cst) != seen || stop.load(); });
                    sleepers.fetch_sub(1);
                } else {
                    cpu_relax();
                }
            }
            if (stop.load(std::memory_order_relaxed)) return;
            seen = g;
            work();
        }
    }
    // runs fn(0..n-1), the calling thread included; returns when all slices are done.  One job at a time (callers hold the
    // context's lock).
    void run(int n, const std::function<void(int)>& fn)
    {
        if (n <= 0) return;
        if (n >= (1 << 20) - 4096) throw std::runtime_error("too many slices for one staging job");
        job = &fn;
        pending.store(n, std::memory_order_relaxed);
        const uint64_t g = generation.load(std::memory_order_relaxed) + 1;
        claim.store(((g & 0xFFFFFF) << 40) | ((uint64_t)n << 20), std::memory_order_release);
        // Dekker-style handshake with loop(): store generation, THEN read sleepers -- against a worker's increment sleepers, THEN read
        // generation.  Both sides must be sequentially consistent (a release store may pass the later load on x86: a missed wake-up
        // would leave the worker asleep through the job).
        generation.store(g, std::memory_order_seq_cst);
        if (sleepers.load(std::memory_order_seq_cst) > 0) { { std::lock_guard<std::mutex> lk(m); } cv_work.notify_all(); }
        work();
        while (pending.load(std::memory_order_acquire) > 0) cpu_relax();
        // (a worker that wakes late finds no slice left in `claim` and leaves `job` alone: it is only dereferenced for a claimed slice)
    }
};

namespace amt {

void upload_pool_default(AmtGpuContext* c)
{
    // staging threads: enough to pass PCIe Gen5 x16 with headroom, never more than a quarter of the host's cores.  The pool itself
    // is created on the first large upload.
    const unsigned hc = std::max(1u, std::thread::hardware_concurrency());
    c->upload_threads = (int)std::max(1u, std::min(4u, hc / 4));      // measured (profiles/r04_notes.md): 4 threads carry the link, more only contend
}

void context_stop_threads(AmtGpuContext* c)
{
    delete c->pool;
    c->pool = nullptr;
}

} // namespace amt

#ifdef AMT_TRACE_CALLS
__global__ void trace_stamp_kernel(unsigned long long* slot) { *slot = wall_clock64(); }
namespace amt {
void trace_stamp(AmtGpuContext* c, hipStream_t st, int slot)
{
    if (!c->tr_stamps) {
        AMT_HIP(hipHostMalloc((void**)&c->tr_stamps, 8 * sizeof(unsigned long long), hipHostMallocDefault));
        for (int i = 0; i < 8; ++i) c->tr_stamps[i] = 0;
        // calibration: the stamp of a kernel that is known to run between two host clock readings (best of 20)
        double best = 1e30;
        for (int i = 0; i < 20; ++i) {
            AMT_HIP(hipStreamSynchronize(c->stream));
            const double h0 = AmtTrace::now();
            hipLaunchKernelGGL(trace_stamp_kernel, dim3(1), dim3(1), 0, c->stream, c->tr_stamps + 7);
            AMT_HIP(hipStreamSynchronize(c->stream));
            const double h1 = AmtTrace::now();
            if (h1 - h0 < best) { best = h1 - h0; c->tr_offset_us = 0.5 * (h0 + h1) - (double)c->tr_stamps[7] / 100.0; }
        }
    }
    hipLaunchKernelGGL(trace_stamp_kernel, dim3(1), dim3(1), 0, st, c->tr_stamps + slot);
}
}
#endif

namespace {
constexpr size_t kSlotBytes = 32u << 20;
constexpr size_t kParallelMin = 1u << 20;          // staging copies below this stay on the calling thread
constexpr size_t kSliceBytes = 512u << 10;

// `n` bytes (<= kSlotBytes) of the slot being filled.  A slot that cannot take them is closed -- an event behind its last copy -- and
// the next one in the ring, once ITS copies (issued three slots ago) have drained, becomes the slot being filled.  Small uploads
// (one frame's logo rows) thus share a slot and cost a memcpy and a copy launch each, no event wait.
uint8_t* stage_acquire(AmtGpuContext* c, size_t n)
{
    if (!c->pinned) {
        AMT_HIP(hipHostMalloc(&c->pinned, kSlotBytes * AmtGpuContext::kRingSlots, hipHostMallocDefault));
        c->pinned_bytes = kSlotBytes;
    }
    if (c->slot_fill + n > kSlotBytes) {
        AMT_HIP(hipEventRecord(c->slot_free[c->next_slot], c->copy_stream));
        c->next_slot = (c->next_slot + 1) % AmtGpuContext::kRingSlots;
        { AMT_TRACE_SCOPE("stage_acquire.event_wait"); AMT_HIP(hipEventSynchronize(c->slot_free[c->next_slot])); }
        c->slot_fill = 0;
    }
    uint8_t* p = (uint8_t*)c->pinned + (size_t)c->next_slot * kSlotBytes + c->slot_fill;
    c->slot_fill += (n + 255) & ~(size_t)255;
    return p;
}

AmtGpuContext::UploadPool* pool_of(AmtGpuContext* c)
{
    if (c->upload_threads <= 1) return nullptr;
    if (!c->pool || (int)c->pool->workers.size() != c->upload_threads - 1) {
        delete c->pool;
        c->pool = nullptr;
        c->pool = new AmtGpuContext::UploadPool(c->upload_threads - 1);
    }
    return c->pool;
}

// The staging copy proper.  The destination is a pinned slot that the CPU never reads again (the DMA engine does): non-temporal
// stores skip the read-for-ownership of every destination line -- a third of the copy's memory traffic -- and keep the slot out of
// the caches.  dst is 16-byte aligned for every piece this file hands out whose size is a multiple of 16; anything else (ragged
// heads and tails, small pieces) goes through memcpy.
inline void stream_copy(uint8_t* dst, const uint8_t* src, size_t n)
{
#if !defined(__x86_64__)
    std::memcpy(dst, src, n);
#else
    if (n < 4096 || ((uintptr_t)dst & 15)) { std::memcpy(dst, src, n); return; }
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
        const __m128i a = _mm_loadu_si128((const __m128i*)(src + i)), b = _mm_loadu_si128((const __m128i*)(src + i + 16));
        const __m128i c = _mm_loadu_si128((const __m128i*)(src + i + 32)), d = _mm_loadu_si128((const __m128i*)(src + i + 48));
        _mm_stream_si128((__m128i*)(dst + i), a);
        _mm_stream_si128((__m128i*)(dst + i + 16), b);
        _mm_stream_si128((__m128i*)(dst + i + 32), c);
        _mm_stream_si128((__m128i*)(dst + i + 48), d);
    }
    _mm_sfence();
    if (i < n) std::memcpy(dst + i, src + i, n - i);
#endif
}

// dst <- src, n bytes, shared out over the staging threads in kSliceBytes pieces
void staged_copy(AmtGpuContext* c, uint8_t* dst, const uint8_t* src, size_t n)
{
    AmtGpuContext::UploadPool* P = n >= kParallelMin ? pool_of(c) : nullptr;
    if (!P) { stream_copy(dst, src, n); return; }
    const int slices = (int)((n + kSliceBytes - 1) / kSliceBytes);
    const std::function<void(int)> fn = [&](int s) {
        const size_t o = (size_t)s * kSliceBytes;
        stream_copy(dst + o, src + o, std::min(kSliceBytes, n - o));
    };
    P->run(slices, fn);
}

// `count` pieces of chunk bytes, piece i read from src_of(i), packed back to back at dst
template <typename SrcOf> void staged_gather(AmtGpuContext* c, uint8_t* dst, size_t chunk, int64_t count, SrcOf src_of)
{
    AmtGpuContext::UploadPool* P = (size_t)count * chunk >= kParallelMin ? pool_of(c) : nullptr;
    if (!P) {
        for (int64_t i = 0; i < count; ++i) std::memcpy(dst + (size_t)i * chunk, src_of(i), chunk);
        return;
    }
    const int64_t per = std::max<int64_t>(1, (int64_t)(kSliceBytes / chunk));
    const int slices = (int)((count + per - 1) / per);
    const std::function<void(int)> fn = [&](int s) {
        const int64_t i1 = std::min(count, (int64_t)(s + 1) * per);
        for (int64_t i = (int64_t)s * per; i < i1; ++i) std::memcpy(dst + (size_t)i * chunk, src_of(i), chunk);
    };
    P->run(slices, fn);
}

// [p, p + n) lies inside a range registered with amtgpu_frames_register
bool is_registered(const AmtGpuContext* c, const void* p, size_t n)
{
    const uintptr_t a = (uintptr_t)p;
    for (const auto& r : c->registered)
        if (a >= r.first && a + n <= r.first + r.second) return true;
    return false;
}

// the address a kernel reads page-locked host memory at (the ring, or a registered range)
const void* device_view(const void* host_pinned)
{
    void* d = nullptr;
    AMT_HIP(hipHostGetDevicePointer(&d, const_cast<void*>(host_pinned), 0));
    return d;
}

void copies_issued(AmtGpuContext* c)
{
#ifdef AMT_TRACE_CALLS
    static const bool no_events = std::getenv("AMT_TRACE_NO_EVENTS") != nullptr;     // (with AMT_SAME_STREAM: no event is needed)
    if (no_events) return;
#endif
    AMT_HIP(hipEventRecord(c->copy_done, c->copy_stream));
    c->copies_pending = true;
#ifdef AMT_TRACE_CALLS
    if (!c->tr_copies_done) AMT_HIP(hipEventCreate(&c->tr_copies_done));
    AMT_HIP(hipEventRecord(c->tr_copies_done, c->copy_stream));
    if (c->tr_block_open) amt::trace_stamp(c, c->copy_stream, 1);
#endif
}
#ifdef AMT_TRACE_CALLS
void trace_block_begin(AmtGpuContext* c)
{
    if (c->tr_block_open) return;
    c->tr_host_block_begin = AmtTrace::now();
    amt::trace_stamp(c, c->copy_stream, 0);
    if (!c->tr_first_copy) AMT_HIP(hipEventCreate(&c->tr_first_copy));
    AMT_HIP(hipEventRecord(c->tr_first_copy, c->copy_stream));
    c->tr_block_open = true;
}
#define AMT_TRACE_BLOCK_BEGIN(c) trace_block_begin(c)
#else
#define AMT_TRACE_BLOCK_BEGIN(c) do { } while (0)
#endif
} // namespace

namespace {
void land_pinned(AmtGpuContext* c, const void* dsrc, uint64_t bytes);
}
namespace amt {
void download_via_pinned(AmtGpuContext* c, void* hdst, const void* dsrc, size_t bytes)
{
    if (!bytes) { AMT_HIP(hipStreamSynchronize(c->stream)); return; }
    land_pinned(c, dsrc, bytes);
    std::memcpy(hdst, c->pinned_down, bytes);
}
} // namespace amt
namespace {
void land_pinned(AmtGpuContext* c, const void* dsrc, uint64_t bytes)
{
    if (bytes > c->pinned_down_bytes) {
        if (c->pinned_down) { (void)hipHostFree(c->pinned_down); c->pinned_down = nullptr; c->pinned_down_bytes = 0; }
        AMT_HIP(hipHostMalloc(&c->pinned_down, (size_t)bytes, hipHostMallocDefault));
        c->pinned_down_bytes = (size_t)bytes;
    }
    if (bytes) {
        AMT_HIP(hipMemcpyAsync(c->pinned_down, dsrc, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
        AMT_HIP(hipStreamSynchronize(c->stream));
    }
}
} // namespace


extern "C" {

void* amtgpu_device_alloc(AmtGpuContext* c, uint64_t bytes)
{
    void* p = nullptr;
    if (!guard(c, [&] { c->bind(); AMT_HIP(hipMalloc(&p, bytes)); })) return nullptr;
    return p;
}
void amtgpu_device_free(AmtGpuContext* c, void* p)
{
    if (c && p) { (void)hipSetDevice(c->device); (void)hipFree(p); }
}

int amtgpu_context_set_upload_threads(AmtGpuContext* c, int nthreads)
{
    return guard(c, [&] {
        if (!c) throw std::runtime_error("no context");
        if (nthreads < 1 || nthreads > 64) throw std::runtime_error("upload threads must be 1..64");
        c->upload_threads = nthreads;
    });
}

int amtgpu_frames_register(AmtGpuContext* c, void* hptr, uint64_t bytes)
{
    return guard(c, [&] {
        if (!c) throw std::runtime_error("no context");
        if (!hptr || !bytes) throw std::runtime_error("empty host range");
        c->bind();
        AMT_HIP(hipHostRegister(hptr, (size_t)bytes, hipHostRegisterDefault));
        c->registered.emplace_back((uintptr_t)hptr, (size_t)bytes);
    });
}

int amtgpu_frames_unregister(AmtGpuContext* c, void* hptr)
{
    return guard(c, [&] {
        if (!c) throw std::runtime_error("no context");
        for (size_t i = 0; i < c->registered.size(); ++i)
            if (c->registered[i].first == (uintptr_t)hptr) {
                c->bind();
                // copies out of the range may still be in flight on the side stream
                AMT_HIP(hipStreamSynchronize(c->copy_stream));
                AMT_HIP(hipHostUnregister(hptr));
                c->registered.erase(c->registered.begin() + (long)i);
                return;
            }
        throw std::runtime_error("host range was not registered");
    });
}

int amtgpu_frames_upload(AmtGpuContext* c, void* ddst, const void* hsrc, uint64_t bytes)
{
    return guard(c, [&] {
        c->bind();
        if (bytes && is_registered(c, hsrc, (size_t)bytes)) {            // page-locked in place: the DMA engine reads it where it is
            AMT_HIP(hipMemcpyAsync(ddst, hsrc, (size_t)bytes, hipMemcpyHostToDevice, c->copy_stream));
            copies_issued(c);
            return;
        }
        uint64_t done = 0;
        while (done < bytes) {
            const size_t n = (size_t)std::min<uint64_t>(kSlotBytes, bytes - done);
            uint8_t* stage = stage_acquire(c, n);
            staged_copy(c, stage, (const uint8_t*)hsrc + done, n);
            AMT_HIP(hipMemcpyAsync((uint8_t*)ddst + done, stage, n, hipMemcpyHostToDevice, c->copy_stream));
            done += n;
        }
        copies_issued(c);
    });
}

// the same ring for `nchunks` equally sized pieces that sit `dst_stride` apart on the device (e.g. the logo rectangle's rows of
// every frame of a batch: the logo passes read nothing else of a frame): pieces are packed into the pinned slot and leave
// as ONE 2-D copy per slot
int amtgpu_frames_upload_strided(AmtGpuContext* c, void* ddst, int64_t dst_stride, const void* hsrc, int64_t src_stride,
                                 uint64_t chunk_bytes, int nchunks)
{
    return guard(c, [&] {
        c->bind();
        if (chunk_bytes == 0 || nchunks <= 0) return;
        if (chunk_bytes > kSlotBytes) throw std::runtime_error("chunk larger than a staging slot");
        if (dst_stride < (int64_t)chunk_bytes || src_stride < (int64_t)chunk_bytes) throw std::runtime_error("stride smaller than the chunk");
        if (is_registered(c, hsrc, (size_t)(nchunks - 1) * (size_t)src_stride + (size_t)chunk_bytes)) {
            AMT_HIP(launch_ingest_rows(c->copy_stream, device_view(hsrc), src_stride, ddst, dst_stride, chunk_bytes, nchunks));
            copies_issued(c);
            return;
        }
        const int per_slot = (int)(kSlotBytes / chunk_bytes);
        for (int i0 = 0; i0 < nchunks; i0 += per_slot) {
            const int n = std::min(per_slot, nchunks - i0);
            uint8_t* stage = stage_acquire(c, (size_t)n * chunk_bytes);
            staged_gather(c, stage, (size_t)chunk_bytes, n, [&](int64_t i) { return (const uint8_t*)hsrc + (size_t)(i0 + i) * src_stride; });
            // (rows as wide as their destination pitch are one contiguous run: the copy engine at full rate; narrow rows go by kernel)
            if (dst_stride == (int64_t)chunk_bytes)
                AMT_HIP(hipMemcpyAsync((uint8_t*)ddst + (size_t)i0 * dst_stride, stage, (size_t)n * chunk_bytes, hipMemcpyHostToDevice, c->copy_stream));
            else
                AMT_HIP(launch_ingest_rows(c->copy_stream, device_view(stage), (long long)chunk_bytes, (uint8_t*)ddst + (size_t)i0 * dst_stride, dst_stride,
                                           chunk_bytes, n));
        }
        copies_issued(c);
    });
}

// `nsrc` sources of `chunks_per_src` pieces each (e.g. the logo rectangle's rows of nsrc separately allocated host frames) to
// destinations that continue one another: piece j of source i lands at ddst + (i * chunks_per_src + j) * dst_stride.  Packed into
// the pinned ring and sent as one 2-D copy per slot -- one call and one copy launch for a whole group of frames
int amtgpu_frames_upload_gather(AmtGpuContext* c, void* ddst, int64_t dst_stride, const void* const* hsrc, int64_t src_stride,
                                uint64_t chunk_bytes, int chunks_per_src, int nsrc)
{
    return guard(c, [&] {
        c->bind();
        if (chunk_bytes == 0 || chunks_per_src <= 0 || nsrc <= 0) return;
        if (!hsrc) throw std::runtime_error("null source list");
        if (chunk_bytes > kSlotBytes) throw std::runtime_error("chunk larger than a staging slot");
        if (dst_stride < (int64_t)chunk_bytes || src_stride < (int64_t)chunk_bytes) throw std::runtime_error("stride smaller than the chunk");
        const int64_t total = (int64_t)chunks_per_src * nsrc;
        const int64_t per_slot = (int64_t)(kSlotBytes / chunk_bytes);
        AMT_TRACE_BLOCK_BEGIN(c);
        for (int64_t i0 = 0; i0 < total; i0 += per_slot) {
            const int64_t n = std::min(per_slot, total - i0);
            uint8_t* stage = stage_acquire(c, (size_t)n * chunk_bytes);
            {
                AMT_TRACE_SCOPE("upload_gather.staging_copy");
                staged_gather(c, stage, (size_t)chunk_bytes, n, [&](int64_t i) {
                    const int64_t q = i0 + i;
                    return (const uint8_t*)hsrc[q / chunks_per_src] + (size_t)(q % chunks_per_src) * src_stride;
                });
            }
            AMT_TRACE_SCOPE("upload_gather.rows_to_device");
#ifdef AMT_TRACE_CALLS
            static const bool no_ingest = std::getenv("AMT_TRACE_NO_INGEST") != nullptr;
            if (no_ingest) continue;
#endif
            if (dst_stride == (int64_t)chunk_bytes)
                AMT_HIP(hipMemcpyAsync((uint8_t*)ddst + (size_t)i0 * dst_stride, stage, (size_t)n * chunk_bytes, hipMemcpyHostToDevice, c->copy_stream));
            else
                AMT_HIP(launch_ingest_rows(c->copy_stream, device_view(stage), (long long)chunk_bytes, (uint8_t*)ddst + (size_t)i0 * dst_stride, dst_stride,
                                           chunk_bytes, n));
        }
        copies_issued(c);
    });
}

int amtgpu_frames_upload_wait(AmtGpuContext* c)
{
    return guard(c, [&] {
        c->bind();
        if (c->copies_pending) { AMT_HIP(hipStreamWaitEvent(c->stream, c->copy_done, 0)); c->copies_pending = false; }
#ifdef AMT_TRACE_CALLS
        if (!c->tr_released) AMT_HIP(hipEventCreate(&c->tr_released));
        AMT_HIP(hipEventRecord(c->tr_released, c->stream));
        if (c->tr_block_open) amt::trace_stamp(c, c->stream, 2);
#endif
    });
}

int amtgpu_download(AmtGpuContext* c, void* hdst, const void* dsrc, uint64_t bytes)
{
    return guard(c, [&] {
        c->bind();
        download_via_pinned(c, hdst, dsrc, (size_t)bytes);
    });
}

int amtgpu_download_strided(AmtGpuContext* c, void* hdst, int64_t dst_stride, const void* dsrc, int64_t src_stride, uint64_t chunk_bytes,
                            int nchunks)
{
    return guard(c, [&] {
        c->bind();
        if (chunk_bytes == 0 || nchunks <= 0) return;
        if (dst_stride < (int64_t)chunk_bytes || src_stride < (int64_t)chunk_bytes) throw std::runtime_error("stride smaller than the chunk");
        // packed into the pinned landing buffer by one 2-D copy, handed out to the (pageable) rows from there
        const size_t total = (size_t)chunk_bytes * (size_t)nchunks;
        if (total > c->pinned_down_bytes) {
            if (c->pinned_down) { (void)hipHostFree(c->pinned_down); c->pinned_down = nullptr; c->pinned_down_bytes = 0; }
            AMT_HIP(hipHostMalloc(&c->pinned_down, total, hipHostMallocDefault));
            c->pinned_down_bytes = total;
        }
        // packed on the device side by the library's row kernel writing the pinned buffer directly (no 2-D copy of the runtime)
        AMT_HIP(launch_ingest_rows(c->stream, dsrc, src_stride, const_cast<void*>(device_view(c->pinned_down)), (long long)chunk_bytes, chunk_bytes, nchunks));
        AMT_HIP(hipStreamSynchronize(c->stream));
        for (int i = 0; i < nchunks; ++i)
            std::memcpy((uint8_t*)hdst + (size_t)i * dst_stride, (const uint8_t*)c->pinned_down + (size_t)i * chunk_bytes, (size_t)chunk_bytes);
    });
}

// device -> a pinned landing buffer of the context in ONE asynchronous copy + one wait; *hptr stays valid until the next call ON THIS
// CONTEXT from any thread -- a context shared by several host threads must use amtgpu_download_scatter instead
int amtgpu_download_pinned(AmtGpuContext* c, const void* dsrc, uint64_t bytes, const void** hptr)
{
    return guard(c, [&] {
        c->bind();
        if (!hptr) throw std::runtime_error("null result pointer");
        land_pinned(c, dsrc, bytes);
        *hptr = c->pinned_down;
    });
}

// `bytes` from the device in one copy, then handed out to host memory piece by piece while the context is still locked: the landing
// buffer never leaves the library, so several filters on several threads may share one context
int amtgpu_download_scatter(AmtGpuContext* c, const void* dsrc, uint64_t bytes, const AmtGpuScatter* pieces, int npieces)
{
    return guard(c, [&] {
        c->bind();
        if (npieces < 0 || (npieces > 0 && !pieces)) throw std::runtime_error("null piece list");
        for (int i = 0; i < npieces; ++i) {
            const AmtGpuScatter& p = pieces[i];
            if (p.nchunks < 0 || (p.nchunks > 0 && (!p.hdst || p.dst_stride < (int64_t)p.chunk_bytes)))
                throw std::runtime_error("bad scatter piece");
            if (p.src_offset > bytes || (uint64_t)p.nchunks * p.chunk_bytes > bytes - p.src_offset) throw std::runtime_error("scatter piece outside the downloaded range");
        }
        land_pinned(c, dsrc, bytes);
        const uint8_t* back = (const uint8_t*)c->pinned_down;
        for (int i = 0; i < npieces; ++i) {
            const AmtGpuScatter& p = pieces[i];
            const uint8_t* s = back + p.src_offset;
            for (int k = 0; k < p.nchunks; ++k) std::memcpy((uint8_t*)p.hdst + (size_t)k * p.dst_stride, s + (size_t)k * p.chunk_bytes, (size_t)p.chunk_bytes);
        }
    });
}

// markers on the compute stream: record(id) after a batch's launches, wait(id) on the host before the batch's device buffer is
// written again -- what a double-buffered caller needs instead of amtgpu_context_synchronize (which also waits for the NEXT batch)
int amtgpu_marker_record(AmtGpuContext* c, int id)
{
    return guard(c, [&] {
        c->bind();
        if (id < 0 || id >= 16) throw std::runtime_error("marker id out of range (0..15)");
        if (!c->markers[id]) AMT_HIP(hipEventCreateWithFlags(&c->markers[id], hipEventDisableTiming));
        AMT_HIP(hipEventRecord(c->markers[id], c->stream));
    });
}
int amtgpu_marker_wait(AmtGpuContext* c, int id)
{
    return guard(c, [&] {
        c->bind();
        if (id < 0 || id >= 16) throw std::runtime_error("marker id out of range (0..15)");
        if (c->markers[id]) AMT_HIP(hipEventSynchronize(c->markers[id]));          // never recorded: nothing to wait for
    });
}

// marker OBJECTS: the same record / wait pair on an event the caller owns, so that two users of a shared context (two LogoFrame
// scans, a filter and user code) cannot re-record each other's markers; the event remembers the stream it was recorded on
AmtGpuMarker* amtgpu_marker_create(AmtGpuContext* c)
{
    AmtGpuMarker* m = nullptr;
    guard(c, [&] {
        if (!c) throw std::runtime_error("no context");
        c->bind();
        hipEvent_t e;
        AMT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        m = reinterpret_cast<AmtGpuMarker*>(e);
    });
    return m;
}
void amtgpu_marker_destroy(AmtGpuContext* c, AmtGpuMarker* m)
{
    if (c && m) { (void)hipSetDevice(c->device); (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(m)); }
}
int amtgpu_marker_record_on(AmtGpuContext* c, AmtGpuMarker* m)
{
    return guard(c, [&] {
        if (!m) throw std::runtime_error("null marker");
        c->bind();
        AMT_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(m), c->stream));
    });
}
int amtgpu_marker_wait_on(AmtGpuContext* c, AmtGpuMarker* m)
{
    return guard(c, [&] {
        if (!m) throw std::runtime_error("null marker");
        c->bind();
        AMT_HIP(hipEventSynchronize(reinterpret_cast<hipEvent_t>(m)));             // never recorded: returns at once
    });
}

} // extern "C"
