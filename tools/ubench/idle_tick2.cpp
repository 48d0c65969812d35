// idle_tick2.cpp -- does work enqueued on a stream that went idle WITHOUT the host having synchronised with it start late?
// A block = k1 (copy-like kernel), host pause of `gap` ms (a decoder producing the next frames), k2, short compute kernel, small
// device-to-host copy, host waits.  Reported: ms from the moment k2 was enqueued to the host seeing the block complete.
// Variants: plain; "query": hipStreamQuery on the stream right before enqueuing after the pause; "sync": hipStreamSynchronize before
// the pause (the queue idles in a synchronised state); "hb": an empty kernel on another stream every millisecond meanwhile;
// "hbsame": the same empty kernel on the SAME stream every millisecond (the stream never idles longer than that).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void busy_ms(double ms) { const double t0 = now_ms(); while (now_ms() - t0 < ms) { } }
// what a frame source does while the GPU waits: allocate a frame-sized buffer, touch it, free it (mmap / munmap for sizes above
// malloc's mmap threshold)
static void churn_ms(double ms, size_t bytes)
{
    const double t0 = now_ms();
    while (now_ms() - t0 < ms) {
        volatile char* p = (volatile char*)std::malloc(bytes);
        for (size_t i = 0; i < bytes; i += 4096) p[i] = 1;
        std::free((void*)p);
    }
}

__global__ void work_kernel(const unsigned* __restrict__ src, unsigned* __restrict__ out, int n)
{
    unsigned acc = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += src[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void noop_kernel() {}

int main()
{
    const size_t bytes = 2u << 20;
    void *pin = nullptr, *land = nullptr;
    unsigned* dout = nullptr;
    CK(hipHostMalloc(&pin, bytes, hipHostMallocDefault));
    CK(hipHostMalloc(&land, 4096, hipHostMallocDefault));
    CK(hipMalloc((void**)&dout, 4096));
    for (size_t i = 0; i < bytes / 4; ++i) ((unsigned*)pin)[i] = (unsigned)i;
    hipStream_t sk, shb;
    CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&shb, hipStreamNonBlocking));
    std::atomic<int> hb_mode{0};
    std::atomic<bool> hb_stop{false};
    std::mutex hm;
    std::thread hb([&] {
        CK(hipSetDevice(0));
        while (!hb_stop.load()) {
            const int m = hb_mode.load();
            if (m == 1 && hipStreamQuery(shb) == hipSuccess) hipLaunchKernelGGL(noop_kernel, dim3(1), dim3(64), 0, shb);
            if (m == 2) { std::lock_guard<std::mutex> lk(hm); hipLaunchKernelGGL(noop_kernel, dim3(1), dim3(64), 0, sk); }
            std::this_thread::sleep_for(std::chrono::microseconds(1000));
        }
    });
    size_t churn_bytes = 0;
    auto block = [&](int variant, int gap, bool sleep_gap) {
        { std::lock_guard<std::mutex> lk(hm); hipLaunchKernelGGL(work_kernel, dim3(256), dim3(256), 0, sk, (const unsigned*)pin, dout, (int)(bytes / 4)); }
        if (variant == 2) CK(hipStreamSynchronize(sk));
        if (gap) { if (churn_bytes) churn_ms(gap, churn_bytes); else if (sleep_gap) std::this_thread::sleep_for(std::chrono::milliseconds(gap)); else busy_ms(gap); }
        if (variant == 1) (void)hipStreamQuery(sk);
        const double t0 = now_ms();
        {
            std::lock_guard<std::mutex> lk(hm);
            hipLaunchKernelGGL(work_kernel, dim3(256), dim3(256), 0, sk, (const unsigned*)pin, dout, (int)(bytes / 4));
            hipLaunchKernelGGL(work_kernel, dim3(256), dim3(256), 0, sk, (const unsigned*)pin, dout, (int)(bytes / 4));
            CK(hipMemcpyAsync(land, dout, 4096, hipMemcpyDeviceToHost, sk));
        }
        CK(hipStreamSynchronize(sk));
        return now_ms() - t0;
    };
    // (1) memory churn in the pause
    std::printf("{");
    for (size_t cb : {(size_t)2400000, (size_t)64 << 20}) {
        churn_bytes = cb;
        for (int gap : {1, 3, 8}) {
            for (int w = 0; w < 3; ++w) block(0, 0, false);
            std::vector<double> t;
            for (int i = 0; i < 20; ++i) t.push_back(block(0, gap, false));
            std::sort(t.begin(), t.end());
            std::printf("\"churn%zuMB_gap%d\": [%.2f, %.2f, %.2f], ", cb >> 20, gap, t.front(), t[t.size() / 2], t.back());
        }
    }
    churn_bytes = 0;
    const char* vn[3] = {"plain", "query", "sync"};
    bool first = true;
    for (int hbm = 0; hbm < 1; ++hbm) {
        hb_mode.store(hbm);
        for (int variant = 0; variant < 3; ++variant) {
            for (int sg = 0; sg < 2; ++sg) {
                for (int gap : {0, 1, 3, 8, 15}) {
                    for (int w = 0; w < 3; ++w) block(variant, 0, false);
                    std::vector<double> t;
                    for (int i = 0; i < 20; ++i) t.push_back(block(variant, gap, sg != 0));
                    std::sort(t.begin(), t.end());
                    std::printf("%s\"%s%s_%s_gap%d\": [%.2f, %.2f, %.2f]", first ? "" : ", ", vn[variant], hbm == 1 ? "_hb" : hbm == 2 ? "_hbsame" : "", sg ? "sleep" : "busy", gap,
                                t.front(), t[t.size() / 2], t.back());
                    first = false;
                }
            }
        }
    }
    std::printf("}\n");
    hb_stop.store(true);
    hb.join();
    return 0;
}
