// idle_tick.cpp -- when does work submitted to an idle queue start late on this box?
// One "block" = what a frame-by-frame host does per GetFrame block (include/amt_filters.hpp): a host-to-device copy of a few MB on a
// side stream, an event, the compute stream waits for it and runs a short kernel, a small device-to-host copy, host waits.  The loop
// idles the host for `gap` ms between blocks and reports the latency distribution of a block for several ways of doing the same work:
//   sdma2d    copy = hipMemcpy2DAsync (rows of 256 B, pitched destination) on the side stream          [what round 3 did]
//   sdma1d    copy = one linear hipMemcpyAsync on the side stream
//   same1d    the linear copy on the COMPUTE stream (one queue)
//   zerocopy  no copy engine: the kernel reads the pinned buffer over PCIe itself
//   + "hb": a helper thread launches an empty kernel on a third stream every 1 ms (amtgpu_context_set_keepalive)
// hipcc -O2 -o idle_tick idle_tick.cpp -pthread ; one JSON line
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void work_kernel(const unsigned* __restrict__ src, unsigned* __restrict__ out, int n, int iters)
{
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += src[i] * (it + 1);
    if (acc == 0x12345678u) out[0] = acc;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[1] = src[0];
}
__global__ void noop_kernel() {}

int main()
{
    const size_t bytes = 2u << 20;                 // one upload group of the filter layer: 64 frames x 128 rows x 256 B
    const int rows = 8192, roww = 256, dpitch = 1472;
    void *pin = nullptr, *dev = nullptr, *dev2d = nullptr, *land = nullptr;
    unsigned* dout = nullptr;
    CK(hipHostMalloc(&pin, bytes, hipHostMallocDefault));
    CK(hipHostMalloc(&land, 4096, hipHostMallocDefault));
    CK(hipMalloc(&dev, bytes));
    CK(hipMalloc(&dev2d, (size_t)rows * dpitch));
    CK(hipMalloc((void**)&dout, 4096));
    for (size_t i = 0; i < bytes / 4; ++i) ((unsigned*)pin)[i] = (unsigned)i;
    void* pin_dev = nullptr;
    CK(hipHostGetDevicePointer(&pin_dev, pin, 0));
    hipStream_t sc, sk, shb;
    CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&shb, hipStreamNonBlocking));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    std::atomic<bool> hb_on{false}, hb_stop{false};
    std::thread hb([&] {
        CK(hipSetDevice(0));
        while (!hb_stop.load()) {
            if (hb_on.load() && hipStreamQuery(shb) == hipSuccess) hipLaunchKernelGGL(noop_kernel, dim3(1), dim3(64), 0, shb);
            std::this_thread::sleep_for(std::chrono::microseconds(1000));
        }
    });
    auto block = [&](int mode) {
        const double t0 = now_ms();
        if (mode == 0) {
            CK(hipMemcpy2DAsync(dev2d, dpitch, pin, roww, roww, rows, hipMemcpyHostToDevice, sc));
            CK(hipEventRecord(ev, sc)); CK(hipStreamWaitEvent(sk, ev, 0));
            hipLaunchKernelGGL(work_kernel, dim3(512), dim3(256), 0, sk, (const unsigned*)dev2d, dout, (int)(bytes / 4), 4);
        } else if (mode == 1) {
            CK(hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, sc));
            CK(hipEventRecord(ev, sc)); CK(hipStreamWaitEvent(sk, ev, 0));
            hipLaunchKernelGGL(work_kernel, dim3(512), dim3(256), 0, sk, (const unsigned*)dev, dout, (int)(bytes / 4), 4);
        } else if (mode == 2) {
            CK(hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, sk));
            hipLaunchKernelGGL(work_kernel, dim3(512), dim3(256), 0, sk, (const unsigned*)dev, dout, (int)(bytes / 4), 4);
        } else {
            hipLaunchKernelGGL(work_kernel, dim3(512), dim3(256), 0, sk, (const unsigned*)pin_dev, dout, (int)(bytes / 4), 4);
        }
        CK(hipMemcpyAsync(land, dout, 4096, hipMemcpyDeviceToHost, sk));
        CK(hipStreamSynchronize(sk));
        return now_ms() - t0;
    };
    const char* names[4] = {"sdma2d", "sdma1d", "same1d", "zerocopy"};
    std::printf("{");
    bool first = true;
    for (int hbm = 0; hbm < 2; ++hbm) {
        hb_on.store(hbm != 0);
        for (int mode = 0; mode < 4; ++mode) {
            for (int gap : {0, 2, 5, 12, 30}) {
                for (int w = 0; w < 3; ++w) block(mode);
                std::vector<double> t;
                for (int i = 0; i < 25; ++i) {
                    if (gap) std::this_thread::sleep_for(std::chrono::milliseconds(gap));
                    t.push_back(block(mode));
                }
                std::sort(t.begin(), t.end());
                std::printf("%s\"%s%s_gap%dms\": {\"min\": %.2f, \"median\": %.2f, \"p90\": %.2f, \"max\": %.2f}", first ? "" : ", ", names[mode], hbm ? "_hb" : "", gap,
                            t.front(), t[t.size() / 2], t[(t.size() * 9) / 10], t.back());
                first = false;
            }
        }
    }
    std::printf("}\n");
    hb_stop.store(true);
    hb.join();
    return 0;
}
