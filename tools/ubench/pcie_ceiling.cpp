// pcie_ceiling.cpp -- what the host-to-device link of THIS box carries, to put amtgpu_frames_upload's rate next to:
//   plain pinned hipMemcpyAsync (hipHostMalloc'ed source) at a frame's size, a ring slot's size and 256 MiB; the same from
//   memory page-locked in place (hipHostRegister); two streams at once; and one core's / T cores' memcpy into pinned memory
//   (the staging copy of the ring).  One JSON line.
//   hipcc -O2 -o pcie_ceiling pcie_ceiling.cpp -pthread
#include <hip/hip_runtime.h>

#include <emmintrin.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t total = 1ull << 30;                       // bytes moved per measurement
    void *dev = nullptr, *pin = nullptr;
    CK(hipMalloc(&dev, total));
    CK(hipHostMalloc(&pin, total, hipHostMallocDefault));
    std::memset(pin, 1, total);
    char* pageable = (char*)std::aligned_alloc(4096, total);
    std::memset(pageable, 2, total);
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    auto h2d = [&](const char* src, size_t chunk, int nstreams) {
        double best = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            const double t0 = now();
            size_t k = 0;
            for (size_t o = 0; o < total; o += chunk, ++k)
                CK(hipMemcpyAsync((char*)dev + o, src + o, std::min(chunk, total - o), hipMemcpyHostToDevice, (nstreams == 2 && (k & 1)) ? s1 : s0));
            CK(hipStreamSynchronize(s0));
            CK(hipStreamSynchronize(s1));
            best = std::max(best, total / (now() - t0) / 1e9);
        }
        return best;
    };
    std::printf("{\"bytes_per_measurement\": %zu", total);
    const size_t sizes[3] = {1589760, 16u << 20, 256u << 20};
    const char* names[3] = {"frame_1.6MB", "slot_16MB", "256MB"};
    for (int i = 0; i < 3; ++i) std::printf(", \"pinned_h2d_GBs_%s\": %.2f", names[i], h2d((const char*)pin, sizes[i], 1));
    std::printf(", \"pinned_h2d_GBs_slot_16MB_two_streams\": %.2f", h2d((const char*)pin, 16u << 20, 2));
    double t0 = now();
    CK(hipHostRegister(pageable, total, hipHostRegisterDefault));
    std::printf(", \"host_register_GBs\": %.2f", total / (now() - t0) / 1e9);
    for (int i = 0; i < 3; ++i) std::printf(", \"registered_h2d_GBs_%s\": %.2f", names[i], h2d(pageable, sizes[i], 1));
    CK(hipHostUnregister(pageable));
    std::printf(", \"pageable_hipMemcpy_GBs\": ");
    {
        CK(hipDeviceSynchronize());
        t0 = now();
        CK(hipMemcpy(dev, pageable, total, hipMemcpyHostToDevice));
        std::printf("%.2f", total / (now() - t0) / 1e9);
    }
    // the staging copy alone: pageable -> pinned with T threads
    for (int T : {1, 2, 4, 8, 16}) {
        double best = 0;
        for (int rep = 0; rep < 3; ++rep) {
            t0 = now();
            std::vector<std::thread> th;
            const size_t per = total / T;
            for (int t = 0; t < T; ++t) th.emplace_back([&, t] { std::memcpy((char*)pin + t * per, pageable + t * per, per); });
            for (auto& x : th) x.join();
            best = std::max(best, total / (now() - t0) / 1e9);
        }
        std::printf(", \"memcpy_to_pinned_GBs_%dthreads\": %.2f", T, best);
    }
    // ... with non-temporal stores (what the ring's staging copy uses)
    for (int T : {1, 2, 4, 8}) {
        double best = 0;
        for (int rep = 0; rep < 3; ++rep) {
            t0 = now();
            std::vector<std::thread> th;
            const size_t per = total / T;
            for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
                const char* s = pageable + t * per; char* d = (char*)pin + t * per;
                for (size_t i = 0; i + 64 <= per; i += 64) {
                    const __m128i a = _mm_loadu_si128((const __m128i*)(s + i)), b = _mm_loadu_si128((const __m128i*)(s + i + 16));
                    const __m128i c = _mm_loadu_si128((const __m128i*)(s + i + 32)), e = _mm_loadu_si128((const __m128i*)(s + i + 48));
                    _mm_stream_si128((__m128i*)(d + i), a); _mm_stream_si128((__m128i*)(d + i + 16), b);
                    _mm_stream_si128((__m128i*)(d + i + 32), c); _mm_stream_si128((__m128i*)(d + i + 48), e);
                }
                _mm_sfence();
            });
            for (auto& x : th) x.join();
            best = std::max(best, total / (now() - t0) / 1e9);
        }
        std::printf(", \"nt_copy_to_pinned_GBs_%dthreads\": %.2f", T, best);
    }
    std::printf(", \"host_cores\": %u}\n", std::thread::hardware_concurrency());
    return 0;
}
