// abi_tick.cpp -- the C ABI alone (no filter layer, no synthetic source): how long does one analysis block take when the host pauses
// between blocks / between the block's uploads?  g++ -O2 -std=c++17 -I include -o abi_tick abi_tick.cpp -L amatsukaze_amd -lamt_gpu
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "amt_gpu.h"

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void busy_ms(double ms) { const double t0 = now_ms(); while (now_ms() - t0 < ms) { } }

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: abi_tick logo.lgd\n"); return 2; }
    AmtGpuContext* c = amtgpu_context_create(0);
    if (!c) { std::fprintf(stderr, "no context\n"); return 1; }
    AmtGpuAnalyze* an = amtgpu_analyze_create(c, argv[1], 0.35f);
    if (!an) { std::fprintf(stderr, "%s\n", amtgpu_last_error(c)); return 1; }
    const int W = 1440, H = 1080, pitch = 1472, N = 256, rows = 128, row0 = 64, col0 = 1120, roww = 256;
    const size_t plane = (size_t)pitch * rows;                   // resident part of a frame: the rectangle's rows
    uint8_t* d = (uint8_t*)amtgpu_device_alloc(c, plane * N + 64);
    std::vector<std::vector<uint8_t>> frames(N, std::vector<uint8_t>((size_t)pitch * H, 100));
    std::vector<float> out((size_t)N * 33);
    (void)W;
    auto block = [&](double gap_between_uploads, bool uploads) {
        const double t0 = now_ms();
        if (uploads) {
            for (int g0 = 0; g0 < N; g0 += 64) {
                std::vector<const void*> src;
                for (int i = 0; i < 64; ++i) src.push_back(frames[g0 + i].data() + (size_t)row0 * pitch + col0);
                if (g0 == 128 && gap_between_uploads > 0) busy_ms(gap_between_uploads);
                if (!amtgpu_frames_upload_gather(c, d + plane * g0 + col0, pitch, src.data(), pitch, roww, rows, 64)) return -1.0;
            }
            if (!amtgpu_frames_upload_wait(c)) return -1.0;
        }
        if (!amtgpu_analyze_batch_host(an, d - (size_t)row0 * pitch, (int64_t)plane, pitch, 8, N, out.data())) return -1.0;
        return now_ms() - t0 - (uploads ? gap_between_uploads : 0.0);
    };
    std::printf("{");
    bool first = true;
    for (int up = 0; up < 2; ++up)
        for (double gap_in : {0.0, 8.0})
            for (double gap_between : {0.0, 10.0}) {
                if (!up && gap_in > 0) continue;
                for (int w = 0; w < 3; ++w) block(0, up);
                std::vector<double> t;
                for (int i = 0; i < 12; ++i) { if (gap_between > 0) busy_ms(gap_between); t.push_back(block(gap_in, up)); }
                std::printf("%s\"%s_gapin%.0f_gapbetween%.0f\": [", first ? "" : ", ", up ? "upload+analysis" : "analysis_only", gap_in, gap_between);
                for (size_t i = 0; i < t.size(); ++i) std::printf("%s%.2f", i ? ", " : "", t[i]);
                std::printf("]");
                first = false;
            }
    std::printf("}\n");
    amtgpu_analyze_destroy(an);
    amtgpu_device_free(c, d);
    amtgpu_context_destroy(c);
    return 0;
}
