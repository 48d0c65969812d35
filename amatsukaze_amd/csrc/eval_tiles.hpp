// eval_tiles.hpp -- host-side plan of the tile evaluation kernels (eval_pair_kernels.hip, eval_linear_kernels.hip): bands of
// raster-consecutive mask pixels, each split into up to kTileWaves wave tiles.  Pure C++ (no HIP) so that tests/cpp/eval_tiles_test.cpp can replay the
// kernel's addressing on the CPU.
//
// Why tiles: CorrelationScore (LogoScan.hpp:288-318) adds the per-pixel terms in raster order, so the terms of a band of
// <= kTileBandPix raster-consecutive mask pixels go to an LDS row that one lane adds front to back.  WHICH thread evaluates a mask
// pixel is free, though: a band's pixels are sorted by column and dealt to the evaluation waves 64 at a time, and every wave stages
// only the bounding box of its own pixels' 5x5 windows -- a tile of ~36 x 10 samples instead of a share of the band's full-width
// rows -- into LDS that no other wave reads.  Staging, window reads and evaluation of a wave then need no workgroup barrier; the
// waves meet once per band (G frames), when the summing wave takes over the band's rows.  (The linear analysis kernel uses the
// tiles alone: its summation order is free, so its waves never meet.)
#pragma once

#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <vector>

namespace amt {

#ifndef AMT_TILE_WAVES
#define AMT_TILE_WAVES 11
#endif
constexpr int kTileWaves = AMT_TILE_WAVES;            // evaluation waves per workgroup = tiles per band
#ifndef AMT_TILE_G
#define AMT_TILE_G 8
#endif
constexpr int kTileMaxFrames = AMT_TILE_G;            // frames per workgroup: 2 * kTileMaxFrames score rows per band, twice (double buffer)
constexpr int kTileLanes = 64;                        // mask pixels per tile
constexpr int kTileBandPix = kTileWaves * kTileLanes; // 704: mask pixels per band, one LDS score row per (frame, fade)
constexpr int kTileCap = 512;                         // {s, bg} pairs an LDS tile plane holds (4 KB)
constexpr int kTileUnits = kTileCap / 4 / kTileLanes; // staging units (one row x four columns) per lane per frame: 2

// one wave's tile of one band; read with scalar loads (32 bytes)
struct TileDesc {
    int x0, y0;          // tile origin in logo coordinates (x0 even: the units' 16-byte LDS stores stay aligned)
    int nrows, ncol4;    // rows, four-column units per row
    int tp;              // LDS row pitch in pairs (even, >= 4 * ncol4), chosen on the host for the fewest bank conflicts
    int npix;            // mask pixels of this tile = active lanes (<= 64); 0: the wave idles through this band
    int rcp;             // unit u -> row u / ncol4 == (u * rcp) >> 16 for every u < 64 * kTileUnits
    int pad;
};
struct TileBandDesc {
    int m0, npix;        // the band's mask pixels [m0, m0 + npix) in raster order
};
#if defined(__HIPCC__)
#define AMT_TILE_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define AMT_TILE_HD inline
#endif
// staging unit u of a tile (one tile row x four columns; units beyond the tile's last repeat it): the logo row and first column
// it covers and its pair offset in the tile plane.  Shared by the kernel and the CPU replay of its addressing.
struct TileUnit { int y, xs, lds; };
AMT_TILE_HD TileUnit tile_unit(const TileDesc& T, int u, int w)
{
    const int last = T.nrows * T.ncol4 - 1;
    u = u < last ? u : (last < 0 ? 0 : last);
    const int row = (u * T.rcp) >> 16;
    const int c4 = u - row * T.ncol4;
    const int x = T.x0 + 4 * c4;
    const int xs = x < w - 4 ? x : w - 4;             // a ragged right edge: the last unit moves left
    return TileUnit{T.y0 + row, xs, row * T.tp + (xs - T.x0)};
}

// Workgroup -> (logo, frame group) of the scan's tile kernel.  Workgroups are dealt to the 8 XCDs round-robin by their linear id, each XCD
// has its own 4 MB L2, and the candidate logos of a scan all read the same rectangle of the same frames: the workgroups of ONE frame group
// sit on one XCD, back to back in dispatch order, so its rows are fetched once per group instead of once per logo.
// Workgroup id = ((group / 8) * nlogos + logo) * 8 + group % 8; the grid is rounded up to whole blocks of 8 groups, ids beyond the last
// group leave at once.  Measured (round 6, 3 logos, 4 096 frames): 160 KB instead of 179 KB of L2 misses per frame, 2.49 against 2.51 ms --
// what is left is the logos' tap and scale tables (4.1 MB per logo, walked once per workgroup), not the rows.
// The LINEAR kernel keeps logo-major ids on purpose: its deint / top / bottom logos have different tables (1.9 + 0.95 + 0.95 MB with the
// coefficient planes), logo-major dispatch keeps ONE of them in an XCD's L2 at a time, and putting the three side by side for the sake
// of the shared rows costs more table misses than it saves row misses (269 KB against 176 KB per frame, 2.12 against 1.92 ms).
constexpr int kXcds = 8;
struct WgMap { int logo, grp; };
AMT_TILE_HD WgMap wg_map_shared_rows(int bid, int nlogos, int ngroups)
{
#ifdef AMT_WG_PLAIN_MAP             /* (the A/B of the map: logo-major ids) */
    if (bid >= nlogos * ngroups) return WgMap{0, ngroups};
    return WgMap{bid / ngroups, bid % ngroups};
#endif
    (void)ngroups;
    const int x = bid & (kXcds - 1), r = bid >> 3;
    const int blk = r / nlogos;
    return WgMap{r - blk * nlogos, blk * kXcds + x};
}
inline long long wg_grid_shared_rows(long long ngroups, int nlogos) { return (ngroups + kXcds - 1) / kXcds * kXcds * nlogos; }

// per lane of a tile (slot = (band * kTileWaves + wave) * 64 + lane)
inline uint32_t tile_slot_info(int woff, int ridx, bool valid) { return (uint32_t)woff | ((uint32_t)ridx << 12) | (valid ? 0x80000000u : 0u); }

struct TilePlan {
    std::vector<TileBandDesc> bands;
    std::vector<TileDesc> tiles;          // bands.size() * kTileWaves
    std::vector<uint32_t> sinfo;          // per slot: window offset in the tile (pairs, bits 0-11), index in the band's score row (bits 12-23), valid (bit 31)
    std::vector<int> slot_pixel;          // per slot: mask pixel m, -1 for an idle lane
    int nslots() const { return (int)sinfo.size(); }
};

namespace tiles_detail {

struct Px { int x, y, m; };

// LDS cycles of one 8-byte window read of a wave: lanes 0-31 and 32-63 are served as two groups, a bank pair holds pair index mod 32,
// distinct addresses on one bank pair serialise (MI355X_MICROARCH.md, LDS: ds_read_b64).  The 25 reads of a window differ by a
// constant, so one count covers them all.
//
// WHICH lane of a tile evaluates which of its pixels is free, and so is the half of the wave a pixel sits in: split_lanes() deals the
// pixels of every bank pair over the two halves so that the sum of the halves' worst bank loads is as small as the pitch allows (2 = no
// conflict at all when no bank pair holds more than two of the tile's windows).  Dealing lanes by column, as rounds 2-5 did, put a
// stroke's pixels of rows y and y + 4 (32 / pitch 8 apart) into the same half: 3.65 cycles per read on the bench logo, 2.2 now.
// half[i] = 0 / 1 for pixel i of g; returns the cycles.
inline int split_lanes(const std::vector<Px>& g, int x0, int y0, int tp, std::vector<uint8_t>* half)
{
    const int n = (int)g.size();
    int k[32] = {0};
    std::vector<int> bank(n);
    for (int i = 0; i < n; ++i) { bank[i] = ((g[i].y - 2 - y0) * tp + (g[i].x - 2 - x0)) & 31; ++k[bank[i]]; }
    for (int total = 2;; ++total)
        for (int ca = (total + 1) / 2; ca < total; ++ca) {
            const int cb = total - ca;                                           // half A may carry ca windows of a bank pair, half B cb
            int lo = 0, hi = 0;
            bool ok = true;
            for (int b = 0; b < 32; ++b) { ok = ok && k[b] <= total; lo += std::max(0, k[b] - cb); hi += std::min(k[b], ca); }
            const int smin = std::max(lo, n - 32), smax = std::min(hi, 32);      // pixels in half A
            if (!ok || smin > smax) continue;
            if (half) {
                int a[32], sum = 0;
                for (int b = 0; b < 32; ++b) { a[b] = std::max(0, k[b] - cb); sum += a[b]; }
                const int want = std::min(std::max((n + 1) / 2, smin), smax);
                for (bool moved = true; sum < want && moved;) {
                    moved = false;
                    for (int b = 0; b < 32 && sum < want; ++b)
                        if (a[b] < std::min(k[b], ca)) { ++a[b]; ++sum; moved = true; }
                }
                half->assign(n, 1);
                for (int i = 0; i < n; ++i)
                    if (a[bank[i]] > 0) { (*half)[i] = 0; --a[bank[i]]; }
            }
            return total;
        }
}

struct Geometry { int x0, y0, nrows, ncol4; };
inline Geometry bbox(const std::vector<Px>& g, size_t a, size_t b)
{
    int xmin = 1 << 30, xmax = -1, ymin = 1 << 30, ymax = -1;
    for (size_t i = a; i < b; ++i) {
        xmin = std::min(xmin, g[i].x); xmax = std::max(xmax, g[i].x);
        ymin = std::min(ymin, g[i].y); ymax = std::max(ymax, g[i].y);
    }
    Geometry G;
    G.x0 = (xmin - 2) & ~1;
    G.y0 = ymin - 2;
    G.nrows = ymax + 2 - G.y0 + 1;
    G.ncol4 = (xmax + 2 - G.x0 + 1 + 3) / 4;
    return G;
}
inline bool fits(const Geometry& G) { return G.nrows * G.ncol4 * 4 <= kTileCap && G.nrows * G.ncol4 <= kTileLanes * kTileUnits; }

} // namespace tiles_detail

// pos[m] = (y << 16) | x of mask pixel m in raster order (MaskTables::pos); every pixel has 2 <= x < w-2, 2 <= y < h-2, w >= 5
inline TilePlan build_tile_plan(const std::vector<uint32_t>& pos, int count, int w, int h)
{
    using namespace tiles_detail;
    (void)h;
    if (w < 5) throw std::runtime_error("logo too narrow for the tile kernel");
    TilePlan P;
    for (int m = 0; m < count;) {
        int n = std::min(kTileBandPix, count - m);
        std::vector<std::vector<Px>> groups;
        for (;;) {
            // the band's pixels by column, then row: a wave's 64 pixels cover few columns and all of the band's rows
            std::vector<Px> px(n);
            for (int i = 0; i < n; ++i) px[i] = Px{(int)(pos[m + i] & 0xFFFFu), (int)(pos[m + i] >> 16), m + i};
            std::stable_sort(px.begin(), px.end(), [](const Px& a, const Px& b) { return a.x < b.x; });
            groups.clear();
            for (size_t a = 0; a < px.size();) {
                size_t b = a + 1;
                while (b < px.size() && b - a < (size_t)kTileLanes && fits(bbox(px, a, b + 1))) ++b;
                groups.emplace_back(px.begin() + a, px.begin() + b);
                a = b;
            }
            if ((int)groups.size() <= kTileWaves) break;
            n = std::max(1, n - std::max(1, n / 8));       // sparse stretch: a shorter band
        }
        // Which wave takes which tile is free.  Waves are dealt to the CU's four SIMDs round-robin (wave w on SIMD w % 4: with eleven
        // evaluation waves three each on SIMDs 0-2, two and the summing wave on SIMD 3), a SIMD issues for one wave at a time, and a
        // tile with more than 64 staging units costs a second staging pass: the tiles go heaviest first to the SIMD with the least
        // work so far (scan kernel 2.759 -> 2.712 ms per 10 000 frames; the records do not depend on it).
        {
            std::vector<size_t> order(groups.size());
            for (size_t i = 0; i < order.size(); ++i) order[i] = i;
            auto cost = [&](size_t i) { const Geometry G = bbox(groups[i], 0, groups[i].size()); return 131 + 31 * (G.nrows * G.ncol4 > kTileLanes ? 2 : 1); };
            std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cost(a) > cost(b); });
            int load[4] = {0, 0, 0, 0}, used[4] = {0, 0, 0, 0}, cap[4] = {0, 0, 0, 0};
            for (int w = 0; w < kTileWaves; ++w) ++cap[w % 4];
            std::vector<std::vector<Px>> placed(kTileWaves);
            for (size_t i : order) {
                int best = -1;
                for (int k = 0; k < 4; ++k)
                    if (used[k] < cap[k] && (best < 0 || load[k] < load[best])) best = k;
                placed[best + 4 * used[best]] = groups[i];
                load[best] += cost(i); ++used[best];
            }
            groups.swap(placed);                                  // (a wave without a tile keeps an empty group: npix 0, it idles through the band)
        }
        const int band = (int)P.bands.size();
        P.bands.push_back(TileBandDesc{m, n});
        P.tiles.resize((size_t)(band + 1) * kTileWaves, TileDesc{0, 0, 0, 1, 4, 0, 65536, 0});
        P.sinfo.resize((size_t)(band + 1) * kTileBandPix, tile_slot_info(0, 0, false));
        P.slot_pixel.resize((size_t)(band + 1) * kTileBandPix, -1);
        for (size_t gi = 0; gi < groups.size(); ++gi) {
            const std::vector<Px>& g = groups[gi];
            if (g.empty()) continue;
            const Geometry G = bbox(g, 0, g.size());
            // pitch: the candidate with the fewest LDS cycles per window read, the narrowest among equals
            int best_tp = G.ncol4 * 4, best_cyc = 1 << 30;
            for (int tp = G.ncol4 * 4; tp * G.nrows <= kTileCap && tp < G.ncol4 * 4 + 32; tp += 2) {
                const int c = split_lanes(g, G.x0, G.y0, tp, nullptr);
                if (c < best_cyc) { best_cyc = c; best_tp = tp; }
            }
            std::vector<uint8_t> half;
            split_lanes(g, G.x0, G.y0, best_tp, &half);
            TileDesc& T = P.tiles[(size_t)band * kTileWaves + gi];
            T.x0 = G.x0; T.y0 = G.y0; T.nrows = G.nrows; T.ncol4 = G.ncol4; T.tp = best_tp; T.npix = (int)g.size();
            T.rcp = (65536 + G.ncol4 - 1) / G.ncol4;
            for (int u = 0; u < kTileLanes * kTileUnits; ++u)
                if (((u * T.rcp) >> 16) != u / G.ncol4) throw std::runtime_error("tile plan: unit row magic");
            // lanes 0-31 take half 0's pixels, lanes 32-63 half 1's; a lane without a pixel reads the window of a pixel of its own half
            // (the same address: a broadcast, no bank cycle of its own) with zero taps
            int next[2] = {0, kTileLanes / 2}, filler[2] = {-1, -1};
            for (size_t i = 0; i < g.size(); ++i) {
                const int l = next[half[i]]++;
                const size_t slot = ((size_t)band * kTileWaves + gi) * kTileLanes + l;
                const int woff = (g[i].y - 2 - G.y0) * best_tp + (g[i].x - 2 - G.x0);
                if (l >= kTileLanes / 2 * (half[i] + 1)) throw std::runtime_error("tile plan: more than 32 pixels in half a wave");
                if (woff < 0 || woff + 4 * best_tp + 4 > G.nrows * best_tp - 1 || G.nrows * best_tp > kTileCap) throw std::runtime_error("tile plan: window outside its tile");
                P.sinfo[slot] = tile_slot_info(woff, g[i].m - m, true);
                P.slot_pixel[slot] = g[i].m;
                if (filler[half[i]] < 0) filler[half[i]] = woff;
            }
            for (int hf = 0; hf < 2; ++hf) {
                const int woff = filler[hf] >= 0 ? filler[hf] : filler[1 - hf];
                for (int l = next[hf]; l < kTileLanes / 2 * (hf + 1); ++l)
                    P.sinfo[((size_t)band * kTileWaves + gi) * kTileLanes + l] = tile_slot_info(woff, 0, false);
            }
        }
        m += n;
    }
    return P;
}

} // namespace amt
