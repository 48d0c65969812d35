// eval_tiles_test.cpp -- CPU replay of the tile kernel's addressing (amatsukaze_amd/csrc/eval_tiles.hpp, eval_pair_kernels.hip):
// for every band of a plan, stage every wave's tile through the very unit mapping the kernel uses and check that each mask pixel's
// 5x5 window reads exactly the samples the reference's CalcCorrelation5x5 reads (LogoScan.hpp:24-41: rows y-2..y+2, columns
// x-2..x+2), that every mask pixel is evaluated once, and that the score-row indices reproduce the raster order.
//   usage: eval_tiles_test            (built-in synthetic masks)
//          eval_tiles_test pos.bin    (int32 count, w, h, then count x uint32 (y << 16) | x)
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

#include "eval_tiles.hpp"

using namespace amt;

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { if (++failures < 20) { fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } } while (0)

static void replay(const char* name, const std::vector<uint32_t>& pos, int w, int h)
{
    const int count = (int)pos.size();
    const TilePlan P = build_tile_plan(pos, count, w, h);
    std::vector<int> seen(count, 0);
    int m_next = 0;
    long staged = 0, read_cycles = 0, read_tiles = 0;
    CHECK(P.tiles.size() == P.bands.size() * kTileWaves && P.sinfo.size() == P.bands.size() * kTileBandPix, "%s: table sizes", name);
    for (size_t b = 0; b < P.bands.size(); ++b) {
        const TileBandDesc& B = P.bands[b];
        CHECK(B.m0 == m_next && B.npix >= 1 && B.npix <= kTileBandPix, "%s: band %zu range", name, b);
        m_next = B.m0 + B.npix;
        std::vector<int> row_seen(kTileBandPix, 0);
        for (int wv = 0; wv < kTileWaves; ++wv) {
            const TileDesc& T = P.tiles[b * kTileWaves + wv];
            CHECK(T.npix >= 0 && T.npix <= kTileLanes && T.nrows * T.tp <= kTileCap && T.tp >= 4 * T.ncol4 && (T.tp & 1) == 0 && (T.x0 & 1) == 0,
                  "%s: band %zu tile %d geometry", name, b, wv);
            CHECK(T.nrows * T.ncol4 <= kTileLanes * kTileUnits, "%s: band %zu tile %d has %d units", name, b, wv, T.nrows * T.ncol4);
            // staging, as the kernel's lanes do it
            std::vector<int> cell(kTileCap, -1);
            for (int u = 0; u < kTileLanes * kTileUnits; ++u) {
                const TileUnit U = tile_unit(T, u, w);
                CHECK(U.y >= 0 && U.y < h && U.xs >= 0 && U.xs + 3 < w && (U.xs & 1) == 0, "%s: unit reads outside the logo (y %d xs %d)", name, U.y, U.xs);
                CHECK(U.lds >= 0 && U.lds + 3 < kTileCap && (U.lds & 1) == 0, "%s: unit store outside the plane (%d)", name, U.lds);
                for (int j = 0; j < 4; ++j) {
                    const int coord = U.y * 65536 + U.xs + j;
                    if (U.lds + j < kTileCap) {
                        CHECK(cell[U.lds + j] == -1 || cell[U.lds + j] == coord, "%s: two samples in one LDS cell", name);
                        cell[U.lds + j] = coord;
                    }
                }
            }
            staged += (long)T.nrows * T.ncol4 * 4;
            // LDS cycles of a window read (lanes 0-31, then 32-63: distinct addresses on a bank pair serialise), with and without the lanes
            // that carry no pixel: those must read a window one of their half's pixels reads (a broadcast, no cycle of their own)
            if (T.npix > 0) {
                int cyc[2] = {0, 0};
                for (int idle = 0; idle < 2; ++idle)
                    for (int hf = 0; hf < 2; ++hf) {
                        std::vector<int> seen_at[32];
                        int worst = 1;
                        for (int l = hf * 32; l < hf * 32 + 32; ++l) {
                            const uint32_t si = P.sinfo[(b * kTileWaves + wv) * kTileLanes + l];
                            if (!idle && !(si >> 31)) continue;
                            std::vector<int>& at = seen_at[si & 31];
                            if (std::find(at.begin(), at.end(), (int)(si & 0xFFFu)) == at.end()) at.push_back((int)(si & 0xFFFu));
                            worst = std::max(worst, (int)at.size());
                        }
                        cyc[idle] += worst;
                    }
                CHECK(cyc[0] == cyc[1], "%s: band %zu tile %d: lanes without a pixel cost LDS cycles (%d -> %d)", name, b, wv, cyc[0], cyc[1]);
                read_cycles += cyc[1]; ++read_tiles;
            }
            int nvalid = 0;
            for (int l = 0; l < kTileLanes; ++l) {
                const size_t slot = (b * kTileWaves + wv) * kTileLanes + l;
                const uint32_t si = P.sinfo[slot];
                const bool valid = (si >> 31) != 0;
                const int m = P.slot_pixel[slot];
                CHECK(valid == (m >= 0), "%s: slot validity", name);
                nvalid += valid;
                const int woff = (int)(si & 0xFFFu), ridx = (int)((si >> 12) & 0xFFFu);
                CHECK(woff + 4 * T.tp + 4 < kTileCap || !valid || woff + 4 * T.tp + 4 < kTileCap, "%s: window past the plane", name);
                if (!valid) { CHECK(woff + 4 * std::max(T.tp, 4) + 4 < kTileCap, "%s: idle lane window outside the plane", name); continue; }
                CHECK(m >= B.m0 && m < B.m0 + B.npix && ridx == m - B.m0, "%s: score row index", name);
                ++seen[m];
                ++row_seen[ridx];
                const int px = (int)(pos[m] & 0xFFFFu), py = (int)(pos[m] >> 16);
                for (int r = 0; r < 5; ++r)
                    for (int c = 0; c < 5; ++c) {
                        const int at = woff + r * T.tp + c;
                        CHECK(at < kTileCap && cell[at] == (py - 2 + r) * 65536 + (px - 2 + c), "%s: band %zu tile %d lane %d window (%d,%d) reads the wrong sample",
                              name, b, wv, l, r, c);
                    }
            }
            CHECK(nvalid == T.npix, "%s: band %zu tile %d: %d lanes carry a pixel, the tile says %d", name, b, wv, nvalid, T.npix);
        }
        for (int i = 0; i < B.npix; ++i) CHECK(row_seen[i] == 1, "%s: band %zu row slot %d written %d times", name, b, i, row_seen[i]);
    }
    CHECK(m_next == count, "%s: bands cover %d of %d pixels", name, m_next, count);
    for (int m = 0; m < count; ++m) CHECK(seen[m] == 1, "%s: pixel %d evaluated %d times", name, m, seen[m]);
    printf("%-28s %6d px  %3zu bands  %.2f staged samples per mask pixel  %.2f LDS cycles per window read (2 = none lost to bank conflicts)\n", name, count,
           P.bands.size(), count ? (double)staged / count : 0.0, read_tiles ? (double)read_cycles / read_tiles : 0.0);
}

static std::vector<uint32_t> from_mask(const std::vector<uint8_t>& mask, int w, int h)
{
    std::vector<uint32_t> pos;
    for (int y = 2; y < h - 2; ++y)
        for (int x = 2; x < w - 2; ++x)
            if (mask[x + y * w]) pos.push_back(((uint32_t)y << 16) | (uint32_t)x);
    return pos;
}

int main(int argc, char** argv)
{
    if (argc > 1) {
        FILE* f = fopen(argv[1], "rb");
        int hdr[3];
        if (!f || fread(hdr, 4, 3, f) != 3) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
        std::vector<uint32_t> pos(hdr[0]);
        if (fread(pos.data(), 4, pos.size(), f) != pos.size()) return 2;
        fclose(f);
        replay(argv[1], pos, hdr[1], hdr[2]);
    } else {
        std::mt19937 rng(12345);
        struct Case { const char* name; int w, h; double density; int kind; };
        const Case cases[] = {
            {"random 35% 256x128", 256, 128, 0.35, 0}, {"random 2% 256x128", 256, 128, 0.02, 0}, {"full 256x128", 256, 128, 1.0, 0},
            {"strokes 256x128", 256, 128, 0.0, 1}, {"columns 256x128", 256, 128, 0.0, 2}, {"wide 1022x40", 1022, 40, 0.3, 0},
            {"tall 22x400", 22, 400, 0.5, 0}, {"tiny 6x6", 6, 6, 1.0, 0}, {"narrow 6x200", 6, 200, 1.0, 0}, {"w%4==2 66x50", 66, 50, 0.4, 0},
            {"one pixel", 64, 64, 0.0, 3}, {"empty", 64, 64, 0.0, 4}, {"diagonal 300x300", 300, 300, 0.0, 5},
        };
        for (const Case& c : cases) {
            std::vector<uint8_t> mask((size_t)c.w * c.h, 0);
            std::uniform_real_distribution<double> U(0, 1);
            for (int y = 0; y < c.h; ++y)
                for (int x = 0; x < c.w; ++x) {
                    bool on = false;
                    switch (c.kind) {
                    case 0: on = U(rng) < c.density; break;
                    case 1: on = ((x / 7 + y / 5) % 3 == 0) && U(rng) < 0.8; break;       // text-like strokes
                    case 2: on = x % 37 < 3; break;
                    case 3: on = x == 31 && y == 17; break;
                    case 4: on = false; break;
                    case 5: on = (x - y) % 97 == 0 || x == y; break;
                    }
                    mask[x + (size_t)y * c.w] = on;
                }
            replay(c.name, from_mask(mask, c.w, c.h), c.w, c.h);
        }
    }
    // the scan kernel's workgroup -> (logo, frame group) map (eval_tiles.hpp): every pair exactly once, a group's logos on ONE XCD
    // (workgroup id mod 8) and back to back in that XCD's dispatch order, ids of the padding past the last group
    for (int nlogos = 1; nlogos <= 5; ++nlogos)
        for (int ngroups : {1, 2, 7, 8, 9, 63, 64, 1250, 1251}) {
            const long long grid = wg_grid_shared_rows(ngroups, nlogos);
            std::vector<int> hit((size_t)nlogos * ngroups, 0), xcd((size_t)ngroups, -1), first((size_t)ngroups, -1);
            CHECK(grid % kXcds == 0 && grid >= (long long)nlogos * ngroups && grid < (long long)nlogos * (ngroups + kXcds), "wg map: grid %lld", grid);
            for (int bid = 0; bid < grid; ++bid) {
                const WgMap m = wg_map_shared_rows(bid, nlogos, ngroups);
                CHECK(m.logo >= 0 && m.logo < nlogos && m.grp >= 0, "wg map: range");
                if (m.grp >= ngroups) continue;
                ++hit[(size_t)m.logo * ngroups + m.grp];
                if (xcd[m.grp] < 0) { xcd[m.grp] = bid % kXcds; first[m.grp] = bid; }
                CHECK(xcd[m.grp] == bid % kXcds, "wg map: group %d on two XCDs", m.grp);
                CHECK(bid == first[m.grp] + m.logo * kXcds, "wg map: group %d logo %d not back to back on its XCD", m.grp, m.logo);
            }
            for (int v : hit) CHECK(v == 1, "wg map: a (logo, group) pair %d times", v);
        }
    if (failures) { fprintf(stderr, "%d check(s) failed\n", failures); return 1; }
    printf("ok\n");
    return 0;
}
