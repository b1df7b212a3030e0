// tools/experimental/lbm_slide_plan.hpp (EXPERIMENT of round 4, see lbm_kernel_slide.hpp) — host side of the sliding register kernel (lbm_kernel_slide.hpp): who does which rows.
//
// k_steps_col's blocks are persistent — one segment of one column strip each, all resident at once (two per CU) — so the
// slowest block is the launch, and the boundary logic is 1.3-1.5 x the cost of a plain fluid cell. The work is therefore
// dealt here, on the host, once per (grid, depth):
//   * column strips of 64 - 2(D-1) output columns; a strip that touches x = 0 or x = nx-1 within its halo is GENERAL for all its rows;
//   * an interior strip whose halo meets the cylinder's bounding box is GENERAL for the rows around the box (with the margin the
//     device-side check tile_near_cylinder needs: a lean tile must never see the box) and LEAN below and above it;
//   * every other strip is LEAN for all rows — walls included (a wave-uniform swap in the lean path);
//   * every contiguous range is cut into pieces of equal weight: rows x 1 (lean) or x GENERAL_WEIGHT, so that all blocks finish together;
//     the lean strips share their cuts, so x-neighbours work on the same rows at the same time (their common halo meets in L2);
//   * order: lean pieces piece-row-major (an XCD's run of the walk = x-neighbours), the general ones spread evenly between them.
// Pure C++ (no HIP): unit-tested on the CPU (tests/test_slide_plan_cpu.py through lbm_debug_slide_plan).
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

namespace lbmk {

struct SlideSegHost { int bx, ya, yb, general; };

struct SlideGeom {
    int nx, y_lo, y_cnt;        // columns; local rows [y_lo, y_lo + y_cnt) of the launch
    int y_start, ny_glob;       // global row of local row 0; global rows
    int cyl_x, cyl_y, cyl_r;    // integer cylinder (LBMConfig.h:61-63); cyl_r < 0: none
    int depth;                  // iterations per launch (D)
    int tile_rows;              // H = R * NW
    int slots;                  // resident blocks the device holds (2 x CUs)
};

constexpr double SLIDE_GENERAL_WEIGHT = 1.5;

// the strip's halo-inclusive column range [Xo - HW, Xo + OW - 1 + HW] lies strictly inside the lattice (no inlet / outlet column, no ghost column)
inline bool slide_strip_interior(const SlideGeom& g, int bx) {
    const int HW = g.depth - 1, OW = 64 - 2 * HW, Xo = bx * OW;
    return Xo >= HW + 1 && Xo + OW + HW <= g.nx - 1;
}
// ... and meets the cylinder's box (the same test as tile_near_cylinder's column half: r = cyl_r + 1)
inline bool slide_strip_meets_cylinder(const SlideGeom& g, int bx) {
    if (g.cyl_r < 0) return false;
    const int HW = g.depth - 1, OW = 64 - 2 * HW, Xo = bx * OW, r = g.cyl_r + 1;
    return Xo - HW <= g.cyl_x + r && Xo + OW - 1 + HW >= g.cyl_x - r;
}

inline std::vector<SlideSegHost> slide_plan(const SlideGeom& g) {
    const int HW = g.depth - 1, OW = 64 - 2 * HW, H = g.tile_rows;
    const int nbx = (g.nx + OW - 1) / OW;
    const int y0 = g.y_lo, y1 = g.y_lo + g.y_cnt;
    struct Range { int bx, ya, yb, general; };
    std::vector<Range> ranges;
    // rows (local) around the cylinder that a lean tile of any segment below / above must stay clear of
    int box_lo = y1, box_hi = y1;      // [box_lo, box_hi): general rows of a cylinder strip
    if (g.cyl_r >= 0) {
        const int r = g.cyl_r + 1;
        const int lo = g.cyl_y - r - g.y_start, hi = g.cyl_y + r - g.y_start;      // local rows, inclusive
        box_lo = std::clamp(lo - 2 * H, y0, y1);
        box_hi = std::clamp(hi + H + 2 * HW + 1, y0, y1);
        if (hi + H + 2 * HW + 1 <= y0 || lo - 2 * H >= y1) { box_lo = y1; box_hi = y1; }      // the box is not in this launch's rows
    }
    for (int bx = 0; bx < nbx; ++bx) {
        if (!slide_strip_interior(g, bx)) { ranges.push_back({bx, y0, y1, 1}); continue; }
        if (slide_strip_meets_cylinder(g, bx) && box_lo < box_hi) {
            if (box_lo > y0) ranges.push_back({bx, y0, box_lo, 0});
            ranges.push_back({bx, box_lo, box_hi, 1});
            if (box_hi < y1) ranges.push_back({bx, box_hi, y1, 0});
        } else {
            ranges.push_back({bx, y0, y1, 0});
        }
    }
    double weight = 0;
    for (const Range& r : ranges) weight += (r.yb - r.ya) * (r.general ? SLIDE_GENERAL_WEIGHT : 1.0);
    // pieces per range: as many as keep every piece at the target weight; shrink the count until the blocks fit the slots
    auto pieces_of = [&](const Range& r, double target) {
        const double wr = (r.yb - r.ya) * (r.general ? SLIDE_GENERAL_WEIGHT : 1.0);
        int n = (int)std::floor(wr / target + 0.5);
        n = std::min(n, std::max(1, (r.yb - r.ya) / H));      // at least one tile of rows per piece
        return std::max(1, n);
    };
    double target = weight / std::max(1, g.slots);
    int total = 0;
    for (int it = 0; it < 200; ++it) {
        total = 0;
        for (const Range& r : ranges) total += pieces_of(r, target);
        if (total <= g.slots) break;
        target *= 1.02;
    }
    struct Piece { SlideSegHost s; int idx; };
    std::vector<Piece> lean, general;
    for (const Range& r : ranges) {
        const int n = pieces_of(r, target), len = r.yb - r.ya;
        for (int p = 0; p < n; ++p) {
            const int a = r.ya + (int)((long)len * p / n), b = r.ya + (int)((long)len * (p + 1) / n);
            (r.general ? general : lean).push_back({{r.bx, a, b, r.general}, p});
        }
    }
    // lean: by the rows they start at, then by strip (x-neighbours adjacent); general: spread evenly between them
    std::stable_sort(lean.begin(), lean.end(), [](const Piece& p, const Piece& q) { return p.s.ya != q.s.ya ? p.s.ya < q.s.ya : p.s.bx < q.s.bx; });
    std::vector<SlideSegHost> out;
    out.reserve(lean.size() + general.size());
    const size_t nl = lean.size(), ng = general.size();
    size_t gi = 0;
    for (size_t i = 0; i < nl; ++i) {
        while (gi < ng && gi * (nl + 1) <= i * ng) out.push_back(general[gi++].s);
        out.push_back(lean[i].s);
    }
    while (gi < ng) out.push_back(general[gi++].s);
    return out;
}

}  // namespace lbmk
