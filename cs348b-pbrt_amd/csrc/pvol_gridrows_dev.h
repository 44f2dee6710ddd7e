// pvol_gridrows_dev.h -- which photons a ball query has to look at: the rows of grid cells that can touch the ball, one row per
// lane, each ONE contiguous range of the sorted photon arrays (pvol_grid.hip).  Shared by lphoton() (pvol_march.hip) and the
// bucket staging (pvol_group_dev.h).
//
// Coarse level: rows of whole cells along x, (2R+1)^2 of them.  Second level (clumpy maps only, GridView::subStart): a ball
// smaller than 3/4 of a cell walks rows of 4x4x4 SUB-cells instead -- a sub-row lives inside one cell, so a row of the ball's
// (y,z) footprint splits into one segment per cell it crosses in x (at most three).  Pruning bounds carry a slack of 1e-4 cell
// (photons are assigned to sub-cells by a floor() that may round across a boundary); acceptance is always the exact d2 < T.
#ifndef PVOL_GRIDROWS_DEV_H
#define PVOL_GRIDROWS_DEV_H

// A photon grid as the search code sees it (the volume map of DevScene, or the caustic map of DevSurface)
struct GridView {
    float cellSize, invCell;
    float gridLo[3];
    int32_t gdim[3];
    const uint32_t *cellStart;
    const uint32_t *subStart;   // 0: no second level
    const float4 *pos4;
};
__device__ __forceinline__ GridView volume_grid(const DevScene &S) {
    GridView g;
    g.cellSize = S.cellSize; g.invCell = S.invCell;
    for (int i = 0; i < 3; ++i) { g.gridLo[i] = S.gridLo[i]; g.gdim[i] = S.gdim[i]; }
    g.cellStart = S.cellStart; g.subStart = S.subStart; g.pos4 = S.pos4;
    return g;
}

struct GridRows {
    bool fine;
    float cw, invw;      // edge of the cells the rows are made of (cell, or cell / 4)
    int cy, cz;          // the centre's row coordinates (in those cells)
    int Rt, side;        // rows cover cy - Rt .. cy + Rt, cz - Rt .. cz + Rt
    int cx0, nseg;       // fine: first coarse cell in x and how many the ball crosses
    int nrows;
};
// ringMax: upper bound of Rt for the coarse level (radius <= maxDist), <= 0: none.  FINE = false compiles the second level out
// (the hot kernels' lookups are wider than a cell; the coarse level alone is always complete).
template <bool FINE>
__device__ __forceinline__ GridRows grid_rows(const GridView &g, V3 c, float R, int ringMax) {
    GridRows P;
    const float eps = g.cellSize * 1e-4f;
    P.fine = FINE && g.subStart != 0 && R < 0.75f * g.cellSize;
    P.cw = P.fine ? g.cellSize * 0.25f : g.cellSize;
    P.invw = P.fine ? g.invCell * 4.f : g.invCell;
    P.cy = (int)floorf((c.y - g.gridLo[1]) * P.invw);
    P.cz = (int)floorf((c.z - g.gridLo[2]) * P.invw);
    P.Rt = (int)ceilf(R * P.invw + 1e-3f);
    if (!P.fine && ringMax > 0) P.Rt = min(P.Rt, ringMax);
    P.side = 2 * P.Rt + 1;
    P.cx0 = 0; P.nseg = 1;
    if (FINE && P.fine) {
        P.cx0 = max(0, (int)floorf((c.x - R - eps - g.gridLo[0]) * g.invCell));
        const int cx1 = min(g.gdim[0] - 1, (int)floorf((c.x + R + eps - g.gridLo[0]) * g.invCell));
        P.nseg = max(1, cx1 - P.cx0 + 1);
    }
    P.nrows = P.side * P.side * P.nseg;
    return P;
}
// row r (< P.nrows) of the ball (c, T = radius^2): the photons [start, start + rlen) or rlen == 0
template <bool FINE>
__device__ __forceinline__ void grid_row_range(const GridView &g, const GridRows &P, int r, V3 c, float T, uint32_t *start, uint32_t *rlen) {
    *start = 0u; *rlen = 0u;
    if (r >= P.nrows) return;
    const float eps = g.cellSize * 1e-4f;
    int seg = 0, rr = r;
    if (FINE && P.nseg > 1) { rr = r / P.nseg; seg = r - rr * P.nseg; }
    const int iy = rr / P.side;
    const int y = P.cy + iy - P.Rt, z = P.cz + (rr - iy * P.side) - P.Rt;
    const int ny = (FINE && P.fine) ? g.gdim[1] * 4 : g.gdim[1], nz = (FINE && P.fine) ? g.gdim[2] * 4 : g.gdim[2];
    if (y < 0 || y >= ny || z < 0 || z >= nz) return;
    const float ylo = g.gridLo[1] + y * P.cw, zlo = g.gridLo[2] + z * P.cw;
    const float ddy = fmaxf(0.f, fmaxf(ylo - c.y, c.y - (ylo + P.cw)) - eps);
    const float ddz = fmaxf(0.f, fmaxf(zlo - c.z, c.z - (zlo + P.cw)) - eps);
    const float rd2 = ddy * ddy + ddz * ddz;
    if (!(rd2 < T)) return;
    const float hw = sqrtf(fmaxf(0.f, T - rd2)) + eps;
    int x0 = (int)floorf((c.x - hw - g.gridLo[0]) * P.invw), x1 = (int)floorf((c.x + hw - g.gridLo[0]) * P.invw);
    if (!FINE || !P.fine) {
        x0 = max(x0, 0);
        x1 = min(x1, g.gdim[0] - 1);
        if (x0 > x1) return;
        const size_t base = ((size_t)z * g.gdim[1] + y) * g.gdim[0];
        *start = g.cellStart[base + x0];
        *rlen = g.cellStart[base + x1 + 1] - *start;
    } else {
        const int cx = P.cx0 + seg;
        if (cx >= g.gdim[0]) return;
        const int sx0 = max(x0, 4 * cx) - 4 * cx, sx1 = min(x1, 4 * cx + 3) - 4 * cx;
        if (sx0 > sx1) return;
        const size_t kb = (((size_t)(z >> 2) * g.gdim[1] + (y >> 2)) * g.gdim[0] + cx) * 64 + (size_t)(((z & 3) * 4 + (y & 3)) * 4);
        *start = g.subStart[kb + sx0];
        *rlen = g.subStart[kb + sx1 + 1] - *start;
    }
}
#endif
