// pvol_fixgrp_dev.h -- li_fixup_group_kernel: the exact k-nearest lookups li_group_kernel hands over when nused lies beyond its
// bucket plan (C3: nused 500), one lookup per LANE, the 64 neighbouring lookups of a list run sharing one staged bucket.
//
// Why: a k = 500 lookup reads 500 photon rows of 128 B.  One wave per lookup (li_fixup_kernel) moves 64 KB per lookup and
// spends its time on dependent memory round trips; the 64 lookups of a group at one march step (neighbouring rays, same depth)
// want nearly the same photons, so here the union of their search balls is staged ONCE, every lane selects its own k nearest
// out of it, and positions and flux rows are read once per 64 lookups through the scalar cache.
//
// The list comes in padded 64-slot runs, one per group-step (LiArgs::fixGroup).  fxg_compact() decides from the run alone who
// serves it: this kernel when the 64 points lie within a quarter of the probed radius of the first one, li_fixup_kernel (a wave
// per lookup) when they are scattered relative to it -- few samples per pixel inside a dense beam.  Both kernels evaluate the
// same function; neither writes anything for the other.
//
// Bucket: the INDICES of the photons within (radius + spread) of the run's first point, <= FXG_CAP of them in LDS (16 KB).
// Every lane looks at the same photon at the same time, so its position is a scalar load (eight in flight, FXG_FETCH8).
// Selection per lane (kdtree.h:157-206 keeps the nused nearest and shrinks maxDistSquared to the farthest of them):
//   A  histogram of DistanceSquared over the bucket: 64 bins on [0, T_lane), 8-bit counters packed four to an LDS word per
//      lane (ds_add_u32); a byte that wrapped shows as sum != the exact in-range count and sends the lane to the exact pass
//   B  the bin that holds the k-th: histogram again over 64 sub-bins of that bin
//   C  the sub-bin that holds it: its members (<= 8, else the exact pass) are collected and ordered in registers: the k-th
//      DistanceSquared itself, bit for bit, and how many photons AT that value still belong (ties, bucket order)
//   D  flux: sum of alpha over the members, one 128-B row per photon through the scalar cache, a 0/1 factor per lane
// Bins are floor(d2 * 64 / T) and floor(frac * 64): monotone in d2, so every pass narrows the same order statistics; what
// decides membership in the end is the comparison with the k-th value found in C, never a bin.
//
// Radii.  The first radius^2 of a run comes from a density probe (photon counts of the grid cells around the run's first
// point, read from the cell-start prefix sums), not from a neighbour's result: runs arrive from all over the frame.  A bucket
// that overflows makes the cluster's balls smaller; a lane whose ball held fewer than nused photons grows it by what its count
// says (like lphoton()) and joins a later bucket; a lane that has seen both is bisected between them.  After FXG_TRIES
// corrections, or when that bracket closes below 25 % (a point beside a beam: its own ball nearly empty, the cluster's already in
// the core), the wave-cooperative lphoton() serves the lookup -- the slow path, exact as well.
// Instrumented builds: -DPVOL_FXG_TIME (cycle split into the stats counters), -DPVOL_FXG_DEBUG (per-run prints).
#ifndef PVOL_FIXGRP_DEV_H
#define PVOL_FIXGRP_DEV_H

#ifndef FXG_CORE
#define FXG_CORE 1   // photons that are members for every lookup of a cluster are summed once for the wave
#endif
#define FXG_CAP 4080   // bucket slots: photon indices only (4 B each), eight times nused 500; (4080 + 8) x 4 B + the 4 KB that paint list and
                       // histogram share = 20 448 B: eight waves per CU (two per SIMD, what 255 VGPRs allow)
#define FXG_MINI 8
#define FXG_TRIES 16  // radius corrections per lookup before the exact pass takes it

// photons in the grid cells that overlap the cube c +- h, and the volume of those cells
__device__ __forceinline__ uint32_t box_count(const GridView &g, V3 c, float h, int lane, float *vol) {
    const float inv = g.invCell;
    int x0 = (int)floorf((c.x - h - g.gridLo[0]) * inv), x1 = (int)floorf((c.x + h - g.gridLo[0]) * inv);
    int y0 = (int)floorf((c.y - h - g.gridLo[1]) * inv), y1 = (int)floorf((c.y + h - g.gridLo[1]) * inv);
    int z0 = (int)floorf((c.z - h - g.gridLo[2]) * inv), z1 = (int)floorf((c.z + h - g.gridLo[2]) * inv);
    // volume of the cells counted; where the cube sticks out of the photons' bounds those cells are empty, and counted as such
    *vol = (float)(x1 - x0 + 1) * (float)(y1 - y0 + 1) * (float)(z1 - z0 + 1) * (g.cellSize * g.cellSize * g.cellSize);
    x0 = max(x0, 0); y0 = max(y0, 0); z0 = max(z0, 0);
    x1 = min(x1, g.gdim[0] - 1); y1 = min(y1, g.gdim[1] - 1); z1 = min(z1, g.gdim[2] - 1);
    if (x0 > x1 || y0 > y1 || z0 > z1) return 0u;
    const int ny = y1 - y0 + 1, nz = z1 - z0 + 1, rows = ny * nz;
    uint32_t tot = 0u;
    for (int rb = 0; rb < rows; rb += LANES) {
        const int r = rb + lane;
        if (r < rows) {
            const int iz = r / ny, iy = r - iz * ny;
            const size_t base = ((size_t)(z0 + iz) * g.gdim[1] + (y0 + iy)) * g.gdim[0];
            tot += g.cellStart[base + x1 + 1] - g.cellStart[base + x0];
        }
    }
    for (int off = 32; off > 0; off >>= 1) tot += (uint32_t)__shfl_xor((int)tot, off);
    return tot;
}

// Radius^2 of the ball around c expected to hold ~1.4 k photons, from the photon counts of the grid cells around it (cube of
// half side sqrt(T), at most four refinements of up to 20x the volume each).  Only a first guess: the lookups correct it.
__device__ float probe_radius(const GridView &g, V3 c, float T0, int k, float maxT, int lane, float aim = 1.4f) {
    float T = fminf(T0, maxT);
    for (int it = 0; it < 4; ++it) {
        float vol;
        const uint32_t cnt = box_count(g, c, sqrtf(T), lane, &vol);
        const float pred = (float)cnt * (4.18879020478639f * T * sqrtf(T)) / vol;   // photons in the ball at the cube's density
        const float want = aim * (float)k;
        if (pred >= fmaxf(1.05f, aim - 0.25f) * (float)k && pred <= (aim + 0.6f) * (float)k) break;
        if (pred < (float)k && T >= maxT) break;
        const float ratio = want / fmaxf(pred, 0.05f * want);
        T = fminf(maxT, T * __builtin_amdgcn_exp2f(0.6666667f * __builtin_amdgcn_logf(ratio)));
    }
    return T;
}

// Which of the two hand-over kernels serves a 64-slot run -- decided from the run alone, so both kernels agree without talking:
// COMPACT (the lookups lie within a quarter of the probed radius of the first one: one staged bucket serves them all, lane per
// lookup, li_fixup_group_kernel) or SCATTERED (few samples per pixel in a dense beam: every lookup has its own ball, and a wave
// per lookup with 64 lanes on one candidate set is the better shape, li_fixup_kernel).  *Tprobe: first radius^2 for either.
__device__ __forceinline__ bool fxg_compact(const GridView &g, const DevScene &S, bool valid, V3 p, int lane, float *Tprobe, float aim = 1.4f) {
    const unsigned long long m = __ballot(valid);
    *Tprobe = 0.f;
    if (!m) return false;
    const int piv = __ffsll((long long)m) - 1;
    const V3 c = v3(lane_f(p.x, piv), lane_f(p.y, piv), lane_f(p.z, piv));
    const float T = probe_radius(g, c, S.rkEstimate, S.nUsed, S.maxDistSq, lane, aim);
    *Tprobe = T;
    const float spread = wave_max(valid ? len(p - c) : 0.f);
    return spread <= 0.25f * sqrtf(T);
}

__device__ __forceinline__ void fxg_clear(uint32_t *hist, int lane) {
#pragma unroll
    for (int w = 0; w < 16; ++w) hist[w * LANES + lane] = 0u;
}

// first bin whose running count reaches `want` (>= 1); returns false if the 64 bins hold fewer
__device__ __forceinline__ bool fxg_scan(const uint32_t *hist, int lane, uint32_t want, uint32_t *bin, uint32_t *below, uint32_t *inBin, uint32_t *total) {
    uint32_t cum = 0u, b = 0u, bel = 0u, cnt = 0u;
    bool found = false;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const uint32_t word = hist[w * LANES + lane];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint32_t c = (word >> (8 * s)) & 255u;
            if (!found && cum + c >= want) { found = true; b = (uint32_t)(4 * w + s); bel = cum; cnt = c; }
            cum += c;
        }
    }
    *bin = b; *below = bel; *inBin = cnt; *total = cum;
    return found;
}

template <bool SPECTRAL>
__global__ __launch_bounds__(LANES, 2) void li_fixup_group_kernel(LiArgs A) {
    extern __shared__ __align__(16) unsigned char lds[];
    const DevScene &S = *A.scene;
    const int lane = threadIdx.x;
    if (*A.needSeq != 0u) return;   // the whole batch is redone sequentially
    const uint32_t n = min(*A.deferCount, A.deferCap);
    float *bucket = reinterpret_cast<float *>(lds);
    const uint32_t *bIdx = reinterpret_cast<const uint32_t *>(lds);          // FXG_CAP + 8 photon indices
    uint32_t *paint = reinterpret_cast<uint32_t *>(lds) + (FXG_CAP + 8);
    uint32_t *hist = paint;   // [16 words][64 lanes]: the staging's paint list and the selection's histogram are never live together (PAINT_CAP = 16 x 64)
    Gather G;                             // the slow path's candidate lists alias the bucket (never live together)
    G.cap = S.candCap; G.cd = bucket; G.ci = reinterpret_cast<uint32_t *>(bucket + S.candCap); G.paint = paint;
    const GridView gv = volume_grid(S);
    const int k = S.nUsed;
    const float wIso = 1.f / (4.f * K_PI);
    const float widen = A.fxgWiden > 0.f ? A.fxgWiden : 1.3f, aim = A.fxgAim > 0.f ? A.fxgAim : 1.4f;
    const int q = lane & 7;
    const f4 sigA4 = ld4(S.sigA, q), sigS4 = ld4(S.sigS, q);
    const f4 sigT4 = sigA4 + sigS4;
    const f4 albedo4 = clean4(fdiv4(sigS4, sigT4), q);
    const f4 X4 = ld4(S.cieX, q), Y4 = ld4(S.cieY, q), Z4 = ld4(S.cieZ, q);
    WaveCounters wc = {};
    typedef const __attribute__((address_space(4))) nf4 cf4;
    cf4 *posRows = (cf4 *)gv.pos4;   // positions come through the scalar cache: every lane looks at the same photon
#ifdef PVOL_FXG_TIME
    unsigned long long cyProbe = 0, cyStage = 0, cySel = 0, cyFlux = 0, cySlow = 0, nRuns = 0, nSkipped = 0;
    const unsigned long long tK0 = stamp();
#define FXG_T(var) var += stamp() - tS; tS = stamp();
#else
#define FXG_T(var)
#endif
    unsigned long long nStagings = 0, nWrap = 0, nCrowded = 0, nTries = 0;   // diag[2..5]
    // the eight photons bucket[i .. i+8): indices out of LDS (one address for the wave), positions as scalar loads
#define FXG_FETCH8(i)                                                                                                         \
    uint32_t id_[8];                                                                                                          \
    nf4 P_[8];                                                                                                                \
    {                                                                                                                         \
        const uint4 ia = *reinterpret_cast<const uint4 *>(bIdx + (i)), ib = *reinterpret_cast<const uint4 *>(bIdx + (i) + 4); \
        const uint32_t iv[8] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w};                                              \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) id_[u] = (uint32_t)__builtin_amdgcn_readfirstlane((int)iv[u]);         \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) P_[u] = posRows[id_[u]];                                                \
    }
    // eight consecutive runs per wave at a time (neighbouring runs are the same rays one march step apart: their buckets overlap)
    const uint32_t nRuns = (n + (uint32_t)LANES - 1u) / (uint32_t)LANES;
    for (uint32_t runI = blockIdx.x * 8u; runI < nRuns; runI = ((runI & 7u) == 7u) ? runI + 1u + (gridDim.x - 1u) * 8u : runI + 1u) {
        const uint32_t r0 = runI * (uint32_t)LANES;
        const uint32_t e = r0 + (uint32_t)lane;
        DeferRec r;
        r.ray = 0xffffffffu; r.px = r.py = r.pz = 0.f; r.kRem = 0.f; r.stepD = 0.f; r.guess = 0.f; r.dens = 1.f;
        if (e < n) r = A.defer[e];
        const bool valid = r.ray != 0xffffffffu;
        const V3 p = v3(r.px, r.py, r.pz);
        float acc[32];
#pragma unroll
        for (int b = 0; b < 32; ++b) acc[b] = 0.f;
        float distSq = 0.f;       // what the density estimate divides by: the k-th distance^2, or the farthest of a short set
        int nFound = 0;
        bool served = false;      // acc / distSq / nFound hold this lane's lookup
        float Twant = 0.f;
        float Tlo = 0.f, Thi = INFINITY;   // largest radius^2 seen to hold fewer than nused photons / smallest whose bucket overflowed
        int tries = 0;
        float reach = 0.5f;       // a cluster takes the waiting lookups within this fraction of the pivot's radius of the pivot
        unsigned long long pending = __ballot(valid), slow = 0ull;
        float Trun;
#ifdef PVOL_FXG_TIME
        unsigned long long tS = stamp();
        if (!fxg_compact(gv, S, valid, p, lane, &Trun, aim)) { FXG_T(cyProbe) ++nSkipped; continue; }
        FXG_T(cyProbe) ++nRuns;
#else
        if (!fxg_compact(gv, S, valid, p, lane, &Trun, aim)) continue;   // empty, or scattered: li_fixup_kernel's
#endif
        // the probe aims at 1.4 nused photons; the bucket holds eight times nused, and a ball that turns out too small costs a
        // second staging: start at about twice the probe's volume
        if (valid) Twant = fminf(S.maxDistSq, Trun * widen);
        int reprobes = 0;
        while (pending) {
            const bool waiting = ((pending >> lane) & 1ull) != 0ull;
            const int piv = __ffsll((long long)pending) - 1;
            const V3 c = v3(lane_f(p.x, piv), lane_f(p.y, piv), lane_f(p.z, piv));
            // ---- cluster: the waiting lookups within half a radius of the pivot (all of a compact run, the first time)
            const float Tp = lane_f(Twant, piv);
            const float dPiv = len(p - c);
            const bool in = waiting && dPiv <= lane_f(reach, piv) * sqrtf(Tp);
            const float Tmax = wave_max(in ? Twant : 0.f);
            const float spread = wave_max(in ? dPiv : 0.f);
            const float Rs = (sqrtf(Tmax) + spread) * 1.0001f + 1e-6f;
            unsigned long long tst = 0;
            __syncthreads();
            const int Mb = stage_bucket_g<FXG_CAP, true, true>(gv, G, bucket, c, Rs, lane, tst);
            ++nStagings;
            FXG_T(cyStage)
            if (Mb < 0) {
                // more photons around the cluster than the bucket holds: smaller balls for all of it -- between the largest radius
                // seen to hold too few and this one for a lane that has been there, else half the volume
                // (a run whose first point lies beside the beam the others are in was probed at the wrong place: probe again here)
                float Tprobe2 = INFINITY;
                if (reprobes < 3) { ++reprobes; Tprobe2 = widen * probe_radius(gv, c, Tp * 0.5f, k, S.maxDistSq, lane, aim); }
                bool toSlowO = false;
                if (in) {
                    ++tries;
                    Thi = fminf(Thi, Twant);
                    Twant = Tlo > 0.f ? sqrtf(Tlo * Thi) : fminf(Twant * 0.62996f, Tprobe2);
                    // beside a beam the lane's own ball holds few photons while the cluster's (radius + spread) already reaches the
                    // core: no radius serves both.  Such lookups come in neighbourhoods: they wait for a TIGHTER cluster (their
                    // bracket's upper end was learnt with the wide one and is forgotten); only when even a cluster of a sixteenth of
                    // the radius overflows does the wave-per-lookup pass look at the lane's ball alone
                    if (Tlo > 0.f && Thi < 1.25f * Tlo) {
                        if (reach > 0.07f) { reach *= 0.35f; Thi = INFINITY; }   // Twant: the bracket's middle, as set above
                        else toSlowO = true;
                    }
                    if (tries >= FXG_TRIES) toSlowO = true;
                }
#ifdef PVOL_FXG_DEBUG
                { const int trp = lane_i(tries, piv); const float tl = lane_f(Tlo, piv), th = lane_f(Thi, piv), tw = lane_f(Twant, piv);
                    if (lane == 0 && trp >= 8 && atomicAdd(&A.counters->pad, 1ull) < 80ull) printf("fxg: OVERFLOW blk %d run %u piv %d Tp %g -> %g Tlo %g Thi %g tries %d in %016llx spread %g probe2 %g\n", (int)blockIdx.x, r0, piv, (double)Tp, (double)tw, (double)tl, (double)th, trp, __ballot(in), (double)spread, (double)Tprobe2); }
#endif
                const unsigned long long ts = __ballot(toSlowO);
                nTries += (unsigned long long)__popcll(ts);
                slow |= ts; pending &= ~ts;
                continue;
            }
            // ---- A: in-range count and 64-bin histogram of this lane's DistanceSquared values
            const float Tl = Twant;
            const float scale = in ? 64.f / Tl : 0.f;
            fxg_clear(hist, lane);
            __syncthreads();
            int nIn = 0;
            float maxIn = 0.f;
            for (int i = 0; i < Mb; i += 8) {
                FXG_FETCH8(i)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float dx = P_[u].x - p.x, dy = P_[u].y - p.y, dz = P_[u].z - p.z;
                    const float d2 = dx * dx + dy * dy + dz * dz;   // DistanceSquared(photon.p, p), kdtree.h:180
                    const bool inside = in && (i + u < Mb) && d2 < Tl;
                    nIn += inside ? 1 : 0;
                    maxIn = inside ? fmaxf(maxIn, d2) : maxIn;
                    const uint32_t bin = (uint32_t)fminf(d2 * scale, 63.f);
                    atomicAdd(&hist[(bin >> 2) * LANES + lane], inside ? 1u << ((bin & 3u) << 3) : 0u);
                }
            }
            __syncthreads();
            bool ok = false, toSlow = false;
            float kth = INFINITY;     // members: d2 < kth, plus tieQuota photons at d2 == kth
            int tieQuota = 0;
            uint32_t bstar = 0u, below = 0u, inBin = 0u, total = 0u;
            const bool full = in && nIn >= k;
            if (in && !full && Tl >= S.maxDistSq) ok = true;   // fewer than nused within maxdist: all of them (photonvolume.cpp:83 decides on the count)
            if (__ballot(full)) {
                const bool f1 = fxg_scan(hist, lane, (uint32_t)k, &bstar, &below, &inBin, &total);
                if (full && (!f1 || total != (uint32_t)nIn)) toSlow = true;   // a byte wrapped
                nWrap += (unsigned long long)__popcll(__ballot(toSlow));
                // ---- B: 64 sub-bins of the bin that holds the k-th
                const bool selB = full && !toSlow;
                const float fb = (float)bstar;
                fxg_clear(hist, lane);
                __syncthreads();
                for (int i = 0; i < Mb; i += 8) {
                    FXG_FETCH8(i)
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float dx = P_[u].x - p.x, dy = P_[u].y - p.y, dz = P_[u].z - p.z;
                        const float d2 = dx * dx + dy * dy + dz * dz;
                        const float f = d2 * scale;
                        const bool hitB = selB && (i + u < Mb) && d2 < Tl && (uint32_t)fminf(f, 63.f) == bstar;
                        const uint32_t sub = (uint32_t)fminf(fmaxf((f - fb) * 64.f, 0.f), 63.f);
                        atomicAdd(&hist[(sub >> 2) * LANES + lane], hitB ? 1u << ((sub & 3u) << 3) : 0u);
                    }
                }
                __syncthreads();
                uint32_t sstar = 0u, below2 = 0u, inSub = 0u, total2 = 0u;
                const uint32_t need1 = (uint32_t)k - below;   // rank of the k-th inside its bin, >= 1
                const bool f2 = fxg_scan(hist, lane, need1, &sstar, &below2, &inSub, &total2);
                if (selB && (!f2 || total2 != inBin || inSub > (uint32_t)FXG_MINI)) toSlow = true;
                nCrowded += (unsigned long long)__popcll(__ballot(selB && toSlow));
                // ---- C: the members of that sub-bin, ordered
                const bool selC = selB && !toSlow;
                float mini[FXG_MINI];
#pragma unroll
                for (int m = 0; m < FXG_MINI; ++m) mini[m] = INFINITY;
                for (int i = 0; i < Mb; i += 8) {
                    FXG_FETCH8(i)
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float dx = P_[u].x - p.x, dy = P_[u].y - p.y, dz = P_[u].z - p.z;
                        const float d2 = dx * dx + dy * dy + dz * dz;
                        const float f = d2 * scale;
                        const bool hitC = selC && (i + u < Mb) && d2 < Tl && (uint32_t)fminf(f, 63.f) == bstar &&
                                          (uint32_t)fminf(fmaxf((f - fb) * 64.f, 0.f), 63.f) == sstar;
                        if (__ballot(hitC)) {
                            float v = hitC ? d2 : INFINITY;   // insertion into the ascending list
#pragma unroll
                            for (int m = 0; m < FXG_MINI; ++m) { const float lo = fminf(mini[m], v); v = fmaxf(mini[m], v); mini[m] = lo; }
                        }
                    }
                }
                if (selC) {
                    const uint32_t need2 = need1 - below2;   // 1 .. inSub
                    float kv = mini[0];
#pragma unroll
                    for (int m = 1; m < FXG_MINI; ++m) kv = (need2 == (uint32_t)(m + 1)) ? mini[m] : kv;
                    int lessInList = 0;
#pragma unroll
                    for (int m = 0; m < FXG_MINI; ++m) lessInList += mini[m] < kv ? 1 : 0;
                    kth = kv;
                    tieQuota = k - (int)(below + below2) - lessInList;
                    ok = true;
                }
            }
            FXG_T(cySel)
            // ---- D: flux of the members
            const bool doFlux = ok && (full || nIn >= 10);
            if (__ballot(doFlux)) {
                int tq = tieQuota;
                // photons that are members for EVERY lookup of the cluster are summed once for the wave after the scan (their
                // indices wait in the histogram words, free by now): the 64 balls of a compact run overlap in most of their volume
                const unsigned long long fluxLanes = __ballot(doFlux);
                uint32_t *clist = hist;
                int nCore = 0;
                for (int i = 0; i < Mb; i += 8) {
                    FXG_FETCH8(i)
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float dx = P_[u].x - p.x, dy = P_[u].y - p.y, dz = P_[u].z - p.z;
                        const float d2 = dx * dx + dy * dy + dz * dz;
                        bool mem = doFlux && (i + u < Mb) && d2 < Tl && d2 < kth;
                        if (doFlux && (i + u < Mb) && d2 == kth && tq > 0) { mem = true; --tq; }
                        const unsigned long long mm = __ballot(mem);
                        if (!mm) continue;
                        if (FXG_CORE && mm == fluxLanes && nCore < 16 * LANES) {
                            if (lane == 0) clist[nCore] = id_[u];
                            ++nCore;
                            continue;
                        }
                        const float wgt = mem ? wIso : 0.f;
                        cf4 *row = (cf4 *)(S.alpha4 + (size_t)id_[u] * 8);
#pragma unroll
                        for (int qq = 0; qq < 8; ++qq) {
                            const nf4 rr = row[qq];
                            acc[4 * qq] = __builtin_fmaf(rr.x, wgt, acc[4 * qq]); acc[4 * qq + 1] = __builtin_fmaf(rr.y, wgt, acc[4 * qq + 1]);
                            acc[4 * qq + 2] = __builtin_fmaf(rr.z, wgt, acc[4 * qq + 2]); acc[4 * qq + 3] = __builtin_fmaf(rr.w, wgt, acc[4 * qq + 3]);
                        }
                    }
                }
                if (FXG_CORE && nCore > 0) {
                    __syncthreads();
                    // eight 128-B rows per load: lane l reads the (l & 7)-th float4 of the row of list entry 8 q + (l >> 3)
                    const int sub = lane >> 3, quart = lane & 7;
                    nf4 cs4 = {0.f, 0.f, 0.f, 0.f};
                    for (int c0 = 0; c0 < nCore; c0 += 32) {
                        nf4 v[4];
#pragma unroll
                        for (int qv = 0; qv < 4; ++qv) {
                            const int at = c0 + 8 * qv + sub;
                            const bool on = at < nCore;
                            const uint32_t idx = clist[on ? at : 0];
                            const float4 w = S.alpha4[(size_t)idx * 8 + quart];
                            v[qv] = on ? nf4{w.x, w.y, w.z, w.w} : nf4{0.f, 0.f, 0.f, 0.f};
                        }
                        cs4 += (v[0] + v[1]) + (v[2] + v[3]);
                    }
                    float csv[4] = {cs4.x, cs4.y, cs4.z, cs4.w};
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {   // sum over the eight row groups (lane bits 3, 4, 5)
                        float x = csv[cc];
                        x += dppf<DPP_ROW_ROR8>(x);
                        { auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false); x = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
                        { auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false); x = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
                        csv[cc] = x;
                    }
                    const float wCore = doFlux ? wIso : 0.f;
#pragma unroll
                    for (int b = 0; b < 30; ++b)
                        acc[b] = __builtin_fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(csv[b & 3]), b >> 2)), wCore, acc[b]);
                    __syncthreads();
                }
            }
            FXG_T(cyFlux)
            if (ok) {
                served = true;
                nFound = full ? k : nIn;
                distSq = full ? kth : maxIn;
            }
            // ---- lanes of the cluster that are not done: a larger ball next time, or the exact pass
            const bool failed = in && !ok && !toSlow;   // fewer than nused inside a ball smaller than maxdist
            if (failed) {
                ++tries;
                Tlo = Tl;
                const float grow = nIn > 0 ? 1.35f * __builtin_amdgcn_exp2f(0.6666667f * __builtin_amdgcn_logf((float)k / (float)nIn)) : 8.f;
                const float Tg = fminf(S.maxDistSq, Tl * fmaxf(1.5f, grow));
                Twant = Tg < Thi ? Tg : sqrtf(Tlo * Thi);   // never back into a ball whose bucket overflowed
                if (Thi < 1.25f * Tlo) {   // (as after an overflow: a tighter cluster first)
                    if (reach > 0.07f) { reach *= 0.35f; Thi = INFINITY; Twant = Tg; }
                    else toSlow = true;
                }
                if (tries >= FXG_TRIES) toSlow = true;
            }
#ifdef PVOL_FXG_DEBUG
            { const unsigned long long fl = __ballot(failed && tries >= 8); if (fl) { const int j = __ffsll((long long)fl) - 1;
                const float tl = lane_f(Tl, j), tw = lane_f(Twant, j), th = lane_f(Thi, j); const int ni = lane_i(nIn, j), trj = lane_i(tries, j);
                if (lane == 0 && atomicAdd(&A.counters->pad, 1ull) < 80ull) printf("fxg: FAILED blk %d run %u lane %d Tl %g nIn %d -> Twant %g Thi %g tries %d Mb %d\n", (int)blockIdx.x, r0, j, (double)tl, ni, (double)tw, (double)th, trj, Mb); } }
#endif
            nTries += (unsigned long long)__popcll(__ballot(failed && toSlow));
            slow |= __ballot(toSlow);
            pending &= ~__ballot(ok || toSlow);
        }
        { const unsigned long long sv = __ballot(served); if (lane == 0 && sv) atomicAdd(&A.counters->diag[1], (unsigned long long)__popcll(sv)); }   // lookups served from shared buckets
        // ---- terms of the lanes served above:  exp(-sigma_t R_j) sigma_s step albedo L_ii,  L_ii = sum(alpha) phase / (4/3 pi r^3 sigma_s(p))
        if (served && nFound >= 10) {
            const float dV = distSq * sqrtf(distSq);
            float liiScale = 0.f;
            if (dV != 0.f && r.dens != 0.f) liiScale = __builtin_amdgcn_rcpf(float(4.0 / 3.0 * (double)K_PI * (double)dV)) * __builtin_amdgcn_rcpf(r.dens);
            float x = 0.f, y = 0.f, z = 0.f;
            float *ops = SPECTRAL ? A.out + (size_t)r.ray * 60 : 0;
#pragma unroll
            for (int b = 0; b < 30; ++b) {
                const float sA = S.sigA[b], sS = S.sigS[b];
                const float sT = sA + sS;
                const float alb = __fdividef(sS, sT);
                const float Lii = sS != 0.f ? acc[b] * liiScale * __builtin_amdgcn_rcpf(sS) : 0.f;
                const float t = __builtin_amdgcn_exp2f(sT * r.kRem) * (sS * (alb * Lii) * r.stepD);
                const float tc = (t == t) ? t : 0.f;
                if (SPECTRAL) { if (tc != 0.f) atomicAdd(ops + b, tc); }
                else { x += S.cieX[b] * tc; y += S.cieY[b] * tc; z += S.cieZ[b] * tc; }
            }
            if (!SPECTRAL) {
                const float scale = float(700 - 400) / float(106.856895f * 30);
                float *op = A.out + (size_t)r.ray * 4;
                atomicAdd(op, x * scale); atomicAdd(op + 1, y * scale); atomicAdd(op + 2, z * scale);
            }
        }
        // ---- the exact pass for what is left (one wave-cooperative lookup each)
        FXG_T(cySel)
        while (slow) {
            const int j = __ffsll((long long)slow) - 1;
            slow &= slow - 1ull;
            const V3 pj = v3(lane_f(p.x, j), lane_f(p.y, j), lane_f(p.z, j));
            const float dens = lane_f(r.dens, j), kRem = lane_f(r.kRem, j), stepD = lane_f(r.stepD, j);
            const uint32_t ray = (uint32_t)lane_i((int)r.ray, j);
            float rk;
            __syncthreads();
            const float Tj = lane_f(Twant, j);   // what the lane last asked for (lphoton widens its first radius by 1.3 and grows it when it fails)
            const f4 Lii = lphoton<false, 12, true>(S, G, v3(0.f, 0.f, 1.f), pj, sigS4 * dens, lane, wc, Tj < S.maxDistSq ? Tj : 0.f, &rk);
            const f4 kk = sigT4 * kRem;
            f4 c = make_float4(__builtin_amdgcn_exp2f(kk.x), __builtin_amdgcn_exp2f(kk.y), __builtin_amdgcn_exp2f(kk.z), __builtin_amdgcn_exp2f(kk.w));
            c = clean4(c * (sigS4 * (albedo4 * Lii) * stepD), q);
            if (SPECTRAL) {
                if (lane < 8) {
                    float *op = A.out + (size_t)ray * 60 + 4 * q;
                    const float cv[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) if (4 * q + cc < 30 && cv[cc] != 0.f) atomicAdd(op + cc, cv[cc]);
                }
            } else {
                const float scale = float(700 - 400) / float(106.856895f * 30);
                const float x = group8_sum(X4.x * c.x + X4.y * c.y + X4.z * c.z + X4.w * c.w) * scale;
                const float y = group8_sum(Y4.x * c.x + Y4.y * c.y + Y4.z * c.z + Y4.w * c.w) * scale;
                const float z = group8_sum(Z4.x * c.x + Z4.y * c.y + Z4.z * c.z + Z4.w * c.w) * scale;
                if (lane == 0) {
                    float *op = A.out + (size_t)ray * 4;
                    atomicAdd(op, x); atomicAdd(op + 1, y); atomicAdd(op + 2, z);
                }
            }
            if (lane == 0) atomicAdd(&A.counters->diag[0], 1ull);   // exact-pass lookups (reported with the stats)
        }
        FXG_T(cySlow)
    }
#undef FXG_FETCH8
#ifdef PVOL_FXG_TIME
    if (lane == 0) {
        atomicAdd(&A.counters->cySearch, cyStage); atomicAdd(&A.counters->cySelect, cySel); atomicAdd(&A.counters->cyFlux, cyFlux);
        atomicAdd(&A.counters->cyTotal, stamp() - tK0); atomicAdd(&A.counters->nTested, cyProbe); atomicAdd(&A.counters->nKept, cySlow);
        atomicAdd(&A.counters->nLookupsLt10, nRuns); atomicAdd(&A.counters->nShadowUnoccluded, nSkipped);
    }
#endif
    if (lane == 0) {
        if (nStagings) atomicAdd(&A.counters->diag[2], nStagings);
        if (nWrap) atomicAdd(&A.counters->diag[3], nWrap);
        if (nCrowded) atomicAdd(&A.counters->diag[4], nCrowded);
        if (nTries) atomicAdd(&A.counters->diag[5], nTries);
    }
}

extern "C" size_t pvol_fixgrp_lds_bytes(int candCap) {
    const size_t bucket = (size_t)(FXG_CAP + 8) * 4;
    static_assert(PAINT_CAP == 16 * LANES, "paint list and histogram share one region");
    return std::max(bucket, (size_t)candCap * 8) + PAINT_CAP * 4;
}
#endif
