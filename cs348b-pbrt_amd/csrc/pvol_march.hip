// pvol_march.hip -- the hot kernels: PhotonVolumeIntegrator::Li (integrators/photonvolume.cpp:112-222)
// with the radius-bounded k-NN photon gather LPhoton (:65-108), hand-written for gfx950.
//
// Two launch shapes share one per-ray march (`march_ray`):
//   li_seq_kernel  one 64-lane wavefront == one MT19937 stream == one render tile
//                  (renderers/samplerrenderer.cpp:73).  The wave walks the tile's rays in order with the
//                  MT state in LDS, so every RandomUInt() is the reference's.  Needed whenever drawn
//                  VALUES reach the result: more than one light (lightNum[] picks the light per step),
//                  a VolumeGrid (tau() offsets), or a step dark enough for Russian roulette.
//   li_par_kernel  one wavefront per camera ray, rays handed out in chunks from a global counter.  Legal
//                  when no drawn value can reach the result (one light, analytic tau(), no roulette):
//                  Li() then only COUNTS draws (4+6n+n+u), so rays are independent and each stream's
//                  end position is a sum.  A ray that would roll the roulette raises a flag and the
//                  batch is redone by li_seq_kernel -- never guessed.
// Per lookup (the gather):
//   * rows of grid cells that can hold a photon nearer than sqrt(T) are enumerated one per lane, their
//     contiguous photon ranges concatenated by a wave scan; the 64 lanes stride the concatenation with
//     coalesced float4 loads, NCH chunks in flight.  T starts from the k-th distance^2 of the previous
//     ray at the same march step (x1.5): exact whenever >= k photons are found inside it, else redone
//     with maxDist^2.
//   * accepted candidates go to an LDS list by ballot compaction; k-selection is a bit-pattern bisection
//     held in registers (ballot + popcount), ties kept in list order.
//   * flux: 8 photons per pass, one 128-B alpha row per 8-lane group, 3 shuffle-adds to finish.
//   * spectra live as "8 lanes x float4".  No MFMA: nothing on this path is a dense contraction.
// Built with -ffp-contract=off: every geometric decision (step counts, slab clips, cell indices, shadow
// hits) is taken on the fp32 values the CPU reference computes.
#include "pvol_math.h"

#define NCH 4        // chunks of 64 candidate loads in flight
#define PREV_N 256   // march steps for which the previous ray's k-th distance^2 is remembered
#define CHUNK_RAYS 64
#ifndef PVOL_GUESS_SCALE
#define PVOL_GUESS_SCALE 1.3f   // first search radius^2 = this x the neighbouring k-th distance^2
#endif
#define PAINT_ROW 16   // rows up to this many photons go through the painted index list
#define PAINT_CAP (64 * PAINT_ROW)
#ifndef PVOL_WPE
#ifndef PVOL_WPE_BIG
#define PVOL_WPE_BIG 2   // k > 64 instantiations (NREG 12): 2 waves/SIMD measured best on a C3-like load (0.129 vs 0.124 at 1, 0.079 Msamples/s at 3)
#endif
#define PVOL_WPE 3   // minimum waves per SIMD the register allocator must leave room for (3 measured best: profiles/)
#endif

#include "pvol_rng_dev.h"
#include "pvol_gridrows_dev.h"

// ------------------------------------------------------------------------------------------ k-NN gather
struct Gather {
    float *cd;      // LDS: candidate dist^2
    uint32_t *ci;   // LDS: candidate photon index (sorted order)
    uint32_t *paint; // LDS: PAINT_CAP photon indices of the short rows, in concatenated order
    int cap;
};
// per-wave work counters (stats build), flushed with one atomic each when the wave retires
struct WaveCounters { unsigned long long tested, kept, lt10, retries, cySearch, cySelect, cyFlux, rays, steps, unocc, diag0, diag1, diag2, diag3, diag4, diag5; };
__device__ __forceinline__ unsigned long long stamp() { return __builtin_amdgcn_s_memtime(); }

// Keep the k nearest of the M > k candidates of the LDS list; returns the k-th smallest dist^2 (the new
// radius^2).  The list is pulled into NREG registers per lane once; the bisection over the fp32 bit
// pattern (monotone for non-negative floats) then runs on ballots and popcounts only.  Ties at the k-th
// value are resolved in list order.
template <int NREG>
__device__ float select_k(Gather &G, int M, int k, float T, int lane) {
    uint32_t bits[NREG], id[NREG];
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
        int i = r * LANES + lane;
        bool on = i < M;
        bits[r] = on ? __float_as_uint(G.cd[i]) : 0x7f800000u;
        id[r] = on ? G.ci[i] : 0u;
    }
    uint32_t lo = 0u, hi = __float_as_uint(T);   // every candidate is < T
    {   // the k-th of M points spread in a ball sits near T*(k/M)^(2/3): T/16 is almost always below it
        uint32_t probe = __float_as_uint(T * 0.0625f);
        int c = 0;
#pragma unroll
        for (int r = 0; r < NREG; ++r) c += __popcll(__ballot(bits[r] <= probe));
        if (c < k) lo = probe + 1; else hi = probe;
    }
    uint32_t tau;
    for (;;) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        int c = 0;
#pragma unroll
        for (int r = 0; r < NREG; ++r) c += __popcll(__ballot(bits[r] <= mid));
        if (c == k) { tau = mid; break; }
        if (c < k) lo = mid + 1; else hi = mid;
        if (lo >= hi) { tau = lo; break; }
    }
    int nLess = 0;
#pragma unroll
    for (int r = 0; r < NREG; ++r) nLess += __popcll(__ballot(bits[r] < tau));
    int quota = k - nLess;
    int out = 0;
    uint32_t keptMax = 0u;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
        bool less = bits[r] < tau;
        bool eq = bits[r] == tau;
        uint64_t me = __ballot(eq);
        bool keep = less || (eq && (int)lanes_below(me, lane) < quota);
        quota -= __popcll(me);
        uint64_t mk = __ballot(keep);
        if (keep) {
            int pos = out + (int)lanes_below(mk, lane);
            G.cd[pos] = __uint_as_float(bits[r]);
            G.ci[pos] = id[r];
            keptMax = max(keptMax, bits[r]);
        }
        out += __popcll(mk);
    }
    __syncthreads();
    return wave_max(__uint_as_float(keptMax));
}

// PhotonVolumeIntegrator::LPhoton (photonvolume.cpp:65-108): k nearest photons within maxDist of pt;
// returns total flux / (4/3 pi r^3 sigma_s) in the float4 layout, or 0 when fewer than 10 are found.
// `guess` (> 0) is a previous k-th distance^2 at a nearby point; `rkOut` returns this lookup's
// (0 when fewer than k photons lie within maxDist).
template <bool STATS, int NREG, bool FINE = false>
__device__ f4 lphoton(const DevScene &S, Gather &G, V3 w, V3 pt, f4 sigS_at_p, int lane, WaveCounters &wc, float guess, float *rkOut) {
    const int q = lane & 7;
    f4 zero = mk4(0.f);
    *rkOut = 0.f;
    if (S.nPhotons == 0u) return zero;
    const GridView gv = volume_grid(S);
    const int k = S.nUsed;
    unsigned long long tested = 0;
    float T = S.maxDistSq;
    bool guessed = false;
    if (guess > 0.f) {
        float Tg = guess * PVOL_GUESS_SCALE;
        if (Tg < T) { T = Tg; guessed = true; }
    }
    int count = 0;
    unsigned long long t0s = 0, selCy = 0;
    if (STATS) t0s = stamp();
    for (;;) {
        count = 0;
        const GridRows rows = grid_rows<FINE>(gv, pt, sqrtf(T), S.ringMax);
        for (int rb = 0; rb < rows.nrows; rb += LANES) {
            uint32_t start, rlen;
            grid_row_range<FINE>(gv, rows, rb + lane, pt, T, &start, &rlen);
            // ---- short rows (<= PAINT_ROW photons, the usual case): their photon indices are PAINTED into an LDS
            // list in concatenated order -- lane j writes start_j + it at off_j + it -- so that the candidate
            // loop below is one LDS read per 64 candidates instead of a per-lane search for "which row am I in"
            const uint32_t lenS = rlen <= PAINT_ROW ? rlen : 0u;
            // inclusive scan of the row lengths over the wave with DPP adds (no LDS round trips):
            // row_shr 1,2,4,8 scan each row of 16 lanes, row_bcast15/31 carry the row totals across rows
            uint32_t incl = lenS;
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, true);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, true);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, true);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, true);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xa, 0xf, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xc, 0xf, false);
            const uint32_t off = incl - lenS;
            const uint32_t total = (uint32_t)lane_i((int)incl, LANES - 1);   // <= 64 * PAINT_ROW == PAINT_CAP
            if (STATS) tested += total;
            uint64_t longRows = __ballot(rlen > PAINT_ROW);
            if (total) {
                __syncthreads();
                for (uint32_t it = 0; it < PAINT_ROW; ++it) {
                    if (!wave_any(it < lenS)) break;
                    if (it < lenS) G.paint[off + it] = start + it;
                }
                __syncthreads();
            }
            // candidate segments: first the painted list, then every long row by itself (contiguous photons)
            uint32_t segBase = 0u, segLen = total;
            bool painted = true;
            for (;;) {
                for (uint32_t cb = 0; cb < segLen; cb += LANES * NCH) {
                    f4 P[NCH];
                    uint32_t I[NCH];
                    bool on[NCH];
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        on[c] = false;
                        I[c] = 0u;
                        P[c] = mk4(0.f);
                        if (cb + c * LANES < segLen) {   // wave-uniform
                            uint32_t g = cb + c * LANES + lane;
                            on[c] = g < segLen;
                            if (on[c]) {
                                I[c] = painted ? G.paint[g] : segBase + g;
                                P[c] = S.pos4[I[c]];
                            }
                        }
                    }
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        if (cb + c * LANES >= segLen) continue;   // wave-uniform
                        float dx = P[c].x - pt.x, dyy = P[c].y - pt.y, dzz = P[c].z - pt.z;
                        float d2 = dx * dx + dyy * dyy + dzz * dzz;   // DistanceSquared(photon.p, p), kdtree.h:180
                        bool acc = on[c] && d2 < T;
                        uint64_t m = __ballot(acc);
                        if (m) {
                            if (acc) {
                                int pos = count + (int)lanes_below(m, lane);
                                G.cd[pos] = d2;
                                G.ci[pos] = I[c];
                            }
                            count += __popcll(m);
                            __syncthreads();
                            if (count > G.cap - LANES) {
                                unsigned long long ts = 0;
                                if (STATS) ts = stamp();
                                T = select_k<NREG>(G, count, k, T, lane);
                                if (STATS) selCy += stamp() - ts;
                                count = k;
                            }
                        }
                    }
                }
                if (!longRows) break;
                const int j = __ffsll((unsigned long long)longRows) - 1;
                longRows &= longRows - 1;
                painted = false;
                segBase = (uint32_t)lane_i((int)start, j);
                segLen = (uint32_t)lane_i((int)rlen, j);
                if (STATS) tested += segLen;
            }
        }
        if (guessed && count < k) {   // the guessed radius held fewer than k photons: grow it by what the count seen says about
            if (STATS) wc.retries += 1;   // the local density (k photons need ~ (k / count)^(2/3) x T), up to the full radius
            const float grow = count > 0 ? 1.35f * __builtin_amdgcn_exp2f(0.6666667f * __builtin_amdgcn_logf((float)k / (float)count)) : 1.0e9f;
            T = fminf(S.maxDistSq, T * fmaxf(2.f, grow));
            guessed = T < S.maxDistSq;
            continue;
        }
        break;
    }
    __syncthreads();
    unsigned long long t1s = 0;
    if (STATS) { t1s = stamp(); wc.cySearch += (t1s - t0s) - selCy; }
    int nFound = count;
    float maxmd;
    if (count > k) {
        // the usual case after a guessed radius: ~1.5-2 k candidates, two registers per lane are enough
        maxmd = (count <= 2 * LANES) ? select_k<2>(G, count, k, T, lane) : select_k<NREG>(G, count, k, T, lane);
        nFound = k;
    } else {
        float m = 0.f;
        for (int base = 0; base < nFound; base += LANES) {
            int i = base + lane;
            if (i < nFound) m = fmaxf(m, G.cd[i]);
        }
        maxmd = wave_max(m);
    }
    if (STATS) {
        unsigned long long t2s = stamp();
        wc.cySelect += selCy + (t2s - t1s);
        t1s = t2s;
        wc.tested += tested;
        if (nFound < 10) wc.lt10 += 1; else wc.kept += (unsigned long long)nFound;
    }
    if (nFound == k) *rkOut = maxmd;
    if (nFound < 10) return zero;   // photonvolume.cpp:83-84
    // totalFlux += alpha * p(pt_i, wi_i, -w): 8 photons per pass, one 128-B row per 8-lane group
    const int grp = lane >> 3;
    f4 acc = zero;
    const bool iso = (S.g == 0.f);
    const float wIso = 1.f / (4.f * K_PI);   // PhaseHG with g == 0: (1 - 0) / powf(1, 1.5) == 1
    V3 mw = -w;
    for (int base = 0; base < nFound; base += 64) {   // 8 rows per 8-lane group in flight
        uint32_t id[8];
        f4 a[8];
        float wg[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int e = base + j * 8 + grp;
            id[j] = e < nFound ? G.ci[e] : 0xffffffffu;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a[j] = zero;
            wg[j] = wIso;
            if (id[j] != 0xffffffffu) {
                a[j] = S.alpha4[(size_t)id[j] * 8 + q];
                if (!iso) { f4 wi = S.wi4[id[j]]; wg[j] = phase_hg(v3(wi.x, wi.y, wi.z), mw, S.g); }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = acc + a[j] * wg[j];
    }
    // sum the 8 photon groups (lanes l, l^8, l^16, l^32 ...): DPP row rotate, then the two permlane swaps
    acc.x += dppf<DPP_ROW_ROR8>(acc.x); acc.y += dppf<DPP_ROW_ROR8>(acc.y); acc.z += dppf<DPP_ROW_ROR8>(acc.z); acc.w += dppf<DPP_ROW_ROR8>(acc.w);
    acc.x = xor16_sum(acc.x); acc.y = xor16_sum(acc.y); acc.z = xor16_sum(acc.z); acc.w = xor16_sum(acc.w);
    acc.x = xor32_sum(acc.x); acc.y = xor32_sum(acc.y); acc.z = xor32_sum(acc.z); acc.w = xor32_sum(acc.w);
    float distSq = maxmd;
    float dV = distSq * sqrtf(distSq);
    f4 L = zero;
    if (dV != 0.f && !spec_is_black(sigS_at_p)) {
        float f = float(4.0 / 3.0 * (double)K_PI * (double)dV);   // photonvolume.cpp:103: double product -> float
        L = clean4(fdiv4(acc, sigS_at_p * f), q);
    }
    if (STATS) wc.cyFlux += stamp() - t1s;
    return L;
}

// ------------------------------------------------------------------------------------------ Li
#include "pvol_liargs.h"

// PhotonVolumeIntegrator::Transmittance with sample == NULL (photonvolume.cpp:15-30)
template <bool SEQ>
__device__ __forceinline__ f4 transmittance(const DevScene &S, const RayD &ray, Rng &rng, f4 sigT, int lane) {
    if (S.volKind == PVOL_VOLUME_NONE) return mk4(1.f);
    float step = 4.f * S.stepSize;
    float offset = rng_float<SEQ>(rng, lane);
    f4 tau = vol_tau(S, ray, step, offset, sigT);
    return exp4(neg4(tau));
}

// Per-lane precomputed part of Triangle::IntersectP (shapes/trianglemesh.cpp:211-243) for a fixed ray
// direction: e1, e2, s1 = d x e2 and 1/(s1.e1) do not depend on the ray origin, so for a distant light
// they are computed once per ray instead of once per march step (same operations, same values).
struct TriPre { V3 p1, e1, e2, s1; float invDivisor; bool valid; };
__device__ __forceinline__ TriPre tri_prepare(const DevScene &S, V3 d, int lane) {
    TriPre t;
    t.valid = lane < S.nTris;
    const DevTri &tr = S.tris[t.valid ? lane : 0];
    t.p1 = v3(tr.p1[0], tr.p1[1], tr.p1[2]);
    V3 p2 = v3(tr.p2[0], tr.p2[1], tr.p2[2]), p3 = v3(tr.p3[0], tr.p3[1], tr.p3[2]);
    t.e1 = p2 - t.p1;
    t.e2 = p3 - t.p1;
    t.s1 = cross(d, t.e2);
    float divisor = dot(t.s1, t.e1);
    t.valid = t.valid && divisor != 0.f;
    t.invDivisor = 1.f / divisor;
    return t;
}
__device__ __forceinline__ bool tri_test(const TriPre &t, V3 o, V3 d, float mint, float maxt) {
    if (!t.valid) return false;
    V3 s = o - t.p1;
    float b1 = dot(s, t.s1) * t.invDivisor;
    if (b1 < 0.f || b1 > 1.f) return false;
    V3 s2 = cross(s, t.e1);
    float b2 = dot(d, s2) * t.invDivisor;
    if (b2 < 0.f || b1 + b2 > 1.f) return false;
    float tt = dot(t.e2, s2) * t.invDivisor;
    if (tt < mint || tt > maxt) return false;
    return true;
}
// The same precomputation parked in LDS, one 16-float row per triangle (p1 | e1 | e2 | s1 | invDivisor, valid), for the
// kernels that test one shadow ray PER LANE against every triangle: with a distant light all shadow rays share their
// direction, so the first cross product (double precision as in geometry.h:477-484), the divisor and its reciprocal are
// computed once per kernel instead of once per lane, step and triangle.
__device__ void tri_rows_prepare(const DevScene &S, V3 d, float *rows, int lane) {
    if (lane < S.nTris) {
        const TriPre t = tri_prepare(S, d, lane);
        float *r = rows + 16 * lane;
        r[0] = t.p1.x; r[1] = t.p1.y; r[2] = t.p1.z; r[3] = t.e1.x; r[4] = t.e1.y; r[5] = t.e1.z; r[6] = t.e2.x; r[7] = t.e2.y;
        r[8] = t.e2.z; r[9] = t.s1.x; r[10] = t.s1.y; r[11] = t.s1.z; r[12] = t.invDivisor; r[13] = t.valid ? 1.f : 0.f; r[14] = 0.f; r[15] = 0.f;
    }
    __syncthreads();
}
__device__ __forceinline__ bool tri_rows_occluded(const float *rows, int nTris, V3 o, V3 d, float mint, float maxt) {
    bool occ = false;
    for (int t = 0; t < nTris; ++t) {
        const f4 r0 = *reinterpret_cast<const f4 *>(rows + 16 * t), r1 = *reinterpret_cast<const f4 *>(rows + 16 * t + 4),
                 r2 = *reinterpret_cast<const f4 *>(rows + 16 * t + 8), r3 = *reinterpret_cast<const f4 *>(rows + 16 * t + 12);
        TriPre tp;
        tp.p1 = v3(r0.x, r0.y, r0.z); tp.e1 = v3(r0.w, r1.x, r1.y); tp.e2 = v3(r1.z, r1.w, r2.x); tp.s1 = v3(r2.y, r2.z, r2.w);
        tp.invDivisor = r3.x; tp.valid = r3.y != 0.f;
        occ = occ | tri_test(tp, o, d, mint, maxt);
    }
    return occ;
}
// tau() of a homogeneous extent along a ray that STARTS INSIDE it (mint == 0): BBox::IntersectP
// (core/geometry.cpp:68-86) then leaves t0 == 0 exactly, so only the far slab distances remain.
// pv = WorldToVolume(o), dvInv = 1 / WorldToVolume(d) per axis.  Returns Distance(ray(0), ray(t1)).
__device__ __forceinline__ float inside_exit_length(const DevScene &S, V3 o, V3 d, V3 pv, V3 dvInv, float maxt) {
    float t1 = maxt;
    const float pp[3] = {pv.x, pv.y, pv.z}, ii[3] = {dvInv.x, dvInv.y, dvInv.z};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float tn = (S.extLo[i] - pp[i]) * ii[i];
        float tf = (S.extHi[i] - pp[i]) * ii[i];
        if (tn > tf) tf = tn;
        t1 = tf < t1 ? tf : t1;
    }
    V3 a = o + d * 0.f, b = o + d * t1;
    return len(a - b);
}

struct MarchLds {
    Gather G;
    float *lightNum;   // SEQ with > 1 light: LDShuffleScrambled1D output
    float *prevRk;     // k-th distance^2 per march step of the previous ray handled by this wave
};

// One call of PhotonVolumeIntegrator::Li (or Transmittance).  Returns false only in !SEQ mode when the ray
// needs a drawn VALUE (Russian roulette): the caller must redo the batch sequentially.
// How a march consumes the render tile's RNG stream:
//   MODE_SEQ      draws from the MT19937 state in LDS, everything in one pass (always legal)
//   MODE_PAR      only counts draws: legal when no drawn value can reach the result
//   MODE_RESOLVE  draws like MODE_SEQ but only RECORDS what the values decide (light per step, roulette
//                 outcome, tau() offsets) and skips gather and radiance: the cheap sequential pre-pass
//   MODE_REPLAY   no RNG: reads the records, so rays are independent and run one wave each
#define MODE_SEQ 0
#define MODE_PAR 1
#define MODE_RESOLVE 2
#define MODE_REPLAY 3
// Per-ray record slot: header {u32 nEff, u32 flags (bit0: the roulette killed the ray at step nEff), u32 draws,
// u32 pad}, then one byte per step (bits 0-2 light, bit 7 "survived a roulette: Tr /= 0.5"), then -- VolumeGrid
// only -- two floats per step (tau() offset of the step, offset of the shadow ray).
struct RayRec {
    uint32_t *hdr;
    unsigned char *stepByte;
    float *stepU;   // null unless VolumeGrid
};
__device__ __forceinline__ RayRec ray_rec(unsigned char *slot, int maxSteps, bool grid) {
    RayRec r;
    r.hdr = reinterpret_cast<uint32_t *>(slot);
    r.stepByte = slot + 16;
    r.stepU = grid ? reinterpret_cast<float *>(slot + 16 + ((maxSteps + 15) & ~15)) : 0;
    return r;
}

// One call of PhotonVolumeIntegrator::Li (or Transmittance).  Returns false only in MODE_PAR when the ray
// needs a drawn VALUE (Russian roulette): the caller must redo the batch sequentially.
template <bool STATS, int MODE, int NREG>
__device__ bool march_ray(const DevScene &S, const LiArgs &A, const pvol_ray &pr, Rng &rng, MarchLds &M, int lane,
                          WaveCounters &wc, f4 *LvOut, f4 *TrOut, RayRec rec) {
    constexpr bool RNGON = (MODE == MODE_SEQ || MODE == MODE_RESOLVE);
    constexpr bool RADIANCE = (MODE != MODE_RESOLVE);
    const int q = lane & 7;
    const f4 sigA = ld4(S.sigA, q), sigS = ld4(S.sigS, q);
    const f4 sigT = sigA + sigS;
    const f4 Y = ld4(S.cieY, q);
    const int nLights = S.nLights;
    const bool rainbow = (S.volKind == PVOL_VOLUME_RAINBOW);
    const bool grid = (S.volKind == PVOL_VOLUME_GRID);
    RayD ray;
    ray.o = v3(pr.o[0], pr.o[1], pr.o[2]);
    ray.d = v3(pr.d[0], pr.d[1], pr.d[2]);
    ray.mint = pr.mint;
    ray.maxt = pr.maxt;
    f4 Lv = mk4(0.f), Tr = mk4(1.f);
    ++wc.rays;   // rays and march steps are counted in every build (wave-uniform adds, one atomic per wave at the end)
    if (A.transmittanceOnly) {
        *TrOut = transmittance<RNGON>(S, ray, rng, sigT, lane);
        *LvOut = Lv;
        return true;
    }
    float t0, t1;
    bool hit = S.volKind != PVOL_VOLUME_NONE && vol_intersect(S, ray, &t0, &t1) && (t1 - t0) != 0.f;
    int nSamples = hit ? (int)ceilf((t1 - t0) / S.stepSize) : 0;
    if (MODE != MODE_PAR && hit && nSamples > S.maxSteps && (nLights > 1 || MODE >= MODE_RESOLVE)) {
        // the LDS / record plan cannot hold this ray's per-step arrays: report, never guess
        if (lane == 0) atomicAdd(&A.counters->nErrors, 1ull);
        hit = false;
        nSamples = 0;
    }
    int nEff = nSamples;
    bool killed = false;
    if (MODE == MODE_REPLAY) { nEff = (int)rec.hdr[0]; killed = (rec.hdr[1] & 1u) != 0u; }
    if (hit) {
        float step = (t1 - t0) / nSamples;
        V3 p = ray.o + ray.d * t0, pPrev;
        V3 w = -ray.d;
        t0 += pr.scatter_u * step;
        // LDShuffleScrambled1D(1,n,lightNum) + (1,n,lightComp) + 2D(1,n,lightPos): 4+6n draws
        // (photonvolume.cpp:137-142).  The three delta lights ignore lightComp/lightPos, and lightNum only
        // matters with more than one light: what is never read is only counted.
        if (RNGON && nLights > 1) {
            float *lightNum = M.lightNum;
            uint32_t scramble = rng_uint<RNGON>(rng, lane);
            for (int i = lane; i < nSamples; i += LANES) lightNum[i] = van_der_corput((uint32_t)i, scramble);
            rng_skip<RNGON>(rng, (unsigned long long)nSamples, lane);  // n one-element shuffles (montecarlo.h:308-309)
            __syncthreads();
            for (int base = 0; base < nSamples; base += LANES) {       // Shuffle(samples, n, 1), montecarlo.h:174-181: draws and
                const int cnt = min(LANES, nSamples - base);            // remainders 64 at a time (as lite_ray does), the swaps serial
                const uint32_t y = rng_bulk(rng, cnt, lane);
                const int iMine = base + lane;
                const uint32_t oth = lane < cnt ? (uint32_t)iMine + (y % (uint32_t)(nSamples - iMine)) : 0u;
                for (int j = 0; j < cnt; ++j) {
                    const uint32_t other = (uint32_t)__builtin_amdgcn_readlane((int)oth, j);
                    if (lane == 0) {
                        float a = lightNum[base + j], b = lightNum[other];
                        lightNum[base + j] = b;
                        lightNum[other] = a;
                    }
                }
            }
            __syncthreads();
            rng_skip<RNGON>(rng, 3ull + 4ull * (unsigned long long)nSamples, lane);
        } else {
            rng_skip<RNGON>(rng, 4ull + 6ull * (unsigned long long)nSamples, lane);
        }
        const f4 le = ld4(S.le, q);
        // step invariants where the density factor is exactly 1 (inside a homogeneous extent):
        // sigma * 1.f == sigma, so these are the very values the per-step expressions would produce
        const float ySa1 = spec_y(sigA, Y), ySs1 = spec_y(sigS, Y);
        const bool blackS1 = spec_is_black(sigS);
        const f4 albedo1 = clean4(fdiv4(sigS, sigA + sigS), q);
        float lastRk = 0.f;
        const bool analytic = !grid;
        bool inPrev = analytic && box_inside(S.extLo, S.extHi, xform_point(S.w2v, p));
        int cachedLn = -1;          // light whose direction-only shadow-ray terms are cached (distant lights)
        TriPre triPre;
        V3 cDvInv = v3(0.f, 0.f, 0.f);
        triPre.valid = false;
        for (int i = 0; i < nEff; ++i, t0 += step) {
            ++wc.steps;
            pPrev = p;
            p = ray.o + ray.d * t0;
            const V3 pv = xform_point(S.w2v, p);
            const bool inP = analytic && box_inside(S.extLo, S.extHi, pv);
            unsigned int recByte = 0u;
            if (MODE == MODE_REPLAY) recByte = rec.stepByte[i];
            f4 stepTau;
            float uTau = rng_float<RNGON>(rng, lane);
            if (MODE == MODE_REPLAY && grid) uTau = rec.stepU[2 * i];
            if (MODE == MODE_RESOLVE && grid && lane == 0) rec.stepU[2 * i] = uTau;
            if (inPrev && inP) {
                // both ends inside the extent: the slab clip of the [0,1] segment is the identity
                // (homogeneous.h:80-84 -> Distance(ray(0), ray(1)) * sigma_t)
                V3 dseg = p - pPrev;
                V3 a = pPrev + dseg * 0.f, b = pPrev + dseg * 1.f;
                stepTau = sigT * len(a - b);
            } else {
                RayD tauRay;
                tauRay.o = pPrev; tauRay.d = p - pPrev; tauRay.mint = 0.f; tauRay.maxt = 1.f;
                stepTau = vol_tau(S, tauRay, .5f * S.stepSize, uTau, sigT);
            }
            inPrev = inP;
            Tr = exp4(neg4(stepTau));   // assigned, not accumulated (photonvolume.cpp:155)
            if (MODE == MODE_REPLAY) {
                if (recByte & 0x80u) Tr = Tr / .5f;
            } else if (spec_y(Tr, Y) < 1e-3) {
                if (MODE == MODE_PAR) return false;  // the roulette compares a drawn value
                const float continueProb = .5f;
                if (rng_float<RNGON>(rng, lane) > continueProb) {
                    Tr = mk4(0.f);
                    if (MODE == MODE_RESOLVE) { nEff = i; killed = true; }
                    break;
                }
                Tr = Tr / continueProb;
                recByte |= 0x80u;
            }
            const float dens = analytic ? (inP ? 1.f : 0.f) : grid_density(S, pv);   // homogeneous.h:64-75 / volume.h:81-92
            f4 ss = sigS * dens, sa = sigA * dens;
            f4 L_d = mk4(0.f), L_ii = mk4(0.f), L_i;
            const bool unit = (dens == 1.f);
            if (!(unit ? blackS1 : spec_is_black(ss)) && nLights > 0) {
                int ln = 0;
                if (RNGON && nLights > 1) ln = min((int)floorf(M.lightNum[i] * nLights), nLights - 1);
                if (MODE == MODE_REPLAY) ln = (int)(recByte & 7u);
                recByte |= (unsigned int)ln;
                const DevLight &light = S.lights[ln];
                // Light::Sample_L(p, 0, ...): distant.cpp:48-55, point.cpp:50-57, spot.cpp:50-57
                V3 wo;
                RayD vis;
                f4 L = ld4(light.intensity, q);
                if (light.kind == PVOL_LIGHT_DISTANT) {
                    wo = v3(light.dir[0], light.dir[1], light.dir[2]);
                    vis.o = p; vis.d = wo; vis.mint = 0.f; vis.maxt = INFINITY;
                } else {
                    V3 lp = v3(light.pos[0], light.pos[1], light.pos[2]);
                    wo = normalize(lp - p);
                    float dist = len(p - lp);
                    vis.o = p; vis.d = vdiv(lp - p, dist); vis.mint = 0.f; vis.maxt = dist * (1.f - 0.f);
                    float d2 = len_sq(lp - p);
                    if (light.kind == PVOL_LIGHT_SPOT) {
                        // SpotLight::Falloff(-wi), spot.cpp:60-69
                        V3 wl = normalize(v3(light.w2l[0] * -wo.x + light.w2l[1] * -wo.y + light.w2l[2] * -wo.z,
                                             light.w2l[4] * -wo.x + light.w2l[5] * -wo.y + light.w2l[6] * -wo.z,
                                             light.w2l[8] * -wo.x + light.w2l[9] * -wo.y + light.w2l[10] * -wo.z));
                        float costheta = wl.z, fall;
                        if (costheta < light.cosTotalWidth) fall = 0.f;
                        else if (costheta > light.cosFalloffStart) fall = 1.f;
                        else {
                            float delta = (costheta - light.cosTotalWidth) / (light.cosFalloffStart - light.cosTotalWidth);
                            fall = delta * delta * delta * delta;
                        }
                        L = L * fall / d2;
                    } else {
                        L = L / d2;
                    }
                }
                const float pdf = 1.f;
                const bool distant = (light.kind == PVOL_LIGHT_DISTANT);
                if (distant && cachedLn != ln && (S.nTris <= LANES && !S.bvhNodes && !S.nSpheres)) {
                    triPre = tri_prepare(S, vis.d, lane);
                    V3 dv = xform_vector(S.w2v, vis.d);
                    cDvInv = v3(1.f / dv.x, 1.f / dv.y, 1.f / dv.z);
                    cachedLn = ln;
                }
                bool lit = !spec_is_black(L) && pdf > 0.f;
                if (lit) {
                    if (distant && (S.nTris <= LANES && !S.bvhNodes && !S.nSpheres)) lit = !wave_any(tri_test(triPre, vis.o, vis.d, vis.mint, vis.maxt));
                    else lit = !scene_occluded(S, vis, lane);
                }
                if (lit) {
                    if (STATS) ++wc.unocc;
                    // vis.Transmittance(..., NULL, rng, ...): one draw (photonvolume.cpp:24-27)
                    float uSh = rng_float<RNGON>(rng, lane);
                    if (MODE == MODE_REPLAY && grid) uSh = rec.stepU[2 * i + 1];
                    if (MODE == MODE_RESOLVE && grid && lane == 0) rec.stepU[2 * i + 1] = uSh;
                    if (RADIANCE) {
                        f4 Ttr;
                        if (inP) {   // analytic tau() from a point inside the extent
                            V3 dvInv = cDvInv;
                            if (!(distant && (S.nTris <= LANES && !S.bvhNodes && !S.nSpheres))) {
                                V3 dv = xform_vector(S.w2v, vis.d);
                                dvInv = v3(1.f / dv.x, 1.f / dv.y, 1.f / dv.z);
                            }
                            Ttr = exp4(neg4(sigT * inside_exit_length(S, vis.o, vis.d, pv, dvInv, vis.maxt)));
                        } else {
                            Ttr = exp4(neg4(vol_tau(S, vis, 4.f * S.stepSize, uSh, sigT)));
                        }
                        f4 Ld = L * Ttr;
                        if (rainbow) L_d = rainbow_reflection(Ld, ray.d, wo, q);
                        else {
                            // vr->p(p, w, -wo): homogeneous.h:76-79 (0 outside the extent), DensityRegion::p otherwise
                            float ph = (analytic && !inP) ? 0.f : phase_hg(w, -wo, S.g);
                            L_d = Ld * ph * float(nLights) / pdf;
                        }
                    }
                }
            }
            if (MODE == MODE_RESOLVE && lane == 0) rec.stepByte[i] = (unsigned char)recByte;
            if (RADIANCE) {
                if (!rainbow) {
                    float guess = i < PREV_N ? fmaxf(M.prevRk[i], lastRk) : lastRk;
                    float rk;
                    L_ii = lphoton<STATS, NREG>(S, M.G, w, p, ss, lane, wc, guess, &rk);
                    lastRk = rk;
                    if (i < PREV_N && lane == 0) M.prevRk[i] = rk;
                }
                const float ySa = unit ? ySa1 : spec_y(sa, Y), ySs = unit ? ySs1 : spec_y(ss, Y);
                if (ySa != 0.0 || ySs != 0.0) L_i = L_d + (unit ? albedo1 : clean4(fdiv4(ss, sa + ss), q)) * L_ii;
                else L_i = L_d;
                Lv = (sa * (le * dens) * step) + (ss * L_i * step) + (Tr * Lv);
            }
        }
        if (MODE == MODE_REPLAY && killed) Tr = mk4(0.f);
    }
    if (MODE == MODE_RESOLVE && lane == 0) { rec.hdr[0] = (uint32_t)nEff; rec.hdr[1] = killed ? 1u : 0u; }
    *LvOut = Lv;
    *TrOut = Tr;
    return true;
}

// ---- blocked march for the two ray-parallel modes over an analytic (homogeneous / rainbow) medium.
// Everything about a march step that is SCALAR -- sample point, inside test, segment length, light
// geometry, shadow-ray occlusion, exit distance of the shadow ray -- is evaluated for 64 steps at once,
// one step per lane; the serial loop that follows only does the spectral arithmetic and the gather, reading
// each step's scalars back with wave-uniform lane reads.  Returns 0 = done, 1 = the ray needs the roulette
// (MODE_PAR: redo sequentially), 2 = not applicable (the caller runs march_ray).
// tau() length of a homogeneous extent along a general segment (homogeneous.h:80-84): Distance(ray(t0), ray(t1))
__device__ __forceinline__ float analytic_tau_length(const DevScene &S, V3 o, V3 d, float mint, float maxt) {
    RayD r;
    r.o = o; r.d = d; r.mint = mint; r.maxt = maxt;
    float t0, t1;
    if (!vol_intersect(S, r, &t0, &t1)) return 0.f;
    V3 a = o + d * t0, b = o + d * t1;
    return len(a - b);
}
__device__ __forceinline__ bool lane_occluded(const DevScene &S, const RayD &vis) {
    if (S.nSpheres && spheres_occluded(S, vis.o, vis.d, vis.mint, vis.maxt)) return true;
    if (S.bvhNodes) return bvh_occluded(S, vis.o, vis.d, vis.mint, vis.maxt);
    bool hit = false;
    for (int t = 0; t < S.nTris && !hit; ++t) hit = tri_hit(S.tris[t], vis);
    return hit;
}

template <bool STATS, int MODE, int NREG>
__device__ int march_ray_blocked(const DevScene &S, const LiArgs &A, const pvol_ray &pr, Rng &rng, MarchLds &M, int lane,
                                 WaveCounters &wc, f4 *LvOut, f4 *TrOut, RayRec rec) {
    static_assert(MODE == MODE_PAR || MODE == MODE_REPLAY, "ray-parallel modes only");
    if (S.volKind == PVOL_VOLUME_GRID || S.volKind == PVOL_VOLUME_NONE || A.transmittanceOnly) return 2;
    const int q = lane & 7;
    const f4 sigA = ld4(S.sigA, q), sigS = ld4(S.sigS, q);
    const f4 sigT = sigA + sigS;
    const f4 Y = ld4(S.cieY, q);
    const int nLights = S.nLights;
    const bool rainbow = (S.volKind == PVOL_VOLUME_RAINBOW);
    RayD ray;
    ray.o = v3(pr.o[0], pr.o[1], pr.o[2]);
    ray.d = v3(pr.d[0], pr.d[1], pr.d[2]);
    ray.mint = pr.mint;
    ray.maxt = pr.maxt;
    f4 Lv = mk4(0.f), Tr = mk4(1.f);
    float t0, t1;
    bool hit = vol_intersect(S, ray, &t0, &t1) && (t1 - t0) != 0.f;
    int nSamples = hit ? (int)ceilf((t1 - t0) / S.stepSize) : 0;
    if (MODE == MODE_REPLAY && hit && nSamples > S.maxSteps) return 2;   // march_ray reports it
    ++wc.rays;
    int nEff = nSamples;
    bool killed = false;
    if (MODE == MODE_REPLAY) { nEff = (int)rec.hdr[0]; killed = (rec.hdr[1] & 1u) != 0u; }
    if (hit) {
        const float step = (t1 - t0) / nSamples;
        const V3 pEntry = ray.o + ray.d * t0;
        const V3 w = -ray.d;
        float tcur = t0 + pr.scatter_u * step;
        rng_skip<false>(rng, 4ull + 6ull * (unsigned long long)nSamples, lane);
        const f4 le = ld4(S.le, q);
        const float ySa1 = spec_y(sigA, Y), ySs1 = spec_y(sigS, Y);
        const bool blackS1 = spec_is_black(sigS);
        const f4 albedo1 = clean4(fdiv4(sigS, sigA + sigS), q);
        // largest sigma_t bin: Tr.y() >= Y(1) * exp(-len * sigTmax), so the roulette (Tr.y() < 1e-3) cannot
        // fire while len * sigTmax < 6.8
        float sigTmax = fmaxf(fmaxf(sigT.x, sigT.y), fmaxf(sigT.z, sigT.w));
        sigTmax = wave_max(sigTmax);
        unsigned int lightBlackMask = 0u;   // lights whose intensity spectrum is black
        for (int l = 0; l < nLights; ++l) if (spec_is_black(ld4(S.lights[l].intensity, q))) lightBlackMask |= 1u << l;
        float lastRk = 0.f;
        V3 pCarry = pEntry;                                                  // p of the step before the block
        bool inCarry = box_inside(S.extLo, S.extHi, xform_point(S.w2v, pEntry));
        for (int base = 0; base < nEff; base += LANES) {
            const int cnt = min(LANES, nEff - base);
            // t0 is ACCUMULATED in the reference (`t0 += step` per iteration): replay the same additions
            float tMine = 0.f;
            for (int j = 0; j < cnt; ++j) {
                if (lane == j) tMine = tcur;
                tcur += step;
            }
            const bool on = lane < cnt;
            const V3 p = ray.o + ray.d * tMine;
            const V3 pv = xform_point(S.w2v, p);
            const bool inP = on && box_inside(S.extLo, S.extHi, pv);
            // pPrev / inPrev of lane j = p / inP of lane j-1 (block carry for lane 0)
            V3 pPrev;
            pPrev.x = __shfl_up(p.x, 1); pPrev.y = __shfl_up(p.y, 1); pPrev.z = __shfl_up(p.z, 1);
            int inPrevI = __shfl_up(inP ? 1 : 0, 1);
            if (lane == 0) { pPrev = pCarry; inPrevI = inCarry ? 1 : 0; }
            const bool inPrev = inPrevI != 0;
            float lenStep;
            {
                V3 dseg = p - pPrev;
                if (inPrev && inP) {
                    V3 a = pPrev + dseg * 0.f, b = pPrev + dseg * 1.f;
                    lenStep = len(a - b);
                } else {
                    lenStep = analytic_tau_length(S, pPrev, dseg, 0.f, 1.f);
                }
            }
            // carry for the next block
            pCarry.x = lane_f(p.x, cnt - 1); pCarry.y = lane_f(p.y, cnt - 1); pCarry.z = lane_f(p.z, cnt - 1);
            inCarry = lane_i(inP ? 1 : 0, cnt - 1) != 0;
            if (wave_any(on && !(lenStep * sigTmax < 6.8f))) {
                if (MODE == MODE_PAR) return 1;
            }
            // direct lighting geometry of this lane's step
            unsigned int recByte = 0u;
            if (MODE == MODE_REPLAY && on) recByte = rec.stepByte[base + lane];
            const int ln = (MODE == MODE_REPLAY) ? (int)(recByte & 7u) : 0;
            float fallReg = 1.f, d2Reg = 1.f, exitLen = 0.f, ph = 0.f;
            V3 wo = v3(0.f, 0.f, 0.f);
            bool lit = false;
            if (on && inP && !blackS1 && nLights > 0) {   // dens == 1 inside, 0 outside: sigma_s is black outside
                const DevLight &light = S.lights[ln];
                RayD vis;
                if (light.kind == PVOL_LIGHT_DISTANT) {   // distant.cpp:48-55
                    wo = v3(light.dir[0], light.dir[1], light.dir[2]);
                    vis.o = p; vis.d = wo; vis.mint = 0.f; vis.maxt = INFINITY;
                } else {                                  // point.cpp:50-57, spot.cpp:50-57
                    V3 lp = v3(light.pos[0], light.pos[1], light.pos[2]);
                    wo = normalize(lp - p);
                    float dist = len(p - lp);
                    vis.o = p; vis.d = vdiv(lp - p, dist); vis.mint = 0.f; vis.maxt = dist * (1.f - 0.f);
                    d2Reg = len_sq(lp - p);
                    if (light.kind == PVOL_LIGHT_SPOT) {  // SpotLight::Falloff(-wi), spot.cpp:60-69
                        V3 wl = normalize(v3(light.w2l[0] * -wo.x + light.w2l[1] * -wo.y + light.w2l[2] * -wo.z,
                                             light.w2l[4] * -wo.x + light.w2l[5] * -wo.y + light.w2l[6] * -wo.z,
                                             light.w2l[8] * -wo.x + light.w2l[9] * -wo.y + light.w2l[10] * -wo.z));
                        float costheta = wl.z;
                        if (costheta < light.cosTotalWidth) fallReg = 0.f;
                        else if (costheta > light.cosFalloffStart) fallReg = 1.f;
                        else {
                            float delta = (costheta - light.cosTotalWidth) / (light.cosFalloffStart - light.cosTotalWidth);
                            fallReg = delta * delta * delta * delta;
                        }
                    }
                }
                // L.IsBlack(): I * fall / d2 has no non-zero bin
                const bool black = (fallReg == 0.f) || ((lightBlackMask >> ln) & 1u);
                if (!black && !lane_occluded(S, vis)) {
                    lit = true;
                    V3 dv = xform_vector(S.w2v, vis.d);
                    V3 dvInv = v3(1.f / dv.x, 1.f / dv.y, 1.f / dv.z);
                    exitLen = inside_exit_length(S, vis.o, vis.d, pv, dvInv, vis.maxt);
                    ph = phase_hg(w, -wo, S.g);
                }
            }
            // draws of the block: one tau() offset per step + one per lit shadow ray (roulette excluded above)
            rng.draws += (unsigned long long)cnt + (unsigned long long)__popcll(__ballot(lit));
            wc.steps += cnt;
            if (STATS) wc.unocc += __popcll(__ballot(lit));
            // ---- serial part: spectral arithmetic + gather, one step at a time
            for (int j = 0; j < cnt; ++j) {
                const float lenJ = lane_f(lenStep, j);
                const bool inJ = lane_i(inP ? 1 : 0, j) != 0;
                const bool litJ = lane_i(lit ? 1 : 0, j) != 0;
                const V3 pJ = v3(lane_f(p.x, j), lane_f(p.y, j), lane_f(p.z, j));
                Tr = exp4(neg4(sigT * lenJ));   // assigned, not accumulated (photonvolume.cpp:155)
                if (MODE == MODE_REPLAY) {
                    const unsigned int rb = (unsigned int)lane_i((int)recByte, j);
                    if (rb & 0x80u) Tr = Tr / .5f;
                }
                const float dens = inJ ? 1.f : 0.f;
                f4 ss = sigS * dens, sa = sigA * dens;
                f4 L_d = mk4(0.f), L_ii = mk4(0.f), L_i;
                if (litJ) {
                    const int lnJ = (MODE == MODE_REPLAY) ? (lane_i((int)recByte, j) & 7) : 0;
                    const DevLight &light = S.lights[lnJ];
                    f4 L = ld4(light.intensity, q);
                    if (light.kind != PVOL_LIGHT_DISTANT) L = L * lane_f(fallReg, j) / lane_f(d2Reg, j);
                    f4 Ttr = exp4(neg4(sigT * lane_f(exitLen, j)));
                    f4 Ld = L * Ttr;
                    if (rainbow) {
                        V3 woJ = v3(lane_f(wo.x, j), lane_f(wo.y, j), lane_f(wo.z, j));
                        L_d = rainbow_reflection(Ld, ray.d, woJ, q);
                    } else {
                        L_d = Ld * lane_f(ph, j) * float(nLights) / 1.f;
                    }
                }
                if (!rainbow) {
                    const int i = base + j;
                    float guess = i < PREV_N ? fmaxf(M.prevRk[i], lastRk) : lastRk;
                    float rk;
                    L_ii = lphoton<STATS, NREG>(S, M.G, w, pJ, ss, lane, wc, guess, &rk);
                    lastRk = rk;
                    if (i < PREV_N && lane == 0) M.prevRk[i] = rk;
                }
                if (inJ ? (ySa1 != 0.0 || ySs1 != 0.0) : false) L_i = L_d + albedo1 * L_ii;
                else L_i = L_d;
                Lv = (sa * (le * dens) * step) + (ss * L_i * step) + (Tr * Lv);
            }
        }
        if (MODE == MODE_REPLAY && killed) Tr = mk4(0.f);
    }
    *LvOut = Lv;
    *TrOut = Tr;
    return 0;
}

__device__ __forceinline__ void write_outputs(const DevScene &S, const LiArgs &A, size_t ri, f4 Lv, f4 Tr, int lane) {
    const int q = lane & 7;
    if (A.outputKind == PVOL_OUT_SPECTRAL) {
        if (lane < 8) {
            float *o = A.out + ri * 60;
            const float lv[4] = {Lv.x, Lv.y, Lv.z, Lv.w}, tr[4] = {Tr.x, Tr.y, Tr.z, Tr.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                int bin = 4 * q + c;
                if (bin < 30) { o[bin] = lv[c]; o[30 + bin] = tr[c]; }
            }
        }
    } else {
        f4 X4 = ld4(S.cieX, q), Y = ld4(S.cieY, q), Z4 = ld4(S.cieZ, q);
        float scale = float(700 - 400) / float(106.856895f * 30);   // core/spectrum.h:427-428
        float x = group8_sum(X4.x * Lv.x + X4.y * Lv.y + X4.z * Lv.z + X4.w * Lv.w) * scale;
        float y = group8_sum(Y.x * Lv.x + Y.y * Lv.y + Y.z * Lv.z + Y.w * Lv.w) * scale;
        float z = group8_sum(Z4.x * Lv.x + Z4.y * Lv.y + Z4.z * Lv.z + Z4.w * Lv.w) * scale;
        float ty = spec_y(Tr, Y);
        if (lane == 0) *reinterpret_cast<f4 *>(A.out + ri * 4) = make_float4(x, y, z, ty);
    }
}

template <bool STATS>
__device__ __forceinline__ void flush_counters(DevCounters *c, const WaveCounters &wc, unsigned long long t0, int lane) {
    if (lane == 0) {   // Li() calls and march steps (= photon lookups): always, they feed the bench's per-sample figures
        if (wc.rays) atomicAdd(&c->nRays, wc.rays);
        if (wc.steps) atomicAdd(&c->nSteps, wc.steps);
    }
    if (STATS && lane == 0) {
        atomicAdd(&c->nShadowUnoccluded, wc.unocc);
        atomicAdd(&c->nTested, wc.tested);
        atomicAdd(&c->nKept, wc.kept);
        atomicAdd(&c->nLookupsLt10, wc.lt10);
        atomicAdd(&c->pad, wc.retries);
        atomicAdd(&c->cySearch, wc.cySearch);
        atomicAdd(&c->cySelect, wc.cySelect);
        atomicAdd(&c->cyFlux, wc.cyFlux);
        atomicAdd(&c->cyTotal, stamp() - t0);
        atomicAdd(&c->diag[0], wc.diag0); atomicAdd(&c->diag[1], wc.diag1); atomicAdd(&c->diag[2], wc.diag2);
        atomicAdd(&c->diag[3], wc.diag3); atomicAdd(&c->diag[4], wc.diag4); atomicAdd(&c->diag[5], wc.diag5);
    }
}

// LDS plan (bytes): [MT 2496 (SEQ)] | cand d2 cap*4 | cand idx cap*4 | lightNum maxSteps*4 (SEQ) | prevRk PREV_N*4 | paint PAINT_CAP*4
template <bool STATS, int NREG>
__global__ __launch_bounds__(LANES, (NREG > 4 ? PVOL_WPE_BIG : PVOL_WPE)) void li_seq_kernel(LiArgs A) {
    extern __shared__ __align__(16) unsigned char lds[];
    const DevScene &S = *A.scene;
    const int lane = threadIdx.x;
    const uint32_t sidx = blockIdx.x;
    if (sidx >= A.nStreams) return;
    if (A.gated && *A.needSeq == 0u) return;
    uint32_t *mt = reinterpret_cast<uint32_t *>(lds);
    MarchLds M;
    M.G.cap = S.candCap;
    M.G.cd = reinterpret_cast<float *>(lds + MT_N * 4);
    M.G.ci = reinterpret_cast<uint32_t *>(lds + MT_N * 4 + (size_t)M.G.cap * 4);
    M.lightNum = reinterpret_cast<float *>(lds + MT_N * 4 + (size_t)M.G.cap * 8);
    M.prevRk = M.lightNum + S.maxSteps;
    M.G.paint = reinterpret_cast<uint32_t *>(M.prevRk + PREV_N);
    for (int i = lane; i < PREV_N; i += LANES) M.prevRk[i] = 0.f;
    __syncthreads();

    pvol_stream st = A.streams[sidx];
    Rng rng;
    rng.mt = mt;
    rng.draws = 0;
    if (A.initState) {
        const uint32_t *src = A.initState + (size_t)sidx * (MT_N + 1);
        for (int i = lane; i < MT_N; i += LANES) mt[i] = src[i];
        rng.mti = (int)src[MT_N];
        rng.draws = st.start_draw;
        __syncthreads();
    } else {
        mt_seed(mt, st.seed, lane);
        rng.mti = MT_N;
        rng_skip<true>(rng, st.start_draw, lane);
    }
    WaveCounters wc = {};
    unsigned long long tk0 = STATS ? stamp() : 0ull;
    for (uint32_t k = 0; k < st.n_rays; ++k) {
        const size_t ri = (size_t)st.first_ray + k;
        const pvol_ray pr = A.rays[ri];
        rng_skip<true>(rng, pr.rng_skip, lane);
        const unsigned long long d0 = rng.draws;
        f4 Lv, Tr;
        RayRec none = {0, 0, 0};
        march_ray<STATS, MODE_SEQ, NREG>(S, A, pr, rng, M, lane, wc, &Lv, &Tr, none);
        write_outputs(S, A, ri, Lv, Tr, lane);
        if (A.draws && lane == 0) A.draws[ri] = (uint32_t)(rng.draws - d0);
    }
    if (lane == 0) A.streams[sidx].end_draw = rng.draws;
    if (A.finalState) {
        uint32_t *dst = A.finalState + (size_t)sidx * (MT_N + 1);
        __syncthreads();
        for (int i = lane; i < MT_N; i += LANES) dst[i] = mt[i];
        if (lane == 0) dst[MT_N] = (uint32_t)rng.mti;
    }
    flush_counters<STATS>(A.counters, wc, tk0, lane);
}

// stream of ray `ri`: streams are sorted by first_ray and cover disjoint ranges
__device__ __forceinline__ uint32_t stream_of(const pvol_stream *st, uint32_t n, uint32_t ri) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (st[mid].first_ray <= ri) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ void stream_begin_kernel(pvol_stream *st, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) st[i].end_draw = st[i].start_draw;
}

template <bool STATS, int NREG>
__global__ __launch_bounds__(LANES, (NREG > 4 ? PVOL_WPE_BIG : PVOL_WPE)) void li_par_kernel(LiArgs A) {
    extern __shared__ __align__(16) unsigned char lds[];
    const DevScene &S = *A.scene;
    const int lane = threadIdx.x;
    MarchLds M;
    M.G.cap = S.candCap;
    M.G.cd = reinterpret_cast<float *>(lds);
    M.G.ci = reinterpret_cast<uint32_t *>(lds + (size_t)M.G.cap * 4);
    M.lightNum = 0;
    M.prevRk = reinterpret_cast<float *>(lds + (size_t)M.G.cap * 8);
    M.G.paint = reinterpret_cast<uint32_t *>(M.prevRk + PREV_N);
    for (int i = lane; i < PREV_N; i += LANES) M.prevRk[i] = 0.f;
    __syncthreads();
    Rng rng;
    rng.mt = 0;
    rng.mti = 0;
    WaveCounters wc = {};
    unsigned long long tk0 = STATS ? stamp() : 0ull;
    for (;;) {
        uint32_t chunk = 0;
        if (lane == 0) chunk = atomicAdd(A.chunkCounter, 1u);
        chunk = (uint32_t)lane_i((int)chunk, 0);
        unsigned long long r0 = (unsigned long long)chunk * CHUNK_RAYS;
        if (r0 >= A.nRays) break;
        uint32_t r1 = (uint32_t)min((unsigned long long)A.nRays, r0 + CHUNK_RAYS);
        uint32_t sidx = stream_of(A.streams, A.nStreams, (uint32_t)r0);
        uint32_t sEnd = A.streams[sidx].first_ray + A.streams[sidx].n_rays;
        unsigned long long acc = 0;   // draws of the current stream seen by this wave
        for (uint32_t ri = (uint32_t)r0; ri < r1; ++ri) {
            while (ri >= sEnd && sidx + 1 < A.nStreams) {   // chunk crosses into the next stream
                if (lane == 0 && acc) atomicAdd((unsigned long long *)&A.streams[sidx].end_draw, acc);
                acc = 0;
                ++sidx;
                sEnd = A.streams[sidx].first_ray + A.streams[sidx].n_rays;
            }
            const pvol_ray pr = A.rays[ri];
            rng.draws = 0;
            f4 Lv, Tr;
            RayRec none = {0, 0, 0};
            int brc = march_ray_blocked<STATS, MODE_PAR, NREG>(S, A, pr, rng, M, lane, wc, &Lv, &Tr, none);
            bool okRay = (brc == 0);
            if (brc == 2) { rng.draws = 0; okRay = march_ray<STATS, MODE_PAR, NREG>(S, A, pr, rng, M, lane, wc, &Lv, &Tr, none); }
            if (!okRay) {
                if (lane == 0) atomicOr(A.needSeq, 1u);
                continue;
            }
            write_outputs(S, A, ri, Lv, Tr, lane);
            if (A.draws && lane == 0) A.draws[ri] = (uint32_t)rng.draws;
            acc += rng.draws + pr.rng_skip;
        }
        if (lane == 0 && acc) atomicAdd((unsigned long long *)&A.streams[sidx].end_draw, acc);
    }
    flush_counters<STATS>(A.counters, wc, tk0, lane);
}

// Sequential pre-pass of slice sliceK: one wave per stream draws through the slice's rays in order and records
// what the drawn values decide.  No gather, no radiance: a few hundred instructions per march step.
template <bool STATS, int NREG>
__global__ __launch_bounds__(LANES, PVOL_WPE) void li_resolve_kernel(LiArgs A) {
    extern __shared__ __align__(16) unsigned char lds[];
    const DevScene &S = *A.scene;
    const int lane = threadIdx.x;
    const uint32_t sidx = blockIdx.x;
    if (sidx >= A.nStreams) return;
    uint32_t *mt = reinterpret_cast<uint32_t *>(lds);
    MarchLds M;
    M.G.cap = 0; M.G.cd = 0; M.G.ci = 0; M.G.paint = 0;
    M.lightNum = reinterpret_cast<float *>(lds + MT_N * 4);
    M.prevRk = 0;
    pvol_stream st = A.streams[sidx];
    const uint32_t begin = A.sliceK * A.sliceM;
    if (begin >= st.n_rays && !(A.sliceK == 0)) return;
    Rng rng;
    rng.mt = mt;
    rng.draws = 0;
    uint32_t *state = A.state + (size_t)sidx * (MT_N + 1);
    if (A.sliceK == 0 && !A.initState) {
        mt_seed(mt, st.seed, lane);
        rng.mti = MT_N;
        rng_skip<true>(rng, st.start_draw, lane);
    } else {
        const uint32_t *src = (A.sliceK == 0) ? A.initState + (size_t)sidx * (MT_N + 1) : state;
        for (int i = lane; i < MT_N; i += LANES) mt[i] = src[i];
        rng.mti = (int)src[MT_N];
        rng.draws = (A.sliceK == 0) ? st.start_draw : st.end_draw;
        __syncthreads();
    }
    WaveCounters wc = {};
    const bool grid = (S.volKind == PVOL_VOLUME_GRID);
    const uint32_t end = min(st.n_rays, begin + A.sliceM);
    for (uint32_t k = begin; k < end; ++k) {
        const size_t ri = (size_t)st.first_ray + k;
        const pvol_ray pr = A.rays[ri];
        rng_skip<true>(rng, pr.rng_skip, lane);
        const unsigned long long d0 = rng.draws;
        RayRec rec = ray_rec(A.records + ((size_t)sidx * A.sliceM + (k - begin)) * A.recStride, S.maxSteps, grid);
        f4 Lv, Tr;
        march_ray<false, MODE_RESOLVE, NREG>(S, A, pr, rng, M, lane, wc, &Lv, &Tr, rec);
        if (lane == 0) rec.hdr[2] = (uint32_t)(rng.draws - d0);
        if (A.draws && lane == 0) A.draws[ri] = (uint32_t)(rng.draws - d0);
    }
    if (lane == 0) A.streams[sidx].end_draw = rng.draws;
    __syncthreads();
    for (int i = lane; i < MT_N; i += LANES) state[i] = mt[i];
    if (lane == 0) state[MT_N] = (uint32_t)rng.mti;
    if (A.finalState && end >= st.n_rays) {
        uint32_t *dst = A.finalState + (size_t)sidx * (MT_N + 1);
        for (int i = lane; i < MT_N; i += LANES) dst[i] = mt[i];
        if (lane == 0) dst[MT_N] = (uint32_t)rng.mti;
    }
}

// Heavy pass of slice sliceK: one wave per ray, chunk c = 64 consecutive rays of one stream's slice.
template <bool STATS, int NREG>
__global__ __launch_bounds__(LANES, (NREG > 4 ? PVOL_WPE_BIG : PVOL_WPE)) void li_replay_kernel(LiArgs A) {
    extern __shared__ __align__(16) unsigned char lds[];
    const DevScene &S = *A.scene;
    const int lane = threadIdx.x;
    if (A.gated && *A.needSeq == 0u) return;   // backup of li_group_kernel's replay form: runs only if its hand-over list overflowed
    MarchLds M;
    M.G.cap = S.candCap;
    M.G.cd = reinterpret_cast<float *>(lds);
    M.G.ci = reinterpret_cast<uint32_t *>(lds + (size_t)M.G.cap * 4);
    M.lightNum = 0;
    M.prevRk = reinterpret_cast<float *>(lds + (size_t)M.G.cap * 8);
    M.G.paint = reinterpret_cast<uint32_t *>(M.prevRk + PREV_N);
    for (int i = lane; i < PREV_N; i += LANES) M.prevRk[i] = 0.f;
    __syncthreads();
    Rng rng;
    rng.mt = 0;
    rng.mti = 0;
    rng.draws = 0;
    WaveCounters wc = {};
    unsigned long long tk0 = STATS ? stamp() : 0ull;
    const bool grid = (S.volKind == PVOL_VOLUME_GRID);
    const uint32_t chunksPerSlice = (A.sliceM + CHUNK_RAYS - 1) / CHUNK_RAYS;
    const unsigned long long nChunks = (unsigned long long)chunksPerSlice * A.nStreams;
    for (;;) {
        uint32_t chunk = 0;
        if (lane == 0) chunk = atomicAdd(A.gated ? A.chunkCounter + 3 : A.chunkCounter, 1u);
        chunk = (uint32_t)lane_i((int)chunk, 0);
        if (chunk >= nChunks) break;
        const uint32_t sidx = chunk / chunksPerSlice, j = chunk - sidx * chunksPerSlice;
        const uint32_t nr = A.streams[sidx].n_rays, first = A.streams[sidx].first_ray;
        const uint32_t begin = A.sliceK * A.sliceM;
        if (begin >= nr) continue;
        const uint32_t sliceLen = min(nr - begin, A.sliceM);
        const uint32_t l0 = j * CHUNK_RAYS;
        if (l0 >= sliceLen) continue;
        const uint32_t l1 = min(sliceLen, l0 + CHUNK_RAYS);
        for (uint32_t l = l0; l < l1; ++l) {
            const size_t ri = (size_t)first + begin + l;
            const pvol_ray pr = A.rays[ri];
            RayRec rec = ray_rec(A.records + ((size_t)sidx * A.sliceM + l) * A.recStride, S.maxSteps, grid);
            f4 Lv, Tr;
            if (march_ray_blocked<STATS, MODE_REPLAY, NREG>(S, A, pr, rng, M, lane, wc, &Lv, &Tr, rec) == 2)
                march_ray<STATS, MODE_REPLAY, NREG>(S, A, pr, rng, M, lane, wc, &Lv, &Tr, rec);
            write_outputs(S, A, ri, Lv, Tr, lane);
        }
    }
    flush_counters<STATS>(A.counters, wc, tk0, lane);
}

// ------------------------------------------------------------------------------------------ geometry pre-pass + lite resolve
// When no march step can reach the Russian roulette (decided on the host from stepSize * max sigma_t * max density), the
// only things the RNG pre-pass needs from a ray's geometry are the step count n and, per step, which lights would be
// sampled unoccluded -- none of it depends on drawn values.  li_geo_kernel computes them ray-parallel (one wave per ray,
// one march step per lane) into the record slots; li_resolve_lite_kernel then walks each stream's rays in order touching
// only the MT19937 state: LDShuffleScrambled1D for lightNum, one draw per step, one per unoccluded sample (A.1).
// Geometry of one ray's march steps, one step per lane: which lights are unoccluded at every step (bit ln; 0x80 = sigma_s is not
// black there).  Shared by li_geo_kernel and the tile pre-pass of scenes with several lights (pvol_tile_dev.h).
__device__ int geo_ray(const DevScene &S, const LiArgs &A, const pvol_ray &pr, RayRec rec, int lane, bool grid, bool blackS1,
                       unsigned int lightBlackMask) {   // returns the ray's march-step count (0: nothing to do)
    RayD ray;
    ray.o = v3(pr.o[0], pr.o[1], pr.o[2]); ray.d = v3(pr.d[0], pr.d[1], pr.d[2]); ray.mint = pr.mint; ray.maxt = pr.maxt;
    float t0, t1;
    bool hit = S.volKind != PVOL_VOLUME_NONE && vol_intersect(S, ray, &t0, &t1) && (t1 - t0) != 0.f;
    int nSamples = hit ? (int)ceilf((t1 - t0) / S.stepSize) : 0;
    if (hit && nSamples > S.maxSteps) {   // the record plan cannot hold this ray: report, never guess
        if (lane == 0) atomicAdd(&A.counters->nErrors, 1ull);
        nSamples = 0;
    }
    if (lane == 0) { rec.hdr[0] = (uint32_t)nSamples; rec.hdr[1] = 0u; }
    if (nSamples == 0) return 0;
    const float step = (t1 - t0) / nSamples;
    float tcur = t0 + pr.scatter_u * step;
    for (int base = 0; base < nSamples; base += LANES) {
        const int cnt = min(LANES, nSamples - base);
        float tMine = 0.f;
        for (int j = 0; j < cnt; ++j) {   // t0 is ACCUMULATED in the reference: replay the additions
            if (lane == j) tMine = tcur;
            tcur += step;
        }
        const bool on = lane < cnt;
        const V3 p = ray.o + ray.d * tMine;
        const V3 pv = xform_point(S.w2v, p);
        const float dens = !on ? 0.f : (grid ? grid_density(S, pv) : (box_inside(S.extLo, S.extHi, pv) ? 1.f : 0.f));
        unsigned int mask = 0u;
        if (on && dens != 0.f && !blackS1 && S.nLights > 0) {   // sigma_s * dens is black iff dens == 0 or sigma_s is
            mask = 0x80u;
            for (int ln = 0; ln < S.nLights; ++ln) {
                const DevLight &light = S.lights[ln];
                RayD vis;
                float fall = 1.f;
                if (light.kind == PVOL_LIGHT_DISTANT) {
                    vis.o = p; vis.d = v3(light.dir[0], light.dir[1], light.dir[2]); vis.mint = 0.f; vis.maxt = INFINITY;
                } else {
                    V3 lp = v3(light.pos[0], light.pos[1], light.pos[2]);
                    V3 wo = normalize(lp - p);
                    float dist = len(p - lp);
                    vis.o = p; vis.d = vdiv(lp - p, dist); vis.mint = 0.f; vis.maxt = dist * (1.f - 0.f);
                    if (light.kind == PVOL_LIGHT_SPOT) {
                        V3 wl = normalize(v3(light.w2l[0] * -wo.x + light.w2l[1] * -wo.y + light.w2l[2] * -wo.z,
                                             light.w2l[4] * -wo.x + light.w2l[5] * -wo.y + light.w2l[6] * -wo.z,
                                             light.w2l[8] * -wo.x + light.w2l[9] * -wo.y + light.w2l[10] * -wo.z));
                        float costheta = wl.z;
                        if (costheta < light.cosTotalWidth) fall = 0.f;
                        else if (costheta > light.cosFalloffStart) fall = 1.f;
                        else {
                            float delta = (costheta - light.cosTotalWidth) / (light.cosFalloffStart - light.cosTotalWidth);
                            fall = delta * delta * delta * delta;
                        }
                    }
                }
                const bool black = (fall == 0.f) || ((lightBlackMask >> ln) & 1u);
                if (!black && !lane_occluded(S, vis)) mask |= 1u << ln;
            }
        }
        if (on) rec.stepByte[base + lane] = (unsigned char)mask;
    }
    return nSamples;
}

template <int NREG>
__global__ __launch_bounds__(LANES) void li_geo_kernel(LiArgs A) {
    const DevScene &S = *A.scene;
    const int lane = threadIdx.x;
    const int q = lane & 7;
    const bool grid = (S.volKind == PVOL_VOLUME_GRID);
    const f4 sigS = ld4(S.sigS, q);
    const bool blackS1 = spec_is_black(sigS);
    unsigned int lightBlackMask = 0u;
    for (int l = 0; l < S.nLights; ++l) if (spec_is_black(ld4(S.lights[l].intensity, q))) lightBlackMask |= 1u << l;
    const uint32_t chunksPerSlice = (A.sliceM + CHUNK_RAYS - 1) / CHUNK_RAYS;
    const unsigned long long nChunks = (unsigned long long)chunksPerSlice * A.nStreams;
    for (unsigned long long chunk = blockIdx.x; chunk < nChunks; chunk += gridDim.x) {
        const uint32_t sidx = (uint32_t)(chunk / chunksPerSlice), jc = (uint32_t)(chunk - (unsigned long long)sidx * chunksPerSlice);
        const uint32_t nr = A.streams[sidx].n_rays, first = A.streams[sidx].first_ray;
        const uint32_t begin = A.sliceK * A.sliceM;
        if (begin >= nr) continue;
        const uint32_t sliceLen = min(nr - begin, A.sliceM);
        const uint32_t l0 = jc * CHUNK_RAYS;
        if (l0 >= sliceLen) continue;
        const uint32_t l1 = min(sliceLen, l0 + CHUNK_RAYS);
        for (uint32_t l = l0; l < l1; ++l) {
            const pvol_ray pr = A.rays[(size_t)first + begin + l];
            RayRec rec = ray_rec(A.records + ((size_t)sidx * A.sliceM + l) * A.recStride, S.maxSteps, grid);
            geo_ray(S, A, pr, rec, lane, grid, blackS1, lightBlackMask);
        }
    }
}

// The RNG side of one ray's Li() when no drawn value reaches a decision beyond the light choice: LDShuffleScrambled1D for
// lightNum, one draw per step, one per unoccluded sample (A.1), 64 steps at a time.  Reads the masks geo_ray left, leaves the
// light of every step (and the drawn tau offsets for a VolumeGrid).  Shared by li_resolve_lite_kernel and the tile pre-pass.
__device__ void lite_ray(const DevScene &S, RayRec rec, int n, Rng &rng, float *lightNum, int lane, bool grid, int nLights, bool noSwaps = false) {
    if (n > 0) {
        // LDShuffleScrambled1D(1, n, lightNum) + (1, n, lightComp) + 2D(1, n, lightPos): 4 + 6n draws (photonvolume.cpp:137-142)
        if (nLights > 1) {
            const uint32_t scramble = rng_uint<true>(rng, lane);
            for (int i = lane; i < n; i += LANES) lightNum[i] = van_der_corput((uint32_t)i, scramble);
            rng_skip<true>(rng, (unsigned long long)n, lane);
            __syncthreads();
            // Shuffle(samples, n, 1), montecarlo.h:174-181: swap(samp[i], samp[i + RandomUInt() % (n - i)]) for i = 0 .. n-1.  The n draws
            // and their remainders do not depend on the swaps: 64 at a time, one per lane (draw j of a batch lands in lane j, the
            // stream order of the serial loop); only the swaps themselves stay a serial chain on lane 0.
            for (int base = 0; base < n; base += LANES) {
                const int cnt = min(LANES, n - base);
                const uint32_t y = rng_bulk(rng, cnt, lane);
                const int iMine = base + lane;
                const uint32_t oth = lane < cnt ? (uint32_t)iMine + (y % (uint32_t)(n - iMine)) : 0u;
                for (int j = 0; j < (noSwaps ? 0 : cnt); ++j) {   // noSwaps: timing knob 256 of PVOL_TILE_DEBUG (wrong results)
                    const uint32_t other = (uint32_t)__builtin_amdgcn_readlane((int)oth, j);
                    if (lane == 0) {
                        const float a = lightNum[base + j], b = lightNum[other];
                        lightNum[base + j] = b;
                        lightNum[other] = a;
                    }
                }
            }
            __syncthreads();
            rng_skip<true>(rng, 3ull + 4ull * (unsigned long long)n, lane);
        } else {
            rng_skip<true>(rng, 4ull + 6ull * (unsigned long long)n, lane);
        }
        // one march step per lane: step i draws uTau, then uSh if its light sample is unoccluded -- the positions are the
        // exclusive scan of (1 + shadowed) over the steps, the values come 64 at a time out of the LDS state
        for (int base = 0; base < n; base += LANES) {
            const int i = base + lane;
            const bool on = i < n;
            const unsigned int mask = on ? rec.stepByte[i] : 0u;
            int ln = 0;
            if (nLights > 1 && on) ln = min((int)floorf(lightNum[i] * nLights), nLights - 1);
            const bool lit = on && (mask & 0x80u) != 0u;
            const bool sh = lit && ((mask >> ln) & 1u) != 0u;
            const uint32_t need = on ? (sh ? 2u : 1u) : 0u;
            uint32_t incl = need;
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, true);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, true);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, true);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, true);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xa, 0xf, false);
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xc, 0xf, false);
            const uint32_t off = incl - need;
            const int total = lane_i((int)incl, LANES - 1);   // <= 128
            const uint32_t dA = rng_bulk(rng, min(total, LANES), lane);
            const uint32_t dB = total > LANES ? rng_bulk(rng, total - LANES, lane) : 0u;
            const uint32_t o1 = off + 1u;
            const uint32_t t0 = (uint32_t)__shfl((int)dA, (int)(off & 63u)), t1 = (uint32_t)__shfl((int)dB, (int)(off & 63u));
            const uint32_t s0 = (uint32_t)__shfl((int)dA, (int)(o1 & 63u)), s1 = (uint32_t)__shfl((int)dB, (int)(o1 & 63u));
            const float uTau = ((off < 64u ? t0 : t1) & 0xffffff) / float(1 << 24);   // core/rng.cpp:59-65
            const float uSh = sh ? ((o1 < 64u ? s0 : s1) & 0xffffff) / float(1 << 24) : 0.f;
            if (on) {
                rec.stepByte[i] = (unsigned char)(lit ? (unsigned int)ln : 0u);
                if (grid) { rec.stepU[2 * i] = uTau; rec.stepU[2 * i + 1] = uSh; }
            }
        }
    }
}

template <int NREG>
__global__ __launch_bounds__(LANES) void li_resolve_lite_kernel(LiArgs A) {
    extern __shared__ __align__(16) unsigned char lds[];
    const DevScene &S = *A.scene;
    const int lane = threadIdx.x;
    const uint32_t sidx = blockIdx.x;
    if (sidx >= A.nStreams) return;
    uint32_t *mt = reinterpret_cast<uint32_t *>(lds);
    float *lightNum = reinterpret_cast<float *>(lds + MT_N * 4);
    pvol_stream st = A.streams[sidx];
    const uint32_t begin = A.sliceK * A.sliceM;
    if (begin >= st.n_rays && !(A.sliceK == 0)) return;
    Rng rng;
    rng.mt = mt;
    rng.draws = 0;
    uint32_t *state = A.state + (size_t)sidx * (MT_N + 1);
    if (A.sliceK == 0 && !A.initState) {
        mt_seed(mt, st.seed, lane);
        rng.mti = MT_N;
        rng_skip<true>(rng, st.start_draw, lane);
    } else {
        const uint32_t *src = (A.sliceK == 0) ? A.initState + (size_t)sidx * (MT_N + 1) : state;
        for (int i = lane; i < MT_N; i += LANES) mt[i] = src[i];
        rng.mti = (int)src[MT_N];
        rng.draws = (A.sliceK == 0) ? st.start_draw : st.end_draw;
        __syncthreads();
    }
    const bool grid = (S.volKind == PVOL_VOLUME_GRID);
    const int nLights = S.nLights;
    const uint32_t end = min(st.n_rays, begin + A.sliceM);
    for (uint32_t k = begin; k < end; ++k) {
        const size_t ri = (size_t)st.first_ray + k;
        rng_skip<true>(rng, A.rays[ri].rng_skip, lane);
        const unsigned long long d0 = rng.draws;
        RayRec rec = ray_rec(A.records + ((size_t)sidx * A.sliceM + (k - begin)) * A.recStride, S.maxSteps, grid);
        lite_ray(S, rec, (int)rec.hdr[0], rng, lightNum, lane, grid, nLights);
        if (lane == 0) rec.hdr[2] = (uint32_t)(rng.draws - d0);
        if (A.draws && lane == 0) A.draws[ri] = (uint32_t)(rng.draws - d0);
    }
    if (lane == 0) A.streams[sidx].end_draw = rng.draws;
    __syncthreads();
    for (int i = lane; i < MT_N; i += LANES) state[i] = mt[i];
    if (lane == 0) state[MT_N] = (uint32_t)rng.mti;
    if (A.finalState && end >= st.n_rays) {
        uint32_t *dst = A.finalState + (size_t)sidx * (MT_N + 1);
        for (int i = lane; i < MT_N; i += LANES) dst[i] = mt[i];
        if (lane == 0) dst[MT_N] = (uint32_t)rng.mti;
    }
}

extern "C" hipError_t pvol_launch_li_seq(const LiArgs *args, size_t ldsBytes, int candCap, bool stats, hipStream_t stream) {
    dim3 grid(args->nStreams), block(LANES);
    // NREG = candidate registers per lane in select_k: 4 covers nused <= 64, 12 covers nused <= 576
    if (candCap <= 4 * LANES) {
        if (stats) hipLaunchKernelGGL((li_seq_kernel<true, 4>), grid, block, ldsBytes, stream, *args);
        else hipLaunchKernelGGL((li_seq_kernel<false, 4>), grid, block, ldsBytes, stream, *args);
    } else {
        if (stats) hipLaunchKernelGGL((li_seq_kernel<true, 12>), grid, block, ldsBytes, stream, *args);
        else hipLaunchKernelGGL((li_seq_kernel<false, 12>), grid, block, ldsBytes, stream, *args);
    }
    return hipGetLastError();
}

extern "C" hipError_t pvol_launch_li_par(const LiArgs *args, size_t ldsBytes, int candCap, bool stats, uint32_t nWaves, hipStream_t stream) {
    hipLaunchKernelGGL(stream_begin_kernel, dim3((args->nStreams + 255) / 256), dim3(256), 0, stream, args->streams, args->nStreams);
    dim3 grid(nWaves), block(LANES);
    if (candCap <= 4 * LANES) {
        if (stats) hipLaunchKernelGGL((li_par_kernel<true, 4>), grid, block, ldsBytes, stream, *args);
        else hipLaunchKernelGGL((li_par_kernel<false, 4>), grid, block, ldsBytes, stream, *args);
    } else {
        if (stats) hipLaunchKernelGGL((li_par_kernel<true, 12>), grid, block, ldsBytes, stream, *args);
        else hipLaunchKernelGGL((li_par_kernel<false, 12>), grid, block, ldsBytes, stream, *args);
    }
    return hipGetLastError();
}

// one slice: resolve (wave per stream) then replay; chunkCounter must be zero before the replay.
// groupForm: 0 = li_replay_kernel (one wave per ray), 1 / 2 = li_group_kernel's replay form (homogeneous / VolumeGrid: one ray per lane)
// followed by its exact-lookup pass and, gated on a hand-over list overflow, li_replay_kernel as the backup
extern "C" hipError_t pvol_launch_li_group(const LiArgs *args, size_t ldsBytes, int candCap, bool stats, uint32_t nWaves, uint32_t nFixWaves,
                                           int replay, hipStream_t stream);
extern "C" hipError_t pvol_launch_li_slice(const LiArgs *args, size_t ldsResolve, size_t ldsReplay, int candCap, bool stats,
                                           uint32_t nWaves, hipStream_t stream, bool resolve, int groupForm, size_t ldsGroup, uint32_t nGroupWaves,
                                           uint32_t nFixWaves) {
    dim3 block(LANES);
    if (resolve && args->liteResolve) {   // geometry pre-pass (ray-parallel) + RNG-only sequential pass
        hipLaunchKernelGGL((li_geo_kernel<4>), dim3(nWaves), block, 0, stream, *args);
        hipLaunchKernelGGL((li_resolve_lite_kernel<4>), dim3(args->nStreams), block, ldsResolve, stream, *args);
        resolve = false;
    }
    if (resolve) {
        if (candCap <= 4 * LANES) hipLaunchKernelGGL((li_resolve_kernel<false, 4>), dim3(args->nStreams), block, ldsResolve, stream, *args);
        else hipLaunchKernelGGL((li_resolve_kernel<false, 12>), dim3(args->nStreams), block, ldsResolve, stream, *args);
    }
    LiArgs a2 = *args;
    if (groupForm) {
        hipError_t e = pvol_launch_li_group(args, ldsGroup, candCap, false, nGroupWaves, nFixWaves, groupForm, stream);
        if (e != hipSuccess) return e;
        a2.gated = 1;
    }
    if (candCap <= 4 * LANES) {
        if (stats) hipLaunchKernelGGL((li_replay_kernel<true, 4>), dim3(nWaves), block, ldsReplay, stream, a2);
        else hipLaunchKernelGGL((li_replay_kernel<false, 4>), dim3(nWaves), block, ldsReplay, stream, a2);
    } else {
        if (stats) hipLaunchKernelGGL((li_replay_kernel<true, 12>), dim3(nWaves), block, ldsReplay, stream, a2);
        else hipLaunchKernelGGL((li_replay_kernel<false, 12>), dim3(nWaves), block, ldsReplay, stream, a2);
    }
    return hipGetLastError();
}

// li_replay_kernel on its own (the segment pool of the specular recursion: records written by the tile pre-pass, one wave per ray)
extern "C" hipError_t pvol_launch_li_replay(const LiArgs *args, size_t ldsReplay, int candCap, uint32_t nWaves, hipStream_t stream) {
    LiArgs a = *args;
    a.gated = 0;
    if (candCap <= 4 * LANES) hipLaunchKernelGGL((li_replay_kernel<false, 4>), dim3(nWaves), dim3(LANES), ldsReplay, stream, a);
    else hipLaunchKernelGGL((li_replay_kernel<false, 12>), dim3(nWaves), dim3(LANES), ldsReplay, stream, a);
    return hipGetLastError();
}

#include "pvol_group_dev.h"
#include "pvol_surface_dev.h"
#include "pvol_spec_dev.h"
#include "pvol_tile_dev.h"
