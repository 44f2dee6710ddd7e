// pvol_tile_dev.h -- included at the end of pvol_march.hip (needs march_ray<MODE_RESOLVE> and the LDS MT19937).
// The sequential half of a SamplerRendererTask (renderers/samplerrenderer.cpp:59-157): one wave per render
// task walks the task's pixels in LDSampler order (samplers/lowdiscrepancy.cpp:69-80), draws LDPixelSample
// (core/montecarlo.cpp:200-254) from the task's MT19937 stream, turns the samples into camera rays
// (cameras/perspective.cpp:80-134), clips them at the closest surface (Scene::Intersect) and finds out how many
// draws each Li() call will make, because the NEXT pixel's samples depend on where the stream then stands:
//   COUNT  (scenes where no drawn value reaches Li's result and the roulette cannot fire): one camera sample per
//          lane counts 4 + 6n + n + u; the heavy kernel is li_par_kernel
//   FUSED  (everything else): every ray is walked by march_ray<MODE_RESOLVE>, which draws for real and leaves the
//          per-step records li_replay_kernel reads -- this kernel then stands in for li_resolve_kernel
// The rays it writes carry rng_skip = the sampler's draws in front of the pixel's first sample, so the batch is
// also a valid input of every other Li kernel.

// core/montecarlo.h:289-293
__device__ __forceinline__ float sobol2(uint32_t n, uint32_t scramble) {
    for (uint32_t v = 1u << 31; n != 0; n >>= 1, v ^= v >> 1)
        if (n & 0x1) scramble ^= v;
    return fminf(((scramble >> 8) & 0xffffff) / float(1 << 24), 0x1.fffffep-1f);
}

// Shuffle(samp, count, dims, rng), core/montecarlo.h:174-181: the `other` indices are drawn 64 at a time, the
// swaps themselves are order-dependent and run on lane 0
template <bool WG1 = true>
__device__ void tile_shuffle(float *samp, uint32_t count, uint32_t dims, uint32_t *oth, Rng &rng, int lane, bool skipSwaps = false) {
    for (uint32_t base = 0; base < count; base += LANES) {
        const int cnt = (int)min((uint32_t)LANES, count - base);
        const uint32_t r = rng_bulk<WG1>(rng, cnt, lane);
        const uint32_t i = base + (uint32_t)lane;
        if (lane < cnt) oth[i] = i + (r % (count - i));
    }
    rng_sync<WG1>();
    if (lane == 0 && !skipSwaps) {   // the swaps depend on each other through the array: one lane, next index fetched a step ahead
        uint32_t oNext = oth[0];
        for (uint32_t i = 0; i < count; ++i) {
            const uint32_t o = oNext;
            if (i + 1 < count) oNext = oth[i + 1];
            if (dims == 2) {
                float2 *s2 = reinterpret_cast<float2 *>(samp);
                const float2 a = s2[i], b = s2[o];
                s2[i] = b;
                s2[o] = a;
            } else {
                for (uint32_t j = 0; j < dims; ++j) {
                    float a = samp[dims * i + j], b = samp[dims * o + j];
                    samp[dims * i + j] = b;
                    samp[dims * o + j] = a;
                }
            }
        }
    }
    rng_sync<WG1>();
}

struct TileLds {
    float *image;    // 2 * spp
    float *time;     // spp
    float *scatter;  // spp
    uint32_t *oth;   // spp
};

// LDPixelSample, core/montecarlo.cpp:200-254.  Arrays nobody on this path reads (lens, tau and other
// integrators' requests) only advance the stream by what generating them draws.
template <bool WG1 = true>
__device__ void tile_pixel_sample(const TileArgs &T, TileLds &L, Rng &rng, int lane) {
    const uint32_t n = T.spp;
    // image: LDShuffleScrambled2D(1, n): 2 scrambles, n no-op Shuffles of one element (one draw each), Shuffle(n, 2)
    {
        const uint32_t s0 = rng_uint<true, WG1>(rng, lane), s1 = rng_uint<true, WG1>(rng, lane);
        for (uint32_t i = lane; i < n; i += LANES) { L.image[2 * i] = van_der_corput(i, s0); L.image[2 * i + 1] = sobol2(i, s1); }
        rng_skip<true, WG1>(rng, n, lane);
        rng_sync<WG1>();
        tile_shuffle<WG1>(L.image, n, 2, L.oth, rng, lane, (T.debugSkip & 1u) != 0u);
    }
    rng_skip<true, WG1>(rng, 2ull + 2ull * n, lane);   // lens
    {
        const uint32_t s = rng_uint<true, WG1>(rng, lane);
        for (uint32_t i = lane; i < n; i += LANES) L.time[i] = van_der_corput(i, s);
        rng_skip<true, WG1>(rng, n, lane);
        rng_sync<WG1>();
        tile_shuffle<WG1>(L.time, n, 1, L.oth, rng, lane, (T.debugSkip & 1u) != 0u);
    }
    // the arrays nobody reads (light / BSDF / gather samples of the surface integrator) are drawn, not kept: their draws are skipped,
    // whole runs of them at once -- thousands per pixel with a final gather's 2 x 32 samples -- through the two-round-trip regeneration
    unsigned long long pend = 0ull;
    for (uint32_t a = 0; a < T.n1dCount; ++a) {
        if (a == T.scatterIndex) {   // n1d == 1 (checked on the host)
            if (pend) { rng_skip<true, WG1, true>(rng, pend, lane); pend = 0ull; }
            const uint32_t s = rng_uint<true, WG1>(rng, lane);
            for (uint32_t i = lane; i < n; i += LANES) L.scatter[i] = van_der_corput(i, s);
            rng_skip<true, WG1>(rng, n, lane);
            rng_sync<WG1>();
            tile_shuffle<WG1>(L.scatter, n, 1, L.oth, rng, lane, (T.debugSkip & 1u) != 0u);
        } else {
            pend += 1ull + (unsigned long long)T.n1d[a] * n + n;
        }
    }
    for (uint32_t a = 0; a < T.n2dCount; ++a) pend += 2ull + (unsigned long long)T.n2d[a] * n + n;
    if (pend) rng_skip<true, WG1, true>(rng, pend, lane);
}

// PerspectiveCamera::GenerateRayDifferential without a lens + CameraToWorld (static transform)
__device__ __forceinline__ void tile_camera_ray(const TileArgs &T, float imageX, float imageY, V3 *o, V3 *d) {
    const float *m = T.r2c;
    V3 pc;
    pc.x = m[0] * imageX + m[1] * imageY + m[2] * 0.f + m[3];
    pc.y = m[4] * imageX + m[5] * imageY + m[6] * 0.f + m[7];
    pc.z = m[8] * imageX + m[9] * imageY + m[10] * 0.f + m[11];
    const float w = m[12] * imageX + m[13] * imageY + m[14] * 0.f + m[15];
    if (w != 1.f) { const float inv = 1.f / w; pc.x *= inv; pc.y *= inv; pc.z *= inv; }
    const V3 dir = normalize(pc);
    const float *c = T.c2w;
    V3 ow = v3(c[0] * 0.f + c[1] * 0.f + c[2] * 0.f + c[3], c[4] * 0.f + c[5] * 0.f + c[6] * 0.f + c[7],
               c[8] * 0.f + c[9] * 0.f + c[10] * 0.f + c[11]);
    const float ww = c[12] * 0.f + c[13] * 0.f + c[14] * 0.f + c[15];
    if (ww != 1.f) { const float inv = 1.f / ww; ow.x *= inv; ow.y *= inv; ow.z *= inv; }
    *o = ow;
    *d = xform_vector(c, dir);
}

// Scene::Intersect as SamplerRenderer::Li uses it (samplerrenderer.cpp:236-249): only the clipped maxt matters here
__device__ __forceinline__ float tile_clip(const float *ltri, int nTris, V3 o, V3 d) {
    float mt = INFINITY;
    for (int i = 0; i < nTris; ++i) {
        const f4 a = *reinterpret_cast<const f4 *>(ltri + 12 * i), b = *reinterpret_cast<const f4 *>(ltri + 12 * i + 4),
                 c = *reinterpret_cast<const f4 *>(ltri + 12 * i + 8);
        float t;
        if (tri_closest_v(v3(a.x, a.y, a.z), v3(a.w, b.x, b.y), v3(b.z, b.w, c.x), o, d, 0.f, mt, &t)) mt = t;
    }
    return mt;
}

// What the draw count reads of the scene, taken into registers once per kernel: inside the per-step loop a reference to
// the DevScene in global memory would be re-read every iteration (the loop's control flow keeps the compiler from hoisting
// the loads), and that latency dominated the pre-pass.
struct CountConsts {
    float w2v[16];
    float lo[3], hi[3];
    float stepSize;
    int volKind, nLights, nTris, lightKind;
    bool bvh;
    int nSpheres;
    float ldir[3], lpos[3], w2l[9], cosTotalWidth, cosFalloffStart;
};
__device__ __forceinline__ CountConsts count_consts(const DevScene &S) {
    CountConsts C;
    for (int i = 0; i < 16; ++i) C.w2v[i] = S.w2v[i];
    for (int i = 0; i < 3; ++i) { C.lo[i] = S.extLo[i]; C.hi[i] = S.extHi[i]; }
    C.stepSize = S.stepSize; C.volKind = S.volKind; C.nLights = S.nLights; C.nTris = S.nTris; C.bvh = S.bvhNodes != 0; C.nSpheres = S.nSpheres;
    const DevLight &light = S.lights[0];
    C.lightKind = light.kind;
    for (int i = 0; i < 3; ++i) { C.ldir[i] = light.dir[i]; C.lpos[i] = light.pos[i]; }
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) C.w2l[3 * r + c] = light.w2l[4 * r + c];
    C.cosTotalWidth = light.cosTotalWidth; C.cosFalloffStart = light.cosFalloffStart;
    return C;
}

// Number of RandomUInt calls of one Li() when no roulette can fire and at most one light exists: 4 + 6n + n + u
// (photonvolume.cpp:112-222; same per-step tests as march_ray_blocked's scalar phase).  One ray per lane.
// `slice` of `nSlices` (multi-wave pre-pass): the march steps are dealt round-robin to the waves that share a sample group; every
// wave still walks tcur through ALL the additions (the reference accumulates t0 step by step), tests only its own steps, and
// slice 0 alone adds the 4 + 7n that do not depend on a test.
__device__ uint32_t tile_count_draws(const DevScene &S, const CountConsts &C, const float *ltri, const float *trows, V3 o, V3 d, float maxt, float scatterU, bool blackS, bool lightBlack, uint32_t dbg = 0u,
                                     int slice = 0, int nSlices = 1, float mint = 0.f) {
    float t0, t1;
    // vol_intersect: BBox::IntersectP of the ray taken to volume space (core/geometry.cpp:68-86); mint > 0 for a spawned ray
    if (C.volKind == PVOL_VOLUME_NONE || !box_intersect(C.lo, C.hi, xform_point(C.w2v, o), xform_vector(C.w2v, d), mint, maxt, &t0, &t1) || (t1 - t0) == 0.f) return 0u;
    const int nSamples = (int)ceilf((t1 - t0) / C.stepSize);
    const float step = (t1 - t0) / nSamples;
    float tcur = t0 + scatterU * step;
    uint32_t u = 0;
    const bool tryLight = !blackS && C.nLights > 0 && !lightBlack;
    const uint32_t fixedDraws = slice == 0 ? 4u + 7u * (uint32_t)nSamples : 0u;
    if (!tryLight || (dbg & 16u)) return fixedDraws;
    if (trows && !C.bvh && !C.nSpheres && !(dbg & 8u)) {
        // A distant light over the precomputed triangle rows: two march steps per trip (written out: the same loop over
        // small arrays compiled to slower code, and four steps per trip measured slower than one -- registers) -- every row is read from LDS once for
        // both (the shadow rays share their direction), and the two steps' arithmetic overlaps the reads' latency.  tcur takes
        // the same additions in the same order; the count is an integer sum.
        const V3 ld = v3(C.ldir[0], C.ldir[1], C.ldir[2]);
        for (int j = 0; j < nSamples; j += 2) {
            const V3 pA = o + d * tcur;
            tcur += step;
            const V3 pB = o + d * tcur;
            tcur += step;
            if (nSlices > 1 && ((j >> 1) % nSlices) != slice) continue;
            const V3 pvA = xform_point(C.w2v, pA), pvB = xform_point(C.w2v, pB);
            bool inA, inB = j + 1 < nSamples;
            if (C.volKind == PVOL_VOLUME_GRID) { inA = grid_density(S, pvA) != 0.f; inB = inB && grid_density(S, pvB) != 0.f; }
            else { inA = box_inside(C.lo, C.hi, pvA); inB = inB && box_inside(C.lo, C.hi, pvB); }
            if (!inA && !inB) continue;
            bool occA = false, occB = false;
            for (int t = 0; t < C.nTris; ++t) {
                const f4 r0 = *reinterpret_cast<const f4 *>(trows + 16 * t), r1 = *reinterpret_cast<const f4 *>(trows + 16 * t + 4),
                         r2 = *reinterpret_cast<const f4 *>(trows + 16 * t + 8), r3 = *reinterpret_cast<const f4 *>(trows + 16 * t + 12);
                TriPre tp;
                tp.p1 = v3(r0.x, r0.y, r0.z); tp.e1 = v3(r0.w, r1.x, r1.y); tp.e2 = v3(r1.z, r1.w, r2.x); tp.s1 = v3(r2.y, r2.z, r2.w);
                tp.invDivisor = r3.x; tp.valid = r3.y != 0.f;
                occA = occA | tri_test(tp, pA, ld, 0.f, INFINITY);
                occB = occB | tri_test(tp, pB, ld, 0.f, INFINITY);
            }
            u += ((inA && !occA) ? 1u : 0u) + ((inB && !occB) ? 1u : 0u);
        }
        return fixedDraws + u;
    }
    for (int j = 0; j < nSamples; ++j) {
        const V3 p = o + d * tcur;
        tcur += step;
        if (nSlices > 1 && (j % nSlices) != slice) continue;
        const V3 pv = xform_point(C.w2v, p);
        // sigma_s(p) black: outside a homogeneous extent, or zero density of a VolumeGrid (volumegrid.cpp:39-57)
        if (C.volKind == PVOL_VOLUME_GRID ? grid_density(S, pv) == 0.f : !box_inside(C.lo, C.hi, pv)) continue;
        RayD vis;
        if (C.lightKind == PVOL_LIGHT_DISTANT) {
            vis.o = p; vis.d = v3(C.ldir[0], C.ldir[1], C.ldir[2]); vis.mint = 0.f; vis.maxt = INFINITY;
        } else {
            const V3 lp = v3(C.lpos[0], C.lpos[1], C.lpos[2]);
            const V3 wo = normalize(lp - p);
            const float dist = len(p - lp);
            vis.o = p; vis.d = vdiv(lp - p, dist); vis.mint = 0.f; vis.maxt = dist * (1.f - 0.f);
            if (C.lightKind == PVOL_LIGHT_SPOT) {
                V3 wl = normalize(v3(C.w2l[0] * -wo.x + C.w2l[1] * -wo.y + C.w2l[2] * -wo.z,
                                     C.w2l[3] * -wo.x + C.w2l[4] * -wo.y + C.w2l[5] * -wo.z,
                                     C.w2l[6] * -wo.x + C.w2l[7] * -wo.y + C.w2l[8] * -wo.z));
                const float costheta = wl.z;   // SpotLight::Falloff, spot.cpp:60-69; 0 means L.IsBlack()
                float fall = 1.f;
                if (costheta < C.cosTotalWidth) fall = 0.f;
                else if (!(costheta > C.cosFalloffStart)) {
                    const float delta = (costheta - C.cosTotalWidth) / (C.cosFalloffStart - C.cosTotalWidth);
                    fall = delta * delta * delta * delta;
                }
                if (fall == 0.f) continue;
            }
        }
        // Scene::IntersectP over the world triangles, staged in LDS (12 floats each): every lane reads the same words
        // (broadcast), no early exit, so the loads of all triangles are in flight together
        bool occ = false;
        if (C.nSpheres && spheres_occluded(S, vis.o, vis.d, vis.mint, vis.maxt)) {
            occ = true;
        } else if (C.bvh) {   // more triangles than LDS rows: the device-built hierarchy (pvol_bvh_dev.h)
            occ = bvh_occluded(S, vis.o, vis.d, vis.mint, vis.maxt);
        } else if (trows) {   // distant light: direction-only terms precomputed per triangle
            if (!(dbg & 8u)) occ = tri_rows_occluded(trows, C.nTris, vis.o, vis.d, vis.mint, vis.maxt);
        } else {
            for (int t = 0; t < ((dbg & 8u) ? 0 : C.nTris); ++t) {
                const f4 a = *reinterpret_cast<const f4 *>(ltri + 12 * t), b = *reinterpret_cast<const f4 *>(ltri + 12 * t + 4),
                         c = *reinterpret_cast<const f4 *>(ltri + 12 * t + 8);
                occ = occ | tri_hit_v(v3(a.x, a.y, a.z), v3(a.w, b.x, b.y), v3(b.z, b.w, c.x), vis);
            }
        }
        if (!occ) ++u;
    }
    return fixedDraws + u;
}

// ---- specular recursion in the pre-pass (pvol_spec_dev.h) ------------------------------------------------------------------
// COUNT mode, one camera sample per lane.  Pass 1 adds up what the tree draws (every segment's own volume Li() included: the
// stream only needs the total), pass 2 lays the segments out in the pool.
struct SpecCountPol {
    const DevScene *S; const CountConsts *C; const float *ltri, *trows; float su; bool blackS, lightBlack;
    uint32_t total;
    __device__ void visit(SpecCtx &X, int, int, int, float, float, float, V3 o, V3 d, float mint, float maxt) {
        total += X.pending + tile_count_draws(*S, *C, ltri, trows, o, d, maxt, su, blackS, lightBlack, 0u, 0, 1, mint);
        X.pending = 0u;
    }
};
struct SpecEmitPol {
    pvol_ray *rays; SegInfo *info; uint32_t base, sample; float su, time;
    __device__ void visit(SpecCtx &X, int D, int lobe, int mat, float Fs, float awz, float g, V3 o, V3 d, float mint, float maxt) {
        const uint32_t idx = base + X.nSeg;
        pvol_ray r;
        r.o[0] = o.x; r.o[1] = o.y; r.o[2] = o.z; r.mint = mint;
        r.d[0] = d.x; r.d[1] = d.y; r.d[2] = d.z; r.maxt = maxt;
        r.time = time; r.scatter_u = su; r.rng_skip = X.pending; r.flags = 0u;
        rays[idx] = r;
        SegInfo si;
        si.sample = sample; si.depthLobeMat = (uint32_t)D | ((uint32_t)lobe << 8) | ((uint32_t)mat << 16);
        si.Fs = Fs; si.awz = awz; si.g = g; si.pad[0] = si.pad[1] = si.pad[2] = 0u;
        info[idx] = si;
        X.pending = 0u;
    }
};
// What PhotonIntegrator::Li draws for a camera sample whose ray hit `sh` (in front of the sample's own volume Li()), the
// segments of a specular hit laid out on the way.  Returns the draw count; *link = (first segment << 6) | count.
template <bool SPEC>
__device__ uint32_t tile_surface_draws(const DevScene &S, const TileArgs &T, const CountConsts &CC, const float *ltri, const float *trows, const SurfHit &sh,
                                       V3 d, float su, float tm, size_t ri, bool blackS, bool lightBlack, unsigned blackMask, DevCounters *counters, uint32_t *link) {
    *link = 0u;
    if (!SPEC || S.shootScene->mats[sh.mat].kind == PVOL_MATERIAL_MATTE || !T.specOn) return surf_count_draws(S, sh, d, blackMask);
    if constexpr (SPEC) {
    SpecCtx X;
    X.blackMask = blackMask; X.pending = 0u; X.nSeg = 0u;
    SpecCountPol cp;
    cp.S = &S; cp.C = &CC; cp.ltri = ltri; cp.trows = trows; cp.su = su; cp.blackS = blackS; cp.lightBlack = lightBlack; cp.total = 0u;
    spec_surface<0>(S, X, cp, d, sh);
    const uint32_t draws = cp.total + X.pending, n = X.nSeg;
    if (n) {
        const uint32_t base = atomicAdd(T.segCounter, n);
        if (base + n <= T.segCap) {
            SpecEmitPol ep;
            ep.rays = T.segRays; ep.info = T.segInfo; ep.base = base; ep.sample = (uint32_t)ri; ep.su = su; ep.time = tm;
            X.pending = 0u; X.nSeg = 0u;
            spec_surface<0>(S, X, ep, d, sh);
            *link = (base << SPEC_LINK_COUNT_BITS) | n;
        } else {
            atomicAdd(&counters->nErrors, 1ull);   // the segment pool is exhausted: reported, never dropped silently
        }
    }
    return draws;
    }
    return 0u;
}

// FUSED mode (several lights: drawn values choose the light of a march step), the whole wave on one camera sample.  Pass 1 only
// counts the segments; pass 2 walks every segment's stream for real -- the surface integrator's draws in front of it are skipped,
// its march geometry goes to its record (geo_ray), its Li() is drawn (lite_ray) -- in the order the reference draws them.
struct SpecNullPol {
    __device__ void visit(SpecCtx &X, int, int, int, float, float, float, V3, V3, float, float) { X.pending = 0u; }
};
struct SpecWalkPol {
    const DevScene *S; const LiArgs *A; const TileArgs *T; Rng *rng; float *lightNum;
    uint32_t base, sample; float su, time; int lane; bool grid, blackS; unsigned blackMask;
    __device__ void visit(SpecCtx &X, int D, int lobe, int mat, float Fs, float awz, float g, V3 o, V3 d, float mint, float maxt) {
        rng_skip<true>(*rng, X.pending, lane);
        const uint32_t idx = base + X.nSeg;
        pvol_ray r;
        r.o[0] = o.x; r.o[1] = o.y; r.o[2] = o.z; r.mint = mint;
        r.d[0] = d.x; r.d[1] = d.y; r.d[2] = d.z; r.maxt = maxt;
        r.time = time; r.scatter_u = su; r.rng_skip = X.pending; r.flags = 0u;
        if (lane == 0) {
            T->segRays[idx] = r;
            SegInfo si;
            si.sample = sample; si.depthLobeMat = (uint32_t)D | ((uint32_t)lobe << 8) | ((uint32_t)mat << 16);
            si.Fs = Fs; si.awz = awz; si.g = g; si.pad[0] = si.pad[1] = si.pad[2] = 0u;
            T->segInfo[idx] = si;
        }
        X.pending = 0u;
        RayRec rec = ray_rec(T->segRecords + (size_t)idx * A->recStride, S->maxSteps, grid);
        const unsigned long long r0 = rng->draws;
        const int nSteps = geo_ray(*S, *A, r, rec, lane, grid, blackS, blackMask);
        lite_ray(*S, rec, nSteps, *rng, lightNum, lane, grid, S->nLights);
        if (lane == 0) rec.hdr[2] = (uint32_t)(rng->draws - r0);
    }
};

// SPEC: the scene holds a specular material and the surface integrator is on (the recursion of pvol_spec_dev.h is compiled in;
// kept out of the other instantiations, whose register allocation it would otherwise weigh down)
template <bool FUSED, int NREG, bool SPEC>
__global__ __launch_bounds__(LANES, FUSED ? 2 : 4) void tile_kernel(LiArgs A, TileArgs T) {   // FUSED: 256 VGPRs (it wants 385: one wave per SIMD, four tasks per CU at a time)
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
    TileLds L;
    L.image = M.lightNum + (FUSED ? ((S.maxSteps + 1) & ~1) : 0);   // 8-byte aligned: the image samples are swapped as float2
    L.time = L.image + 2 * T.spp;
    L.scatter = L.time + T.spp;
    L.oth = reinterpret_cast<uint32_t *>(L.scatter + T.spp);
    const size_t triOff = (((size_t)MT_N * 4 + (FUSED ? (size_t)((S.maxSteps + 1) & ~1) * 4 : 0) + (size_t)T.spp * 5 * 4) + 15) & ~(size_t)15;
    float *ltri = reinterpret_cast<float *>(lds + triOff);   // 16-byte aligned: read as float4
    for (int i = lane; i < S.nTris * 12; i += LANES) {
        const int t = i / 12, c = i - 12 * t;
        const DevTri &tr = S.tris[t];
        ltri[i] = c < 3 ? tr.p1[c] : c < 6 ? tr.p2[c - 3] : c < 9 ? tr.p3[c - 6] : 0.f;
    }
    __syncthreads();
    float *trows = 0;
    if (!FUSED && S.nLights > 0 && S.lights[0].kind == PVOL_LIGHT_DISTANT) {
        trows = ltri + S.nTris * 12;
        tri_rows_prepare(S, v3(S.lights[0].dir[0], S.lights[0].dir[1], S.lights[0].dir[2]), trows, lane);
    }
    const pvol_stream st = A.streams[sidx];
    const uint32_t begin = A.sliceK * A.sliceM;
    if (begin >= st.n_rays && !(A.sliceK == 0)) return;
    Rng rng;
    rng.mt = mt;
    rng.draws = 0;
    uint32_t *state = A.state ? A.state + (size_t)sidx * (MT_N + 1) : 0;
    if (A.sliceK == 0) {
        mt_seed(mt, st.seed, lane);
        rng.mti = MT_N;
        rng_skip<true>(rng, st.start_draw, lane);
    } else {
        for (int i = lane; i < MT_N; i += LANES) mt[i] = state[i];
        rng.mti = (int)state[MT_N];
        rng.draws = st.end_draw;
        __syncthreads();
    }
    WaveCounters wc = {};
    const bool grid = (S.volKind == PVOL_VOLUME_GRID);
    const uint32_t end = min(st.n_rays, begin + A.sliceM);
    const int4 w = T.windows[sidx];
    const uint32_t width = (uint32_t)(w.y - w.x);
    const int q = lane & 7;
    const bool blackS = spec_is_black(ld4(S.sigS, q));
    const bool lightBlack = S.nLights > 0 ? spec_is_black(ld4(S.lights[0].intensity, q)) : true;
    unsigned blackMask = 0u;   // per light, for the surface integrator's direct lighting
    for (int l = 0; l < S.nLights; ++l) if (spec_is_black(ld4(S.lights[l].intensity, q))) blackMask |= 1u << l;
    const bool surfOn = S.surf.enabled != 0;
    pvol_ray *rays = T.rays;
    const CountConsts CC = count_consts(S);
    for (uint32_t k = begin; k < end; k += T.spp) {
        const uint32_t pix = k / T.spp;
        const int xPos = w.x + (int)(pix % width), yPos = w.z + (int)(pix / width);
        const unsigned long long d0 = rng.draws;
        tile_pixel_sample(T, L, rng, lane);
        const uint32_t samplerDraws = (uint32_t)(rng.draws - d0);
        for (uint32_t s0 = 0; s0 < T.spp; s0 += LANES) {
            const uint32_t i = s0 + (uint32_t)lane;
            const bool on = i < T.spp;
            const int cnt = (int)min((uint32_t)LANES, T.spp - s0);
            float imageX = 0.f, imageY = 0.f, tm = 0.f, su = 0.f, maxt = INFINITY;
            V3 o = v3(0.f, 0.f, 0.f), d = v3(0.f, 0.f, 1.f);
            const size_t ri = (size_t)st.first_ray + k + i;
            uint32_t surfDraws = 0u;   // what PhotonIntegrator::Li draws in front of this sample's volume Li() (pvol_surface_dev.h)
            int glass = 0;
            if (on) {
                imageX = xPos + L.image[2 * i];
                imageY = yPos + L.image[2 * i + 1];
                tm = lerpf(L.time[i], T.shutterOpen, T.shutterClose);
                su = L.scatter[i];
                tile_camera_ray(T, imageX, imageY, &o, &d);
                if (S.bvhNodes) { float th; if (bvh_closest(S, o, d, 0.f, INFINITY, &th) >= 0) maxt = th; }
                else maxt = tile_clip(ltri, S.nTris, o, d);
                if (S.nSpheres) { V3 ph; spheres_closest(S, o, d, 0.f, &maxt, &ph); }
                uint32_t link = 0u;
                if (surfOn && maxt < INFINITY) {
                    SurfHit sh;
                    if (surf_closest(S, o, d, 0.f, &sh)) {
                        if (!FUSED) surfDraws = tile_surface_draws<SPEC>(S, T, CC, ltri, trows, sh, d, su, tm, ri, blackS, lightBlack, blackMask, A.counters, &link);
                        else if (!SPEC || S.shootScene->mats[sh.mat].kind == PVOL_MATERIAL_MATTE || !T.specOn) surfDraws = surf_count_draws(S, sh, d, blackMask);
                        else glass = 1;   // the tree is walked below, with the whole wave
                    }
                }
                if (!FUSED && T.specLink) T.specLink[ri] = link;
                pvol_ray pr;
                pr.o[0] = o.x; pr.o[1] = o.y; pr.o[2] = o.z; pr.mint = 0.f;
                pr.d[0] = d.x; pr.d[1] = d.y; pr.d[2] = d.z; pr.maxt = maxt;
                pr.time = tm; pr.scatter_u = su; pr.rng_skip = ((i == 0) ? samplerDraws : 0u) + surfDraws; pr.flags = 0u;
                rays[ri] = pr;
                T.xy[2 * ri] = imageX;
                T.xy[2 * ri + 1] = imageY;
            }
            if (!FUSED) {
                uint32_t nd = (on && !(T.debugSkip & 2u)) ? tile_count_draws(S, CC, ltri, trows, o, d, maxt, su, blackS, lightBlack, T.debugSkip) : ((T.debugSkip & 2u) ? 270u : 0u);
                // wave total (integer adds in any order are exact)
                unsigned long long tot = nd + surfDraws;
                for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
                if (!(T.debugSkip & 4u)) rng_skip<true, true, true>(rng, tot, lane);

            } else {
                for (int j = 0; j < cnt; ++j) {
                    pvol_ray pr;
                    pr.o[0] = lane_f(o.x, j); pr.o[1] = lane_f(o.y, j); pr.o[2] = lane_f(o.z, j); pr.mint = 0.f;
                    pr.d[0] = lane_f(d.x, j); pr.d[1] = lane_f(d.y, j); pr.d[2] = lane_f(d.z, j); pr.maxt = lane_f(maxt, j);
                    pr.time = lane_f(tm, j); pr.scatter_u = lane_f(su, j); pr.rng_skip = 0u; pr.flags = 0u;
                    const size_t rj = (size_t)st.first_ray + k + s0 + (uint32_t)j;
                    if (surfOn) {   // the surface integrator's draws in front of this sample's own volume Li()
                        uint32_t linkj = 0u;
                        bool walked = false;
                        if constexpr (SPEC) if (lane_i(glass, j)) {
                            walked = true;
                            const unsigned long long t0 = rng.draws;
                            const V3 oj = v3(pr.o[0], pr.o[1], pr.o[2]), dj = v3(pr.d[0], pr.d[1], pr.d[2]);
                            SurfHit shj;
                            shj.tri = 0; shj.mat = 0; shj.t = 0.f; shj.rayEps = 0.f; shj.p = shj.nn = shj.dpdu = v3(0.f, 0.f, 0.f);
                            surf_closest(S, oj, dj, 0.f, &shj);   // lane j found this hit above; every lane repeats it (wave-uniform)
                            SpecCtx X;
                            X.blackMask = blackMask; X.pending = 0u; X.nSeg = 0u;
                            SpecNullPol np;
                            spec_surface<0>(S, X, np, dj, shj);
                            const uint32_t nSeg = X.nSeg;
                            uint32_t base = 0u;
                            if (nSeg) { if (lane == 0) base = atomicAdd(T.segCounter, nSeg); base = (uint32_t)lane_i((int)base, 0); }
                            X.pending = 0u; X.nSeg = 0u;
                            if (nSeg && base + nSeg <= T.segCap && A.liteResolve) {
                                SpecWalkPol wp;
                                wp.S = &S; wp.A = &A; wp.T = &T; wp.rng = &rng; wp.lightNum = M.lightNum; wp.base = base; wp.sample = (uint32_t)rj;
                                wp.su = pr.scatter_u; wp.time = pr.time; wp.lane = lane; wp.grid = grid; wp.blackS = blackS; wp.blackMask = blackMask;
                                spec_surface<0>(S, X, wp, dj, shj);
                                linkj = (base << SPEC_LINK_COUNT_BITS) | nSeg;
                            } else {
                                if (nSeg && lane == 0) atomicAdd(&A.counters->nErrors, 1ull);   // pool exhausted (or a roulette-capable medium): reported
                                spec_surface<0>(S, X, np, dj, shj);
                            }
                            rng_skip<true>(rng, X.pending, lane);   // what the primary hit itself still draws (the lobes that spawned nothing)
                            if (lane == j) rays[rj].rng_skip += (uint32_t)(rng.draws - t0);
                        }
                        if (!walked) rng_skip<true>(rng, (unsigned long long)lane_i((int)surfDraws, j), lane);
                        if (T.specLink && lane == j) T.specLink[rj] = linkj;
                    }
                    const unsigned long long r0 = rng.draws;
                    RayRec rec = ray_rec(A.records + ((size_t)sidx * A.sliceM + (k + s0 + (uint32_t)j - begin)) * A.recStride, S.maxSteps, grid);
                    if (A.liteResolve) {   // no drawn value decides more than the light of a step: geometry one step per lane, then the RNG alone
                        // PVOL_TILE_DEBUG timing knobs (results are wrong with any of them): 32 no visibility tests, 64 no lightNum
                        // shuffle, 128 no RNG walk at all, 256 the shuffle's draws without its swaps
                        const int nSteps = geo_ray(S, A, pr, rec, lane, grid, (T.debugSkip & 32u) ? true : blackS, blackMask);
                        if (T.debugSkip & 128u) rng_skip<true>(rng, 4ull + 7ull * (unsigned long long)nSteps, lane);
                        else lite_ray(S, rec, nSteps, rng, M.lightNum, lane, grid, (T.debugSkip & 64u) ? 1 : S.nLights, (T.debugSkip & 256u) != 0u);
                    } else {
                        f4 Lv, Tr;
                        march_ray<false, MODE_RESOLVE, NREG>(S, A, pr, rng, M, lane, wc, &Lv, &Tr, rec);
                    }
                    if (lane == 0) rec.hdr[2] = (uint32_t)(rng.draws - r0);
                    if (A.draws && lane == 0) A.draws[rj] = (uint32_t)(rng.draws - r0);
                }
            }
        }
    }
    if (lane == 0) A.streams[sidx].end_draw = rng.draws;
    if (state) {
        __syncthreads();
        for (int i = lane; i < MT_N; i += LANES) state[i] = mt[i];
        if (lane == 0) state[MT_N] = (uint32_t)rng.mti;
    }
}

// COUNT mode with NW waves per render task.  One task is one MT19937 stream, so its pixels stay a serial chain -- but with few
// tasks per CU (one rank of a multi-GPU render: 512 tasks on 256 CUs) the chip idles while every task's single wave works
// through 4 trips of 64 camera samples and ~17 double march steps each.  Here wave 0 owns the stream (LDPixelSample and its
// shuffles, the advance) and ALL waves share the draw count of a pixel: wave = slice * G + g handles the samples of group g
// (64 at a time) and the march steps of its slice (tile_count_draws: steps dealt round-robin), the partial counts meet in LDS.
// Two workgroup barriers per pixel; the stream only ever needs the pixel's TOTAL, so one skip replaces the per-trip skips.
template <int NW, bool SPEC>
// Register budget = residency: the kernel wants ~340 VGPRs, and at that size a CU holds one wave per SIMD -- ONE task's workgroup (four waves)
// or two tasks' (two waves) at a time, while a rank of an 8-GPU frame has two tasks per CU and one of a 4-GPU frame four.  Two waves per SIMD
// (256 VGPRs, 83 spilled) keeps every task of the CU resident: 512-task pre-pass 85.9 (eight waves, 256 VGPRs) -> 56.5 ms (four waves),
// 1 024-task pre-pass 134 -> 77 ms (two waves).  Eight and sixteen waves per task ask for four per SIMD (128 VGPRs).
__global__ __launch_bounds__(LANES * NW, (NW >= 8 ? 4 : 2)) void tile_mw_kernel(LiArgs A, TileArgs T, uint32_t partOff) {
    extern __shared__ __align__(16) unsigned char lds[];
    constexpr bool W1 = (NW == 1);   // a one-wave workgroup (debugging form) may use the workgroup barrier inside the RNG helpers
    const DevScene &S = *A.scene;
    const int tid = threadIdx.x, lane = tid & (LANES - 1), wave = tid / LANES;
    const uint32_t sidx = blockIdx.x;
    if (sidx >= A.nStreams) return;
    uint32_t *mt = reinterpret_cast<uint32_t *>(lds);
    TileLds L;
    L.image = reinterpret_cast<float *>(lds + MT_N * 4);
    L.time = L.image + 2 * T.spp;
    L.scatter = L.time + T.spp;
    L.oth = reinterpret_cast<uint32_t *>(L.scatter + T.spp);
    const size_t triOff = (((size_t)MT_N * 4 + (size_t)T.spp * 5 * 4) + 15) & ~(size_t)15;
    float *ltri = reinterpret_cast<float *>(lds + triOff);
    const bool shadowRows = S.nLights > 0 && S.lights[0].kind == PVOL_LIGHT_DISTANT;
    const size_t partHere = (triOff + (size_t)S.nTris * 12 * 4 + (shadowRows ? (size_t)S.nTris * 16 * 4 : 0) + 64 + 15) & ~(size_t)15;
    unsigned long long *part = reinterpret_cast<unsigned long long *>(lds + partHere);
    for (int i = tid; i < S.nTris * 12; i += LANES * NW) {
        const int t = i / 12, c = i - 12 * t;
        const DevTri &tr = S.tris[t];
        ltri[i] = c < 3 ? tr.p1[c] : c < 6 ? tr.p2[c - 3] : c < 9 ? tr.p3[c - 6] : 0.f;
    }
    __syncthreads();
    float *trows = 0;
    if (S.nLights > 0 && S.lights[0].kind == PVOL_LIGHT_DISTANT) {
        trows = ltri + S.nTris * 12;
        tri_rows_prepare(S, v3(S.lights[0].dir[0], S.lights[0].dir[1], S.lights[0].dir[2]), trows, tid);   // workgroup barrier inside
    }
    const pvol_stream st = A.streams[sidx];
    Rng rng;
    rng.mt = mt; rng.draws = 0; rng.mti = MT_N;
    if (wave == 0) {
        mt_seed<W1>(mt, st.seed, lane);
        rng_skip<true, W1>(rng, st.start_draw, lane);
    }
    const int4 w = T.windows[sidx];
    const uint32_t width = (uint32_t)(w.y - w.x);
    const int q = lane & 7;
    const bool blackS = spec_is_black(ld4(S.sigS, q));
    const bool lightBlack = S.nLights > 0 ? spec_is_black(ld4(S.lights[0].intensity, q)) : true;
    unsigned blackMask = 0u;
    for (int l = 0; l < S.nLights; ++l) if (spec_is_black(ld4(S.lights[l].intensity, q))) blackMask |= 1u << l;
    const bool surfOn = S.surf.enabled != 0;
    pvol_ray *rays = T.rays;
    const CountConsts CC = count_consts(S);
    // sample groups served at once, and march-step slices per group
    const int groupsPerPixel = (int)((T.spp + LANES - 1) / LANES);
    const int G = groupsPerPixel < NW ? groupsPerPixel : NW;   // both powers of two
    const int NSL = NW / G;
    const int g = wave % G, slice = wave / G;
    for (uint32_t k = 0; k < st.n_rays; k += T.spp) {
        const uint32_t pix = k / T.spp;
        const int xPos = w.x + (int)(pix % width), yPos = w.z + (int)(pix / width);
        uint32_t samplerDraws = 0u;
        if (wave == 0) {
            const unsigned long long d0 = rng.draws;
            tile_pixel_sample<W1>(T, L, rng, lane);
            samplerDraws = (uint32_t)(rng.draws - d0);
        }
        __syncthreads();   // the pixel's samples are in LDS
        unsigned long long mine = 0ull;
        for (uint32_t s0 = (uint32_t)g * LANES; s0 < T.spp; s0 += (uint32_t)G * LANES) {
            const uint32_t i = s0 + (uint32_t)lane;
            if (i >= T.spp) continue;
            const float imageX = xPos + L.image[2 * i], imageY = yPos + L.image[2 * i + 1];
            const float tm = lerpf(L.time[i], T.shutterOpen, T.shutterClose), su = L.scatter[i];
            V3 o, d;
            tile_camera_ray(T, imageX, imageY, &o, &d);
            float maxt = INFINITY;
            if (S.bvhNodes) { float th; if (bvh_closest(S, o, d, 0.f, INFINITY, &th) >= 0) maxt = th; }
            else maxt = tile_clip(ltri, S.nTris, o, d);
            if (S.nSpheres) { V3 ph; spheres_closest(S, o, d, 0.f, &maxt, &ph); }
            if (slice == 0) {
                uint32_t surfDraws = 0u;
                const size_t ri = (size_t)st.first_ray + k + i;
                uint32_t link = 0u;
                if (surfOn && maxt < INFINITY) {
                    SurfHit sh;
                    if (surf_closest(S, o, d, 0.f, &sh)) surfDraws = tile_surface_draws<SPEC>(S, T, CC, ltri, trows, sh, d, su, tm, ri, blackS, lightBlack, blackMask, A.counters, &link);
                }
                if (T.specLink) T.specLink[ri] = link;
                pvol_ray pr;
                pr.o[0] = o.x; pr.o[1] = o.y; pr.o[2] = o.z; pr.mint = 0.f;
                pr.d[0] = d.x; pr.d[1] = d.y; pr.d[2] = d.z; pr.maxt = maxt;
                pr.time = tm; pr.scatter_u = su; pr.rng_skip = ((i == 0) ? samplerDraws : 0u) + surfDraws; pr.flags = 0u;
                rays[ri] = pr;
                T.xy[2 * ri] = imageX;
                T.xy[2 * ri + 1] = imageY;
                mine += surfDraws;
            }
            if (!(T.debugSkip & 2u)) mine += tile_count_draws(S, CC, ltri, trows, o, d, maxt, su, blackS, lightBlack, 0u, slice, NSL);
            else if (slice == 0) mine += 270ull;   // timing knob (PVOL_TIMING_KNOBS builds only)
        }
        for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
        if (lane == 0) part[wave] = mine;
        __syncthreads();   // every wave's partial count is in LDS (and nobody reads this pixel's samples any more)
        if (wave == 0) {
            unsigned long long tot = lane < NW ? part[lane] : 0ull;
            for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
            if (!(T.debugSkip & 4u)) rng_skip<true, W1>(rng, tot, lane);   // (timing knob 4)
        }
    }
    if (wave == 0 && lane == 0) A.streams[sidx].end_draw = rng.draws;
}

// LDS: MT19937 state | lightNum (fused) | the pixel's samples (5 words each) | triangles (12 words each) | their shadow rows
// for a distant light (16 words each).  Sized by the scene: the resident waves per CU are LDS-bound (one task per wave).
extern "C" size_t pvol_tile_lds_bytes(int maxSteps, uint32_t spp, bool fused, int nTris, bool shadowRows) {
    return (((size_t)MT_N * 4 + (fused ? (size_t)((maxSteps + 1) & ~1) * 4 : 0) + (size_t)spp * 5 * 4 + 15) & ~(size_t)15) + (size_t)nTris * 12 * 4 +
           (shadowRows ? (size_t)nTris * 16 * 4 : 0) + 64;
}

template <bool SPEC>
static hipError_t launch_tile_t(const LiArgs *args, const TileArgs *tile, bool fused, size_t ldsBytes, int candCap, hipStream_t stream, int wavesPerTask) {
    dim3 grid(args->nStreams), block(LANES);
    if (!fused && wavesPerTask > 1 && args->state == 0 && args->sliceK == 0) {
        const uint32_t partOff = (uint32_t)((ldsBytes + 15) & ~(size_t)15);
        const size_t bytes = partOff + 16 * 8;
        if (wavesPerTask >= 16) hipLaunchKernelGGL((tile_mw_kernel<16, SPEC>), grid, dim3(LANES * 16), bytes, stream, *args, *tile, partOff);
        else if (wavesPerTask >= 8) hipLaunchKernelGGL((tile_mw_kernel<8, SPEC>), grid, dim3(LANES * 8), bytes, stream, *args, *tile, partOff);
        else if (wavesPerTask >= 4) hipLaunchKernelGGL((tile_mw_kernel<4, SPEC>), grid, dim3(LANES * 4), bytes, stream, *args, *tile, partOff);
        else hipLaunchKernelGGL((tile_mw_kernel<2, SPEC>), grid, dim3(LANES * 2), bytes, stream, *args, *tile, partOff);
        return hipGetLastError();
    }
    if (!fused) hipLaunchKernelGGL((tile_kernel<false, 4, SPEC>), grid, block, ldsBytes, stream, *args, *tile);
    else if (candCap <= 4 * LANES) hipLaunchKernelGGL((tile_kernel<true, 4, SPEC>), grid, block, ldsBytes, stream, *args, *tile);
    else hipLaunchKernelGGL((tile_kernel<true, 12, SPEC>), grid, block, ldsBytes, stream, *args, *tile);
    return hipGetLastError();
}
// wavesPerTask (COUNT mode, whole batch in one launch): 1 = the one-wave kernel; 4 / 8 / 16 = tile_mw_kernel, chosen by the host when
// few tasks share a CU (pvol_api.hip)
extern "C" hipError_t pvol_launch_tile(const LiArgs *args, const TileArgs *tile, bool fused, size_t ldsBytes, int candCap, hipStream_t stream, int wavesPerTask) {
    return tile->specOn ? launch_tile_t<true>(args, tile, fused, ldsBytes, candCap, stream, wavesPerTask)
                        : launch_tile_t<false>(args, tile, fused, ldsBytes, candCap, stream, wavesPerTask);
}
