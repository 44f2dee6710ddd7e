// pvol_surface_dev.h -- included by pvol_march.hip.  SURVEY 8(f)-2, the part of PhotonIntegrator::Li (integrators/
// photonmap.cpp:154-319) that the BASELINE scenes' camera rays need on their matte walls:
//   L = UniformSampleAllLights (core/integrator.cpp:47-79 -> EstimateDirect :117-174, delta lights)
//     + LPhoton(causticMap)     (photonmap.cpp:62-108, diffuse branch: kernel-weighted k-NN estimate)
// (`indirectphotons 0` in both scenes leaves the final gather and the indirect estimate without a map), plus every RNG
// draw of the rest of that Li(): 2 x 72 of BSDF::rho inside LPhoton, 2 x 3 of the BSDFSample(rng) that SpecularReflect /
// SpecularTransmit construct, one per unoccluded light sample (visibility.Transmittance(..., NULL, rng)).  None of the drawn
// VALUES reaches the result on a Lambertian surface in an analytic medium, so the draws are COUNTED by the tile pre-pass
// (they precede the sample's volume Li() in the task's stream) and the radiance is added by surface_kernel afterwards:
//   sample = T * Lsurface + Lvi        (SamplerRenderer::Li, renderers/samplerrenderer.cpp:238-250)
// Not covered (the host refuses such scenes when the surface integrator is on): specular BSDFs (the recursion of
// SpecularReflect/Transmit), an indirect map / final gather, VolumeGrid media (the shadow-ray tau() offset is a drawn value).
#define SRF_AIM 1.6f    // the search ball is re-aimed at this multiple of nused photons for the sparsest sample of a cluster
#define SRF_CAP 1024   // caustic bucket capacity (photons within the search ball + spread of a group's hit points).  Measured on the C2 frame with the
                       // scene's surface integrator on: 2048 -> 1 509 ms, 1024 -> 1 104 ms, 512 -> 3 580 ms; 640 with SRF_AIM 1.35: 1 278 ms: occupancy (LDS) against overflows

struct SurfHit {
    int tri, mat;
    float t;
    float rayEps;   // Intersection::rayEpsilon: 1e-3 t for a triangle (trianglemesh.cpp:205), 5e-4 t for a sphere (sphere.cpp:155)
    V3 p, nn;
    V3 dpdu;        // the BSDF's frame starts from it (core/reflection.cpp:619-627)
};
// Scene::Intersect for one lane: closest hit, the later triangle on equal t (as the linear scans of this library and its oracle),
// DifferentialGeometry normal as shapes/trianglemesh.cpp:163-181 + core/diffgeom.cpp:46-54 build it
__device__ bool surf_closest(const DevScene &S, V3 o, V3 d, float mint, SurfHit *h) {
    float mt = INFINITY;
    int best = -1;
    V3 p1, p2, p3;
    bool flip;
    if (S.bvhNodes) {
        const int slot = bvh_closest(S, o, d, mint, INFINITY, &mt);
        if (slot < 0 && !S.nSpheres) return false;
        if (slot >= 0) {
        const float4 q1 = S.bvhTris[3 * slot], q2 = S.bvhTris[3 * slot + 1], q3 = S.bvhTris[3 * slot + 2];
        p1 = v3(q1.x, q1.y, q1.z); p2 = v3(q2.x, q2.y, q2.z); p3 = v3(q3.x, q3.y, q3.z);
        best = __float_as_int(q1.w);
        h->mat = __float_as_int(q2.w);
        flip = __float_as_int(q3.w) != 0;
        }
    } else {
        for (int i = 0; i < S.nTris; ++i) {
            float t;
            if (tri_closest(S.tris[i], o, d, mint, mt, &t)) { mt = t; best = i; }
        }
        if (best < 0 && !S.nSpheres) return false;
        if (best >= 0) {
            const DevTri &tr = S.tris[best];
            p1 = v3(tr.p1[0], tr.p1[1], tr.p1[2]); p2 = v3(tr.p2[0], tr.p2[1], tr.p2[2]); p3 = v3(tr.p3[0], tr.p3[1], tr.p3[2]);
            h->mat = S.shootScene->triMat[best];
            flip = S.shootScene->triFlip[best] != 0;
        }
    }
    if (S.nSpheres) {   // Shape "sphere" (pvol_sphere_dev.h): tested after the triangles
        V3 ph, dpduW;
        const int si = spheres_closest(S, o, d, mint, &mt, &ph);
        if (si >= 0) {
            h->tri = -1 - si;
            h->mat = S.spheres[si].mat;
            h->t = mt;
            h->rayEps = 5e-4f * mt;
            sphere_dg(S.spheres[si], ph, &h->p, &dpduW, &h->nn);
            h->dpdu = dpduW;
            return true;
        }
        if (best < 0) return false;
    }
    const float du1 = 0.f - 1.f, du2 = 1.f - 1.f, dv1 = 0.f - 1.f, dv2 = 0.f - 1.f;
    const V3 dp1 = p1 - p3, dp2 = p2 - p3;
    const float invdet = 1.f / (du1 * dv2 - dv1 * du2);
    const V3 dpdu = (dp1 * dv2 - dp2 * dv1) * invdet;
    const V3 dpdv = (dp1 * (-du2) + dp2 * du1) * invdet;
    h->tri = best;
    h->t = mt;
    h->rayEps = 1e-3f * mt;
    h->p = o + d * mt;
    h->nn = normalize(cross(dpdu, dpdv));
    h->dpdu = dpdu;
    if (flip) h->nn = h->nn * -1.f;
    return true;
}

// One delta light seen from a surface point: Light::Sample_L(p, eps, ...) (spot.cpp:50-57, point.cpp:50-57, distant.cpp:48-55),
// the shadow ray of its VisibilityTester (core/light.h:85-101) and whether EstimateDirect would take the sample.
struct SurfLight { bool take; V3 wi; float scale; RayD vis; };   // radiance = intensity * scale
__device__ __forceinline__ SurfLight surf_light(const DevScene &S, int ln, V3 p, float eps, V3 n, V3 wo, bool lambert, unsigned blackMask) {
    SurfLight r;
    const DevLight &light = S.lights[ln];
    r.scale = 1.f;
    if (light.kind == PVOL_LIGHT_DISTANT) {
        r.wi = v3(light.dir[0], light.dir[1], light.dir[2]);
        r.vis.o = p; r.vis.d = r.wi; r.vis.mint = eps; r.vis.maxt = INFINITY;
    } else {
        const V3 lp = v3(light.pos[0], light.pos[1], light.pos[2]);
        r.wi = normalize(lp - p);
        const float dist = len(p - lp);
        r.vis.o = p; r.vis.d = vdiv(lp - p, dist); r.vis.mint = eps; r.vis.maxt = dist * (1.f - 0.f);
        const float d2 = len_sq(lp - p);
        float fall = 1.f;
        if (light.kind == PVOL_LIGHT_SPOT) {
            const V3 w = -r.wi;
            const V3 wl = normalize(v3(light.w2l[0] * w.x + light.w2l[1] * w.y + light.w2l[2] * w.z, light.w2l[4] * w.x + light.w2l[5] * w.y + light.w2l[6] * w.z,
                                       light.w2l[8] * w.x + light.w2l[9] * w.y + light.w2l[10] * w.z));
            const float ct = wl.z;
            if (ct < light.cosTotalWidth) fall = 0.f;
            else if (!(ct > light.cosFalloffStart)) {
                const float delta = (ct - light.cosTotalWidth) / (light.cosFalloffStart - light.cosTotalWidth);
                fall = delta * delta * delta * delta;
            }
        }
        r.scale = fall / d2;
    }
    const bool liBlack = r.scale == 0.f || ((blackMask >> ln) & 1u);
    // BSDF::f: the Lambertian lobe counts only when wi and wo lie on the same side of the geometric normal (reflection.cpp:627-644)
    const bool fOk = lambert && dot(r.wi, n) * dot(wo, n) > 0.f;
    r.take = !liBlack && fOk && !lane_occluded(S, r.vis);
    return r;
}

// RandomUInt calls of PhotonIntegrator::Li for a ray of depth `depth` that hit h on a MATTE surface (in front of that ray's volume Li())
__device__ uint32_t surf_count_draws(const DevScene &S, const SurfHit &h, V3 d, unsigned blackMask, int depth = 0) {
    const DevMaterial &m = S.shootScene->mats[h.mat];
    const bool lambert = m.kind == PVOL_MATERIAL_MATTE && m.nBxdf > 0;   // MatteMaterial::GetBSDF adds the Lambertian only for a non-black Kd
    const V3 wo = -d;
    uint32_t n = 0;
    for (int ln = 0; ln < S.nLights; ++ln) n += surf_light(S, ln, h.p, h.rayEps, h.nn, wo, lambert, blackMask).take ? 1u : 0u;
    if (S.surf.nPhotons > 0u && lambert) n += 144u;     // LPhoton(causticMap): two BSDF::rho(wo, rng)
    if (depth + 1 < S.surf.maxSpecularDepth) n += 6u;   // SpecularReflect + SpecularTransmit: BSDFSample(rng) each
    return n;
}


__device__ __forceinline__ V3 lane_v3(V3 v, int l) { return v3(lane_f(v.x, l), lane_f(v.y, l), lane_f(v.z, l)); }
// The caustic lookup of ONE hit point whose ball holds more photons than the LDS bucket (a caustic's focus): the whole wave streams
// the photons of the ball's cell rows from global memory -- how many lie inside maxdist; when at least nused do, the nused-th
// smallest distance^2 by bisection over the fp32 bit pattern (one pass per step, as the bucket form does over LDS); then the flux
// sum over the members (ties at the nused-th distance in photon order).  Every lane returns the wave's sums in acc[0..29]:
// Lr = sum kernel / (nPaths r^2) alpha over the members on the wo side (photonmap.cpp:62-108).
__device__ __attribute__((noinline)) void caustic_stream(const DevScene &S, const GridView &cg, V3 P, V3 Nf, int k, int lane, float *acc) {
    const float maxD2 = S.surf.maxDistSq;
    const GridRows R = grid_rows<false>(cg, P, sqrtf(maxD2), 0);
    // pass 0: count inside; pass 1..: bisection steps; each pass visits every photon of the rows once
    uint32_t lo = 0u, hi = __float_as_uint(maxD2), kth = 0xffffffffu;
    int nIn = 0, less = 0;
    for (int pass = 0; pass < 40; ++pass) {
        const bool counting = pass == 0;
        const uint32_t mid = lo + ((hi - lo) >> 1);
        int cnt = 0;
        for (int r = 0; r < R.nrows; ++r) {
            uint32_t start, rlen;
            grid_row_range<false>(cg, R, r, P, maxD2, &start, &rlen);
            for (uint32_t b = 0; b < rlen; b += LANES) {
                const uint32_t i = b + (uint32_t)lane;
                bool hitp = false;
                if (i < rlen) {
                    const float4 q = cg.pos4[start + i];
                    const float dx = q.x - P.x, dy = q.y - P.y, dz = q.z - P.z;
                    const float d2 = dx * dx + dy * dy + dz * dz;
                    hitp = counting ? d2 < maxD2 : __float_as_uint(d2) <= mid;
                }
                cnt += __popcll(__ballot(hitp));
            }
        }
        if (counting) { nIn = cnt; if (nIn < k) break; continue; }
        if (cnt >= k) hi = mid; else lo = mid + 1u;
        if (!(lo < hi)) { kth = lo; break; }
    }
    float r2 = maxD2;
    int tieQuota = 0;
    if (kth != 0xffffffffu) {
        r2 = __uint_as_float(kth);
        for (int r = 0; r < R.nrows; ++r) {
            uint32_t start, rlen;
            grid_row_range<false>(cg, R, r, P, maxD2, &start, &rlen);
            for (uint32_t b = 0; b < rlen; b += LANES) {
                const uint32_t i = b + (uint32_t)lane;
                bool lt = false;
                if (i < rlen) {
                    const float4 q = cg.pos4[start + i];
                    const float dx = q.x - P.x, dy = q.y - P.y, dz = q.z - P.z;
                    lt = __float_as_uint(dx * dx + dy * dy + dz * dz) < kth;
                }
                less += __popcll(__ballot(lt));
            }
        }
        tieQuota = k - less;
    }
    float a[32];
#pragma unroll
    for (int b = 0; b < 32; ++b) a[b] = 0.f;
    const float norm = 1.f / ((float)S.surf.nCausticPaths * r2);
    int tiesSeen = 0;
    for (int r = 0; r < R.nrows; ++r) {
        uint32_t start, rlen;
        grid_row_range<false>(cg, R, r, P, maxD2, &start, &rlen);
        for (uint32_t b = 0; b < rlen; b += LANES) {
            const uint32_t i = b + (uint32_t)lane;
            bool mem = false, tie = false;
            float d2 = 0.f;
            if (i < rlen) {
                const float4 q = cg.pos4[start + i];
                const float dx = q.x - P.x, dy = q.y - P.y, dz = q.z - P.z;
                d2 = dx * dx + dy * dy + dz * dz;
                if (kth == 0xffffffffu) mem = d2 < maxD2;
                else { const uint32_t bits = __float_as_uint(d2); mem = bits < kth; tie = bits == kth; }
            }
            const unsigned long long tb = __ballot(tie);
            if (tie) {
                const int rank = tiesSeen + __popcll(tb & ((1ull << lane) - 1ull));
                mem = rank < tieQuota;
            }
            tiesSeen += __popcll(tb);
            if (mem) {
                const f4 wi4 = S.surf.wi4[start + i];
                if (Nf.x * wi4.x + Nf.y * wi4.y + Nf.z * wi4.z > 0.f) {
                    const float sW = 1.f - d2 / r2;
                    const float wgt = (3.f * 0.31830988618379067154f * sW * sW) * norm;
                    const float4 *row = S.surf.alpha4 + (size_t)(start + i) * 8;
#pragma unroll
                    for (int qq = 0; qq < 8; ++qq) {
                        const float4 rr = row[qq];
                        a[4 * qq] = __builtin_fmaf(rr.x, wgt, a[4 * qq]); a[4 * qq + 1] = __builtin_fmaf(rr.y, wgt, a[4 * qq + 1]);
                        a[4 * qq + 2] = __builtin_fmaf(rr.z, wgt, a[4 * qq + 2]); a[4 * qq + 3] = __builtin_fmaf(rr.w, wgt, a[4 * qq + 3]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int b = 0; b < 30; ++b) {
        float v = a[b];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        acc[b] = v;
    }
}

// One camera sample per lane, 64 consecutive samples (one pixel's, at >= 64 spp) per group: their hit points lie within a
// pixel footprint, so the caustic lookups share one LDS bucket of the photons within maxdist + spread of the group's centre.
__global__ __launch_bounds__(LANES, 2) void surface_kernel(SurfArgs A) {   // 256 VGPRs: its 20.5 KB of LDS allow seven waves per CU anyway (at 168 VGPRs it spilled 531)
    extern __shared__ __align__(16) unsigned char lds[];
    const DevScene &S = *A.scene;
    const int lane = threadIdx.x;
    constexpr int PITCH = SRF_CAP + 4;
    float *bucket = reinterpret_cast<float *>(lds);
    Gather G;
    G.cap = 0; G.cd = 0; G.ci = 0;
    G.paint = reinterpret_cast<uint32_t *>(bucket + 4 * PITCH);
    const float *bX = bucket, *bY = bucket + PITCH, *bZ = bucket + 2 * PITCH, *bI = bucket + 3 * PITCH;
    unsigned blackMask = 0u;
    { const int q = lane & 7; for (int l = 0; l < S.nLights; ++l) if (spec_is_black(ld4(S.lights[l].intensity, q))) blackMask |= 1u << l; }
    GridView cg;
    cg.cellSize = S.surf.cellSize; cg.invCell = S.surf.invCell;
    for (int i = 0; i < 3; ++i) { cg.gridLo[i] = S.surf.gridLo[i]; cg.gdim[i] = S.surf.gdim[i]; }
    cg.cellStart = S.surf.cellStart; cg.subStart = 0; cg.pos4 = S.surf.pos4;
    const int k = S.surf.nLookup;
    // groups of 64 samples in runs of sixteen (four 256-spp pixels) per wave: the search radius that served one group is the first
    // guess for the next (the caustic density changes slowly along a wall)
    const float maxDist = sqrtf(S.surf.maxDistSq);
    float Rtry = maxDist;   // wave-uniform
    const unsigned long long nGroups = (A.nRays + LANES - 1) / LANES;
    for (unsigned long long gi = (unsigned long long)blockIdx.x * 16ull; gi < nGroups; gi = ((gi & 15ull) == 15ull) ? gi + 1ull + ((unsigned long long)gridDim.x - 1ull) * 16ull : gi + 1ull) {
        const unsigned long long g0 = gi * LANES;
        const size_t ri = (size_t)g0 + lane;
        const bool have = ri < A.nRays;
        const pvol_ray pr = A.rays[have ? ri : 0];
        const V3 o = v3(pr.o[0], pr.o[1], pr.o[2]), d = v3(pr.d[0], pr.d[1], pr.d[2]);
        SurfHit h;
        h.tri = 0; h.mat = 0; h.t = 0.f; h.rayEps = 0.f; h.p = h.nn = h.dpdu = v3(0.f, 0.f, 0.f);
        // (the ray's maxt is this very t: the tile pre-pass clipped it; an unused slot of the segment pool has maxt < mint)
        const bool hit = have && pr.maxt >= pr.mint && surf_closest(S, o, d, pr.mint, &h);
        float Ls[32];
#pragma unroll
        for (int b = 0; b < 32; ++b) Ls[b] = 0.f;
        const DevMaterial &mat = S.shootScene->mats[h.mat];
        const bool lambert = hit && mat.kind == PVOL_MATERIAL_MATTE && mat.nBxdf > 0;
        const V3 wo = -d;
        // ---- direct lighting: Ld = f * Li * (AbsDot(wi, n) / pdf), Li = light radiance * Transmittance of the shadow ray
        if (lambert) {
            for (int ln = 0; ln < S.nLights; ++ln) {
                const SurfLight sl = surf_light(S, ln, h.p, h.rayEps, h.nn, wo, true, blackMask);
                if (!sl.take) continue;
                float lenAB = 0.f;   // tau of the analytic medium along the shadow ray (homogeneous.h:80-84)
                if (S.volKind != PVOL_VOLUME_NONE) {
                    float t0, t1;
                    if (vol_intersect(S, sl.vis, &t0, &t1)) { const V3 a = sl.vis.o + sl.vis.d * t0, b = sl.vis.o + sl.vis.d * t1; lenAB = len(a - b); }
                }
                const float geo = fabsf(dot(sl.wi, h.nn)) / 1.f;
                const float kSh = -1.442695041f * lenAB;
#pragma unroll
                for (int b = 0; b < 30; ++b) {
                    const float sT = S.sigA[b] + S.sigS[b];
                    const float Li = (S.lights[ln].intensity[b] * sl.scale) * __builtin_amdgcn_exp2f(sT * kSh);
                    Ls[b] += (mat.kd[b] * 0.31830988618379067154f) * Li * geo;
                }
            }
        }
        // ---- caustic estimate (photonmap.cpp:62-108), shared buckets: the samples still waiting are served cluster by cluster
        // -- all of them at once when they lie within maxdist of their common centre (one pixel's samples on one wall), else
        // those within maxdist of the first waiting sample's hit point, and that sample alone if even this bucket overflows.
        const bool needAny = lambert && S.surf.nPhotons > 0u;
        unsigned long long pending = __ballot(needAny);
        bool alone = false;
        // Where more photons lie within maxdist of the samples than the bucket holds (a caustic's focus; the beam's footprint on a wall),
        // the nused nearest lie far inside maxdist: the cluster is then searched with a SMALLER radius -- halved area per overflow,
        // bracketed between a radius that overflowed and one that held fewer than nused for some sample -- and a sample is served as
        // soon as nused photons lie inside the searched ball (they are its nused nearest).  Only when no radius serves the cluster
        // does one sample at a time go through the full-radius path (round 2 went there at the first overflow: 64 stagings, or 64
        // streamed lookups, per group of samples -- 3.1 s of a 4.9 s C2 frame with the scene's surface integrator on).
        float Rhi = INFINITY, Rlo = 0.f;   // wave-uniform (Rtry carries over from the previous group)
        while (pending) {
            const bool waiting = ((pending >> lane) & 1ull) != 0ull;
            const float big = 3.0e38f;
            float lx = waiting ? h.p.x : big, ly = waiting ? h.p.y : big, lz = waiting ? h.p.z : big;
            float hx = waiting ? h.p.x : -big, hy = waiting ? h.p.y : -big, hz = waiting ? h.p.z : -big;
            lx = -wave_max(-lx); ly = -wave_max(-ly); lz = -wave_max(-lz);
            hx = wave_max(hx); hy = wave_max(hy); hz = wave_max(hz);
            V3 c = v3(0.5f * (lx + hx), 0.5f * (ly + hy), 0.5f * (lz + hz));
            float rho = wave_max(waiting ? len(h.p - c) : 0.f);
            bool need = waiting;
            int pivLane = 0;
            if (alone || rho > maxDist) {
                const int piv = __ffsll((long long)pending) - 1;
                pivLane = piv;
                c = v3(lane_f(h.p.x, piv), lane_f(h.p.y, piv), lane_f(h.p.z, piv));
                const float dPiv = len(h.p - c);
                need = waiting && (alone ? dPiv == 0.f : dPiv <= maxDist);
                rho = alone ? 0.f : maxDist;
            }
            const float Rq = alone ? maxDist : Rtry;
            const float Rs = (Rq + rho) * 1.0001f + 1e-6f;
            unsigned long long tst = 0;
            __syncthreads();
            const int Mb = stage_bucket_g<SRF_CAP>(cg, G, bucket, c, Rs, lane, tst);
            if (Mb < 0 && !alone) {   // overflow: a smaller ball, or -- the bracket closed -- the first waiting sample on its own
                Rhi = Rq;
                // (a cluster whose own spread overflows the bucket cannot be helped by a smaller ball: every trip either serves a sample,
                // narrows the bracket or -- below a tenth of maxdist, or below the spread -- ends in the one-sample path: the loop ends)
                if ((Rlo > 0.f && Rhi < 1.1f * Rlo) || Rq < 0.1f * maxDist || Rq < rho) alone = true;
                else Rtry = Rlo > 0.f ? sqrtf(Rlo * Rhi) : 0.70710678f * Rq;
                continue;
            }
            const float T2 = (alone || !(Rq < maxDist)) ? S.surf.maxDistSq : Rq * Rq;   // the searched ball's radius^2
            const bool wasAlone = alone;
            alone = false;
            if (Mb < 0) {   // more photons within maxdist of ONE point than the bucket holds: streamed from the grid for that point
                pending &= ~__ballot(need);
                float sa[32];
                caustic_stream(S, cg, c, dot(lane_v3(h.nn, pivLane), lane_v3(wo, pivLane)) < 0.f ? lane_v3(h.nn, pivLane) * -1.f : lane_v3(h.nn, pivLane), k, lane, sa);
                if (need) {
#pragma unroll
                    for (int b = 0; b < 30; ++b) Ls[b] += sa[b] * mat.kd[b] * 0.31830988618379067154f;
                }
            } else {
                // pass A: how many photons lie inside maxdist (kdtree.h:180: dist2 < maxDistSquared), and -- when that is at least
                // nused -- the nused-th smallest distance^2 by bisection over the fp32 bit pattern (the heap of PhotonProcess keeps
                // the nused nearest and leaves maxDistSquared at the farthest of them)
                // (round 3: the nused-th distance^2 comes from three histogram passes as in li_fixup_group_kernel -- 64 bins on [0, maxdist^2)
                // with 8-bit counters, 64 sub-bins of the bin that holds it, the <= 8 members of that sub-bin ordered in registers --
                // instead of ~30 bisection passes over the bucket, which were 3 s of a C2 frame with the scene's surface integrator on;
                // the bisection remains for a lane whose counters wrapped or whose sub-bin is crowded)
                uint32_t *hist = G.paint;   // [16 words][64 lanes]: the paint list is dead once the bucket is staged
                const float hscale = 64.f / T2;
                int nIn = 0;
                fxg_clear(hist, lane);
                __syncthreads();
                for (int i = 0; i < Mb; ++i) {
                    const float dx = bX[i] - h.p.x, dy = bY[i] - h.p.y, dz = bZ[i] - h.p.z;
                    const float d2 = dx * dx + dy * dy + dz * dz;
                    const bool inside = need && d2 < T2;
                    nIn += inside ? 1 : 0;
                    const uint32_t bin = (uint32_t)fminf(d2 * hscale, 63.f);
                    atomicAdd(&hist[(bin >> 2) * LANES + lane], inside ? 1u << ((bin & 3u) << 3) : 0u);
                }
                __syncthreads();
                float r2 = T2;
                uint32_t kth = 0xffffffffu;   // members: d2 < maxDistSq when fewer than nused lie inside, else the nused nearest
                int tieQuota = 0;
                if (__ballot(nIn >= k)) {
                    const bool sel = nIn >= k;
                    bool fast = sel;   // the histogram passes hold for this lane
                    uint32_t bstar = 0u, below = 0u, inBin = 0u, total = 0u;
                    const bool f1 = fxg_scan(hist, lane, (uint32_t)k, &bstar, &below, &inBin, &total);
                    if (!f1 || total != (uint32_t)nIn) fast = false;   // a byte wrapped
                    const float fb = (float)bstar;
                    fxg_clear(hist, lane);
                    __syncthreads();
                    for (int i = 0; i < Mb; ++i) {
                        const float dx = bX[i] - h.p.x, dy = bY[i] - h.p.y, dz = bZ[i] - h.p.z;
                        const float d2 = dx * dx + dy * dy + dz * dz;
                        const float f = d2 * hscale;
                        const bool hitB = fast && d2 < T2 && (uint32_t)fminf(f, 63.f) == bstar;
                        const uint32_t sub = (uint32_t)fminf(fmaxf((f - fb) * 64.f, 0.f), 63.f);
                        atomicAdd(&hist[(sub >> 2) * LANES + lane], hitB ? 1u << ((sub & 3u) << 3) : 0u);
                    }
                    __syncthreads();
                    uint32_t sstar = 0u, below2 = 0u, inSub = 0u, total2 = 0u;
                    const uint32_t need1 = (uint32_t)k - below;   // rank of the nused-th inside its bin, >= 1
                    const bool f2 = fxg_scan(hist, lane, need1, &sstar, &below2, &inSub, &total2);
                    if (!f2 || total2 != inBin || inSub > 8u) fast = false;
                    float mini[8];
#pragma unroll
                    for (int m = 0; m < 8; ++m) mini[m] = INFINITY;
                    if (__ballot(fast)) {
                        for (int i = 0; i < Mb; ++i) {
                            const float dx = bX[i] - h.p.x, dy = bY[i] - h.p.y, dz = bZ[i] - h.p.z;
                            const float d2 = dx * dx + dy * dy + dz * dz;
                            const float f = d2 * hscale;
                            const bool hitC = fast && d2 < T2 && (uint32_t)fminf(f, 63.f) == bstar &&
                                              (uint32_t)fminf(fmaxf((f - fb) * 64.f, 0.f), 63.f) == sstar;
                            if (__ballot(hitC)) {
                                float v = hitC ? d2 : INFINITY;   // insertion into the ascending list
#pragma unroll
                                for (int m = 0; m < 8; ++m) { const float lo = fminf(mini[m], v); v = fmaxf(mini[m], v); mini[m] = lo; }
                            }
                        }
                    }
                    if (fast) {
                        const uint32_t need2 = need1 - below2;   // 1 .. inSub
                        float kv = mini[0];
#pragma unroll
                        for (int m = 1; m < 8; ++m) kv = (need2 == (uint32_t)(m + 1)) ? mini[m] : kv;
                        int lessInList = 0;
#pragma unroll
                        for (int m = 0; m < 8; ++m) lessInList += mini[m] < kv ? 1 : 0;
                        kth = __float_as_uint(kv);
                        r2 = kv;
                        tieQuota = k - (int)(below + below2) - lessInList;
                    }
                    // the lanes the histograms could not serve: bisection over the fp32 bit pattern (the heap of PhotonProcess keeps the
                    // nused nearest and leaves maxDistSquared at the farthest of them)
                    const bool slowSel = sel && !fast;
                    if (__ballot(slowSel)) {
                        uint32_t lo = 0u, hi = __float_as_uint(T2);
                        while (__ballot(slowSel && lo < hi)) {   // smallest bit pattern v with count(d2 <= v) >= k
                            const uint32_t mid = lo + ((hi - lo) >> 1);
                            int cnt = 0;
                            for (int i = 0; i < Mb; ++i) {
                                const float dx = bX[i] - h.p.x, dy = bY[i] - h.p.y, dz = bZ[i] - h.p.z;
                                cnt += (__float_as_uint(dx * dx + dy * dy + dz * dz) <= mid) ? 1 : 0;
                            }
                            if (slowSel && lo < hi) { if (cnt >= k) hi = mid; else lo = mid + 1u; }
                        }
                        if (slowSel) {
                            kth = lo;
                            r2 = __uint_as_float(lo);
                            int less = 0;
                            for (int i = 0; i < Mb; ++i) {
                                const float dx = bX[i] - h.p.x, dy = bY[i] - h.p.y, dz = bZ[i] - h.p.z;
                                less += (__float_as_uint(dx * dx + dy * dy + dz * dz) < kth) ? 1 : 0;
                            }
                            tieQuota = k - less;   // photons AT the k-th distance^2 that still belong (bucket order)
                        }
                    }
                }
                // a sample is served when its nused nearest lie inside the searched ball, or the ball is the full one
                const bool served = need && (nIn >= k || !(T2 < S.surf.maxDistSq));
                pending &= ~__ballot(served);
                if (!wasAlone) {
                    // photons on a surface: the count grows with the radius squared.  The next ball aims at 1.6 nused photons for the
                    // sparsest sample (smaller buckets make every pass cheaper), inside the bracket if there is one
                    const int minIn = -(int)wave_max(need ? -(float)nIn : -3.0e9f);
                    const float aim = Rq * sqrtf(SRF_AIM * (float)k / (float)max(minIn, 1));
                    if (__ballot(need && !served)) {   // fewer than nused inside the reduced ball for some sample: a larger one next
                        Rlo = Rq;
                        if (Rhi < INFINITY && Rhi < 1.1f * Rlo) alone = true;
                        else Rtry = fminf(maxDist, Rhi < INFINITY ? fminf(fmaxf(aim, 1.05f * Rq), sqrtf(Rlo * Rhi)) : fmaxf(aim, 1.25f * Rq));
                    } else {   // the next cluster / group starts from the radius that fits this one, with an open bracket
                        Rlo = 0.f; Rhi = INFINITY;
                        Rtry = fminf(maxDist, fmaxf(aim, 0.1f * maxDist));
                    }
                }
                // pass B: Lr = sum over members on the wo side of  kernel / (nPaths r^2) * alpha   (the -wo side carries rho_t == 0)
                const V3 Nf = dot(h.nn, wo) < 0.f ? h.nn * -1.f : h.nn;
                const float norm = 1.f / ((float)S.surf.nCausticPaths * r2);
                // acc[sample][bin] += weight[sample][photon] * alpha[photon][bin] on the matrix pipe (v_mfma_f32_32x32x2_f32 is bit for
                // bit the fmaf chain the per-lane loop made, bucket order kept): A = two flux rows (lane l: bin l & 31 of the photon of
                // its half-wave, one coalesced 128-B row per half-wave), B = the two photons' weights for sample l & 31 -- ONE
                // v_permlane32_swap of (w0, w1) builds both operands --, samples 0..31 into CA, 32..63 into CB.  The round-2 loop
                // read every row through the scalar cache and waited for it, photon by photon: 3.1 s of a 4.9 s C2 frame with the
                // scene's own surface integrator (buckets of 500-2000 caustic photons where the light enters).
                f32x16 CA, CB;
#pragma unroll
                for (int b = 0; b < 16; ++b) { CA[b] = 0.f; CB[b] = 0.f; }
                const int half = lane >> 5, binL = lane & 31;
                typedef const __attribute__((address_space(1))) float gfloat;
                typedef const __attribute__((address_space(1))) nf4 gfloat4;
                gfloat *alphaF = (gfloat *)(S.surf.alpha4);
                gfloat4 *wiG = (gfloat4 *)(S.surf.wi4);
                for (int i0 = 0; i0 < Mb; i0 += 8) {   // four photon pairs per trip: their rows are requested before the first is used
                    float av[4], w0[4], w1[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        av[t] = 0.f; w0[t] = 0.f; w1[t] = 0.f;
                        const int ia = i0 + 2 * t, ib = ia + 1;
                        if (ia >= Mb) continue;   // wave-uniform
                        const bool two = ib < Mb;
                        const uint32_t pa = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(bI[ia]));
                        const uint32_t pb = two ? (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(bI[ib])) : pa;
                        av[t] = alphaF[(size_t)(half ? pb : pa) * 32 + binL];
                        const nf4 wiA = wiG[pa], wiB = wiG[pb];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {   // bucket order: the tie quota is spent in it
                            if (u == 1 && !two) continue;
                            const int i = ia + u;
                            const nf4 wi4 = u ? wiB : wiA;
                            const float dx = bX[i] - h.p.x, dy = bY[i] - h.p.y, dz = bZ[i] - h.p.z;
                            const float d2 = dx * dx + dy * dy + dz * dz;
                            bool mem = served && d2 < T2;
                            if (kth != 0xffffffffu) {
                                const uint32_t bits = __float_as_uint(d2);
                                mem = served && bits < kth;
                                if (served && bits == kth && tieQuota > 0) { mem = true; --tieQuota; }
                            }
                            const float sW = 1.f - d2 / r2;
                            float wgt = (3.f * 0.31830988618379067154f * sW * sW) * norm;
                            if (!mem || !(Nf.x * wi4.x + Nf.y * wi4.y + Nf.z * wi4.z > 0.f)) wgt = 0.f;
                            if (u) w1[t] = wgt; else w0[t] = wgt;
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (i0 + 2 * t >= Mb) continue;   // wave-uniform
                        if (!__ballot(w0[t] != 0.f || w1[t] != 0.f)) continue;
                        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(w0[t]), __float_as_uint(w1[t]), false, false);
                        CA = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], __uint_as_float(r[0]), CA, 0, 0, 0);
                        CB = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], __uint_as_float(r[1]), CB, 0, 0, 0);
                    }
                }
                float acc[32];
#pragma unroll
                for (int b = 0; b < 16; ++b) {   // every lane takes its own sample's column: CA <- bins 8 g + c, CB <- bins 8 g + 4 + c
                    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(CA[b]), __float_as_uint(CB[b]), false, false);
                    acc[8 * (b >> 2) + (b & 3)] = __uint_as_float(r[0]);
                    acc[8 * (b >> 2) + 4 + (b & 3)] = __uint_as_float(r[1]);
                }
#pragma unroll
                for (int b = 0; b < 30; ++b) Ls[b] += acc[b] * mat.kd[b] * 0.31830988618379067154f;   // Lr * rho(wo) * INV_PI, rho == Kd
            }
        }
        // ---- compose: sample = T * Lsurface + Lvi
        if (hit && A.spectral) {   // a segment of the specular recursion: its T is in the record, the sum stays spectral
            float *op = A.out + ri * 60;
#pragma unroll
            for (int b = 0; b < 30; ++b) op[b] += op[30 + b] * Ls[b];
        } else if (hit) {
            const float kT = -1.442695041f * A.tau[ri];
            float x = 0.f, y = 0.f, z = 0.f, sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
            for (int b = 0; b < 30; ++b) {
                const float Tb = __builtin_amdgcn_exp2f((S.sigA[b] + S.sigS[b]) * kT);
                const float v = Tb * Ls[b];
                x += S.cieX[b] * v; y += S.cieY[b] * v; z += S.cieZ[b] * v;
                sx += S.cieX[b] * Ls[b]; sy += S.cieY[b] * Ls[b]; sz += S.cieZ[b] * Ls[b];
            }
            const float scale = float(700 - 400) / float(106.856895f * 30);
            float *op = A.out + ri * 4;
            op[0] += x * scale; op[1] += y * scale; op[2] += z * scale;
            if (A.surfOut && !(A.link && A.link[ri])) { A.surfOut[3 * ri] = sx * scale; A.surfOut[3 * ri + 1] = sy * scale; A.surfOut[3 * ri + 2] = sz * scale; }
        } else if (have && A.surfOut && !A.spectral) {
            A.surfOut[3 * ri] = 0.f; A.surfOut[3 * ri + 1] = 0.f; A.surfOut[3 * ri + 2] = 0.f;
        }
    }
}

extern "C" hipError_t pvol_launch_surface(const SurfArgs *a, uint32_t nWaves, hipStream_t stream) {
    const size_t ldsBytes = (size_t)(SRF_CAP + 4) * 16 + PAINT_CAP * 4;
    hipLaunchKernelGGL(surface_kernel, dim3(nWaves), dim3(LANES), ldsBytes, stream, *a);
    return hipGetLastError();
}
