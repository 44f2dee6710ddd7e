// pvol_shoot.hip -- device photon shooter: PhotonShootingTask::Run / followPhoton
// (core/photonshooter.cpp:47-357) for the VOLUME photon store, hand-written for gfx950.
//
// Execution model: ONE WAVE == ONE VIRTUAL PhotonShootingTask.  Task t owns RNG(31*t) and its own Halton
// permutation (photonshooter.cpp:235,243), exactly as a reference run on that many cores would; a launch advances
// every live task by one 4096-path block, then the host merges the blocks in task order with the running nshot
// scaling (photonshooter.cpp:280-351) -- the reference's mutex-ordered merge made deterministic.  n_tasks == 1 is the
// reference at --ncores 1.
//
// A task is one sequential RNG stream with data-dependent control flow, so the wave's control flow is UNIFORM and
// its 64 lanes work inside one path:
//   * the MT19937 state lives in LDS (regenerated 64 words per step);
//   * spectra are one register: lane b holds bin b (multiplications, exp, stores are one instruction for all 30
//     bins); luminance sums replay the reference's serial order (readlane chain), because `xi > Tr.y()` and the
//     roulette are decisions and a different rounding would send the task down a different path;
//   * the transmittance march (photonshooter.cpp:71-80: one draw + one tau() + one 30-bin exp per stepSize of path)
//     runs 64 STEPS at a time for analytic media, one step per lane, then takes the first lane whose test fires;
//   * closest hit: one triangle per lane, minimum by DPP, ties to the later triangle as the serial scan has them;
//   * followPhoton's recursion is an explicit frame stack in LDS (scalars by lane 0, alpha by the bin lanes);
//   * the six Halton dimensions of a path are computed by six lanes.
// The path is VALU/latency bound; the only compulsory HBM traffic is the 144-B photon record.
#include <stdlib.h>
#include <algorithm>
#include "pvol_rng_dev.h"

#define SH_MAX_DEPTH 24   // frames; deeper recursion aborts the path and is counted
#define NBIN 30
#define SH_STATE_WORDS 640   // per task in global memory: mt[624], mti, totalPaths, pad

// BxDFType bits, core/reflection.h:107-121

struct ShootArgs {
    const DevScene *scene;
    const DevShootScene *shoot;
    uint32_t nTasks;
    const uint32_t *stateIn;  // [nTasks][SH_STATE_WORDS]
    uint32_t *stateOut;       // same layout; a round that has to be redone (block buffer too small) restarts from stateIn
    uint32_t *halton;         // [nTasks][48] permutation tables (bases 2,3,5,7,11,13: 41 entries)
    const uint32_t *flags;    // [nTasks] bit0 causticDone, bit1 indirectDone, bit2 volumeDone, bit3 finished
    float *localPhotons;      // [nTasks][cap][36]: p(3) wi(3) alpha(30)
    uint32_t *localCounts;    // [nTasks][8]: volume, caustic, direct, indirect deposits of this block, surface records kept, radiance photons kept
    uint32_t cap;
    // the surface stores of photonshooter.cpp:148-189, kept only on request (pvol_params.keep_surface_photons): every deposit
    // is one record Photon(p, alpha, wo) with its kind (0 caustic, 1 direct, 2 indirect), in deposit order
    float *localSurf;         // [nTasks][capS][36]: p(3) wo(3) alpha(30)
    uint32_t *localSurfKind;  // [nTasks][capS]
    uint32_t capS;
    float *localRad;          // [nTasks][capR][8]: p(3) n(3) material index, pad  (RadiancePhoton + whose rho it carries)
    uint32_t capR;
    int keepSurface;
    unsigned long long *stats;  // paths, follow_calls, no_hit, march_steps, interactions, absorbed, split_children, overflow
    int init;                 // 1: seed RNG + Halton tables instead of shooting
    uint32_t blockPaths;      // paths per task and round (4096: PhotonShootingTask::Run's block, photonshooter.cpp:247)
    int gridVolume;           // the medium is a VolumeGrid: the kernel takes GRID_KMAX x 64 more LDS words (march_grid)
};

// ------------------------------------------------------------------------------------------ spectra: lane b == bin b
__device__ __forceinline__ float sp_y(float a, float cieYl) {   // core/spectrum.h:433-439, the serial sum
    const float prod = cieYl * a;
    float yy = 0.f;
#pragma unroll
    for (int i = 0; i < NBIN; ++i) yy += lane_f(prod, i);
    return yy * float(700 - 400) / float(106.856895f * NBIN);
}
__device__ __forceinline__ bool sp_black(float a, bool binLane) { return __ballot(binLane && a != 0.f) == 0ull; }
__device__ __forceinline__ int sp_lambda(float a, bool binLane) {   // extractLambda, core/spectrum.h:266-279
    const uint64_t m = __ballot(binLane && a > 0.f);
    if (__popcll(m) != 1) return -1;
    return 400 + (__ffsll((unsigned long long)m) - 1) * 10;
}

// ------------------------------------------------------------------------------------------ scene queries
struct Hit {
    int tri, mat;
    float t, rayEps;
    V3 p, dpdu, nn;
};
// Scene::Intersect (core/scene.h:50-56): closest hit.  The serial scan tightens maxt with `t > maxt` rejections, so of
// several triangles at the same t the LATER one wins: one triangle per lane, minimum over the wave, highest lane of the tie.
__device__ bool scene_closest(const DevScene &S, const DevShootScene &H, V3 o, V3 d, float mint, float *maxt, Hit *hit, int lane) {
    float tk = INFINITY;
    bool any = false;
    int best = -1;
    V3 p1, p2, p3;
    bool flip;
    if (S.bvhNodes) {   // large scene: every lane walks the hierarchy with the same ray (pvol_bvh_dev.h)
        const int slot = bvh_closest(S, o, d, mint, *maxt, &tk);
        if (slot < 0 && !S.nSpheres) return false;
        if (slot < 0) tk = INFINITY;
        else {
        const float4 q1 = S.bvhTris[3 * slot], q2 = S.bvhTris[3 * slot + 1], q3 = S.bvhTris[3 * slot + 2];
        p1 = v3(q1.x, q1.y, q1.z); p2 = v3(q2.x, q2.y, q2.z); p3 = v3(q3.x, q3.y, q3.z);
        best = __float_as_int(q1.w);
        hit->mat = __float_as_int(q2.w);
        flip = __float_as_int(q3.w) != 0;
        }
    } else {
        for (int base = 0; base < S.nTris; base += LANES) {   // at most PVOL_MAX_TRIS == 64: one pass
            float t = 0.f;
            const bool h = (base + lane < S.nTris) && tri_closest(S.tris[base + lane], o, d, mint, *maxt, &t);
            const float tm = -wave_max(h ? -t : -INFINITY);
            if (__ballot(h) && tm <= tk) {
                const uint64_t m = __ballot(h && t == tm);
                tk = tm;
                best = base + 63 - __clzll((unsigned long long)m);
                any = true;
            }
        }
        if (!any && !S.nSpheres) return false;
        if (any) {
            const DevTri &tr = S.tris[best];
            p1 = v3(tr.p1[0], tr.p1[1], tr.p1[2]); p2 = v3(tr.p2[0], tr.p2[1], tr.p2[2]); p3 = v3(tr.p3[0], tr.p3[1], tr.p3[2]);
            hit->mat = H.triMat[best];
            flip = H.triFlip[best] != 0;
        }
    }
    if (S.nSpheres) {   // Shape "sphere" (pvol_sphere_dev.h), tested after the triangles with the ray shortened to their hit
        float ts = best >= 0 ? tk : *maxt;
        V3 ph;
        const int si = spheres_closest(S, o, d, mint, &ts, &ph);
        if (si >= 0) {
            *maxt = ts;
            hit->tri = -1 - si;
            hit->mat = S.spheres[si].mat;
            hit->t = ts;
            hit->rayEps = 5e-4f * ts;   // sphere.cpp:155
            sphere_dg(S.spheres[si], ph, &hit->p, &hit->dpdu, &hit->nn);
            return true;
        }
        if (best < 0) return false;
    }
    *maxt = tk;
    // shapes/trianglemesh.cpp:163-181 with the default uvs (0,0),(1,0),(1,1)
    float du1 = 0.f - 1.f, du2 = 1.f - 1.f, dv1 = 0.f - 1.f, dv2 = 0.f - 1.f;
    V3 dp1 = p1 - p3, dp2 = p2 - p3;
    float determinant = du1 * dv2 - dv1 * du2;
    float invdet = 1.f / determinant;
    hit->dpdu = (dp1 * dv2 - dp2 * dv1) * invdet;
    V3 dpdv = (dp1 * (-du2) + dp2 * du1) * invdet;
    hit->tri = best;
    hit->t = tk;
    hit->p = o + d * tk;
    hit->rayEps = 1e-3f * tk;
    hit->nn = normalize(cross(hit->dpdu, dpdv));   // core/diffgeom.cpp:46-54
    if (flip) hit->nn = hit->nn * -1.f;
    return true;
}

// PhotonVolumeIntegrator::Transmittance with sample == NULL (photonvolume.cpp:15-30): one draw, Exp(-tau), one bin per lane.
// sigTl = sigma_a + sigma_s of the lane's bin (0 on the pad lanes, which therefore return 1).
__device__ float transmittance_bins(const DevScene &S, V3 o, V3 d, float mint, float maxt, Rng &rng, float sigTl, int lane) {
    const float step = 4.f * S.stepSize;
    const float offset = rng_float<true>(rng, lane);
    if (S.volKind == PVOL_VOLUME_NONE) return 1.f;
    RayD r;
    r.o = o; r.d = d; r.mint = mint; r.maxt = maxt;
    if (S.volKind != PVOL_VOLUME_GRID) {   // homogeneous.h:80-84
        float t0, t1, lenAB = 0.f;
        const bool hit = vol_intersect(S, r, &t0, &t1);
        if (hit) { V3 a = o + d * t0, b = o + d * t1; lenAB = len(a - b); }
        const float tau = hit ? lenAB * sigTl : 0.f;
        return expf(-tau);
    }
    // DensityRegion::tau, core/volume.cpp:296-310: tau_b += sigma_t_b * D per sample, then * step
    float t0, t1;
    const float length = len(d);
    if (length == 0.f) return 1.f;
    RayD rn;
    rn.o = o; rn.d = vdiv(d, length); rn.mint = mint * length; rn.maxt = maxt * length;
    if (!vol_intersect(S, rn, &t0, &t1)) return 1.f;
    t0 += offset * step;
    float tau = 0.f;
    while (t0 < t1) {
        const float D = grid_density(S, xform_point(S.w2v, rn.o + rn.d * t0));
        tau += sigTl * D;
        t0 += step;
    }
    return expf(-(tau * step));
}

// core/reflection.cpp:60-67 + 115-135 with scalar indices
__device__ __forceinline__ float fresnel_dielectric(float cosi, float eta_i, float eta_t) {
    cosi = cosi < -1.f ? -1.f : (cosi > 1.f ? 1.f : cosi);
    bool entering = cosi > 0.f;
    float ei = eta_i, et = eta_t;
    if (!entering) { float t = ei; ei = et; et = t; }
    float sint = ei / et * sqrtf(fmaxf(0.f, 1.f - cosi * cosi));
    if (sint >= 1.f) return 1.f;
    float cost = sqrtf(fmaxf(0.f, 1.f - sint * sint));
    float ac = fabsf(cosi);
    float Rparl = ((et * ac) - (ei * cost)) / ((et * ac) + (ei * cost));
    float Rperp = ((ei * ac) - (et * cost)) / ((ei * ac) + (et * cost));
    return (Rparl * Rparl + Rperp * Rperp) / 2.f;
}

__device__ __forceinline__ int num_components(const DevMaterial &m, int flags) {
    int n = 0;
    for (int i = 0; i < m.nBxdf; ++i) if ((m.bxdfType[i] & flags) == m.bxdfType[i]) ++n;
    return n;
}

// ConcentricSampleDisk, core/montecarlo.cpp:306-348
__device__ __forceinline__ void concentric_disk(float u0, float u1, float *dx, float *dy) {
    float r, theta;
    float sx = 2 * u0 - 1, sy = 2 * u1 - 1;
    if (sx == 0.f && sy == 0.f) { *dx = 0.f; *dy = 0.f; return; }
    if (sx >= -sy) {
        if (sx > sy) { r = sx; if (sy > 0.f) theta = sy / r; else theta = 8.0f + sy / r; }
        else { r = sy; theta = 2.0f - sx / r; }
    } else {
        if (sx <= sy) { r = -sx; theta = 4.0f - sy / r; }
        else { r = -sy; theta = 6.0f + sx / r; }
    }
    theta *= K_PI / 4.f;
    *dx = r * cosf(theta);
    *dy = r * sinf(theta);
}

// BSDF::Sample_f with BSDF_ALL (core/reflection.cpp:534-598); `lambda` = extractLambda of the incoming alpha.  The sampled f
// is returned in factored form so the caller can rebuild it per bin in the reference's operation order:
//   fWhich 0: f_b = Kd_b * fFac (Lambertian, fFac = INV_PI)
//          1: f_b = (fFac * Kr_b) / fDiv (specular reflection, fFac = F, fDiv = |cos wi|)
//          2: f_b = (fFac * Kt_b) / fDiv (specular transmission, fFac = 1 - F)
//         -1: black
__device__ void bsdf_sample(const DevMaterial &m, V3 dpdu, V3 nn, V3 woW, float u0, float u1, float ucomp, int lambda,
                            V3 *wiW, float *pdf, int *sampledType, int *fWhich, float *fFac, float *fDiv) {
    *pdf = 0.f; *sampledType = 0; *fWhich = -1; *fFac = 0.f; *fDiv = 1.f;
    int matching = num_components(m, BSDF_ALL);
    if (matching == 0) return;
    int which = min((int)floorf(ucomp * matching), matching - 1);
    int type = m.bxdfType[which];
    V3 sn = normalize(dpdu);
    V3 tn = cross(nn, sn);
    V3 wo = v3(dot(woW, sn), dot(woW, tn), dot(woW, nn));
    V3 wi = v3(0.f, 0.f, 0.f);
    if (type == (BSDF_REFLECTION | BSDF_DIFFUSE)) {
        // BxDF::Sample_f (reflection.cpp:323-330): CosineSampleHemisphere
        float dx, dy;
        concentric_disk(u0, u1, &dx, &dy);
        wi = v3(dx, dy, sqrtf(fmaxf(0.f, 1.f - dx * dx - dy * dy)));
        if (wo.z < 0.f) wi.z *= -1.f;
        *pdf = (wo.z * wi.z > 0.f) ? fabsf(wi.z) * 0.31830988618379067154f : 0.f;
        *fWhich = 0; *fFac = 0.31830988618379067154f;
    } else if (type == (BSDF_REFLECTION | BSDF_SPECULAR)) {
        wi = v3(-wo.x, -wo.y, wo.z);
        *pdf = 1.f;
        *fWhich = 1; *fFac = fresnel_dielectric(wo.z, 1.f, m.ior); *fDiv = fabsf(wi.z);
    } else {
        // SpecularTransmission::Sample_f with the fork's Cauchy dispersion (reflection.cpp:147-182)
        bool entering = wo.z > 0.f;
        float ei = 1.f, et = m.ior;
        if (lambda > 0 && m.vn > 0.f) {
            float l = lambda / 1000.f;
            float B = (float)(((double)((et - 1) / m.vn)) * 0.52345);
            float A = (float)((double)et - ((double)B / 0.34522792));
            et = (float)((double)A + (double)B / pow((double)l, 2.0));
        }
        if (!entering) { float t = ei; ei = et; et = t; }
        float sini2 = fmaxf(0.f, 1.f - wo.z * wo.z);
        float eta = ei / et;
        float sint2 = eta * eta * sini2;
        if (sint2 >= 1.f) return;   // total internal reflection: f = 0, pdf stays 0
        float cost = sqrtf(fmaxf(0.f, 1.f - sint2));
        if (entering) cost = -cost;
        wi = v3(eta * -wo.x, eta * -wo.y, cost);
        *pdf = 1.f;
        float F = fresnel_dielectric(wo.z, 1.f, m.ior);   // undispersed index (reflection.h:331-338)
        *fWhich = 2; *fFac = 1.f - F; *fDiv = fabsf(wi.z);
    }
    if (*pdf == 0.f) { *fWhich = -1; return; }
    *sampledType = type;
    *wiW = v3(sn.x * wi.x + tn.x * wi.y + nn.x * wi.z, sn.y * wi.x + tn.y * wi.y + nn.y * wi.z, sn.z * wi.x + tn.z * wi.y + nn.z * wi.z);
    if (matching > 1) *pdf /= matching;
    if (!(type & BSDF_SPECULAR)) {
        // reflection.cpp:583-592: f re-evaluated over the components on the sampled side; only the
        // Lambertian has a non-zero f()
        int fl = BSDF_ALL;
        if (dot(*wiW, nn) * dot(woW, nn) > 0.f) fl &= ~BSDF_TRANSMISSION; else fl &= ~BSDF_REFLECTION;
        bool lamb = false;
        for (int i = 0; i < m.nBxdf; ++i)
            if ((m.bxdfType[i] & fl) == m.bxdfType[i] && m.bxdfType[i] == (BSDF_REFLECTION | BSDF_DIFFUSE)) lamb = true;
        if (!lamb) { *fWhich = -1; *fFac = 0.f; }
    }
}

// Light::Sample_L(scene, ls, u1, u2, time, &ray, &Ns, &pdf): spot.cpp:106-114, point.cpp:80-88, distant.cpp:82-102
__device__ float light_emit(const DevScene &S, const DevShootScene &H, int ln, float u0, float u1, V3 *o, V3 *d, float *pdf) {
    const DevLight &l = S.lights[ln];
    if (l.kind == PVOL_LIGHT_SPOT) {
        float costheta = (1.f - u0) + u0 * l.cosTotalWidth;   // UniformSampleCone, montecarlo.cpp:405-410
        float sintheta = sqrtf(1.f - costheta * costheta);
        float phi = u1 * 2.f * K_PI;
        V3 v = v3(cosf(phi) * sintheta, sinf(phi) * sintheta, costheta);
        const float *m = H.l2w[ln];
        *o = v3(l.pos[0], l.pos[1], l.pos[2]);
        *d = v3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
        *pdf = 1.f / (2.f * K_PI * (1.f - l.cosTotalWidth));
        // Falloff(ray->d), spot.cpp:60-69
        V3 wl = normalize(v3(l.w2l[0] * d->x + l.w2l[1] * d->y + l.w2l[2] * d->z, l.w2l[4] * d->x + l.w2l[5] * d->y + l.w2l[6] * d->z,
                             l.w2l[8] * d->x + l.w2l[9] * d->y + l.w2l[10] * d->z));
        float ct = wl.z;
        if (ct < l.cosTotalWidth) return 0.f;
        if (ct > l.cosFalloffStart) return 1.f;
        float delta = (ct - l.cosTotalWidth) / (l.cosFalloffStart - l.cosTotalWidth);
        return delta * delta * delta * delta;
    }
    if (l.kind == PVOL_LIGHT_POINT) {
        float z = 1.f - 2.f * u0;   // UniformSampleSphere, montecarlo.cpp:283-290
        float r = sqrtf(fmaxf(0.f, 1.f - z * z));
        float phi = 2.f * K_PI * u1;
        *o = v3(l.pos[0], l.pos[1], l.pos[2]);
        *d = v3(r * cosf(phi), r * sinf(phi), z);
        *pdf = 1.f / (4.f * K_PI);
        return 1.f;
    }
    V3 ld = v3(l.dir[0], l.dir[1], l.dir[2]);
    V3 v1, v2;
    if (fabsf(ld.x) > fabsf(ld.y)) {   // CoordinateSystem, geometry.h:508-519
        float invLen = 1.f / sqrtf(ld.x * ld.x + ld.z * ld.z);
        v1 = v3(-ld.z * invLen, 0.f, ld.x * invLen);
    } else {
        float invLen = 1.f / sqrtf(ld.y * ld.y + ld.z * ld.z);
        v1 = v3(0.f, ld.z * invLen, -ld.y * invLen);
    }
    v2 = cross(ld, v1);
    float dx, dy;
    concentric_disk(u0, u1, &dx, &dy);
    V3 wc = v3(H.worldCenter[0], H.worldCenter[1], H.worldCenter[2]);
    V3 Pdisk = wc + (v1 * dx + v2 * dy) * H.worldRadius;
    *o = Pdisk + ld * H.worldRadius;
    *d = -ld;
    *pdf = 1.f / (K_PI * H.worldRadius * H.worldRadius);
    return 1.f;
}

// PermutedHalton::Sample dimension with the lane's own base (montecarlo.h:206-218): digit loop in double, `n *= invBase`
// truncating through double exactly as the reference does (its quotient can fall one short of n / base; the digit is
// still the true n % base).
__device__ __forceinline__ float halton_lane(uint32_t n, uint32_t base, const uint32_t *perm) {
    double val = 0;
    const double invBase = 1. / base;
    double invBi = invBase;
    while (n > 0) {
        const uint32_t qd = (uint32_t)(n * invBase);
        uint32_t r = n - qd * base;
        if (r >= base) r -= base;
        const uint32_t d_i = perm[r];
        val += d_i * invBi;
        n = qd;
        invBi *= invBase;
    }
    return fminf((float)val, 0x1.fffffep-1f);
}

// One followPhoton activation waiting to be resumed (its callee is running); `alpha` travels beside it, one bin per lane
struct Frame {
    V3 rayO, rayD;       // photonRay as last assigned in this activation
    float rayMint, rayMaxt;   // maxt: the hit distance Scene::Intersect left in the ray, or INFINITY after a reassignment
    Hit hit;             // this activation's photonIsect
    float tag;           // Spectrum::lambda of alpha
    V3 wo;
    int nInt;
    int state;           // 1 = resume at the surface code, 2 = resume the children loop
    int nextChild;       // CHILDREN: next bin (split) or next child ordinal (no split)
    bool split, spec;
};
#define SH_FRAME_WORDS 32
__device__ void frame_push(float *fs, float *fa, int sp, const Frame &F, float alpha, int lane) {
    float *w = fs + sp * SH_FRAME_WORDS;
    if (lane == 0) {
        w[0] = F.rayO.x; w[1] = F.rayO.y; w[2] = F.rayO.z; w[3] = F.rayD.x; w[4] = F.rayD.y; w[5] = F.rayD.z;
        w[6] = F.rayMint; w[7] = F.rayMaxt;
        w[8] = __int_as_float(F.hit.tri); w[9] = F.hit.t; w[10] = F.hit.rayEps;
        w[11] = F.hit.p.x; w[12] = F.hit.p.y; w[13] = F.hit.p.z; w[14] = F.hit.dpdu.x; w[15] = F.hit.dpdu.y; w[16] = F.hit.dpdu.z;
        w[17] = F.hit.nn.x; w[18] = F.hit.nn.y; w[19] = F.hit.nn.z;
        w[20] = F.tag; w[21] = F.wo.x; w[22] = F.wo.y; w[23] = F.wo.z;
        w[24] = __int_as_float(F.nInt); w[25] = __int_as_float(F.state); w[26] = __int_as_float(F.nextChild);
        w[27] = __int_as_float((F.split ? 1 : 0) | (F.spec ? 2 : 0));
        w[28] = __int_as_float(F.hit.mat);
    }
    if (lane < 32) fa[sp * 32 + lane] = alpha;
    __syncthreads();
}
__device__ void frame_pop(const float *fs, const float *fa, int sp, Frame &F, float &alpha, int lane) {
    const float *w = fs + sp * SH_FRAME_WORDS;
    F.rayO = v3(w[0], w[1], w[2]); F.rayD = v3(w[3], w[4], w[5]);
    F.rayMint = w[6]; F.rayMaxt = w[7];
    F.hit.tri = __float_as_int(w[8]); F.hit.mat = __float_as_int(w[28]); F.hit.t = w[9]; F.hit.rayEps = w[10];
    F.hit.p = v3(w[11], w[12], w[13]); F.hit.dpdu = v3(w[14], w[15], w[16]); F.hit.nn = v3(w[17], w[18], w[19]);
    F.tag = w[20]; F.wo = v3(w[21], w[22], w[23]);
    F.nInt = __float_as_int(w[24]); F.state = __float_as_int(w[25]); F.nextChild = __float_as_int(w[26]);
    const int fl = __float_as_int(w[27]);
    F.split = (fl & 1) != 0; F.spec = (fl & 2) != 0;
    alpha = lane < 32 ? fa[sp * 32 + lane] : 0.f;
    __syncthreads();
}

struct PathCtx {
    const DevScene *S;
    const DevShootScene *H;
    Rng rng;
    float *fs, *fa;      // LDS frame stack: scalars, alphas
    float sigAl, sigSl, sigTl, cieYl;   // the lane's bin of sigma_a, sigma_s, their sum, the CIE Y weight (0 on the pad lanes)
    const float *cstT, *cstY;           // LDS: sigma_t[32], cieY[32] for the lane-per-step march
    float *dbuf;                        // LDS, VolumeGrid only: GRID_KMAX x 64 densities (one column per lane) of the lane-per-step march
    bool causticDone, indirectDone, volumeDone;
    float *outPhotons;   // this task's local block buffer
    uint32_t cap;
    float *outSurf; uint32_t *outSurfKind; float *outRad;   // surface stores of this task (null unless kept)
    uint32_t capS, capR, nSurf, nRad;
    uint32_t nVol, nCaustic, nDirect, nIndirect;
    unsigned long long follow, noHit, march, inter, absorbed, splitc, overflow;
};

// The transmittance march of photonshooter.cpp:71-80 for an ANALYTIC medium: `while (t0 < t1) { Tr = Transmittance(ray(t_i .. t0));
// if (xi > Tr.y()) break; t0 += stepSize; }`, 64 iterations at a time, one per lane.  Every iteration draws once (the
// offset inside Transmittance, unused by the analytic tau).  Returns true at an interaction; *t0io = the loop variable there.
__device__ bool march_analytic(PathCtx &C, V3 rayO, V3 dn, float t_i, float *t0io, float t1, float xi, int lane) {
    const DevScene &S = *C.S;
    const float stepSize = C.H->shooterStep;
    float t0 = *t0io;
    while (t0 < t1) {
        // lane j's loop variable: j repeated additions, as the serial loop makes them
        float tj = t0;
        const int span = (int)fminf(63.f, (t1 - t0) / stepSize + 2.f);
        for (int i = 0; i < span; ++i) tj += (i < lane) ? stepSize : 0.f;
        const bool active = (lane <= span) && tj < t1;
        // Tr.y() of lane j: tau = Distance(ray(t_i), ray(t_j)) * sigma_t (homogeneous.h:80-84) inside the extent's slab clip
        RayD r;
        r.o = rayO; r.d = dn; r.mint = t_i; r.maxt = tj;
        float ta, tb, lenAB = 0.f;
        const bool hit = vol_intersect(S, r, &ta, &tb);
        if (hit) { V3 a = rayO + dn * ta, b = rayO + dn * tb; lenAB = len(a - b); }
        float yy = 0.f;
#pragma unroll 6
        for (int b = 0; b < NBIN; ++b) {
            const float tau = hit ? lenAB * C.cstT[b] : 0.f;
            yy += C.cstY[b] * expf(-tau);
        }
        const float y = yy * float(700 - 400) / float(106.856895f * NBIN);
        const uint64_t fire = __ballot(active && xi > y);
        const int nAct = __popcll(__ballot(active));   // the active lanes are a prefix
        if (fire) {
            const int js = __ffsll((unsigned long long)fire) - 1;
            C.march += (unsigned long long)(js + 1);
            rng_skip<true>(C.rng, (unsigned long long)(js + 1), lane);
            *t0io = lane_f(tj, js);
            return true;
        }
        C.march += (unsigned long long)nAct;
        rng_skip<true>(C.rng, (unsigned long long)nAct, lane);
        if (nAct <= span) { *t0io = t1; return false; }   // some lane reached t1: the loop ended
        t0 = lane_f(tj, span) + stepSize;
    }
    *t0io = t0;
    return false;
}

// The same march through a VolumeGrid (round 3; until then one Transmittance() at a time, every lane repeating the same density
// fetches: C4's 2 M-photon map took 67 s).  Here too 64 iterations at a time, one per lane: lane j evaluates
// Transmittance(ray(t_i .. t_j)) = Exp(-DensityRegion::tau) (core/volume.cpp:296-310) with ITS OWN drawn offset -- the values are
// read ahead out of the MT19937 state without consuming them; only as many as the loop really made are skipped afterwards, and
// a trip never reaches across a regeneration.  The decisions must be the serial loop's bit for bit, so the stepped sum is kept
// per bin in the reference's order: the lane's densities D_k go to an LDS column first (the fetches are the expensive part,
// now 64 wide), then tau_b = (sum_k sigma_t,b D_k) step for each bin, and y() in bin order.  A lane that needs more than
// GRID_KMAX densities makes the trip fall back to the serial form (returns -1).
#define GRID_KMAX 48
__device__ int march_grid(PathCtx &C, V3 rayO, V3 dn, float t_i, float *t0io, float t1, float xi, int lane) {
    const DevScene &S = *C.S;
    const float stepSize = C.H->shooterStep;
    const float step = 4.f * S.stepSize;   // PhotonVolumeIntegrator::Transmittance with sample == NULL (photonvolume.cpp:24-27)
    float t0 = *t0io;
    while (t0 < t1) {
        if (C.rng.mti >= MT_N) { mt_regenerate(C.rng.mt, lane); C.rng.mti = 0; }
        const int avail = MT_N - C.rng.mti;
        float tj = t0;
        int span = (int)fminf(63.f, (t1 - t0) / stepSize + 2.f);
        span = min(span, avail - 1);
        for (int i = 0; i < span; ++i) tj += (i < lane) ? stepSize : 0.f;
        const bool active = (lane <= span) && tj < t1;
        // the offset this iteration's Transmittance() would draw: draw number `lane` from here
        uint32_t yv = C.rng.mt[C.rng.mti + min(lane, avail - 1)];
        yv ^= (yv >> 11); yv ^= (yv << 7) & 0x9d2c5680u; yv ^= (yv << 15) & 0xefc60000u; yv ^= (yv >> 18);
        const float offset = (yv & 0xffffff) / float(1 << 24);
        // transmittance_bins' geometry for (mint, maxt) = (t_i, t_j)
        const float length = len(dn);
        RayD rn;
        rn.o = rayO; rn.d = vdiv(dn, length); rn.mint = t_i * length; rn.maxt = tj * length;
        float ta = 0.f, tb = 0.f;
        const bool hit = active && length != 0.f && vol_intersect(S, rn, &ta, &tb);
        int K = 0;
        if (hit) {
            float tt = ta + offset * step;
            while (tt < tb && K < GRID_KMAX) {
                C.dbuf[K * LANES + lane] = grid_density(S, xform_point(S.w2v, rn.o + rn.d * tt));
                ++K;
                tt += step;
            }
            if (tt < tb) K = GRID_KMAX + 1;   // does not fit: serial form for this trip
        }
        if (__ballot(K > GRID_KMAX)) { *t0io = t0; return -1; }
        // (measured on C4's map, 2.05 M photons: 67 s serial -> 22 s in this form; four sample points per trip of the fetch loop with
        // register accumulators for the bins made the 238-VGPR kernel spill: 43 s)
        float yy = 0.f;
        for (int b = 0; b < NBIN; ++b) {
            const float sT = C.cstT[b];
            float tau = 0.f;
            for (int k = 0; k < K; ++k) tau += sT * C.dbuf[k * LANES + lane];
            yy += C.cstY[b] * expf(-(tau * step));   // K == 0 (no overlap with the extent): exp(-0) = 1, as Transmittance returns 1
        }
        const float y = yy * float(700 - 400) / float(106.856895f * NBIN);
        const uint64_t fire = __ballot(active && xi > y);
        const int nAct = __popcll(__ballot(active));   // the active lanes are a prefix
        if (fire) {
            const int js = __ffsll((unsigned long long)fire) - 1;
            C.march += (unsigned long long)(js + 1);
            rng_skip<true>(C.rng, (unsigned long long)(js + 1), lane);
            *t0io = lane_f(tj, js);
            return 1;
        }
        C.march += (unsigned long long)nAct;
        rng_skip<true>(C.rng, (unsigned long long)nAct, lane);
        if (nAct <= span) { *t0io = t1; return 0; }   // some lane reached t1: the loop ended
        t0 = lane_f(tj, span) + stepSize;
    }
    *t0io = t0;
    return 0;
}

__device__ void follow_photon(PathCtx &C, V3 rayO, V3 rayD, float rayMint, float alpha, float tag, int lane) {
    const DevScene &S = *C.S;
    const DevShootScene &H = *C.H;
    const bool binLane = lane < NBIN;
    int sp = 0;
    // "call arguments" of the activation being entered
    int nInt = 0;
    bool spec = true;
    int mode = 0;   // 0 = CALL, 1 = SURFACE, 2 = CHILDREN, 3 = RETURN
    Frame F;        // the activation in SURFACE / CHILDREN mode
    float Falpha = 0.f;
    F.hit.tri = 0; F.hit.mat = 0; F.hit.t = 0.f; F.hit.rayEps = 0.f; F.hit.p = F.hit.dpdu = F.hit.nn = v3(0.f, 0.f, 0.f);
    F.rayO = F.rayD = F.wo = v3(0.f, 0.f, 0.f); F.rayMint = F.rayMaxt = F.tag = 0.f; F.nInt = F.state = F.nextChild = 0; F.split = F.spec = false;
    for (;;) {
        if (mode == 0) {
            // ---- followPhoton entry (photonshooter.cpp:54-128)
            ++C.follow;
            float maxt = INFINITY;
            Hit h;
            if (!scene_closest(S, H, rayO, rayD, rayMint, &maxt, &h, lane)) { ++C.noHit; mode = 3; continue; }
            ++nInt;
            float length = len(rayD);
            if (length == 0.f) { mode = 3; continue; }
            RayD rn;
            rn.o = rayO; rn.d = vdiv(rayD, length); rn.mint = rayMint * length; rn.maxt = maxt * length;
            float t0, t1;
            if (S.volKind == PVOL_VOLUME_NONE || !vol_intersect(S, rn, &t0, &t1)) { t0 = 1.0f; t1 = 0.0f; }
            t0 += rng_float<true>(C.rng, lane) * H.shooterStep;
            const float t_i = t0;
            const float xi = rng_float<true>(C.rng, lane);
            bool interaction = false;
            if (S.volKind != PVOL_VOLUME_GRID) {
                interaction = march_analytic(C, rayO, rn.d, t_i, &t0, t1, xi, lane);
            } else {
                int g = march_grid(C, rayO, rn.d, t_i, &t0, t1, xi, lane);
                while (g < 0) {   // a segment with more density samples than the LDS columns hold: one serial step, then try again
                    if (!(t0 < t1)) { g = 0; break; }
                    ++C.march;
                    const float tr = transmittance_bins(S, rayO, rn.d, t_i, t0, C.rng, C.sigTl, lane);
                    if (xi > sp_y(tr, C.cieYl)) { g = 1; break; }
                    t0 += H.shooterStep;
                    g = march_grid(C, rayO, rn.d, t_i, &t0, t1, xi, lane);
                }
                interaction = g == 1;
            }
            bool toSurface = true;
            if (interaction) {
                ++C.inter;
                V3 ip = rn.o + rn.d * t0;
                float dens = vol_density(S, ip);
                const float ss = C.sigSl * dens, sa = C.sigAl * dens;
                float ys = sp_y(ss, C.cieYl), ya = sp_y(sa, C.cieYl);
                bool scatter = (rng_float<true>(C.rng, lane) > ys / (ya + ys));
                if (!scatter) { ++C.absorbed; mode = 3; continue; }
                if (!C.volumeDone) {
                    if (nInt > 1) {
                        if (C.nVol < C.cap) {
                            float *o = C.outPhotons + (size_t)C.nVol * 36;
                            if (lane < 6) o[lane] = lane == 0 ? ip.x : lane == 1 ? ip.y : lane == 2 ? ip.z : lane == 3 ? rn.d.x : lane == 4 ? rn.d.y : rn.d.z;
                            if (binLane) o[6 + lane] = alpha;
                        }   // a full block buffer is seen by the host in the count (it redoes the round with a larger one)
                        ++C.nVol;
                    }
                    float u1 = rng_float<true>(C.rng, lane), u2 = rng_float<true>(C.rng, lane);
                    float z = 1.f - 2.f * u1;   // UniformSampleSphere
                    float r = sqrtf(fmaxf(0.f, 1.f - z * z));
                    float phi = 2.f * K_PI * u2;
                    V3 dir = v3(r * cosf(phi), r * sinf(phi), z);
                    float pdf = 1.f / (4.f * K_PI);
                    float ref = vol_phase(S, ip, rn.d, dir);
                    if (ref == 0.f) { mode = 3; continue; }
                    alpha *= ref;
                    alpha /= pdf;
                    // the outer activation resumes at its surface code with the REASSIGNED ray (photonshooter.cpp:123-133)
                    if (sp >= SH_MAX_DEPTH) { ++C.overflow; mode = 3; continue; }
                    Frame R;
                    R.rayO = ip; R.rayD = dir; R.rayMint = 0.f; R.rayMaxt = INFINITY; R.hit = h; R.tag = tag; R.nInt = nInt; R.spec = spec;
                    R.state = 1; R.nextChild = 0; R.split = false; R.wo = v3(0.f, 0.f, 0.f);
                    frame_push(C.fs, C.fa, sp++, R, alpha, lane);
                    rayO = ip; rayD = dir; rayMint = 0.f;   // callee arguments
                    toSurface = false;
                    mode = 0;
                }
            }
            if (toSurface) {
                F.rayO = rayO; F.rayD = rayD; F.rayMint = rayMint; F.rayMaxt = maxt; F.hit = h; Falpha = alpha; F.tag = tag; F.nInt = nInt; F.spec = spec;
                mode = 1;
            }
            continue;
        }
        if (mode == 1) {
            // ---- surface code (photonshooter.cpp:131-197), F holds the activation
            const float tr = transmittance_bins(S, F.rayO, F.rayD, F.rayMint, F.rayMaxt, C.rng, C.sigTl, lane);
            Falpha *= tr;
            const DevMaterial &m = H.mats[F.hit.mat];
            bool hasNonSpecular = m.nBxdf > num_components(m, BSDF_REFLECTION | BSDF_TRANSMISSION | BSDF_SPECULAR);
            bool hasTransmission = num_components(m, BSDF_TRANSMISSION | BSDF_DIFFUSE | BSDF_GLOSSY | BSDF_SPECULAR) > 0;
            bool dispersive = (m.kind == PVOL_MATERIAL_GLASS && m.vn > 0.f);
            F.split = hasTransmission && F.tag < 0.f && dispersive;
            if (F.split) C.splitc += (unsigned long long)__popcll(__ballot(binLane && Falpha != 0.f));
            F.wo = -F.rayD;
            if (hasNonSpecular) {
                bool deposited = false;
                int kind = 0;
                if (F.spec && F.nInt > 1) {
                    if (!C.causticDone) { deposited = true; ++C.nCaustic; kind = 0; }
                } else {
                    if (F.nInt == 1 && !C.indirectDone && H.finalGather) { deposited = true; ++C.nDirect; kind = 1; }
                    else if (F.nInt > 1 && !C.indirectDone) { deposited = true; ++C.nIndirect; kind = 2; }
                }
                if (deposited && C.outSurf) {   // Photon(photonIsect.dg.p, alpha, wo), photonshooter.cpp:150
                    if (C.nSurf < C.capS) {
                        float *o = C.outSurf + (size_t)C.nSurf * 36;
                        if (lane < 6) o[lane] = lane == 0 ? F.hit.p.x : lane == 1 ? F.hit.p.y : lane == 2 ? F.hit.p.z : lane == 3 ? F.wo.x : lane == 4 ? F.wo.y : F.wo.z;
                        if (binLane) o[6 + lane] = Falpha;
                        if (lane == 0) C.outSurfKind[C.nSurf] = (uint32_t)kind;
                    }
                    ++C.nSurf;
                }
                if (deposited && H.finalGather && rng_float<true>(C.rng, lane) < .125f) {
                    rng_skip<true>(C.rng, 288ull, lane);   // 2 x BSDF::rho (reflection.cpp:647-658)
                    if (C.outRad) {   // RadiancePhoton(p, Faceforward(nn, -photonRay.d)), photonshooter.cpp:182-189
                        if (C.nRad < C.capR) {
                            const V3 nf = dot(F.hit.nn, F.wo) < 0.f ? F.hit.nn * -1.f : F.hit.nn;
                            float *o = C.outRad + (size_t)C.nRad * 8;
                            if (lane < 8) o[lane] = lane == 0 ? F.hit.p.x : lane == 1 ? F.hit.p.y : lane == 2 ? F.hit.p.z : lane == 3 ? nf.x : lane == 4 ? nf.y :
                                                    lane == 5 ? nf.z : lane == 6 ? __int_as_float(F.hit.mat) : 0.f;
                        }
                        ++C.nRad;
                    }
                }
            }
            if (F.nInt >= H.maxPhotonDepth) { mode = 3; continue; }
            F.nextChild = 0;
            mode = 2;
            continue;
        }
        if (mode == 2) {
            // ---- children loop (photonshooter.cpp:199-227)
            bool called = false;
            for (;;) {
                float a;
                if (F.split) {
                    const uint64_t nz = __ballot(binLane && Falpha != 0.f) & ~((1ull << F.nextChild) - 1ull);
                    if (!nz) break;
                    const int b = __ffsll((unsigned long long)nz) - 1;
                    F.nextChild = b + 1;
                    a = (lane == b) ? Falpha : 0.f;   // splitSpectrum child: one bin, core/spectrum.cpp:100-111
                } else {
                    if (F.nextChild != 0) break;
                    F.nextChild = 1;
                    a = Falpha;
                }
                float ud0 = rng_float<true>(C.rng, lane), ud1 = rng_float<true>(C.rng, lane), uc = rng_float<true>(C.rng, lane);
                const DevMaterial &m = H.mats[F.hit.mat];
                V3 wi;
                float pdf, fFac, fDiv;
                int flags, fWhich;
                bsdf_sample(m, F.hit.dpdu, F.hit.nn, F.wo, ud0, ud1, uc, sp_lambda(a, binLane), &wi, &pdf, &flags, &fWhich, &fFac, &fDiv);
                if (fWhich < 0 || pdf == 0.f) continue;
                const float *K = fWhich == 0 ? m.kd : (fWhich == 1 ? m.kr : m.kt);
                const float Kl = binLane ? K[lane] : 0.f;
                float absdot = fabsf(dot(wi, F.hit.nn));
                const float fr = (fWhich == 0) ? Kl * fFac : (fFac * Kl) / fDiv;
                if (sp_black(fr, binLane)) continue;
                const float anew = binLane ? a * fr * absdot / pdf : 0.f;   // alpha * fr * AbsDot(wi, nn) / pdf (photonshooter.cpp:208-209)
                float continueProb = fminf(1.f, sp_y(anew, C.cieYl) / sp_y(a, C.cieYl));
                if (rng_float<true>(C.rng, lane) > continueProb) continue;
                const float a2 = anew / continueProb;
                float tag2 = (float)sp_lambda(a2, binLane);
                F.spec = F.spec && ((flags & BSDF_SPECULAR) != 0);
                if (C.indirectDone && !F.spec) continue;
                if (sp >= SH_MAX_DEPTH) { ++C.overflow; continue; }
                F.state = 2;
                frame_push(C.fs, C.fa, sp++, F, Falpha, lane);
                rayO = F.hit.p; rayD = wi; rayMint = F.hit.rayEps;
                alpha = a2; tag = tag2; nInt = F.nInt; spec = F.spec;
                called = true;
                break;
            }
            mode = called ? 0 : 3;
            continue;
        }
        // ---- RETURN
        if (sp == 0) return;
        frame_pop(C.fs, C.fa, --sp, F, Falpha, lane);
        mode = F.state;
    }
}

template <int WPE>
__global__ __launch_bounds__(LANES, WPE) void shoot_kernel(ShootArgs A) {
    extern __shared__ __align__(16) unsigned char lds[];
    const uint32_t task = blockIdx.x;
    const int lane = threadIdx.x;
    if (task >= A.nTasks) return;
    const DevScene &S = *A.scene;
    const DevShootScene &H = *A.shoot;
    uint32_t *mt = reinterpret_cast<uint32_t *>(lds);
    float *cst = reinterpret_cast<float *>(lds + MT_N * 4);          // sigT[32] | cieY[32]
    uint32_t *perm = reinterpret_cast<uint32_t *>(cst + 64);         // 48 words
    float *fs = reinterpret_cast<float *>(perm + 48);                // SH_MAX_DEPTH x SH_FRAME_WORDS
    float *fa = fs + SH_MAX_DEPTH * SH_FRAME_WORDS;                  // SH_MAX_DEPTH x 32
    float *dbuf = fa + SH_MAX_DEPTH * 32;                            // VolumeGrid only: GRID_KMAX x 64 (march_grid)
    uint32_t *haltonG = A.halton + (size_t)task * 48;
    Rng rng;
    rng.mt = mt;
    rng.draws = 0;
    if (A.init) {
        // RNG rng(31 * taskNum) (photonshooter.cpp:235), then PermutedHalton(6, rng) (montecarlo.cpp:380-397)
        mt_seed(mt, 31u * task, lane);
        rng.mti = MT_N;
        const uint32_t bases[6] = {2, 3, 5, 7, 11, 13};
        uint32_t off = 0;
        for (int d = 0; d < 6; ++d) {
            const uint32_t b = bases[d];
            if ((uint32_t)lane < b) perm[off + lane] = (uint32_t)lane;
            __syncthreads();
            for (uint32_t i = 0; i < b; ++i) {   // Shuffle(buf, b, 1, rng), montecarlo.h:174-181
                const uint32_t other = i + (rng_uint<true>(rng, lane) % (b - i));
                if (lane == 0) { const uint32_t t = perm[off + i]; perm[off + i] = perm[off + other]; perm[off + other] = t; }
                __syncthreads();
            }
            off += b;
        }
        if (lane < 48) haltonG[lane] = lane < 41 ? perm[lane] : 0u;
        uint32_t *so = A.stateOut + (size_t)task * SH_STATE_WORDS;
        for (int i = lane; i < MT_N; i += LANES) so[i] = mt[i];
        if (lane == 0) { so[MT_N] = (uint32_t)rng.mti; so[MT_N + 1] = 0u; }
        return;
    }
    const uint32_t fl = A.flags[task];
    uint32_t *lc = A.localCounts + (size_t)task * 8;
    const uint32_t *si = A.stateIn + (size_t)task * SH_STATE_WORDS;
    uint32_t *so = A.stateOut + (size_t)task * SH_STATE_WORDS;
    if (fl & 8u) {   // finished: the state is carried over untouched
        if (lane < 8) lc[lane] = 0u;
        for (int i = lane; i < MT_N + 2; i += LANES) so[i] = si[i];
        return;
    }
    for (int i = lane; i < MT_N; i += LANES) mt[i] = si[i];
    if (lane < 48) perm[lane] = haltonG[lane];
    const bool binLane = lane < NBIN;
    const float sigAl = binLane ? S.sigA[lane] : 0.f, sigSl = binLane ? S.sigS[lane] : 0.f;
    if (lane < 32) { cst[lane] = sigAl + sigSl; cst[32 + lane] = binLane ? S.cieY[lane] : 0.f; }
    __syncthreads();
    rng.mti = (int)si[MT_N];
    uint32_t totalPaths = si[MT_N + 1];
    PathCtx C;
    C.S = &S; C.H = &H; C.rng = rng; C.fs = fs; C.fa = fa;
    C.sigAl = sigAl; C.sigSl = sigSl; C.sigTl = sigAl + sigSl; C.cieYl = binLane ? S.cieY[lane] : 0.f;
    C.cstT = cst; C.cstY = cst + 32;
    C.dbuf = dbuf;
    C.causticDone = fl & 1u; C.indirectDone = fl & 2u; C.volumeDone = fl & 4u;
    C.outPhotons = A.localPhotons + (size_t)task * A.cap * 36;
    C.cap = A.cap;
    C.outSurf = A.keepSurface ? A.localSurf + (size_t)task * A.capS * 36 : 0;
    C.outSurfKind = A.keepSurface ? A.localSurfKind + (size_t)task * A.capS : 0;
    C.outRad = A.keepSurface ? A.localRad + (size_t)task * A.capR * 8 : 0;
    C.capS = A.capS; C.capR = A.capR; C.nSurf = C.nRad = 0;
    C.nVol = C.nCaustic = C.nDirect = C.nIndirect = 0;
    C.follow = C.noHit = C.march = C.inter = C.absorbed = C.splitc = C.overflow = 0;
    const uint32_t blockSize = A.blockPaths;
    unsigned long long paths = 0;
    // the lane's Halton dimension: base and offset of its permutation table (lanes 0..5)
    const uint32_t hBase = lane == 0 ? 2u : lane == 1 ? 3u : lane == 2 ? 5u : lane == 3 ? 7u : lane == 4 ? 11u : 13u;
    const uint32_t hOff = lane == 0 ? 0u : lane == 1 ? 2u : lane == 2 ? 5u : lane == 3 ? 10u : lane == 4 ? 17u : 28u;
    for (uint32_t i = 0; i < blockSize; ++i) {
        ++totalPaths;
        ++paths;
        const float um = lane < 6 ? halton_lane(totalPaths, hBase, perm + hOff) : 0.f;
        const float u0 = lane_f(um, 0), u1 = lane_f(um, 1), u2 = lane_f(um, 2);   // u[3..5] are drawn and never read (photonshooter.cpp:247-262)
        // Distribution1D::SampleDiscrete (montecarlo.h:99-107): upper_bound over cdf[0..n]
        int n = S.nLights;
        int ub = 0;
        while (ub < n + 1 && !(u0 < H.lightCdf[ub])) ++ub;
        int ln = max(0, ub - 1);
        if (ln >= n) ln = n - 1;
        float lightPdf = H.lightFunc[ln] / (H.lightFuncInt * n);
        V3 o, d;
        float pdf;
        float scale = light_emit(S, H, ln, u1, u2, &o, &d, &pdf);
        const DevLight &L = S.lights[ln];
        const float Il = binLane ? L.intensity[lane] : 0.f;
        const float Le = (L.kind == PVOL_LIGHT_SPOT) ? Il * scale : Il;
        if (pdf == 0.f || sp_black(Le, binLane)) continue;
        // alpha = (AbsDot(Nl, d) * Le) / (pdf * lightPdf), Nl == d (photonshooter.cpp:264)
        float ad = fabsf(dot(d, d));
        float den = pdf * lightPdf;
        const float alpha = binLane ? (Le * ad) / den : 0.f;
        if (sp_black(alpha, binLane)) continue;
        follow_photon(C, o, d, 0.f, alpha, (float)sp_lambda(alpha, binLane), lane);
    }
    __syncthreads();
    for (int i = lane; i < MT_N; i += LANES) so[i] = mt[i];
    if (lane == 0) {
        so[MT_N] = (uint32_t)C.rng.mti;
        so[MT_N + 1] = totalPaths;
        lc[0] = C.nVol; lc[1] = C.nCaustic; lc[2] = C.nDirect; lc[3] = C.nIndirect; lc[4] = C.nSurf; lc[5] = C.nRad; lc[6] = lc[7] = 0u;
        atomicAdd(&A.stats[0], paths);
        atomicAdd(&A.stats[1], C.follow);
        atomicAdd(&A.stats[2], C.noHit);
        atomicAdd(&A.stats[3], C.march);
        atomicAdd(&A.stats[4], C.inter);
        atomicAdd(&A.stats[5], C.absorbed);
        atomicAdd(&A.stats[6], C.splitc);
        atomicAdd(&A.stats[7], C.overflow);
    }
}

// merge of one task's block into the global photon arrays: alpha /= float(nshot) with the RUNNING nshot
// of that task's turn (photonshooter.cpp:333)
struct MergeArgs {
    const float *localPhotons;
    uint32_t cap;
    const uint32_t *srcTask;   // per merged segment: task, count, destination offset, nshot
    const uint32_t *count;
    const uint32_t *dstOff;
    const float *nshot;
    uint32_t nSeg;
    float *p, *wi, *alpha;     // destination raw arrays
};
__global__ void merge_kernel(MergeArgs M) {
    for (uint32_t seg = blockIdx.x; seg < M.nSeg; seg += gridDim.x) {
        const uint32_t cnt = M.count[seg];
        const float *src = M.localPhotons + (size_t)M.srcTask[seg] * M.cap * 36;
        const float ns = M.nshot[seg];
        for (uint32_t i = threadIdx.x; i < cnt * 36; i += blockDim.x) {
            uint32_t ph = i / 36, f = i - ph * 36;
            float v = src[(size_t)ph * 36 + f];
            size_t dst = (size_t)M.dstOff[seg] + ph;
            if (f < 3) M.p[dst * 3 + f] = v;
            else if (f < 6) M.wi[dst * 3 + (f - 3)] = v;
            else M.alpha[dst * 30 + (f - 6)] = v / ns;
        }
    }
}

// merge of one task's surface records into the per-kind arrays, in deposit order (photonshooter.cpp:303-327), and of its
// radiance photons (:341-349).  One wave per segment; `take` has bit k set when kind k is merged at this task's turn.
struct SurfMergeArgs {
    const float *localSurf; const uint32_t *localSurfKind; uint32_t capS;
    const float *localRad; uint32_t capR;
    const uint32_t *srcTask, *nSurf, *take, *dstOff;   // dstOff: [nSeg][4] = caustic, direct, indirect, radiance
    const uint32_t *nRad;
    uint32_t nSeg;
    float *p[3], *wo[3], *alpha[3];
    float *rad;   // [n][8]
};
__global__ __launch_bounds__(64) void merge_surface_kernel(SurfMergeArgs M) {
    const int lane = threadIdx.x;
    for (uint32_t seg = blockIdx.x; seg < M.nSeg; seg += gridDim.x) {
        const uint32_t task = M.srcTask[seg], n = M.nSurf[seg], take = M.take[seg];
        const float *src = M.localSurf + (size_t)task * M.capS * 36;
        const uint32_t *kinds = M.localSurfKind + (size_t)task * M.capS;
        uint32_t at[3] = {M.dstOff[4 * seg], M.dstOff[4 * seg + 1], M.dstOff[4 * seg + 2]};
        for (uint32_t base = 0; base < n; base += 64) {
            const uint32_t i = base + lane;
            const uint32_t kd = i < n ? kinds[i] : 3u;
#pragma unroll
            for (uint32_t k = 0; k < 3; ++k) {
                const bool mine = kd == k && ((take >> k) & 1u);
                const uint64_t m = __ballot(mine);
                if (mine) {
                    const size_t dst = (size_t)at[k] + __popcll(m & ((1ull << lane) - 1ull));
                    const float *r = src + (size_t)i * 36;
                    for (int f = 0; f < 3; ++f) { M.p[k][dst * 3 + f] = r[f]; M.wo[k][dst * 3 + f] = r[3 + f]; }
                    for (int f = 0; f < 30; ++f) M.alpha[k][dst * 30 + f] = r[6 + f];
                }
                at[k] += (uint32_t)__popcll(m);
            }
        }
        const uint32_t nr = M.nRad[seg];
        const float *rs = M.localRad + (size_t)task * M.capR * 8;
        for (uint32_t i = lane; i < nr * 8; i += 64) M.rad[(size_t)M.dstOff[4 * seg + 3] * 8 + i] = rs[i];
    }
}
extern "C" hipError_t pvol_launch_merge_surface(const SurfMergeArgs *m, hipStream_t stream) {
    if (!m->nSeg) return hipSuccess;
    hipLaunchKernelGGL(merge_surface_kernel, dim3(std::min<uint32_t>(m->nSeg, 4096u)), dim3(64), 0, stream, *m);
    return hipGetLastError();
}

extern "C" size_t pvol_shoot_state_words(void) { return SH_STATE_WORDS; }

extern "C" hipError_t pvol_launch_shoot(const ShootArgs *a, hipStream_t stream) {
    const size_t ldsBytes = MT_N * 4 + 64 * 4 + 48 * 4 + (size_t)SH_MAX_DEPTH * SH_FRAME_WORDS * 4 + (size_t)SH_MAX_DEPTH * 32 * 4 +
                            (a->gridVolume ? (size_t)GRID_KMAX * LANES * 4 : 0);
    // registers: the path state is ~240 wave-uniform values; 2 waves per SIMD hold them without spills, 4 spill ~160 of
    // them to scratch but hide more latency (PVOL_SHOOT_WPE picks; measured in profiles/)
    static const int wpe = [] { const char *e = getenv("PVOL_SHOOT_WPE"); const int v = e ? atoi(e) : 2; return (v == 4 || v == 3) ? v : 2; }();
    if (wpe == 4) hipLaunchKernelGGL((shoot_kernel<4>), dim3(a->nTasks), dim3(LANES), ldsBytes, stream, *a);
    else if (wpe == 3) hipLaunchKernelGGL((shoot_kernel<3>), dim3(a->nTasks), dim3(LANES), ldsBytes, stream, *a);
    else hipLaunchKernelGGL((shoot_kernel<2>), dim3(a->nTasks), dim3(LANES), ldsBytes, stream, *a);
    return hipGetLastError();
}
extern "C" hipError_t pvol_launch_merge(const MergeArgs *m, hipStream_t stream) {
    if (!m->nSeg) return hipSuccess;
    hipLaunchKernelGGL(merge_kernel, dim3(std::min<uint32_t>(m->nSeg, 4096u)), dim3(256), 0, stream, *m);
    return hipGetLastError();
}
