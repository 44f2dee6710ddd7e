// pvol_shoot.hip -- device photon shooter: PhotonShootingTask::Run / followPhoton
// (core/photonshooter.cpp:47-357) for the VOLUME photon store, hand-written for gfx950.
//
// Execution model: ONE LANE == ONE VIRTUAL PhotonShootingTask.  Task t owns RNG(31*t) and its own
// Halton permutation (photonshooter.cpp:235,243), exactly as a reference run on that many cores would;
// a launch advances every live task by one 4096-path block, then the host merges the blocks in task
// order with the running nshot scaling (photonshooter.cpp:280-351) -- the reference's mutex-ordered
// merge made deterministic.  n_tasks == 1 is the reference at --ncores 1.
// The path is VALU/latency/divergence bound: per stored photon hundreds of transmittance
// march steps (30 exp + a luminance dot each) and closest-hit queries against a handful of triangles;
// the only compulsory HBM traffic is the 144-B photon record.  Per-lane state: MT19937 words in global
// memory laid out [word][task] so a wave touches whole lines, photon weight alpha[30] in registers,
// followPhoton's recursion as an explicit frame stack in scratch.
#include "pvol_math.h"

#define SH_MAX_DEPTH 24   // frames; deeper recursion aborts the path and is counted
#define NBIN 30

// BxDFType bits, core/reflection.h:107-121
#define BSDF_REFLECTION 1
#define BSDF_TRANSMISSION 2
#define BSDF_DIFFUSE 4
#define BSDF_GLOSSY 8
#define BSDF_SPECULAR 16
#define BSDF_ALL 31

struct ShootArgs {
    const DevScene *scene;
    const DevShootScene *shoot;
    uint32_t nTasks;
    uint32_t *mt;            // [625][nTasks]: words 0..623, word 624 = mti
    uint32_t *halton;        // [41][nTasks] permutation tables (bases 2,3,5,7,11,13)
    uint32_t *totalPaths;    // [nTasks]
    uint32_t *flags;         // [nTasks] bit0 causticDone, bit1 indirectDone, bit2 volumeDone, bit3 finished
    float *localPhotons;     // [nTasks][cap][36]: p(3) wi(3) alpha(30)
    uint32_t *localCounts;   // [nTasks][4]: volume, caustic, direct, indirect stored in this block
    uint32_t cap;
    unsigned long long *stats;  // paths, follow_calls, no_hit, march_steps, interactions, absorbed, split_children, overflow
    int init;                // 1: seed RNG + Halton tables instead of shooting
};

// ------------------------------------------------------------------------------------------ per-lane RNG
struct LaneRng {
    uint32_t *mt;     // base + task; stride nTasks
    uint32_t stride;
    int mti;
};
__device__ __forceinline__ uint32_t &MTW(LaneRng &r, int i) { return r.mt[(size_t)i * r.stride]; }
__device__ void lane_regen(LaneRng &r) {   // core/rng.cpp:80-92
    const uint32_t A = 0x9908b0dfu, UP = 0x80000000u, LO = 0x7fffffffu;
    int kk;
    uint32_t y;
    for (kk = 0; kk < MT_N - MT_M; kk++) { y = (MTW(r, kk) & UP) | (MTW(r, kk + 1) & LO); MTW(r, kk) = MTW(r, kk + MT_M) ^ (y >> 1) ^ ((y & 1u) ? A : 0u); }
    for (; kk < MT_N - 1; kk++) { y = (MTW(r, kk) & UP) | (MTW(r, kk + 1) & LO); MTW(r, kk) = MTW(r, kk + (MT_M - MT_N)) ^ (y >> 1) ^ ((y & 1u) ? A : 0u); }
    y = (MTW(r, MT_N - 1) & UP) | (MTW(r, 0) & LO);
    MTW(r, MT_N - 1) = MTW(r, MT_M - 1) ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
    r.mti = 0;
}
__device__ __forceinline__ uint32_t lane_uint(LaneRng &r) {
    if (r.mti >= MT_N) lane_regen(r);
    uint32_t y = MTW(r, r.mti++);
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
__device__ __forceinline__ float lane_float(LaneRng &r) { return (lane_uint(r) & 0xffffff) / float(1 << 24); }
__device__ __forceinline__ void lane_skip(LaneRng &r, int n) { for (int i = 0; i < n; ++i) (void)lane_uint(r); }

// ------------------------------------------------------------------------------------------ per-lane spectra
struct Spec { float c[NBIN]; };
__device__ __forceinline__ float sp_y(const DevScene &S, const Spec &a) {   // core/spectrum.h:433-439
    float yy = 0.f;
#pragma unroll
    for (int i = 0; i < NBIN; ++i) yy += S.cieY[i] * a.c[i];
    return yy * float(700 - 400) / float(106.856895f * NBIN);
}
__device__ __forceinline__ bool sp_black(const Spec &a) {
    bool b = true;
#pragma unroll
    for (int i = 0; i < NBIN; ++i) b = b && (a.c[i] == 0.f);
    return b;
}
__device__ __forceinline__ int sp_lambda(const Spec &a) {   // extractLambda, core/spectrum.h:266-279
    bool first = true;
    int l = -1;
#pragma unroll
    for (int i = 0; i < NBIN; ++i) {
        if (a.c[i] > 0.f && !first) return -1;
        if (a.c[i] > 0.f && first) { l = 400 + i * 10; first = false; }
    }
    return l;
}

// ------------------------------------------------------------------------------------------ scene queries
struct Hit {
    int tri;
    float t, rayEps;
    V3 p, dpdu, nn;
};
// Scene::Intersect (core/scene.h:50-56): closest hit, linear scan (ties in t: the later triangle wins,
// like a maxt that is only tightened with `t > maxt` rejections)
__device__ bool scene_closest(const DevScene &S, const DevShootScene &H, V3 o, V3 d, float mint, float *maxt, Hit *hit) {
    bool any = false;
    float mt = *maxt;
    int best = -1;
    for (int i = 0; i < S.nTris; ++i) {
        float t;
        if (!tri_closest(S.tris[i], o, d, mint, mt, &t)) continue;
        any = true;
        mt = t;
        best = i;
    }
    if (!any) return false;
    *maxt = mt;
    const DevTri &tr = S.tris[best];
    V3 p1 = v3(tr.p1[0], tr.p1[1], tr.p1[2]), p2 = v3(tr.p2[0], tr.p2[1], tr.p2[2]), p3 = v3(tr.p3[0], tr.p3[1], tr.p3[2]);
    // shapes/trianglemesh.cpp:163-181 with the default uvs (0,0),(1,0),(1,1)
    float du1 = 0.f - 1.f, du2 = 1.f - 1.f, dv1 = 0.f - 1.f, dv2 = 0.f - 1.f;
    V3 dp1 = p1 - p3, dp2 = p2 - p3;
    float determinant = du1 * dv2 - dv1 * du2;
    float invdet = 1.f / determinant;
    hit->dpdu = (dp1 * dv2 - dp2 * dv1) * invdet;
    V3 dpdv = (dp1 * (-du2) + dp2 * du1) * invdet;
    hit->tri = best;
    hit->t = mt;
    hit->p = o + d * mt;
    hit->rayEps = 1e-3f * mt;
    hit->nn = normalize(cross(hit->dpdu, dpdv));   // core/diffgeom.cpp:46-54
    if (H.triFlip[best]) hit->nn = hit->nn * -1.f;
    return true;
}

// PhotonVolumeIntegrator::Transmittance with sample == NULL (photonvolume.cpp:15-30): one draw, Exp(-tau)
__device__ void lane_transmittance(const DevScene &S, V3 o, V3 d, float mint, float maxt, LaneRng &rng, Spec *out) {
    float step = 4.f * S.stepSize;
    float offset = lane_float(rng);
    if (S.volKind == PVOL_VOLUME_NONE) {
#pragma unroll
        for (int i = 0; i < NBIN; ++i) out->c[i] = 1.f;
        return;
    }
    RayD r;
    r.o = o; r.d = d; r.mint = mint; r.maxt = maxt;
    if (S.volKind != PVOL_VOLUME_GRID) {
        float t0, t1, lenAB = 0.f;
        bool hit = vol_intersect(S, r, &t0, &t1);
        if (hit) { V3 a = o + d * t0, b = o + d * t1; lenAB = len(a - b); }
#pragma unroll
        for (int i = 0; i < NBIN; ++i) {
            float tau = hit ? lenAB * (S.sigA[i] + S.sigS[i]) : 0.f;
            out->c[i] = expf(-tau);
        }
        return;
    }
    // DensityRegion::tau, core/volume.cpp:296-310
    float t0, t1;
    float length = len(d);
    float dsum = 0.f;
    bool any = false;
    if (length != 0.f) {
        RayD rn;
        rn.o = o; rn.d = vdiv(d, length); rn.mint = mint * length; rn.maxt = maxt * length;
        if (vol_intersect(S, rn, &t0, &t1)) {
            any = true;
            t0 += offset * step;
            // tau_b = (sum_j D_j * sigT_b) * step: accumulate per bin like the reference does
            Spec tau;
#pragma unroll
            for (int i = 0; i < NBIN; ++i) tau.c[i] = 0.f;
            while (t0 < t1) {
                float D = grid_density(S, xform_point(S.w2v, rn.o + rn.d * t0));
#pragma unroll
                for (int i = 0; i < NBIN; ++i) tau.c[i] += (S.sigA[i] + S.sigS[i]) * D;
                t0 += step;
            }
#pragma unroll
            for (int i = 0; i < NBIN; ++i) out->c[i] = expf(-(tau.c[i] * step));
        }
    }
    (void)dsum;
    if (!any) {
#pragma unroll
        for (int i = 0; i < NBIN; ++i) out->c[i] = 1.f;
    }
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

// BSDF::Sample_f with BSDF_ALL (core/reflection.cpp:534-598).  The sampled f is returned in factored form so
// the caller can rebuild it per bin in the reference's operation order:
//   fWhich 0: f_b = Kd_b * fFac (Lambertian, fFac = INV_PI)
//          1: f_b = (fFac * Kr_b) / fDiv (specular reflection, fFac = F, fDiv = |cos wi|)
//          2: f_b = (fFac * Kt_b) / fDiv (specular transmission, fFac = 1 - F)
//         -1: black
__device__ void bsdf_sample(const DevMaterial &m, V3 dpdu, V3 nn, V3 woW, float u0, float u1, float ucomp, const Spec &alpha,
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
        // BxDF::Sample_f (reflection.cpp:323-330): CosineSampleHemisphere via ConcentricSampleDisk (montecarlo.cpp:306-348)
        float r, theta;
        float sx = 2 * u0 - 1, sy = 2 * u1 - 1;
        float dx, dy;
        if (sx == 0.f && sy == 0.f) { dx = 0.f; dy = 0.f; }
        else {
            if (sx >= -sy) {
                if (sx > sy) { r = sx; if (sy > 0.f) theta = sy / r; else theta = 8.0f + sy / r; }
                else { r = sy; theta = 2.0f - sx / r; }
            } else {
                if (sx <= sy) { r = -sx; theta = 4.0f - sy / r; }
                else { r = -sy; theta = 6.0f + sx / r; }
            }
            theta *= K_PI / 4.f;
            dx = r * cosf(theta);
            dy = r * sinf(theta);
        }
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
        int lambda = sp_lambda(alpha);
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
    float r, theta, dx, dy;   // ConcentricSampleDisk
    float sx = 2 * u0 - 1, sy = 2 * u1 - 1;
    if (sx == 0.f && sy == 0.f) { dx = 0.f; dy = 0.f; }
    else {
        if (sx >= -sy) {
            if (sx > sy) { r = sx; if (sy > 0.f) theta = sy / r; else theta = 8.0f + sy / r; }
            else { r = sy; theta = 2.0f - sx / r; }
        } else {
            if (sx <= sy) { r = -sx; theta = 4.0f - sy / r; }
            else { r = -sy; theta = 6.0f + sx / r; }
        }
        theta *= K_PI / 4.f;
        dx = r * cosf(theta);
        dy = r * sinf(theta);
    }
    V3 wc = v3(H.worldCenter[0], H.worldCenter[1], H.worldCenter[2]);
    V3 Pdisk = wc + (v1 * dx + v2 * dy) * H.worldRadius;
    *o = Pdisk + ld * H.worldRadius;
    *d = -ld;
    *pdf = 1.f / (K_PI * H.worldRadius * H.worldRadius);
    return 1.f;
}

// montecarlo.h:206-218: digit loop in double, `n *= invBase` truncating through double
__device__ __forceinline__ float halton_dim(uint32_t n, uint32_t base, const uint32_t *perm, uint32_t stride) {
    double val = 0;
    double invBase = 1. / base, invBi = invBase;
    while (n > 0) {
        uint32_t d_i = perm[(size_t)(n % base) * stride];
        val += d_i * invBi;
        n = (uint32_t)(n * invBase);
        invBi *= invBase;
    }
    return fminf((float)val, 0x1.fffffep-1f);
}

// One followPhoton activation waiting to be resumed (its callee is running)
struct Frame {
    V3 rayO, rayD;       // photonRay as last assigned in this activation
    float rayMint, rayMaxt;   // maxt: the hit distance Scene::Intersect left in the ray, or INFINITY after a reassignment
    Hit hit;             // this activation's photonIsect
    Spec alpha;          // SURFACE: alpha entering the surface code; CHILDREN: alpha after Transmittance (split source)
    float tag;           // Spectrum::lambda of `alpha`
    V3 wo;
    int nInt;
    int state;           // 1 = resume at the surface code, 2 = resume the children loop
    int nextChild;       // CHILDREN: next bin (split) or next child ordinal (no split)
    bool split, spec;
};

struct PathCtx {
    const DevScene *S;
    const DevShootScene *H;
    LaneRng rng;
    bool causticDone, indirectDone, volumeDone;
    float *outPhotons;   // this task's local block buffer
    uint32_t cap;
    uint32_t nVol, nCaustic, nDirect, nIndirect;
    unsigned long long follow, noHit, march, inter, absorbed, splitc, overflow;
};

__device__ void follow_photon(PathCtx &C, V3 rayO, V3 rayD, float rayMint, Spec alpha, float tag) {
    const DevScene &S = *C.S;
    const DevShootScene &H = *C.H;
    Frame stack[SH_MAX_DEPTH];
    int sp = 0;
    // "call arguments" of the activation being entered
    int nInt = 0;
    bool spec = true;
    Hit hit;
    hit.tri = 0; hit.t = 0.f; hit.rayEps = 0.f; hit.p = hit.dpdu = hit.nn = v3(0.f, 0.f, 0.f);
    int mode = 0;   // 0 = CALL, 1 = SURFACE, 2 = CHILDREN, 3 = RETURN
    Frame F;        // the activation in SURFACE / CHILDREN mode
    for (;;) {
        if (mode == 0) {
            // ---- followPhoton entry (photonshooter.cpp:54-128)
            ++C.follow;
            float maxt = INFINITY;
            Hit h;
            if (!scene_closest(S, H, rayO, rayD, rayMint, &maxt, &h)) { ++C.noHit; mode = 3; continue; }
            ++nInt;
            float length = len(rayD);
            if (length == 0.f) { mode = 3; continue; }
            RayD rn;
            rn.o = rayO; rn.d = vdiv(rayD, length); rn.mint = rayMint * length; rn.maxt = maxt * length;
            float t0, t1;
            if (S.volKind == PVOL_VOLUME_NONE || !vol_intersect(S, rn, &t0, &t1)) { t0 = 1.0f; t1 = 0.0f; }
            t0 += lane_float(C.rng) * H.shooterStep;
            float t_i = t0;
            float xi = lane_float(C.rng);
            bool interaction = false;
            while (t0 < t1) {
                ++C.march;
                Spec tr;
                lane_transmittance(S, rayO, rn.d, t_i, t0, C.rng, &tr);
                if (xi > sp_y(S, tr)) { interaction = true; break; }
                t0 += H.shooterStep;
            }
            bool toSurface = true;
            if (interaction) {
                ++C.inter;
                V3 ip = rn.o + rn.d * t0;
                float dens = vol_density(S, ip);
                Spec ss, sa;
#pragma unroll
                for (int i = 0; i < NBIN; ++i) { ss.c[i] = S.sigS[i] * dens; sa.c[i] = S.sigA[i] * dens; }
                float ys = sp_y(S, ss), ya = sp_y(S, sa);
                bool scatter = (lane_float(C.rng) > ys / (ya + ys));
                if (!scatter) { ++C.absorbed; mode = 3; continue; }
                if (!C.volumeDone) {
                    if (nInt > 1) {
                        if (C.nVol < C.cap) {
                            float *o = C.outPhotons + (size_t)C.nVol * 36;
                            o[0] = ip.x; o[1] = ip.y; o[2] = ip.z; o[3] = rn.d.x; o[4] = rn.d.y; o[5] = rn.d.z;
#pragma unroll
                            for (int i = 0; i < NBIN; ++i) o[6 + i] = alpha.c[i];
                        } else {
                            ++C.overflow;
                        }
                        ++C.nVol;
                    }
                    float u1 = lane_float(C.rng), u2 = lane_float(C.rng);
                    float z = 1.f - 2.f * u1;   // UniformSampleSphere
                    float r = sqrtf(fmaxf(0.f, 1.f - z * z));
                    float phi = 2.f * K_PI * u2;
                    V3 dir = v3(r * cosf(phi), r * sinf(phi), z);
                    float pdf = 1.f / (4.f * K_PI);
                    float ref = vol_phase(S, ip, rn.d, dir);
                    if (ref == 0.f) { mode = 3; continue; }
#pragma unroll
                    for (int i = 0; i < NBIN; ++i) { alpha.c[i] *= ref; alpha.c[i] /= pdf; }
                    // the outer activation resumes at its surface code with the REASSIGNED ray (photonshooter.cpp:123-133)
                    if (sp >= SH_MAX_DEPTH) { ++C.overflow; mode = 3; continue; }
                    Frame &R = stack[sp++];
                    R.rayO = ip; R.rayD = dir; R.rayMint = 0.f; R.rayMaxt = INFINITY; R.hit = h; R.alpha = alpha; R.tag = tag; R.nInt = nInt; R.spec = spec;
                    R.state = 1; R.nextChild = 0; R.split = false; R.wo = v3(0.f, 0.f, 0.f);
                    rayO = ip; rayD = dir; rayMint = 0.f;   // callee arguments
                    toSurface = false;
                    mode = 0;
                }
            }
            if (toSurface) {
                F.rayO = rayO; F.rayD = rayD; F.rayMint = rayMint; F.rayMaxt = maxt; F.hit = h; F.alpha = alpha; F.tag = tag; F.nInt = nInt; F.spec = spec;
                mode = 1;
            }
            continue;
        }
        if (mode == 1) {
            // ---- surface code (photonshooter.cpp:131-197), F holds the activation
            Spec tr;
            lane_transmittance(S, F.rayO, F.rayD, F.rayMint, F.rayMaxt, C.rng, &tr);
#pragma unroll
            for (int i = 0; i < NBIN; ++i) F.alpha.c[i] *= tr.c[i];
            const DevMaterial &m = H.mats[H.triMat[F.hit.tri]];
            bool hasNonSpecular = m.nBxdf > num_components(m, BSDF_REFLECTION | BSDF_TRANSMISSION | BSDF_SPECULAR);
            bool hasTransmission = num_components(m, BSDF_TRANSMISSION | BSDF_DIFFUSE | BSDF_GLOSSY | BSDF_SPECULAR) > 0;
            bool dispersive = (m.kind == PVOL_MATERIAL_GLASS && m.vn > 0.f);
            F.split = hasTransmission && F.tag < 0.f && dispersive;
            if (F.split) {
                int nc = 0;
#pragma unroll
                for (int i = 0; i < NBIN; ++i) nc += (F.alpha.c[i] != 0.f) ? 1 : 0;
                C.splitc += nc;
            }
            F.wo = -F.rayD;
            if (hasNonSpecular) {
                bool deposited = false;
                if (F.spec && F.nInt > 1) {
                    if (!C.causticDone) { deposited = true; ++C.nCaustic; }
                } else {
                    if (F.nInt == 1 && !C.indirectDone && H.finalGather) { deposited = true; ++C.nDirect; }
                    else if (F.nInt > 1 && !C.indirectDone) { deposited = true; ++C.nIndirect; }
                }
                if (deposited && H.finalGather && lane_float(C.rng) < .125f) lane_skip(C.rng, 288);   // 2 x BSDF::rho (reflection.cpp:647-658)
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
                Spec a;
                float atag;
                if (F.split) {
                    int b = F.nextChild;
                    while (b < NBIN && F.alpha.c[b] == 0.f) ++b;
                    if (b >= NBIN) break;
                    F.nextChild = b + 1;
#pragma unroll
                    for (int i = 0; i < NBIN; ++i) a.c[i] = (i == b) ? F.alpha.c[i] : 0.f;
                    atag = 400 + b * ((700 - 400) / (float)(NBIN - 1));   // core/spectrum.cpp:101,106
                } else {
                    if (F.nextChild != 0) break;
                    F.nextChild = 1;
                    a = F.alpha;
                    atag = F.tag;
                }
                float ud0 = lane_float(C.rng), ud1 = lane_float(C.rng), uc = lane_float(C.rng);
                const DevMaterial &m = H.mats[H.triMat[F.hit.tri]];
                V3 wi;
                float pdf, fFac, fDiv;
                int flags, fWhich;
                bsdf_sample(m, F.hit.dpdu, F.hit.nn, F.wo, ud0, ud1, uc, a, &wi, &pdf, &flags, &fWhich, &fFac, &fDiv);
                if (fWhich < 0 || pdf == 0.f) continue;
                const float *K = fWhich == 0 ? m.kd : (fWhich == 1 ? m.kr : m.kt);
                Spec anew;
                float absdot = fabsf(dot(wi, F.hit.nn));
                bool frBlack = true;
#pragma unroll
                for (int i = 0; i < NBIN; ++i) {
                    float fr = (fWhich == 0) ? K[i] * fFac : (fFac * K[i]) / fDiv;
                    frBlack = frBlack && fr == 0.f;
                    anew.c[i] = a.c[i] * fr * absdot / pdf;   // alpha * fr * AbsDot(wi, nn) / pdf (photonshooter.cpp:208-209)
                }
                if (frBlack) continue;
                float continueProb = fminf(1.f, sp_y(S, anew) / sp_y(S, a));
                if (lane_float(C.rng) > continueProb) continue;
                Spec a2;
#pragma unroll
                for (int i = 0; i < NBIN; ++i) a2.c[i] = anew.c[i] / continueProb;
                float tag2 = (float)sp_lambda(a2);
                F.spec = F.spec && ((flags & BSDF_SPECULAR) != 0);
                if (C.indirectDone && !F.spec) continue;
                if (sp >= SH_MAX_DEPTH) { ++C.overflow; continue; }
                F.state = 2;
                stack[sp++] = F;
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
        F = stack[--sp];
        mode = F.state;
    }
}

__global__ __launch_bounds__(64) void shoot_kernel(ShootArgs A) {
    const uint32_t task = blockIdx.x * blockDim.x + threadIdx.x;
    if (task >= A.nTasks) return;
    const DevScene &S = *A.scene;
    const DevShootScene &H = *A.shoot;
    LaneRng rng;
    rng.mt = A.mt + task;
    rng.stride = A.nTasks;
    if (A.init) {
        // RNG rng(31 * taskNum) (photonshooter.cpp:235), then PermutedHalton(6, rng) (montecarlo.cpp:380-397)
        uint32_t x = 31u * task;
        MTW(rng, 0) = x;
        for (int i = 1; i < MT_N; ++i) { x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i; MTW(rng, i) = x; }
        rng.mti = MT_N;
        const uint32_t bases[6] = {2, 3, 5, 7, 11, 13};
        uint32_t off = 0;
        for (int d = 0; d < 6; ++d) {
            uint32_t b = bases[d];
            for (uint32_t j = 0; j < b; ++j) A.halton[(size_t)(off + j) * A.nTasks + task] = j;
            for (uint32_t i = 0; i < b; ++i) {   // Shuffle(buf, b, 1, rng), montecarlo.h:174-181
                uint32_t other = i + (lane_uint(rng) % (b - i));
                uint32_t *pa = &A.halton[(size_t)(off + i) * A.nTasks + task], *pb = &A.halton[(size_t)(off + other) * A.nTasks + task];
                uint32_t t = *pa; *pa = *pb; *pb = t;
            }
            off += b;
        }
        A.mt[(size_t)MT_N * A.nTasks + task] = (uint32_t)rng.mti;
        A.totalPaths[task] = 0;
        uint32_t fl = 0;
        if (H.nCausticWanted == 0) fl |= 1u;
        if (H.nIndirectWanted == 0) fl |= 2u;
        if (H.nVolumeWanted == 0) fl |= 4u;
        A.flags[task] = fl;
        return;
    }
    uint32_t fl = A.flags[task];
    uint32_t *lc = A.localCounts + (size_t)task * 4;
    lc[0] = lc[1] = lc[2] = lc[3] = 0;
    if (fl & 8u) return;   // finished
    rng.mti = (int)A.mt[(size_t)MT_N * A.nTasks + task];
    PathCtx C;
    C.S = &S; C.H = &H; C.rng = rng;
    C.causticDone = fl & 1u; C.indirectDone = fl & 2u; C.volumeDone = fl & 4u;
    C.outPhotons = A.localPhotons + (size_t)task * A.cap * 36;
    C.cap = A.cap;
    C.nVol = C.nCaustic = C.nDirect = C.nIndirect = 0;
    C.follow = C.noHit = C.march = C.inter = C.absorbed = C.splitc = C.overflow = 0;
    uint32_t totalPaths = A.totalPaths[task];
    const uint32_t blockSize = 4096;
    unsigned long long paths = 0;
    for (uint32_t i = 0; i < blockSize; ++i) {
        ++totalPaths;
        ++paths;
        float u[6];
        const uint32_t bases[6] = {2, 3, 5, 7, 11, 13};
        uint32_t off = 0;
        for (int d = 0; d < 6; ++d) { u[d] = halton_dim(totalPaths, bases[d], A.halton + (size_t)off * A.nTasks + task, A.nTasks); off += bases[d]; }
        // Distribution1D::SampleDiscrete (montecarlo.h:99-107): upper_bound over cdf[0..n]
        int n = S.nLights;
        int ub = 0;
        while (ub < n + 1 && !(u[0] < H.lightCdf[ub])) ++ub;
        int ln = max(0, ub - 1);
        if (ln >= n) ln = n - 1;
        float lightPdf = H.lightFunc[ln] / (H.lightFuncInt * n);
        V3 o, d;
        float pdf;
        float scale = light_emit(S, H, ln, u[1], u[2], &o, &d, &pdf);
        const DevLight &L = S.lights[ln];
        Spec Le;
        bool black = true;
#pragma unroll
        for (int b = 0; b < NBIN; ++b) { Le.c[b] = (L.kind == PVOL_LIGHT_SPOT) ? L.intensity[b] * scale : L.intensity[b]; black = black && Le.c[b] == 0.f; }
        if (pdf == 0.f || black) continue;
        // alpha = (AbsDot(Nl, d) * Le) / (pdf * lightPdf), Nl == d (photonshooter.cpp:264)
        float ad = fabsf(dot(d, d));
        float den = pdf * lightPdf;
        Spec alpha;
        bool ablack = true;
#pragma unroll
        for (int b = 0; b < NBIN; ++b) { alpha.c[b] = (Le.c[b] * ad) / den; ablack = ablack && alpha.c[b] == 0.f; }
        if (ablack) continue;
        follow_photon(C, o, d, 0.f, alpha, (float)sp_lambda(alpha));
    }
    A.totalPaths[task] = totalPaths;
    A.mt[(size_t)MT_N * A.nTasks + task] = (uint32_t)C.rng.mti;
    lc[0] = C.nVol; lc[1] = C.nCaustic; lc[2] = C.nDirect; lc[3] = C.nIndirect;
    atomicAdd(&A.stats[0], paths);
    atomicAdd(&A.stats[1], C.follow);
    atomicAdd(&A.stats[2], C.noHit);
    atomicAdd(&A.stats[3], C.march);
    atomicAdd(&A.stats[4], C.inter);
    atomicAdd(&A.stats[5], C.absorbed);
    atomicAdd(&A.stats[6], C.splitc);
    atomicAdd(&A.stats[7], C.overflow);
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
    uint32_t seg = blockIdx.y;
    if (seg >= M.nSeg) return;
    uint32_t cnt = M.count[seg];
    const float *src = M.localPhotons + (size_t)M.srcTask[seg] * M.cap * 36;
    float ns = M.nshot[seg];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt * 36; i += gridDim.x * blockDim.x) {
        uint32_t ph = i / 36, f = i - ph * 36;
        float v = src[(size_t)ph * 36 + f];
        size_t dst = (size_t)M.dstOff[seg] + ph;
        if (f < 3) M.p[dst * 3 + f] = v;
        else if (f < 6) M.wi[dst * 3 + (f - 3)] = v;
        else M.alpha[dst * 30 + (f - 6)] = v / ns;
    }
}

extern "C" hipError_t pvol_launch_shoot(const ShootArgs *a, hipStream_t stream) {
    hipLaunchKernelGGL(shoot_kernel, dim3((a->nTasks + 63) / 64), dim3(64), 0, stream, *a);
    return hipGetLastError();
}
extern "C" hipError_t pvol_launch_merge(const MergeArgs *m, hipStream_t stream) {
    if (!m->nSeg) return hipSuccess;
    hipLaunchKernelGGL(merge_kernel, dim3(32, m->nSeg), dim3(256), 0, stream, *m);
    return hipGetLastError();
}
