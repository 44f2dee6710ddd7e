// pvol_spec_dev.h -- included by pvol_march.hip after pvol_surface_dev.h, before pvol_tile_dev.h.
//
// The specular recursion of the surface integrator: PhotonIntegrator::Li ends with SpecularReflect +
// SpecularTransmit (integrators/photonmap.cpp:310-317 -> core/integrator.cpp:177-262).  Each draws BSDFSample(rng) (3 numbers),
// asks the BSDF for its one specular lobe (BSDF::Sample_f restricted to BSDF_REFLECTION|BSDF_SPECULAR resp. BSDF_TRANSMISSION|
// BSDF_SPECULAR: no randomness in the direction), and if the lobe exists spawns a ray (mint = rayEpsilon, depth + 1) through
// Renderer::Li -- closest hit, surface integrator (recursively), then the VOLUME integrator on the same Sample -- all from the
// tile's one RNG stream, BEFORE the parent ray's own volume term (renderers/samplerrenderer.cpp:228-251).
//
// So a camera sample that meets glass owns a small tree of spawned rays ("segments").  Its shape is a function of the camera
// ray alone; what the stream sees is, in post-order, for every segment: the draws of the surface integrator on the way (3 per
// lobe tried, the matte leaf's light samples / rho() draws) and then that segment's volume Li().  This header walks the tree
// (compile-time bounded recursion, maxspeculardepth <= 5) and hands every segment to a policy in exactly that order; the tile
// pre-pass (pvol_tile_dev.h) uses it to count draws, to lay the segments out as extra rays and -- where drawn values matter
// (several lights) -- to walk their streams for real.  spec_compose_kernel then folds the segments' radiance back:
//     value(seg) = Lv(seg) + T(seg) (.) [ Ls_matte(seg's hit) + sum over the hit's lobes  f (.) value(child) |cos| / pdf ]
//     sample    += T(primary) (.) sum over the primary hit's lobes f (.) value(child) |cos| / pdf
// with f = (F Kr) / |cos theta_i| resp. ((1 - F) Kt) / |cos theta_t| (SpecularReflection / SpecularTransmission::Sample_f,
// core/reflection.cpp:138-182; the camera side refracts at the undispersed index: no `alpha` is passed).
#ifndef PVOL_SPEC_DEV_H
#define PVOL_SPEC_DEV_H


__device__ __forceinline__ float spec_fresnel(float cosi, float eta_i, float eta_t) {   // FresnelDielectric::Evaluate, reflection.cpp:60-67,115-135
    cosi = cosi < -1.f ? -1.f : (cosi > 1.f ? 1.f : cosi);
    const bool entering = cosi > 0.f;
    float ei = eta_i, et = eta_t;
    if (!entering) { const float t = ei; ei = et; et = t; }
    const float sint = ei / et * sqrtf(fmaxf(0.f, 1.f - cosi * cosi));
    if (sint >= 1.f) return 1.f;
    const float cost = sqrtf(fmaxf(0.f, 1.f - sint * sint));
    const float ac = fabsf(cosi);
    const float Rparl = ((et * ac) - (ei * cost)) / ((et * ac) + (ei * cost));
    const float Rperp = ((ei * ac) - (et * cost)) / ((ei * ac) + (et * cost));
    return (Rparl * Rparl + Rperp * Rperp) / 2.f;
}

struct SpecLobe { bool valid; V3 wi; float Fs, awz, g; };
// BSDF::Sample_f(wo, &wi, BSDFSample, &pdf, lobe | BSDF_SPECULAR) on a glass hit and the test SpecularReflect / SpecularTransmit
// make of the result (core/integrator.cpp:186,227: pdf > 0 && !f.IsBlack() && AbsDot(wi, n) != 0).  lobe: 1 reflection, 2 transmission.
__device__ SpecLobe spec_lobe(const DevMaterial &m, const SurfHit &h, V3 woW, int lobe) {
    SpecLobe r;
    r.valid = false; r.wi = v3(0.f, 0.f, 0.f); r.Fs = 0.f; r.awz = 1.f; r.g = 0.f;
    const int want = (lobe == 1 ? BSDF_REFLECTION : BSDF_TRANSMISSION) | BSDF_SPECULAR;
    bool have = false;
    for (int i = 0; i < m.nBxdf; ++i) have = have || m.bxdfType[i] == want;   // GlassMaterial adds a lobe only for a non-black Kr / Kt (glass.cpp:52-57)
    if (!have) return r;
    const V3 sn = normalize(h.dpdu);
    const V3 tn = cross(h.nn, sn);
    const V3 wo = v3(dot(woW, sn), dot(woW, tn), dot(woW, h.nn));
    V3 wi;
    if (lobe == 1) {
        wi = v3(-wo.x, -wo.y, wo.z);
        r.Fs = spec_fresnel(wo.z, 1.f, m.ior);
    } else {
        const bool entering = wo.z > 0.f;
        float ei = 1.f, et = m.ior;
        if (!entering) { const float t = ei; ei = et; et = t; }
        const float sini2 = fmaxf(0.f, 1.f - wo.z * wo.z);
        const float eta = ei / et;
        const float sint2 = eta * eta * sini2;
        if (sint2 >= 1.f) return r;   // total internal reflection
        float cost = sqrtf(fmaxf(0.f, 1.f - sint2));
        if (entering) cost = -cost;
        wi = v3(eta * -wo.x, eta * -wo.y, cost);
        r.Fs = 1.f - spec_fresnel(wo.z, 1.f, m.ior);
    }
    r.awz = fabsf(wi.z);
    r.wi = v3(sn.x * wi.x + tn.x * wi.y + h.nn.x * wi.z, sn.y * wi.x + tn.y * wi.y + h.nn.y * wi.z, sn.z * wi.x + tn.z * wi.y + h.nn.z * wi.z);
    const float ad = fabsf(dot(r.wi, h.nn));
    r.g = ad / 1.f;
    r.valid = r.Fs != 0.f && ad != 0.f;
    return r;
}

struct SpecCtx {
    unsigned blackMask;   // lights whose intensity is black (no draw for their samples)
    uint32_t pending;     // RandomUInt calls since the last visited ray's volume Li()
    uint32_t nSeg;        // segments visited so far
};

// Renderer::Li for a spawned ray at depth D >= 1, then the policy's turn (the ray's own volume Li() comes last).
template <int D, class P> __device__ void spec_ray(const DevScene &S, SpecCtx &X, P &pol, V3 o, V3 d, float mint, int lobe, int mat, float Fs, float awz, float g);

// PhotonIntegrator::Li at a hit of a ray of depth D: what it draws, and the rays it spawns
template <int D, class P> __device__ void spec_surface(const DevScene &S, SpecCtx &X, P &pol, V3 d, const SurfHit &h) {
    const DevMaterial &m = S.shootScene->mats[h.mat];
    if (m.kind == PVOL_MATERIAL_MATTE) {
        X.pending += surf_count_draws(S, h, d, X.blackMask, D);
        return;
    }
    // glass: UniformSampleAllLights finds f == 0 for every light (no draw), LPhoton returns before its rho() draws
    if constexpr (D + 1 < SPEC_MAX_DEPTH) {
        if (D + 1 < S.surf.maxSpecularDepth) {
#pragma unroll 1
            for (int lobe = 1; lobe <= 2; ++lobe) {
                X.pending += 3u;   // BSDFSample(rng): uDir[0], uDir[1], uComponent
                const SpecLobe L = spec_lobe(m, h, -d, lobe);
                if (L.valid) spec_ray<D + 1>(S, X, pol, h.p, L.wi, h.rayEps, lobe, h.mat, L.Fs, L.awz, L.g);
            }
        }
    }
}

template <int D, class P> __device__ void spec_ray(const DevScene &S, SpecCtx &X, P &pol, V3 o, V3 d, float mint, int lobe, int mat, float Fs, float awz, float g) {
    SurfHit h;
    h.tri = 0; h.mat = 0; h.t = 0.f; h.rayEps = 0.f; h.p = h.nn = h.dpdu = v3(0.f, 0.f, 0.f);
    const bool hit = surf_closest(S, o, d, mint, &h);
    if (hit) spec_surface<D>(S, X, pol, d, h);
    pol.visit(X, D, lobe, mat, Fs, awz, g, o, d, mint, hit ? h.t : INFINITY);
    ++X.nSeg;
}

// ---- composition: one camera sample per lane; samples without segments leave at once
__global__ __launch_bounds__(256) void spec_compose_kernel(SpecComposeArgs A) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= A.nRays) return;
    const size_t ri = (size_t)A.first + i;
    const uint32_t link = A.link[ri];
    if (!link || (link & SPEC_LINK_DONE)) return;
    A.link[ri] = link | SPEC_LINK_DONE;
    const DevScene &S = *A.scene;
    const uint32_t base = link >> SPEC_LINK_COUNT_BITS, n = link & ((1u << SPEC_LINK_COUNT_BITS) - 1u);
    const float kT = -1.442695041f * A.tau[ri];
    float x = 0.f, y = 0.f, z = 0.f, sx = 0.f, sy = 0.f, sz = 0.f;
    for (int b = 0; b < 30; ++b) {
        float acc[SPEC_MAX_DEPTH + 1];   // acc[D]: what the children of the pending depth-(D-1) ray have brought in so far
#pragma unroll
        for (int dd = 0; dd <= SPEC_MAX_DEPTH; ++dd) acc[dd] = 0.f;
        for (uint32_t s = 0; s < n; ++s) {   // post-order: a ray's children come before it
            const SegInfo si = A.info[base + s];
            const int D = (int)(si.depthLobeMat & 0xffu), lobe = (int)((si.depthLobeMat >> 8) & 0xffu), mat = (int)(si.depthLobeMat >> 16);
            const float *o = A.segOut + (size_t)(base + s) * 60;
            float below = 0.f;   // the children's sum, taken and cleared
#pragma unroll
            for (int dd = 1; dd <= SPEC_MAX_DEPTH; ++dd) if (dd == D + 1) { below = acc[dd]; acc[dd] = 0.f; }
            const float val = o[b] + o[30 + b] * below;
            const DevMaterial &m = S.shootScene->mats[mat];
            const float K = lobe == 1 ? m.kr[b] : m.kt[b];
            const float add = ((si.Fs * K) / si.awz) * val * si.g;   // f * Li * (AbsDot(wi, n) / pdf), integrator.cpp:196,240
#pragma unroll
            for (int dd = 1; dd <= SPEC_MAX_DEPTH; ++dd) if (dd == D) acc[dd] += add;
        }
        const float Ls = acc[1];
        const float Tb = __builtin_amdgcn_exp2f((S.sigA[b] + S.sigS[b]) * kT);
        const float v = Tb * Ls;
        x += S.cieX[b] * v; y += S.cieY[b] * v; z += S.cieZ[b] * v;
        sx += S.cieX[b] * Ls; sy += S.cieY[b] * Ls; sz += S.cieZ[b] * Ls;
    }
    const float scale = float(700 - 400) / float(106.856895f * 30);
    float *op = A.out + ri * 4;
    op[0] += x * scale; op[1] += y * scale; op[2] += z * scale;
    if (A.surfOut) { A.surfOut[3 * ri] = sx * scale; A.surfOut[3 * ri + 1] = sy * scale; A.surfOut[3 * ri + 2] = sz * scale; }
}

extern "C" hipError_t pvol_launch_spec_compose(const SpecComposeArgs *a, hipStream_t stream) {
    if (!a->nRays) return hipSuccess;
    hipLaunchKernelGGL(spec_compose_kernel, dim3((a->nRays + 255) / 256), dim3(256), 0, stream, *a);
    return hipGetLastError();
}

// every slot of the segment pool starts as a ray that meets nothing (BBox::IntersectP fails at once: mint > maxt), so kernels that
// run over the pool's capacity find nothing to do beyond the segments actually laid out
__global__ void spec_fill_kernel(pvol_ray *rays, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    pvol_ray r;
    r.o[0] = 0.f; r.o[1] = 0.f; r.o[2] = 0.f; r.mint = 1.f;
    r.d[0] = 0.f; r.d[1] = 0.f; r.d[2] = 1.f; r.maxt = 0.f;
    r.time = 0.f; r.scatter_u = 0.f; r.rng_skip = 0u; r.flags = 0u;
    rays[i] = r;
}
extern "C" hipError_t pvol_launch_spec_fill(pvol_ray *rays, uint32_t n, hipStream_t stream) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(spec_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, rays, n);
    return hipGetLastError();
}
#endif
