// orc_surface.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// PhotonIntegrator::Li (integrators/photonmap.cpp:154-319) for the part of SURVEY 8(f)-2 the product implements: camera rays
// that end on NON-SPECULAR surfaces (the matte walls of both BASELINE scenes), no indirect map (both scenes set
// `indirectphotons 0`, which leaves the final gather and the indirect estimate without a map, photonmap.cpp:183,315).
// What remains is
//   L = UniformSampleAllLights (core/integrator.cpp:47-79 -> EstimateDirect :117-174, delta lights)
//     + LPhoton(causticMap)     (photonmap.cpp:62-108, diffuse branch)
// and the RNG traffic of the rest: two BSDF::rho(wo, rng) inside LPhoton (2 x 72 draws, core/reflection.cpp:660-670), two
// BSDFSample(rng) of SpecularReflect / SpecularTransmit (2 x 3 draws, core/integrator.cpp:184,225), one draw per unoccluded
// light sample (visibility.Transmittance(..., NULL, rng), core/integrator.cpp:133).  Pinned by the reference's own records
// (tests/golden/render_vh_surf*.bin, written by oracle/ref_capture.cpp `render ... surface`).
#ifndef ORC_SURFACE_H
#define ORC_SURFACE_H
#include "orc_integrator.h"
#include "orc_shooter.h"   // material_bxdfs, bsdf_sample_f: the BSDFs the photon side already restates

namespace orc {

struct SurfaceIntegrator {
    int nLookup;               // "nused"
    float maxDistSquared;      // "maxdist"^2
    int maxSpecularDepth;      // "maxspeculardepth" (5)
    bool finalGather;
    const KdTree *causticMap;  // may be NULL
    int nCausticPaths;
};

// BSDF::f(wo, wi, flags) of a matte surface (core/reflection.cpp:627-644 + Lambertian::f :212-214): the Lambertian lobe
// counts only when wi and wo lie on the same side of the geometric normal
inline Spec matte_f(const Material &m, V3 ng, V3 woW, V3 wiW) {
    if (!(dot(wiW, ng) * dot(woW, ng) > 0.f)) return spec_const(0.f);   // BSDF_REFLECTION is masked out otherwise
    return m.kd * kInvPi;
}

// photonmap.cpp:56-60
inline float photon_kernel(V3 photonP, V3 p, float maxDist2) {
    float s = (1.f - length_sq(photonP - p) / maxDist2);
    return 3.f * kInvPi * s * s;
}

struct SpecCtx {   // what the specular recursion carries besides the ray: the camera sample's scatter offset (every nested
    float scatterU;             // volume Li() reads the SAME Sample, samplerrenderer.cpp:247) and the volume integrator's scratch
    std::vector<float> *scratch;
};
inline Spec surface_li(const Integrator &I, const SurfaceIntegrator &S, const Ray &ray, const Hit &isect, Rng &rng, Counters *ctr,
                       std::vector<ClosePhoton> &lookupBuf, bool *supported, int depth = 0, const SpecCtx *sx = 0);

// SamplerRenderer::Li for a spawned ray (renderers/samplerrenderer.cpp:228-251): closest hit, surface integrator, then the
// volume integrator on the (clipped) ray; delta lights add no Le for rays that leave the scene.
inline Spec renderer_li(const Integrator &I, const SurfaceIntegrator &S, Ray ray, int depth, Rng &rng, Counters *ctr,
                        std::vector<ClosePhoton> &lookupBuf, bool *supported, const SpecCtx &sx) {
    Hit hit;
    Spec L = spec_const(0.f);
    if (scene_intersect(*I.scene, &ray, &hit)) L = surface_li(I, S, ray, hit, rng, ctr, lookupBuf, supported, depth, &sx);
    Spec T;
    Spec Lvi = li(I, ray, sx.scatterU, rng, &T, ctr, *sx.scratch, lookupBuf);
    return T * L + Lvi;
}

// SpecularReflect / SpecularTransmit (core/integrator.cpp:177-262) without the ray differentials (nothing on this path reads
// them): BSDFSample(rng) draws three numbers, Sample_f is restricted to the one specular lobe, the spawned ray starts at
// rayEpsilon and carries depth + 1.
inline Spec specular_bounce(const Integrator &I, const SurfaceIntegrator &S, const Ray &ray, const Hit &isect, int lobe, int depth, Rng &rng,
                            Counters *ctr, std::vector<ClosePhoton> &lookupBuf, bool *supported, const SpecCtx &sx) {
    const float u0 = rng.random_float(), u1 = rng.random_float(), uc = rng.random_float();   // BSDFSample(RNG&), reflection.h:491-495
    const V3 wo = -ray.d;
    V3 wi = v3(0.f, 0.f, 0.f);
    float pdf = 0.f;
    int type = 0;
    Spec f = bsdf_sample_f(*I.scene, isect.tri, isect.dpdu, isect.nn, wo, &wi, u0, u1, uc, &pdf, &type, spec_const(1.f), lobe | BSDF_SPECULAR);
    Spec L = spec_const(0.f);
    if (pdf > 0.f && !is_black(f) && fabsf(dot(wi, isect.nn)) != 0.f) {
        Ray rd = make_ray(isect.p, wi, isect.rayEpsilon, kInfinity, ray.time);
        Spec Li = renderer_li(I, S, rd, depth + 1, rng, ctr, lookupBuf, supported, sx);
        L = f * Li * (fabsf(dot(wi, isect.nn)) / pdf);
    }
    return L;
}

// Returns Li of the surface integrator at a hit of `ray`; advances rng exactly as the reference does.  Specular surfaces recurse
// when `sx` is given (the tile driver passes it); without it `supported` is cleared for them as before.
inline Spec surface_li(const Integrator &I, const SurfaceIntegrator &S, const Ray &ray, const Hit &isect, Rng &rng, Counters *ctr,
                       std::vector<ClosePhoton> &lookupBuf, bool *supported, int depth, const SpecCtx *sx) {
    const Scene &sc = *I.scene;
    const Material &mat = sc.mats[prim_material(sc, isect.tri)];
    Spec L = spec_const(0.f);
    if (mat.kind != PVOL_MATERIAL_MATTE) {
        if (!sx) { if (supported) *supported = false; return L; }
        // glass (materials/glass.cpp:42-59): purely specular.  UniformSampleAllLights finds f == 0 for every light sample (no
        // draw: the transmittance of the shadow ray is only asked for when f is not black, core/integrator.cpp:131-134);
        // LPhoton returns before its rho() draws (no non-specular component, photonmap.cpp:66-68).
        if (depth + 1 < S.maxSpecularDepth) {
            L = L + specular_bounce(I, S, ray, isect, BSDF_REFLECTION, depth, rng, ctr, lookupBuf, supported, *sx);
            L = L + specular_bounce(I, S, ray, isect, BSDF_TRANSMISSION, depth, rng, ctr, lookupBuf, supported, *sx);
        }
        return L;
    }
    const bool hasLambert = !is_black(mat.kd);   // MatteMaterial::GetBSDF adds the Lambertian only for a non-black Kd (matte.cpp:55-60)
    const V3 wo = -ray.d;
    const V3 p = isect.p, n = isect.nn;          // dgShading == dg for a triangle without normals
    // ---- UniformSampleAllLights: one sample per delta light
    for (size_t li = 0; li < sc.lights.size(); ++li) {
        V3 wi; float lightPdf; Ray vis;
        Spec Li = light_sample_L(sc.lights[li], p, isect.rayEpsilon, ray.time, &wi, &lightPdf, &vis);
        Spec Ld = spec_const(0.f);
        if (lightPdf > 0. && !is_black(Li)) {
            Spec f = matte_f(mat, n, wo, wi);
            if (!is_black(f) && !scene_intersect_p(sc, vis)) {
                Li = Li * transmittance(I, vis, rng, ctr);                   // sample == NULL: one draw
                Ld = Ld + f * Li * (fabsf(dot(wi, n)) / lightPdf);          // IsDeltaLight()
            }
        }
        L = L + Ld / 1.f;                                                   // nSamples == 1
    }
    // ---- caustic estimate: LPhoton, diffuse branch (the matte BSDF has no glossy / transmissive component)
    if (S.causticMap && hasLambert) {   // LPhoton: `map && bsdf->NumComponents(nonSpecular) > 0` (photonmap.cpp:68)
        if ((int)lookupBuf.size() < S.nLookup) lookupBuf.resize(S.nLookup);
        PhotonProcess proc;
        proc.photons = &lookupBuf[0]; proc.nLookup = (uint32_t)S.nLookup; proc.nFound = 0; proc.ctr = 0;
        float maxDist2 = S.maxDistSquared;
        kd_lookup(*S.causticMap, 0, p, proc, maxDist2);
        const V3 Nf = dot(n, wo) < 0.f ? -n : n;                            // Faceforward(nn, wo)
        Spec Lr = spec_const(0.f), Lt = spec_const(0.f);
        for (uint32_t i = 0; i < proc.nFound; ++i) {
            const Photon &ph = S.causticMap->data[proc.photons[i].photon];
            float k = photon_kernel(ph.p, p, maxDist2);
            if (dot(Nf, ph.wi) > 0.f) Lr = Lr + (k / (S.nCausticPaths * maxDist2)) * ph.alpha;
            else Lt = Lt + (k / (S.nCausticPaths * maxDist2)) * ph.alpha;
        }
        rng.skip(72);                                                       // bsdf->rho(wo, rng, BSDF_ALL_REFLECTION): 6x6 jittered samples, unused by Lambertian::rho
        const Spec rhoR = mat.kd;                                           // Lambertian::rho == R (reflection.h:222-223)
        rng.skip(72);                                                       // bsdf->rho(wo, rng, BSDF_ALL_TRANSMISSION): no such component -> 0
        L = L + Lr * rhoR * kInvPi + Lt * spec_const(0.f) * kInvPi;
    }
    // indirect: finalGather && indirectMap != NULL is false without an indirect map; LPhoton(NULL map) adds nothing, draws nothing
    // ---- SpecularReflect + SpecularTransmit: BSDFSample(rng) is constructed before Sample_f finds no specular component
    if (depth + 1 < S.maxSpecularDepth) rng.skip(6);
    return L;
}

}  // namespace orc
#endif
