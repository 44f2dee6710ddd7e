// oracle/ref_capture.cpp -- TEST INFRASTRUCTURE.  Capture tool linked against the REFERENCE's
// own objects (oracle/_ref/libpbrtref.a, built by oracle/Makefile from the sources where they
// lie).  It builds the BASELINE.json scenes through the reference's public Create*() functions
// exactly as core/api.cpp would from the .pbrt files, flattens them into the pvol_scene layout
// (the same flattening a drop-in shim performs), and records outputs of the
// reference's PhotonVolumeIntegrator::Li / Transmittance, lights, shapes, BSDFs, RNG and samplers
// as golden vectors for tests/golden/.
//
// Nothing here restates reference algorithms: every number written comes out of reference code.
// The only liberty: `private`/`protected` are opened for the reference headers so that object
// state can be read back (light positions, volume extents, PhotonShooter::volumeMap).  The
// reference's photon SHOOTER cannot run here: PhotonShootingTask needs core/parallel.cpp, which is
// unbuildable in this image (see Makefile); photon maps are therefore inputs to this tool.
#include "ref_scenes.h"
#include "../integration/hip_flatten.h"   // the reference-side binding's Scene -> pvol_scene walk, tested by `shimscene`


// ----------------------------------------------------------------------------- flattening
static void putSpec(std::vector<float> &v, const Spectrum &s) { for (int i = 0; i < nSpectralSamples; ++i) v.push_back(s.c[i]); }
static void putMat(std::vector<float> &v, const Matrix4x4 &m) { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) v.push_back(m.m[r][c]); }

static void flatten(const BuiltScene &B, Blob &out, bool withDensity) {
    out.puti1("vol.kind", B.volumeKind);
    std::vector<float> ext, w2v, v2w, sa, ss, le;
    float g = 0;
    const BBox *e = 0;
    const Transform *W2V = 0;
    if (B.volumeKind == PVOL_VOLUME_GRID) {
        const VolumeGridDensity *v = (const VolumeGridDensity *)B.volume;
        e = &v->extent; W2V = &v->WorldToVolume; putSpec(sa, v->sig_a); putSpec(ss, v->sig_s); putSpec(le, v->le); g = v->g;
    } else {
        const HomogeneousVolumeDensity *v = (const HomogeneousVolumeDensity *)B.volume;
        e = &v->extent; W2V = &v->WorldToVolume; putSpec(sa, v->sig_a); putSpec(ss, v->sig_s); putSpec(le, v->le); g = v->g;
    }
    float ex[6] = {e->pMin.x, e->pMin.y, e->pMin.z, e->pMax.x, e->pMax.y, e->pMax.z};
    out.putf("vol.extent", ex, 6);
    putMat(w2v, W2V->m); putMat(v2w, W2V->mInv);
    out.putf("vol.w2v", w2v); out.putf("vol.v2w", v2w);
    out.putf("vol.sigma_a", sa); out.putf("vol.sigma_s", ss); out.putf("vol.le", le);
    out.putf1("vol.g", g);
    int32_t dims[3] = {B.nx, B.ny, B.nz};
    if (B.volumeKind != PVOL_VOLUME_GRID) dims[0] = dims[1] = dims[2] = 0;
    out.put("vol.dims", blob::I32, dims, 3);
    if (B.volumeKind == PVOL_VOLUME_GRID && withDensity) out.putf("vol.density", ((const VolumeGridDensity *)B.volume)->density, (size_t)B.nx * B.ny * B.nz);

    std::vector<int32_t> lk;
    std::vector<float> lpos, ldir, l2w, w2l, lint, lcos;
    for (size_t i = 0; i < B.lights.size(); ++i) {
        lk.push_back(B.lightKinds[i]);
        const Light *L = B.lights[i];
        putMat(l2w, L->LightToWorld.m); putMat(w2l, L->WorldToLight.m);
        if (B.lightKinds[i] == PVOL_LIGHT_DISTANT) {
            const DistantLight *d = (const DistantLight *)L;
            lpos.insert(lpos.end(), 3, 0.f);
            ldir.push_back(d->lightDir.x); ldir.push_back(d->lightDir.y); ldir.push_back(d->lightDir.z);
            putSpec(lint, d->L); lcos.push_back(0.f); lcos.push_back(0.f);
        } else if (B.lightKinds[i] == PVOL_LIGHT_SPOT) {
            const SpotLight *s = (const SpotLight *)L;
            lpos.push_back(s->lightPos.x); lpos.push_back(s->lightPos.y); lpos.push_back(s->lightPos.z);
            ldir.insert(ldir.end(), 3, 0.f);
            putSpec(lint, s->Intensity); lcos.push_back(s->cosTotalWidth); lcos.push_back(s->cosFalloffStart);
        } else {
            const PointLight *p = (const PointLight *)L;
            lpos.push_back(p->lightPos.x); lpos.push_back(p->lightPos.y); lpos.push_back(p->lightPos.z);
            ldir.insert(ldir.end(), 3, 0.f);
            putSpec(lint, p->Intensity); lcos.push_back(0.f); lcos.push_back(0.f);
        }
    }
    out.put("lights.kind", blob::I32, lk.data(), lk.size());
    out.putf("lights.pos", lpos); out.putf("lights.dir", ldir); out.putf("lights.l2w", l2w); out.putf("lights.w2l", w2l);
    out.putf("lights.intensity", lint); out.putf("lights.cos", lcos);

    std::vector<float> tp;
    std::vector<int32_t> tm, tf;
    for (size_t k = 0; k < B.meshes.size(); ++k) {
        const TriangleMesh *mesh = B.meshes[k];
        for (int t = 0; t < mesh->ntris; ++t) {
            for (int c = 0; c < 3; ++c) {
                const Point &p = mesh->p[mesh->vertexIndex[3 * t + c]];
                tp.push_back(p.x); tp.push_back(p.y); tp.push_back(p.z);
            }
            tm.push_back(B.meshMaterial[k]);
            tf.push_back((mesh->ReverseOrientation ^ mesh->TransformSwapsHandedness) ? 1 : 0);
        }
    }
    out.putf("tris.p", tp);
    out.put("tris.material", blob::I32, tm.data(), tm.size());
    out.put("tris.flip", blob::I32, tf.data(), tf.size());

    if (!B.spheres.empty()) {   // Shape "sphere": what Sphere::Sphere stored (shapes/sphere.cpp:41-49)
        std::vector<float> so2w, sw2o, sf;
        std::vector<int32_t> sm, sfl;
        for (size_t k = 0; k < B.spheres.size(); ++k) {
            const Sphere *sp = B.spheres[k];
            putMat(so2w, sp->ObjectToWorld->m); putMat(sw2o, sp->WorldToObject->m);
            float f[6] = {sp->radius, sp->zmin, sp->zmax, sp->thetaMin, sp->thetaMax, sp->phiMax};
            sf.insert(sf.end(), f, f + 6);
            sm.push_back(B.sphereMaterial[k]);
            sfl.push_back((sp->ReverseOrientation ^ sp->TransformSwapsHandedness) ? 1 : 0);
        }
        out.putf("spheres.o2w", so2w); out.putf("spheres.w2o", sw2o); out.putf("spheres.f", sf);
        out.put("spheres.material", blob::I32, sm.data(), sm.size());
        out.put("spheres.flip", blob::I32, sfl.data(), sfl.size());
    }

    std::vector<int32_t> mk;
    std::vector<float> kd, kr, kt, ior, vn;
    for (size_t i = 0; i < B.mats.size(); ++i) {
        mk.push_back(B.mats[i].kind);
        putSpec(kd, B.mats[i].kd); putSpec(kr, B.mats[i].kr); putSpec(kt, B.mats[i].kt);
        ior.push_back(B.mats[i].ior); vn.push_back(B.mats[i].vn);
    }
    out.put("mats.kind", blob::I32, mk.data(), mk.size());
    out.putf("mats.kd", kd); out.putf("mats.kr", kr); out.putf("mats.kt", kt); out.putf("mats.ior", ior); out.putf("mats.vn", vn);

    const BBox &wb = B.scene->WorldBound();
    float w[6] = {wb.pMin.x, wb.pMin.y, wb.pMin.z, wb.pMax.x, wb.pMax.y, wb.pMax.z};
    out.putf("world", w, 6);
    std::vector<float> cx, cy, cz;
    putSpec(cx, SampledSpectrum::X); putSpec(cy, SampledSpectrum::Y); putSpec(cz, SampledSpectrum::Z);
    out.putf("cie.x", cx); out.putf("cie.y", cy); out.putf("cie.z", cz);
    out.putf1("xyz_scale", float(sampledLambdaEnd - sampledLambdaStart) / float(CIE_Y_integral * nSpectralSamples));

    float pf[3] = {B.stepSize, B.maxDist, B.shooterStep};
    int32_t pi[6] = {B.nUsed, B.nVolumePhotons, B.maxPhotonDepth, B.nCaustic, B.nIndirect, B.finalGather};
    out.putf("params.f", pf, 3);
    out.put("params.i", blob::I32, pi, 6);
    std::vector<float> c2w;
    putMat(c2w, B.camToWorld.m);
    out.putf("camera.c2w", c2w);
    out.putf1("camera.fov", B.fov);
    int32_t film[3] = {B.xres, B.yres, B.spp};
    out.put("film", blob::I32, film, 3);
}

// ----------------------------------------------------------------------------- capture helpers
// The caller side of the hot path: SamplerRenderer::Transmittance (renderers/samplerrenderer.cpp:253-258)
// forwards to the volume integrator.  SamplerRenderer itself cannot be linked (task system).
class CaptureRenderer : public Renderer {
public:
    explicit CaptureRenderer(VolumeIntegrator *v, SurfaceIntegrator *s = NULL) : vi(v), si(s) {}
    void Render(const Scene *) {}
    // SamplerRenderer::Li (renderers/samplerrenderer.cpp:228-251), reached only from the surface integrator's specular bounces
    Spectrum Li(const Scene *scene, const RayDifferential &ray, const Sample *sample, RNG &rng, MemoryArena &arena, Intersection *isect, Spectrum *T) const {
        Spectrum localT;
        if (!T) T = &localT;
        Intersection localIsect;
        if (!isect) isect = &localIsect;
        Spectrum L = 0.f;
        if (scene->Intersect(ray, isect)) { if (si) L = si->Li(scene, this, ray, *isect, sample, rng, arena); }
        else for (uint32_t i = 0; i < scene->lights.size(); ++i) L += scene->lights[i]->Le(ray);
        Spectrum Lvi = vi->Li(scene, this, ray, sample, rng, T, arena);
        return *T * L + Lvi;
    }
    Spectrum Transmittance(const Scene *scene, const RayDifferential &ray, const Sample *sample, RNG &rng, MemoryArena &arena) const {
        return vi->Transmittance(scene, this, ray, sample, rng, arena);
    }
    VolumeIntegrator *vi;
    SurfaceIntegrator *si;
};

static uint64_t drawsBetween(RNG &shadow, const RNG &live, uint64_t limit) {
    // advance `shadow` until it is in the same state as `live`
    uint64_t k = 0;
    for (;;) {
        if (shadow.mti == live.mti && shadow.mt[0] == live.mt[0] && shadow.mt[1] == live.mt[1] && shadow.mt[397] == live.mt[397] &&
            shadow.mt[623] == live.mt[623])
            return k;
        shadow.RandomUInt();
        if (++k > limit) { fprintf(stderr, "drawsBetween: limit exceeded\n"); exit(2); }
    }
}

static Spectrum specFrom(const float *c) { Spectrum s(0.f); for (int i = 0; i < nSpectralSamples; ++i) s.c[i] = c[i]; s.lambda = s.extractLambda(); return s; }

static int cmdScene(const std::string &name, const char *outPath) {
    BuiltScene B;
    memset(&B.nx, 0, sizeof(int) * 3);
    if (!buildByName(B, name)) { fprintf(stderr, "unknown scene %s\n", name.c_str()); return 1; }
    Blob out;
    flatten(B, out, name != "volumescene_grid128");  // the 128^3 blob is regenerated by the repo's own generator
    return out.save(outPath) ? 0 : 1;
}

// What the reference-side binding (integration/hip_flatten.h) makes of the LIVE scene object -- aggregate, lights, volume as
// the renderer holds them -- written with the keys of flatten() above, which reads the builders' own lists instead.  The two
// must agree array for array (tests/test_shim_flatten.py).
static int cmdShimScene(const std::string &name, const char *outPath) {
    BuiltScene B;
    memset(&B.nx, 0, sizeof(int) * 3);
    if (!buildByName(B, name)) { fprintf(stderr, "unknown scene %s\n", name.c_str()); return 1; }
    HipFlatScene F;
    const char *why = HipFlattenScene(B.scene, &F);
    if (why) { fprintf(stderr, "HipFlattenScene: %s\n", why); return 3; }
    const pvol_scene &S = F.scene;
    Blob out;
    out.puti1("vol.kind", S.volume.kind);
    float ex[6] = {S.volume.extent_min[0], S.volume.extent_min[1], S.volume.extent_min[2], S.volume.extent_max[0], S.volume.extent_max[1], S.volume.extent_max[2]};
    out.putf("vol.extent", ex, 6);
    out.putf("vol.w2v", S.volume.world_to_volume, 16); out.putf("vol.v2w", S.volume.volume_to_world, 16);
    out.putf("vol.sigma_a", S.volume.sigma_a.c, 30); out.putf("vol.sigma_s", S.volume.sigma_s.c, 30); out.putf("vol.le", S.volume.le.c, 30);
    out.putf1("vol.g", S.volume.g);
    int32_t dims[3] = {S.volume.nx, S.volume.ny, S.volume.nz};
    out.put("vol.dims", blob::I32, dims, 3);
    if (S.volume.kind == PVOL_VOLUME_GRID && name != "volumescene_grid128")
        out.putf("vol.density", S.volume.density, (size_t)S.volume.nx * S.volume.ny * S.volume.nz);
    std::vector<int32_t> lk, tm, tf, sm, sfl, mk;
    std::vector<float> lpos, ldir, l2w, w2l, lint, lcos, tp, so2w, sw2o, sf, kd, kr, kt, ior, vn;
    for (uint32_t i = 0; i < S.n_lights; ++i) {
        const pvol_light &l = S.lights[i];
        lk.push_back(l.kind);
        lpos.insert(lpos.end(), l.pos, l.pos + 3); ldir.insert(ldir.end(), l.dir, l.dir + 3);
        l2w.insert(l2w.end(), l.light_to_world, l.light_to_world + 16); w2l.insert(w2l.end(), l.world_to_light, l.world_to_light + 16);
        lint.insert(lint.end(), l.intensity.c, l.intensity.c + 30);
        lcos.push_back(l.cos_total_width); lcos.push_back(l.cos_falloff_start);
    }
    out.put("lights.kind", blob::I32, lk.data(), lk.size());
    out.putf("lights.pos", lpos); out.putf("lights.dir", ldir); out.putf("lights.l2w", l2w); out.putf("lights.w2l", w2l);
    out.putf("lights.intensity", lint); out.putf("lights.cos", lcos);
    for (uint32_t i = 0; i < S.n_triangles; ++i) {
        const pvol_triangle &t = S.triangles[i];
        tp.insert(tp.end(), &t.p[0][0], &t.p[0][0] + 9);
        tm.push_back(t.material); tf.push_back(t.flip_normal);
    }
    out.putf("tris.p", tp);
    out.put("tris.material", blob::I32, tm.data(), tm.size());
    out.put("tris.flip", blob::I32, tf.data(), tf.size());
    if (S.n_spheres) {
        for (uint32_t i = 0; i < S.n_spheres; ++i) {
            const pvol_sphere &q = S.spheres[i];
            so2w.insert(so2w.end(), q.object_to_world, q.object_to_world + 16); sw2o.insert(sw2o.end(), q.world_to_object, q.world_to_object + 16);
            float f[6] = {q.radius, q.z_min, q.z_max, q.theta_min, q.theta_max, q.phi_max};
            sf.insert(sf.end(), f, f + 6);
            sm.push_back(q.material); sfl.push_back(q.flip_normal);
        }
        out.putf("spheres.o2w", so2w); out.putf("spheres.w2o", sw2o); out.putf("spheres.f", sf);
        out.put("spheres.material", blob::I32, sm.data(), sm.size());
        out.put("spheres.flip", blob::I32, sfl.data(), sfl.size());
    }
    for (uint32_t i = 0; i < S.n_materials; ++i) {
        const pvol_material &m = S.materials[i];
        mk.push_back(m.kind);
        kd.insert(kd.end(), m.kd.c, m.kd.c + 30); kr.insert(kr.end(), m.kr.c, m.kr.c + 30); kt.insert(kt.end(), m.kt.c, m.kt.c + 30);
        ior.push_back(m.ior); vn.push_back(m.vn);
    }
    out.put("mats.kind", blob::I32, mk.data(), mk.size());
    out.putf("mats.kd", kd); out.putf("mats.kr", kr); out.putf("mats.kt", kt); out.putf("mats.ior", ior); out.putf("mats.vn", vn);
    float w[6] = {S.world_min[0], S.world_min[1], S.world_min[2], S.world_max[0], S.world_max[1], S.world_max[2]};
    out.putf("world", w, 6);
    out.putf("cie.x", S.cie_x.c, 30); out.putf("cie.y", S.cie_y.c, 30); out.putf("cie.z", S.cie_z.c, 30);
    out.putf1("xyz_scale", S.xyz_scale);
    // the BVH's own leaf order, so that the test can show it differs from creation order (why the walk sorts)
    std::vector<int32_t> leafIds;
    if (const BVHAccel *bvh = dynamic_cast<const BVHAccel *>(B.scene->aggregate))
        for (size_t i = 0; i < bvh->primitives.size(); ++i)
            leafIds.push_back((int32_t)static_cast<const GeometricPrimitive *>(bvh->primitives[i].GetPtr())->shape->shapeId);
    out.put("bvh.leaf_shape_ids", blob::I32, leafIds.data(), leafIds.size());
    return out.save(outPath) ? 0 : 1;
}

static int cmdTables(const char *outPath) {
    Blob out;
    std::vector<float> v;
    putSpec(v, SampledSpectrum::X); out.putf("cie.x", v); v.clear();
    putSpec(v, SampledSpectrum::Y); out.putf("cie.y", v); v.clear();
    putSpec(v, SampledSpectrum::Z); out.putf("cie.z", v); v.clear();
    out.putf1("spectrum1.y", Spectrum(1.f).y());
    // RNG: first 1300 draws (two table regenerations) for a few seeds
    const uint32_t seeds[6] = {0u, 1u, 31u, 62u, 4095u, 5489u};
    out.put("rng.seeds", blob::U32, seeds, 6);
    std::vector<uint32_t> draws;
    std::vector<float> fl;
    for (int s = 0; s < 6; ++s) {
        RNG r(seeds[s]);
        for (int i = 0; i < 1300; ++i) draws.push_back(r.RandomUInt());
        RNG r2(seeds[s]);
        for (int i = 0; i < 64; ++i) fl.push_back(r2.RandomFloat());
    }
    out.putu("rng.draws", draws);
    out.putf("rng.floats", fl);
    // PermutedHalton(6, RNG(31*t)) for t = 0,1,2: samples 1..512 plus a few large indices
    std::vector<float> hal;
    std::vector<uint32_t> halIdx;
    for (uint32_t i = 1; i <= 512; ++i) halIdx.push_back(i);
    const uint32_t big[8] = {4096u, 65535u, 1000003u, 16777216u, 43200000u, 123456789u, 2147483647u, 4294967295u};
    for (int i = 0; i < 8; ++i) halIdx.push_back(big[i]);
    for (uint32_t t = 0; t < 3; ++t) {
        RNG r(31 * t);
        PermutedHalton h(6, r);
        for (size_t i = 0; i < halIdx.size(); ++i) { float u[6]; h.Sample(halIdx[i], u); hal.insert(hal.end(), u, u + 6); }
    }
    out.putu("halton.index", halIdx);
    out.putf("halton.samples", hal);
    // LDShuffleScrambled1D/2D as Li() calls them (nSamples = 1, nPixel = n) and as LDPixelSample does
    const int ns[5] = {1, 7, 34, 171, 600};
    std::vector<float> l1, l2;
    std::vector<uint32_t> lnext;
    for (int k = 0; k < 5; ++k) {
        RNG r(1000 + k);
        std::vector<float> a(ns[k]), b(2 * ns[k]);
        LDShuffleScrambled1D(1, ns[k], &a[0], r);
        LDShuffleScrambled2D(1, ns[k], &b[0], r);
        l1.insert(l1.end(), a.begin(), a.end());
        l2.insert(l2.end(), b.begin(), b.end());
        lnext.push_back(r.RandomUInt());
    }
    out.put("ld.n", blob::I32, ns, 5);
    out.putf("ld.1d", l1); out.putf("ld.2d", l2); out.putu("ld.next", lnext);
    {
        RNG r(77);
        std::vector<float> a(4 * 16), b(2 * 4 * 16);
        LDShuffleScrambled1D(4, 16, &a[0], r);
        LDShuffleScrambled2D(4, 16, &b[0], r);
        out.putf("ld.1d_4x16", a); out.putf("ld.2d_4x16", b);
        out.putu1("ld.next_4x16", r.RandomUInt());
    }
    // sampling routines on a lattice
    std::vector<float> us, sph, cone, disk, cosh;
    for (int i = 0; i < 17; ++i)
        for (int j = 0; j < 17; ++j) {
            float u1 = std::min(i / 16.f, OneMinusEpsilon), u2 = std::min(j / 16.f, OneMinusEpsilon);
            us.push_back(u1); us.push_back(u2);
            Vector a = UniformSampleSphere(u1, u2); sph.push_back(a.x); sph.push_back(a.y); sph.push_back(a.z);
            Vector b = UniformSampleCone(u1, u2, 0.95f); cone.push_back(b.x); cone.push_back(b.y); cone.push_back(b.z);
            float dx, dy; ConcentricSampleDisk(u1, u2, &dx, &dy); disk.push_back(dx); disk.push_back(dy);
            Vector c = CosineSampleHemisphere(u1, u2); cosh.push_back(c.x); cosh.push_back(c.y); cosh.push_back(c.z);
        }
    out.putf("mc.u", us); out.putf("mc.sphere", sph); out.putf("mc.cone95", cone); out.putf("mc.disk", disk); out.putf("mc.coshemi", cosh);
    // phase functions
    std::vector<float> ph;
    for (int i = 0; i <= 32; ++i) {
        float c = -1.f + i / 16.f;
        Vector w(0, 0, 1), wp(sqrtf(std::max(0.f, 1 - c * c)), 0, c);
        ph.push_back(c);
        ph.push_back(PhaseHG(w, wp, 0.f)); ph.push_back(PhaseHG(w, wp, 0.6f)); ph.push_back(PhaseHG(w, wp, -0.3f)); ph.push_back(PhaseMieHazy(w, wp));
    }
    out.putf("phase", ph);
    // RGB -> spectrum for the colours the scenes use (core/paramset.cpp:103 path)
    const float cols[7] = {.05f, .1f, 150.f, .01f, .001f, 15000.f, 4.f};
    std::vector<float> sp, spy;
    for (int i = 0; i < 7; ++i) { float rgb[3] = {cols[i], cols[i], cols[i]}; Spectrum s = Spectrum::FromRGB(rgb); putSpec(sp, s); spy.push_back(s.y()); }
    out.putf("rgb.grey", cols, 7); out.putf("rgb.spectra", sp); out.putf("rgb.y", spy);
    // the Smits tables behind FromRGB (core/spectrum.cpp:699-958, public externs of spectrum.h:77-92), as compiled into the reference,
    // and FromRGB of some coloured inputs, both kinds (spectrum.cpp:154-241): what the scene-file front end is checked against
    out.putf("rgb2spect.lambda", RGB2SpectLambda, nRGB2SpectSamples);
    {
        const float *refl[7] = {RGBRefl2SpectWhite, RGBRefl2SpectCyan, RGBRefl2SpectMagenta, RGBRefl2SpectYellow, RGBRefl2SpectRed, RGBRefl2SpectGreen, RGBRefl2SpectBlue};
        const float *illum[7] = {RGBIllum2SpectWhite, RGBIllum2SpectCyan, RGBIllum2SpectMagenta, RGBIllum2SpectYellow, RGBIllum2SpectRed, RGBIllum2SpectGreen, RGBIllum2SpectBlue};
        std::vector<float> a, b;
        for (int i = 0; i < 7; ++i) { a.insert(a.end(), refl[i], refl[i] + nRGB2SpectSamples); b.insert(b.end(), illum[i], illum[i] + nRGB2SpectSamples); }
        out.putf("rgb2spect.refl", a); out.putf("rgb2spect.illum", b);
        const float rgbs[8][3] = {{.8f, .2f, .1f}, {.1f, .7f, .3f}, {.2f, .3f, .9f}, {.9f, .9f, .1f}, {.05f, .6f, .6f}, {.7f, .1f, .7f}, {1.f, 1.f, 1.f}, {0.f, 0.f, 0.f}};
        std::vector<float> in, outR, outI;
        for (int i = 0; i < 8; ++i) {
            in.insert(in.end(), rgbs[i], rgbs[i] + 3);
            putSpec(outR, SampledSpectrum::FromRGB(rgbs[i], SPECTRUM_REFLECTANCE));
            putSpec(outI, SampledSpectrum::FromRGB(rgbs[i], SPECTRUM_ILLUMINANT));
        }
        out.putf("rgb.colour_in", in); out.putf("rgb.colour_refl", outR); out.putf("rgb.colour_illum", outI);
    }
    return out.save(outPath) ? 0 : 1;
}

// Li()/Transmittance() records.  rays blob: rays.o rays.d [3n], rays.mint rays.maxt rays.time rays.u [n], rays.skip u32[n];
// streams.seed streams.first streams.n u32[s], streams.start u64[s].  photons blob: p wi [3m], alpha [30m].
static int cmdLi(const std::string &name, const char *photonPath, const char *rayPath, const char *outPath, int argc, char **argv) {
    BuiltScene B;
    memset(&B.nx, 0, sizeof(int) * 3);
    if (!buildByName(B, name)) return 1;
    for (int i = 0; i + 1 < argc; i += 2) {  // parameter overrides
        if (!strcmp(argv[i], "stepsize")) B.stepSize = (float)atof(argv[i + 1]);
        else if (!strcmp(argv[i], "nused")) B.nUsed = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "maxdist")) B.maxDist = (float)atof(argv[i + 1]);
    }
    Blob pb, rb;
    if (!rb.load(rayPath)) { fprintf(stderr, "cannot read %s\n", rayPath); return 1; }
    // The shooter object Li() reads volumeMap from (photonvolume.cpp:118), made as core/api.cpp:1225-1230 makes it.
    ParamSet surfp, volp;
    PhotonShooter *psh = CreatePhotonShooter(surfp, volp);
    if (strcmp(photonPath, "-")) {
        if (!pb.load(photonPath)) { fprintf(stderr, "cannot read %s\n", photonPath); return 1; }
        size_t m = pb.get("p").count() / 3;
        const float *pp = pb.get("p").f32(), *pw = pb.get("wi").f32(), *pa = pb.get("alpha").f32();
        vector<Photon> photons;
        for (size_t i = 0; i < m; ++i)
            photons.push_back(Photon(Point(pp[3 * i], pp[3 * i + 1], pp[3 * i + 2]), specFrom(pa + 30 * i), Vector(pw[3 * i], pw[3 * i + 1], pw[3 * i + 2])));
        if (m) psh->volumeMap = new KdTree<Photon>(photons);  // photonshooter.cpp:502-503
    }
    ParamSet vp;
    vp.AddFloat("stepsize", &B.stepSize, 1);
    vp.AddInt("nused", &B.nUsed, 1);
    vp.AddFloat("maxdist", &B.maxDist, 1);
    PhotonVolumeIntegrator *vi = CreatePhotonVolumeIntegrator(vp, psh);
    CaptureRenderer renderer(vi);
    Sample sample(NULL, NULL, vi, B.scene);  // RequestSamples: tauSampleOffset = 0, scatterSampleOffset = 1
    MemoryArena arena;

    size_t n = rb.get("rays.u").count();
    const float *ro = rb.get("rays.o").f32(), *rd = rb.get("rays.d").f32(), *rmin = rb.get("rays.mint").f32(), *rmax = rb.get("rays.maxt").f32(),
                *rt = rb.get("rays.time").f32(), *ru = rb.get("rays.u").f32();
    const uint32_t *rskip = rb.get("rays.skip").u32();
    size_t ns = rb.get("streams.seed").count();
    const uint32_t *sseed = rb.get("streams.seed").u32(), *sfirst = rb.get("streams.first").u32(), *sn = rb.get("streams.n").u32();
    const uint64_t *sstart = rb.get("streams.start").u64();
    bool transOnly = rb.has("transmittance_only");
    std::vector<float> Lv(30 * n, 0.f), T(30 * n, 0.f);
    std::vector<uint32_t> draws(n, 0), nextRng(ns, 0);
    std::vector<uint64_t> send(ns, 0);
    for (size_t s = 0; s < ns; ++s) {
        RNG rng(sseed[s]), shadow(sseed[s]);
        uint64_t total = 0;
        for (uint64_t k = 0; k < sstart[s]; ++k) { rng.RandomUInt(); shadow.RandomUInt(); }
        total = sstart[s];
        for (uint32_t k = 0; k < sn[s]; ++k) {
            size_t i = sfirst[s] + k;
            for (uint32_t q = 0; q < rskip[i]; ++q) { rng.RandomUInt(); shadow.RandomUInt(); }
            total += rskip[i];
            RayDifferential ray(Point(ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]), Vector(rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]), rmin[i], rmax[i], rt[i]);
            Spectrum Tr(1.f), L(0.f);
            if (transOnly) {
                Tr = vi->Transmittance(B.scene, &renderer, ray, NULL, rng, arena);
            } else {
                sample.oneD[vi->scatterSampleOffset][0] = ru[i];
                L = vi->Li(B.scene, &renderer, ray, &sample, rng, &Tr, arena);
            }
            arena.FreeAll();
            uint64_t d = drawsBetween(shadow, rng, 10000000);
            draws[i] = (uint32_t)d;
            total += d;
            for (int b = 0; b < 30; ++b) { Lv[30 * i + b] = L.c[b]; T[30 * i + b] = Tr.c[b]; }
        }
        send[s] = total;
        nextRng[s] = rng.RandomUInt();
    }
    Blob out;
    out.putf("Lv", Lv); out.putf("T", T); out.putu("draws", draws); out.putu("next_rng", nextRng);
    out.put("streams.end", blob::U64, send.data(), send.size());
    float pf[3] = {B.stepSize, B.maxDist, B.shooterStep};
    out.putf("params.f", pf, 3);
    out.puti1("params.nused", B.nUsed);
    return out.save(outPath) ? 0 : 1;
}

// The caller and the consumer of Li(): LDSampler, PerspectiveCamera, ImageFilm, made by the reference's own
// Create*() functions.  SamplerRendererTask itself cannot be linked (it is a Task: core/parallel.cpp), so the
// sample loop of SamplerRendererTask::Run (samplerrenderer.cpp:85-146) is walked here in the same order with
// the reference's objects; the surface term of SamplerRenderer::Li is left out (Ls = Lvi), as in include/pvol.h.
static int cmdRender(const std::string &name, const char *photonPath, const char *outPath, int argc, char **argv) {
    BuiltScene B;
    memset(&B.nx, 0, sizeof(int) * 3);
    if (!buildByName(B, name)) return 1;
    int xres = 32, yres = 18, spp = 4, nTasks = 8;
    // `surface CAUSTIC.bin`: also run the reference's PhotonIntegrator (integrators/photonmap.cpp) with the scene's own
    // SurfaceIntegrator parameters on a caustic map built from the given photons (oracle-shot: the reference's shooter does
    // not link here); Ls then is T * Lsurface + Lvi as SamplerRenderer::Li composes it
    const char *causticPath = NULL;
    int surfNused = 300, surfGatherSamples = 64;
    float surfMaxDist = .15f;
    bool surfFinalGather = true;
    std::vector<uint32_t> tasks;
    for (int i = 0; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "stepsize")) B.stepSize = (float)atof(argv[i + 1]);
        else if (!strcmp(argv[i], "nused")) B.nUsed = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "maxdist")) B.maxDist = (float)atof(argv[i + 1]);
        else if (!strcmp(argv[i], "xres")) xres = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "yres")) yres = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "spp")) spp = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "ntasks")) nTasks = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "surface")) causticPath = argv[i + 1];
        else if (!strcmp(argv[i], "surfnused")) surfNused = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "surfmaxdist")) surfMaxDist = (float)atof(argv[i + 1]);
        else if (!strcmp(argv[i], "surffinalgather")) surfFinalGather = atoi(argv[i + 1]) != 0;
        else if (!strcmp(argv[i], "tasks")) {
            const char *q = argv[i + 1];
            while (*q) { tasks.push_back((uint32_t)strtoul(q, (char **)&q, 10)); if (*q == ',') ++q; }
        }
    }
    if (tasks.empty()) for (int t = 0; t < nTasks; ++t) tasks.push_back(t);
    ParamSet surfp, volp;
    PhotonShooter *psh = CreatePhotonShooter(surfp, volp);
    Blob pb;
    if (strcmp(photonPath, "-")) {
        if (!pb.load(photonPath)) { fprintf(stderr, "cannot read %s\n", photonPath); return 1; }
        size_t m = pb.get("p").count() / 3;
        const float *pp = pb.get("p").f32(), *pw = pb.get("wi").f32(), *pa = pb.get("alpha").f32();
        vector<Photon> photons;
        for (size_t i = 0; i < m; ++i)
            photons.push_back(Photon(Point(pp[3 * i], pp[3 * i + 1], pp[3 * i + 2]), specFrom(pa + 30 * i), Vector(pw[3 * i], pw[3 * i + 1], pw[3 * i + 2])));
        if (m) psh->volumeMap = new KdTree<Photon>(photons);
    }
    ParamSet vp;
    vp.AddFloat("stepsize", &B.stepSize, 1);
    vp.AddInt("nused", &B.nUsed, 1);
    vp.AddFloat("maxdist", &B.maxDist, 1);
    PhotonVolumeIntegrator *vi = CreatePhotonVolumeIntegrator(vp, psh);
    PhotonIntegrator *si = NULL;
    Blob cb;
    if (causticPath) {
        if (strcmp(causticPath, "-")) {
            if (!cb.load(causticPath)) { fprintf(stderr, "cannot read %s\n", causticPath); return 1; }
            size_t m = cb.get("p").count() / 3;
            const float *pp = cb.get("p").f32(), *pw = cb.get("wo").f32(), *pa = cb.get("alpha").f32();
            vector<Photon> photons;
            for (size_t i = 0; i < m; ++i)
                photons.push_back(Photon(Point(pp[3 * i], pp[3 * i + 1], pp[3 * i + 2]), specFrom(pa + 30 * i), Vector(pw[3 * i], pw[3 * i + 1], pw[3 * i + 2])));
            if (m) psh->causticMap = new KdTree<Photon>(photons);
            psh->nCausticPaths = (int)cb.get("n_paths").u32()[0];
        }
        ParamSet sp;
        sp.AddInt("nused", &surfNused, 1);
        sp.AddFloat("maxdist", &surfMaxDist, 1);
        sp.AddBool("finalgather", &surfFinalGather, 1);
        sp.AddInt("finalgathersamples", &surfGatherSamples, 1);
        si = CreatePhotonMapSurfaceIntegrator(sp, psh);
    }
    CaptureRenderer renderer(vi, si);

    // Film "image" + PixelFilter "gaussian" (defaults), Camera "perspective", Sampler "lowdiscrepancy": core/api.cpp:1221-1288
    ParamSet filtp, filmp, camp, sampp;
    Filter *filter = CreateGaussianFilter(filtp);
    std::string fname = "ref_capture_unused.tga";
    filmp.AddInt("xresolution", &xres, 1);
    filmp.AddInt("yresolution", &yres, 1);
    filmp.AddString("filename", &fname, 1);
    ImageFilm *film = CreateImageFilm(filmp, filter);
    camp.AddFloat("fov", &B.fov, 1);
    Transform *c2w = keep(B.camToWorld);
    AnimatedTransform ac2w(c2w, 0.f, c2w, 1.f);
    PerspectiveCamera *camera = CreatePerspectiveCamera(camp, ac2w, film);
    sampp.AddInt("pixelsamples", &spp, 1);
    LDSampler *mainSampler = CreateLowDiscrepancySampler(sampp, film, camera);
    Sample *origSample = new Sample(mainSampler, si, vi, B.scene);

    Blob out;
    std::vector<float> r2c, c2wm, ftab(film->filterTable, film->filterTable + 256);
    putMat(r2c, camera->RasterToCamera.m);
    putMat(c2wm, camera->CameraToWorld.startTransform->m);
    out.putf("camera.raster_to_camera", r2c);
    out.putf("camera.camera_to_world", c2wm);
    float shutter[4] = {camera->shutterOpen, camera->shutterClose, camera->lensRadius, camera->focalDistance};
    out.putf("camera.shutter_lens", shutter, 4);
    out.putf("film.filter_table", ftab);
    float fw[2] = {filter->xWidth, filter->yWidth};
    out.putf("film.filter_width", fw, 2);
    int32_t ext[4];
    film->GetSampleExtent(&ext[0], &ext[1], &ext[2], &ext[3]);
    out.put("sampler.extent", blob::I32, ext, 4);
    int32_t smpI[6] = {xres, yres, mainSampler->samplesPerPixel, nTasks, (int32_t)vi->tauSampleOffset, (int32_t)vi->scatterSampleOffset};
    out.put("sampler.i", blob::I32, smpI, 6);
    std::vector<uint32_t> n1d(origSample->n1D.begin(), origSample->n1D.end()), n2d(origSample->n2D.begin(), origSample->n2D.end());
    out.putu("sampler.n1d", n1d);
    out.putu("sampler.n2d", n2d);
    out.putu("tasks", tasks);
    float pf[3] = {B.stepSize, B.maxDist, B.shooterStep};
    out.putf("params.f", pf, 3);
    out.puti1("params.nused", B.nUsed);

    std::vector<float> sImg, sTime, sLens, sTau, sScat, rayO, rayD, rayT, xyzT, surfXYZ;
    std::vector<uint32_t> skip, nextRng, nSamples, surfDraws;
    std::vector<int32_t> windows;
    std::vector<uint64_t> endDraws;
    MemoryArena arena;
    for (size_t ti = 0; ti < tasks.size(); ++ti) {
        int taskNum = (int)tasks[ti];
        int w[4] = {0, 0, 0, 0};
        mainSampler->ComputeSubWindow(taskNum, nTasks, &w[0], &w[1], &w[2], &w[3]);
        for (int k = 0; k < 4; ++k) windows.push_back(w[k]);
        Sampler *sampler = mainSampler->GetSubSampler(taskNum, nTasks);
        if (!sampler) { endDraws.push_back(0); nextRng.push_back(0); nSamples.push_back(0); continue; }
        RNG rng(taskNum), shadow(taskNum);
        uint64_t total = 0;
        uint32_t count = 0;
        int maxSamples = sampler->MaximumSampleCount();
        Sample *samples = origSample->Duplicate(maxSamples);
        std::vector<Spectrum> LsAll(maxSamples);
        int sampleCount;
        while ((sampleCount = sampler->GetMoreSamples(samples, rng)) > 0) {
            uint64_t ds = drawsBetween(shadow, rng, 100000000);
            total += ds;
            for (int i = 0; i < sampleCount; ++i) {
                RayDifferential ray;
                float rayWeight = camera->GenerateRayDifferential(samples[i], &ray);
                ray.ScaleDifferentials(1.f / sqrtf(sampler->samplesPerPixel));
                Intersection isect;
                Spectrum Lsurf(0.f);
                const bool hitSurface = B.scene->Intersect(ray, &isect);   // SamplerRenderer::Li, samplerrenderer.cpp:236-249
                if (hitSurface && si) Lsurf = si->Li(B.scene, &renderer, ray, isect, &samples[i], rng, arena);
                const uint64_t dsurf = si ? drawsBetween(shadow, rng, 100000000) : 0;
                total += dsurf;
                Spectrum T(1.f);
                Spectrum Lvi = vi->Li(B.scene, &renderer, ray, &samples[i], rng, &T, arena);
                Spectrum Ls = rayWeight * (T * Lsurf + Lvi);
                if (Ls.HasNaNs() || Ls.y() < -1e-5 || isinf(Ls.y())) Ls = Spectrum(0.f);
                uint64_t d = drawsBetween(shadow, rng, 100000000);
                total += d;
                LsAll[i] = Ls;
                float xyz[3];
                Ls.ToXYZ(xyz);
                sImg.push_back(samples[i].imageX); sImg.push_back(samples[i].imageY);
                sTime.push_back(samples[i].time);
                sLens.push_back(samples[i].lensU); sLens.push_back(samples[i].lensV);
                sTau.push_back(samples[i].oneD[vi->tauSampleOffset][0]);
                sScat.push_back(samples[i].oneD[vi->scatterSampleOffset][0]);
                rayO.push_back(ray.o.x); rayO.push_back(ray.o.y); rayO.push_back(ray.o.z);
                rayD.push_back(ray.d.x); rayD.push_back(ray.d.y); rayD.push_back(ray.d.z);
                rayT.push_back(ray.mint); rayT.push_back(ray.maxt);
                xyzT.push_back(xyz[0]); xyzT.push_back(xyz[1]); xyzT.push_back(xyz[2]); xyzT.push_back(T.y());
                skip.push_back((i == 0 ? (uint32_t)ds : 0u) + (uint32_t)dsurf);   // what the caller drew in front of this sample's volume Li()
                if (si) { float sx[3]; Lsurf.ToXYZ(sx); surfXYZ.push_back(sx[0]); surfXYZ.push_back(sx[1]); surfXYZ.push_back(sx[2]); surfDraws.push_back((uint32_t)dsurf); }
                ++count;
            }
            if (sampler->ReportResults(samples, NULL, NULL, NULL, sampleCount))   // samplerrenderer.cpp:137-146
                for (int i = 0; i < sampleCount; ++i) film->AddSample(samples[i], LsAll[i]);
            arena.FreeAll();
        }
        endDraws.push_back(total);
        nextRng.push_back(rng.RandomUInt());
        nSamples.push_back(count);
        delete sampler;
    }
    out.putf("samples.image", sImg); out.putf("samples.time", sTime); out.putf("samples.lens", sLens);
    out.putf("samples.tau", sTau); out.putf("samples.scatter", sScat);
    out.putf("rays.o", rayO); out.putf("rays.d", rayD); out.putf("rays.t", rayT); out.putf("xyzT", xyzT);
    out.putu("rays.skip", skip); out.putu("next_rng", nextRng); out.putu("task.n_samples", nSamples);
    if (si) {
        out.putf("surf.xyz", surfXYZ); out.putu("surf.draws", surfDraws);
        float sf[1] = {surfMaxDist};
        int32_t sn[3] = {surfNused, surfFinalGather ? 1 : 0, psh->nCausticPaths};
        out.putf("surf.params.f", sf, 1);
        out.put("surf.params.i", blob::I32, sn, 3);
    }
    out.put("task.window", blob::I32, windows.data(), windows.size());
    out.put("task.end_draw", blob::U64, endDraws.data(), endDraws.size());
    std::vector<float> pix;
    for (int y = 0; y < yres; ++y)
        for (int x = 0; x < xres; ++x) {
            const ImageFilm::Pixel &px = (*film->pixels)(x, y);
            pix.push_back(px.Lxyz[0]); pix.push_back(px.Lxyz[1]); pix.push_back(px.Lxyz[2]); pix.push_back(px.weightSum);
        }
    out.putf("film.pixels", pix);
    float *rgb = NULL;
    int fwid = 0, fhei = 0;
    film->WriteRGB(&rgb, &fwid, &fhei, 1.f);
    out.putf("film.rgb", rgb, (size_t)3 * fwid * fhei);
    return out.save(outPath) ? 0 : 1;
}

// Building blocks of the photon shooter, evaluated by the reference's objects.
static int cmdUnits(const std::string &name, const char *outPath) {
    BuiltScene B;
    memset(&B.nx, 0, sizeof(int) * 3);
    if (!buildByName(B, name)) return 1;
    Blob out;
    const Scene *scene = B.scene;
    // light power CDF inputs (core/integrator.cpp:261-268)
    std::vector<float> pw;
    for (size_t i = 0; i < B.lights.size(); ++i) pw.push_back(B.lights[i]->Power(scene).y());
    out.putf("light.power_y", pw);
    // emission: Sample_L(scene, ls, u1, u2, time, &ray, &Ns, &pdf) on a lattice + RNG points
    RNG rng(2024);
    std::vector<float> eu, eo;
    for (size_t li = 0; li < B.lights.size(); ++li)
        for (int k = 0; k < 64; ++k) {
            float u0 = rng.RandomFloat(), u1 = rng.RandomFloat();
            Ray ray; Normal Ns; float pdf;
            Spectrum Le = B.lights[li]->Sample_L(scene, LightSample(u0, u1, 0.5f), 0.25f, 0.75f, 0.f, &ray, &Ns, &pdf);
            eu.push_back((float)li); eu.push_back(u0); eu.push_back(u1);
            float rec[10] = {ray.o.x, ray.o.y, ray.o.z, ray.d.x, ray.d.y, ray.d.z, Ns.x, Ns.y, Ns.z, pdf};
            eo.insert(eo.end(), rec, rec + 10);
            putSpec(eo, Le);
        }
    out.putf("emit.in", eu); out.putf("emit.out", eo);
    // Sample_L(p, ...) at random points of the world bound
    const BBox &wb = scene->WorldBound();
    std::vector<float> sp, so;
    for (size_t li = 0; li < B.lights.size(); ++li)
        for (int k = 0; k < 64; ++k) {
            Point p = wb.Lerp(rng.RandomFloat(), rng.RandomFloat(), rng.RandomFloat());
            Vector wi; float pdf; VisibilityTester vis;
            Spectrum L = B.lights[li]->Sample_L(p, 0.f, LightSample(0.5f, 0.5f, 0.5f), 0.f, &wi, &pdf, &vis);
            sp.push_back((float)li); sp.push_back(p.x); sp.push_back(p.y); sp.push_back(p.z);
            float rec[12] = {wi.x, wi.y, wi.z, pdf, vis.r.o.x, vis.r.o.y, vis.r.o.z, vis.r.d.x, vis.r.d.y, vis.r.d.z, vis.r.mint, vis.r.maxt};
            so.insert(so.end(), rec, rec + 12);
            putSpec(so, L);
        }
    out.putf("sample.in", sp); out.putf("sample.out", so);
    // closest hit / any hit / BSDF sampling along random rays
    std::vector<float> ri, ro_, bs;
    MemoryArena arena;
    int nr = 0;
    const bool aimed = B.meshRadius > 0.f;
    const int wantRays = aimed ? 2000 : 600;
    for (int k = 0; k < (aimed ? 40000 : 4000) && nr < wantRays; ++k) {
        Point o = wb.Lerp(rng.RandomFloat(), rng.RandomFloat(), rng.RandomFloat());
        Vector d = UniformSampleSphere(rng.RandomFloat(), rng.RandomFloat());
        if (aimed && (k & 1)) {   // towards a random point of the mesh's bounding ball
            Vector off = B.meshRadius * rng.RandomFloat() * UniformSampleSphere(rng.RandomFloat(), rng.RandomFloat());
            d = Normalize(Point(B.meshCenter[0], B.meshCenter[1], B.meshCenter[2]) + off - o);
        }
        float maxt = (k % 3 == 0) ? 4.f * rng.RandomFloat() : INFINITY;
        RayDifferential ray(o, d, 0.f, maxt, 0.f);
        Intersection isect;
        bool hitP = scene->IntersectP(ray);
        bool hit = scene->Intersect(ray, &isect);
        if (!hit && (k % 4)) continue;  // keep some misses
        ++nr;
        float in[7] = {o.x, o.y, o.z, d.x, d.y, d.z, maxt};
        ri.insert(ri.end(), in, in + 7);
        float rec[12] = {hit ? 1.f : 0.f, hitP ? 1.f : 0.f, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (hit) {
            rec[2] = ray.maxt; rec[3] = isect.dg.p.x; rec[4] = isect.dg.p.y; rec[5] = isect.dg.p.z;
            rec[6] = isect.dg.nn.x; rec[7] = isect.dg.nn.y; rec[8] = isect.dg.nn.z;
            rec[9] = isect.dg.dpdu.x; rec[10] = isect.dg.dpdu.y; rec[11] = isect.dg.dpdu.z;
        }
        ro_.insert(ro_.end(), rec, rec + 12);
        // BSDF::Sample_f exactly as followPhoton calls it (photonshooter.cpp:202-203), once with a
        // polychromatic alpha and once with a monochromatic one (dispersion path)
        for (int variant = 0; variant < 2; ++variant) {
            float u0 = rng.RandomFloat(), u1 = rng.RandomFloat(), uc = rng.RandomFloat();
            int bin = (int)(rng.RandomFloat() * 30);
            float rec2[41];
            memset(rec2, 0, sizeof(rec2));
            rec2[0] = u0; rec2[1] = u1; rec2[2] = uc; rec2[3] = (float)bin;
            if (hit) {
                Spectrum alpha(1.f);
                alpha.lambda = -1;
                if (variant == 1) { alpha = Spectrum(0.f); alpha.c[bin] = 0.7f; alpha.lambda = alpha.extractLambda(); }
                BSDF *bsdf = isect.GetBSDF(ray, arena);
                Vector wo = -ray.d, wi(0, 0, 0);
                float pdf = 0; BxDFType flags = BxDFType(0);
                Spectrum f = bsdf->Sample_f(wo, &wi, BSDFSample(u0, u1, uc), &pdf, BSDF_ALL, &flags, &alpha);
                rec2[4] = wi.x; rec2[5] = wi.y; rec2[6] = wi.z; rec2[7] = pdf; rec2[8] = (float)(int)flags;
                rec2[9] = (float)bsdf->NumComponents();
                rec2[10] = isect.primitive->dispersive() ? 1.f : 0.f;
                for (int b = 0; b < 30; ++b) rec2[11 + b] = f.c[b];
                arena.FreeAll();
            }
            bs.insert(bs.end(), rec2, rec2 + 41);
        }
    }
    out.putf("hit.in", ri); out.putf("hit.out", ro_); out.putf("bsdf.out", bs);
    // volume queries
    std::vector<float> vq, vo;
    const VolumeRegion *vr = scene->volumeRegion;
    BBox vb = vr->WorldBound();
    vb.Expand(0.5f);
    for (int k = 0; k < 256; ++k) {
        Point p = vb.Lerp(rng.RandomFloat(), rng.RandomFloat(), rng.RandomFloat());
        Vector d = UniformSampleSphere(rng.RandomFloat(), rng.RandomFloat());
        float len = 0.2f + 6.f * rng.RandomFloat();
        Ray r(p, d, 0.f, len, 0.f);
        float t0 = -1, t1 = -1;
        bool hit = vr->IntersectP(r, &t0, &t1);
        float step = 0.3f, offs = rng.RandomFloat();
        Spectrum tau = vr->tau(r, step, offs);
        float in[9] = {p.x, p.y, p.z, d.x, d.y, d.z, len, step, offs};
        vq.insert(vq.end(), in, in + 9);
        vo.push_back(hit ? 1.f : 0.f); vo.push_back(t0); vo.push_back(t1);
        putSpec(vo, tau);
        putSpec(vo, vr->sigma_a(p, d, 0.f));
        putSpec(vo, vr->sigma_s(p, d, 0.f));
        vo.push_back(vr->p(p, d, -d, 0.f));
    }
    out.putf("vol.in", vq); out.putf("vol.out", vo);
    if (B.volumeKind == PVOL_VOLUME_RAINBOW) {
        std::vector<float> rbw;
        RainbowVolume *rv = (RainbowVolume *)vr;
        float rgb[3] = {3, 3, 3};
        Spectrum Ld = Spectrum::FromRGB(rgb);
        for (int k = 0; k < 512; ++k) {
            // sweep the scattering angle through both bows plus random directions
            float th = (k < 400) ? Radians(38.f + 18.f * k / 400.f) : acosf(1 - 2 * rng.RandomFloat());
            Vector w(0, 0, 1), wi(sinf(th), 0, -cosf(th));
            Spectrum r = rv->rainbowReflection(Ld, w, wi);
            rbw.push_back(w.x); rbw.push_back(w.y); rbw.push_back(w.z); rbw.push_back(wi.x); rbw.push_back(wi.y); rbw.push_back(wi.z);
            putSpec(rbw, r);
        }
        std::vector<float> ldv; putSpec(ldv, Ld);
        out.putf("rainbow.Ld", ldv);
        out.putf("rainbow", rbw);
    }
    return out.save(outPath) ? 0 : 1;
}

int main(int argc, char **argv) {
    SampledSpectrum::Init();  // pbrtInit (core/api.cpp) does this
    if (argc >= 3 && !strcmp(argv[1], "tables")) return cmdTables(argv[2]);
    if (argc >= 4 && !strcmp(argv[1], "scene")) return cmdScene(argv[2], argv[3]);
    if (argc >= 4 && !strcmp(argv[1], "shimscene")) return cmdShimScene(argv[2], argv[3]);
    if (argc >= 4 && !strcmp(argv[1], "units")) return cmdUnits(argv[2], argv[3]);
    if (argc >= 5 && !strcmp(argv[1], "render")) return cmdRender(argv[2], argv[3], argv[4], argc - 5, argv + 5);
    if (argc >= 6 && !strcmp(argv[1], "li")) return cmdLi(argv[2], argv[3], argv[4], argv[5], argc - 6, argv + 6);
    fprintf(stderr,
            "usage: ref_capture tables OUT | scene NAME OUT | shimscene NAME OUT | units NAME OUT | li NAME PHOTONS|- RAYS OUT [stepsize v] [nused v] [maxdist v]\n");
    return 64;
}
