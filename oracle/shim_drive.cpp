// oracle/shim_drive.cpp -- TEST INFRASTRUCTURE (GPU).  The drop-in, end to end, inside the reference's own object graph:
// a BASELINE scene built through the reference's Create*() functions (ref_scenes.h), the reference's PhotonShooter holding
// the volume map, and BOTH volume integrators constructed beside each other -- the reference's PhotonVolumeIntegrator and
// the binding of integration/hip_photonvolume.cpp (HipPhotonVolumeIntegrator over libpvol.so) -- driven through the
// VolumeIntegrator virtuals exactly as SamplerRenderer drives them: RequestSamples (Sample's constructor), Preprocess(scene),
// Li(scene, renderer, ray, sample, rng, &T, arena) per camera sample with a live RNG.
//
//   shim_drive li NAME PHOTONS CASE OUT        the rays of a golden Li() case (tests/golden/li_*.bin) through both
//   shim_drive render NAME PHOTONS CASE OUT    whole SamplerRendererTasks of a golden render case, three ways: the reference
//                                              integrator, the binding called per sample from the reference's sampler loop,
//                                              and the binding's RenderTasks (the device tile driver) -- three films
//
// Built by `make -C oracle shim_drive` from the reference's objects (oracle/_ref/libpbrtref.a) + libpvol.so; lives in
// oracle/_ref/ (git-ignored, travels to the GPU box); run by tests/test_gpu_shim.py.  Needs a HIP device.
#include "ref_scenes.h"
#include "../integration/hip_photonvolume.cpp"   // the binding itself, as a maintainer would compile it into the renderer

#include <hip/hip_runtime_api.h>

class DriveRenderer : public Renderer {   // SamplerRenderer::Transmittance / Li (renderers/samplerrenderer.cpp:228-258) for one integrator
public:
    explicit DriveRenderer(VolumeIntegrator *v) : vi(v) {}
    void Render(const Scene *) {}
    Spectrum Li(const Scene *scene, const RayDifferential &ray, const Sample *sample, RNG &rng, MemoryArena &arena, Intersection *isect, Spectrum *T) const {
        Spectrum localT;
        if (!T) T = &localT;
        Intersection localIsect;
        if (!isect) isect = &localIsect;
        Spectrum L = 0.f;
        if (!scene->Intersect(ray, isect)) for (uint32_t i = 0; i < scene->lights.size(); ++i) L += scene->lights[i]->Le(ray);
        Spectrum Lvi = vi->Li(scene, this, ray, sample, rng, T, arena);
        return *T * L + Lvi;
    }
    Spectrum Transmittance(const Scene *scene, const RayDifferential &ray, const Sample *sample, RNG &rng, MemoryArena &arena) const {
        return vi->Transmittance(scene, this, ray, sample, rng, arena);
    }
    VolumeIntegrator *vi;
};

static Spectrum specFrom(const float *c) { Spectrum s(0.f); for (int i = 0; i < nSpectralSamples; ++i) s.c[i] = c[i]; s.lambda = s.extractLambda(); return s; }

static uint64_t drawsBetween(RNG &shadow, const RNG &live, uint64_t limit) {
    uint64_t k = 0;
    for (;;) {
        if (shadow.mti == live.mti && shadow.mt[0] == live.mt[0] && shadow.mt[1] == live.mt[1] && shadow.mt[397] == live.mt[397] && shadow.mt[623] == live.mt[623]) return k;
        shadow.RandomUInt();
        if (++k > limit) { fprintf(stderr, "drawsBetween: limit exceeded\n"); exit(2); }
    }
}

struct Pair {
    PhotonShooter *psh;
    PhotonVolumeIntegrator *ref;
    HipPhotonVolumeIntegrator *hip;
};

static bool makePair(BuiltScene &B, const char *photonPath, Pair &P) {
    ParamSet surfp, volp;
    P.psh = CreatePhotonShooter(surfp, volp);   // core/api.cpp:1225-1230
    Blob pb;
    if (strcmp(photonPath, "-")) {
        if (!pb.load(photonPath)) { fprintf(stderr, "cannot read %s\n", photonPath); return false; }
        size_t m = pb.get("p").count() / 3;
        const float *pp = pb.get("p").f32(), *pw = pb.get("wi").f32(), *pa = pb.get("alpha").f32();
        vector<Photon> photons;
        for (size_t i = 0; i < m; ++i)
            photons.push_back(Photon(Point(pp[3 * i], pp[3 * i + 1], pp[3 * i + 2]), specFrom(pa + 30 * i), Vector(pw[3 * i], pw[3 * i + 1], pw[3 * i + 2])));
        if (m) P.psh->volumeMap = new KdTree<Photon>(photons);   // photonshooter.cpp:502-503
    }
    ParamSet vp;
    vp.AddFloat("stepsize", &B.stepSize, 1);
    vp.AddInt("nused", &B.nUsed, 1);
    vp.AddFloat("maxdist", &B.maxDist, 1);
    P.ref = CreatePhotonVolumeIntegrator(vp, P.psh);
    bool deviceShoot = false;   // take the shooter's map: the same photons on both sides
    vp.AddBool("deviceshoot", &deviceShoot, 1);
    P.hip = dynamic_cast<HipPhotonVolumeIntegrator *>(CreateHipPhotonVolumeIntegrator(vp, surfp, P.psh));
    return P.hip != NULL;
}

static void applyOverrides(BuiltScene &B, const Blob &c) {
    const float *pf = c.get("params.f").f32();
    B.stepSize = pf[0]; B.maxDist = pf[1];
    B.nUsed = c.get("params.nused").i32()[0];
}

static int cmdLi(const std::string &name, const char *photonPath, const char *casePath, const char *outPath) {
    BuiltScene B;
    memset(&B.nx, 0, sizeof(int) * 3);
    if (!buildByName(B, name)) return 1;
    Blob rb;
    if (!rb.load(casePath)) { fprintf(stderr, "cannot read %s\n", casePath); return 1; }
    applyOverrides(B, rb);
    Pair P;
    if (!makePair(B, photonPath, P)) return 1;
    VolumeIntegrator *vis[2] = {P.ref, P.hip};
    DriveRenderer rr(P.ref), rh(P.hip);
    const Renderer *rens[2] = {&rr, &rh};
    Sample sampleRef(NULL, NULL, P.ref, B.scene), sampleHip(NULL, NULL, P.hip, B.scene);   // RequestSamples on each
    Sample *samples[2] = {&sampleRef, &sampleHip};
    const int scat[2] = {P.ref->scatterSampleOffset, P.hip->ScatterSampleOffset()};
    P.ref->Preprocess(B.scene, NULL, &rr);
    P.hip->Preprocess(B.scene, NULL, &rh);   // HipFlattenScene -> pvol_set_scene -> pvol_upload_photons(shooter's map)
    MemoryArena arena;
    size_t n = rb.get("rays.u").count();
    const float *ro = rb.get("rays.o").f32(), *rd = rb.get("rays.d").f32(), *rmin = rb.get("rays.mint").f32(), *rmax = rb.get("rays.maxt").f32(),
                *rt = rb.get("rays.time").f32(), *ru = rb.get("rays.u").f32();
    const uint32_t *rskip = rb.get("rays.skip").u32();
    size_t ns = rb.get("streams.seed").count();
    const uint32_t *sseed = rb.get("streams.seed").u32(), *sn = rb.get("streams.n").u32();
    const uint64_t *sstart = rb.get("streams.start").u64();
    Blob out;
    for (int w = 0; w < 2; ++w) {
        std::vector<float> Lv(30 * n, 0.f), T(30 * n, 0.f);
        std::vector<uint32_t> draws(n, 0), nextRng(ns, 0);
        size_t first = 0;
        for (size_t s = 0; s < ns; ++s) {
            RNG rng(sseed[s]), shadow(sseed[s]);
            for (uint64_t k = 0; k < sstart[s]; ++k) { rng.RandomUInt(); shadow.RandomUInt(); }
            for (uint32_t k = 0; k < sn[s]; ++k) {
                size_t i = first + k;
                for (uint32_t q = 0; q < rskip[i]; ++q) { rng.RandomUInt(); shadow.RandomUInt(); }
                RayDifferential ray(Point(ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]), Vector(rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]), rmin[i], rmax[i], rt[i]);
                Spectrum Tr(1.f);
                samples[w]->oneD[scat[w]][0] = ru[i];
                Spectrum L = vis[w]->Li(B.scene, rens[w], ray, samples[w], rng, &Tr, arena);
                arena.FreeAll();
                draws[i] = (uint32_t)drawsBetween(shadow, rng, 10000000);
                for (int b = 0; b < 30; ++b) { Lv[30 * i + b] = L.c[b]; T[30 * i + b] = Tr.c[b]; }
            }
            first += sn[s];
            nextRng[s] = rng.RandomUInt();
        }
        const std::string pre = w ? "hip." : "ref.";
        out.putf(pre + "Lv", Lv); out.putf(pre + "T", T); out.putu(pre + "draws", draws); out.putu(pre + "next_rng", nextRng);
    }
    return out.save(outPath) ? 0 : 1;
}

static int cmdRender(const std::string &name, const char *photonPath, const char *casePath, const char *outPath) {
    BuiltScene B;
    memset(&B.nx, 0, sizeof(int) * 3);
    if (!buildByName(B, name)) return 1;
    Blob cb;
    if (!cb.load(casePath)) { fprintf(stderr, "cannot read %s\n", casePath); return 1; }
    applyOverrides(B, cb);
    const int32_t *si = cb.get("sampler.i").i32();
    int xres = si[0], yres = si[1], spp = si[2], nTasks = si[3];
    std::vector<uint32_t> tasks(cb.get("tasks").u32(), cb.get("tasks").u32() + cb.get("tasks").count());
    Pair P;
    if (!makePair(B, photonPath, P)) return 1;
    VolumeIntegrator *vis[2] = {P.ref, P.hip};
    DriveRenderer rr(P.ref), rh(P.hip);
    const Renderer *rens[2] = {&rr, &rh};
    P.ref->Preprocess(B.scene, NULL, &rr);
    P.hip->Preprocess(B.scene, NULL, &rh);
    Blob out;
    ImageFilm *lastFilm = NULL;
    PerspectiveCamera *lastCamera = NULL;
    LDSampler *lastSampler = NULL;
    Sample *lastOrig = NULL;
    for (int w = 0; w < 2; ++w) {
        // Film "image" + PixelFilter "gaussian" (defaults), Camera "perspective", Sampler "lowdiscrepancy": core/api.cpp:1221-1288
        ParamSet filtp, filmp, camp, sampp;
        Filter *filter = CreateGaussianFilter(filtp);
        std::string fname = "shim_drive_unused.tga";
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
        Sample *origSample = new Sample(mainSampler, NULL, vis[w], B.scene);
        lastFilm = film; lastCamera = camera; lastSampler = mainSampler; lastOrig = origSample;
        std::vector<uint32_t> nextRng;
        MemoryArena arena;
        for (size_t ti = 0; ti < tasks.size(); ++ti) {   // SamplerRendererTask::Run (samplerrenderer.cpp:60-164)
            int taskNum = (int)tasks[ti];
            Sampler *sampler = mainSampler->GetSubSampler(taskNum, nTasks);
            if (!sampler) { nextRng.push_back(0); continue; }
            RNG rng(taskNum);
            int maxSamples = sampler->MaximumSampleCount();
            Sample *samples = origSample->Duplicate(maxSamples);
            std::vector<Spectrum> LsAll(maxSamples);
            int sampleCount;
            while ((sampleCount = sampler->GetMoreSamples(samples, rng)) > 0) {
                for (int i = 0; i < sampleCount; ++i) {
                    RayDifferential ray;
                    float rayWeight = camera->GenerateRayDifferential(samples[i], &ray);
                    ray.ScaleDifferentials(1.f / sqrtf(sampler->samplesPerPixel));
                    Intersection isect;
                    B.scene->Intersect(ray, &isect);   // clips the ray (samplerrenderer.cpp:236); no surface integrator here: Ls = Lvi
                    Spectrum T(1.f);
                    Spectrum Ls = rayWeight * vis[w]->Li(B.scene, rens[w], ray, &samples[i], rng, &T, arena);
                    if (Ls.HasNaNs() || Ls.y() < -1e-5 || isinf(Ls.y())) Ls = Spectrum(0.f);
                    LsAll[i] = Ls;
                }
                if (sampler->ReportResults(samples, NULL, NULL, NULL, sampleCount))
                    for (int i = 0; i < sampleCount; ++i) film->AddSample(samples[i], LsAll[i]);
                arena.FreeAll();
            }
            nextRng.push_back(rng.RandomUInt());
            delete sampler;
        }
        std::vector<float> pix;
        for (int y = 0; y < yres; ++y)
            for (int x = 0; x < xres; ++x) {
                const ImageFilm::Pixel &px = (*film->pixels)(x, y);
                pix.push_back(px.Lxyz[0]); pix.push_back(px.Lxyz[1]); pix.push_back(px.Lxyz[2]); pix.push_back(px.weightSum);
            }
        const std::string pre = w ? "hip." : "ref.";
        out.putf(pre + "film.pixels", pix);
        out.putu(pre + "next_rng", nextRng);
    }
    // the binding's RenderTasks: the same tasks on the device tile driver, from the reference's camera / film / sampler objects
    {
        const size_t npix = (size_t)xres * yres;
        float *dPixels = NULL, *dRgb = NULL;
        if (hipMalloc((void **)&dPixels, npix * 4 * sizeof(float)) != hipSuccess || hipMalloc((void **)&dRgb, npix * 3 * sizeof(float)) != hipSuccess) return 5;
        hipMemset(dPixels, 0, npix * 4 * sizeof(float));
        int rc = P.hip->RenderTasks(lastCamera, lastFilm, lastSampler, lastOrig, nTasks, dPixels, dRgb, NULL, &tasks);
        if (rc == PVOL_OK) rc = pvol_check_errors(P.hip->context());
        if (rc != PVOL_OK) { fprintf(stderr, "RenderTasks: %s\n", pvol_strerror(rc)); return 6; }
        std::vector<float> pix(npix * 4), rgb(npix * 3);
        hipDeviceSynchronize();
        hipMemcpy(pix.data(), dPixels, pix.size() * sizeof(float), hipMemcpyDeviceToHost);
        hipMemcpy(rgb.data(), dRgb, rgb.size() * sizeof(float), hipMemcpyDeviceToHost);
        out.putf("tiles.film.pixels", pix);
        out.putf("tiles.film.rgb", rgb);
        hipFree(dPixels); hipFree(dRgb);
    }
    return out.save(outPath) ? 0 : 1;
}

int main(int argc, char **argv) {
    SampledSpectrum::Init();
    if (argc >= 6 && !strcmp(argv[1], "li")) return cmdLi(argv[2], argv[3], argv[4], argv[5]);
    if (argc >= 6 && !strcmp(argv[1], "render")) return cmdRender(argv[2], argv[3], argv[4], argv[5]);
    fprintf(stderr, "usage: shim_drive li|render NAME PHOTONS|- CASE OUT\n");
    return 64;
}
