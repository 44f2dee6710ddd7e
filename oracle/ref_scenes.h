// oracle/ref_scenes.h -- TEST INFRASTRUCTURE.  The BASELINE.json scenes built through the REFERENCE's own public
// Create*() functions exactly as core/api.cpp would from the .pbrt files (shared by ref_capture.cpp and shim_drive.cpp).
// `private`/`protected` are opened for the reference headers only, so that object state can be read back.
#ifndef ORACLE_REF_SCENES_H
#define ORACLE_REF_SCENES_H
#include <algorithm>
#include <cassert>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <list>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <string>
#include <vector>
#include <stdint.h>
#include <pthread.h>

#define private public
#define protected public
#include "stdafx.h"
#include "pbrt.h"
#include "spectrum.h"
#include "rng.h"
#include "montecarlo.h"
#include "geometry.h"
#include "transform.h"
#include "paramset.h"
#include "scene.h"
#include "light.h"
#include "volume.h"
#include "shape.h"
#include "primitive.h"
#include "material.h"
#include "reflection.h"
#include "intersection.h"
#include "renderer.h"
#include "sampler.h"
#include "kdtree.h"
#include "texture.h"
#include "photonshooter.h"
#include "integrators/photonvolume.h"
#include "integrators/photonmap.h"
#include "accelerators/bvh.h"
#include "accelerators/grid.h"
#include "accelerators/kdtreeaccel.h"
#include "textures/constant.h"
#include "lights/distant.h"
#include "lights/point.h"
#include "lights/spot.h"
#include "materials/glass.h"
#include "materials/matte.h"
#include "shapes/trianglemesh.h"
#include "shapes/sphere.h"
#include "volumes/homogeneous.h"
#include "volumes/rainbow.h"
#include "volumes/volumegrid.h"
#include "camera.h"
#include "film.h"
#include "filter.h"
#include "cameras/perspective.h"
#include "film/image.h"
#include "filters/gaussian.h"
#include "samplers/lowdiscrepancy.h"
#undef private
#undef protected

#include "pvol.h"
#include "blob.h"

using blob::Blob;

// ----------------------------------------------------------------------------- scene building
struct BuiltScene {
    Scene *scene;
    VolumeRegion *volume;
    int volumeKind;
    std::vector<Light *> lights;
    std::vector<int> lightKinds;
    std::vector<TriangleMesh *> meshes;
    std::vector<int> meshMaterial;
    std::vector<Sphere *> spheres;
    std::vector<int> sphereMaterial;
    struct Mat { int kind; Spectrum kd, kr, kt; float ior, vn; };
    std::vector<Mat> mats;
    std::vector<Reference<Material> > matRefs;
    std::vector<Reference<Primitive> > prims;
    // integrator parameters from the scene file
    float stepSize, maxDist, shooterStep;
    int nUsed, nVolumePhotons, maxPhotonDepth, nCaustic, nIndirect, finalGather;
    Transform camToWorld;
    float fov;
    int xres, yres, spp;
    std::vector<float> density;
    int nx, ny, nz;
    float meshCenter[3] = {0.f, 0.f, 0.f}, meshRadius = 0.f;   // meshroom: where the hit-record rays are aimed half of the time (0: nowhere)
};

static Transform *keep(const Transform &t) { return new Transform(t); }

static int addMatte(BuiltScene &B, float r, float g, float b) {
    ParamSet geom, mat;
    float rgb[3] = {r, g, b};
    mat.AddRGBSpectrum("Kd", rgb, 3);
    map<string, Reference<Texture<float> > > ft;
    map<string, Reference<Texture<Spectrum> > > st;
    TextureParams mp(geom, mat, ft, st);
    Reference<Material> m = CreateMatteMaterial(Transform(), mp);
    BuiltScene::Mat M;
    M.kind = PVOL_MATERIAL_MATTE;
    M.kd = mat.FindOneSpectrum("Kd", Spectrum(0.5f)).Clamp();
    M.kr = M.kt = Spectrum(0.f);
    M.ior = 1.f; M.vn = 0.f;
    B.mats.push_back(M);
    B.matRefs.push_back(m);
    return (int)B.mats.size() - 1;
}

static int addGlass(BuiltScene &B, float ior, float vn, const float kr[3], const float kt[3]) {
    ParamSet geom, mat;
    mat.AddFloat("index", &ior, 1);
    mat.AddFloat("Vn", &vn, 1);
    mat.AddRGBSpectrum("Kr", kr, 3);
    mat.AddRGBSpectrum("Kt", kt, 3);
    map<string, Reference<Texture<float> > > ft;
    map<string, Reference<Texture<Spectrum> > > st;
    TextureParams mp(geom, mat, ft, st);
    Reference<Material> m = CreateGlassMaterial(Transform(), mp);
    BuiltScene::Mat M;
    M.kind = PVOL_MATERIAL_GLASS;
    M.kd = Spectrum(0.f);
    M.kr = mat.FindOneSpectrum("Kr", Spectrum(1.f)).Clamp();
    M.kt = mat.FindOneSpectrum("Kt", Spectrum(1.f)).Clamp();
    M.ior = ior; M.vn = vn;
    B.mats.push_back(M);
    B.matRefs.push_back(m);
    return (int)B.mats.size() - 1;
}

static void addMesh(BuiltScene &B, const Transform &ctm, const float *P, int nverts, const int *idx, int nidx, int material) {
    ParamSet ps;
    std::vector<Point> pts(nverts);
    for (int i = 0; i < nverts; ++i) pts[i] = Point(P[3 * i], P[3 * i + 1], P[3 * i + 2]);
    ps.AddPoint("P", &pts[0], nverts);
    ps.AddInt("indices", idx, nidx);
    Transform *o2w = keep(ctm), *w2o = keep(Inverse(ctm));
    TriangleMesh *mesh = CreateTriangleMeshShape(o2w, w2o, false, ps, NULL);
    B.meshes.push_back(mesh);
    B.meshMaterial.push_back(material);
    Reference<Shape> shape(mesh);
    B.prims.push_back(new GeometricPrimitive(shape, B.matRefs[material], NULL));
}

static void addSphere(BuiltScene &B, const Transform &ctm, float radius, float zmin, float zmax, float phimax, int material) {
    ParamSet ps;
    ps.AddFloat("radius", &radius, 1);
    ps.AddFloat("zmin", &zmin, 1);
    ps.AddFloat("zmax", &zmax, 1);
    ps.AddFloat("phimax", &phimax, 1);
    Transform *o2w = keep(ctm), *w2o = keep(Inverse(ctm));
    Sphere *sph = CreateSphereShape(o2w, w2o, false, ps);
    B.spheres.push_back(sph);
    B.sphereMaterial.push_back(material);
    Reference<Shape> shape(sph);
    B.prims.push_back(new GeometricPrimitive(shape, B.matRefs[material], NULL));
}

static void addQuad(BuiltScene &B, const Transform &ctm, const float P[12], int material) {
    static const int idx[6] = {0, 1, 2, 2, 3, 0};
    addMesh(B, ctm, P, 4, idx, 6, material);
}

static void finish(BuiltScene &B) {
    ParamSet accelParams;
    Primitive *accel = CreateBVHAccelerator(B.prims, accelParams);  // core/api.cpp default accelerator "bvh"
    B.scene = new Scene(accel, B.lights, B.volume);
}

static VolumeRegion *makeVolume(BuiltScene &B, const Transform &ctm, const char *kind, const float p0[3], const float p1[3],
                                float sa, float ss, int gridN = 0, float g = 0.f) {
    ParamSet ps;
    float rgbA[3] = {sa, sa, sa}, rgbS[3] = {ss, ss, ss};
    if (g != 0.f) ps.AddFloat("g", &g, 1);   // Henyey-Greenstein asymmetry (volumes/homogeneous.cpp:44)
    ps.AddRGBSpectrum("sigma_a", rgbA, 3);
    ps.AddRGBSpectrum("sigma_s", rgbS, 3);
    Point a(p0[0], p0[1], p0[2]), b(p1[0], p1[1], p1[2]);
    ps.AddPoint("p0", &a, 1);
    ps.AddPoint("p1", &b, 1);
    if (!strcmp(kind, "homogeneous")) { B.volumeKind = PVOL_VOLUME_HOMOGENEOUS; return CreateHomogeneousVolumeDensityRegion(ctm, ps); }
    if (!strcmp(kind, "rainbow")) { B.volumeKind = PVOL_VOLUME_RAINBOW; return CreateRainbowVolumeDensityRegion(ctm, ps); }
    // synthetic heterogeneous grid (BASELINE.json config 4)
    B.volumeKind = PVOL_VOLUME_GRID;
    B.nx = B.ny = B.nz = gridN;
    B.density.resize((size_t)gridN * gridN * gridN);
    uint32_t h = 348u;
    for (int z = 0; z < gridN; ++z)
        for (int y = 0; y < gridN; ++y)
            for (int x = 0; x < gridN; ++x) {
                float xh = (x + .5f) / gridN, yh = (y + .5f) / gridN, zh = (z + .5f) / gridN;
                h = h * 1664525u + 1013904223u;
                float noise = ((h >> 8) & 0xffff) / 65536.f - .5f;
                float d = 0.5f + 0.5f * sinf(7.f * xh) * sinf(5.f * yh) * sinf(3.f * zh) + 0.25f * noise;
                B.density[(size_t)z * gridN * gridN + (size_t)y * gridN + x] = d < 0.f ? 0.f : (d > 1.5f ? 1.5f : d);
            }
    ps.AddFloat("density", &B.density[0], (int)B.density.size());
    ps.AddInt("nx", &gridN, 1);
    ps.AddInt("ny", &gridN, 1);
    ps.AddInt("nz", &gridN, 1);
    return CreateGridVolumeRegion(ctm, ps);
}

static void addDistant(BuiltScene &B, const Transform &ctm, const float from[3], const float to[3], float L) {
    ParamSet ps;
    Point f(from[0], from[1], from[2]), t(to[0], to[1], to[2]);
    float rgb[3] = {L, L, L};
    ps.AddPoint("from", &f, 1);
    ps.AddPoint("to", &t, 1);
    ps.AddRGBSpectrum("L", rgb, 3);
    B.lights.push_back(CreateDistantLight(ctm, ps));
    B.lightKinds.push_back(PVOL_LIGHT_DISTANT);
}
static void addSpot(BuiltScene &B, const Transform &ctm, const float from[3], const float to[3], float I, float cone) {
    ParamSet ps;
    Point f(from[0], from[1], from[2]), t(to[0], to[1], to[2]);
    float rgb[3] = {I, I, I};
    ps.AddPoint("from", &f, 1);
    ps.AddPoint("to", &t, 1);
    ps.AddRGBSpectrum("I", rgb, 3);
    ps.AddFloat("coneangle", &cone, 1);
    B.lights.push_back(CreateSpotLight(ctm, ps));
    B.lightKinds.push_back(PVOL_LIGHT_SPOT);
}
static void addPoint(BuiltScene &B, const Transform &ctm, const float from[3], float I) {
    ParamSet ps;
    Point f(from[0], from[1], from[2]);
    float rgb[3] = {I, I, I};
    ps.AddPoint("from", &f, 1);
    ps.AddRGBSpectrum("I", rgb, 3);
    B.lights.push_back(CreatePointLight(ctm, ps));
    B.lightKinds.push_back(PVOL_LIGHT_POINT);
}

// obj/prism.pbrt (scene input data)
static const float kPrismP[18] = {1, -1, -1, 1, -1, 1, -1, -1, 1, -1, -1, -1, 1, 1, 9.999999975e-07f, -1, 1, -0.f};
static const int kPrismIdx[24] = {0, 1, 2, 0, 2, 3, 1, 4, 5, 1, 5, 2, 0, 4, 1, 2, 5, 3, 4, 0, 3, 4, 3, 5};

// projectScene/volumescene_png.pbrt; `volKind` swaps the Volume statement, `gridN` > 0
// makes the synthetic config-4 variant.
static void buildVolumeSceneNoFinish(BuiltScene &B, const char *volKind, int gridN, float g) {
    B.stepSize = .15f; B.nUsed = 50; B.maxDist = 0.5f; B.nVolumePhotons = 5000;
    B.shooterStep = 0.1f; B.maxPhotonDepth = 5; B.nCaustic = 5000; B.nIndirect = 0; B.finalGather = 1;
    B.xres = B.yres = 300; B.spp = 1; B.fov = 70.f;
    Transform camCtm = Rotate(0, Vector(0, 1, 0));      // "Rotate 0 1 0 0" before Camera
    B.camToWorld = Inverse(camCtm);
    Transform ctm = Transform() * Translate(Vector(0, -0.5f, 3.5f));
    float p0[3] = {-10, 0, -5}, p1[3] = {5, 5, 5};
    B.volume = makeVolume(B, ctm, volKind, p0, p1, .05f, .1f, gridN, g);
    float from[3] = {0, 3, 0}, to[3] = {0, 2, 5};
    addDistant(B, ctm, from, to, 150.f);
    int m = addMatte(B, .01f, .01f, .01f);
    float q1[12] = {-5, 0, -5, 5, 0, -5, 5, 0, 5, -5, 0, 5};
    float q2[12] = {-5, 0, 3, 5, 0, 3, 5, 10, 3, -5, 10, 3};
    float q3[12] = {5, 0, 3, 5, 0, -3, 5, 10, -3, 5, 10, 3};
    addQuad(B, ctm, q1, m);
    addQuad(B, ctm, q2, m);
    addQuad(B, ctm, q3, m);
}
static void buildVolumeScene(BuiltScene &B, const char *volKind, int gridN, float g = 0.f) {
    buildVolumeSceneNoFinish(B, volKind, gridN, g);
    finish(B);
}

// projectScene/pinkfloyd.pbrt
static void buildPinkFloyd(BuiltScene &B) {
    B.stepSize = .05f; B.nUsed = 500; B.maxDist = 0.4f; B.nVolumePhotons = 5000000;
    B.shooterStep = 0.1f; B.maxPhotonDepth = 5; B.nCaustic = 1; B.nIndirect = 0; B.finalGather = 0;
    B.xres = B.yres = 512; B.spp = 32; B.fov = 70.f;
    Transform camCtm = Rotate(5, Vector(1, 0, 0));
    B.camToWorld = Inverse(camCtm);
    Transform ctm = Transform() * Translate(Vector(0, -0.5f, 3.5f));
    float p0[3] = {-10, -10, -10}, p1[3] = {5, 5, 5};
    B.volume = makeVolume(B, ctm, "homogeneous", p0, p1, .05f, .1f);
    float sf[3] = {-3, 0.72f, 0}, st[3] = {0, 1.55f, 0};
    addSpot(B, ctm, sf, st, 15000.f, 0.8f);
    float pf[3] = {0.1f, 1.35f, -4};
    addPoint(B, ctm, pf, 4.f);
    float kr[3] = {0, 0, 0}, kt[3] = {1, 1, 1};
    int glass = addGlass(B, 1.3f, 2.75f, kr, kt);
    Transform pctm = ctm * Translate(Vector(0.1f, 1.35f, 0)) * Rotate(90, Vector(0, 1, 0)) * Rotate(0, Vector(1, 0, 0)) *
                     Scale(0.05f, 0.7f, 0.85f);
    addMesh(B, pctm, kPrismP, 6, kPrismIdx, 24, glass);
    int matte = addMatte(B, .001f, .001f, .001f);
    float q[12] = {5, -20, 3, 5, -20, -3, 5, 20, -3, 5, 20, 3};
    addQuad(B, ctm, q, matte);
    finish(B);
}

// BASELINE.json config 5: homogeneous slab, prism, enclosing matte box, one spot light.
static void buildShootBench(BuiltScene &B) {
    B.stepSize = .15f; B.nUsed = 50; B.maxDist = 0.5f; B.nVolumePhotons = 100000000;
    B.shooterStep = 0.1f; B.maxPhotonDepth = 5; B.nCaustic = 0; B.nIndirect = 0; B.finalGather = 0;
    B.xres = B.yres = 256; B.spp = 1; B.fov = 70.f;
    B.camToWorld = Transform();
    Transform ctm = Transform() * Translate(Vector(0, -0.5f, 3.5f));
    float p0[3] = {-5, 0, -5}, p1[3] = {5, 5, 5};
    B.volume = makeVolume(B, ctm, "homogeneous", p0, p1, .05f, .1f);
    float sf[3] = {-3, 2.5f, 0}, st[3] = {0, 2.5f, 0};
    addSpot(B, ctm, sf, st, 15000.f, 20.f);
    float kr[3] = {0, 0, 0}, kt[3] = {1, 1, 1};
    int glass = addGlass(B, 1.3f, 2.75f, kr, kt);
    Transform pctm = ctm * Translate(Vector(0.1f, 2.5f, 0)) * Rotate(90, Vector(0, 1, 0)) * Scale(0.3f, 0.7f, 0.85f);
    addMesh(B, pctm, kPrismP, 6, kPrismIdx, 24, glass);
    int matte = addMatte(B, .5f, .5f, .5f);
    // enclosing box (-6,-1,-6)-(6,6,6): photons that never hit a surface are dropped (photonshooter.cpp:54)
    float bx[24] = {-6, -1, -6, 6, -1, -6, 6, -1, 6, -6, -1, 6, -6, 6, -6, 6, 6, -6, 6, 6, 6, -6, 6, 6};
    int bi[36] = {0, 1, 2, 2, 3, 0, 4, 6, 5, 6, 4, 7, 0, 4, 5, 5, 1, 0, 1, 5, 6, 6, 2, 1, 2, 6, 7, 7, 3, 2, 3, 7, 4, 4, 0, 3};
    addMesh(B, ctm, bx, 8, bi, 36, matte);
    finish(B);
}

// the volumescene room with a tessellated, bumpy matte ball in the medium (nu x nv quads -> 2 nu nv - 2 nu
// triangles): more triangles than a linear scan is meant for, so the reference answers through its BVHAccel proper.
static void buildMeshRoom(BuiltScene &B, int nu, int nv) {
    buildVolumeSceneNoFinish(B, "homogeneous", 0, 0.f);
    Transform ctm = Transform() * Translate(Vector(0, -0.5f, 3.5f));
    int m = addMatte(B, .4f, .3f, .2f);
    std::vector<float> P;
    std::vector<int> idx;
    const float R = 0.9f, cx = 0.5f, cy = 1.8f, cz = 0.5f;
    for (int j = 0; j <= nv; ++j)
        for (int i = 0; i < nu; ++i) {
            const float th = M_PI * j / nv, ph = 2.f * M_PI * i / nu;
            const float r = R * (1.f + 0.15f * sinf(5.f * th) * sinf(4.f * ph));
            P.push_back(cx + r * sinf(th) * cosf(ph)); P.push_back(cy + r * cosf(th)); P.push_back(cz + r * sinf(th) * sinf(ph));
        }
    for (int j = 0; j < nv; ++j)
        for (int i = 0; i < nu; ++i) {
            const int a = j * nu + i, b = j * nu + (i + 1) % nu, c = (j + 1) * nu + (i + 1) % nu, d = (j + 1) * nu + i;
            if (j > 0) { idx.push_back(a); idx.push_back(b); idx.push_back(c); }
            if (j < nv - 1) { idx.push_back(a); idx.push_back(c); idx.push_back(d); }
        }
    addMesh(B, ctm, P.data(), (int)P.size() / 3, idx.data(), (int)idx.size(), m);
    B.meshCenter[0] = cx; B.meshCenter[1] = cy - 0.5f; B.meshCenter[2] = cz + 3.5f; B.meshRadius = R * 1.2f;
    finish(B);
}

// projectScene/scene.pbrt: a glass ball (Shape "sphere", index 1.5, Vn 0: no dispersion) in the medium, spot + point light,
// three matte walls.  `extra` adds a second, partial matte sphere under a rotation and a non-uniform scale, so that zmin / zmax
// / phimax clipping, the second root and a general ObjectToWorld are exercised too.
static void buildSphereScene(BuiltScene &B, bool extra) {
    B.stepSize = .05f; B.nUsed = 300; B.maxDist = 0.5f; B.nVolumePhotons = 1000000;
    B.shooterStep = 0.1f; B.maxPhotonDepth = 5; B.nCaustic = 50000; B.nIndirect = 0; B.finalGather = 1;
    B.xres = B.yres = 300; B.spp = 8; B.fov = 70.f;
    Transform camCtm = Rotate(5, Vector(1, 0, 0));
    B.camToWorld = Inverse(camCtm);
    Transform ctm = Transform() * Translate(Vector(-1, -1, 3.5f));
    float p0[3] = {-10, 0, -5}, p1[3] = {5, 5, 5};
    B.volume = makeVolume(B, ctm, "homogeneous", p0, p1, .05f, .1f);
    float sf[3] = {-3, 5, 0}, st[3] = {0, 2, 0};
    addSpot(B, ctm, sf, st, 2500.f, 6.f);
    float pf[3] = {0, 2, -4};
    addPoint(B, ctm, pf, 8.f);
    float kr[3] = {0, 0, 0}, kt[3] = {1, 1, 1};
    int glass = addGlass(B, 1.5f, 0.f, kr, kt);
    addSphere(B, ctm * Translate(Vector(0, 2, 0)), .6f, -.6f, .6f, 360.f, glass);
    int matte = addMatte(B, .6f, .6f, .9f);
    float q1[12] = {-5, 0, -5, 5, 0, -5, 5, 0, 5, -5, 0, 5};
    float q2[12] = {-5, 0, 3, 5, 0, 3, 5, 10, 3, -5, 10, 3};
    float q3[12] = {5, 0, 3, 5, 0, -3, 5, 10, -3, 5, 10, 3};
    addQuad(B, ctm, q1, matte);
    addQuad(B, ctm, q2, matte);
    addQuad(B, ctm, q3, matte);
    if (extra) {
        int m2 = addMatte(B, .5f, .4f, .1f);
        Transform t2 = ctm * Translate(Vector(2, 1, 1)) * Rotate(30, Vector(1, 1, 0)) * Scale(1.f, 0.7f, 1.2f);
        addSphere(B, t2, .5f, -.3f, .4f, 270.f, m2);
        B.meshCenter[0] = 0.f; B.meshCenter[1] = 0.5f; B.meshCenter[2] = 4.f; B.meshRadius = 2.f;
    }
    finish(B);
}

static bool buildByName(BuiltScene &B, const std::string &name) {
    if (name == "volumescene_h") buildVolumeScene(B, "homogeneous", 0);
    else if (name == "volumescene_hg") buildVolumeScene(B, "homogeneous", 0, 0.6f);   // anisotropic phase function (row a16)
    else if (name == "volumescene_rainbow") buildVolumeScene(B, "rainbow", 0);
    else if (name == "volumescene_grid16") buildVolumeScene(B, "grid", 16);
    else if (name == "volumescene_grid128") buildVolumeScene(B, "grid", 128);
    else if (name == "pinkfloyd") buildPinkFloyd(B);
    else if (name == "shootbench") buildShootBench(B);
    else if (name == "meshroom") buildMeshRoom(B, 32, 16);
    else if (name == "spherescene") buildSphereScene(B, false);
    else if (name == "sphereroom") buildSphereScene(B, true);
    else if (name == "meshroom_big") buildMeshRoom(B, 256, 128);
    else return false;
    return true;
}

#endif  // ORACLE_REF_SCENES_H
