// pvol_dev.h -- device-side data layout shared by the HIP kernels and the C-ABI host code.
//
// Layout in HBM (all fp32 unless noted):
//   photon map, sorted by grid cell (x fastest), SoA:
//     pos4  [n]      float4 {x, y, z, original index as bits}   -- 16 B per distance test
//     alpha [n][32]  30 spectral bins + 2 zero pads = one 128-B row, read as 8 x float4 by an
//                    8-lane group (rows of photons outside a homogeneous extent are zeroed:
//                    HomogeneousVolumeDensity::p() returns 0 there, volumes/homogeneous.h:76-79)
//     wi4   [n]      float4 {wi.x, wi.y, wi.z, 0}               -- only read when g != 0
//   cellStart [ncells+1] u32: first sorted photon of each cell; a row of cells along x is one
//     contiguous photon range, so a lookup touches (2R+1)^2 ranges.
// Spectra live in an "8 lanes x float4" register layout: lane l holds bins 4*(l&7)..4*(l&7)+3;
// the 8 groups of a wavefront hold copies (or, during the flux sum, 8 different photons).
#ifndef PVOL_DEV_H
#define PVOL_DEV_H

#include <stdint.h>
#include "../../include/pvol.h"

#define PVOL_MAX_LIGHTS 8
#define PVOL_MAX_TRIS 64            // triangles in the DevScene itself (scalar loads); more go through the hierarchy
#define PVOL_BVH_MAX_TRIS (1 << 24)  // pvol_bvh.hip
#define PVOL_MAX_SPHERES 8
#define PVOL_MAX_RING 8   // search radius in cells: rings of (dy,dz) rows, 8r rows per ring <= 64 lanes

struct DevLight {
    int32_t kind;
    float pos[3];
    float dir[3];
    float w2l[12];  // rows 0..2 of WorldToLight (vectors only)
    float cosTotalWidth, cosFalloffStart;
    float intensity[32];
};

struct DevSphere {   // shapes/sphere.cpp: what Sphere::Sphere stores
    float o2w[16], w2o[16];
    float radius, zmin, zmax, thetaMin, thetaMax, phiMax;
    int32_t mat, flip;
};

struct DevTri {
    float p1[3], p2[3], p3[3];
};

// PhotonIntegrator (integrators/photonmap.cpp), matte subset: parameters + the caustic photon map (same grid layout as the
// volume map; alpha rows are the photons' raw weights, divided by nCausticPaths at lookup as photonmap.cpp:89 does)
struct DevSurface {
    int32_t enabled, nLookup, maxSpecularDepth, nCausticPaths;
    float maxDistSq;
    uint32_t nPhotons;
    float gridLo[3];
    float cellSize, invCell;
    int32_t gdim[3];
    const uint32_t *cellStart;
    const float4 *pos4, *alpha4, *wi4;
};
struct DevShootScene;

struct DevScene {
    // volume
    int32_t volKind;
    float extLo[3], extHi[3];
    float w2v[16];
    float sigA[32], sigS[32], le[32];
    float g;
    int32_t nx, ny, nz;
    const float *density;
    // lights / triangles
    int32_t nLights;
    DevLight lights[PVOL_MAX_LIGHTS];
    int32_t nTris;
    DevTri tris[PVOL_MAX_TRIS];
    // more than PVOL_MAX_TRIS triangles: nTris is 0 and the scene is the device-built hierarchy (pvol_bvh_dev.h, pvol_bvh.hip)
    const float4 *bvhNodes;   // [nBvhTris - 1][4], 0 = no hierarchy
    const float4 *bvhTris;    // [nBvhTris][3] in Morton order: {p1, original index} {p2, material} {p3, flip_normal}
    int32_t nBvhTris;
    int32_t nSpheres;         // analytic spheres, tested after the triangles (pvol_math.h sphere_hit)
    DevSphere spheres[PVOL_MAX_SPHERES];
    // colour matching
    float cieX[32], cieY[32], cieZ[32];
    // integrator parameters
    float stepSize, maxDist, maxDistSq;
    int32_t nUsed;
    // photon grid
    uint32_t nPhotons;
    float gridLo[3];
    float cellSize, invCell;
    int32_t gdim[3];
    int32_t ringMax;          // ceil(maxDist / cellSize), <= PVOL_MAX_RING
    const uint32_t *cellStart;
    const uint32_t *subStart; // 4x4x4 sub-cells per cell (clumpy maps only, else 0): start of every sub-cell, x fastest
    const float4 *pos4;
    const float4 *alpha4;     // n * 8 float4
    const float4 *wi4;
    // LDS plan (bytes offsets are derived in the kernel from these)
    int32_t candCap;          // candidate list capacity (multiple of 64, >= nUsed + 128)
    int32_t maxSteps;         // upper bound of march steps per ray (lightNum array length)
    float rkEstimate;         // k-th nearest distance^2 expected at the map's mean density (first guess of a cold lookup)
    DevSurface surf;                  // surface integrator, disabled unless pvol_set_surface_integrator enabled it
    const DevShootScene *shootScene;  // materials of the triangles (device copy of the shooter's scene)
};

// One lookup li_group_kernel hands to li_fixup_kernel (pvol_group_dev.h)
struct DeferRec {
    uint32_t ray;
    float px, py, pz;
    float kRem;       // -log2(e) * optical length behind the step
    float stepD;      // step length x density factor
    float guess;      // a k-th distance^2 nearby, 0 = none
    float dens;       // density factor at the point (1 inside a homogeneous extent)
};

// What only the photon shooter reads (core/photonshooter.cpp): surface materials, emission frames,
// the light-power CDF and the shooter's own parameters.
#define PVOL_MAX_MATERIALS 8
struct DevMaterial {
    int32_t kind;
    float ior, vn;
    int32_t nBxdf;        // components in BSDF::Add order (matte.cpp:55-60, glass.cpp:52-57)
    int32_t bxdfType[2];  // BxDFType bits (core/reflection.h:107-121)
    float kd[30], kr[30], kt[30];
};
struct DevShootScene {
    int32_t triMat[PVOL_MAX_TRIS];
    int32_t triFlip[PVOL_MAX_TRIS];
    int32_t nMats;
    DevMaterial mats[PVOL_MAX_MATERIALS];
    float l2w[PVOL_MAX_LIGHTS][12];     // rows 0..2 of LightToWorld (spot emission, spot.cpp:106-114)
    float worldCenter[3], worldRadius;  // BoundingSphere of Scene::WorldBound() (distant.cpp:86-100)
    float lightFunc[PVOL_MAX_LIGHTS], lightCdf[PVOL_MAX_LIGHTS + 1], lightFuncInt;  // Distribution1D (montecarlo.h:54-76)
    float shooterStep;
    int32_t maxPhotonDepth, finalGather;
    uint32_t nCausticWanted, nIndirectWanted, nVolumeWanted;
};

struct DevCounters {
    unsigned long long nRays, nSteps, nTested, nKept, nLookupsLt10, nShadowUnoccluded, nErrors, pad;
    unsigned long long cySearch, cySelect, cyFlux, cyTotal;  // s_memtime cycles summed over waves (stats build only)
    unsigned long long diag[6];   // li_group_kernel (stats build): lookups whose radius guess failed / whose bucket plan was skipped / cycles spent in the exact fallback lookups
};

// surface_kernel (pvol_surface_dev.h)
// ---- specular recursion of the surface integrator (pvol_spec_dev.h)
#define SPEC_MAX_DEPTH 5          // "maxspeculardepth" values up to this are walked (the reference's default); more is refused on the host
#define SPEC_LINK_DONE 0x80000000u // the sample's segments have been folded in (their pool slots may be reused by the next slice)
#define SPEC_LINK_COUNT_BITS 6    // specLink = (first segment << 6) | segment count; a tree has at most 2 + 4 + 8 + 16 = 30 segments

struct SegInfo {                  // one per segment ray, 32 bytes
    uint32_t sample;              // index of the camera sample's primary ray in the batch
    uint32_t depthLobeMat;        // depth (8 bits) | lobe (8: 1 reflection, 2 transmission) | material of the PARENT hit (16)
    float Fs, awz, g;             // f (.) |cos| / pdf = ((Fs K) / awz) g:  Fs = F or 1 - F, awz = |cos| in the BSDF frame, g = AbsDot(wi, n) / pdf
    uint32_t pad[3];
};

struct SpecComposeArgs {
    const DevScene *scene;
    uint32_t *link;           // per primary ray of the range: (first segment << 6) | count, 0 = none; SPEC_LINK_DONE is set once composed
    const SegInfo *info;
    const float *segOut;      // 60 floats per segment: Lv[30] (volume term + T (.) matte surface term), T[30]
    const float *tau;         // per primary ray: optical length of Li()'s last march step
    float *out;               // per primary ray X, Y, Z, T.y: the specular surface term is ADDED to X, Y, Z
    float *surfOut;           // optional: the surface integrator's Li as X, Y, Z (3 floats per ray)
    uint32_t first, nRays;    // the range of primary rays
};

// BxDFType bits (core/reflection.h:107-121)
#ifndef BSDF_ALL
#define BSDF_REFLECTION 1
#define BSDF_TRANSMISSION 2
#define BSDF_DIFFUSE 4
#define BSDF_GLOSSY 8
#define BSDF_SPECULAR 16
#define BSDF_ALL 31
#endif

struct SurfArgs {
    const DevScene *scene;
    const pvol_ray *rays;
    uint32_t nRays;
    float *out;            // per ray X, Y, Z, T.y of the volume term: the surface term is ADDED to X, Y, Z
    const float *tau;      // per ray: optical length of the last march step (T = exp(-sigma_t * tau))
    float *surfOut;        // optional: the surface integrator's Li as X, Y, Z (3 floats per ray)
    DevCounters *counters;
    const uint32_t *link;  // optional: per ray the specular-recursion link (pvol_spec_dev.h); linked samples keep the surfOut the composition wrote
    int32_t spectral;      // 1: `out` holds 60 floats per ray (Lv[30], T[30], the segments of the specular recursion): T (.) Ls is added to Lv
};

// Tile driver (pvol_tile_dev.h, pvol_tile.hip): what a SamplerRendererTask needs besides the scene.
struct TileArgs {
    float r2c[16], c2w[16];        // PerspectiveCamera::RasterToCamera, Camera::CameraToWorld
    float shutterOpen, shutterClose;
    uint32_t spp;                  // LDSampler::nPixelSamples
    uint32_t n1dCount, n2dCount;
    uint32_t n1d[PVOL_MAX_SAMPLE_ARRAYS], n2d[PVOL_MAX_SAMPLE_ARRAYS];
    uint32_t scatterIndex;
    const int4 *windows;           // per stream of the batch: x0, x1, y0, y1 (Sampler::ComputeSubWindow)
    pvol_ray *rays;                // out: camera rays, stream-major, pixel-major, sample-minor
    float *xy;                     // out: imageX, imageY per ray
    // specular recursion of the surface integrator (pvol_spec_dev.h): the spawned rays of camera samples that meet glass
    int32_t specOn;
    pvol_ray *segRays;             // pool of segment rays, laid out in stream order per camera sample
    SegInfo *segInfo;
    uint32_t *specLink;            // per primary ray: (first segment << 6) | count, 0 = none
    uint32_t *segCounter;          // pool allocation counter
    uint32_t segCap;
    unsigned char *segRecords;     // FUSED mode: per segment slot one record of recStride bytes
    uint32_t debugSkip;            // timing experiments only (PVOL_TILE_DEBUG): 1 skip the swaps, 2 skip the draw count, 4 skip advancing the stream
};

#endif
