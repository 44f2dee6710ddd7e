// pvol_api.hip -- the C ABI of include/pvol.h: context, scene flattening to the device layout,
// photon-map upload + grid build, kernel launches.  There is NO CPU fallback: every entry point
// that needs the GPU returns PVOL_E_NO_DEVICE when HIP is unusable.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <mutex>
#include <new>
#include <vector>

#include "pvol_dev.h"

#include "pvol_host.h"

static bool ok(hipError_t e) { return e == hipSuccess; }

extern "C" {

int pvol_abi_version(void) { return PVOL_ABI_VERSION; }

const char *pvol_strerror(int s) {
    switch (s) {
    case PVOL_OK: return "ok";
    case PVOL_E_INVALID: return "invalid argument";
    case PVOL_E_NO_DEVICE: return "no usable HIP device (this library has no CPU path)";
    case PVOL_E_NO_SCENE: return "no scene set";
    case PVOL_E_NO_MEMORY: return "out of memory";
    case PVOL_E_UNSUPPORTED: return "unsupported volume/light/material kind or size";
    case PVOL_E_LIMIT: return "a ray needs more march steps than the kernel's LDS plan";
    case PVOL_E_SHOOT_FAILED: return "unable to store enough photons";
    default: return "unknown status";
    }
}

int pvol_device_count(void) {
    int n = 0;
    if (!ok(hipGetDeviceCount(&n))) return 0;
    return n;
}

void pvol_default_params(pvol_params *p) {
    memset(p, 0, sizeof(*p));
    p->step_size = 1.f;          // photonvolume.cpp:225
    p->n_used = 250;             // photonvolume.cpp:226
    p->max_dist = 0.1f;          // photonvolume.cpp:227
    p->n_volume_photons = 0;     // photonshooter.cpp:532
    p->shooter_step_size = 0.1f; // photonshooter.cpp:533
    p->max_photon_depth = 5;     // photonshooter.cpp:539
    p->n_caustic_photons = 20000;
    p->n_indirect_photons = 10000;
    p->final_gather = 1;
    p->device = 0;
    p->grid_cell_scale = 0.f;
    p->keep_surface_photons = 0;
}

int pvol_create(const pvol_params *params, pvol_ctx **out) {
    if (!params || !out) return PVOL_E_INVALID;
    *out = 0;
    if (params->n_used < 1 || !(params->step_size > 0.f) || !(params->max_dist > 0.f)) return PVOL_E_INVALID;
    if (params->n_used > 576) return PVOL_E_UNSUPPORTED;   // select_k holds the candidate list in 12 registers per lane
    int n = 0;
    if (!ok(hipGetDeviceCount(&n)) || n <= 0 || params->device < 0 || params->device >= n) return PVOL_E_NO_DEVICE;
    if (!ok(hipSetDevice(params->device))) return PVOL_E_NO_DEVICE;
    pvol_ctx *c = new (std::nothrow) pvol_ctx();
    if (!c) return PVOL_E_NO_MEMORY;
    c->params = *params;
    c->haveScene = false;
    memset(&c->hs, 0, sizeof(c->hs));
    c->ds = 0; c->dDensity = 0;
    c->nPhotons = 0;
    c->dRawP = c->dRawWi = c->dRawAlpha = 0;
    c->dPos4 = c->dAlpha4 = c->dWi4 = 0;
    c->dCellStart = 0;
    c->dCounters = 0;
    c->dWords = 0;
    c->nCU = 256;
    { const char *fs = getenv("PVOL_FORCE_SEQ"); c->forceSeq = fs && fs[0] == '1'; }
    { const char *ng = getenv("PVOL_NO_GROUP"); c->noGroup = ng && ng[0] == '1'; }
    { const char *nl = getenv("PVOL_NO_LITE"); c->noLite = nl && nl[0] == '1'; }
    { const char *gw = getenv("PVOL_GROUP_WAVES"); c->groupWavesPerCU = gw ? std::max(1, atoi(gw)) : 12; }
    { const char *tw = getenv("PVOL_TILE_WAVES"); c->tileWaves = tw ? std::max(0, atoi(tw)) : 0; }
    // li_fixup_kernel / li_fixup_group_kernel waves per CU (C3, 8 spp frame: 16.3 s at 8, 13.9 s at 16)
    { const char *fw = getenv("PVOL_FIX_WAVES"); c->fixWavesPerCU = fw ? std::max(1, atoi(fw)) : 16; }
    { hipDeviceProp_t prop; if (ok(hipGetDeviceProperties(&prop, params->device))) c->nCU = prop.multiProcessorCount; }
    c->statsOn = false;
    c->timeMs = 0; c->launches = 0;
    c->dsh = 0;
    memset(&c->hsh, 0, sizeof(c->hsh));
    memset(c->shootStats, 0, sizeof(c->shootStats));
    if (!ok(hipMalloc(&c->ds, sizeof(DevScene))) || !ok(hipMalloc(&c->dCounters, sizeof(DevCounters))) || !ok(hipMalloc(&c->dWords, 16)) ||
        !ok(hipMalloc(&c->dsh, sizeof(DevShootScene))) || !ok(hipMemset(c->dCounters, 0, sizeof(DevCounters)))) {
        if (c->ds) hipFree(c->ds);
        if (c->dsh) hipFree(c->dsh);
        if (c->dCounters) hipFree(c->dCounters);
        if (c->dWords) hipFree(c->dWords);
        delete c;
        return PVOL_E_NO_DEVICE;
    }
    *out = c;
    return PVOL_OK;
}

void pvol_free_photons(pvol_ctx *c) {
    if (c->dRawP) hipFree(c->dRawP);
    if (c->dRawWi) hipFree(c->dRawWi);
    if (c->dRawAlpha) hipFree(c->dRawAlpha);
    if (c->dPos4) hipFree(c->dPos4);
    if (c->dAlpha4) hipFree(c->dAlpha4);
    if (c->dWi4) hipFree(c->dWi4);
    if (c->dCellStart) hipFree(c->dCellStart);
    if (c->dSubStart) hipFree(c->dSubStart);
    c->dSubStart = 0;
    c->dRawP = c->dRawWi = c->dRawAlpha = 0;
    c->dPos4 = c->dAlpha4 = c->dWi4 = 0;
    c->dCellStart = 0;
    c->nPhotons = 0;
}

void pvol_destroy(pvol_ctx *c) {
    if (!c) return;
    hipSetDevice(c->params.device);
    hipDeviceSynchronize();
    pvol_free_photons(c);
    pvol_free_surface_stores(c);
    pvol_free_caustic_map(c);
    if (c->dTau) hipFree(c->dTau);
    if (c->dSegRays) hipFree(c->dSegRays);
    if (c->dSegInfo) hipFree(c->dSegInfo);
    if (c->dSegOut) hipFree(c->dSegOut);
    if (c->dSegRecords) hipFree(c->dSegRecords);
    if (c->dSegCounter) hipFree(c->dSegCounter);
    if (c->dSegStream) hipFree(c->dSegStream);
    if (c->dSpecLink) hipFree(c->dSpecLink);
    for (auto &p : c->pending) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    for (auto &p : c->pool) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    for (auto &p : c->phaseMarks) hipEventDestroy(p.second);
    for (auto &e : c->phasePool) hipEventDestroy(e);
    if (c->dDensity) hipFree(c->dDensity);
    if (c->dBvhNodes) hipFree(c->dBvhNodes);
    if (c->dBvhTris) hipFree(c->dBvhTris);
    if (c->dRecords) hipFree(c->dRecords);
    if (c->dState) hipFree(c->dState);
    if (c->dDefer) hipFree(c->dDefer);
    for (int i = 0; i < 6; ++i) if (c->dTile[i]) hipFree(c->dTile[i]);
    if (c->ds) hipFree(c->ds);
    if (c->dsh) hipFree(c->dsh);
    if (c->dCounters) hipFree(c->dCounters);
    if (c->dWords) hipFree(c->dWords);
    delete c;
}

// ---- host evaluation of the light-power CDF (ComputeLightSamplingCDF core/integrator.cpp:261-268,
// Distribution1D montecarlo.h:56-76), fp32 in the reference's order (this file is built with
// -ffp-contract=off)
static float host_spec_y(const pvol_scene *s, const float *c30) {
    float yy = 0.f;
    for (int i = 0; i < 30; ++i) yy += s->cie_y.c[i] * c30[i];
    return yy * float(700 - 400) / float(106.856895f * 30);
}
static void world_sphere(const pvol_scene *s, float c[3], float *rad) {   // BBox::BoundingSphere, core/geometry.cpp:60-63
    bool inside = true;
    for (int a = 0; a < 3; ++a) {
        c[a] = .5f * s->world_min[a] + .5f * s->world_max[a];
        inside = inside && c[a] >= s->world_min[a] && c[a] <= s->world_max[a];
    }
    float dx = c[0] - s->world_max[0], dy = c[1] - s->world_max[1], dz = c[2] - s->world_max[2];
    *rad = inside ? sqrtf(dx * dx + dy * dy + dz * dz) : 0.f;
}
static const float kPiF = 3.14159265358979323846f;
static float light_power_y(const pvol_scene *s, const pvol_light &l, float worldRadius) {
    float p[30];
    for (int i = 0; i < 30; ++i) {
        float I = l.intensity.c[i];
        if (l.kind == PVOL_LIGHT_SPOT) p[i] = I * 2.f * kPiF * (1.f - .5f * (l.cos_falloff_start + l.cos_total_width));   // spot.cpp:72-75
        else if (l.kind == PVOL_LIGHT_POINT) p[i] = I * (4.f * kPiF);                                                      // point.cpp:60-62
        else p[i] = I * kPiF * worldRadius * worldRadius;                                                                  // distant.cpp:58-63
    }
    return host_spec_y(s, p);
}

static int fill_shoot_scene(const pvol_ctx *c, const pvol_scene *s, DevShootScene &H) {
    memset(&H, 0, sizeof(H));
    if (s->n_materials > PVOL_MAX_MATERIALS) return PVOL_E_UNSUPPORTED;
    if (s->n_materials && !s->materials) return PVOL_E_INVALID;
    H.nMats = (int)s->n_materials;
    for (uint32_t i = 0; i < s->n_materials; ++i) {
        const pvol_material &m = s->materials[i];
        if (m.kind != PVOL_MATERIAL_MATTE && m.kind != PVOL_MATERIAL_GLASS) return PVOL_E_UNSUPPORTED;
        DevMaterial &d = H.mats[i];
        d.kind = m.kind; d.ior = m.ior; d.vn = m.vn; d.nBxdf = 0;
        bool kdBlack = true, krBlack = true, ktBlack = true;
        for (int b = 0; b < 30; ++b) {
            d.kd[b] = m.kd.c[b]; d.kr[b] = m.kr.c[b]; d.kt[b] = m.kt.c[b];
            kdBlack = kdBlack && m.kd.c[b] == 0.f; krBlack = krBlack && m.kr.c[b] == 0.f; ktBlack = ktBlack && m.kt.c[b] == 0.f;
        }
        if (m.kind == PVOL_MATERIAL_MATTE) { if (!kdBlack) d.bxdfType[d.nBxdf++] = 1 | 4; }          // Lambertian
        else { if (!krBlack) d.bxdfType[d.nBxdf++] = 1 | 16; if (!ktBlack) d.bxdfType[d.nBxdf++] = 2 | 16; }
    }
    for (uint32_t i = 0; i < s->n_triangles; ++i) {
        int mi = s->triangles[i].material;
        if (mi < 0 || (uint32_t)mi >= std::max(1u, s->n_materials)) return PVOL_E_INVALID;
        if (s->n_triangles <= PVOL_MAX_TRIS) {   // a larger scene keeps both in its hierarchy's leaves (pvol_bvh.hip)
            H.triMat[i] = mi;
            H.triFlip[i] = s->triangles[i].flip_normal;
        }
    }
    world_sphere(s, H.worldCenter, &H.worldRadius);
    int n = (int)s->n_lights;
    for (int i = 0; i < n; ++i) {
        memcpy(H.l2w[i], s->lights[i].light_to_world, sizeof(float) * 12);
        H.lightFunc[i] = light_power_y(s, s->lights[i], H.worldRadius);
    }
    if (n > 0) {
        H.lightCdf[0] = 0.f;
        for (int i = 1; i < n + 1; ++i) H.lightCdf[i] = H.lightCdf[i - 1] + H.lightFunc[i - 1] / n;
        H.lightFuncInt = H.lightCdf[n];
        if (H.lightFuncInt == 0.f) { for (int i = 1; i < n + 1; ++i) H.lightCdf[i] = float(i) / float(n); }
        else { for (int i = 1; i < n + 1; ++i) H.lightCdf[i] /= H.lightFuncInt; }
    }
    H.shooterStep = c->params.shooter_step_size;
    H.maxPhotonDepth = c->params.max_photon_depth;
    H.finalGather = c->params.final_gather;
    H.nCausticWanted = c->params.n_caustic_photons;
    H.nIndirectWanted = c->params.n_indirect_photons;
    H.nVolumeWanted = c->params.n_volume_photons;
    return PVOL_OK;
}

static void pad32(float *dst, const pvol_spectrum &s) {
    for (int i = 0; i < 30; ++i) dst[i] = s.c[i];
    dst[30] = dst[31] = 0.f;
}

int pvol_get_accel_info(pvol_ctx *c, double *out2) {
    if (!c || !out2) return PVOL_E_INVALID;
    out2[0] = (double)c->hs.nBvhTris; out2[1] = c->bvhBuildMs;
    return PVOL_OK;
}

int pvol_push_scene(pvol_ctx *c) {
    return ok(hipMemcpy(c->ds, &c->hs, sizeof(DevScene), hipMemcpyHostToDevice)) ? PVOL_OK : PVOL_E_NO_DEVICE;
}

int pvol_set_scene(pvol_ctx *c, const pvol_scene *s) {
    if (!c || !s) return PVOL_E_INVALID;
    std::lock_guard<std::recursive_mutex> api(c->apiMu);
    if (!ok(hipSetDevice(c->params.device))) return PVOL_E_NO_DEVICE;
    // Everything is validated and built into temporaries first; the context changes only when nothing can fail any more
    // (a rejected scene leaves the previous one, including its density grid, in place).
    const pvol_volume &v = s->volume;
    if (v.kind != PVOL_VOLUME_NONE && v.kind != PVOL_VOLUME_HOMOGENEOUS && v.kind != PVOL_VOLUME_GRID && v.kind != PVOL_VOLUME_RAINBOW)
        return PVOL_E_UNSUPPORTED;
    if (s->n_lights > PVOL_MAX_LIGHTS || s->n_triangles > PVOL_BVH_MAX_TRIS) return PVOL_E_UNSUPPORTED;
    if ((s->n_lights && !s->lights) || (s->n_triangles && !s->triangles)) return PVOL_E_INVALID;
    if (s->n_spheres > PVOL_MAX_SPHERES) return PVOL_E_UNSUPPORTED;
    if (s->n_spheres && !s->spheres) return PVOL_E_INVALID;
    for (uint32_t i = 0; i < s->n_spheres; ++i) {
        const pvol_sphere &sp = s->spheres[i];
        if (!(sp.radius > 0.f) || sp.material < 0 || (uint32_t)sp.material >= std::max(1u, s->n_materials)) return PVOL_E_INVALID;
    }
    if (v.kind == PVOL_VOLUME_GRID && (!v.density || v.nx < 1 || v.ny < 1 || v.nz < 1)) return PVOL_E_INVALID;
    for (uint32_t i = 0; i < s->n_lights; ++i) {
        const int k = s->lights[i].kind;
        if (k != PVOL_LIGHT_POINT && k != PVOL_LIGHT_SPOT && k != PVOL_LIGHT_DISTANT) return PVOL_E_UNSUPPORTED;
    }
    DevScene h = c->hs;   // keeps the photon-map fields, everything else is replaced
    memset(&h.surf, 0, sizeof(h.surf));   // the surface integrator belongs to the scene it was enabled on
    h.shootScene = c->dsh;
    h.volKind = v.kind;
    for (int i = 0; i < 3; ++i) { h.extLo[i] = v.extent_min[i]; h.extHi[i] = v.extent_max[i]; }
    memcpy(h.w2v, v.world_to_volume, sizeof(h.w2v));
    pad32(h.sigA, v.sigma_a); pad32(h.sigS, v.sigma_s); pad32(h.le, v.le);
    h.g = v.g;
    h.nx = v.nx; h.ny = v.ny; h.nz = v.nz;
    h.density = 0;
    h.nLights = (int)s->n_lights;
    for (uint32_t i = 0; i < s->n_lights; ++i) {
        const pvol_light &l = s->lights[i];
        DevLight &d = h.lights[i];
        d.kind = l.kind;
        for (int k = 0; k < 3; ++k) { d.pos[k] = l.pos[k]; d.dir[k] = l.dir[k]; }
        memcpy(d.w2l, l.world_to_light, sizeof(float) * 12);
        d.cosTotalWidth = l.cos_total_width;
        d.cosFalloffStart = l.cos_falloff_start;
        pad32(d.intensity, l.intensity);
    }
    h.nSpheres = (int)s->n_spheres;
    memset(h.spheres, 0, sizeof(h.spheres));
    for (uint32_t i = 0; i < s->n_spheres; ++i) {
        const pvol_sphere &sp = s->spheres[i];
        DevSphere &d = h.spheres[i];
        memcpy(d.o2w, sp.object_to_world, sizeof(d.o2w));
        memcpy(d.w2o, sp.world_to_object, sizeof(d.w2o));
        d.radius = sp.radius; d.zmin = sp.z_min; d.zmax = sp.z_max; d.thetaMin = sp.theta_min; d.thetaMax = sp.theta_max; d.phiMax = sp.phi_max;
        d.mat = sp.material; d.flip = sp.flip_normal;
    }
    const bool big = s->n_triangles > PVOL_MAX_TRIS;
    h.nTris = big ? 0 : (int)s->n_triangles;
    h.bvhNodes = 0; h.bvhTris = 0; h.nBvhTris = 0;
    for (uint32_t i = 0; i < s->n_triangles && !big; ++i) {
        const pvol_triangle &t = s->triangles[i];
        for (int k = 0; k < 3; ++k) { h.tris[i].p1[k] = t.p[0][k]; h.tris[i].p2[k] = t.p[1][k]; h.tris[i].p3[k] = t.p[2][k]; }
    }
    pad32(h.cieX, s->cie_x); pad32(h.cieY, s->cie_y); pad32(h.cieZ, s->cie_z);
    h.stepSize = c->params.step_size;
    h.maxDist = c->params.max_dist;
    h.maxDistSq = c->params.max_dist * c->params.max_dist;  // photonvolume.h:18
    h.nUsed = c->params.n_used;
    h.candCap = ((c->params.n_used + 63) / 64) * 64 + 192;
    // march-step bound: diagonal of the volume's world bound / stepSize (rays are clipped to the extent)
    h.maxSteps = 0;
    // (every scene with a medium: a one-light homogeneous scene reaches the record plan too -- pvol_li with the caller's live
    // RNG state takes the RESOLVE + REPLAY path -- and with maxSteps 0 every such ray was reported as PVOL_E_LIMIT)
    if (v.kind != PVOL_VOLUME_NONE) {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = 0; k < 8; ++k) {
            float x = (k & 1) ? v.extent_max[0] : v.extent_min[0], y = (k & 2) ? v.extent_max[1] : v.extent_min[1],
                  z = (k & 4) ? v.extent_max[2] : v.extent_min[2];
            const float *m = v.volume_to_world;
            float w[3] = {m[0] * x + m[1] * y + m[2] * z + m[3], m[4] * x + m[5] * y + m[6] * z + m[7], m[8] * x + m[9] * y + m[10] * z + m[11]};
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], w[a]); hi[a] = std::max(hi[a], w[a]); }
        }
        double diag = sqrt((double)(hi[0] - lo[0]) * (hi[0] - lo[0]) + (double)(hi[1] - lo[1]) * (hi[1] - lo[1]) + (double)(hi[2] - lo[2]) * (hi[2] - lo[2]));
        double steps = diag / c->params.step_size * 1.05 + 4;
        if (steps > 12000) return PVOL_E_LIMIT;
        h.maxSteps = ((int)steps + 63) & ~63;
    }
    DevShootScene hsh;
    {
        int rc = fill_shoot_scene(c, s, hsh);
        if (rc != PVOL_OK) return rc;
    }
    // more triangles than the embedded array: LBVH on the device
    float4 *newNodes = 0, *newTris = 0;
    double bvhMs = 0.0;
    std::vector<int32_t> triMat(s->n_triangles);
    for (uint32_t i = 0; i < s->n_triangles; ++i) triMat[i] = s->triangles[i].material;
    for (uint32_t i = 0; i < s->n_spheres; ++i) triMat.push_back(s->spheres[i].material);   // the surface integrator's matte check covers them
    if (big) {
        const uint32_t n = s->n_triangles;
        std::vector<float> tv((size_t)n * 9);
        std::vector<int32_t> fl(n);
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = 0; i < n; ++i) {
            const pvol_triangle &t = s->triangles[i];
            for (int v3i = 0; v3i < 3; ++v3i)
                for (int k = 0; k < 3; ++k) {
                    const float x = t.p[v3i][k];
                    if (!(x == x) || fabsf(x) == INFINITY) return PVOL_E_INVALID;
                    tv[(size_t)i * 9 + 3 * v3i + k] = x;
                    lo[k] = std::min(lo[k], x); hi[k] = std::max(hi[k], x);
                }
            fl[i] = t.flip_normal;
        }
        const double diag = sqrt((double)(hi[0] - lo[0]) * (hi[0] - lo[0]) + (double)(hi[1] - lo[1]) * (hi[1] - lo[1]) +
                                 (double)(hi[2] - lo[2]) * (hi[2] - lo[2]));
        float *dTri = 0;
        int32_t *dMat = 0, *dFlip = 0;
        bool good = ok(hipMalloc(&dTri, tv.size() * 4)) && ok(hipMalloc(&dMat, (size_t)n * 4)) && ok(hipMalloc(&dFlip, (size_t)n * 4)) &&
                    ok(hipMalloc(&newTris, (size_t)n * 3 * sizeof(float4))) && ok(hipMalloc(&newNodes, (size_t)(n - 1) * 4 * sizeof(float4))) &&
                    ok(hipMemcpy(dTri, tv.data(), tv.size() * 4, hipMemcpyHostToDevice)) &&
                    ok(hipMemcpy(dMat, triMat.data(), (size_t)n * 4, hipMemcpyHostToDevice)) &&
                    ok(hipMemcpy(dFlip, fl.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        if (good) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            good = ok(pvol_build_bvh(dTri, dMat, dFlip, n, (float)(1e-5 * diag), newTris, newNodes, 0));
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            bvhMs = ms;
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
        if (dTri) hipFree(dTri);
        if (dMat) hipFree(dMat);
        if (dFlip) hipFree(dFlip);
        if (!good) { if (newTris) hipFree(newTris); if (newNodes) hipFree(newNodes); return PVOL_E_NO_MEMORY; }
        h.bvhNodes = newNodes; h.bvhTris = newTris; h.nBvhTris = (int)n;
    }
    float *newDensity = 0;
    float maxDensity = 1.f;
    if (v.kind == PVOL_VOLUME_GRID) {
        size_t nb = sizeof(float) * (size_t)v.nx * v.ny * v.nz;
        auto dropBvh = [&]() { if (newTris) hipFree(newTris); if (newNodes) hipFree(newNodes); };
        if (!ok(hipMalloc(&newDensity, nb))) { dropBvh(); return PVOL_E_NO_MEMORY; }
        if (!ok(hipMemcpy(newDensity, v.density, nb, hipMemcpyHostToDevice))) { hipFree(newDensity); dropBvh(); return PVOL_E_NO_DEVICE; }
        float md = 0.f;
        for (size_t i = 0; i < nb / sizeof(float); ++i) md = std::max(md, v.density[i]);
        maxDensity = md;
        h.density = newDensity;
    }
    // commit: kernels of earlier batches may still read the old scene and grid
    if (!ok(hipDeviceSynchronize()) || !ok(hipMemcpy(c->dsh, &hsh, sizeof(hsh), hipMemcpyHostToDevice)) ||
        !ok(hipMemcpy(c->ds, &h, sizeof(DevScene), hipMemcpyHostToDevice))) {
        if (newDensity) hipFree(newDensity);
        if (newTris) hipFree(newTris);
        if (newNodes) hipFree(newNodes);
        pvol_push_scene(c);   // best effort: put the device copy of the previous scene back
        hipMemcpy(c->dsh, &c->hsh, sizeof(c->hsh), hipMemcpyHostToDevice);
        return PVOL_E_NO_DEVICE;
    }
    if (c->dDensity) hipFree(c->dDensity);
    c->dDensity = newDensity;
    if (c->dBvhNodes) hipFree(c->dBvhNodes);
    if (c->dBvhTris) hipFree(c->dBvhTris);
    c->dBvhNodes = newNodes; c->dBvhTris = newTris;
    c->bvhBuildMs = bvhMs;
    c->triMatHost.swap(triMat);
    c->maxDensity = maxDensity;
    pvol_free_caustic_map(c);   // the surface integrator belongs to the scene it was enabled on (h.surf is zero)
    c->hs = h;
    c->hsh = hsh;
    c->haveScene = true;
    return PVOL_OK;
}

static void choose_grid(pvol_ctx *c, const float *p, uint32_t n) {
    DevScene &h = c->hs;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[3 * i + a]); hi[a] = std::max(hi[a], p[3 * i + a]); }
    double ext[3], vol = 1;
    const float maxDist = c->params.max_dist;
    for (int a = 0; a < 3; ++a) { ext[a] = std::max((double)hi[a] - lo[a], 1e-3 * maxDist); vol *= ext[a]; }
    // aim at ~4 photons per cell, never more than PVOL_MAX_RING rings per lookup, at most 2^24 cells
    double cell = cbrt(vol * 1.4 / std::max(1u, n));   // ~1.4 photons per cell measured best on MI355X (profiles/)
    if (c->params.grid_cell_scale > 0.f) cell *= c->params.grid_cell_scale;
    cell = std::max(cell, (double)maxDist / PVOL_MAX_RING * 1.0001);
    for (;;) {
        double cells = 1;
        for (int a = 0; a < 3; ++a) cells *= floor(ext[a] / cell) + 1;
        if (cells <= 16777216.0) break;
        cell *= 1.26;
    }
    h.cellSize = (float)cell;
    h.invCell = 1.f / h.cellSize;
    for (int a = 0; a < 3; ++a) {
        h.gridLo[a] = lo[a];
        h.gdim[a] = (int)floor(ext[a] / cell) + 1;
    }
    h.ringMax = (int)ceil(maxDist / h.cellSize);
    if (h.ringMax > PVOL_MAX_RING) h.ringMax = PVOL_MAX_RING;
    if (h.ringMax < 1) h.ringMax = 1;
    // radius^2 of the ball that holds nUsed photons at the map's mean density: where a lookup with nothing better starts
    h.rkEstimate = (float)pow((double)c->params.n_used * vol / ((double)std::max(1u, n) * 4.18879020478639), 2.0 / 3.0);
}

int pvol_finish_map(pvol_ctx *c, uint32_t n, const float *hostPositions) {
    DevScene &h = c->hs;
    choose_grid(c, hostPositions, n);
    size_t ncells = (size_t)h.gdim[0] * h.gdim[1] * h.gdim[2];
    bool good = ok(hipMalloc(&c->dPos4, sizeof(float4) * (size_t)n)) && ok(hipMalloc(&c->dAlpha4, sizeof(float4) * 8 * (size_t)n)) &&
                ok(hipMalloc(&c->dWi4, sizeof(float4) * (size_t)n)) && ok(hipMalloc(&c->dCellStart, sizeof(uint32_t) * (ncells + 1)));
    if (!good) { pvol_free_photons(c); pvol_push_scene(c); return PVOL_E_NO_MEMORY; }
    GridBuildArgs g;
    g.p = c->dRawP; g.wi = c->dRawWi; g.alpha = c->dRawAlpha; g.n = n;
    for (int a = 0; a < 3; ++a) { g.lo[a] = h.gridLo[a]; g.gdim[a] = h.gdim[a]; g.extLo[a] = h.extLo[a]; g.extHi[a] = h.extHi[a]; }
    g.inv = h.invCell;
    g.sub = 1;
    g.volKind = h.volKind;
    memcpy(g.w2v, h.w2v, sizeof(g.w2v));
    good = ok(pvol_build_grid(&g, c->dPos4, c->dAlpha4, c->dWi4, c->dCellStart, 0, 0));
    if (!good) { pvol_free_photons(c); pvol_push_scene(c); return PVOL_E_NO_DEVICE; }
    // clumpy map (an average photon shares its cell with more than 64 others -- pinkfloyd's beams: 4 700): sort again with
    // the 4 x 4 x 4 second level.  PVOL_SUBGRID=0/1 forces it off/on.
    h.subStart = 0;
    {
        double sq = 0.0;
        const char *ev = getenv("PVOL_SUBGRID");
        bool want = false;
        if (ev) want = atoi(ev) != 0;
        else if (ok(pvol_grid_occupancy(c->dCellStart, (uint32_t)ncells, &sq, 0))) want = sq / (double)n > 64.0;
        if (want && ncells * 64 < 0xfffffff0ull) {
            if (ok(hipMalloc(&c->dSubStart, sizeof(uint32_t) * (ncells * 64 + 1)))) {
                g.sub = 4;
                if (!ok(pvol_build_grid(&g, c->dPos4, c->dAlpha4, c->dWi4, c->dCellStart, c->dSubStart, 0))) { pvol_free_photons(c); pvol_push_scene(c); return PVOL_E_NO_DEVICE; }
                h.subStart = c->dSubStart;
            } else {
                c->dSubStart = 0;   // no room for the table: the coarse level alone is complete
                (void)hipGetLastError();
            }
        }
    }
    c->nPhotons = n;
    h.nPhotons = n; h.cellStart = c->dCellStart; h.pos4 = c->dPos4; h.alpha4 = c->dAlpha4; h.wi4 = c->dWi4;
    return pvol_push_scene(c);
}

int pvol_upload_photons(pvol_ctx *c, const float *p, const float *wi, const float *alpha, uint32_t n) {
    if (!c) return PVOL_E_INVALID;
    if (!c->haveScene) return PVOL_E_NO_SCENE;
    if (n && (!p || !wi || !alpha)) return PVOL_E_INVALID;
    std::lock_guard<std::recursive_mutex> api(c->apiMu);
    if (!ok(hipSetDevice(c->params.device))) return PVOL_E_NO_DEVICE;
    hipDeviceSynchronize();
    pvol_free_photons(c);
    DevScene &h = c->hs;
    h.nPhotons = 0; h.cellStart = 0; h.subStart = 0; h.pos4 = 0; h.alpha4 = 0; h.wi4 = 0;
    if (n == 0) return pvol_push_scene(c);
    bool good = ok(hipMalloc(&c->dRawP, sizeof(float) * 3 * (size_t)n)) && ok(hipMalloc(&c->dRawWi, sizeof(float) * 3 * (size_t)n)) &&
                ok(hipMalloc(&c->dRawAlpha, sizeof(float) * 30 * (size_t)n));
    if (!good) { pvol_free_photons(c); pvol_push_scene(c); return PVOL_E_NO_MEMORY; }
    good = ok(hipMemcpy(c->dRawP, p, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice)) &&
           ok(hipMemcpy(c->dRawWi, wi, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice)) &&
           ok(hipMemcpy(c->dRawAlpha, alpha, sizeof(float) * 30 * (size_t)n, hipMemcpyHostToDevice));
    if (!good) { pvol_free_photons(c); pvol_push_scene(c); return PVOL_E_NO_DEVICE; }
    return pvol_finish_map(c, n, p);
}

void pvol_free_caustic_map(pvol_ctx *c) {
    if (c->dCPos4) hipFree(c->dCPos4);
    if (c->dCAlpha4) hipFree(c->dCAlpha4);
    if (c->dCWi4) hipFree(c->dCWi4);
    if (c->dCCellStart) hipFree(c->dCCellStart);
    c->dCPos4 = c->dCAlpha4 = c->dCWi4 = 0; c->dCCellStart = 0;
    memset(&c->hs.surf, 0, sizeof(c->hs.surf));
    c->specOn = false;
}

// PhotonIntegrator::Li in front of the volume term (include/pvol.h).  The caustic map gets the volume map's cell layout
// (pvol_grid.hip) with cells of about maxdist / 2: a lookup gathers everything within maxdist, never fewer.
int pvol_set_surface_integrator(pvol_ctx *c, const pvol_surface_params *sp, const float *p, const float *wo, const float *alpha, uint32_t n) {
    if (!c) return PVOL_E_INVALID;
    if (!c->haveScene) return PVOL_E_NO_SCENE;
    std::lock_guard<std::recursive_mutex> api(c->apiMu);
    if (!ok(hipSetDevice(c->params.device))) return PVOL_E_NO_DEVICE;
    if (!sp) {
        hipDeviceSynchronize();
        pvol_free_caustic_map(c);
        c->specOn = false;
        return pvol_push_scene(c);
    }
    if (sp->n_used < 1 || !(sp->max_dist > 0.f) || sp->max_specular_depth < 0) return PVOL_E_INVALID;
    // matte and glass: a specular BSDF brings the recursion of SpecularReflect / SpecularTransmit (core/integrator.cpp:177-262,
    // pvol_spec_dev.h), walked for "maxspeculardepth" up to SPEC_MAX_DEPTH (the reference's default)
    bool anySpecular = false;
    for (size_t i = 0; i < c->triMatHost.size(); ++i) {
        const int kind = c->hsh.mats[c->triMatHost[i]].kind;
        if (kind == PVOL_MATERIAL_GLASS) anySpecular = true;
        else if (kind != PVOL_MATERIAL_MATTE) return PVOL_E_UNSUPPORTED;
    }
    for (int i = 0; i < c->hs.nSpheres; ++i) {
        const int kind = c->hsh.mats[c->hs.spheres[i].mat].kind;
        if (kind == PVOL_MATERIAL_GLASS) anySpecular = true;
        else if (kind != PVOL_MATERIAL_MATTE) return PVOL_E_UNSUPPORTED;
    }
    if (anySpecular && sp->max_specular_depth > SPEC_MAX_DEPTH) return PVOL_E_UNSUPPORTED;
    // an indirect map makes PhotonIntegrator::Li gather: the final gather, or LPhoton(indirectMap) with 144 more rho draws
    // (photonmap.cpp:183-309).  None of that radiance and none of those draws exist here yet, so such an integrator is
    // refused by name rather than rendered wrong -- whether the map is the caller's (n_indirect_photons) or the store of
    // the last pvol_preprocess.
    if (sp->n_indirect_photons > 0) return PVOL_E_UNSUPPORTED;
    if (sp->use_preprocess_store) {
        if (!c->surfKept) return PVOL_E_INVALID;   // nothing was kept: params.keep_surface_photons was 0, or no pvol_preprocess yet
        if (c->surf[2].n > 0) return PVOL_E_UNSUPPORTED;
    }
    uint32_t nPaths = sp->n_caustic_paths;
    std::vector<float> hp;
    const float *dP = 0, *dWo = 0, *dAlpha = 0;
    float *up[3] = {0, 0, 0};
    auto drop = [&]() { for (int i = 0; i < 3; ++i) if (up[i]) hipFree(up[i]); };
    if (sp->use_preprocess_store) {   // device to device; the positions come back once for the grid bounds
        const pvol_ctx::SurfStore &st = c->surf[0];
        n = st.n; nPaths = st.nPaths;
        dP = st.p; dWo = st.wo; dAlpha = st.alpha;
        hp.resize(3 * (size_t)n);
        if (n && !ok(hipMemcpy(hp.data(), dP, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost))) return PVOL_E_NO_DEVICE;
        p = hp.data();
    } else if (n) {
        if (!p || !wo || !alpha) return PVOL_E_INVALID;
        const size_t nb[3] = {sizeof(float) * 3 * (size_t)n, sizeof(float) * 3 * (size_t)n, sizeof(float) * 30 * (size_t)n};
        const float *src[3] = {p, wo, alpha};
        for (int i = 0; i < 3; ++i) {
            if (!ok(hipMalloc(&up[i], nb[i]))) { drop(); return PVOL_E_NO_MEMORY; }
            if (!ok(hipMemcpy(up[i], src[i], nb[i], hipMemcpyHostToDevice))) { drop(); return PVOL_E_NO_DEVICE; }
        }
        dP = up[0]; dWo = up[1]; dAlpha = up[2];
    }
    if (n && nPaths == 0) { drop(); return PVOL_E_INVALID; }
    hipDeviceSynchronize();
    pvol_free_caustic_map(c);
    DevSurface &sf = c->hs.surf;
    sf.enabled = 1; sf.nLookup = sp->n_used; sf.maxSpecularDepth = sp->max_specular_depth; sf.nCausticPaths = (int32_t)nPaths;
    c->specOn = anySpecular && sp->max_specular_depth > 1;
    sf.maxDistSq = sp->max_dist * sp->max_dist;   // photonmap.cpp:345-346
    sf.nPhotons = 0;
    if (n) {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = 0; i < n; ++i)
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[3 * i + a]); hi[a] = std::max(hi[a], p[3 * i + a]); }
        double cell = 0.5 * sp->max_dist, ext[3];
        for (int a = 0; a < 3; ++a) ext[a] = std::max((double)hi[a] - lo[a], 1e-3 * sp->max_dist);
        for (;;) {
            double cells = 1;
            for (int a = 0; a < 3; ++a) cells *= floor(ext[a] / cell) + 1;
            if (cells <= 16777216.0) break;
            cell *= 1.26;
        }
        sf.cellSize = (float)cell; sf.invCell = 1.f / sf.cellSize;
        size_t ncells = 1;
        for (int a = 0; a < 3; ++a) { sf.gridLo[a] = lo[a]; sf.gdim[a] = (int)floor(ext[a] / cell) + 1; ncells *= (size_t)sf.gdim[a]; }
        bool good = ok(hipMalloc(&c->dCPos4, sizeof(float4) * (size_t)n)) && ok(hipMalloc(&c->dCAlpha4, sizeof(float4) * 8 * (size_t)n)) &&
                    ok(hipMalloc(&c->dCWi4, sizeof(float4) * (size_t)n)) && ok(hipMalloc(&c->dCCellStart, sizeof(uint32_t) * (ncells + 1)));
        if (!good) { drop(); pvol_free_caustic_map(c); pvol_push_scene(c); return PVOL_E_NO_MEMORY; }
        GridBuildArgs g;
        memset(&g, 0, sizeof(g));
        g.p = dP; g.wi = dWo; g.alpha = dAlpha; g.n = n;
        for (int a = 0; a < 3; ++a) { g.lo[a] = sf.gridLo[a]; g.gdim[a] = sf.gdim[a]; }
        g.inv = sf.invCell;
        g.volKind = PVOL_VOLUME_GRID;   // no Inside() filter: surface photons count wherever they lie
        g.sub = 1;
        good = ok(pvol_build_grid(&g, c->dCPos4, c->dCAlpha4, c->dCWi4, c->dCCellStart, 0, 0));
        if (!good) { drop(); pvol_free_caustic_map(c); pvol_push_scene(c); return PVOL_E_NO_DEVICE; }
        sf.nPhotons = n; sf.cellStart = c->dCCellStart; sf.pos4 = c->dCPos4; sf.alpha4 = c->dCAlpha4; sf.wi4 = c->dCWi4;
    }
    drop();
    return pvol_push_scene(c);
}

int pvol_photon_count(pvol_ctx *c, uint32_t *n) {
    if (!c || !n) return PVOL_E_INVALID;
    *n = c->nPhotons;
    return PVOL_OK;
}

int pvol_download_photons(pvol_ctx *c, float *p, float *wi, float *alpha, uint32_t capacity) {
    if (!c || !p || !wi || !alpha) return PVOL_E_INVALID;
    uint32_t n = std::min(capacity, c->nPhotons);
    if (!n) return PVOL_OK;
    if (!ok(hipSetDevice(c->params.device))) return PVOL_E_NO_DEVICE;
    bool good = ok(hipMemcpy(p, c->dRawP, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost)) &&
                ok(hipMemcpy(wi, c->dRawWi, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost)) &&
                ok(hipMemcpy(alpha, c->dRawAlpha, sizeof(float) * 30 * (size_t)n, hipMemcpyDeviceToHost));
    return good ? PVOL_OK : PVOL_E_NO_DEVICE;
}

// LDS plans of the two kernels (pvol_march.hip)
static size_t lds_bytes_seq(const pvol_ctx *c) { return 624 * 4 + (size_t)c->hs.candCap * 8 + (size_t)c->hs.maxSteps * 4 + 256 * 4 + 1024 * 4; }
static size_t lds_bytes_par(const pvol_ctx *c) { return (size_t)c->hs.candCap * 8 + 256 * 4 + 1024 * 4; }

// Drawn VALUES cannot reach Li()'s result with at most one light and an analytic tau():
// such scenes take the ray-parallel kernel, backed by the sequential one if a ray reaches the roulette.
static bool par_eligible(const pvol_ctx *c) { return c->hs.nLights <= 1 && c->hs.volKind != PVOL_VOLUME_GRID; }

static int launch(pvol_ctx *c, const pvol_ray *dRays, uint32_t nRays, pvol_stream *dStreams, uint32_t nStreams, int outputKind,
                  float *dOut, uint32_t *dDraws, const uint32_t *dInit, uint32_t *dFinal, int transOnly, uint32_t maxRaysPerStream,
                  hipStream_t stream) {
    return pvol_launch_batch(c, dRays, nRays, dStreams, nStreams, outputKind, dOut, dDraws, dInit, dFinal, transOnly, maxRaysPerStream, 0, stream);
}

// Whether a single march step can reach the Russian roulette (Tr.y() < 1e-3, photonvolume.cpp:156-161): Tr is
// ASSIGNED per step, a step is at most stepSize long, so exp(-stepSize * max sigma_t) bounds it from below.
static bool roulette_possible(const pvol_ctx *c) {
    float m = 0.f;
    for (int i = 0; i < PVOL_NBINS; ++i) m = std::max(m, c->hs.sigA[i] + c->hs.sigS[i]);
    // VolumeGrid: trilinear interpolation never exceeds the grid maximum, and the stepped tau() of a segment (samples every
    // stepSize/2, DensityRegion::tau) can overshoot the segment by one sample: 1.5 x stepSize bounds it
    const float dens = c->hs.volKind == PVOL_VOLUME_GRID ? 1.5f * c->maxDensity : 1.f;
    return !(c->hs.stepSize * m * dens < 6.8f);
}

// Timing events: every batch records a pair on its launch stream.  Finished pairs are folded into the running sum and
// recycled here, at the next launch and in pvol_kernel_time_ms, so the list stays short however many batches a caller
// issues without ever asking for the time (the per-sample shim issues one per camera sample).  Caller holds c->mu.
static bool harvest_events(pvol_ctx *c, bool wait) {
    size_t keep = 0;
    for (size_t i = 0; i < c->pending.size(); ++i) {
        std::pair<hipEvent_t, hipEvent_t> p = c->pending[i];
        // more than 64 batches in flight: wait for the oldest instead of growing
        const bool block = wait || c->pending.size() - i > 64;
        hipError_t q = block ? hipEventSynchronize(p.second) : hipEventQuery(p.second);
        if (q == hipSuccess) {
            float ms = 0.f;
            if (ok(hipEventElapsedTime(&ms, p.first, p.second))) { c->timeMs += ms; c->launches += 1; }
            c->pool.push_back(p);
        } else if (q == hipErrorNotReady) {
            c->pending[keep++] = p;
        } else {
            (void)hipGetLastError();
            c->pool.push_back(p);
            if (wait) { c->pending.erase(c->pending.begin(), c->pending.begin() + (i + 1 - keep)); return false; }
        }
    }
    c->pending.resize(keep);
    return true;
}

// Phase timing (off by default: two events per kernel group otherwise).  A mark is an event on the launch stream.
void pvol_phase_mark(pvol_ctx *c, hipStream_t stream, int id) {
    if (!c->phaseOn) return;
    std::lock_guard<std::mutex> g(c->mu);
    hipEvent_t e;
    if (!c->phasePool.empty()) { e = c->phasePool.back(); c->phasePool.pop_back(); }
    else if (!ok(hipEventCreate(&e))) return;
    hipEventRecord(e, stream);
    c->phaseMarks.push_back(std::make_pair(id, e));
}

// Waves per render task of the COUNT-mode tile pre-pass.  A task is a serial chain (one MT19937 stream); with many tasks per CU
// the chip is kept busy by running them side by side (one wave each), with few -- one rank's share of a multi-GPU frame -- the
// draw count of every pixel is spread over several waves instead (tile_mw_kernel).  PVOL_TILE_WAVES overrides.
static int tile_waves_per_task(const pvol_ctx *c, uint32_t nTasks) {
    if (c->tileWaves > 0) return c->tileWaves;
    // measured on the C2 frame (tools/cmp_mt.sh, profiles/r03_tile_waves.txt): 16 tasks per CU 231 ms with one wave against 345 with two,
    // 8 per CU 210 with one, 210 with two, 270 with eight; 4 per CU 198 with one, 132 with two, 144 with eight; 2 per CU 194 against 86
    // with eight
    const double perCU = (double)nTasks / (double)std::max(1, c->nCU);
    // (second pass, with the multi-wave kernels held to two waves per SIMD -- 256 VGPRs, so that the workgroups of ALL a CU's tasks are resident:
    // 4 per CU 77 ms with two waves (134 before), 99 with four; 2 per CU 56.5 ms with four waves, 80.8 with eight at 128 VGPRs, 85.9 at 256)
    // 8 per CU: 131.5 ms with two waves, 146.9 with one; 16 per CU: 174.7 with one, 190.7 with two
    return perCU >= 12.0 ? 1 : (perCU >= 3.0 ? 2 : 4);
}

// ---- specular recursion (pvol_spec_dev.h): the pool of segment rays of one batch (COUNT mode) or one slice (FUSED mode)
static int spec_pool_prepare(pvol_ctx *c, size_t cap, size_t recStride, hipStream_t stream) {
    if (cap > c->segCap) {
        hipStreamSynchronize(stream);
        if (c->dSegRays) hipFree(c->dSegRays);
        if (c->dSegInfo) hipFree(c->dSegInfo);
        if (c->dSegOut) hipFree(c->dSegOut);
        c->dSegRays = 0; c->dSegInfo = 0; c->dSegOut = 0; c->segCap = 0;
        if (!ok(hipMalloc(&c->dSegRays, sizeof(pvol_ray) * cap)) || !ok(hipMalloc(&c->dSegInfo, sizeof(SegInfo) * cap)) ||
            !ok(hipMalloc(&c->dSegOut, sizeof(float) * 60 * cap))) return PVOL_E_NO_MEMORY;
        c->segCap = cap;
    }
    if (recStride * cap > c->segRecBytes) {
        hipStreamSynchronize(stream);
        if (c->dSegRecords) hipFree(c->dSegRecords);
        c->dSegRecords = 0; c->segRecBytes = 0;
        if (!ok(hipMalloc(&c->dSegRecords, recStride * cap))) return PVOL_E_NO_MEMORY;
        c->segRecBytes = recStride * cap;
    }
    if (!c->dSegCounter && !ok(hipMalloc(&c->dSegCounter, 16))) return PVOL_E_NO_MEMORY;
    if (!c->dSegStream && !ok(hipMalloc(&c->dSegStream, sizeof(pvol_stream)))) return PVOL_E_NO_MEMORY;
    memset(&c->hSegStream, 0, sizeof(pvol_stream));
    c->hSegStream.n_rays = (uint32_t)cap;
    if (!ok(hipMemsetAsync(c->dSegCounter, 0, 16, stream)) || !ok(hipMemcpyAsync(c->dSegStream, &c->hSegStream, sizeof(pvol_stream), hipMemcpyHostToDevice, stream)) ||
        !ok(pvol_launch_spec_fill(c->dSegRays, (uint32_t)cap, stream)) || !ok(hipMemsetAsync(c->dSegOut, 0, sizeof(float) * 60 * cap, stream)))
        return PVOL_E_NO_DEVICE;
    return PVOL_OK;
}
static size_t spec_pool_cap(size_t raysInFlight) {   // segments of one batch / slice: twice its camera samples, 64 k .. 16 M (the link holds 25 bits)
    size_t cap = std::max<size_t>(65536, 2 * raysInFlight);
    if (const char *ev = getenv("PVOL_SPEC_POOL")) { long long v = atoll(ev); if (v > 0) cap = (size_t)v; }
    return std::min<size_t>(cap, (size_t)1 << 24);
}
static void spec_fill_tile(const pvol_ctx *c, TileArgs *t, size_t cap) {
    t->specOn = 1; t->segRays = c->dSegRays; t->segInfo = c->dSegInfo; t->segCounter = c->dSegCounter; t->segCap = (uint32_t)cap;
    t->segRecords = c->dSegRecords;
}
// The segments' own volume Li() (spectral), the surface term at their matte hits, and the fold into the camera samples.
// `replay`: the pool's records were written by the FUSED tile pre-pass (li_replay_kernel); else no drawn value matters (li_par_kernel).
static int spec_finish(pvol_ctx *c, const LiArgs &a, size_t cap, bool replay, uint32_t nPrimary, float *primaryOut, float *surfOut, hipStream_t stream) {
    LiArgs sa = a;
    sa.rays = c->dSegRays; sa.nRays = (uint32_t)cap; sa.streams = c->dSegStream; sa.nStreams = 1; sa.outputKind = PVOL_OUT_SPECTRAL;
    sa.out = c->dSegOut; sa.draws = 0; sa.initState = 0; sa.finalState = 0; sa.tauOut = 0; sa.defer = 0; sa.deferCount = 0; sa.deferCap = 0; sa.gated = 0;
    sa.records = c->dSegRecords; sa.sliceM = (uint32_t)cap; sa.sliceK = 0; sa.state = 0;
    const uint32_t nWaves = (uint32_t)std::min<unsigned long long>((cap + 63) / 64, (unsigned long long)c->nCU * 16ull);
    if (!ok(hipMemsetAsync(c->dWords, 0, 4 * sizeof(uint32_t), stream))) return PVOL_E_NO_DEVICE;
    hipError_t e = replay ? pvol_launch_li_replay(&sa, lds_bytes_par(c), c->hs.candCap, nWaves, stream)
                          : pvol_launch_li_par(&sa, lds_bytes_par(c), c->hs.candCap, false, nWaves, stream);
    if (!ok(e)) return PVOL_E_NO_DEVICE;
    SurfArgs su;
    memset(&su, 0, sizeof(su));
    su.scene = c->ds; su.rays = c->dSegRays; su.nRays = (uint32_t)cap; su.out = c->dSegOut; su.tau = 0; su.surfOut = 0; su.counters = c->dCounters; su.link = 0; su.spectral = 1;
    if (!ok(pvol_launch_surface(&su, (uint32_t)std::min<unsigned long long>((cap + 63) / 64, (unsigned long long)c->nCU * 24ull), stream))) return PVOL_E_NO_DEVICE;
    SpecComposeArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.scene = c->ds; ca.link = c->dSpecLink; ca.info = c->dSegInfo; ca.segOut = c->dSegOut; ca.tau = a.tauOut; ca.out = primaryOut; ca.surfOut = surfOut;
    ca.first = 0; ca.nRays = nPrimary;
    return ok(pvol_launch_spec_compose(&ca, stream)) ? PVOL_OK : PVOL_E_NO_DEVICE;
}

// `tile` != 0: the rays do not exist yet -- the tile kernel (pvol_tile_dev.h) generates them stream by stream
// (LD sampler + camera) in front of the march, and takes the place of the RESOLVE pre-pass where one is needed.
int pvol_launch_batch(pvol_ctx *c, const pvol_ray *dRays, uint32_t nRays, pvol_stream *dStreams, uint32_t nStreams, int outputKind,
                      float *dOut, uint32_t *dDraws, const uint32_t *dInit, uint32_t *dFinal, int transOnly, uint32_t maxRaysPerStream,
                      const TileArgs *tile, hipStream_t stream) {
    TileArgs tloc;   // the tile driver's arguments, completed here where the batch is cut into slices (segment pool of the specular recursion)
    if (tile) { tloc = *tile; tile = &tloc; }
    const bool spec = tile && tloc.specOn;
    size_t specCap = 0;
    LiArgs a;
    memset(&a, 0, sizeof(a));
    a.scene = c->ds; a.rays = dRays; a.streams = dStreams; a.nStreams = nStreams; a.nRays = nRays; a.outputKind = outputKind;
    a.out = dOut; a.draws = dDraws; a.initState = dInit; a.finalState = dFinal; a.counters = c->dCounters;
    a.transmittanceOnly = transOnly;
    a.chunkCounter = c->dWords; a.needSeq = c->dWords + 1; a.gated = 0;
    a.tauOut = c->dTauNext;   // render driver with the surface integrator on: li_group_kernel also reports the optical length
    // li_group_kernel bucket radius^2 = this x the guessed k-th distance^2 (measured at 64 spp: 1.3 55.5, 1.2 57.5, 1.12 58.0,
    // 1.06 56.6, 1.0 51.0 Msamples/s)
    { const char *gs = getenv("PVOL_GROUP_GUESS"); a.grpGuess = gs ? (float)atof(gs) : 1.15f; if (!(a.grpGuess >= 1.f)) a.grpGuess = 1.15f; }
    std::pair<hipEvent_t, hipEvent_t> ev;
    {
        std::lock_guard<std::mutex> g(c->mu);
        harvest_events(c, false);
        if (!c->pool.empty()) { ev = c->pool.back(); c->pool.pop_back(); }
        else if (!ok(hipEventCreate(&ev.first)) || !ok(hipEventCreate(&ev.second))) return PVOL_E_NO_DEVICE;
    }
    // the pair goes back to the pool on every early return below (an UNSUPPORTED batch must not leak events)
    struct EventReturn {
        pvol_ctx *c; std::pair<hipEvent_t, hipEvent_t> ev; bool keep = false;
        ~EventReturn() { if (!keep) { std::lock_guard<std::mutex> g(c->mu); c->pool.push_back(ev); } }
    } evGuard{c, ev};
    hipError_t e;
    // the tile pre-pass can COUNT Li()'s draws (instead of drawing them) under the same conditions as li_par_kernel,
    // provided no march step can reach the roulette
    const bool tileCount = tile && par_eligible(c) && !dInit && !roulette_possible(c);
    // a VolumeGrid with at most one light: the draw COUNT is still geometry only (4 + 6n + n + u; the drawn offsets change values,
    // not counts), so the tile pre-pass counts and the values come from the RNG-only resolve pass of each slice
    const bool tileGridCount = tile && c->hs.nLights <= 1 && c->hs.volKind == PVOL_VOLUME_GRID && !dInit && !roulette_possible(c) && !c->forceSeq && !c->noLite;
    const bool par = par_eligible(c) && !dInit && !c->forceSeq && (!tile || tileCount);
    // scenes where drawn values matter: sequential RESOLVE pre-pass + ray-parallel REPLAY, slice by slice
    const bool sliced = !par && !c->forceSeq && !transOnly && (c->hs.volKind != PVOL_VOLUME_NONE || tile);
    if (tile && !par && !sliced && !tileCount) return PVOL_E_UNSUPPORTED;
    uint32_t sliceM = 0, nSlices = 0;
    if (sliced) {
        uint32_t maxRays = maxRaysPerStream;
        if (maxRays == 0) {   // device entry point: the stream table lives on the device
            std::vector<pvol_stream> hs(nStreams);   // read on the caller's stream: ordered behind whatever produced the table
            if (!ok(hipMemcpyAsync(hs.data(), dStreams, sizeof(pvol_stream) * (size_t)nStreams, hipMemcpyDeviceToHost, stream)) ||
                !ok(hipStreamSynchronize(stream))) return PVOL_E_NO_DEVICE;
            for (uint32_t i = 0; i < nStreams; ++i) maxRays = std::max(maxRays, hs[i].n_rays);
        }
        const bool grid = c->hs.volKind == PVOL_VOLUME_GRID;
        size_t stride = 16 + (size_t)((c->hs.maxSteps + 15) & ~15) + (grid ? 8 * (size_t)c->hs.maxSteps : 0);
        stride = (stride + 15) & ~(size_t)15;
        const size_t budget = (size_t)4 << 30;
        size_t m = budget / (stride * (size_t)nStreams);
        // nused beyond the bucket plan hands every dense lookup of a slice to the exact pass: keep that list within 8 GB
        if (c->hs.nUsed > 100) m = std::min<size_t>(m, std::max<size_t>(64, (((size_t)8 << 30) / sizeof(DeferRec) / 64) / (size_t)nStreams));
        if (spec) m = std::min<size_t>(m, std::max<size_t>(64, ((size_t)8 << 20) / (size_t)nStreams));   // keeps a slice's segment pool within its 16 M slots
        m = std::max<size_t>(64, std::min<size_t>(m, ((size_t)maxRays + 63) & ~(size_t)63));
        m &= ~(size_t)63;
        if (const char *ev = getenv("PVOL_SLICE_RAYS")) { long v = atol(ev); if (v >= 64) m = (size_t)v & ~(size_t)63; }   // testing: force many slices
        if (tile) {   // a slice holds whole pixels (both powers of two)
            const size_t unit = std::max<size_t>(64, tile->spp);
            m = std::max(unit, m / unit * unit);
        }
        sliceM = (uint32_t)m;
        nSlices = maxRays ? (maxRays + sliceM - 1) / sliceM : 1;
        size_t recBytes = stride * (size_t)nStreams * sliceM, stBytes = sizeof(uint32_t) * 625 * (size_t)nStreams;
        if (recBytes > c->recBytes) { if (c->dRecords) hipFree(c->dRecords); c->dRecords = 0; c->recBytes = 0;
                                      if (!ok(hipMalloc(&c->dRecords, recBytes))) return PVOL_E_NO_MEMORY; c->recBytes = recBytes; }
        if (stBytes > c->stateBytes) { if (c->dState) hipFree(c->dState); c->dState = 0; c->stateBytes = 0;
                                       if (!ok(hipMalloc(&c->dState, stBytes))) return PVOL_E_NO_MEMORY; c->stateBytes = stBytes; }
        a.records = c->dRecords; a.recStride = (uint32_t)stride; a.sliceM = sliceM; a.state = c->dState;
        // with a tile pre-pass in FUSED mode (several lights) the flag selects the pre-pass's own geometry + RNG-only form
        a.liteResolve = (!c->noLite && !roulette_possible(c)) ? 1 : 0;
    }
    if (par) hipMemsetAsync(c->dWords, 0, 3 * sizeof(uint32_t), stream);
    if (tile && (par || (!sliced && tileCount))) {   // sampler + camera pre-pass, outside the timed region of the march kernel
        a.sliceK = 0; a.sliceM = 0xffffffc0u; a.state = 0;
        if (spec) {
            specCap = spec_pool_cap(nRays);
            int rc = spec_pool_prepare(c, specCap, 0, stream);
            if (rc != PVOL_OK) return rc;
            spec_fill_tile(c, &tloc, specCap);
        }
        pvol_phase_mark(c, stream, PVOL_PHASE_TILE);
        if (!ok(pvol_launch_tile(&a, tile, false, pvol_tile_lds_bytes(0, tile->spp, false, c->hs.nTris, c->hs.nLights > 0 && c->hs.lights[0].kind == PVOL_LIGHT_DISTANT), c->hs.candCap, stream,
                                 tile_waves_per_task(c, nStreams)))) return PVOL_E_NO_DEVICE;
    }
    hipEventRecord(ev.first, stream);
    pvol_phase_mark(c, stream, PVOL_PHASE_MARCH);
    if (par) {
        unsigned long long chunks = ((unsigned long long)nRays + 63ull) / 64ull;
        uint32_t nWaves = (uint32_t)std::min<unsigned long long>(chunks, (unsigned long long)c->nCU * 16ull);
        // homogeneous isotropic medium with a photon map and k <= 64: one ray per lane, gathers of 64 rays share a bucket
        const bool group = !c->noGroup && c->hs.volKind == PVOL_VOLUME_HOMOGENEOUS && c->hs.g == 0.f && c->hs.nPhotons > 0 &&
                           c->hs.nUsed >= 10 && c->hs.nUsed <= 64 && c->hs.candCap <= 4 * 64;
        if (a.tauOut && !group) return PVOL_E_UNSUPPORTED;
        if (group) {
            unsigned long long gchunks = ((unsigned long long)nRays + 511ull) / 512ull;   // GRP_CH rays per chunk
            uint32_t gWaves = (uint32_t)std::min<unsigned long long>(gchunks, (unsigned long long)c->nCU * (unsigned long long)c->groupWavesPerCU);
            // room for the lookups the bucket plan hands over (a fraction of a percent of ~40 per ray on C2); a list that
            // overflows raises needSeq and the batch is redone sequentially, never truncated
            const size_t wantDefer = (size_t)nRays / 2 + 65536;
            if (wantDefer > c->deferCap) {
                hipStreamSynchronize(stream);   // an earlier batch may still read the old list
                if (c->dDefer) hipFree(c->dDefer);
                c->dDefer = 0; c->deferCap = 0;
                if (!ok(hipMalloc(&c->dDefer, wantDefer * sizeof(DeferRec)))) return PVOL_E_NO_MEMORY;
                c->deferCap = wantDefer;
            }
            a.defer = c->dDefer; a.deferCount = c->dWords + 2; a.deferCap = (uint32_t)std::min<size_t>(c->deferCap, 0xffffffffu);
            e = pvol_launch_li_group(&a, pvol_group_lds_bytes(c->hs.candCap), c->hs.candCap, c->statsOn, gWaves, (uint32_t)(c->nCU * c->fixWavesPerCU), 0, stream);
            c->lastKernel = "li_group_kernel";
        } else {
            c->lastKernel = "li_par_kernel";
            e = pvol_launch_li_par(&a, lds_bytes_par(c), c->hs.candCap, c->statsOn, nWaves, stream);
        }
        if (ok(e)) {   // runs only if a ray raised needSeq (gate read on the device: no host sync here)
            a.gated = 1;
            e = pvol_launch_li_seq(&a, lds_bytes_seq(c), c->hs.candCap, c->statsOn, stream);
            a.gated = 0;
        }
        if (ok(e) && spec) {
            int rc = spec_finish(c, a, specCap, false, nRays, dOut, c->specSurfOut, stream);
            if (rc != PVOL_OK) return rc;
        }
    } else if (a.tauOut && (!sliced || c->hs.volKind != PVOL_VOLUME_HOMOGENEOUS)) {
        return PVOL_E_UNSUPPORTED;   // the surface term needs li_group_kernel's optical length of a homogeneous medium
    } else if (sliced) {
        c->lastKernel = "li_replay_kernel";
        e = hipSuccess;
        unsigned long long chunks = (unsigned long long)((sliceM + 63) / 64) * nStreams;
        uint32_t nWaves = (uint32_t)std::min<unsigned long long>(chunks, (unsigned long long)c->nCU * 16ull);
        // li_group_kernel's replay form (one ray per lane, shared photon bucket) where no march step can reach the roulette, the
        // medium is isotropic and a photon map exists; li_replay_kernel (one wave per ray) otherwise, and as the gated backup
        const bool gridVol = c->hs.volKind == PVOL_VOLUME_GRID;
        int groupForm = 0;
        if (!c->noGroup && !c->statsOn && !roulette_possible(c) && c->hs.g == 0.f && c->hs.nPhotons > 0 && c->hs.nUsed >= 10 &&
            (c->hs.volKind == PVOL_VOLUME_HOMOGENEOUS || gridVol))
            groupForm = gridVol ? 2 : 1;
        uint32_t gWaves = 0;
        if (groupForm) {
            const unsigned long long gchunks = (unsigned long long)((sliceM + 511) / 512) * nStreams;
            gWaves = (uint32_t)std::min<unsigned long long>(gchunks, (unsigned long long)c->nCU * (unsigned long long)c->groupWavesPerCU);
            // hand-over list: nused beyond the bucket plan sends every dense lookup to the exact pass (C3: ~14 per ray)
            const size_t perSlice = (size_t)sliceM * nStreams;
            size_t wantDefer = (c->hs.nUsed > 100 ? perSlice * 64 : perSlice) + 65536;
            wantDefer = std::min<size_t>(wantDefer, ((size_t)8 << 30) / sizeof(DeferRec));
            if (wantDefer > c->deferCap) {
                hipStreamSynchronize(stream);
                if (c->dDefer) hipFree(c->dDefer);
                c->dDefer = 0; c->deferCap = 0;
                if (!ok(hipMalloc(&c->dDefer, wantDefer * sizeof(DeferRec)))) return PVOL_E_NO_MEMORY;
                c->deferCap = wantDefer;
            }
            a.defer = c->dDefer; a.deferCount = c->dWords + 2; a.deferCap = (uint32_t)std::min<size_t>(c->deferCap, 0xffffffffu);
            a.fixGroup = (c->hs.nUsed > 100 && !getenv("PVOL_FIX_EXACT")) ? 1 : 0;   // GRP_PLAN_KMAX: see pvol_fixgrp_dev.h
            { const char *ew = getenv("PVOL_FXG_WIDEN"), *ea = getenv("PVOL_FXG_AIM"); a.fxgWiden = ew ? (float)atof(ew) : 0.f; a.fxgAim = ea ? (float)atof(ea) : 0.f; }   // measurement knobs
            c->lastKernel = "li_group_kernel";
        }
        if (a.tauOut && !groupForm) return PVOL_E_UNSUPPORTED;
        if (spec && !(tile && !tileGridCount && a.liteResolve)) return PVOL_E_UNSUPPORTED;   // the FUSED pre-pass walks the segments (geo_ray + lite_ray)
        if (spec) specCap = spec_pool_cap((size_t)sliceM * nStreams);
        if (tileGridCount) {   // sampler + camera + draw COUNT for the whole batch, once
            LiArgs t = a;
            t.sliceK = 0; t.sliceM = 0xffffffc0u; t.state = 0;
            pvol_phase_mark(c, stream, PVOL_PHASE_TILE);
            e = pvol_launch_tile(&t, tile, false, pvol_tile_lds_bytes(0, tile->spp, false, c->hs.nTris, c->hs.nLights > 0 && c->hs.lights[0].kind == PVOL_LIGHT_DISTANT), c->hs.candCap, stream,
                                 tile_waves_per_task(c, nStreams));
        }
        for (uint32_t k = 0; k < nSlices && ok(e); ++k) {
            a.sliceK = k;
            hipMemsetAsync(c->dWords, 0, 4 * sizeof(uint32_t), stream);
            if (spec) {
                int rc = spec_pool_prepare(c, specCap, a.recStride, stream);
                if (rc != PVOL_OK) return rc;
                spec_fill_tile(c, &tloc, specCap);
            }
            if (tile && !tileGridCount) {
                pvol_phase_mark(c, stream, PVOL_PHASE_TILE);
                e = pvol_launch_tile(&a, tile, true, pvol_tile_lds_bytes(c->hs.maxSteps, tile->spp, true, c->hs.nTris, false), c->hs.candCap, stream, 1);
            }
            pvol_phase_mark(c, stream, PVOL_PHASE_MARCH);   // incl. the RNG-only resolve pass of a slice where there is one
            if (ok(e)) e = pvol_launch_li_slice(&a, 624 * 4 + (size_t)c->hs.maxSteps * 4, lds_bytes_par(c), c->hs.candCap, c->statsOn, nWaves, stream,
                                                tile == 0 || tileGridCount, groupForm, pvol_group_lds_bytes(c->hs.candCap), gWaves, (uint32_t)(c->nCU * c->fixWavesPerCU));
            if (ok(e) && spec) {   // this slice's segments: their Li() from the records the pre-pass left, then the fold into the slice's camera samples
                int rc = spec_finish(c, a, specCap, true, nRays, dOut, c->specSurfOut, stream);
                if (rc != PVOL_OK) return rc;
            }
        }
    } else {
        c->lastKernel = "li_seq_kernel";
        e = pvol_launch_li_seq(&a, lds_bytes_seq(c), c->hs.candCap, c->statsOn, stream);
    }
    hipEventRecord(ev.second, stream);
    pvol_phase_mark(c, stream, PVOL_PHASE_END);
    {
        std::lock_guard<std::mutex> g(c->mu);
        c->pending.push_back(ev);
        evGuard.keep = true;
    }
    return ok(e) ? PVOL_OK : PVOL_E_NO_DEVICE;
}

static int check_errors(pvol_ctx *c) {
    DevCounters h;
    if (!ok(hipMemcpy(&h, c->dCounters, sizeof(h), hipMemcpyDeviceToHost))) return PVOL_E_NO_DEVICE;
    if (h.nErrors) {
        unsigned long long zero = 0;
        hipMemcpy(&c->dCounters->nErrors, &zero, sizeof(zero), hipMemcpyHostToDevice);
        return PVOL_E_LIMIT;
    }
    return PVOL_OK;
}

int pvol_li_batch_device(pvol_ctx *c, const pvol_ray *dRays, uint32_t nRays, pvol_stream *dStreams, uint32_t nStreams,
                         int outputKind, float *dOut, uint32_t *dDraws, void *hipStream) {
    if (!c || (nRays && !dRays) || (nStreams && !dStreams) || (nRays && !dOut)) return PVOL_E_INVALID;
    if (outputKind != PVOL_OUT_SPECTRAL && outputKind != PVOL_OUT_XYZ) return PVOL_E_INVALID;
    if (!c->haveScene) return PVOL_E_NO_SCENE;
    if (!nStreams) return PVOL_OK;
    std::lock_guard<std::recursive_mutex> api(c->apiMu);
    if (!ok(hipSetDevice(c->params.device))) return PVOL_E_NO_DEVICE;
    return launch(c, dRays, nRays, dStreams, nStreams, outputKind, dOut, dDraws, 0, 0, 0, 0, (hipStream_t)hipStream);
}

int pvol_check_errors(pvol_ctx *c) {
    if (!c) return PVOL_E_INVALID;
    std::lock_guard<std::recursive_mutex> api(c->apiMu);
    if (!ok(hipSetDevice(c->params.device)) || !ok(hipDeviceSynchronize())) return PVOL_E_NO_DEVICE;
    return check_errors(c);
}

static int host_batch(pvol_ctx *c, const pvol_ray *rays, uint32_t nRays, pvol_stream *streams, uint32_t nStreams, int outputKind,
                      float *out, uint32_t *draws, uint32_t *mtState, int transOnly) {
    if (!c || (nRays && (!rays || !out)) || (nStreams && !streams)) return PVOL_E_INVALID;
    if (!c->haveScene) return PVOL_E_NO_SCENE;
    // VolumeIntegrator::Li() is called by every SamplerRendererTask thread at once (samplerrenderer.cpp:247): calls on one
    // context are serialised here, from the upload to the copy-back (the launches share the context's scratch)
    std::lock_guard<std::recursive_mutex> api(c->apiMu);
    if (!ok(hipSetDevice(c->params.device))) return PVOL_E_NO_DEVICE;
    uint32_t maxRays = 1;
    {   // streams must partition the ray array in order (every ray belongs to exactly one stream)
        uint64_t next = 0;
        for (uint32_t s = 0; s < nStreams; ++s) {
            if (streams[s].first_ray != next) return PVOL_E_INVALID;
            next += streams[s].n_rays;
            maxRays = std::max(maxRays, streams[s].n_rays);
        }
        if (next != nRays) return PVOL_E_INVALID;
    }
    if (!nStreams || !nRays) return PVOL_OK;
    const size_t width = transOnly ? 60 : (outputKind == PVOL_OUT_SPECTRAL ? 60 : 4);
    pvol_ray *dRays = 0; pvol_stream *dStreams = 0; float *dOut = 0; uint32_t *dDraws = 0, *dState = 0;
    int rc = PVOL_OK;
    bool good = ok(hipMalloc(&dRays, sizeof(pvol_ray) * (size_t)nRays)) && ok(hipMalloc(&dStreams, sizeof(pvol_stream) * (size_t)nStreams)) &&
                ok(hipMalloc(&dOut, sizeof(float) * width * nRays)) && ok(hipMalloc(&dDraws, sizeof(uint32_t) * (size_t)nRays));
    if (good && mtState) good = ok(hipMalloc(&dState, sizeof(uint32_t) * 625 * (size_t)nStreams));
    if (!good) rc = PVOL_E_NO_MEMORY;
    if (rc == PVOL_OK) {
        good = ok(hipMemcpy(dRays, rays, sizeof(pvol_ray) * (size_t)nRays, hipMemcpyHostToDevice)) &&
               ok(hipMemcpy(dStreams, streams, sizeof(pvol_stream) * (size_t)nStreams, hipMemcpyHostToDevice)) &&
               ok(hipMemset(dOut, 0, sizeof(float) * width * nRays));
        if (good && mtState) good = ok(hipMemcpy(dState, mtState, sizeof(uint32_t) * 625 * (size_t)nStreams, hipMemcpyHostToDevice));
        if (!good) rc = PVOL_E_NO_DEVICE;
    }
    if (rc == PVOL_OK) rc = launch(c, dRays, nRays, dStreams, nStreams, transOnly ? PVOL_OUT_SPECTRAL : outputKind, dOut, dDraws, dState, dState, transOnly, maxRays, 0);
    if (rc == PVOL_OK && !ok(hipStreamSynchronize(0))) rc = PVOL_E_NO_DEVICE;
    if (rc == PVOL_OK) rc = check_errors(c);
    if (rc == PVOL_OK) {
        if (transOnly) {
            // kernel wrote [Lv(30) | T(30)] rows; the ABI returns T only
            std::vector<float> tmp(width * nRays);
            good = ok(hipMemcpy(tmp.data(), dOut, sizeof(float) * width * nRays, hipMemcpyDeviceToHost));
            if (good) for (uint32_t i = 0; i < nRays; ++i) memcpy(out + (size_t)i * 30, tmp.data() + (size_t)i * 60 + 30, sizeof(float) * 30);
        } else {
            good = ok(hipMemcpy(out, dOut, sizeof(float) * width * nRays, hipMemcpyDeviceToHost));
        }
        good = good && ok(hipMemcpy(streams, dStreams, sizeof(pvol_stream) * (size_t)nStreams, hipMemcpyDeviceToHost));
        if (good && draws) good = ok(hipMemcpy(draws, dDraws, sizeof(uint32_t) * (size_t)nRays, hipMemcpyDeviceToHost));
        if (good && mtState) good = ok(hipMemcpy(mtState, dState, sizeof(uint32_t) * 625 * (size_t)nStreams, hipMemcpyDeviceToHost));
        if (!good) rc = PVOL_E_NO_DEVICE;
    }
    if (dRays) hipFree(dRays);
    if (dStreams) hipFree(dStreams);
    if (dOut) hipFree(dOut);
    if (dDraws) hipFree(dDraws);
    if (dState) hipFree(dState);
    return rc;
}

int pvol_li_batch(pvol_ctx *c, const pvol_ray *rays, uint32_t nRays, pvol_stream *streams, uint32_t nStreams, int outputKind,
                  float *out, uint32_t *draws) {
    if (outputKind != PVOL_OUT_SPECTRAL && outputKind != PVOL_OUT_XYZ) return PVOL_E_INVALID;
    return host_batch(c, rays, nRays, streams, nStreams, outputKind, out, draws, 0, 0);
}

int pvol_transmittance_batch(pvol_ctx *c, const pvol_ray *rays, uint32_t nRays, pvol_stream *streams, uint32_t nStreams, float *out) {
    return host_batch(c, rays, nRays, streams, nStreams, PVOL_OUT_SPECTRAL, out, 0, 0, 1);
}

int pvol_li(pvol_ctx *c, const pvol_ray *ray, uint32_t *mt, int32_t *mti, float *Lv, float *T) {
    if (!c || !ray || !mt || !mti || !Lv || !T) return PVOL_E_INVALID;
    if (*mti < 0 || *mti > 624) return PVOL_E_INVALID;
    uint32_t state[625];
    memcpy(state, mt, sizeof(uint32_t) * 624);
    state[624] = (uint32_t)*mti;
    pvol_stream st;
    memset(&st, 0, sizeof(st));
    st.n_rays = 1;
    float out[60];
    int rc = host_batch(c, ray, 1, &st, 1, PVOL_OUT_SPECTRAL, out, 0, state, 0);
    if (rc != PVOL_OK) return rc;
    memcpy(Lv, out, sizeof(float) * 30);
    memcpy(T, out + 30, sizeof(float) * 30);
    memcpy(mt, state, sizeof(uint32_t) * 624);
    *mti = (int32_t)state[624];
    return PVOL_OK;
}

int pvol_enable_stats(pvol_ctx *c, int on) {
    if (!c) return PVOL_E_INVALID;
    c->statsOn = on != 0;
    return PVOL_OK;
}

int pvol_get_stats(pvol_ctx *c, pvol_stats *out, int reset) {
    if (!c || !out) return PVOL_E_INVALID;
    if (!ok(hipSetDevice(c->params.device)) || !ok(hipDeviceSynchronize())) return PVOL_E_NO_DEVICE;
    DevCounters h;
    if (!ok(hipMemcpy(&h, c->dCounters, sizeof(h), hipMemcpyDeviceToHost))) return PVOL_E_NO_DEVICE;
    memset(out, 0, sizeof(*out));
    out->n_rays = h.nRays; out->n_steps = h.nSteps; out->n_tested = h.nTested; out->n_kept = h.nKept;
    out->n_lookups_lt10 = h.nLookupsLt10; out->n_shadow_unoccluded = h.nShadowUnoccluded;
    out->n_guess_retries = h.pad;
    out->group_guess_failed = h.diag[0]; out->group_plan_skipped = h.diag[1]; out->cy_fallback = h.diag[2];
    out->group_deferred_overflow = h.diag[3]; out->group_deferred_too_few = h.diag[4]; out->group_attempts = h.diag[5];
    out->cy_search = h.cySearch; out->cy_select = h.cySelect; out->cy_flux = h.cyFlux; out->cy_total = h.cyTotal;
    if (reset && !ok(hipMemset(c->dCounters, 0, sizeof(DevCounters)))) return PVOL_E_NO_DEVICE;
    return PVOL_OK;
}

const char *pvol_march_kernel_name(pvol_ctx *c) { return c ? c->lastKernel : ""; }

int pvol_enable_phase_timing(pvol_ctx *c, int on) {
    if (!c) return PVOL_E_INVALID;
    c->phaseOn = on != 0;
    return PVOL_OK;
}

int pvol_get_phase_ms(pvol_ctx *c, double *out6, int reset) {
    if (!c || !out6) return PVOL_E_INVALID;
    if (!ok(hipSetDevice(c->params.device)) || !ok(hipDeviceSynchronize())) return PVOL_E_NO_DEVICE;
    std::lock_guard<std::mutex> g(c->mu);
    for (size_t i = 0; i + 1 < c->phaseMarks.size(); ++i) {
        const int id = c->phaseMarks[i].first;
        float ms = 0.f;
        if (id >= 0 && id < PVOL_N_PHASES && ok(hipEventElapsedTime(&ms, c->phaseMarks[i].second, c->phaseMarks[i + 1].second))) c->phaseMs[id] += ms;
    }
    for (size_t i = 0; i < c->phaseMarks.size(); ++i) c->phasePool.push_back(c->phaseMarks[i].second);
    c->phaseMarks.clear();
    for (int i = 0; i < PVOL_N_PHASES; ++i) out6[i] = c->phaseMs[i];
    if (reset) for (int i = 0; i < PVOL_N_PHASES; ++i) c->phaseMs[i] = 0.0;
    return PVOL_OK;
}

int pvol_kernel_time_ms(pvol_ctx *c, double *avgMs, uint64_t *launches, int reset) {
    if (!c || !avgMs) return PVOL_E_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    if (!harvest_events(c, true)) return PVOL_E_NO_DEVICE;
    *avgMs = c->launches ? c->timeMs / (double)c->launches : 0.0;
    if (launches) *launches = c->launches;
    if (reset) { c->timeMs = 0; c->launches = 0; }
    return PVOL_OK;
}

}  // extern "C"
