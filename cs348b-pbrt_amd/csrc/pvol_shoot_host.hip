// pvol_shoot_host.hip -- host side of PhotonShooter::Preprocess (core/photonshooter.cpp:457-526):
// rounds of one 4096-path block per live virtual task on the device, merged in task order with the
// reference's bookkeeping (running nshot, per-task *Done flags, the "unable to store enough photons"
// abort, photonshooter.cpp:280-356), then the search-structure build.
#include <math.h>
#include <string.h>
#include <chrono>
#include <algorithm>
#include <vector>

#include "pvol_host.h"

struct ShootArgs {
    const DevScene *scene;
    const DevShootScene *shoot;
    uint32_t nTasks;
    const uint32_t *stateIn;
    uint32_t *stateOut;
    uint32_t *halton;
    const uint32_t *flags;
    float *localPhotons;
    uint32_t *localCounts;
    uint32_t cap;
    float *localSurf;
    uint32_t *localSurfKind;
    uint32_t capS;
    float *localRad;
    uint32_t capR;
    int keepSurface;
    unsigned long long *stats;
    int init;
    uint32_t blockPaths;      // paths per task and round (4096: PhotonShootingTask::Run's block, photonshooter.cpp:247)
    int gridVolume;           // the medium is a VolumeGrid: the kernel takes GRID_KMAX x 64 more LDS words (march_grid)
};
struct SurfMergeArgs {
    const float *localSurf; const uint32_t *localSurfKind; uint32_t capS;
    const float *localRad; uint32_t capR;
    const uint32_t *srcTask, *nSurf, *take, *dstOff;
    const uint32_t *nRad;
    uint32_t nSeg;
    float *p[3], *wo[3], *alpha[3];
    float *rad;
};
extern "C" hipError_t pvol_launch_merge_surface(const SurfMergeArgs *m, hipStream_t stream);
struct MergeArgs {
    const float *localPhotons;
    uint32_t cap;
    const uint32_t *srcTask;
    const uint32_t *count;
    const uint32_t *dstOff;
    const float *nshot;
    uint32_t nSeg;
    float *p, *wi, *alpha;
};
extern "C" hipError_t pvol_launch_shoot(const ShootArgs *a, hipStream_t stream);
extern "C" size_t pvol_shoot_state_words(void);
extern "C" hipError_t pvol_launch_merge(const MergeArgs *m, hipStream_t stream);

static bool ok(hipError_t e) { return e == hipSuccess; }

namespace {
struct DevArr {   // device array of floats that grows geometrically, keeping its contents
    float *d = 0;
    size_t cap = 0;
    bool need(size_t used, size_t want) {
        if (want <= cap) return true;
        size_t nc = std::max(want, cap * 2 + 1024);
        float *nd = 0;
        if (!ok(hipMalloc(&nd, sizeof(float) * nc))) return false;
        if (used && d) hipMemcpy(nd, d, sizeof(float) * used, hipMemcpyDeviceToDevice);
        hipFree(d);
        d = nd; cap = nc;
        return true;
    }
};
struct Buffers {
    uint32_t *stateA = 0, *stateB = 0, *halton = 0, *flags = 0, *localCounts = 0, *localSurfKind = 0;
    float *localPhotons = 0, *localSurf = 0, *localRad = 0;
    unsigned long long *stats = 0;
    uint32_t *seg = 0;         // host-built segment tables of a round: 3 words per task for the volume merge, 8 for the surface merge
    float *segNshot = 0;
    DevArr p, wi, alpha;       // merged volume map
    DevArr sp[3], swo[3], salpha[3], rad;   // merged surface stores (kept on request)
    void release(bool keepMaps) {
        hipFree(stateA); hipFree(stateB); hipFree(halton); hipFree(flags); hipFree(localCounts); hipFree(localSurfKind);
        hipFree(localPhotons); hipFree(localSurf); hipFree(localRad); hipFree(stats); hipFree(seg); hipFree(segNshot);
        if (!keepMaps) {
            hipFree(p.d); hipFree(wi.d); hipFree(alpha.d); hipFree(rad.d);
            for (int k = 0; k < 3; ++k) { hipFree(sp[k].d); hipFree(swo[k].d); hipFree(salpha[k].d); }
        }
    }
};
}  // namespace

extern "C" void pvol_free_surface_stores(pvol_ctx *c) {
    for (int k = 0; k < 3; ++k) {
        hipFree(c->surf[k].p); hipFree(c->surf[k].wo); hipFree(c->surf[k].alpha);
        c->surf[k] = pvol_ctx::SurfStore();
    }
    hipFree(c->dRad); c->dRad = 0; c->nRad = 0;
    c->surfKept = false;
}

extern "C" int pvol_preprocess(pvol_ctx *c, uint32_t n_tasks) { return pvol_preprocess_blocks(c, n_tasks, 4096); }

extern "C" int pvol_preprocess_blocks(pvol_ctx *c, uint32_t n_tasks, uint32_t block_paths) {
    if (!c || n_tasks == 0 || n_tasks > 65536 || block_paths == 0 || block_paths > 4096) return PVOL_E_INVALID;
    if (!c->haveScene) return PVOL_E_NO_SCENE;
    std::lock_guard<std::recursive_mutex> api(c->apiMu);
    if (!ok(hipSetDevice(c->params.device))) return PVOL_E_NO_DEVICE;
    hipDeviceSynchronize();
    pvol_free_photons(c);
    pvol_free_surface_stores(c);
    DevScene &h = c->hs;
    h.nPhotons = 0; h.cellStart = 0; h.subStart = 0; h.pos4 = 0; h.alpha4 = 0; h.wi4 = 0;
    memset(c->shootStats, 0, sizeof(c->shootStats));
    if (c->hs.nLights == 0) return pvol_push_scene(c);   // photonshooter.cpp:459
    int rc = pvol_push_scene(c);
    if (rc != PVOL_OK) return rc;

    const auto tShoot0 = std::chrono::steady_clock::now();
    c->prepSeconds[0] = c->prepSeconds[1] = 0.0;
    const uint32_t T = n_tasks;
    const uint32_t blockSize = block_paths;
    const uint32_t giveUpShot = 4096;   // the reference's constant in its give-up test (photonshooter.cpp:283-290), whatever the block
    const size_t SW = pvol_shoot_state_words();
    const bool keep = c->params.keep_surface_photons != 0;
    // Room for one block of one task.  Spectral splitting stores up to ~3 photons per path but the usual yield is
    // ~10 photons per 4096-path block, so the pools start small (T x 256 x 144 B) and a round in which some task outgrew
    // one is REDONE with a larger pool from the saved RNG states (the round is a pure function of them): nothing is dropped
    // and nothing is sized for the worst case.
    uint32_t cap = 256, capS = keep ? 256 : 1, capR = keep ? 64 : 1;
    const uint32_t capMax = (uint32_t)std::min<size_t>(65536, std::max<size_t>(256, ((size_t)48 << 30) / ((size_t)T * 144)));
    Buffers B;
    bool good = ok(hipMalloc(&B.stateA, sizeof(uint32_t) * SW * (size_t)T)) && ok(hipMalloc(&B.stateB, sizeof(uint32_t) * SW * (size_t)T)) &&
                ok(hipMalloc(&B.halton, sizeof(uint32_t) * 48 * (size_t)T)) && ok(hipMalloc(&B.flags, sizeof(uint32_t) * T)) &&
                ok(hipMalloc(&B.localCounts, sizeof(uint32_t) * 8 * (size_t)T)) &&
                ok(hipMalloc(&B.localPhotons, sizeof(float) * 36 * (size_t)cap * T)) && ok(hipMalloc(&B.stats, sizeof(unsigned long long) * 8)) &&
                ok(hipMalloc(&B.localSurf, sizeof(float) * 36 * (size_t)capS * T)) && ok(hipMalloc(&B.localSurfKind, sizeof(uint32_t) * (size_t)capS * T)) &&
                ok(hipMalloc(&B.localRad, sizeof(float) * 8 * (size_t)capR * T)) &&
                ok(hipMalloc(&B.seg, sizeof(uint32_t) * 11 * (size_t)T)) && ok(hipMalloc(&B.segNshot, sizeof(float) * T)) &&
                ok(hipMemset(B.stats, 0, sizeof(unsigned long long) * 8));
    if (!good) { B.release(false); return PVOL_E_NO_MEMORY; }

    ShootArgs A;
    A.scene = c->ds; A.shoot = c->dsh; A.nTasks = T; A.stateIn = B.stateA; A.stateOut = B.stateA; A.halton = B.halton; A.flags = B.flags;
    A.localPhotons = B.localPhotons; A.localCounts = B.localCounts; A.cap = cap; A.stats = B.stats; A.init = 1;
    A.localSurf = B.localSurf; A.localSurfKind = B.localSurfKind; A.capS = capS; A.localRad = B.localRad; A.capR = capR; A.keepSurface = keep ? 1 : 0;
    A.gridVolume = c->hs.volKind == PVOL_VOLUME_GRID ? 1 : 0;
    A.blockPaths = blockSize;
    if (!ok(pvol_launch_shoot(&A, 0)) || !ok(hipDeviceSynchronize())) { B.release(false); return PVOL_E_NO_DEVICE; }
    A.init = 0;
    A.stateOut = B.stateB;

    const pvol_params &P = c->params;
    std::vector<uint32_t> flags(T), counts(8 * (size_t)T);
    // segment tables of a round: volume {task, count, offset} and surface {task, nSurf, take, off[4], nRad}
    std::vector<uint32_t> vTask, vCount, vOff, sTask, sN, sTake, sOff, sRad;
    std::vector<float> vNshot;
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    flags.assign(T, (P.n_caustic_photons == 0 ? 1u : 0u) | (P.n_indirect_photons == 0 ? 2u : 0u) | (P.n_volume_photons == 0 ? 4u : 0u));
    uint32_t nshot = 0;
    uint64_t nCaustic = 0, nIndirect = 0, nDirect = 0, nRadTotal = 0;
    uint32_t nCausticPaths = 0, nIndirectPaths = 0, nDirectPaths = 0;
    size_t nVolume = 0;
    bool abortTasks = false;
    uint32_t stallRounds = 0;
    rc = PVOL_OK;
    auto unsuccessful = [](uint32_t needed, uint64_t found, uint32_t shot) { return (found < needed && (found == 0 || found < shot / 1024)); };   // photonshooter.cpp:37-39
    for (;;) {
        bool anyLive = false;
        for (uint32_t t = 0; t < T; ++t) anyLive = anyLive || !(flags[t] & 8u);
        if (!anyLive) break;
        if (!ok(hipMemcpy(B.flags, flags.data(), sizeof(uint32_t) * T, hipMemcpyHostToDevice))) { rc = PVOL_E_NO_DEVICE; break; }
        bool redo = false;
        do {   // one 4096-path block per live task; redone from the same states if a task's block outgrew a buffer
            redo = false;
            unsigned long long rs[8];
            if (!ok(hipMemset(B.stats, 0, sizeof(rs))) || !ok(pvol_launch_shoot(&A, 0)) ||
                !ok(hipMemcpy(counts.data(), B.localCounts, sizeof(uint32_t) * 8 * (size_t)T, hipMemcpyDeviceToHost)) ||
                !ok(hipMemcpy(rs, B.stats, sizeof(rs), hipMemcpyDeviceToHost))) { rc = PVOL_E_NO_DEVICE; break; }
            uint32_t most = 0, mostS = 0, mostR = 0;
            for (uint32_t t = 0; t < T; ++t) {
                if (flags[t] & 8u) continue;
                if (!(flags[t] & 4u)) most = std::max(most, counts[8 * (size_t)t]);
                mostS = std::max(mostS, counts[8 * (size_t)t + 4]);
                mostR = std::max(mostR, counts[8 * (size_t)t + 5]);
            }
            if (most > cap) {
                if (most > capMax) { rc = PVOL_E_LIMIT; break; }
                cap = std::min<uint32_t>(capMax, std::max<uint32_t>(most + most / 4, cap * 4));
                hipFree(B.localPhotons); B.localPhotons = 0;
                if (!ok(hipMalloc(&B.localPhotons, sizeof(float) * 36 * (size_t)cap * T))) { rc = PVOL_E_NO_MEMORY; break; }
                A.localPhotons = B.localPhotons; A.cap = cap;
                redo = true;
            }
            if (keep && mostS > capS) {
                if (mostS > capMax) { rc = PVOL_E_LIMIT; break; }
                capS = std::min<uint32_t>(capMax, std::max<uint32_t>(mostS + mostS / 4, capS * 4));
                hipFree(B.localSurf); hipFree(B.localSurfKind); B.localSurf = 0; B.localSurfKind = 0;
                if (!ok(hipMalloc(&B.localSurf, sizeof(float) * 36 * (size_t)capS * T)) || !ok(hipMalloc(&B.localSurfKind, sizeof(uint32_t) * (size_t)capS * T))) { rc = PVOL_E_NO_MEMORY; break; }
                A.localSurf = B.localSurf; A.localSurfKind = B.localSurfKind; A.capS = capS;
                redo = true;
            }
            if (keep && mostR > capR) {
                capR = std::max<uint32_t>(mostR + mostR / 4, capR * 4);
                hipFree(B.localRad); B.localRad = 0;
                if (!ok(hipMalloc(&B.localRad, sizeof(float) * 8 * (size_t)capR * T))) { rc = PVOL_E_NO_MEMORY; break; }
                A.localRad = B.localRad; A.capR = capR;
                redo = true;
            }
            if (!redo) for (int i = 0; i < 8; ++i) st[i] += rs[i];
        } while (redo);
        if (rc != PVOL_OK) break;
        { const uint32_t *tmp = A.stateIn; A.stateIn = A.stateOut; A.stateOut = const_cast<uint32_t *>(tmp); }   // the round stands
        // merge in task order (photonshooter.cpp:280-351)
        vTask.clear(); vCount.clear(); vOff.clear(); vNshot.clear();
        sTask.clear(); sN.clear(); sTake.clear(); sOff.clear(); sRad.clear();
        const size_t volBefore = nVolume;
        const uint64_t surfBefore[4] = {nCaustic, nDirect, nIndirect, nRadTotal};
        for (uint32_t t = 0; t < T; ++t) {
            uint32_t &fl = flags[t];
            if (fl & 8u) continue;
            if (abortTasks) { fl |= 8u; continue; }
            if (nshot > 500000 && (unsuccessful(P.n_caustic_photons, nCaustic, giveUpShot) || unsuccessful(P.n_indirect_photons, nIndirect, giveUpShot) ||
                                   unsuccessful(P.n_volume_photons, nVolume, giveUpShot))) {
                nVolume = 0; nCaustic = nIndirect = 0; nRadTotal = 0;   // photonshooter.cpp:292-298 erases caustic, indirect, volume, radiance
                vTask.clear(); vCount.clear(); vOff.clear(); vNshot.clear();
                sTask.clear(); sN.clear(); sTake.clear(); sOff.clear(); sRad.clear();
                abortTasks = true;
                fl |= 8u;
                rc = PVOL_E_SHOOT_FAILED;
                continue;
            }
            nshot += blockSize;
            const uint32_t *lc = &counts[8 * (size_t)t];
            uint32_t take = 0;
            const uint32_t off[4] = {(uint32_t)nCaustic, (uint32_t)nDirect, (uint32_t)nIndirect, (uint32_t)nRadTotal};
            if (!(fl & 2u)) {
                take |= 2u | 4u;
                nIndirectPaths += blockSize; nDirectPaths += blockSize;
                nIndirect += lc[3];
                if (nIndirect >= P.n_indirect_photons) fl |= 2u;
                nDirect += lc[2];
            }
            if (!(fl & 1u)) {
                take |= 1u;
                nCausticPaths += blockSize;
                nCaustic += lc[1];
                if (nCaustic >= P.n_caustic_photons) fl |= 1u;
            }
            if (keep && (lc[4] || lc[5])) {
                sTask.push_back(t); sN.push_back(lc[4]); sTake.push_back(take); sRad.push_back(lc[5]);
                sOff.push_back(off[0]); sOff.push_back(off[1]); sOff.push_back(off[2]); sOff.push_back(off[3]);
            }
            nRadTotal += keep ? lc[5] : 0;
            if (!(fl & 4u)) {
                if (lc[0]) {
                    vTask.push_back(t); vCount.push_back(lc[0]); vOff.push_back((uint32_t)nVolume); vNshot.push_back(float(nshot));
                    nVolume += lc[0];
                }
                if (nVolume >= P.n_volume_photons) fl |= 4u;
            }
            if ((fl & 7u) == 7u) fl |= 8u;
        }
        if (!vTask.empty() && !abortTasks) {
            if (!B.p.need(3 * volBefore, 3 * nVolume) || !B.wi.need(3 * volBefore, 3 * nVolume) || !B.alpha.need(30 * volBefore, 30 * nVolume)) { rc = PVOL_E_NO_MEMORY; break; }
            MergeArgs M;
            const size_t n = vTask.size();
            M.localPhotons = B.localPhotons; M.cap = cap; M.srcTask = B.seg; M.count = B.seg + T; M.dstOff = B.seg + 2 * (size_t)T; M.nshot = B.segNshot;
            M.nSeg = (uint32_t)n; M.p = B.p.d; M.wi = B.wi.d; M.alpha = B.alpha.d;
            bool g2 = ok(hipMemcpy(B.seg, vTask.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice)) &&
                      ok(hipMemcpy(B.seg + T, vCount.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice)) &&
                      ok(hipMemcpy(B.seg + 2 * (size_t)T, vOff.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice)) &&
                      ok(hipMemcpy(B.segNshot, vNshot.data(), sizeof(float) * n, hipMemcpyHostToDevice)) &&
                      ok(pvol_launch_merge(&M, 0)) && ok(hipDeviceSynchronize());
            if (!g2) { rc = PVOL_E_NO_DEVICE; break; }
        }
        if (keep && !sTask.empty() && !abortTasks) {
            const uint64_t after[3] = {nCaustic, nDirect, nIndirect};
            bool g3 = true;
            for (int k = 0; k < 3 && g3; ++k)
                g3 = B.sp[k].need(3 * surfBefore[k], 3 * after[k]) && B.swo[k].need(3 * surfBefore[k], 3 * after[k]) && B.salpha[k].need(30 * surfBefore[k], 30 * after[k]);
            g3 = g3 && B.rad.need(8 * surfBefore[3], 8 * nRadTotal);
            if (!g3) { rc = PVOL_E_NO_MEMORY; break; }
            SurfMergeArgs M;
            const size_t n = sTask.size();
            uint32_t *d = B.seg + 3 * (size_t)T;
            M.localSurf = B.localSurf; M.localSurfKind = B.localSurfKind; M.capS = capS; M.localRad = B.localRad; M.capR = capR;
            M.srcTask = d; M.nSurf = d + T; M.take = d + 2 * (size_t)T; M.nRad = d + 3 * (size_t)T; M.dstOff = d + 4 * (size_t)T;
            M.nSeg = (uint32_t)n;
            for (int k = 0; k < 3; ++k) { M.p[k] = B.sp[k].d; M.wo[k] = B.swo[k].d; M.alpha[k] = B.salpha[k].d; }
            M.rad = B.rad.d;
            g3 = ok(hipMemcpy(d, sTask.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice)) &&
                 ok(hipMemcpy(d + T, sN.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice)) &&
                 ok(hipMemcpy(d + 2 * (size_t)T, sTake.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice)) &&
                 ok(hipMemcpy(d + 3 * (size_t)T, sRad.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice)) &&
                 ok(hipMemcpy(d + 4 * (size_t)T, sOff.data(), sizeof(uint32_t) * 4 * n, hipMemcpyHostToDevice)) &&
                 ok(pvol_launch_merge_surface(&M, 0)) && ok(hipDeviceSynchronize());
            if (!g3) { rc = PVOL_E_NO_DEVICE; break; }
        }
        if (rc == PVOL_E_LIMIT || rc == PVOL_E_NO_DEVICE) break;
        // The reference has no exit for a store that stops growing after a good start (its `unsuccessful` test, photonshooter.cpp:37-39,
        // passes once found >= 4): e.g. a matte scene whose "caustic" photons all come through the medium, after the volume map is
        // full -- it would shoot forever.  Here 256 rounds in a row without a single photon for any store still wanted end the
        // pass the way the reference's own abort does (stores erased, PVOL_E_SHOOT_FAILED).
        if (!abortTasks) {
            const bool progress = nCaustic != surfBefore[0] || nIndirect != surfBefore[2] || nVolume != volBefore;
            stallRounds = progress ? 0u : stallRounds + 1u;
            if (stallRounds >= 256u) {
                nVolume = 0; nCaustic = nIndirect = 0; nRadTotal = 0;
                for (uint32_t t = 0; t < T; ++t) flags[t] |= 8u;
                abortTasks = true;
                rc = PVOL_E_SHOOT_FAILED;
            }
        }
    }
    // paths, follow_calls, no_hit, march_steps, interactions, absorbed, stored_volume, caustic, direct, indirect, split_children, nshot
    c->shootStats[0] = st[0]; c->shootStats[1] = st[1]; c->shootStats[2] = st[2]; c->shootStats[3] = st[3]; c->shootStats[4] = st[4];
    c->shootStats[5] = st[5]; c->shootStats[6] = nVolume; c->shootStats[7] = nCaustic; c->shootStats[8] = nDirect; c->shootStats[9] = nIndirect;
    c->shootStats[10] = st[6]; c->shootStats[11] = nshot;
    c->prepSeconds[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - tShoot0).count();
    if (rc == PVOL_OK && st[7] != 0) rc = PVOL_E_LIMIT;   // a frame stack overflowed: never silently drop photons
    if (rc == PVOL_OK && keep) {   // the surface stores go to the context whatever happens to the volume map
        const uint64_t cnt[3] = {nCaustic, nDirect, nIndirect};
        const uint32_t paths[3] = {nCausticPaths, nDirectPaths, nIndirectPaths};
        for (int k = 0; k < 3; ++k) {
            c->surf[k].p = B.sp[k].d; c->surf[k].wo = B.swo[k].d; c->surf[k].alpha = B.salpha[k].d;
            c->surf[k].n = (uint32_t)cnt[k]; c->surf[k].nPaths = paths[k];
            B.sp[k].d = B.swo[k].d = B.salpha[k].d = 0;
        }
        c->dRad = B.rad.d; c->nRad = (uint32_t)nRadTotal;
        B.rad.d = 0;
        c->surfKept = true;
    }
    if (rc != PVOL_OK || nVolume == 0) {
        B.release(false);
        return rc;
    }
    // hand the merged arrays to the context and build the search structure
    std::vector<float> hostP(3 * nVolume);
    if (!ok(hipMemcpy(hostP.data(), B.p.d, sizeof(float) * 3 * nVolume, hipMemcpyDeviceToHost))) { B.release(false); return PVOL_E_NO_DEVICE; }
    c->dRawP = B.p.d; c->dRawWi = B.wi.d; c->dRawAlpha = B.alpha.d;
    B.p.d = B.wi.d = B.alpha.d = 0;
    B.release(false);
    const auto tBuild0 = std::chrono::steady_clock::now();
    rc = pvol_finish_map(c, (uint32_t)nVolume, hostP.data());
    hipDeviceSynchronize();
    c->prepSeconds[1] = std::chrono::duration<double>(std::chrono::steady_clock::now() - tBuild0).count();
    return rc;
}

extern "C" int pvol_surface_photon_count(pvol_ctx *c, int kind, uint32_t *n, uint32_t *nPaths) {
    if (!c || kind < 0 || kind > 2 || !n) return PVOL_E_INVALID;
    *n = c->surf[kind].n;
    if (nPaths) *nPaths = c->surf[kind].nPaths;
    return PVOL_OK;
}
extern "C" int pvol_download_surface_photons(pvol_ctx *c, int kind, float *p, float *wo, float *alpha, uint32_t capacity) {
    if (!c || kind < 0 || kind > 2 || !p || !wo || !alpha) return PVOL_E_INVALID;
    const uint32_t n = std::min(capacity, c->surf[kind].n);
    if (!n) return PVOL_OK;
    if (!ok(hipSetDevice(c->params.device))) return PVOL_E_NO_DEVICE;
    const bool good = ok(hipMemcpy(p, c->surf[kind].p, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost)) &&
                      ok(hipMemcpy(wo, c->surf[kind].wo, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost)) &&
                      ok(hipMemcpy(alpha, c->surf[kind].alpha, sizeof(float) * 30 * (size_t)n, hipMemcpyDeviceToHost));
    return good ? PVOL_OK : PVOL_E_NO_DEVICE;
}
extern "C" int pvol_radiance_photon_count(pvol_ctx *c, uint32_t *n) {
    if (!c || !n) return PVOL_E_INVALID;
    *n = c->nRad;
    return PVOL_OK;
}
extern "C" int pvol_download_radiance_photons(pvol_ctx *c, float *p, float *nrm, float *rhoR, float *rhoT, uint32_t capacity) {
    if (!c || !p || !nrm || !rhoR || !rhoT) return PVOL_E_INVALID;
    const uint32_t n = std::min(capacity, c->nRad);
    if (!n) return PVOL_OK;
    if (!ok(hipSetDevice(c->params.device))) return PVOL_E_NO_DEVICE;
    std::vector<float> rec(8 * (size_t)n);
    if (!ok(hipMemcpy(rec.data(), c->dRad, sizeof(float) * 8 * (size_t)n, hipMemcpyDeviceToHost))) return PVOL_E_NO_DEVICE;
    for (uint32_t i = 0; i < n; ++i) {
        memcpy(p + 3 * (size_t)i, &rec[8 * (size_t)i], 12);
        memcpy(nrm + 3 * (size_t)i, &rec[8 * (size_t)i + 3], 12);
        int mi;
        memcpy(&mi, &rec[8 * (size_t)i + 6], 4);
        // rho_r / rho_t of the surface's BSDF (photonshooter.cpp:185-188): the only non-specular BxDF on this path is the
        // Lambertian, whose rho() is its reflectance whatever the samples (core/reflection.h:222-223); no transmissive one
        const DevMaterial &m = c->hsh.mats[(mi >= 0 && mi < c->hsh.nMats) ? mi : 0];
        for (int b = 0; b < 30; ++b) { rhoR[30 * (size_t)i + b] = m.kind == PVOL_MATERIAL_MATTE ? m.kd[b] : 0.f; rhoT[30 * (size_t)i + b] = 0.f; }
    }
    return PVOL_OK;
}

extern "C" int pvol_get_preprocess_seconds(pvol_ctx *c, double *out2) {
    if (!c || !out2) return PVOL_E_INVALID;
    out2[0] = c->prepSeconds[0]; out2[1] = c->prepSeconds[1];
    return PVOL_OK;
}

extern "C" int pvol_get_shoot_stats(pvol_ctx *c, uint64_t *out12) {
    if (!c || !out12) return PVOL_E_INVALID;
    memcpy(out12, c->shootStats, sizeof(c->shootStats));
    return PVOL_OK;
}
