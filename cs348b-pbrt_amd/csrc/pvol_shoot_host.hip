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
    unsigned long long *stats;
    int init;
};
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
struct Buffers {
    uint32_t *stateA = 0, *stateB = 0, *halton = 0, *flags = 0, *localCounts = 0;
    float *localPhotons = 0;
    unsigned long long *stats = 0;
    uint32_t *segTask = 0, *segCount = 0, *segOff = 0;
    float *segNshot = 0;
    float *p = 0, *wi = 0, *alpha = 0;   // merged map (capacity photons)
    size_t capacity = 0;
    void release(bool keepMap) {
        hipFree(stateA); hipFree(stateB); hipFree(halton); hipFree(flags); hipFree(localCounts); hipFree(localPhotons); hipFree(stats);
        hipFree(segTask); hipFree(segCount); hipFree(segOff); hipFree(segNshot);
        if (!keepMap) { hipFree(p); hipFree(wi); hipFree(alpha); }
    }
};

bool grow_map(Buffers &B, size_t used, size_t need) {
    if (need <= B.capacity) return true;
    size_t cap = std::max(need, B.capacity * 2 + 1024);
    float *np = 0, *nw = 0, *na = 0;
    if (!ok(hipMalloc(&np, sizeof(float) * 3 * cap)) || !ok(hipMalloc(&nw, sizeof(float) * 3 * cap)) || !ok(hipMalloc(&na, sizeof(float) * 30 * cap))) {
        hipFree(np); hipFree(nw); hipFree(na);
        return false;
    }
    if (used) {
        hipMemcpy(np, B.p, sizeof(float) * 3 * used, hipMemcpyDeviceToDevice);
        hipMemcpy(nw, B.wi, sizeof(float) * 3 * used, hipMemcpyDeviceToDevice);
        hipMemcpy(na, B.alpha, sizeof(float) * 30 * used, hipMemcpyDeviceToDevice);
    }
    hipFree(B.p); hipFree(B.wi); hipFree(B.alpha);
    B.p = np; B.wi = nw; B.alpha = na; B.capacity = cap;
    return true;
}
}  // namespace

extern "C" int pvol_preprocess(pvol_ctx *c, uint32_t n_tasks) {
    if (!c || n_tasks == 0 || n_tasks > 65536) return PVOL_E_INVALID;
    if (!c->haveScene) return PVOL_E_NO_SCENE;
    std::lock_guard<std::recursive_mutex> api(c->apiMu);
    if (!ok(hipSetDevice(c->params.device))) return PVOL_E_NO_DEVICE;
    hipDeviceSynchronize();
    pvol_free_photons(c);
    DevScene &h = c->hs;
    h.nPhotons = 0; h.cellStart = 0; h.pos4 = 0; h.alpha4 = 0; h.wi4 = 0;
    memset(c->shootStats, 0, sizeof(c->shootStats));
    if (c->hs.nLights == 0) return pvol_push_scene(c);   // photonshooter.cpp:459
    int rc = pvol_push_scene(c);
    if (rc != PVOL_OK) return rc;

    const auto tShoot0 = std::chrono::steady_clock::now();
    c->prepSeconds[0] = c->prepSeconds[1] = 0.0;
    const uint32_t T = n_tasks;
    const uint32_t blockSize = 4096;
    const size_t SW = pvol_shoot_state_words();
    // Room for one block of one task.  Spectral splitting stores up to ~3 photons per path but the usual yield is
    // ~10 photons per 4096-path block, so the pool starts small (T x 256 x 144 B) and a round in which some task outgrew
    // it is REDONE with a larger one from the saved RNG states (the round is a pure function of them): nothing is dropped
    // and nothing is sized for the worst case.
    uint32_t cap = 256;
    const uint32_t capMax = (uint32_t)std::min<size_t>(16384, std::max<size_t>(256, ((size_t)48 << 30) / ((size_t)T * 144)));
    Buffers B;
    bool good = ok(hipMalloc(&B.stateA, sizeof(uint32_t) * SW * (size_t)T)) && ok(hipMalloc(&B.stateB, sizeof(uint32_t) * SW * (size_t)T)) &&
                ok(hipMalloc(&B.halton, sizeof(uint32_t) * 48 * (size_t)T)) && ok(hipMalloc(&B.flags, sizeof(uint32_t) * T)) &&
                ok(hipMalloc(&B.localCounts, sizeof(uint32_t) * 4 * (size_t)T)) &&
                ok(hipMalloc(&B.localPhotons, sizeof(float) * 36 * (size_t)cap * T)) && ok(hipMalloc(&B.stats, sizeof(unsigned long long) * 8)) &&
                ok(hipMalloc(&B.segTask, sizeof(uint32_t) * T)) && ok(hipMalloc(&B.segCount, sizeof(uint32_t) * T)) &&
                ok(hipMalloc(&B.segOff, sizeof(uint32_t) * T)) && ok(hipMalloc(&B.segNshot, sizeof(float) * T)) &&
                ok(hipMemset(B.stats, 0, sizeof(unsigned long long) * 8));
    if (!good) { B.release(false); return PVOL_E_NO_MEMORY; }

    ShootArgs A;
    A.scene = c->ds; A.shoot = c->dsh; A.nTasks = T; A.stateIn = B.stateA; A.stateOut = B.stateA; A.halton = B.halton; A.flags = B.flags;
    A.localPhotons = B.localPhotons; A.localCounts = B.localCounts; A.cap = cap; A.stats = B.stats; A.init = 1;
    if (!ok(pvol_launch_shoot(&A, 0)) || !ok(hipDeviceSynchronize())) { B.release(false); return PVOL_E_NO_DEVICE; }
    A.init = 0;
    A.stateOut = B.stateB;

    const pvol_params &P = c->params;
    std::vector<uint32_t> flags(T), counts(4 * (size_t)T), segTask, segCount, segOff;
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<float> segNshot;
    flags.assign(T, (P.n_caustic_photons == 0 ? 1u : 0u) | (P.n_indirect_photons == 0 ? 2u : 0u) | (P.n_volume_photons == 0 ? 4u : 0u));
    uint32_t nshot = 0;
    uint64_t nCaustic = 0, nIndirect = 0, nDirect = 0;
    size_t nVolume = 0;
    bool abortTasks = false;
    rc = PVOL_OK;
    auto unsuccessful = [](uint32_t needed, uint64_t found, uint32_t shot) { return (found < needed && (found == 0 || found < shot / 1024)); };   // photonshooter.cpp:37-39
    for (;;) {
        bool anyLive = false;
        for (uint32_t t = 0; t < T; ++t) anyLive = anyLive || !(flags[t] & 8u);
        if (!anyLive) break;
        if (!ok(hipMemcpy(B.flags, flags.data(), sizeof(uint32_t) * T, hipMemcpyHostToDevice))) { rc = PVOL_E_NO_DEVICE; break; }
        bool redo = false;
        do {   // one 4096-path block per live task; redone from the same states if a task's block outgrew the buffer
            redo = false;
            unsigned long long rs[8];
            if (!ok(hipMemset(B.stats, 0, sizeof(rs))) || !ok(pvol_launch_shoot(&A, 0)) ||
                !ok(hipMemcpy(counts.data(), B.localCounts, sizeof(uint32_t) * 4 * (size_t)T, hipMemcpyDeviceToHost)) ||
                !ok(hipMemcpy(rs, B.stats, sizeof(rs), hipMemcpyDeviceToHost))) { rc = PVOL_E_NO_DEVICE; break; }
            uint32_t most = 0;
            for (uint32_t t = 0; t < T; ++t) if (!(flags[t] & (8u | 4u))) most = std::max(most, counts[4 * (size_t)t]);
            if (most > cap) {
                if (most > capMax) { rc = PVOL_E_LIMIT; break; }
                cap = std::min<uint32_t>(capMax, std::max<uint32_t>(most + most / 4, cap * 4));
                hipFree(B.localPhotons); B.localPhotons = 0;
                if (!ok(hipMalloc(&B.localPhotons, sizeof(float) * 36 * (size_t)cap * T))) { rc = PVOL_E_NO_MEMORY; break; }
                A.localPhotons = B.localPhotons; A.cap = cap;
                redo = true;
                continue;
            }
            for (int i = 0; i < 8; ++i) st[i] += rs[i];
        } while (redo);
        if (rc != PVOL_OK) break;
        { const uint32_t *tmp = A.stateIn; A.stateIn = A.stateOut; A.stateOut = const_cast<uint32_t *>(tmp); }   // the round stands
        // merge in task order (photonshooter.cpp:280-351)
        segTask.clear(); segCount.clear(); segOff.clear(); segNshot.clear();
        for (uint32_t t = 0; t < T; ++t) {
            uint32_t &fl = flags[t];
            if (fl & 8u) continue;
            if (abortTasks) { fl |= 8u; continue; }
            if (nshot > 500000 && (unsuccessful(P.n_caustic_photons, nCaustic, blockSize) || unsuccessful(P.n_indirect_photons, nIndirect, blockSize) ||
                                   unsuccessful(P.n_volume_photons, nVolume, blockSize))) {
                nVolume = 0; nCaustic = nIndirect = 0;
                segTask.clear(); segCount.clear(); segOff.clear(); segNshot.clear();
                abortTasks = true;
                fl |= 8u;
                rc = PVOL_E_SHOOT_FAILED;
                continue;
            }
            nshot += blockSize;
            const uint32_t *lc = &counts[4 * (size_t)t];
            if (!(fl & 2u)) {
                nIndirect += lc[3];
                if (nIndirect >= P.n_indirect_photons) fl |= 2u;
                nDirect += lc[2];
            }
            if (!(fl & 1u)) {
                nCaustic += lc[1];
                if (nCaustic >= P.n_caustic_photons) fl |= 1u;
            }
            if (!(fl & 4u)) {
                if (lc[0]) {
                    segTask.push_back(t); segCount.push_back(lc[0]); segOff.push_back((uint32_t)nVolume); segNshot.push_back(float(nshot));
                    nVolume += lc[0];
                }
                if (nVolume >= P.n_volume_photons) fl |= 4u;
            }
            if ((fl & 7u) == 7u) fl |= 8u;
        }
        if (!segTask.empty() && !abortTasks) {
            if (!grow_map(B, segOff[0], nVolume)) { rc = PVOL_E_NO_MEMORY; break; }
            MergeArgs M;
            M.localPhotons = B.localPhotons; M.cap = cap; M.srcTask = B.segTask; M.count = B.segCount; M.dstOff = B.segOff; M.nshot = B.segNshot;
            M.nSeg = (uint32_t)segTask.size(); M.p = B.p; M.wi = B.wi; M.alpha = B.alpha;
            bool g2 = ok(hipMemcpy(B.segTask, segTask.data(), sizeof(uint32_t) * segTask.size(), hipMemcpyHostToDevice)) &&
                      ok(hipMemcpy(B.segCount, segCount.data(), sizeof(uint32_t) * segCount.size(), hipMemcpyHostToDevice)) &&
                      ok(hipMemcpy(B.segOff, segOff.data(), sizeof(uint32_t) * segOff.size(), hipMemcpyHostToDevice)) &&
                      ok(hipMemcpy(B.segNshot, segNshot.data(), sizeof(float) * segNshot.size(), hipMemcpyHostToDevice)) &&
                      ok(pvol_launch_merge(&M, 0)) && ok(hipDeviceSynchronize());
            if (!g2) { rc = PVOL_E_NO_DEVICE; break; }
        }
        if (rc == PVOL_E_LIMIT || rc == PVOL_E_NO_DEVICE) break;
    }
    // paths, follow_calls, no_hit, march_steps, interactions, absorbed, stored_volume, caustic, direct, indirect, split_children, nshot
    c->shootStats[0] = st[0]; c->shootStats[1] = st[1]; c->shootStats[2] = st[2]; c->shootStats[3] = st[3]; c->shootStats[4] = st[4];
    c->shootStats[5] = st[5]; c->shootStats[6] = nVolume; c->shootStats[7] = nCaustic; c->shootStats[8] = nDirect; c->shootStats[9] = nIndirect;
    c->shootStats[10] = st[6]; c->shootStats[11] = nshot;
    c->prepSeconds[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - tShoot0).count();
    if (rc == PVOL_OK && st[7] != 0) rc = PVOL_E_LIMIT;   // a frame stack or block buffer overflowed: never silently drop photons
    if (rc != PVOL_OK || nVolume == 0) {
        B.release(false);
        return rc;
    }
    // hand the merged arrays to the context and build the search structure
    std::vector<float> hostP(3 * nVolume);
    if (!ok(hipMemcpy(hostP.data(), B.p, sizeof(float) * 3 * nVolume, hipMemcpyDeviceToHost))) { B.release(false); return PVOL_E_NO_DEVICE; }
    c->dRawP = B.p; c->dRawWi = B.wi; c->dRawAlpha = B.alpha;
    B.release(true);
    const auto tBuild0 = std::chrono::steady_clock::now();
    rc = pvol_finish_map(c, (uint32_t)nVolume, hostP.data());
    hipDeviceSynchronize();
    c->prepSeconds[1] = std::chrono::duration<double>(std::chrono::steady_clock::now() - tBuild0).count();
    return rc;
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
