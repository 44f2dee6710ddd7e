// pvol_host.h -- host-side internals shared by pvol_api.hip and pvol_shoot_host.hip
#ifndef PVOL_HOST_H
#define PVOL_HOST_H
#include <hip/hip_runtime.h>
#include <mutex>
#include <vector>
#include "pvol_dev.h"

// ---- kernels' host entry points (pvol_march.hip, pvol_grid.hip)
#include "pvol_liargs.h"
struct GridBuildArgs {
    const float *p, *wi, *alpha;
    uint32_t n;
    float lo[3];
    float inv;
    int32_t gdim[3];
    int32_t sub;
    int32_t volKind;
    float extLo[3], extHi[3];
    float w2v[16];
};
extern "C" hipError_t pvol_launch_li_seq(const LiArgs *args, size_t ldsBytes, int candCap, bool stats, hipStream_t stream);
extern "C" hipError_t pvol_launch_li_slice(const LiArgs *args, size_t ldsResolve, size_t ldsReplay, int candCap, bool stats,
                                           uint32_t nWaves, hipStream_t stream, bool resolve, int groupForm, size_t ldsGroup, uint32_t nGroupWaves,
                                           uint32_t nFixWaves);
extern "C" size_t pvol_group_lds_bytes(int candCap);
extern "C" hipError_t pvol_launch_li_group(const LiArgs *args, size_t ldsBytes, int candCap, bool stats, uint32_t nWaves, uint32_t nFixWaves,
                                           int replay, hipStream_t stream);
extern "C" hipError_t pvol_launch_spec_compose(const SpecComposeArgs *a, hipStream_t stream);
extern "C" hipError_t pvol_launch_spec_fill(pvol_ray *rays, uint32_t n, hipStream_t stream);
extern "C" hipError_t pvol_launch_li_replay(const LiArgs *args, size_t ldsReplay, int candCap, uint32_t nWaves, hipStream_t stream);
extern "C" hipError_t pvol_launch_surface(const SurfArgs *a, uint32_t nWaves, hipStream_t stream);
extern "C" size_t pvol_tile_lds_bytes(int maxSteps, uint32_t spp, bool fused, int nTris, bool shadowRows);
extern "C" hipError_t pvol_launch_tile(const LiArgs *args, const TileArgs *tile, bool fused, size_t ldsBytes, int candCap, hipStream_t stream, int wavesPerTask);
extern "C" hipError_t pvol_launch_li_par(const LiArgs *args, size_t ldsBytes, int candCap, bool stats, uint32_t nWaves, hipStream_t stream);
extern "C" hipError_t pvol_build_grid(const GridBuildArgs *args, float4 *pos4, float4 *alpha4, float4 *wi4,
                                      uint32_t *cellStart, uint32_t *subStart, hipStream_t stream);
extern "C" hipError_t pvol_build_bvh(const float *dTri, const int32_t *dMat, const int32_t *dFlip, uint32_t n, float pad, float4 *tris,
                                     float4 *nodes, hipStream_t stream);
extern "C" hipError_t pvol_grid_occupancy(const uint32_t *cellStart, uint32_t ncells, double *sumSquares, hipStream_t stream);

#define PVOL_N_PHASES 6
enum { PVOL_PHASE_END = -1, PVOL_PHASE_TILE = 0, PVOL_PHASE_RNG = 1, PVOL_PHASE_MARCH = 2, PVOL_PHASE_SURFACE = 3, PVOL_PHASE_FILM = 4, PVOL_PHASE_OTHER = 5 };

struct pvol_ctx {
    pvol_params params;
    bool haveScene;
    DevScene hs;         // host copy
    DevScene *ds;        // device copy
    float *dDensity;
    // triangle hierarchy of a scene with more than PVOL_MAX_TRIS triangles (pvol_bvh.hip), else 0
    float4 *dBvhNodes = 0, *dBvhTris = 0;
    std::vector<int32_t> triMatHost;   // material of every triangle of the scene (host copy, any size)
    double bvhBuildMs = 0.0;
    // photon map
    uint32_t nPhotons;
    float *dRawP, *dRawWi, *dRawAlpha;  // upload order (kept for pvol_download_photons)
    float4 *dPos4, *dAlpha4, *dWi4;
    uint32_t *dCellStart;
    uint32_t *dSubStart = 0;   // second level of a clumpy map (pvol_grid.hip), else 0
    DevCounters *dCounters;
    uint32_t *dWords;   // [0] chunk counter of the ray-parallel kernels, [1] needSeq flag, [2] length of the deferred-lookup list, [3] chunk counter of a gated backup kernel
    DeferRec *dDefer = 0;   // li_group_kernel's deferred lookups (grown on demand)
    size_t deferCap = 0;
    int nCU;
    float maxDensity = 1.f;   // largest density factor of the medium (1 for analytic volumes, max of the grid values)
    bool noLite = false;      // PVOL_NO_LITE=1: keep the geometry inside the sequential resolve pass (testing)
    const char *lastKernel = "";
    int fixWavesPerCU;   // li_fixup_kernel waves per CU; PVOL_FIX_WAVES overrides
    int groupWavesPerCU; // resident li_group_kernel waves per CU (LDS plan: 8); PVOL_GROUP_WAVES overrides
    bool noGroup;       // PVOL_NO_GROUP=1: keep li_par_kernel (one wave per ray) where li_group_kernel (one ray per lane) would run
    bool forceSeq;      // PVOL_FORCE_SEQ=1: always take the stream-sequential kernel (testing)
    bool statsOn;
    // kernel timing (HIP events on the launch stream)
    std::vector<std::pair<hipEvent_t, hipEvent_t> > pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t> > pool;
    double timeMs;
    uint64_t launches;
    std::mutex mu;       // event lists only
    // phase timing of the render driver (pvol_enable_phase_timing): marks on the launch stream, a mark opens phase `id` and
    // closes the one before it; PVOL_PHASE_END closes without opening
    int tileWaves = 0;       // PVOL_TILE_WAVES: waves per render task of the COUNT-mode tile pre-pass (0 = by tasks per CU)
    bool phaseOn = false;
    std::vector<std::pair<int, hipEvent_t> > phaseMarks;
    std::vector<hipEvent_t> phasePool;
    double phaseMs[PVOL_N_PHASES] = {0, 0, 0, 0, 0, 0};
    // One batch at a time per context: the launches of a batch share dWords / dRecords / dState / dCounters and the
    // deferred-lookup list.  Host entry points hold it from upload to copy-back (VolumeIntegrator::Li is called from every
    // SamplerRendererTask thread at once, samplerrenderer.cpp:247); device entry points hold it while they enqueue.
    std::recursive_mutex apiMu;
    // resolve/replay scratch (grown on demand)
    unsigned char *dRecords = 0;
    size_t recBytes = 0;
    uint32_t *dState = 0;
    size_t stateBytes = 0;
    // photon shooter
    DevShootScene hsh;
    DevShootScene *dsh;
    uint64_t shootStats[12];
    // surface stores of the last pvol_preprocess (kept only with params.keep_surface_photons): kind 0 caustic, 1 direct, 2 indirect
    struct SurfStore { float *p = 0, *wo = 0, *alpha = 0; uint32_t n = 0, nPaths = 0; } surf[3];
    bool surfKept = false;   // the last pvol_preprocess ran with keep_surface_photons and left its stores here
    float *dRad = 0;       // radiance photons: [n][8] = p(3) n(3) material index, pad
    uint32_t nRad = 0;
    // caustic map of the surface integrator (pvol_set_surface_integrator), same cell layout as the volume map
    float4 *dCPos4 = 0, *dCAlpha4 = 0, *dCWi4 = 0;
    uint32_t *dCCellStart = 0;
    float *dTau = 0;       // per sample of a render batch: optical length the surface term is attenuated over
    size_t tauBytes = 0;
    float *dTauNext = 0;   // set by the render driver around pvol_launch_batch when the surface integrator is on
    // specular recursion of the surface integrator (pvol_spec_dev.h): segments of the camera samples that meet glass
    bool specOn = false;            // the scene holds a specular material and the surface integrator is on
    pvol_ray *dSegRays = 0;
    SegInfo *dSegInfo = 0;
    float *dSegOut = 0;             // 60 floats per segment
    unsigned char *dSegRecords = 0; // per-step records of the segments (scenes where drawn values matter)
    uint32_t *dSegCounter = 0;
    pvol_stream *dSegStream = 0;    // the pool seen as one stream of a ray batch
    pvol_stream hSegStream;
    size_t segCap = 0, segRecBytes = 0;
    uint32_t *dSpecLink = 0;        // per primary ray of a render batch
    size_t specLinkBytes = 0;
    float *specSurfOut = 0;         // set by the render driver around pvol_launch_batch: where the composition reports the surface term
    double prepSeconds[2] = {0.0, 0.0};   // last pvol_preprocess: shooting (all rounds + merges), search-structure build
    // tile driver work buffers (grown on demand, pvol_tile.hip)
    void *dTile[6] = {0, 0, 0, 0, 0, 0};
    size_t tileBytes[6] = {0, 0, 0, 0, 0, 0};
};


struct pvol_ctx;
extern "C" int pvol_launch_batch(pvol_ctx *c, const pvol_ray *dRays, uint32_t nRays, pvol_stream *dStreams, uint32_t nStreams, int outputKind,
                      float *dOut, uint32_t *dDraws, const uint32_t *dInit, uint32_t *dFinal, int transOnly, uint32_t maxRaysPerStream,
                      const TileArgs *tile, hipStream_t stream);

extern "C" {
void pvol_phase_mark(pvol_ctx *c, hipStream_t stream, int id);
// finish a photon map whose raw arrays (dRawP/dRawWi/dRawAlpha, n photons) are already on the device
int pvol_finish_map(pvol_ctx *c, uint32_t n, const float *hostPositions);
void pvol_free_photons(pvol_ctx *c);
void pvol_free_surface_stores(pvol_ctx *c);
void pvol_free_caustic_map(pvol_ctx *c);
int pvol_push_scene(pvol_ctx *c);
}
#endif
