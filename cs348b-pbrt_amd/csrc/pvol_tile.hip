// pvol_tile.hip -- tile driver, SURVEY 8(f)-1: host side of pvol_render_tasks_device, the image film kernels
// (ImageFilm::AddSample / WriteRGB, film/image.cpp:78-137,178-214) and the small host restatements the
// boundary needs (Gaussian filter table, Sampler::ComputeSubWindow).  The per-task sampler/camera kernel is
// pvol_tile_dev.h (compiled with the march kernels).
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and enums only: the library is bound at run time (pvol_render_frame_ranks)

#include "pvol_host.h"
#include "pvol_math.h"

static inline bool ok(hipError_t e) { return e == hipSuccess; }

// ------------------------------------------------------------------------------------------ host restatements
extern "C" void pvol_gaussian_filter_table(float xw, float yw, float alpha, float *table) {
    // filters/gaussian.h:44-58 (expX, expY, Gaussian()), film/image.cpp:57-68 (table at cell centres)
    const float expX = expf(-alpha * xw * xw), expY = expf(-alpha * yw * yw);
    float *ftp = table;
    for (int y = 0; y < PVOL_FILTER_TABLE_SIZE; ++y) {
        float fy = ((float)y + .5f) * yw / PVOL_FILTER_TABLE_SIZE;
        for (int x = 0; x < PVOL_FILTER_TABLE_SIZE; ++x) {
            float fx = ((float)x + .5f) * xw / PVOL_FILTER_TABLE_SIZE;
            float gx = std::max(0.f, float(expf(-alpha * fx * fx) - expX));
            float gy = std::max(0.f, float(expf(-alpha * fy * fy) - expY));
            *ftp++ = gx * gy;
        }
    }
}

static inline float lerp_host(float t, float a, float b) { return (1.f - t) * a + t * b; }

extern "C" void pvol_compute_sub_window(const pvol_sampler *s, uint32_t num, int32_t out[4]) {   // core/sampler.cpp:55-74
    int count = (int)s->n_tasks;
    int dx = s->x_end - s->x_start, dy = s->y_end - s->y_start;
    int nx = count, ny = 1;
    while ((nx & 0x1) == 0 && 2 * dx * ny < dy * nx) {
        nx >>= 1;
        ny <<= 1;
    }
    int xo = (int)num % nx, yo = (int)num / nx;
    float tx0 = float(xo) / float(nx), tx1 = float(xo + 1) / float(nx);
    float ty0 = float(yo) / float(ny), ty1 = float(yo + 1) / float(ny);
    out[0] = (int)floorf(lerp_host(tx0, (float)s->x_start, (float)s->x_end));
    out[1] = (int)floorf(lerp_host(tx1, (float)s->x_start, (float)s->x_end));
    out[2] = (int)floorf(lerp_host(ty0, (float)s->y_start, (float)s->y_end));
    out[3] = (int)floorf(lerp_host(ty1, (float)s->y_start, (float)s->y_end));
}

extern "C" uint64_t pvol_render_sample_count(const pvol_sampler *s, const uint32_t *taskIds, uint32_t n) {
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; ++i) {
        int32_t w[4];
        pvol_compute_sub_window(s, taskIds[i], w);
        total += (uint64_t)(w[1] - w[0]) * (uint64_t)(w[3] - w[2]) * s->pixel_samples;
    }
    return total;
}

// ------------------------------------------------------------------------------------------ film kernels
struct DevFilm {
    int32_t xres, yres;
    float xw, yw, invXW, invYW;
    float table[PVOL_FILTER_TABLE_SIZE * PVOL_FILTER_TABLE_SIZE];
};

__device__ __forceinline__ float wave_sum_f(float v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ int wave_min_i(int v) {
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
    return v;
}

// ImageFilm::AddSample.  The samples of one pixel are neighbours in the array, so a wave's 64 footprints
// usually share one small pixel window: weights are summed over the wave first and one lane issues the atomics
// (the reference adds sample by sample with AtomicAdd in thread order, i.e. in no particular order either).
// `guard`: the unexpected-radiance checks of samplerrenderer.cpp:118-133 applied to the XYZ record.
__global__ __launch_bounds__(256) void film_add_kernel(DevFilm F, const float *xy, const float *xyz, uint32_t stride,
                                                      unsigned long long n, int guard, float *pixels) {
    __shared__ float table[PVOL_FILTER_TABLE_SIZE * PVOL_FILTER_TABLE_SIZE];
    table[threadIdx.x] = F.table[threadIdx.x];
    __syncthreads();
    const unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool on = i < n;
    float X = 0.f, Y = 0.f, Z = 0.f, dimageX = 0.f, dimageY = 0.f;
    int x0 = 0, x1 = -1, y0 = 0, y1 = -1;
    if (on) {
        X = xyz[i * stride]; Y = xyz[i * stride + 1]; Z = xyz[i * stride + 2];
        if (guard && (isnan(X) || isnan(Y) || isnan(Z) || Y < -1e-5 || isinf(Y))) { X = 0.f; Y = 0.f; Z = 0.f; }
        dimageX = xy[2 * i] - 0.5f;
        dimageY = xy[2 * i + 1] - 0.5f;
        x0 = (int)ceilf(dimageX - F.xw); x1 = (int)floorf(dimageX + F.xw);
        y0 = (int)ceilf(dimageY - F.yw); y1 = (int)floorf(dimageY + F.yw);
        x0 = max(x0, 0); x1 = min(x1, F.xres - 1);
        y0 = max(y0, 0); y1 = min(y1, F.yres - 1);
    }
    const bool valid = on && (x1 - x0) >= 0 && (y1 - y0) >= 0;
    if (!__ballot(valid)) return;
    const int bx = wave_min_i(valid ? x0 : 0x7fffffff), ex = wave_max_i(valid ? x1 : -0x7fffffff);
    const int by = wave_min_i(valid ? y0 : 0x7fffffff), ey = wave_max_i(valid ? y1 : -0x7fffffff);
    if (ex - bx < 8 && ey - by < 8) {
        // the wave's sums for a pixel are four numbers in every lane; lanes 4 g .. 4 g + 3 keep those of the g-th pixel touched, and ONE
        // atomic instruction adds sixteen pixels' worth (the L2 atomic rate is per wave-instruction, MI355X_MICROARCH.md): 2 instead
        // of 100 instructions for a 5 x 5 footprint
        float *myAddr = 0;
        float myVal = 0.f;
        int slot = 0;
        for (int y = by; y <= ey; ++y) {
            const bool iny = valid && y >= y0 && y <= y1;
            const float fy = fabsf((y - dimageY) * F.invYW * PVOL_FILTER_TABLE_SIZE);
            const int iy = min((int)floorf(fy), PVOL_FILTER_TABLE_SIZE - 1);
            for (int x = bx; x <= ex; ++x) {
                float wt = 0.f;
                if (iny && x >= x0 && x <= x1) {
                    const float fx = fabsf((x - dimageX) * F.invXW * PVOL_FILTER_TABLE_SIZE);
                    const int ix = min((int)floorf(fx), PVOL_FILTER_TABLE_SIZE - 1);
                    wt = table[iy * PVOL_FILTER_TABLE_SIZE + ix];
                }
                if (!__ballot(wt != 0.f)) continue;
                const float sx = wave_sum_f(wt * X), sy = wave_sum_f(wt * Y), sz = wave_sum_f(wt * Z), sw = wave_sum_f(wt);
                if ((lane >> 2) == slot) {
                    const int c = lane & 3;
                    myAddr = pixels + 4 * ((size_t)y * F.xres + x) + c;
                    myVal = c == 0 ? sx : (c == 1 ? sy : (c == 2 ? sz : sw));
                }
                if (++slot == 16) {
                    atomicAdd(myAddr, myVal);   // all 64 lanes hold one
                    myAddr = 0; slot = 0;
                }
            }
        }
        if (myAddr) atomicAdd(myAddr, myVal);
    } else if (valid) {
        for (int y = y0; y <= y1; ++y) {
            const float fy = fabsf((y - dimageY) * F.invYW * PVOL_FILTER_TABLE_SIZE);
            const int iy = min((int)floorf(fy), PVOL_FILTER_TABLE_SIZE - 1);
            for (int x = x0; x <= x1; ++x) {
                const float fx = fabsf((x - dimageX) * F.invXW * PVOL_FILTER_TABLE_SIZE);
                const int ix = min((int)floorf(fx), PVOL_FILTER_TABLE_SIZE - 1);
                const float wt = table[iy * PVOL_FILTER_TABLE_SIZE + ix];
                float *p = pixels + 4 * ((size_t)y * F.xres + x);
                atomicAdd(p, wt * X); atomicAdd(p + 1, wt * Y); atomicAdd(p + 2, wt * Z); atomicAdd(p + 3, wt);
            }
        }
    }
}

// ImageFilm::WriteRGB without splats (film/image.cpp:178-214), XYZToRGB core/spectrum.h:51-55
__global__ void film_resolve_kernel(int nPix, const float *pixels, float *rgb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nPix) return;
    const float4 p = reinterpret_cast<const float4 *>(pixels)[i];
    float r = 3.240479f * p.x - 1.537150f * p.y - 0.498535f * p.z;
    float g = -0.969256f * p.x + 1.875991f * p.y + 0.041556f * p.z;
    float b = 0.055648f * p.x - 0.204043f * p.y + 1.057311f * p.z;
    if (p.w != 0.f) {
        const float invWt = 1.f / p.w;
        r = fmaxf(0.f, r * invWt); g = fmaxf(0.f, g * invWt); b = fmaxf(0.f, b * invWt);
    }
    rgb[3 * i] = r; rgb[3 * i + 1] = g; rgb[3 * i + 2] = b;
}

static bool film_ok(const pvol_film *f) {
    return f && f->x_resolution > 0 && f->y_resolution > 0 && f->filter_xwidth > 0.f && f->filter_ywidth > 0.f &&
           f->filter_xwidth <= 3.f && f->filter_ywidth <= 3.f;
}
static DevFilm dev_film(const pvol_film *f) {
    DevFilm F;
    F.xres = f->x_resolution; F.yres = f->y_resolution;
    F.xw = f->filter_xwidth; F.yw = f->filter_ywidth;
    F.invXW = 1.f / f->filter_xwidth; F.invYW = 1.f / f->filter_ywidth;   // Filter ctor, core/filter.h:47-49
    memcpy(F.table, f->filter_table, sizeof(F.table));
    return F;
}

static int film_add(pvol_ctx *c, const pvol_film *film, const float *dXY, const float *dXYZ, uint32_t stride, uint64_t n, int guard,
                    float *dPixels, hipStream_t stream) {
    if (!n) return PVOL_OK;
    const DevFilm F = dev_film(film);
    const unsigned long long blocks = (n + 255ull) / 256ull;
    if (blocks > 0x7fffffffull) return PVOL_E_LIMIT;
    hipLaunchKernelGGL(film_add_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, F, dXY, dXYZ, stride, (unsigned long long)n, guard, dPixels);
    return ok(hipGetLastError()) ? PVOL_OK : PVOL_E_NO_DEVICE;
}

extern "C" int pvol_film_add_samples_device(pvol_ctx *c, const pvol_film *film, const float *dXY, const float *dXYZ, uint32_t stride,
                                            uint64_t n, float *dPixels, void *hipStream) {
    if (!c || !film_ok(film) || (n && (!dXY || !dXYZ || !dPixels)) || stride < 3) return PVOL_E_INVALID;
    if (!ok(hipSetDevice(c->params.device))) return PVOL_E_NO_DEVICE;
    return film_add(c, film, dXY, dXYZ, stride, n, 0, dPixels, (hipStream_t)hipStream);
}

extern "C" int pvol_film_resolve_device(pvol_ctx *c, const pvol_film *film, const float *dPixels, float *dRgb, void *hipStream) {
    if (!c || !film_ok(film) || !dPixels || !dRgb) return PVOL_E_INVALID;
    if (!ok(hipSetDevice(c->params.device))) return PVOL_E_NO_DEVICE;
    const int nPix = film->x_resolution * film->y_resolution;
    hipLaunchKernelGGL(film_resolve_kernel, dim3((nPix + 255) / 256), dim3(256), 0, (hipStream_t)hipStream, nPix, dPixels, dRgb);
    return ok(hipGetLastError()) ? PVOL_OK : PVOL_E_NO_DEVICE;
}

// ------------------------------------------------------------------------------------------ render
static bool grow(pvol_ctx *c, int slot, size_t bytes) {
    if (bytes <= c->tileBytes[slot]) return true;
    if (c->dTile[slot]) hipFree(c->dTile[slot]);
    c->dTile[slot] = 0; c->tileBytes[slot] = 0;
    if (!ok(hipMalloc(&c->dTile[slot], bytes))) return false;
    c->tileBytes[slot] = bytes;
    return true;
}

extern "C" int pvol_render_tasks_device(pvol_ctx *c, const pvol_camera *camera, const pvol_film *film, const pvol_sampler *smp,
                                        const uint32_t *taskIds, uint32_t nTaskIds, float *dPixels, const pvol_render_debug *debug,
                                        void *hipStream) {
    if (!c || !camera || !smp || !film_ok(film) || (nTaskIds && !taskIds) || !dPixels) return PVOL_E_INVALID;
    if (!c->haveScene) return PVOL_E_NO_SCENE;
    const uint32_t spp = smp->pixel_samples;
    if (spp == 0 || (spp & (spp - 1)) || spp > PVOL_MAX_PIXEL_SAMPLES) return PVOL_E_INVALID;   // LDSampler rounds up to a power of two itself
    if (camera->lens_radius != 0.f) return PVOL_E_UNSUPPORTED;
    if (smp->n1d_count > PVOL_MAX_SAMPLE_ARRAYS || smp->n2d_count > PVOL_MAX_SAMPLE_ARRAYS) return PVOL_E_LIMIT;
    if (smp->scatter_index >= smp->n1d_count || smp->n1d[smp->scatter_index] != 1) return PVOL_E_INVALID;
    if (smp->n_tasks == 0 || smp->x_end < smp->x_start || smp->y_end < smp->y_start) return PVOL_E_INVALID;
    for (uint32_t i = 0; i < nTaskIds; ++i) if (taskIds[i] >= smp->n_tasks) return PVOL_E_INVALID;
    std::lock_guard<std::recursive_mutex> api(c->apiMu);   // the work buffers and launch scratch live in the context
    if (!ok(hipSetDevice(c->params.device))) return PVOL_E_NO_DEVICE;
    hipStream_t stream = (hipStream_t)hipStream;

    TileArgs T;
    memset(&T, 0, sizeof(T));
    memcpy(T.r2c, camera->raster_to_camera, sizeof(T.r2c));
    memcpy(T.c2w, camera->camera_to_world, sizeof(T.c2w));
    T.shutterOpen = camera->shutter_open; T.shutterClose = camera->shutter_close;
    T.spp = spp; T.n1dCount = smp->n1d_count; T.n2dCount = smp->n2d_count;
    memcpy(T.n1d, smp->n1d, sizeof(T.n1d));
    memcpy(T.n2d, smp->n2d, sizeof(T.n2d));
    T.scatterIndex = smp->scatter_index;
#ifdef PVOL_TIMING_KNOBS   // timing experiments only (tools/tile_debug_*.sh build with it): every knob gives WRONG images and stream
                           // positions, so the shipped library does not read the variable at all
    if (const char *dbg = getenv("PVOL_TILE_DEBUG")) T.debugSkip = (uint32_t)atoi(dbg);
#endif

    // batches of tasks bounded by the work-buffer budget (rays 48 B + xy 8 B + XYZ 16 B per sample)
    size_t batchRays = (size_t)256 << 20;
    if (const char *ev = getenv("PVOL_TILE_BATCH_RAYS")) { long long v = atoll(ev); if (v > 0) batchRays = (size_t)v; }
    batchRays = std::min<size_t>(batchRays, 0xfffff000u);
    std::vector<int32_t> win(4 * (size_t)nTaskIds);
    std::vector<uint64_t> count(nTaskIds);
    for (uint32_t i = 0; i < nTaskIds; ++i) {
        pvol_compute_sub_window(smp, taskIds[i], &win[4 * i]);
        count[i] = (uint64_t)(win[4 * i + 1] - win[4 * i]) * (uint64_t)(win[4 * i + 3] - win[4 * i + 2]) * spp;
        if (count[i] > batchRays) return PVOL_E_LIMIT;
    }
    uint64_t doneRays = 0;
    for (uint32_t b0 = 0; b0 < nTaskIds;) {
        uint32_t b1 = b0;
        uint64_t nRays = 0;
        uint32_t maxRays = 0;
        while (b1 < nTaskIds && nRays + count[b1] <= batchRays) { nRays += count[b1]; maxRays = std::max<uint32_t>(maxRays, (uint32_t)count[b1]); ++b1; }
        const uint32_t nStreams = b1 - b0;
        std::vector<pvol_stream> hs(nStreams);
        uint64_t first = 0;
        for (uint32_t i = 0; i < nStreams; ++i) {
            memset(&hs[i], 0, sizeof(pvol_stream));
            hs[i].seed = taskIds[b0 + i];          // RNG rng(taskNum), samplerrenderer.cpp:73
            hs[i].first_ray = (uint32_t)first;
            hs[i].n_rays = (uint32_t)count[b0 + i];
            first += count[b0 + i];
        }
        if (!grow(c, 0, std::max<size_t>(sizeof(pvol_ray) * nRays, 64)) || !grow(c, 1, std::max<size_t>(8 * nRays, 64)) ||
            !grow(c, 2, std::max<size_t>(16 * nRays, 64)) || !grow(c, 3, sizeof(pvol_stream) * nStreams) || !grow(c, 4, 16 * (size_t)nStreams))
            return PVOL_E_NO_MEMORY;
        pvol_ray *dRays = (pvol_ray *)c->dTile[0];
        float *dXY = (float *)c->dTile[1], *dOut = (float *)c->dTile[2];
        pvol_stream *dStreams = (pvol_stream *)c->dTile[3];
        int4 *dWin = (int4 *)c->dTile[4];
        // the previous batch's kernels still read the buffers: these copies are ordered behind them on `stream`
        if (!ok(hipMemcpyAsync(dStreams, hs.data(), sizeof(pvol_stream) * nStreams, hipMemcpyHostToDevice, stream)) ||
            !ok(hipMemcpyAsync(dWin, &win[4 * (size_t)b0], 16 * (size_t)nStreams, hipMemcpyHostToDevice, stream)))
            return PVOL_E_NO_DEVICE;
        if (!ok(hipStreamSynchronize(stream))) return PVOL_E_NO_DEVICE;   // hs / win are host temporaries
        T.windows = dWin; T.rays = dRays; T.xy = dXY;
        const bool surfOn = c->hs.surf.enabled != 0;
        const bool specOn = surfOn && c->specOn;
        T.specOn = 0; T.specLink = 0;
        if (specOn && nRays) {   // one link word per camera sample: which segments of the specular recursion belong to it
            const size_t want = 4 * nRays;
            if (want > c->specLinkBytes) {
                hipStreamSynchronize(stream);
                if (c->dSpecLink) hipFree(c->dSpecLink);
                c->dSpecLink = 0; c->specLinkBytes = 0;
                if (!ok(hipMalloc(&c->dSpecLink, want))) return PVOL_E_NO_MEMORY;
                c->specLinkBytes = want;
            }
            T.specOn = 1; T.specLink = c->dSpecLink;
            if (!ok(hipMemsetAsync(c->dSpecLink, 0, want, stream))) return PVOL_E_NO_DEVICE;
            c->specSurfOut = debug && debug->d_surf_xyz ? debug->d_surf_xyz + 3 * doneRays : 0;
        }
        if (surfOn && nRays) {
            const size_t want = 4 * nRays;
            if (want > c->tauBytes) {
                hipStreamSynchronize(stream);
                if (c->dTau) hipFree(c->dTau);
                c->dTau = 0; c->tauBytes = 0;
                if (!ok(hipMalloc(&c->dTau, want))) return PVOL_E_NO_MEMORY;
                c->tauBytes = want;
            }
        }
        if (nRays) {
            c->dTauNext = surfOn ? c->dTau : 0;
            int rc = pvol_launch_batch(c, dRays, (uint32_t)nRays, dStreams, nStreams, PVOL_OUT_XYZ, dOut, 0, 0, 0, 0, maxRays, &T, stream);
            c->dTauNext = 0;
            if (rc != PVOL_OK) return rc;
            if (surfOn) {   // Ls of PhotonIntegrator::Li, composed as T * Ls + Lvi (samplerrenderer.cpp:95-97)
                SurfArgs sa;
                memset(&sa, 0, sizeof(sa));
                sa.link = specOn ? c->dSpecLink : 0;
                sa.scene = c->ds; sa.rays = dRays; sa.nRays = (uint32_t)nRays; sa.out = dOut; sa.tau = c->dTau;
                sa.surfOut = debug ? debug->d_surf_xyz ? debug->d_surf_xyz + 3 * doneRays : 0 : 0;
                sa.counters = c->dCounters;
                const unsigned long long groups = (nRays + 63) / 64;
                pvol_phase_mark(c, stream, PVOL_PHASE_SURFACE);
                if (!ok(pvol_launch_surface(&sa, (uint32_t)std::min<unsigned long long>(groups, (unsigned long long)c->nCU * 24ull), stream)))
                    return PVOL_E_NO_DEVICE;
            }
            pvol_phase_mark(c, stream, PVOL_PHASE_FILM);
            rc = film_add(c, film, dXY, dOut, 4, nRays, 1, dPixels, stream);
            pvol_phase_mark(c, stream, PVOL_PHASE_END);
            if (rc != PVOL_OK) return rc;
        }
        if (debug) {
            if (debug->d_rays && nRays) hipMemcpyAsync(debug->d_rays + doneRays, dRays, sizeof(pvol_ray) * nRays, hipMemcpyDeviceToDevice, stream);
            if (debug->d_image_xy && nRays) hipMemcpyAsync(debug->d_image_xy + 2 * doneRays, dXY, 8 * nRays, hipMemcpyDeviceToDevice, stream);
            if (debug->d_xyz && nRays) hipMemcpyAsync(debug->d_xyz + 4 * doneRays, dOut, 16 * nRays, hipMemcpyDeviceToDevice, stream);
            if (debug->d_streams) hipMemcpyAsync(debug->d_streams + b0, dStreams, sizeof(pvol_stream) * nStreams, hipMemcpyDeviceToDevice, stream);
        }
        doneRays += nRays;
        b0 = b1;
    }
    return PVOL_OK;
}

// ------------------------------------------------------------------------------------------ multi-GPU frame (north_star)
// The frame's render tasks dealt round-robin to the ranks: rank r renders tasks r, r + n, r + 2n, ... so that every rank's share
// is spread over the whole frame (SamplerRenderer::Render's task loop, renderers/samplerrenderer.cpp:206-221, cut n ways).
extern "C" int pvol_partition_tasks(uint32_t nTasks, uint32_t rank, uint32_t nRanks, uint32_t *outIds, uint32_t capacity, uint32_t *nOut) {
    if (!nRanks || rank >= nRanks || !nOut) return PVOL_E_INVALID;
    const uint32_t n = rank < nTasks ? (nTasks - rank + nRanks - 1) / nRanks : 0;
    *nOut = n;
    if (!outIds) return PVOL_OK;
    if (capacity < n) return PVOL_E_INVALID;
    for (uint32_t i = 0; i < n; ++i) outIds[i] = rank + i * nRanks;
    return PVOL_OK;
}

// RCCL is bound at run time: a process that already holds a copy (the application's own, or the one PyTorch ships) keeps using
// that one, and a single-GPU user of this library never loads it.
typedef ncclResult_t (*nccl_reduce_fn)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t);
static nccl_reduce_fn bind_nccl_reduce() {
    static nccl_reduce_fn fn = 0;
    static bool tried = false;
    if (tried) return fn;
    tried = true;
    if (void *sym = dlsym(RTLD_DEFAULT, "ncclReduce")) { fn = (nccl_reduce_fn)sym; return fn; }
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names)
        if (void *h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))
            if (void *sym = dlsym(h, "ncclReduce")) { fn = (nccl_reduce_fn)sym; break; }
    return fn;
}

// One rank of an N-GPU frame: its share of the render tasks into its own full-frame film, ONE ncclReduce(sum) of the film to rank 0
// (the Gaussian filter splats across tile borders, film/image.cpp:82-134, so tiles cannot simply be gathered), resolve on rank 0.
// The photon map is replicated: every rank runs pvol_preprocess with the same seeds beforehand -- nothing else crosses GPUs.
extern "C" int pvol_render_frame_ranks(pvol_ctx *c, const pvol_camera *camera, const pvol_film *film, const pvol_sampler *smp,
                                       uint32_t rank, uint32_t nRanks, void *ncclComm, float *dPixels, float *dRgb, void *hipStream) {
    if (!c || !smp || !film_ok(film) || !dPixels || !nRanks || rank >= nRanks) return PVOL_E_INVALID;
    if (nRanks > 1 && !ncclComm) return PVOL_E_INVALID;
    nccl_reduce_fn reduce = 0;
    if (nRanks > 1 && !(reduce = bind_nccl_reduce())) return PVOL_E_NO_DEVICE;   // no RCCL in reach
    if (!ok(hipSetDevice(c->params.device))) return PVOL_E_NO_DEVICE;
    hipStream_t stream = (hipStream_t)hipStream;
    uint32_t n = 0;
    pvol_partition_tasks(smp->n_tasks, rank, nRanks, 0, 0, &n);
    std::vector<uint32_t> ids(n);
    pvol_partition_tasks(smp->n_tasks, rank, nRanks, ids.data(), n, &n);
    const size_t nFloats = (size_t)film->x_resolution * film->y_resolution * 4;
    if (!ok(hipMemsetAsync(dPixels, 0, nFloats * sizeof(float), stream))) return PVOL_E_NO_DEVICE;
    int rc = pvol_render_tasks_device(c, camera, film, smp, ids.data(), n, dPixels, 0, hipStream);
    if (rc != PVOL_OK) return rc;
    if (nRanks > 1 && reduce(dPixels, dPixels, nFloats, ncclFloat, ncclSum, 0, (ncclComm_t)ncclComm, stream) != ncclSuccess) return PVOL_E_NO_DEVICE;
    if (rank == 0 && dRgb) rc = pvol_film_resolve_device(c, film, dPixels, dRgb, hipStream);
    return rc;
}
