// TEST INFRASTRUCTURE -- CPU restatement of the caller and the consumer of PhotonVolumeIntegrator::Li
// (SURVEY 8(f)-1): LDSampler/LDPixelSample, PerspectiveCamera::GenerateRayDifferential, the sample loop of
// SamplerRendererTask::Run, ImageFilm::AddSample/WriteRGB and the Gaussian filter table.  Pinned against
// the reference's own objects by oracle/ref_capture.cpp `render` (tests/golden/render_*.bin).
#ifndef ORC_TILE_H
#define ORC_TILE_H
#include <algorithm>
#include <vector>

#include "orc_integrator.h"
#include "orc_surface.h"

namespace orc {

inline int floor2int(float v) { return (int)floorf(v); }   // core/pbrt.h Floor2Int
inline int ceil2int(float v) { return (int)ceilf(v); }     // core/pbrt.h Ceil2Int
inline float lerp(float t, float a, float b) { return (1.f - t) * a + t * b; }   // core/pbrt.h Lerp

// filters/gaussian.h:44-58 + film/image.cpp:57-68
inline float gaussian_1d(float d, float alpha, float expv) { return std::max(0.f, float(expf(-alpha * d * d) - expv)); }
inline void gaussian_filter_table(float xw, float yw, float alpha, float *table) {
    const float expX = expf(-alpha * xw * xw), expY = expf(-alpha * yw * yw);
    float *ftp = table;
    for (int y = 0; y < PVOL_FILTER_TABLE_SIZE; ++y) {
        float fy = ((float)y + .5f) * yw / PVOL_FILTER_TABLE_SIZE;
        for (int x = 0; x < PVOL_FILTER_TABLE_SIZE; ++x) {
            float fx = ((float)x + .5f) * xw / PVOL_FILTER_TABLE_SIZE;
            *ftp++ = gaussian_1d(fx, alpha, expX) * gaussian_1d(fy, alpha, expY);
        }
    }
}

// Sampler::ComputeSubWindow, core/sampler.cpp:55-74
inline void compute_sub_window(const pvol_sampler &s, uint32_t num, int32_t out[4]) {
    int count = (int)s.n_tasks;
    int dx = s.x_end - s.x_start, dy = s.y_end - s.y_start;
    int nx = count, ny = 1;
    while ((nx & 0x1) == 0 && 2 * dx * ny < dy * nx) {
        nx >>= 1;
        ny <<= 1;
    }
    int xo = (int)num % nx, yo = (int)num / nx;
    float tx0 = float(xo) / float(nx), tx1 = float(xo + 1) / float(nx);
    float ty0 = float(yo) / float(ny), ty1 = float(yo + 1) / float(ny);
    out[0] = floor2int(lerp(tx0, (float)s.x_start, (float)s.x_end));
    out[1] = floor2int(lerp(tx1, (float)s.x_start, (float)s.x_end));
    out[2] = floor2int(lerp(ty0, (float)s.y_start, (float)s.y_end));
    out[3] = floor2int(lerp(ty1, (float)s.y_start, (float)s.y_end));
}

struct PixelSamples {   // what LDPixelSample leaves in samples[i] that this path reads
    std::vector<float> imageX, imageY, time, lensU, lensV, tau, scatter;
};

// LDPixelSample, core/montecarlo.cpp:200-254 (arrays the volume integrator does not own are generated and dropped)
inline void ld_pixel_sample(int xPos, int yPos, float shutterOpen, float shutterClose, const pvol_sampler &s, PixelSamples &out,
                            std::vector<float> &buf, Rng &rng) {
    const int n = (int)s.pixel_samples;
    size_t need = 5 * (size_t)n;
    for (uint32_t i = 0; i < s.n1d_count; ++i) need += (size_t)s.n1d[i] * n;
    for (uint32_t i = 0; i < s.n2d_count; ++i) need += 2 * (size_t)s.n2d[i] * n;
    buf.resize(need);
    float *b = buf.data();
    float *imageSamples = b; b += 2 * n;
    float *lensSamples = b; b += 2 * n;
    float *timeSamples = b; b += n;
    std::vector<float *> oneD(s.n1d_count), twoD(s.n2d_count);
    for (uint32_t i = 0; i < s.n1d_count; ++i) { oneD[i] = b; b += s.n1d[i] * n; }
    for (uint32_t i = 0; i < s.n2d_count; ++i) { twoD[i] = b; b += 2 * s.n2d[i] * n; }
    ld_shuffle_scrambled_2d(1, n, imageSamples, rng);
    ld_shuffle_scrambled_2d(1, n, lensSamples, rng);
    ld_shuffle_scrambled_1d(1, n, timeSamples, rng);
    for (uint32_t i = 0; i < s.n1d_count; ++i) ld_shuffle_scrambled_1d((int)s.n1d[i], n, oneD[i], rng);
    for (uint32_t i = 0; i < s.n2d_count; ++i) ld_shuffle_scrambled_2d((int)s.n2d[i], n, twoD[i], rng);
    out.imageX.resize(n); out.imageY.resize(n); out.time.resize(n); out.lensU.resize(n); out.lensV.resize(n);
    out.tau.resize(n); out.scatter.resize(n);
    for (int i = 0; i < n; ++i) {
        out.imageX[i] = xPos + imageSamples[2 * i];
        out.imageY[i] = yPos + imageSamples[2 * i + 1];
        out.time[i] = lerp(timeSamples[i], shutterOpen, shutterClose);
        out.lensU[i] = lensSamples[2 * i];
        out.lensV[i] = lensSamples[2 * i + 1];
        out.tau[i] = s.tau_index < s.n1d_count ? oneD[s.tau_index][s.n1d[s.tau_index] * i] : 0.f;
        out.scatter[i] = s.scatter_index < s.n1d_count ? oneD[s.scatter_index][s.n1d[s.scatter_index] * i] : 0.f;
    }
}

// PerspectiveCamera::GenerateRayDifferential with lensRadius == 0, cameras/perspective.cpp:80-134
inline Ray camera_ray(const pvol_camera &cam, float imageX, float imageY, float time) {
    V3 Pras = v3(imageX, imageY, 0.f);
    // Transform::operator()(const Point &, Point *), core/transform.h:204-213: divides by w unless w == 1
    const float *m = cam.raster_to_camera;
    V3 Pc;
    Pc.x = m[0] * Pras.x + m[1] * Pras.y + m[2] * Pras.z + m[3];
    Pc.y = m[4] * Pras.x + m[5] * Pras.y + m[6] * Pras.z + m[7];
    Pc.z = m[8] * Pras.x + m[9] * Pras.y + m[10] * Pras.z + m[11];
    float w = m[12] * Pras.x + m[13] * Pras.y + m[14] * Pras.z + m[15];
    if (w != 1.f) { float inv = 1.f / w; Pc.x *= inv; Pc.y *= inv; Pc.z *= inv; }
    V3 dir = normalize(Pc);
    const float *c = cam.camera_to_world;
    V3 o = v3(0.f, 0.f, 0.f), ow;
    ow.x = c[0] * o.x + c[1] * o.y + c[2] * o.z + c[3];
    ow.y = c[4] * o.x + c[5] * o.y + c[6] * o.z + c[7];
    ow.z = c[8] * o.x + c[9] * o.y + c[10] * o.z + c[11];
    float ww = c[12] * o.x + c[13] * o.y + c[14] * o.z + c[15];
    if (ww != 1.f) { float inv = 1.f / ww; ow.x *= inv; ow.y *= inv; ow.z *= inv; }
    V3 dw = xform_vector(c, dir);
    return make_ray(ow, dw, 0.f, kInfinity, time);
}

// ImageFilm (film/image.cpp): pixels as Lxyz[3], weightSum
struct Film {
    pvol_film f;
    std::vector<float> pix;   // 4 floats per pixel
    void init(const pvol_film &ff) { f = ff; pix.assign((size_t)4 * ff.x_resolution * ff.y_resolution, 0.f); }
    // film/image.cpp:78-137
    void add_sample(float imageX, float imageY, const float xyz[3]) {
        float dimageX = imageX - 0.5f, dimageY = imageY - 0.5f;
        int x0 = ceil2int(dimageX - f.filter_xwidth), x1 = floor2int(dimageX + f.filter_xwidth);
        int y0 = ceil2int(dimageY - f.filter_ywidth), y1 = floor2int(dimageY + f.filter_ywidth);
        x0 = std::max(x0, 0); x1 = std::min(x1, f.x_resolution - 1);
        y0 = std::max(y0, 0); y1 = std::min(y1, f.y_resolution - 1);
        if ((x1 - x0) < 0 || (y1 - y0) < 0) return;
        const float invXW = 1.f / f.filter_xwidth, invYW = 1.f / f.filter_ywidth;   // Filter ctor, core/filter.h:47-49
        for (int y = y0; y <= y1; ++y) {
            float fy = fabsf((y - dimageY) * invYW * PVOL_FILTER_TABLE_SIZE);
            int iy = std::min(floor2int(fy), PVOL_FILTER_TABLE_SIZE - 1);
            for (int x = x0; x <= x1; ++x) {
                float fx = fabsf((x - dimageX) * invXW * PVOL_FILTER_TABLE_SIZE);
                int ix = std::min(floor2int(fx), PVOL_FILTER_TABLE_SIZE - 1);
                float wt = f.filter_table[iy * PVOL_FILTER_TABLE_SIZE + ix];
                float *p = &pix[4 * ((size_t)y * f.x_resolution + x)];
                p[0] += wt * xyz[0]; p[1] += wt * xyz[1]; p[2] += wt * xyz[2]; p[3] += wt;
            }
        }
    }
    // film/image.cpp:178-214 with no splats; XYZToRGB core/spectrum.h:51-55
    void write_rgb(float *rgb) const {
        size_t n = (size_t)f.x_resolution * f.y_resolution;
        for (size_t i = 0; i < n; ++i) {
            const float *xyz = &pix[4 * i];
            float *o = rgb + 3 * i;
            o[0] = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];
            o[1] = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
            o[2] = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
            float weightSum = xyz[3];
            if (weightSum != 0.f) {
                float invWt = 1.f / weightSum;
                o[0] = std::max(0.f, o[0] * invWt); o[1] = std::max(0.f, o[1] * invWt); o[2] = std::max(0.f, o[2] * invWt);
            }
        }
    }
};

struct TileRecords {   // optional per-sample records of render_task
    std::vector<pvol_ray> *rays;
    std::vector<float> *imageXY, *xyzT;
    std::vector<float> *surfXYZ;   // surface integrator's Li as X, Y, Z (only with a surface integrator)
};

// SamplerRendererTask::Run, renderers/samplerrenderer.cpp:59-157, with Ls = Lvi (no surface term) and the
// unexpected-radiance guards of :118-133.  Returns the number of RandomUInt draws the task made.
// With a surface integrator `SI` the sample's radiance is T * Lsurface + Lvi as SamplerRenderer::Li composes it
// (samplerrenderer.cpp:238-250), the surface term first (it draws before the volume integrator does).
inline uint64_t render_task(const Integrator &I, const pvol_camera &cam, const pvol_sampler &smp, uint32_t taskNum, Film *film,
                            TileRecords *rec, Counters *ctr, const SurfaceIntegrator *SI = 0, bool *supported = 0) {
    int32_t w[4];
    compute_sub_window(smp, taskNum, w);
    if (w[0] == w[1] || w[2] == w[3]) return 0;   // GetSubSampler returns NULL (lowdiscrepancy.cpp:61-66): no RNG either
    Rng rng(taskNum);
    PixelSamples ps;
    std::vector<float> buf, scratch;
    std::vector<ClosePhoton> lookupBuf;
    const Cie &cie = I.scene->cie;
    for (int yPos = w[2]; yPos < w[3]; ++yPos)
        for (int xPos = w[0]; xPos < w[1]; ++xPos) {
            uint64_t d0 = rng.draws;
            ld_pixel_sample(xPos, yPos, cam.shutter_open, cam.shutter_close, smp, ps, buf, rng);
            uint64_t samplerDraws = rng.draws - d0;
            for (uint32_t i = 0; i < smp.pixel_samples; ++i) {
                Ray ray = camera_ray(cam, ps.imageX[i], ps.imageY[i], ps.time[i]);
                Hit hit;
                const bool hitSurface = scene_intersect(*I.scene, &ray, &hit);   // SamplerRenderer::Li, samplerrenderer.cpp:236-249: clips ray.maxt
                Spec Lsurf = spec_const(0.f);
                const uint64_t ds0 = rng.draws;
                SpecCtx sx = {ps.scatter[i], &scratch};
                if (SI && hitSurface) Lsurf = surface_li(I, *SI, ray, hit, rng, ctr, lookupBuf, supported, 0, &sx);
                const uint64_t surfDraws = rng.draws - ds0;
                Spec T;
                Spec Lv = li(I, ray, ps.scatter[i], rng, &T, ctr, scratch, lookupBuf);
                if (SI) Lv = T * Lsurf + Lv;
                bool bad = false;
                for (int b = 0; b < NB; ++b) if (std::isnan(Lv.c[b])) bad = true;
                float y = spec_y(cie, Lv);
                if (bad || y < -1e-5 || std::isinf(y)) Lv = spec_const(0.f);
                float xyz[3];
                spec_xyz(cie, Lv, xyz);
                if (film) film->add_sample(ps.imageX[i], ps.imageY[i], xyz);
                if (rec) {
                    pvol_ray pr;
                    memset(&pr, 0, sizeof(pr));
                    pr.o[0] = ray.o.x; pr.o[1] = ray.o.y; pr.o[2] = ray.o.z;
                    pr.d[0] = ray.d.x; pr.d[1] = ray.d.y; pr.d[2] = ray.d.z;
                    pr.mint = ray.mint; pr.maxt = ray.maxt; pr.time = ray.time; pr.scatter_u = ps.scatter[i];
                    pr.rng_skip = ((i == 0) ? (uint32_t)samplerDraws : 0u) + (uint32_t)surfDraws;
                    rec->rays->push_back(pr);
                    rec->imageXY->push_back(ps.imageX[i]); rec->imageXY->push_back(ps.imageY[i]);
                    rec->xyzT->push_back(xyz[0]); rec->xyzT->push_back(xyz[1]); rec->xyzT->push_back(xyz[2]);
                    rec->xyzT->push_back(spec_y(cie, T));
                    if (rec->surfXYZ) { float sx[3]; spec_xyz(cie, Lsurf, sx); rec->surfXYZ->push_back(sx[0]); rec->surfXYZ->push_back(sx[1]); rec->surfXYZ->push_back(sx[2]); }
                }
            }
        }
    return rng.draws;
}

}  // namespace orc
#endif
