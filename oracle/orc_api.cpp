// oracle/orc_api.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see orc_core.h).
//
// C entry points of the CPU oracle (liborc.so), bound with ctypes by tests/ and by
// bench.py's cpu_baseline leg.  The POD layouts are shared with include/pvol.h so the same
// scene/ray/stream buffers can be handed to the oracle and to the HIP path.
#include <thread>
#include <atomic>
#include <mutex>

#include "orc_integrator.h"
#include "orc_shooter.h"
#include "orc_tile.h"

using namespace orc;

struct orc_ctx {
    pvol_params params;
    Scene scene;
    std::vector<Photon> photons;  // in upload / merge order
    KdTree *map;
    Counters ctr;
    ShootStats shoot_stats;
    bool keep_surface;
    SurfaceStores surf;
    // surface integrator (orc_surface.h), optional
    bool have_si;
    SurfaceIntegrator si;
    std::vector<Photon> causticPhotons;
    KdTree *causticMap;
    orc_ctx() : map(0), keep_surface(false), have_si(false), causticMap(0) {}
    ~orc_ctx() { delete map; delete causticMap; }
};

static Integrator make_integrator(const orc_ctx *c) {
    Integrator I;
    I.stepSize = c->params.step_size;
    I.maxDist = c->params.max_dist;
    I.maxDistSquared = c->params.max_dist * c->params.max_dist;  // photonvolume.h:18
    I.nUsed = c->params.n_used;
    I.scene = &c->scene;
    I.volumeMap = c->map;
    return I;
}

extern "C" {

orc_ctx *orc_create(const pvol_params *params, const pvol_scene *scene) {
    orc_ctx *c = new orc_ctx;
    c->params = *params;
    scene_from_pod(scene, &c->scene);
    return c;
}

void orc_destroy(orc_ctx *c) { delete c; }

int orc_set_photons(orc_ctx *c, const float *p, const float *wi, const float *alpha, uint32_t n) {
    c->photons.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        Photon &ph = c->photons[i];
        ph.p = v3(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
        ph.wi = v3(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]);
        memcpy(ph.alpha.c, alpha + (size_t)NB * i, sizeof(float) * NB);
    }
    delete c->map;
    c->map = n ? new KdTree(c->photons) : 0;  // photonshooter.cpp:502-503
    return 0;
}

uint32_t orc_photon_count(orc_ctx *c) { return (uint32_t)c->photons.size(); }

int orc_get_photons(orc_ctx *c, float *p, float *wi, float *alpha, uint32_t capacity) {
    uint32_t n = std::min<uint32_t>(capacity, (uint32_t)c->photons.size());
    for (uint32_t i = 0; i < n; ++i) {
        const Photon &ph = c->photons[i];
        p[3 * i] = ph.p.x; p[3 * i + 1] = ph.p.y; p[3 * i + 2] = ph.p.z;
        wi[3 * i] = ph.wi.x; wi[3 * i + 1] = ph.wi.y; wi[3 * i + 2] = ph.wi.z;
        memcpy(alpha + (size_t)NB * i, ph.alpha.c, sizeof(float) * NB);
    }
    return (int)n;
}

static void write_out(const Cie &cie, int kind, float *out, size_t ray, const Spec &Lv, const Spec &T) {
    if (kind == PVOL_OUT_SPECTRAL) {
        memcpy(out + ray * 60, Lv.c, sizeof(float) * NB);
        memcpy(out + ray * 60 + NB, T.c, sizeof(float) * NB);
    } else {
        float xyz[3];
        spec_xyz(cie, Lv, xyz);
        out[ray * 4 + 0] = xyz[0]; out[ray * 4 + 1] = xyz[1]; out[ray * 4 + 2] = xyz[2];
        out[ray * 4 + 3] = spec_y(cie, T);
    }
}

// One MT19937 stream per render tile, rays of a stream in array order
// (renderers/samplerrenderer.cpp:73,85-111).  n_threads > 1 distributes streams.
int orc_li_batch(orc_ctx *c, const pvol_ray *rays, uint32_t n_rays, pvol_stream *streams, uint32_t n_streams,
                 int output_kind, float *out, uint32_t *draws, int n_threads) {
    (void)n_rays;
    Integrator I = make_integrator(c);
    std::atomic<uint32_t> next(0);
    std::mutex mu;
    auto worker = [&]() {
        Counters local;
        std::vector<float> scratch;
        std::vector<ClosePhoton> lookupBuf;
        for (;;) {
            uint32_t s = next.fetch_add(1);
            if (s >= n_streams) break;
            pvol_stream &st = streams[s];
            Rng rng(st.seed);
            rng.skip(st.start_draw);
            for (uint32_t k = 0; k < st.n_rays; ++k) {
                size_t ri = (size_t)st.first_ray + k;
                const pvol_ray &pr = rays[ri];
                rng.skip(pr.rng_skip);
                Ray ray = make_ray(v3(pr.o[0], pr.o[1], pr.o[2]), v3(pr.d[0], pr.d[1], pr.d[2]), pr.mint, pr.maxt, pr.time);
                uint64_t d0 = rng.draws;
                Spec T;
                Spec Lv = li(I, ray, pr.scatter_u, rng, &T, &local, scratch, lookupBuf);
                if (draws) draws[ri] = (uint32_t)(rng.draws - d0);
                write_out(c->scene.cie, output_kind, out, ri, Lv, T);
            }
            st.end_draw = rng.draws;
        }
        std::lock_guard<std::mutex> g(mu);
        c->ctr.add(local);
    };
    if (n_threads <= 1) worker();
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t) th.emplace_back(worker);
        for (auto &t : th) t.join();
    }
    return 0;
}

int orc_transmittance_batch(orc_ctx *c, const pvol_ray *rays, uint32_t n_rays, pvol_stream *streams, uint32_t n_streams, float *out) {
    (void)n_rays;
    Integrator I = make_integrator(c);
    for (uint32_t s = 0; s < n_streams; ++s) {
        pvol_stream &st = streams[s];
        Rng rng(st.seed);
        rng.skip(st.start_draw);
        for (uint32_t k = 0; k < st.n_rays; ++k) {
            size_t ri = (size_t)st.first_ray + k;
            const pvol_ray &pr = rays[ri];
            rng.skip(pr.rng_skip);
            Ray ray = make_ray(v3(pr.o[0], pr.o[1], pr.o[2]), v3(pr.d[0], pr.d[1], pr.d[2]), pr.mint, pr.maxt, pr.time);
            Spec T = transmittance(I, ray, rng, &c->ctr);
            memcpy(out + ri * NB, T.c, sizeof(float) * NB);
        }
        st.end_draw = rng.draws;
    }
    return 0;
}

// counters: n_rays, n_steps, n_lookups, n_nodes_visited, n_heap_offers, n_kept, n_lookups_lt10,
//           n_shadow_unoccluded, n_density_evals, n_draws
int orc_get_counters(orc_ctx *c, uint64_t *out10, int reset) {
    const Counters &k = c->ctr;
    uint64_t v[10] = {k.n_rays, k.n_steps, k.n_lookups, k.n_nodes_visited, k.n_heap_offers, k.n_kept,
                      k.n_lookups_lt10, k.n_shadow_unoccluded, k.n_density_evals, k.n_draws};
    memcpy(out10, v, sizeof(v));
    if (reset) c->ctr = Counters();
    return 0;
}

// Photon gather alone at explicit query points (LPhoton, photonvolume.cpp:65-108): out 30 floats per query.
int orc_lphoton_batch(orc_ctx *c, const float *pts, const float *w, uint32_t n, float *out) {
    Integrator I = make_integrator(c);
    std::vector<ClosePhoton> buf(std::max(1, I.nUsed));
    for (uint32_t i = 0; i < n; ++i) {
        Spec L = lphoton(I, &buf[0], v3(w[3 * i], w[3 * i + 1], w[3 * i + 2]), v3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]), &c->ctr);
        memcpy(out + (size_t)i * NB, L.c, sizeof(float) * NB);
    }
    return 0;
}

// PhotonShooter::Preprocess (photonshooter.cpp:457-526) with n_tasks virtual tasks merged in
// task order per block round; n_tasks == 1 is the reference at --ncores 1.
int orc_shoot_blocks(orc_ctx *c, uint32_t n_tasks, int n_threads, uint32_t block_paths);
int orc_shoot(orc_ctx *c, uint32_t n_tasks, int n_threads) { return orc_shoot_blocks(c, n_tasks, n_threads, 4096); }
// the same with `block_paths` paths per task and round (the product's pvol_preprocess_blocks)
int orc_shoot_blocks(orc_ctx *c, uint32_t n_tasks, int n_threads, uint32_t block_paths) {
    std::vector<Photon> vol;
    int rc = shoot_photons(c->scene, c->params, n_tasks, n_threads, &vol, &c->shoot_stats, c->keep_surface ? &c->surf : 0, block_paths);
    c->photons.swap(vol);
    delete c->map;
    c->map = c->photons.size() ? new KdTree(c->photons) : 0;
    return rc;
}

// The surface stores of PhotonShooter::Preprocess (photonshooter.cpp:461-470): kind 0 caustic, 1 direct, 2 indirect
void orc_keep_surface_photons(orc_ctx *c, int on) { c->keep_surface = on != 0; }
uint32_t orc_surface_photon_count(orc_ctx *c, int kind, uint32_t *n_paths) {
    const std::vector<Photon> &v = kind == 0 ? c->surf.caustic : (kind == 1 ? c->surf.direct : c->surf.indirect);
    if (n_paths) *n_paths = kind == 0 ? c->surf.nCausticPaths : (kind == 1 ? c->surf.nDirectPaths : c->surf.nIndirectPaths);
    return (uint32_t)v.size();
}
int orc_get_surface_photons(orc_ctx *c, int kind, float *p, float *wo, float *alpha, uint32_t capacity) {
    const std::vector<Photon> &v = kind == 0 ? c->surf.caustic : (kind == 1 ? c->surf.direct : c->surf.indirect);
    uint32_t n = std::min<uint32_t>(capacity, (uint32_t)v.size());
    for (uint32_t i = 0; i < n; ++i) {
        p[3 * i] = v[i].p.x; p[3 * i + 1] = v[i].p.y; p[3 * i + 2] = v[i].p.z;
        wo[3 * i] = v[i].wi.x; wo[3 * i + 1] = v[i].wi.y; wo[3 * i + 2] = v[i].wi.z;
        memcpy(alpha + (size_t)i * NB, v[i].alpha.c, sizeof(float) * NB);
    }
    return 0;
}
uint32_t orc_radiance_photon_count(orc_ctx *c) { return (uint32_t)c->surf.radiance.size(); }
int orc_get_radiance_photons(orc_ctx *c, float *p, float *n, float *rho_r, float *rho_t, uint32_t capacity) {
    uint32_t m = std::min<uint32_t>(capacity, (uint32_t)c->surf.radiance.size());
    for (uint32_t i = 0; i < m; ++i) {
        const RadPhoton &r = c->surf.radiance[i];
        p[3 * i] = r.p.x; p[3 * i + 1] = r.p.y; p[3 * i + 2] = r.p.z;
        n[3 * i] = r.n.x; n[3 * i + 1] = r.n.y; n[3 * i + 2] = r.n.z;
        memcpy(rho_r + (size_t)i * NB, r.rho_r.c, sizeof(float) * NB);
        memcpy(rho_t + (size_t)i * NB, r.rho_t.c, sizeof(float) * NB);
    }
    return 0;
}

// shoot stats: paths, follow_calls, no_hit, march_steps, interactions, absorbed, stored_volume,
//              stored_caustic, stored_direct, stored_indirect, split_children, nshot
int orc_get_shoot_stats(orc_ctx *c, uint64_t *out12) {
    const ShootStats &s = c->shoot_stats;
    uint64_t v[12] = {s.paths, s.follow_calls, s.no_hit, s.march_steps, s.interactions, s.absorbed, s.stored_volume,
                      s.stored_caustic, s.stored_direct, s.stored_indirect, s.split_children, s.nshot};
    memcpy(out12, v, sizeof(v));
    return 0;
}

// ---- unit-level entry points used to pin the value layer against the reference fixtures
void orc_rng_draws(uint32_t seed, uint32_t n, uint32_t *out) {
    Rng r(seed);
    for (uint32_t i = 0; i < n; ++i) out[i] = r.random_uint();
}
void orc_rng_floats(uint32_t seed, uint32_t n, float *out) {
    Rng r(seed);
    for (uint32_t i = 0; i < n; ++i) out[i] = r.random_float();
}
// PermutedHalton(dims, RNG(seed)).Sample(first..first+n-1): n*dims floats
void orc_halton(uint32_t seed, uint32_t dims, uint32_t first, uint32_t n, float *out) {
    Rng r(seed);
    PermutedHalton h(dims, r);
    for (uint32_t i = 0; i < n; ++i) h.sample(first + i, out + (size_t)i * dims);
}
// LDShuffleScrambled1D / 2D with RNG(seed): returns the number of draws consumed
uint32_t orc_ld_shuffle_1d(uint32_t seed, int nSamples, int nPixel, float *out) {
    Rng r(seed);
    ld_shuffle_scrambled_1d(nSamples, nPixel, out, r);
    return (uint32_t)r.draws;
}
uint32_t orc_ld_shuffle_2d(uint32_t seed, int nSamples, int nPixel, float *out) {
    Rng r(seed);
    ld_shuffle_scrambled_2d(nSamples, nPixel, out, r);
    return (uint32_t)r.draws;
}
float orc_spec_y(orc_ctx *c, const float *s30) {
    Spec s; memcpy(s.c, s30, sizeof(s.c));
    return spec_y(c->scene.cie, s);
}
void orc_spec_xyz(orc_ctx *c, const float *s30, float *xyz) {
    Spec s; memcpy(s.c, s30, sizeof(s.c));
    spec_xyz(c->scene.cie, s, xyz);
}
// ComputeLightSamplingCDF (core/integrator.cpp:261-268): power.y() per light
void orc_light_powers(orc_ctx *c, float *out) {
    for (size_t i = 0; i < c->scene.lights.size(); ++i) out[i] = spec_y(c->scene.cie, light_power(c->scene, c->scene.lights[i]));
}
// Light::Sample_L(scene, ...) emission: out = ray o(3) d(3) Ns(3) pdf(1) Le(30)
void orc_light_emit(orc_ctx *c, uint32_t light, float u0, float u1, float *out) {
    Ray ray; V3 Ns; float pdf;
    Spec Le = light_sample_emit(c->scene, c->scene.lights[light], u0, u1, 0.f, &ray, &Ns, &pdf);
    out[0] = ray.o.x; out[1] = ray.o.y; out[2] = ray.o.z; out[3] = ray.d.x; out[4] = ray.d.y; out[5] = ray.d.z;
    out[6] = Ns.x; out[7] = Ns.y; out[8] = Ns.z; out[9] = pdf;
    memcpy(out + 10, Le.c, sizeof(float) * NB);
}
// Light::Sample_L(p, ...) : out = wi(3) pdf(1) vis o(3) d(3) mint maxt L(30)
void orc_light_sample(orc_ctx *c, uint32_t light, const float *p, float *out) {
    V3 wi; float pdf; Ray vis;
    Spec L = light_sample_L(c->scene.lights[light], v3(p[0], p[1], p[2]), 0.f, 0.f, &wi, &pdf, &vis);
    out[0] = wi.x; out[1] = wi.y; out[2] = wi.z; out[3] = pdf;
    out[4] = vis.o.x; out[5] = vis.o.y; out[6] = vis.o.z; out[7] = vis.d.x; out[8] = vis.d.y; out[9] = vis.d.z;
    out[10] = vis.mint; out[11] = vis.maxt;
    memcpy(out + 12, L.c, sizeof(float) * NB);
}
// Scene::Intersect: returns hit flag; out = t, p(3), nn(3), dpdu(3), tri
int orc_intersect(orc_ctx *c, const float *o, const float *d, float mint, float maxt, float *out) {
    Ray r = make_ray(v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]), mint, maxt, 0.f);
    Hit h;
    if (!scene_intersect(c->scene, &r, &h)) return 0;
    out[0] = h.t; out[1] = h.p.x; out[2] = h.p.y; out[3] = h.p.z; out[4] = h.nn.x; out[5] = h.nn.y; out[6] = h.nn.z;
    out[7] = h.dpdu.x; out[8] = h.dpdu.y; out[9] = h.dpdu.z; out[10] = (float)h.tri;
    return 1;
}
int orc_intersect_p(orc_ctx *c, const float *o, const float *d, float mint, float maxt) {
    Ray r = make_ray(v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]), mint, maxt, 0.f);
    return scene_intersect_p(c->scene, r) ? 1 : 0;
}
// BSDF::Sample_f at a surface hit (see orc_shooter.h): out = wi(3) pdf flags f(30)
int orc_bsdf_sample(orc_ctx *c, int tri, const float *wo, const float *alpha30, float u0, float u1, float ucomp,
                    const float *dpdu, const float *nn, float *out) {
    Spec a; memcpy(a.c, alpha30, sizeof(a.c));
    V3 wi; float pdf; int flags;
    Spec f = bsdf_sample_f(c->scene, tri, v3(dpdu[0], dpdu[1], dpdu[2]), v3(nn[0], nn[1], nn[2]), v3(wo[0], wo[1], wo[2]), &wi, u0, u1, ucomp,
                           &pdf, &flags, a);
    out[0] = wi.x; out[1] = wi.y; out[2] = wi.z; out[3] = pdf; out[4] = (float)flags;
    memcpy(out + 5, f.c, sizeof(float) * NB);
    return 0;
}

// volume queries: in = p(3) d(3) len step offs ; out = hit t0 t1 tau(30) sigma_a(30) sigma_s(30) phase(p,d,-d)
void orc_volume_query(orc_ctx *c, const float *in, float *out) {
    V3 p = v3(in[0], in[1], in[2]), d = v3(in[3], in[4], in[5]);
    Ray r = make_ray(p, d, 0.f, in[6], 0.f);
    float t0 = -1, t1 = -1;
    bool hit = vol_intersect(c->scene.vol, r, &t0, &t1);
    out[0] = hit ? 1.f : 0.f; out[1] = t0; out[2] = t1;
    Spec tau = vol_tau(c->scene.vol, r, in[7], in[8]);
    memcpy(out + 3, tau.c, sizeof(float) * NB);
    Spec sa = vol_sigma_a(c->scene.vol, p), ss = vol_sigma_s(c->scene.vol, p);
    memcpy(out + 33, sa.c, sizeof(float) * NB);
    memcpy(out + 63, ss.c, sizeof(float) * NB);
    out[93] = vol_phase(c->scene.vol, p, d, -d);
}
void orc_rainbow(const float *Ld30, const float *w, const float *wi, float *out30) {
    Spec s; memcpy(s.c, Ld30, sizeof(s.c));
    Spec r = rainbow_reflection(s, v3(w[0], w[1], w[2]), v3(wi[0], wi[1], wi[2]));
    memcpy(out30, r.c, sizeof(r.c));
}
// out = sphere(3) cone95(3) disk(2) coshemi(3)
void orc_mc_samples(float u1, float u2, float *out) {
    V3 a = uniform_sample_sphere(u1, u2), b = uniform_sample_cone(u1, u2, 0.95f), h = cosine_sample_hemisphere(u1, u2);
    float dx, dy;
    concentric_sample_disk(u1, u2, &dx, &dy);
    out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = b.x; out[4] = b.y; out[5] = b.z; out[6] = dx; out[7] = dy;
    out[8] = h.x; out[9] = h.y; out[10] = h.z;
}
// out = HG(g=0) HG(g=.6) HG(g=-.3) MieHazy for w=(0,0,1), wp=(sqrt(1-c^2),0,c)
void orc_phase(float c, float *out) {
    V3 w = v3(0, 0, 1), wp = v3(sqrtf(std::max(0.f, 1 - c * c)), 0, c);
    out[0] = phase_hg(w, wp, 0.f); out[1] = phase_hg(w, wp, 0.6f); out[2] = phase_hg(w, wp, -0.3f); out[3] = phase_mie_hazy(w, wp);
}

// ---------------------------------------------------------------- tile driver (orc_tile.h), SURVEY 8(f)-1
void orc_gaussian_filter_table(float xw, float yw, float alpha, float *table256) { gaussian_filter_table(xw, yw, alpha, table256); }

void orc_compute_sub_window(const pvol_sampler *s, uint32_t task, int32_t *out4) { compute_sub_window(*s, task, out4); }

// LDPixelSample for one pixel from RNG(seed) advanced by `skip` draws; out arrays hold pixel_samples floats each
uint64_t orc_ld_pixel_sample(const pvol_sampler *s, int xPos, int yPos, float shutterOpen, float shutterClose, uint32_t seed, uint64_t skip,
                             float *imageX, float *imageY, float *time, float *lensU, float *lensV, float *tau, float *scatter) {
    Rng rng(seed);
    rng.skip(skip);
    uint64_t d0 = rng.draws;
    PixelSamples ps;
    std::vector<float> buf;
    ld_pixel_sample(xPos, yPos, shutterOpen, shutterClose, *s, ps, buf, rng);
    size_t nb = sizeof(float) * s->pixel_samples;
    memcpy(imageX, ps.imageX.data(), nb); memcpy(imageY, ps.imageY.data(), nb); memcpy(time, ps.time.data(), nb);
    memcpy(lensU, ps.lensU.data(), nb); memcpy(lensV, ps.lensV.data(), nb); memcpy(tau, ps.tau.data(), nb); memcpy(scatter, ps.scatter.data(), nb);
    return rng.draws - d0;
}

void orc_camera_rays(const pvol_camera *cam, const float *imageXY, const float *time, uint32_t n, pvol_ray *out) {
    for (uint32_t i = 0; i < n; ++i) {
        Ray r = camera_ray(*cam, imageXY[2 * i], imageXY[2 * i + 1], time ? time[i] : 0.f);
        pvol_ray &pr = out[i];
        memset(&pr, 0, sizeof(pr));
        pr.o[0] = r.o.x; pr.o[1] = r.o.y; pr.o[2] = r.o.z; pr.d[0] = r.d.x; pr.d[1] = r.d.y; pr.d[2] = r.d.z;
        pr.mint = r.mint; pr.maxt = r.maxt; pr.time = r.time;
    }
}

void orc_film_add_samples(const pvol_film *film, const float *imageXY, const float *xyz, uint32_t stride, uint64_t n, float *pixels) {
    Film F;
    F.f = *film;
    size_t np = (size_t)4 * film->x_resolution * film->y_resolution;
    F.pix.assign(pixels, pixels + np);
    for (uint64_t i = 0; i < n; ++i) F.add_sample(imageXY[2 * i], imageXY[2 * i + 1], xyz + (size_t)stride * i);
    memcpy(pixels, F.pix.data(), sizeof(float) * np);
}

void orc_film_resolve(const pvol_film *film, const float *pixels, float *rgb) {
    Film F;
    F.f = *film;
    size_t np = (size_t)4 * film->x_resolution * film->y_resolution;
    F.pix.assign(pixels, pixels + np);
    F.write_rgb(rgb);
}

// SamplerRendererTask::Run for the listed tasks.  pixels (optional) is accumulated into in task order when
// n_threads <= 1; with more threads every thread owns a film and they are summed at the end (bench baseline).
// rays/imageXY/xyzT (optional, sized for all samples) receive the per-sample records in task order; end_draws
// (optional) one per task.
// PhotonIntegrator (integrators/photonmap.cpp) for the tile driver: "nused", "maxdist", "finalgather" and a caustic map built
// from n photons (p, wo: 3 floats, alpha: 30 floats each; n == 0: no map) shot over n_paths paths.  on == 0 removes it.
int orc_set_surface_integrator(orc_ctx *c, int on, int n_used, float max_dist, int final_gather, const float *p, const float *wo,
                               const float *alpha, uint32_t n, uint32_t n_paths) {
    delete c->causticMap;
    c->causticMap = 0;
    c->causticPhotons.clear();
    c->have_si = on != 0;
    if (!on) return 0;
    for (uint32_t i = 0; i < n; ++i) {
        Photon ph;
        ph.p = v3(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
        ph.wi = v3(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]);
        memcpy(ph.alpha.c, alpha + (size_t)i * NB, sizeof(float) * NB);
        c->causticPhotons.push_back(ph);
    }
    if (n) c->causticMap = new KdTree(c->causticPhotons);
    c->si.nLookup = n_used; c->si.maxDistSquared = max_dist * max_dist; c->si.maxSpecularDepth = 5; c->si.finalGather = final_gather != 0;
    c->si.causticMap = c->causticMap; c->si.nCausticPaths = (int)n_paths;
    return 0;
}

// returns 0, or 1 when a camera ray met a surface the surface integrator's restatement does not cover (specular BSDF)
int orc_render_tasks(orc_ctx *c, const pvol_camera *cam, const pvol_film *film, const pvol_sampler *smp, const uint32_t *task_ids,
                     uint32_t n_task_ids, float *pixels, pvol_ray *rays, float *imageXY, float *xyzT, uint64_t *end_draws, int n_threads,
                     float *surfXYZ) {
    Integrator I = make_integrator(c);
    const bool wantRec = rays || imageXY || xyzT || surfXYZ;
    std::atomic<int> unsupported(0);
    std::vector<uint64_t> first(n_task_ids + 1, 0);
    for (uint32_t i = 0; i < n_task_ids; ++i) {
        int32_t w[4];
        compute_sub_window(*smp, task_ids[i], w);
        first[i + 1] = first[i] + (uint64_t)(w[1] - w[0]) * (uint64_t)(w[3] - w[2]) * smp->pixel_samples;
    }
    size_t np = pixels ? (size_t)4 * film->x_resolution * film->y_resolution : 0;
    std::atomic<uint32_t> next(0);
    std::mutex mu;
    auto worker = [&](bool shared) {
        Counters local;
        Film F;
        if (pixels) { F.f = *film; if (shared) F.pix.assign(pixels, pixels + np); else F.pix.assign(np, 0.f); }
        for (;;) {
            uint32_t i = next.fetch_add(1);
            if (i >= n_task_ids) break;
            std::vector<pvol_ray> r;
            std::vector<float> xy, xt, sx;
            TileRecords rec = {&r, &xy, &xt, (surfXYZ && c->have_si) ? &sx : 0};
            bool sup = true;
            uint64_t d = render_task(I, *cam, *smp, task_ids[i], pixels ? &F : 0, wantRec ? &rec : 0, &local, c->have_si ? &c->si : 0, &sup);
            if (!sup) unsupported = 1;
            if (surfXYZ && c->have_si) memcpy(surfXYZ + 3 * first[i], sx.data(), sizeof(float) * sx.size());
            if (end_draws) end_draws[i] = d;
            if (rays) memcpy(rays + first[i], r.data(), sizeof(pvol_ray) * r.size());
            if (imageXY) memcpy(imageXY + 2 * first[i], xy.data(), sizeof(float) * xy.size());
            if (xyzT) memcpy(xyzT + 4 * first[i], xt.data(), sizeof(float) * xt.size());
        }
        std::lock_guard<std::mutex> g(mu);
        c->ctr.add(local);
        if (pixels) { if (shared) memcpy(pixels, F.pix.data(), sizeof(float) * np); else for (size_t k = 0; k < np; ++k) pixels[k] += F.pix[k]; }
    };
    if (n_threads <= 1) worker(true);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t) th.emplace_back(worker, false);
        for (auto &t : th) t.join();
    }
    return unsupported.load();
}

}  // extern "C"
