/*
 * pvol.h -- C ABI of the MI355X-native volumetric photon-mapping hot path.
 *
 * This is the drop-in boundary for ONE path of the pbrt-v2 fork piwell/CS348B-pbrt:
 *   core/photonshooter.cpp   (volume-photon shooting, PhotonShooter::Preprocess)
 *   integrators/photonvolume.cpp (PhotonVolumeIntegrator::Li / Transmittance / LPhoton)
 * Every entry point below names the reference interface (file:line under the reference
 * tree) it replaces.  Plain pointers and sizes only; no C++/torch types.  All functions
 * return PVOL_OK (0) or a negative pvol_status; they never abort, never print, and fail
 * loudly (PVOL_E_NO_DEVICE) instead of falling back to a CPU path when no HIP device
 * is usable.
 *
 * Conventions
 *   - Spectra are 30 fp32 bins, 400..700 nm (core/spectrum.h:44-46).  The two tag floats
 *     of the reference's 128-byte Spectrum are not carried: `lambda` is re-derived by
 *     extractLambda() (core/spectrum.h:266-279), `intensity` is never read on the path.
 *   - Matrices are row-major 4x4, m[r*4+c] == Transform::m.m[r][c] (core/transform.h).
 *   - One MT19937 stream per render tile (renderers/samplerrenderer.cpp:73).  A batch
 *     groups rays by stream; rays of one stream are consumed in array order, exactly as
 *     the reference's tile loop consumes its RNG.
 */
#ifndef PVOL_H
#define PVOL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PVOL_NBINS 30
#define PVOL_MT_N 624
#define PVOL_ABI_VERSION 3

typedef enum pvol_status {
    PVOL_OK = 0,
    PVOL_E_INVALID = -1,      /* bad argument / inconsistent sizes            */
    PVOL_E_NO_DEVICE = -2,    /* no usable HIP device / HIP runtime error     */
    PVOL_E_NO_SCENE = -3,     /* pvol_set_scene has not succeeded yet         */
    PVOL_E_NO_MEMORY = -4,
    PVOL_E_UNSUPPORTED = -5,  /* volume/light/material kind out of scope      */
    PVOL_E_LIMIT = -6,        /* a ray needs more march steps than the kernel's LDS plan */
    PVOL_E_SHOOT_FAILED = -7  /* "Unable to store enough photons" (photonshooter.cpp:285-299) */
} pvol_status;

typedef struct pvol_spectrum { float c[PVOL_NBINS]; } pvol_spectrum;

/* ---- scene: a flattened POD copy of what the path reads from Scene ------------------ */

typedef enum pvol_volume_kind {
    PVOL_VOLUME_NONE = 0,
    PVOL_VOLUME_HOMOGENEOUS = 1,  /* volumes/homogeneous.h:43-89 */
    PVOL_VOLUME_GRID = 2,         /* volumes/volumegrid.{h,cpp}  */
    PVOL_VOLUME_RAINBOW = 3       /* volumes/rainbow.{h,cpp} (homogeneous + rainbowReflection) */
} pvol_volume_kind;

typedef struct pvol_volume {
    int32_t kind;
    float extent_min[3], extent_max[3]; /* BBox extent in volume space                   */
    float world_to_volume[16];          /* WorldToVolume.m                               */
    float volume_to_world[16];          /* its inverse (WorldBound: homogeneous.h:57-59) */
    pvol_spectrum sigma_a, sigma_s, le;
    float g;                            /* HG asymmetry (core/volume.cpp:150-154)        */
    int32_t nx, ny, nz;                 /* grid only                                     */
    const float *density;               /* grid only: nx*ny*nz floats, x fastest (volumegrid.h:60-65) */
} pvol_volume;

typedef enum pvol_light_kind {
    PVOL_LIGHT_POINT = 0,   /* lights/point.cpp   */
    PVOL_LIGHT_SPOT = 1,    /* lights/spot.cpp    */
    PVOL_LIGHT_DISTANT = 2  /* lights/distant.cpp */
} pvol_light_kind;

typedef struct pvol_light {
    int32_t kind;
    float pos[3];               /* lightPos (point, spot)                                 */
    float dir[3];               /* lightDir, already normalised (distant)                 */
    float light_to_world[16];
    float world_to_light[16];
    pvol_spectrum intensity;    /* I (point, spot) or L (distant)                         */
    float cos_total_width;      /* spot.cpp:45                                            */
    float cos_falloff_start;    /* spot.cpp:46                                            */
} pvol_light;

typedef enum pvol_material_kind {
    PVOL_MATERIAL_MATTE = 0,  /* materials/matte.cpp:42-63, sigma == 0 => Lambertian      */
    PVOL_MATERIAL_GLASS = 1   /* materials/glass.cpp:42-59 + dispersive SpecularTransmission */
} pvol_material_kind;

typedef struct pvol_material {
    int32_t kind;
    pvol_spectrum kd;   /* matte: Kd (already clamped >= 0)                               */
    pvol_spectrum kr;   /* glass: Kr                                                      */
    pvol_spectrum kt;   /* glass: Kt                                                      */
    float ior;          /* glass "index"                                                  */
    float vn;           /* glass Abbe number; > 0 => dispersive (materials/glass.h:57)    */
} pvol_material;

typedef struct pvol_triangle {
    float p[3][3];          /* world-space vertices p1,p2,p3 (shapes/trianglemesh.cpp:70-71) */
    int32_t material;       /* index into materials[]                                     */
    int32_t flip_normal;    /* ReverseOrientation ^ TransformSwapsHandedness (core/diffgeom.cpp:52-53) */
} pvol_triangle;

/* Shape "sphere" (shapes/sphere.cpp), intersected analytically in object space exactly as Sphere::Intersect / IntersectP do
 * (:59-157, :160-216); the fields are what the Sphere constructor stores (:41-49).  At most 8 per scene. */
typedef struct pvol_sphere {
    float object_to_world[16];  /* ObjectToWorld.m                                            */
    float world_to_object[16];  /* WorldToObject.m                                            */
    float radius;
    float z_min, z_max;         /* clamped to [-radius, radius] and ordered                   */
    float theta_min, theta_max; /* acosf(Clamp(z/radius, -1, 1)) of z_min, z_max              */
    float phi_max;              /* radians                                                    */
    int32_t material;           /* index into materials[]                                     */
    int32_t flip_normal;        /* ReverseOrientation ^ TransformSwapsHandedness              */
} pvol_sphere;

typedef struct pvol_scene {
    pvol_volume volume;
    uint32_t n_lights;
    const pvol_light *lights;
    uint32_t n_triangles;
    const pvol_triangle *triangles;
    uint32_t n_materials;
    const pvol_material *materials;
    float world_min[3], world_max[3];  /* Scene::WorldBound(): geometry U volume (core/scene.cpp:59-60) */
    pvol_spectrum cie_x, cie_y, cie_z; /* SampledSpectrum::X/Y/Z bin averages (core/spectrum.h:370-381) */
    float xyz_scale;                   /* (700-400)/(CIE_Y_integral*30)  (core/spectrum.h:427-428)     */
    uint32_t n_spheres;                /* ABI 3 */
    const pvol_sphere *spheres;
} pvol_scene;

/* ---- integrator / shooter parameters (CreatePhotonVolumeIntegrator photonvolume.cpp:224-229,
 *      CreatePhotonShooter photonshooter.cpp:529-548) ---------------------------------- */
typedef struct pvol_params {
    float step_size;            /* VolumeIntegrator "stepsize"  (default 1)               */
    int32_t n_used;             /* VolumeIntegrator "nused"     (default 250)             */
    float max_dist;             /* VolumeIntegrator "maxdist"   (default 0.1)             */
    uint32_t n_volume_photons;  /* VolumeIntegrator "volumephotons" (default 0)           */
    float shooter_step_size;    /* SurfaceIntegrator "stepsize" (default 0.1)             */
    int32_t max_photon_depth;   /* SurfaceIntegrator "maxphotondepth" (default 5)         */
    uint32_t n_caustic_photons; /* SurfaceIntegrator "causticphotons" (default 20000)     */
    uint32_t n_indirect_photons;/* SurfaceIntegrator "indirectphotons" (default 10000)    */
    int32_t final_gather;       /* SurfaceIntegrator "finalgather" (default true)         */
    int32_t device;             /* HIP device ordinal                                     */
    float grid_cell_scale;      /* 0 = auto; else photon-grid cell edge as a multiple of the auto choice */
    uint32_t keep_surface_photons; /* 1: pvol_preprocess also KEEPS the caustic / direct / indirect / radiance photons it deposits
                                      (photonshooter.cpp:148-189); 0: they are only counted (their counts steer the shooting)   */
    uint32_t reserved[6];
} pvol_params;

/* ---- ray batches --------------------------------------------------------------------- */

/* One camera ray handed to Li() (PhotonVolumeIntegrator::Li, photonvolume.cpp:112-222). */
typedef struct pvol_ray {
    float o[3];
    float mint;
    float d[3];
    float maxt;
    float time;
    float scatter_u;   /* sample->oneD[scatterSampleOffset][0]  (photonvolume.cpp:135)   */
    uint32_t rng_skip; /* draws of this ray's stream the CALLER consumed since the previous
                          ray of the stream (sampler + surface integrator), skipped before Li */
    uint32_t flags;    /* reserved, 0                                                     */
} pvol_ray;            /* 48 bytes                                                        */

/* One MT19937 stream == one render tile (samplerrenderer.cpp:73 `RNG rng(taskNum)`).     */
typedef struct pvol_stream {
    uint32_t seed;        /* RNG(seed)                                                    */
    uint32_t first_ray;   /* rays [first_ray, first_ray+n_rays) of the batch, in order    */
    uint32_t n_rays;
    uint32_t reserved;
    uint64_t start_draw;  /* RandomUInt() calls already made on this stream before the batch */
    uint64_t end_draw;    /* OUT: calls made after the last ray of the batch               */
} pvol_stream;            /* 32 bytes                                                     */

typedef enum pvol_output_kind {
    PVOL_OUT_SPECTRAL = 0, /* per ray: Lv[30] then T[30]          (60 floats)             */
    PVOL_OUT_XYZ = 1       /* per ray: Lv as X,Y,Z then T.y()     (4 floats)              */
} pvol_output_kind;

typedef struct pvol_stats {
    uint64_t n_rays;           /* Li() calls                                              */
    uint64_t n_steps;          /* march steps == photon lookups (non-rainbow)             */
    uint64_t n_tested;         /* photons distance-tested by the gather                   */
    uint64_t n_kept;           /* photons that entered a flux sum                         */
    uint64_t n_lookups_lt10;   /* lookups that found < 10 photons (photonvolume.cpp:83)   */
    uint64_t n_shadow_unoccluded;
    uint64_t n_guess_retries;  /* lookups redone exactly: predicted radius held < k photons (lphoton), or not served by the bucket plan (group kernel) */
    uint64_t cy_search;        /* s_memtime cycles summed over wavefronts: candidate search ...          */
    uint64_t cy_select;        /* ... k-selection ...                                                     */
    uint64_t cy_flux;          /* ... flux sum ...                                                        */
    uint64_t cy_total;         /* ... whole kernel                                                        */
    uint64_t group_guess_failed;  /* li_group_kernel: lookups whose guessed search radius was too small or too large ...   */
    uint64_t group_plan_skipped;  /* ... whose shared photon bucket overflowed (or k outside the plan) ...                 */
    uint64_t cy_fallback;         /* ... and the cycles of li_fixup_kernel, whose exact wave-cooperative lookups serve what the plan handed over (n_guess_retries counts them) */
    uint64_t group_deferred_overflow;  /* handed-over lookups whose last attempt ended in a bucket overflow ...                */
    uint64_t group_deferred_too_few;   /* ... or found fewer than k photons inside a radius below the maximum                  */
    uint64_t group_attempts;           /* shared-bucket attempts (stagings), per wavefront                                     */
} pvol_stats;

/* ---- tile driver, SURVEY 8(f)-1: the caller of Li() (LD sampler, perspective camera) and its
 * consumer (image film) on the device, so that a whole SamplerRendererTask (renderers/
 * samplerrenderer.cpp:59-157) runs without leaving HBM.  Surface radiance is NOT computed here: the
 * driver is exact for renders whose surface integrator adds nothing and draws nothing (Ls = Lvi). ---- */
#define PVOL_MAX_SAMPLE_ARRAYS 16
#define PVOL_FILTER_TABLE_SIZE 16
#define PVOL_MAX_PIXEL_SAMPLES 1024

typedef struct pvol_camera {     /* PerspectiveCamera (cameras/perspective.cpp:41-49, core/camera.cpp:83-102) */
    float raster_to_camera[16];  /* ProjectiveCamera::RasterToCamera.m, row-major                 */
    float camera_to_world[16];   /* Camera::CameraToWorld of a static camera (startTransform->m)  */
    float shutter_open, shutter_close;
    float lens_radius;           /* must be 0 (pinhole); the lens samples are still drawn          */
    float focal_distance;
} pvol_camera;

typedef struct pvol_film {       /* ImageFilm with the full-frame crop window (film/image.cpp:37-75) */
    int32_t x_resolution, y_resolution;
    float filter_xwidth, filter_ywidth;              /* Filter::xWidth, yWidth                     */
    float filter_table[PVOL_FILTER_TABLE_SIZE * PVOL_FILTER_TABLE_SIZE];  /* ImageFilm::filterTable */
} pvol_film;

typedef struct pvol_sampler {    /* LDSampler (samplers/lowdiscrepancy.cpp:40-80) + the Sample layout  */
    int32_t x_start, x_end, y_start, y_end;          /* Film::GetSampleExtent (film/image.cpp:157-166) */
    uint32_t pixel_samples;                          /* power of two, <= PVOL_MAX_PIXEL_SAMPLES    */
    uint32_t n_tasks;                                /* taskCount of SamplerRenderer::Render (samplerrenderer.cpp:206-208) */
    uint32_t n1d_count, n2d_count;                   /* Sample::n1D / n2D as the integrators requested them */
    uint32_t n1d[PVOL_MAX_SAMPLE_ARRAYS], n2d[PVOL_MAX_SAMPLE_ARRAYS];
    uint32_t tau_index, scatter_index;               /* positions in n1d of tauSampleOffset / scatterSampleOffset (photonvolume.cpp:9-13) */
} pvol_sampler;

typedef struct pvol_render_debug {   /* optional DEVICE buffers that receive the intermediate records, sized for all samples of the call */
    pvol_ray *d_rays;            /* camera rays after Scene::Intersect clipped maxt, tile-major    */
    float *d_image_xy;           /* 2 floats per sample: CameraSample::imageX, imageY              */
    float *d_xyz;                /* 4 floats per sample: Lvi as X,Y,Z and T.y()                    */
    pvol_stream *d_streams;      /* one per task of the call, end_draw filled                      */
    float *d_surf_xyz;           /* 3 floats per sample: the surface integrator's Li as X,Y,Z      */
                                 /* (only with pvol_set_surface_integrator enabled; may be NULL)   */
} pvol_render_debug;

/* Surface integrator in front of the volume term: PhotonIntegrator::Li
 * (integrators/photonmap.cpp:154-319) for camera rays that end on MATTE surfaces, without an indirect
 * map (`indirectphotons 0`, both BASELINE scenes): UniformSampleAllLights over delta lights
 * (core/integrator.cpp:47-79, :117-174), the caustic estimate LPhoton (photonmap.cpp:62-108) and the
 * RNG traffic of the rest of that Li() (2 x BSDF::rho, 2 x BSDFSample, one draw per unoccluded light). */
typedef struct pvol_surface_params {
    int32_t n_used;              /* "nused" (photonmap.cpp:340), lookup size of the caustic estimate */
    float max_dist;              /* "maxdist" (photonmap.cpp:345)                                    */
    int32_t max_specular_depth;  /* "maxspeculardepth" (photonmap.cpp:341): > 1 costs 6 draws per hit */
    int32_t final_gather;        /* "finalgather": without an indirect map it changes nothing              */
    uint32_t n_caustic_paths;    /* nCausticPaths (photonshooter.cpp:215) of the map passed along    */
    int32_t use_preprocess_store;/* 1: take the caustic photons (and path count) the last pvol_preprocess kept
                                    (params.keep_surface_photons) instead of the arrays passed        */
    uint32_t n_indirect_photons; /* photons in the integrator's INDIRECT map (0 with `indirectphotons 0`, both BASELINE
                                    scenes).  > 0 makes Li() gather (photonmap.cpp:183-309): PVOL_E_UNSUPPORTED            */
    uint32_t reserved[1];
} pvol_surface_params;

typedef struct pvol_ctx pvol_ctx;

/* Version of this ABI (PVOL_ABI_VERSION). */
int pvol_abi_version(void);
/* Static string for a pvol_status. */
const char *pvol_strerror(int status);
/* Number of usable HIP devices (0 when none; never a CPU fallback). */
int pvol_device_count(void);

/* Fill *p with the reference defaults (photonvolume.cpp:224-229, photonshooter.cpp:529-548). */
void pvol_default_params(pvol_params *p);

/* Replaces CreatePhotonVolumeIntegrator + CreatePhotonShooter (photonvolume.cpp:224-229,
 * photonshooter.cpp:529-548; constructed from core/api.cpp:572-586,1225-1230). */
int pvol_create(const pvol_params *params, pvol_ctx **out);
/* Replaces ~PhotonShooter / ~PhotonVolumeIntegrator (photonshooter.cpp:423-430). */
void pvol_destroy(pvol_ctx *ctx);

/* Copies the flattened scene to the device (what Li()/followPhoton read through
 * `const Scene *`: core/scene.h:42-73) and, beyond 64 triangles, builds the triangle hierarchy there (pvol_get_accel_info).
 * May be called again to replace the scene; a rejected scene leaves the previous one in place.
 * Limits (PVOL_E_UNSUPPORTED): 8 lights, 2^24 triangles, 8 spheres, 8 materials; PVOL_E_INVALID: a material index out of
 * range, a non-finite vertex, a sphere radius <= 0. */
int pvol_set_scene(pvol_ctx *ctx, const pvol_scene *scene);

/* Installs a volume photon map computed elsewhere (e.g. by the reference's own
 * PhotonShooter) and builds the device search structure; replaces
 * `volumeMap = new KdTree<Photon>(volumePhotons)` (photonshooter.cpp:502-503).
 * p, wi: 3*n floats (xyz interleaved); alpha: 30*n floats.  n == 0 clears the map
 * (a NULL map is legal, photonvolume.cpp:69). */
int pvol_upload_photons(pvol_ctx *ctx, const float *p, const float *wi,
                        const float *alpha, uint32_t n);

/* Replaces PhotonShooter::Preprocess (photonshooter.cpp:457-526): shoots volume photons
 * on the device with `n_tasks` virtual PhotonShootingTasks (task t uses RNG(31*t) and its
 * own Halton permutation, photonshooter.cpp:235,243), merges them in task order per block
 * round (photonshooter.cpp:280-351) and builds the search structure. */
int pvol_preprocess(pvol_ctx *ctx, uint32_t n_tasks);

/* The same with `block_paths` (1 .. 4096) paths per virtual task and round instead of PhotonShootingTask::Run's 4096
 * (photonshooter.cpp:247): the merge rule is unchanged (task order per round, volume photons divided by the running nshot,
 * :280-351), only finer.  What it is for: a store stops growing at the first merge that fills it, so what is stored overshoots
 * the request by up to one round = n_tasks x block_paths paths' worth of photons -- with 4096-path blocks a prism scene cannot
 * use more than a few hundred tasks (C3: 256 waves on a 1024-SIMD chip, and still 6.7 M stored for 4 M asked); with small blocks
 * thousands of tasks keep the chip busy at the same overshoot.  Not the reference's block size, hence not its photon set:
 * statistically the same map (tests/test_gpu_shooter.py: counts, flux, spatial and spectral histograms), and still equal to the
 * oracle photon for photon when the oracle runs the same block size. */
int pvol_preprocess_blocks(pvol_ctx *ctx, uint32_t n_tasks, uint32_t block_paths);

/* Work counters of the last pvol_preprocess (the figures SURVEY 6 reports for the reference shooter):
 * out[12] = paths, followPhoton calls, calls ending without a surface hit, transmittance-march steps,
 * volume interactions, absorbed, stored volume / caustic / direct / indirect photons, spectral-split
 * children, nshot. */
int pvol_get_shoot_stats(pvol_ctx *ctx, uint64_t *out12);

/* Wall-clock seconds of the last pvol_preprocess: out[0] the shooting rounds and merges
 * (PhotonShootingTask::Run, photonshooter.cpp:232-357), out[1] the device search-structure build that replaces
 * the kd-tree construction (photonshooter.cpp:502-503, core/kdtree.h:100-147). */
int pvol_get_preprocess_seconds(pvol_ctx *ctx, double *out2);

/* The triangle accelerator of the scene set last (replaces CreateBVHAccelerator / `new BVHAccel`, core/api.cpp:1284-1290,
 * accelerators/bvh.cpp:301-389): scenes of up to 64 triangles are scanned linearly out of the scalar cache; larger ones
 * (up to 2^24 triangles) get a linear BVH built on the device inside pvol_set_scene.  Closest / any hit results do not
 * depend on which one serves them (smallest t; of several triangles at the same t the one latest in the scene's order).
 * out[0] = triangles in the hierarchy (0: linear scan), out[1] = device build time in milliseconds. */
int pvol_get_accel_info(pvol_ctx *ctx, double *out2);

/* The surface stores of the last pvol_preprocess (params.keep_surface_photons = 1): what PhotonShooter::Preprocess hands to
 * its caustic / direct / indirect kd-trees (photonshooter.cpp:495-503), in the reference's merge order (:303-327).
 * kind: 0 caustic, 1 direct, 2 indirect.  n_paths = nCausticPaths / nDirectPaths / nIndirectPaths.  p, wo: 3 floats,
 * alpha: 30 floats per photon (NOT divided by the path count: the surface estimate divides at lookup, photonmap.cpp:89). */
int pvol_surface_photon_count(pvol_ctx *ctx, int kind, uint32_t *n, uint32_t *n_paths);
int pvol_download_surface_photons(pvol_ctx *ctx, int kind, float *p, float *wo, float *alpha, uint32_t capacity);
/* Radiance photons (photonshooter.cpp:182-189): position, normal facing the arriving photon, rho_r, rho_t (30 floats each). */
int pvol_radiance_photon_count(pvol_ctx *ctx, uint32_t *n);
int pvol_download_radiance_photons(pvol_ctx *ctx, float *p, float *n, float *rho_r, float *rho_t, uint32_t capacity);

/* Number of photons in the current volume map. */
int pvol_photon_count(pvol_ctx *ctx, uint32_t *n);
/* Copies the current map back (same layout as pvol_upload_photons); capacity in photons. */
int pvol_download_photons(pvol_ctx *ctx, float *p, float *wi, float *alpha, uint32_t capacity);

/* Replaces PhotonVolumeIntegrator::Li (photonvolume.cpp:112-222) for a batch of rays
 * grouped into MT19937 streams.  Host pointers; `out` has n_rays*60 (SPECTRAL) or
 * n_rays*4 (XYZ) floats; `draws` (optional) receives per ray the number of RandomUInt
 * calls Li() made (4+6n+n+r+u).  streams[i].end_draw is written.
 * The streams must partition the ray array in order: streams[0].first_ray == 0 and
 * streams[i+1].first_ray == streams[i].first_ray + streams[i].n_rays (PVOL_E_INVALID otherwise). */
int pvol_li_batch(pvol_ctx *ctx, const pvol_ray *rays, uint32_t n_rays,
                  pvol_stream *streams, uint32_t n_streams,
                  int output_kind, float *out, uint32_t *draws);

/* Same, all pointers are DEVICE pointers and the work is enqueued on `hip_stream`
 * (a hipStream_t, NULL = default stream) without synchronising. */
int pvol_li_batch_device(pvol_ctx *ctx, const pvol_ray *d_rays, uint32_t n_rays,
                         pvol_stream *d_streams, uint32_t n_streams,
                         int output_kind, float *d_out, uint32_t *d_draws,
                         void *hip_stream);

/* Device-side error flags of the batches enqueued so far (pvol_li_batch_device, pvol_render_tasks_device): waits
 * for the device and returns PVOL_E_LIMIT if a ray needed more march steps than the kernels' record/LDS plan holds
 * (such a ray's output is zero, never a guess), clearing the flag.  The host entry points check this themselves. */
int pvol_check_errors(pvol_ctx *ctx);

/* Single call with the caller's live RNG (the per-sample shim behind
 * VolumeIntegrator::Li): mt[624] / *mti are core/rng.h:57-58 narrowed to 32 bits and are
 * advanced exactly as the reference would advance them.  Lv, T: 30 floats each.
 * Thread-safe: concurrent calls on one context (every SamplerRendererTask thread calls
 * VolumeIntegrator::Li, samplerrenderer.cpp:247) are serialised inside the library, as are all
 * other host-pointer entry points. */
int pvol_li(pvol_ctx *ctx, const pvol_ray *ray, uint32_t *mt, int32_t *mti,
            float *Lv, float *T);

/* Replaces PhotonVolumeIntegrator::Transmittance with sample == NULL
 * (photonvolume.cpp:15-30): step = 4*stepSize, offset = one RandomFloat per ray.
 * Rays are grouped into streams like pvol_li_batch; out: 30 floats per ray. */
int pvol_transmittance_batch(pvol_ctx *ctx, const pvol_ray *rays, uint32_t n_rays,
                             pvol_stream *streams, uint32_t n_streams, float *out);

/* Work counters accumulated since the last reset (feeds the algorithmic-bytes formula
 * B_lookup = 20*V + 132*K of SURVEY 8(d)). */
int pvol_get_stats(pvol_ctx *ctx, pvol_stats *out, int reset);
/* Enable/disable counter collection in the kernels (off by default: atomics cost time). */
int pvol_enable_stats(pvol_ctx *ctx, int on);

/* Average device time of the dominant (march+gather) kernel over the launches since
 * the last reset, measured with HIP events on the launch stream. */
int pvol_kernel_time_ms(pvol_ctx *ctx, double *avg_ms, uint64_t *launches, int reset);
/* Name of the march+gather kernel the last batch was dispatched to ("li_group_kernel", "li_par_kernel",
 * "li_replay_kernel", "li_seq_kernel"; "" before the first batch): the kernel the time above belongs to. */
const char *pvol_march_kernel_name(pvol_ctx *ctx);

/* Per-phase device time of pvol_render_tasks_device, HIP events on the launch stream (off by default).  out[6], milliseconds
 * summed since the last reset: [0] tile pre-pass (LDSampler + camera + Scene::Intersect clip + draw count: tile_kernel),
 * [1] unused, [2] march + gather (li_group_kernel and its hand-over passes, or the slice loop incl. its RNG pass),
 * [3] surface integrator, [4] ImageFilm::AddSample, [5] unused.  What one rank of a multi-GPU render spends where
 * (bench.py --emulate-rank). */
int pvol_enable_phase_timing(pvol_ctx *ctx, int on);
int pvol_get_phase_ms(pvol_ctx *ctx, double *out6, int reset);

/* GaussianFilter::Evaluate tabulated as ImageFilm's constructor does (filters/gaussian.h:44-58,
 * film/image.cpp:57-68). */
void pvol_gaussian_filter_table(float xwidth, float ywidth, float alpha, float *table256);

/* Sampler::ComputeSubWindow (core/sampler.cpp:55-74) for task `task` of s->n_tasks: out = x0,x1,y0,y1. */
void pvol_compute_sub_window(const pvol_sampler *s, uint32_t task, int32_t out[4]);

/* Total camera samples of the given tasks (pixels of the sub-windows x pixel_samples). */
uint64_t pvol_render_sample_count(const pvol_sampler *s, const uint32_t *task_ids, uint32_t n_task_ids);

/* Runs SamplerRendererTask::Run (samplerrenderer.cpp:59-157) for the listed tasks: per task RNG(taskNum),
 * LDSampler::GetMoreSamples -> LDPixelSample per pixel, PerspectiveCamera::GenerateRayDifferential,
 * Scene::Intersect (clips the ray), PhotonVolumeIntegrator::Li, ImageFilm::AddSample.  d_pixels holds
 * x_resolution*y_resolution*4 floats (Lxyz[3], weightSum as ImageFilm::Pixel) and is ACCUMULATED into,
 * so ranks that render disjoint task sets sum their buffers (RCCL all-reduce) before pvol_film_resolve.
 * Enqueued on hip_stream; work buffers live in the context. */
int pvol_render_tasks_device(pvol_ctx *ctx, const pvol_camera *camera, const pvol_film *film,
                             const pvol_sampler *sampler, const uint32_t *task_ids, uint32_t n_task_ids,
                             float *d_pixels, const pvol_render_debug *debug, void *hip_stream);

/* Enables (sp != NULL) or disables (sp == NULL) the surface integrator for pvol_render_tasks_device:
 * SamplerRendererTask::Run then computes Ls = surface Li, and the film receives T * Ls + Lvi
 * (samplerrenderer.cpp:95-97,239-251).  p/wo/alpha: the caustic map's n photons (p 3, wo 3, alpha 30
 * floats each; n == 0: no caustic map, as when the shooter stored no caustic photon).
 * PVOL_E_UNSUPPORTED: a triangle of the scene carries a non-matte material, or an indirect photon map exists
 * (sp->n_indirect_photons > 0, or use_preprocess_store and the shooter kept indirect photons) (here), or the volume /
 * nused lie outside li_group_kernel's domain (from pvol_render_tasks_device).  A
 * lookup with more than 2048 caustic photons within maxdist of ONE hit point is counted as an error
 * (pvol_check_errors), never truncated.  PVOL_E_INVALID: use_preprocess_store without a pvol_preprocess that ran with
 * params.keep_surface_photons.  pvol_set_scene disables the surface integrator again. */
int pvol_set_surface_integrator(pvol_ctx *ctx, const pvol_surface_params *sp, const float *p, const float *wo,
                                const float *alpha, uint32_t n);

/* ---- multi-GPU frame (north_star: pixel tiles partitioned over the GPUs of a node, one RCCL reduce of the film) ----
 * Rank `rank` of `n_ranks` renders the tasks rank, rank + n_ranks, ... of SamplerRenderer::Render's task loop
 * (renderers/samplerrenderer.cpp:206-221).  out_ids may be NULL to ask for the count only. */
int pvol_partition_tasks(uint32_t n_tasks, uint32_t rank, uint32_t n_ranks, uint32_t *out_ids, uint32_t capacity, uint32_t *n_out);

/* One rank's part of a frame, enqueued on hip_stream: d_pixels (x*y*4 floats, this rank's device) is zeroed, the rank's tasks are
 * rendered into it (pvol_render_tasks_device), ONE ncclReduce(sum, root 0) over `nccl_comm` (an ncclComm_t of n_ranks ranks made by
 * the caller with RCCL: ncclCommInitRank; ignored when n_ranks == 1) adds the ranks' films -- the Gaussian filter splats across tile
 * borders (film/image.cpp:82-134), so the films are summed, not gathered -- and rank 0 resolves into d_rgb (x*y*3 floats; may be
 * NULL on the other ranks).  The photon map is replicated: every rank calls pvol_preprocess with the same task count first.
 * RCCL is bound at run time (the copy already in the process, else librccl.so.1): PVOL_E_NO_DEVICE if none is in reach. */
int pvol_render_frame_ranks(pvol_ctx *ctx, const pvol_camera *camera, const pvol_film *film, const pvol_sampler *sampler,
                            uint32_t rank, uint32_t n_ranks, void *nccl_comm, float *d_pixels, float *d_rgb, void *hip_stream);

/* ImageFilm::AddSample (film/image.cpp:78-137) for n samples: d_image_xy 2 floats, d_xyz `xyz_stride`
 * floats per sample (X,Y,Z first). */
int pvol_film_add_samples_device(pvol_ctx *ctx, const pvol_film *film, const float *d_image_xy,
                                 const float *d_xyz, uint32_t xyz_stride, uint64_t n,
                                 float *d_pixels, void *hip_stream);

/* ImageFilm::WriteRGB (film/image.cpp:178-214) without splats: d_rgb gets 3 floats per pixel. */
int pvol_film_resolve_device(pvol_ctx *ctx, const pvol_film *film, const float *d_pixels, float *d_rgb,
                             void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* PVOL_H */
