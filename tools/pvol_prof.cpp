// tools/pvol_prof.cpp -- minimal native driver of the C ABI for rocprofv3 counter passes
// (`rocprofv3 --pmc ... -- ./tools/pvol_prof DIR`): the profiler's counter mode is unreliable under a
// Python/torch process, so the same march+gather launch is reproduced here from plain files.
//
// DIR holds raw little-endian files written by tools/make_prof_inputs.py:
//   scene.bin   pvol_scene image (fixed part) + light/triangle/material arrays
//   params.bin  pvol_params
//   photons.bin u32 n, then p[3n], wi[3n], alpha[30n] (f32)
//   rays.bin    u32 n_rays, u32 n_streams, pvol_ray[n_rays], pvol_stream[n_streams]
// usage: pvol_prof DIR [launches]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../include/pvol.h"

static std::vector<unsigned char> slurp(const std::string &path) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> b((size_t)n);
    if (n && fread(b.data(), 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read %s\n", path.c_str()); exit(2); }
    fclose(f);
    return b;
}
#define CK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, pvol_strerror(rc_)); return 1; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: pvol_prof DIR [launches]\n"); return 64; }
    std::string dir = argv[1];
    int launches = argc > 2 ? atoi(argv[2]) : 1;
    std::vector<unsigned char> sb = slurp(dir + "/scene.bin"), pb = slurp(dir + "/params.bin"), phb = slurp(dir + "/photons.bin"),
                               rb = slurp(dir + "/rays.bin");
    pvol_scene scene;
    memcpy(&scene, sb.data(), sizeof(scene));
    size_t off = sizeof(scene);
    scene.lights = (const pvol_light *)(sb.data() + off); off += sizeof(pvol_light) * scene.n_lights;
    scene.triangles = (const pvol_triangle *)(sb.data() + off); off += sizeof(pvol_triangle) * scene.n_triangles;
    scene.materials = (const pvol_material *)(sb.data() + off); off += sizeof(pvol_material) * scene.n_materials;
    scene.volume.density = scene.volume.kind == PVOL_VOLUME_GRID ? (const float *)(sb.data() + off) : 0;
    pvol_params params;
    memcpy(&params, pb.data(), sizeof(params));
    uint32_t nPh;
    memcpy(&nPh, phb.data(), 4);
    const float *pp = (const float *)(phb.data() + 4), *pw = pp + 3 * (size_t)nPh, *pa = pw + 3 * (size_t)nPh;
    uint32_t nRays, nStreams;
    memcpy(&nRays, rb.data(), 4);
    memcpy(&nStreams, rb.data() + 4, 4);
    const pvol_ray *rays = (const pvol_ray *)(rb.data() + 8);
    const pvol_stream *streams = (const pvol_stream *)(rb.data() + 8 + sizeof(pvol_ray) * (size_t)nRays);

    pvol_ctx *ctx = 0;
    CK(pvol_create(&params, &ctx));
    CK(pvol_set_scene(ctx, &scene));
    CK(pvol_upload_photons(ctx, pp, pw, pa, nPh));
    pvol_ray *dRays; pvol_stream *dStreams; float *dOut;
    HK(hipMalloc(&dRays, sizeof(pvol_ray) * (size_t)nRays));
    HK(hipMalloc(&dStreams, sizeof(pvol_stream) * (size_t)nStreams));
    HK(hipMalloc(&dOut, sizeof(float) * 4 * (size_t)nRays));
    HK(hipMemcpy(dRays, rays, sizeof(pvol_ray) * (size_t)nRays, hipMemcpyHostToDevice));
    HK(hipMemcpy(dStreams, streams, sizeof(pvol_stream) * (size_t)nStreams, hipMemcpyHostToDevice));
    for (int i = 0; i < launches; ++i) CK(pvol_li_batch_device(ctx, dRays, nRays, dStreams, nStreams, PVOL_OUT_XYZ, dOut, 0, 0));
    HK(hipDeviceSynchronize());
    double ms = 0; uint64_t n = 0;
    CK(pvol_kernel_time_ms(ctx, &ms, &n, 0));
    std::vector<float> out(4 * (size_t)nRays);
    HK(hipMemcpy(out.data(), dOut, sizeof(float) * out.size(), hipMemcpyDeviceToHost));
    double sum = 0;
    for (size_t i = 0; i < nRays; ++i) sum += out[4 * i] + out[4 * i + 1] + out[4 * i + 2];
    printf("{\"rays\": %u, \"streams\": %u, \"photons\": %u, \"launches\": %llu, \"kernel_avg_ms\": %.4f, \"msamples_per_s\": %.4f, \"checksum\": %.6f}\n",
           nRays, nStreams, nPh, (unsigned long long)n, ms, nRays / (ms * 1e-3) / 1e6, sum);
    pvol_destroy(ctx);
    return 0;
}
