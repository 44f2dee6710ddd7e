// pvol_math.h -- device-side value layer of the hot path: float4 spectra, fp32 geometry in the
// reference's evaluation order, volume / phase / rainbow evaluation.  Included by the HIP kernels only.
#ifndef PVOL_MATH_H
#define PVOL_MATH_H
#include <hip/hip_runtime.h>
#include <math.h>

#include "pvol_dev.h"

#define LANES 64
#define MT_N 624
#define MT_M 397
#define K_PI 3.14159265358979323846f  /* core/pbrt.h:191: M_PI is a float literal */

typedef float4 f4;

// ------------------------------------------------------------------------------------------ float4 spectra
__device__ __forceinline__ f4 mk4(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ f4 operator+(f4 a, f4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ f4 operator-(f4 a, f4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ f4 operator*(f4 a, f4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ f4 operator/(f4 a, f4 b) { return make_float4(a.x / b.x, a.y / b.y, a.z / b.z, a.w / b.w); }
__device__ __forceinline__ f4 operator*(f4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ f4 operator/(f4 a, float s) { return make_float4(a.x / s, a.y / s, a.z / s, a.w / s); }
__device__ __forceinline__ f4 neg4(f4 a) { return make_float4(-a.x, -a.y, -a.z, -a.w); }
// Exp(-tau) per bin (core/spectrum.h:227-233).  __expf = v_exp_f32(x * log2 e): ~1e-6 relative for the
// optical depths met here, two instructions instead of ~15; only radiance VALUES pass through it (the
// parity budget is 1e-4 relative L2), never a geometric decision.
__device__ __forceinline__ f4 exp4(f4 a) { return make_float4(__expf(a.x), __expf(a.y), __expf(a.z), __expf(a.w)); }
// per-bin divide of radiance values by v_rcp_f32 + multiply (~1 ulp)
__device__ __forceinline__ f4 fdiv4(f4 a, f4 b) {
    return make_float4(__fdividef(a.x, b.x), __fdividef(a.y, b.y), __fdividef(a.z, b.z), __fdividef(a.w, b.w));
}
// bins 30,31 are padding: force them back to 0 after an operation that could make them non-finite
__device__ __forceinline__ f4 clean4(f4 a, int q) { if (q == 7) { a.z = 0.f; a.w = 0.f; } return a; }
__device__ __forceinline__ f4 ld4(const float *base32, int q) { return *reinterpret_cast<const f4 *>(base32 + 4 * q); }

// ---- cross-lane helpers on DPP / v_readlane / v_permlane*_swap: no LDS crossbar round trips.
// (`__shfl*` lowers to ds_bpermute_b32, ~100+ cycles of dependent latency each.)
#define DPP_QUAD_XOR1 0xB1        /* quad_perm [1,0,3,2] */
#define DPP_QUAD_XOR2 0x4E        /* quad_perm [2,3,0,1] */
#define DPP_ROW_HALF_MIRROR 0x141 /* lane i <-> 7 - i inside each group of 8 */
#define DPP_ROW_ROR8 0x128        /* lane i <- lane (i + 8) mod 16 inside each row of 16 */
template <int CTRL> __device__ __forceinline__ float dppf(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// value of a wave-uniform lane
__device__ __forceinline__ float lane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ int lane_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
// sum over the 8 lanes of a group (every lane of the group gets it)
__device__ __forceinline__ float group8_sum(float v) {
    v += dppf<DPP_QUAD_XOR1>(v);
    v += dppf<DPP_QUAD_XOR2>(v);
    v += dppf<DPP_ROW_HALF_MIRROR>(v);   // after the two quad steps every lane of a quad holds the quad sum
    return v;
}
// v[l] + v[l ^ 16] and v[l] + v[l ^ 32] in every lane: gfx950 v_permlane16_swap / v_permlane32_swap exchange
// odd/even rows (resp. halves) of two registers; with both operands equal the two results are the partners
__device__ __forceinline__ float xor16_sum(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor16_max(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ bool wave_any(bool p) { return __ballot(p) != 0ull; }
__device__ __forceinline__ float wave_max(float v) {   // every lane gets the maximum over the wave
    v = fmaxf(v, dppf<DPP_QUAD_XOR1>(v));
    v = fmaxf(v, dppf<DPP_QUAD_XOR2>(v));
    v = fmaxf(v, dppf<DPP_ROW_HALF_MIRROR>(v));
    v = fmaxf(v, dppf<DPP_ROW_ROR8>(v));
    v = xor16_max(v);
    v = xor32_max(v);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {   // every lane gets the sum over the wave (DPP + permlane swaps, no LDS)
    v += dppf<DPP_QUAD_XOR1>(v);
    v += dppf<DPP_QUAD_XOR2>(v);
    v += dppf<DPP_ROW_HALF_MIRROR>(v);
    v += dppf<DPP_ROW_ROR8>(v);
    v = xor16_sum(v);
    v = xor32_sum(v);
    return v;
}
__device__ __forceinline__ uint32_t lanes_below(uint64_t mask, int lane) {
    return (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}
// SampledSpectrum::y() (core/spectrum.h:433-439); summation order differs from the scalar loop.
__device__ __forceinline__ float spec_y(f4 c, f4 Y) {
    float p = Y.x * c.x + Y.y * c.y + Y.z * c.z + Y.w * c.w;
    p = group8_sum(p);
    return p * 300.f / (106.856895f * 30);
}
__device__ __forceinline__ bool spec_is_black(f4 c) { return !wave_any(c.x != 0.f || c.y != 0.f || c.z != 0.f || c.w != 0.f); }

// core/montecarlo.h:277-286
__device__ __forceinline__ float van_der_corput(uint32_t n, uint32_t scramble) {
    n = __brev(n);
    n ^= scramble;
    return fminf(((n >> 8) & 0xffffff) / float(1 << 24), 0x1.fffffep-1f);
}

// ------------------------------------------------------------------------------------------ geometry
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { V3 v; v.x = x; v.y = y; v.z = z; return v; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float f) { return v3(a.x * f, a.y * f, a.z * f); }
__device__ __forceinline__ V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 vdiv(V3 a, float f) { float inv = 1.f / f; return v3(a.x * inv, a.y * inv, a.z * inv); }  // geometry.h:94-98
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float len_sq(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
__device__ __forceinline__ float len(V3 a) { return sqrtf(len_sq(a)); }
__device__ __forceinline__ V3 normalize(V3 a) { return vdiv(a, len(a)); }
// core/geometry.h:477-484: double products, one rounding
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return v3(float((ay * bz) - (az * by)), float((az * bx) - (ax * bz)), float((ax * by) - (ay * bx)));
}
__device__ __forceinline__ V3 xform_point(const float *m, V3 p) {  // core/transform.h:187-201
    float xp = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    float yp = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    float zp = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    float wp = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    if (wp == 1.f) return v3(xp, yp, zp);
    float inv = 1.f / wp;
    return v3(inv * xp, inv * yp, inv * zp);
}
__device__ __forceinline__ V3 xform_vector(const float *m, V3 v) {  // core/transform.h:220-226
    return v3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
struct RayD { V3 o, d; float mint, maxt; };

// BBox::IntersectP, core/geometry.cpp:68-86
__device__ __forceinline__ bool box_intersect(const float *lo, const float *hi, V3 o, V3 d, float mint, float maxt, float *h0, float *h1) {
    float t0 = mint, t1 = maxt;
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float inv = 1.f / dd[i];
        float tn = (lo[i] - oo[i]) * inv;
        float tf = (hi[i] - oo[i]) * inv;
        if (tn > tf) { float t = tn; tn = tf; tf = t; }
        t0 = tn > t0 ? tn : t0;
        t1 = tf < t1 ? tf : t1;
        if (t0 > t1) return false;
    }
    *h0 = t0;
    *h1 = t1;
    return true;
}
__device__ __forceinline__ bool box_inside(const float *lo, const float *hi, V3 p) {  // geometry.h:404-408
    return p.x >= lo[0] && p.x <= hi[0] && p.y >= lo[1] && p.y <= hi[1] && p.z >= lo[2] && p.z <= hi[2];
}
// VolumeRegion::IntersectP (volumes/homogeneous.h:60-63, volumegrid.h:52-55)
__device__ __forceinline__ bool vol_intersect(const DevScene &S, const RayD &r, float *t0, float *t1) {
    V3 o = xform_point(S.w2v, r.o), d = xform_vector(S.w2v, r.d);
    return box_intersect(S.extLo, S.extHi, o, d, r.mint, r.maxt, t0, t1);
}
// shapes/trianglemesh.cpp:211-243 (any hit) for one triangle
__device__ __forceinline__ bool tri_hit_v(V3 p1, V3 p2, V3 p3, const RayD &ray) {
    V3 e1 = p2 - p1, e2 = p3 - p1;
    V3 s1 = cross(ray.d, e2);
    float divisor = dot(s1, e1);
    if (divisor == 0.f) return false;
    float invDivisor = 1.f / divisor;
    V3 s = ray.o - p1;
    float b1 = dot(s, s1) * invDivisor;
    if (b1 < 0.f || b1 > 1.f) return false;
    V3 s2 = cross(s, e1);
    float b2 = dot(ray.d, s2) * invDivisor;
    if (b2 < 0.f || b1 + b2 > 1.f) return false;
    float t = dot(e2, s2) * invDivisor;
    if (t < ray.mint || t > ray.maxt) return false;
    return true;
}
__device__ __forceinline__ bool tri_hit(const DevTri &tr, const RayD &ray) {
    return tri_hit_v(v3(tr.p1[0], tr.p1[1], tr.p1[2]), v3(tr.p2[0], tr.p2[1], tr.p2[2]), v3(tr.p3[0], tr.p3[1], tr.p3[2]), ray);
}
// shapes/trianglemesh.cpp:116-160 (closest hit: t reported) for one triangle
__device__ __forceinline__ bool tri_closest(const DevTri &tr, V3 o, V3 d, float mint, float maxt, float *tHit) {
    V3 p1 = v3(tr.p1[0], tr.p1[1], tr.p1[2]), p2 = v3(tr.p2[0], tr.p2[1], tr.p2[2]), p3 = v3(tr.p3[0], tr.p3[1], tr.p3[2]);
    V3 e1 = p2 - p1, e2 = p3 - p1;
    V3 s1 = cross(d, e2);
    float divisor = dot(s1, e1);
    if (divisor == 0.f) return false;
    float invDivisor = 1.f / divisor;
    V3 s = o - p1;
    float b1 = dot(s, s1) * invDivisor;
    if (b1 < 0.f || b1 > 1.f) return false;
    V3 s2 = cross(s, e1);
    float b2 = dot(d, s2) * invDivisor;
    if (b2 < 0.f || b1 + b2 > 1.f) return false;
    float t = dot(e2, s2) * invDivisor;
    if (t < mint || t > maxt) return false;
    *tHit = t;
    return true;
}
#include "pvol_bvh_dev.h"
#include "pvol_sphere_dev.h"
// Scene::IntersectP (core/scene.h:57-61): one triangle per lane; a large scene walks its hierarchy (wave-uniform ray)
__device__ __forceinline__ bool scene_occluded(const DevScene &S, const RayD &ray, int lane) {
    if (S.nSpheres && spheres_occluded(S, ray.o, ray.d, ray.mint, ray.maxt)) return true;
    if (S.bvhNodes) return bvh_occluded(S, ray.o, ray.d, ray.mint, ray.maxt);
    bool hit = false;
    for (int base = 0; base < S.nTris; base += LANES) {
        int t = base + lane;
        bool h = (t < S.nTris) && tri_hit(S.tris[t], ray);
        hit = hit || wave_any(h);
    }
    return hit;
}

// ------------------------------------------------------------------------------------------ volume
__device__ __forceinline__ float lerpf(float t, float a, float b) { return (1.f - t) * a + t * b; }
__device__ __forceinline__ float grid_D(const DevScene &S, int x, int y, int z) {  // volumegrid.h:60-65
    x = min(max(x, 0), S.nx - 1);
    y = min(max(y, 0), S.ny - 1);
    z = min(max(z, 0), S.nz - 1);
    return S.density[(size_t)z * S.nx * S.ny + (size_t)y * S.nx + x];
}
// VolumeGridDensity::Density, volumes/volumegrid.cpp:39-57
__device__ float grid_density(const DevScene &S, V3 Pobj) {
    if (!box_inside(S.extLo, S.extHi, Pobj)) return 0.f;
    float vx_ = (Pobj.x - S.extLo[0]) / (S.extHi[0] - S.extLo[0]);
    float vy_ = (Pobj.y - S.extLo[1]) / (S.extHi[1] - S.extLo[1]);
    float vz_ = (Pobj.z - S.extLo[2]) / (S.extHi[2] - S.extLo[2]);
    vx_ = vx_ * S.nx - .5f;
    vy_ = vy_ * S.ny - .5f;
    vz_ = vz_ * S.nz - .5f;
    int vx = (int)floorf(vx_), vy = (int)floorf(vy_), vz = (int)floorf(vz_);
    float dx = vx_ - vx, dy = vy_ - vy, dz = vz_ - vz;
    float d00 = lerpf(dx, grid_D(S, vx, vy, vz), grid_D(S, vx + 1, vy, vz));
    float d10 = lerpf(dx, grid_D(S, vx, vy + 1, vz), grid_D(S, vx + 1, vy + 1, vz));
    float d01 = lerpf(dx, grid_D(S, vx, vy, vz + 1), grid_D(S, vx + 1, vy, vz + 1));
    float d11 = lerpf(dx, grid_D(S, vx, vy + 1, vz + 1), grid_D(S, vx + 1, vy + 1, vz + 1));
    float d0 = lerpf(dy, d00, d10);
    float d1 = lerpf(dy, d01, d11);
    return lerpf(dz, d0, d1);
}
// density factor of sigma_a/sigma_s/sigma_t/Lve at a world point: homogeneous.h:64-75 (Inside ? 1 : 0),
// core/volume.h:81-92 (Density)
__device__ __forceinline__ float vol_density(const DevScene &S, V3 p) {
    V3 q = xform_point(S.w2v, p);
    if (S.volKind == PVOL_VOLUME_GRID) return grid_density(S, q);
    return box_inside(S.extLo, S.extHi, q) ? 1.f : 0.f;
}
// HG phase (core/volume.cpp:150-154); homogeneous p() also tests Inside (homogeneous.h:76-79)
__device__ __forceinline__ float phase_hg(V3 w, V3 wp, float g) {
    // g == 0: (1 - 0) / powf(1, 1.5f) is exactly 1, so the reference's value is 1/(4 pi) bit for bit
    if (g == 0.f) return 1.f / (4.f * K_PI);
    float costheta = dot(w, wp);
    return 1.f / (4.f * K_PI) * (1.f - g * g) / powf(1.f + g * g - 2.f * g * costheta, 1.5f);
}
__device__ __forceinline__ float vol_phase(const DevScene &S, V3 p, V3 wi, V3 wo) {
    if (S.volKind != PVOL_VOLUME_GRID && !box_inside(S.extLo, S.extHi, xform_point(S.w2v, p))) return 0.f;
    return phase_hg(wi, wo, S.g);
}
// tau(): homogeneous.h:80-84 analytic; DensityRegion::tau core/volume.cpp:296-310 stepped.
__device__ f4 vol_tau(const DevScene &S, const RayD &r, float stepSize, float u, f4 sigT) {
    if (S.volKind != PVOL_VOLUME_GRID) {
        float t0, t1;
        if (!vol_intersect(S, r, &t0, &t1)) return mk4(0.f);
        V3 a = r.o + r.d * t0, b = r.o + r.d * t1;
        return sigT * len(a - b);
    }
    float t0, t1;
    float length = len(r.d);
    if (length == 0.f) return mk4(0.f);
    RayD rn;
    rn.o = r.o; rn.d = vdiv(r.d, length); rn.mint = r.mint * length; rn.maxt = r.maxt * length;
    if (!vol_intersect(S, rn, &t0, &t1)) return mk4(0.f);
    // DensityRegion::tau (core/volume.cpp:296-310): tau += sigma_t(p(t0)) per sample, t0 += stepSize.  The ray is
    // wave-uniform, so the samples go one per lane: the t0 sequence is replayed with the reference's additions (cheap,
    // serial), the trilinear density fetches run in parallel, and sigma_t (constant times density, volume.h:81-92)
    // multiplies the summed densities once -- the only difference is the order of that sum.
    const int lane = (int)threadIdx.x & (LANES - 1);
    float dsum = 0.f;
    t0 += u * stepSize;
    while (t0 < t1) {
        float tMine = 0.f;
        bool mine = false;
        for (int j = 0; j < LANES && t0 < t1; ++j) {
            if (lane == j) { tMine = t0; mine = true; }
            t0 += stepSize;
        }
        float dl = mine ? vol_density(S, rn.o + rn.d * tMine) : 0.f;
        for (int off = 32; off > 0; off >>= 1) dl += __shfl_xor(dl, off);
        dsum += dl;
    }
    return sigT * dsum * stepSize;
}

// RainbowVolume::rainbowReflection (volumes/rainbow.cpp:41-78): the direct term of the rainbow medium is the Mie-hazy phase
// intensity times (a mist fraction of the light + the slice of its spectrum the bow puts at this phase angle).  Float4
// form: lane q holds bins 4q..4q+3.  The two bows are rows of a table -- phase-angle band in degrees, the wavelength ramp
// across it in nm, intensity gain -- looked up without branches; CoefficientSpectrum::filter (core/spectrum.h:300-320)
// becomes two per-bin weights.  Every product keeps the reference's operand order (the outputs are compared with its records).
struct BowBand { float theta0, theta1, lambda0, lambda1, gain; };
__device__ f4 rainbow_reflection(f4 Ld, V3 w, V3 wi, int q) {
    const BowBand bows[2] = {{40.4f, 42.3f, 400.f, 700.f, 0.92f},                       // primary
                             {51.0f, 54.4f, 700.f, 400.f, (float)(0.42 * 0.92f)}};      // secondary: 42 % of the primary
    const float mist = 0.08f;
    const float cosTheta = dot(wi, -w);
    const float theta = 57.2957f * acosf(cosTheta);
    // PhaseMieHazy (core/volume.cpp:138-141), dimmed by 10 % across the bow's inner edge (40.4 .. 40.45 degrees)
    float I = (0.5f + 4.5f * powf(0.5 * (1.f + cosTheta), 8.f)) / (4.f * K_PI);
    const float e0 = 40.4f, e1 = 40.45f, g0 = 1.0f, g1 = 0.9f;
    const float ramp = g0 + (theta - e0) * (g1 - g0) / (e1 - e0);
    I *= theta < e0 ? g0 : (e1 < theta ? g1 : ramp);
    float lambda = 0.f, gain = 1.0f;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const BowBand B = bows[b];
        const float l = B.lambda0 + (theta - B.theta0) * (B.lambda1 - B.lambda0) / (B.theta1 - B.theta0);
        const bool hit = lambda == 0.f && !(theta < B.theta0 || B.theta1 < theta) && l != 0.f;
        lambda = hit ? l : lambda;
        gain = hit ? 1.0f * B.gain : gain;
    }
    if (lambda == 0.f) return Ld * (I * mist);
    // filter(lambda): bin `index` keeps the fraction t of the light, bin index + 1 the fraction 1 - t
    const float pos = (lambda - 400) / (float(700 - 400) / 30);
    const int index = int(pos);
    const float t = pos - index;
    const int first = 4 * q;
    const float in[4] = {Ld.x, Ld.y, Ld.z, Ld.w};
    float bow[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int bin = first + c;
        const float wgt = (bin == index) ? t : ((bin == index + 1 && bin < 30) ? (1 - t) : -1.f);
        bow[c] = (wgt >= 0.f && index >= 0 && index < 30) ? in[c] * wgt : 0.f;
    }
    return (Ld * mist + make_float4(bow[0], bow[1], bow[2], bow[3]) * gain) * I;
}

#endif
