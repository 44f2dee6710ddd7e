// pvol_march.hip -- the hot kernel: PhotonVolumeIntegrator::Li (integrators/photonvolume.cpp:112-222)
// with the radius-bounded k-NN photon gather LPhoton (:65-108), hand-written for gfx950.
//
// Execution model
//   * one 64-lane wavefront == one MT19937 stream == one render tile (renderers/samplerrenderer.cpp:73);
//     the wave walks the tile's camera rays in order, so RNG consumption is exactly the reference's.
//   * spectra are "8 lanes x float4": lane l owns bins 4*(l&7)..+3; the 8 lane-groups are copies, except
//     in the flux sum where each group accumulates a different photon (8 photons per 1-KiB load).
//   * the gather walks rings of (dy,dz) cell rows around the query point; one lane per row computes the
//     row's distance bound and its contiguous photon range, then all 64 lanes stride that range with
//     coalesced float4 position loads.  Accepted candidates are appended to an LDS list by ballot
//     compaction; when the list fills, a wave-wide bisection select keeps the k nearest and shrinks the
//     search radius -- the same radius-shrinking contract as core/kdtree.h:157-183 +
//     core/photonshooter.h:186-203, without the tree.
//   * no MFMA: there is no dense contraction on this path.  Bound: HBM/L2 bytes of the gather.
// Built with -ffp-contract=off so that every geometric decision (step counts, slab clips, cell
// indices, shadow hits) is taken on the same fp32 values the CPU reference computes.
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
__device__ __forceinline__ f4 exp4(f4 a) { return make_float4(expf(a.x), expf(a.y), expf(a.z), expf(a.w)); }
// bins 30,31 are padding: force them back to 0 after an operation that could make them non-finite
__device__ __forceinline__ f4 clean4(f4 a, int q) { if (q == 7) { a.z = 0.f; a.w = 0.f; } return a; }
__device__ __forceinline__ f4 ld4(const float *base32, int q) { return *reinterpret_cast<const f4 *>(base32 + 4 * q); }

__device__ __forceinline__ float group8_sum(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    return v;
}
__device__ __forceinline__ bool wave_any(bool p) { return __ballot(p) != 0ull; }
__device__ __forceinline__ float wave_max(float v) {
    for (int m = 1; m < LANES; m <<= 1) v = fmaxf(v, __shfl_xor(v, m));
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

// ------------------------------------------------------------------------------------------ MT19937 in LDS
struct Rng {
    uint32_t *mt;   // LDS, 624 words
    int mti;        // wave-uniform
    unsigned long long draws;
};

__device__ __forceinline__ uint32_t mt_twist(uint32_t a, uint32_t b) {
    uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
// core/rng.cpp:80-92, 64 words per step.  Within a step every lane reads before any lane writes (LDS
// operations of one wave execute in order), so "old" and "new" words are exactly the serial loop's.
__device__ void mt_regenerate(uint32_t *mt, int lane) {
    for (int base = 0; base < MT_N - MT_M; base += LANES) {
        int kk = base + lane;
        uint32_t v = 0;
        bool on = kk < MT_N - MT_M;
        if (on) v = mt[kk + MT_M] ^ mt_twist(mt[kk], mt[kk + 1]);
        __syncthreads();
        if (on) mt[kk] = v;
        __syncthreads();
    }
    for (int base = MT_N - MT_M; base < MT_N - 1; base += LANES) {
        int kk = base + lane;
        uint32_t v = 0;
        bool on = kk < MT_N - 1;
        if (on) v = mt[kk + (MT_M - MT_N)] ^ mt_twist(mt[kk], mt[kk + 1]);
        __syncthreads();
        if (on) mt[kk] = v;
        __syncthreads();
    }
    if (lane == 0) mt[MT_N - 1] = mt[MT_M - 1] ^ mt_twist(mt[MT_N - 1], mt[0]);
    __syncthreads();
}
// core/rng.cpp:43-55
__device__ void mt_seed(uint32_t *mt, uint32_t seed, int lane) {
    uint32_t x = seed;
    if (lane == 0) mt[0] = x;
    for (int i = 1; i < MT_N; ++i) {
        x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
        if (lane == (i & 63)) mt[i] = x;
    }
    __syncthreads();
}
__device__ __forceinline__ uint32_t rng_uint(Rng &r, int lane) {
    if (r.mti >= MT_N) { mt_regenerate(r.mt, lane); r.mti = 0; }
    uint32_t y = r.mt[r.mti++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    ++r.draws;
    return y;
}
__device__ __forceinline__ float rng_float(Rng &r, int lane) { return (rng_uint(r, lane) & 0xffffff) / float(1 << 24); }
__device__ void rng_skip(Rng &r, unsigned long long n, int lane) {
    r.draws += n;
    while (n > 0) {
        if (r.mti >= MT_N) { mt_regenerate(r.mt, lane); r.mti = 0; }
        unsigned long long avail = (unsigned long long)(MT_N - r.mti);
        unsigned long long take = n < avail ? n : avail;
        r.mti += (int)take;
        n -= take;
    }
}

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
__device__ __forceinline__ bool tri_hit(const DevTri &tr, const RayD &ray) {
    V3 p1 = v3(tr.p1[0], tr.p1[1], tr.p1[2]), p2 = v3(tr.p2[0], tr.p2[1], tr.p2[2]), p3 = v3(tr.p3[0], tr.p3[1], tr.p3[2]);
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
// Scene::IntersectP (core/scene.h:57-61): one triangle per lane
__device__ __forceinline__ bool scene_occluded(const DevScene &S, const RayD &ray, int lane) {
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
    f4 tau = mk4(0.f);
    t0 += u * stepSize;
    while (t0 < t1) {
        tau = tau + sigT * vol_density(S, rn.o + rn.d * t0);
        t0 += stepSize;
    }
    return tau * stepSize;
}

// volumes/rainbow.cpp:41-78 in the float4 layout
__device__ __forceinline__ float lerp_or_zero(float theta, float minT, float maxT, float sw, float ew) {
    if (theta < minT || maxT < theta) return 0;
    const float thetaRange = maxT - minT;
    const float wavelengthRange = ew - sw;
    return sw + (theta - minT) * wavelengthRange / thetaRange;
}
__device__ __forceinline__ float lerp_transfer(float x, float xMin, float xMax, float y0, float y1) {
    if (x < xMin) return y0;
    if (xMax < x) return y1;
    const float thetaRange = xMax - xMin;
    const float range = y1 - y0;
    return y0 + (x - xMin) * range / thetaRange;
}
__device__ f4 rainbow_reflection(f4 spectrum, V3 w, V3 wi, int q) {
    float cosTheta = dot(wi, -w);
    const float radToDeg = 57.2957;
    float theta = radToDeg * acosf(cosTheta);
    float I = (0.5f + 4.5f * powf(0.5 * (1.f + dot(wi, -w)), 8.f)) / (4.f * K_PI);  // PhaseMieHazy core/volume.cpp:138-141
    float innerGlow = lerp_transfer(theta, 40.4, 40.45, 1.0, 0.9);
    I *= innerGlow;
    float rainbowI = 1.0f;
    float primaryRainbowI = 0.92f;
    float secondaryRainbowI = 0.42 * primaryRainbowI;
    float mistI = 0.08f;
    float lambda = lerp_or_zero(theta, 40.4, 42.3, 400.0, 700.0);
    if (lambda) {
        rainbowI *= primaryRainbowI;
    } else {
        lambda = lerp_or_zero(theta, 51.0, 54.4, 700.0, 400.0);
        if (lambda) rainbowI *= secondaryRainbowI;
    }
    if (!lambda) return spectrum * (I * mistI);   // I * mistI * spectrum
    // CoefficientSpectrum::filter, core/spectrum.h:300-320
    float deltaLambda = float(700 - 400) / 30;
    float indexWithDecimals = (lambda - 400) / deltaLambda;
    int index = int(indexWithDecimals);
    float t = indexWithDecimals - index;
    float sp[4] = {spectrum.x, spectrum.y, spectrum.z, spectrum.w};
    float rb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        int bin = 4 * q + c;
        float v = 0.f;
        if (index >= 0 && index < 30) {
            if (bin == index) v = sp[c] * t;
            if (bin == index + 1 && index + 1 < 30) v = sp[c] * (1 - t);
        }
        rb[c] = v;
    }
    f4 rainbow = make_float4(rb[0], rb[1], rb[2], rb[3]);
    return (spectrum * mistI + rainbow * rainbowI) * I;
}

// ------------------------------------------------------------------------------------------ k-NN gather
struct Gather {
    float *cd;      // LDS: candidate dist^2
    uint32_t *ci;   // LDS: candidate photon index (sorted order)
    int cap;
};

// Keep the k nearest of the M > k candidates; returns the k-th smallest dist^2 (the new radius^2).
// Bisection on the fp32 bit pattern (monotone for non-negative floats); ties at the k-th value are
// resolved in list order.
__device__ float select_k(Gather &G, int M, int k, int lane) {
    uint32_t lo = 0u, hi = 0x7f800000u;
    uint32_t tau = hi;
    for (;;) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        int c = 0;
        for (int base = 0; base < M; base += LANES) {
            int i = base + lane;
            bool le = (i < M) && (__float_as_uint(G.cd[i]) <= mid);
            c += __popcll(__ballot(le));
        }
        if (c == k) { tau = mid; break; }
        if (c < k) lo = mid + 1; else hi = mid;
        if (lo >= hi) { tau = lo; break; }
    }
    // compact in place: entries with bits < tau, then as many == tau (list order) as still fit
    int out = 0, nLess = 0;
    for (int base = 0; base < M; base += LANES) {
        int i = base + lane;
        nLess += __popcll(__ballot((i < M) && (__float_as_uint(G.cd[i]) < tau)));
    }
    int quota = k - nLess;  // > 0 only in the tie case; in the c == k case every entry <= tau is kept
    float kept_max = 0.f;
    for (int base = 0; base < M; base += LANES) {
        int i = base + lane;
        float d = 0.f; uint32_t id = 0u;
        bool on = i < M;
        if (on) { d = G.cd[i]; id = G.ci[i]; }
        uint32_t bits = __float_as_uint(d);
        bool less = on && bits < tau;
        bool eq = on && bits == tau;
        uint64_t me = __ballot(eq);
        bool keep = less || (eq && (int)lanes_below(me, lane) < quota);
        quota -= __popcll(me);
        uint64_t mk = __ballot(keep);
        __syncthreads();
        if (keep) {
            int pos = out + (int)lanes_below(mk, lane);
            G.cd[pos] = d;
            G.ci[pos] = id;
            kept_max = fmaxf(kept_max, d);
        }
        __syncthreads();
        out += __popcll(mk);
    }
    return wave_max(kept_max);
}

// PhotonVolumeIntegrator::LPhoton (photonvolume.cpp:65-108): k nearest photons within maxDist of pt;
// returns total flux / (4/3 pi r^3 sigma_s) in the float4 layout, or 0 when fewer than 10 are found.
template <bool STATS>
__device__ f4 lphoton(const DevScene &S, Gather &G, V3 w, V3 pt, f4 sigS_at_p, int lane, DevCounters *ctr) {
    const int q = lane & 7;
    f4 zero = mk4(0.f);
    if (S.nPhotons == 0u) return zero;
    float T = S.maxDistSq;
    const float cell = S.cellSize, inv = S.invCell;
    const float eps = cell * 1e-4f;  // slack on the PRUNING bounds only; acceptance stays the exact d2 < T
    int count = 0;
    const int k = S.nUsed;
    unsigned long long tested = 0;
    float fy = (pt.y - S.gridLo[1]) * inv, fz = (pt.z - S.gridLo[2]) * inv;
    int cy = (int)floorf(fy), cz = (int)floorf(fz);
    const int R = S.ringMax;
    for (int r = 0; r <= R; ++r) {
        if (r >= 2) {
            float dmin = (r - 1) * cell - eps;
            if (dmin * dmin >= T) break;
        }
        int nrows = r == 0 ? 1 : 8 * r;
        int dy = 0, dz = 0;
        if (r > 0) {
            int side = lane / (2 * r), t = lane - side * (2 * r);
            if (side == 0) { dy = -r + t; dz = -r; }
            else if (side == 1) { dy = r; dz = -r + t; }
            else if (side == 2) { dy = r - t; dz = r; }
            else { dy = -r; dz = r - t; }
        }
        int y = cy + dy, z = cz + dz;
        bool rowOn = lane < nrows && y >= 0 && y < S.gdim[1] && z >= 0 && z < S.gdim[2];
        float ylo = S.gridLo[1] + y * cell, zlo = S.gridLo[2] + z * cell;
        float ddy = fmaxf(0.f, fmaxf(ylo - pt.y, pt.y - (ylo + cell)) - eps);
        float ddz = fmaxf(0.f, fmaxf(zlo - pt.z, pt.z - (zlo + cell)) - eps);
        float rd2 = ddy * ddy + ddz * ddz;
        rowOn = rowOn && rd2 < T;
        float hw = sqrtf(fmaxf(0.f, T - rd2)) + eps;
        int x0 = (int)floorf((pt.x - hw - S.gridLo[0]) * inv), x1 = (int)floorf((pt.x + hw - S.gridLo[0]) * inv);
        x0 = max(x0, 0);
        x1 = min(x1, S.gdim[0] - 1);
        rowOn = rowOn && x0 <= x1;
        uint32_t start = 0u, end = 0u;
        if (rowOn) {
            size_t base = ((size_t)z * S.gdim[1] + y) * S.gdim[0];
            start = S.cellStart[base + x0];
            end = S.cellStart[base + x1 + 1];
            rowOn = end > start;
        }
        uint64_t rows = __ballot(rowOn);
        while (rows) {
            int j = __ffsll((unsigned long long)rows) - 1;
            rows &= rows - 1;
            float rd2j = __shfl(rd2, j);
            if (rd2j >= T) continue;
            uint32_t s = __shfl(start, j), e = __shfl(end, j);
            if (STATS) tested += e - s;
            for (uint32_t b = s; b < e; b += LANES) {
                uint32_t i = b + lane;
                bool on = i < e;
                f4 P = on ? S.pos4[i] : mk4(0.f);
                float dx = P.x - pt.x, dyy = P.y - pt.y, dzz = P.z - pt.z;
                float d2 = dx * dx + dyy * dyy + dzz * dzz;   // DistanceSquared(photon.p, p), kdtree.h:180
                bool acc = on && d2 < T;
                uint64_t m = __ballot(acc);
                if (m) {
                    if (acc) {
                        int pos = count + (int)lanes_below(m, lane);
                        G.cd[pos] = d2;
                        G.ci[pos] = i;
                    }
                    count += __popcll(m);
                    __syncthreads();
                    if (count > G.cap - LANES) {
                        T = select_k(G, count, k, lane);
                        count = k;
                    }
                }
            }
        }
    }
    __syncthreads();
    int nFound = count;
    float maxmd;
    if (count > k) {
        maxmd = select_k(G, count, k, lane);
        nFound = k;
    } else {
        float m = 0.f;
        for (int base = 0; base < nFound; base += LANES) {
            int i = base + lane;
            if (i < nFound) m = fmaxf(m, G.cd[i]);
        }
        maxmd = wave_max(m);
    }
    if (STATS && lane == 0) {
        atomicAdd(&ctr->nTested, tested);
        if (nFound < 10) atomicAdd(&ctr->nLookupsLt10, 1ull); else atomicAdd(&ctr->nKept, (unsigned long long)nFound);
    }
    if (nFound < 10) return zero;   // photonvolume.cpp:83-84
    // totalFlux += alpha * p(pt_i, wi_i, -w): 8 photons per pass, one 128-B row per 8-lane group
    const int grp = lane >> 3;
    f4 acc = zero;
    const bool iso = (S.g == 0.f);
    const float wIso = 1.f / (4.f * K_PI) * (1.f - 0.f * 0.f) / powf(1.f + 0.f * 0.f - 2.f * 0.f * 0.f, 1.5f);
    V3 mw = -w;
    for (int base = 0; base < nFound; base += 8) {
        int eidx = base + grp;
        if (eidx < nFound) {
            uint32_t id = G.ci[eidx];
            f4 a = S.alpha4[(size_t)id * 8 + q];
            float wgt = wIso;
            if (!iso) {
                f4 wi = S.wi4[id];
                wgt = phase_hg(v3(wi.x, wi.y, wi.z), mw, S.g);
            }
            acc = acc + a * wgt;
        }
    }
    acc.x += __shfl_xor(acc.x, 8); acc.y += __shfl_xor(acc.y, 8); acc.z += __shfl_xor(acc.z, 8); acc.w += __shfl_xor(acc.w, 8);
    acc.x += __shfl_xor(acc.x, 16); acc.y += __shfl_xor(acc.y, 16); acc.z += __shfl_xor(acc.z, 16); acc.w += __shfl_xor(acc.w, 16);
    acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32); acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
    float distSq = maxmd;
    float dV = distSq * sqrtf(distSq);
    f4 L = zero;
    if (dV != 0.f && !spec_is_black(sigS_at_p)) {
        float f = float(4.0 / 3.0 * (double)K_PI * (double)dV);   // photonvolume.cpp:103: double product -> float
        L = clean4(acc / (sigS_at_p * f), q);
    }
    return L;
}

// ------------------------------------------------------------------------------------------ Li
struct LiArgs {
    const DevScene *scene;
    const pvol_ray *rays;
    pvol_stream *streams;
    uint32_t nStreams;
    int outputKind;
    float *out;
    uint32_t *draws;
    const uint32_t *initState;  // optional: nStreams x 625 words (mt[624], mti) instead of seeding
    uint32_t *finalState;       // optional: same layout, written back
    DevCounters *counters;
    int transmittanceOnly;
};

// PhotonVolumeIntegrator::Transmittance with sample == NULL (photonvolume.cpp:15-30)
__device__ __forceinline__ f4 transmittance(const DevScene &S, const RayD &ray, Rng &rng, f4 sigT, int lane) {
    if (S.volKind == PVOL_VOLUME_NONE) return mk4(1.f);
    float step = 4.f * S.stepSize;
    float offset = rng_float(rng, lane);
    f4 tau = vol_tau(S, ray, step, offset, sigT);
    return exp4(neg4(tau));
}

template <bool STATS>
__global__ __launch_bounds__(LANES) void li_kernel(LiArgs A) {
    extern __shared__ __align__(16) unsigned char lds[];
    const DevScene &S = *A.scene;
    const int lane = threadIdx.x;
    const int q = lane & 7;
    const uint32_t sidx = blockIdx.x;
    if (sidx >= A.nStreams) return;
    // LDS plan: MT state | candidate d2 | candidate idx | lightNum
    uint32_t *mt = reinterpret_cast<uint32_t *>(lds);
    Gather G;
    G.cap = S.candCap;
    G.cd = reinterpret_cast<float *>(lds + MT_N * 4);
    G.ci = reinterpret_cast<uint32_t *>(lds + MT_N * 4 + (size_t)G.cap * 4);
    float *lightNum = reinterpret_cast<float *>(lds + MT_N * 4 + (size_t)G.cap * 8);

    pvol_stream st = A.streams[sidx];
    Rng rng;
    rng.mt = mt;
    rng.draws = 0;
    if (A.initState) {
        const uint32_t *src = A.initState + (size_t)sidx * (MT_N + 1);
        for (int i = lane; i < MT_N; i += LANES) mt[i] = src[i];
        rng.mti = (int)src[MT_N];
        rng.draws = st.start_draw;
        __syncthreads();
    } else {
        mt_seed(mt, st.seed, lane);
        rng.mti = MT_N;
        rng_skip(rng, st.start_draw, lane);
    }

    const f4 sigA = ld4(S.sigA, q), sigS = ld4(S.sigS, q), le = ld4(S.le, q);
    const f4 sigT = sigA + sigS;
    const f4 Y = ld4(S.cieY, q);
    const int nLights = S.nLights;
    const bool rainbow = (S.volKind == PVOL_VOLUME_RAINBOW);
    unsigned long long cRays = 0, cSteps = 0, cUnocc = 0;

    for (uint32_t k = 0; k < st.n_rays; ++k) {
        const size_t ri = (size_t)st.first_ray + k;
        const pvol_ray pr = A.rays[ri];
        rng_skip(rng, pr.rng_skip, lane);
        const unsigned long long d0 = rng.draws;
        RayD ray;
        ray.o = v3(pr.o[0], pr.o[1], pr.o[2]);
        ray.d = v3(pr.d[0], pr.d[1], pr.d[2]);
        ray.mint = pr.mint;
        ray.maxt = pr.maxt;
        f4 Lv = mk4(0.f), Tr = mk4(1.f);
        if (STATS) ++cRays;

        if (A.transmittanceOnly) {
            Tr = transmittance(S, ray, rng, sigT, lane);
        } else {
            float t0, t1;
            bool hit = S.volKind != PVOL_VOLUME_NONE && vol_intersect(S, ray, &t0, &t1) && (t1 - t0) != 0.f;
            int nSamples = hit ? (int)ceilf((t1 - t0) / S.stepSize) : 0;
            if (hit && nLights > 1 && nSamples > S.maxSteps) {
                // the LDS plan cannot hold this ray's lightNum[] array: report, never guess
                if (lane == 0) atomicAdd(&A.counters->nErrors, 1ull);
                hit = false;
            }
            if (hit) {
                float step = (t1 - t0) / nSamples;
                V3 p = ray.o + ray.d * t0, pPrev;
                V3 w = -ray.d;
                t0 += pr.scatter_u * step;
                // LDShuffleScrambled1D(1,n,lightNum) + (1,n,lightComp) + 2D(1,n,lightPos): 4+6n draws
                // (photonvolume.cpp:137-142).  The three delta lights ignore lightComp/lightPos, and
                // lightNum only matters with more than one light: skip what is never read.
                if (nLights > 1) {
                    uint32_t scramble = rng_uint(rng, lane);
                    for (int i = lane; i < nSamples; i += LANES) lightNum[i] = van_der_corput((uint32_t)i, scramble);
                    rng_skip(rng, (unsigned long long)nSamples, lane);  // n one-element shuffles (montecarlo.h:308-309)
                    __syncthreads();
                    for (int i = 0; i < nSamples; ++i) {                // Shuffle(samples, n, 1), montecarlo.h:174-181
                        uint32_t other = (uint32_t)i + (rng_uint(rng, lane) % (uint32_t)(nSamples - i));
                        if (lane == 0) {
                            float a = lightNum[i], b = lightNum[other];
                            lightNum[i] = b;
                            lightNum[other] = a;
                        }
                    }
                    __syncthreads();
                    rng_skip(rng, 3ull + 4ull * (unsigned long long)nSamples, lane);
                } else {
                    rng_skip(rng, 4ull + 6ull * (unsigned long long)nSamples, lane);
                }
                for (int i = 0; i < nSamples; ++i, t0 += step) {
                    if (STATS) ++cSteps;
                    pPrev = p;
                    p = ray.o + ray.d * t0;
                    RayD tauRay;
                    tauRay.o = pPrev; tauRay.d = p - pPrev; tauRay.mint = 0.f; tauRay.maxt = 1.f;
                    f4 stepTau = vol_tau(S, tauRay, .5f * S.stepSize, rng_float(rng, lane), sigT);
                    Tr = exp4(neg4(stepTau));   // assigned, not accumulated (photonvolume.cpp:155)
                    if (spec_y(Tr, Y) < 1e-3) {
                        const float continueProb = .5f;
                        if (rng_float(rng, lane) > continueProb) { Tr = mk4(0.f); break; }
                        Tr = Tr / continueProb;
                    }
                    float dens = vol_density(S, p);
                    f4 ss = sigS * dens, sa = sigA * dens;
                    f4 L_d = mk4(0.f), L_ii = mk4(0.f), L_i;
                    if (!spec_is_black(ss) && nLights > 0) {
                        int ln = 0;
                        if (nLights > 1) ln = min((int)floorf(lightNum[i] * nLights), nLights - 1);
                        const DevLight &light = S.lights[ln];
                        // Light::Sample_L(p, 0, ...): distant.cpp:48-55, point.cpp:50-57, spot.cpp:50-57
                        V3 wo;
                        RayD vis;
                        f4 L = ld4(light.intensity, q);
                        if (light.kind == PVOL_LIGHT_DISTANT) {
                            wo = v3(light.dir[0], light.dir[1], light.dir[2]);
                            vis.o = p; vis.d = wo; vis.mint = 0.f; vis.maxt = INFINITY;
                        } else {
                            V3 lp = v3(light.pos[0], light.pos[1], light.pos[2]);
                            wo = normalize(lp - p);
                            float dist = len(p - lp);
                            vis.o = p; vis.d = vdiv(lp - p, dist); vis.mint = 0.f; vis.maxt = dist * (1.f - 0.f);
                            float d2 = len_sq(lp - p);
                            if (light.kind == PVOL_LIGHT_SPOT) {
                                // SpotLight::Falloff(-wi), spot.cpp:60-69
                                V3 wl = normalize(v3(light.w2l[0] * -wo.x + light.w2l[1] * -wo.y + light.w2l[2] * -wo.z,
                                                     light.w2l[4] * -wo.x + light.w2l[5] * -wo.y + light.w2l[6] * -wo.z,
                                                     light.w2l[8] * -wo.x + light.w2l[9] * -wo.y + light.w2l[10] * -wo.z));
                                float costheta = wl.z, fall;
                                if (costheta < light.cosTotalWidth) fall = 0.f;
                                else if (costheta > light.cosFalloffStart) fall = 1.f;
                                else {
                                    float delta = (costheta - light.cosTotalWidth) / (light.cosFalloffStart - light.cosTotalWidth);
                                    fall = delta * delta * delta * delta;
                                }
                                L = L * fall / d2;
                            } else {
                                L = L / d2;
                            }
                        }
                        const float pdf = 1.f;
                        if (!spec_is_black(L) && pdf > 0.f && !scene_occluded(S, vis, lane)) {
                            if (STATS) ++cUnocc;
                            f4 Ld = L * transmittance(S, vis, rng, sigT, lane);
                            if (rainbow) L_d = rainbow_reflection(Ld, ray.d, wo, q);
                            else L_d = Ld * vol_phase(S, p, w, -wo) * float(nLights) / pdf;
                        }
                    }
                    if (!rainbow) L_ii = lphoton<STATS>(S, G, w, p, ss, lane, A.counters);
                    if (spec_y(sa, Y) != 0.0 || spec_y(ss, Y) != 0.0) L_i = L_d + clean4(ss / (sa + ss), q) * L_ii;
                    else L_i = L_d;
                    Lv = (sa * (le * dens) * step) + (ss * L_i * step) + (Tr * Lv);
                }
            }
        }
        // outputs
        if (A.outputKind == PVOL_OUT_SPECTRAL) {
            if (lane < 8) {
                float *o = A.out + ri * 60;
                const float lv[4] = {Lv.x, Lv.y, Lv.z, Lv.w}, tr[4] = {Tr.x, Tr.y, Tr.z, Tr.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    int bin = 4 * q + c;
                    if (bin < 30) { o[bin] = lv[c]; o[30 + bin] = tr[c]; }
                }
            }
        } else {
            f4 X4 = ld4(S.cieX, q), Z4 = ld4(S.cieZ, q);
            float scale = float(700 - 400) / float(106.856895f * 30);   // core/spectrum.h:427-428
            float x = group8_sum(X4.x * Lv.x + X4.y * Lv.y + X4.z * Lv.z + X4.w * Lv.w) * scale;
            float y = group8_sum(Y.x * Lv.x + Y.y * Lv.y + Y.z * Lv.z + Y.w * Lv.w) * scale;
            float z = group8_sum(Z4.x * Lv.x + Z4.y * Lv.y + Z4.z * Lv.z + Z4.w * Lv.w) * scale;
            float ty = spec_y(Tr, Y);
            if (lane == 0) *reinterpret_cast<f4 *>(A.out + ri * 4) = make_float4(x, y, z, ty);
        }
        if (A.draws && lane == 0) A.draws[ri] = (uint32_t)(rng.draws - d0);
    }
    if (lane == 0) A.streams[sidx].end_draw = rng.draws;
    if (A.finalState) {
        uint32_t *dst = A.finalState + (size_t)sidx * (MT_N + 1);
        __syncthreads();
        for (int i = lane; i < MT_N; i += LANES) dst[i] = mt[i];
        if (lane == 0) dst[MT_N] = (uint32_t)rng.mti;
    }
    if (STATS && lane == 0) {
        atomicAdd(&A.counters->nRays, cRays);
        atomicAdd(&A.counters->nSteps, cSteps);
        atomicAdd(&A.counters->nShadowUnoccluded, cUnocc);
    }
}

extern "C" hipError_t pvol_launch_li(const LiArgs *args, size_t ldsBytes, bool stats, hipStream_t stream) {
    dim3 grid(args->nStreams), block(LANES);
    if (stats) hipLaunchKernelGGL(li_kernel<true>, grid, block, ldsBytes, stream, *args);
    else hipLaunchKernelGGL(li_kernel<false>, grid, block, ldsBytes, stream, *args);
    return hipGetLastError();
}
