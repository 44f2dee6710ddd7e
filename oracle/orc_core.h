// oracle/orc_core.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the value layer under the volumetric photon-mapping hot path of
// piwell/CS348B-pbrt.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may use anything under oracle/.  Each function cites the reference file:line it
// follows (paths relative to the reference tree).  Arithmetic is kept in the reference's
// evaluation order and precision (fp32 unless noted); build with -ffp-contract=off.
//
// Pinning: see oracle/README.md -- the restatement is checked against outputs of the
// reference's own objects (oracle/_ref, built by oracle/Makefile) through the fixtures in
// tests/golden/.
#ifndef ORC_CORE_H
#define ORC_CORE_H

#include <math.h>
#include <stdint.h>
#include <string.h>
#include <float.h>
#include <algorithm>
#include <vector>

#include "../include/pvol.h"

namespace orc {

// core/pbrt.h:188-197 -- M_PI is redefined as a float literal; INFINITY is the libm one.
static const float kPi = 3.14159265358979323846f;
static const float kInvPi = 0.31830988618379067154f;
static const float kInfinity = INFINITY;
// core/montecarlo.h:50
static const float kOneMinusEpsilon = 0x1.fffffep-1;

static const int NB = PVOL_NBINS;

// ---------------------------------------------------------------- spectrum (core/spectrum.h)
struct Spec {
    float c[NB];
};
inline Spec spec_const(float v) { Spec s; for (int i = 0; i < NB; ++i) s.c[i] = v; return s; }
inline Spec spec_from(const pvol_spectrum &p) { Spec s; memcpy(s.c, p.c, sizeof(s.c)); return s; }
inline Spec operator+(const Spec &a, const Spec &b) { Spec r; for (int i = 0; i < NB; ++i) r.c[i] = a.c[i] + b.c[i]; return r; }
inline Spec operator-(const Spec &a, const Spec &b) { Spec r; for (int i = 0; i < NB; ++i) r.c[i] = a.c[i] - b.c[i]; return r; }
inline Spec operator*(const Spec &a, const Spec &b) { Spec r; for (int i = 0; i < NB; ++i) r.c[i] = a.c[i] * b.c[i]; return r; }
inline Spec operator/(const Spec &a, const Spec &b) { Spec r; for (int i = 0; i < NB; ++i) r.c[i] = a.c[i] / b.c[i]; return r; }
inline Spec operator*(const Spec &a, float f) { Spec r; for (int i = 0; i < NB; ++i) r.c[i] = a.c[i] * f; return r; }
inline Spec operator*(float f, const Spec &a) { return a * f; }
// core/spectrum.h:183-190 divides each bin (no reciprocal)
inline Spec operator/(const Spec &a, float f) { Spec r; for (int i = 0; i < NB; ++i) r.c[i] = a.c[i] / f; return r; }
inline Spec operator-(const Spec &a) { Spec r; for (int i = 0; i < NB; ++i) r.c[i] = -a.c[i]; return r; }
inline Spec &operator+=(Spec &a, const Spec &b) { for (int i = 0; i < NB; ++i) a.c[i] += b.c[i]; return a; }
inline Spec &operator*=(Spec &a, const Spec &b) { for (int i = 0; i < NB; ++i) a.c[i] *= b.c[i]; return a; }
inline Spec &operator*=(Spec &a, float f) { for (int i = 0; i < NB; ++i) a.c[i] *= f; return a; }
inline Spec &operator/=(Spec &a, float f) { for (int i = 0; i < NB; ++i) a.c[i] /= f; return a; }
inline bool is_black(const Spec &a) { for (int i = 0; i < NB; ++i) if (a.c[i] != 0.f) return false; return true; }
inline bool spec_eq(const Spec &a, const Spec &b) { for (int i = 0; i < NB; ++i) if (a.c[i] != b.c[i]) return false; return true; }
// core/spectrum.h:227-233
inline Spec spec_exp(const Spec &a) { Spec r; for (int i = 0; i < NB; ++i) r.c[i] = expf(a.c[i]); return r; }
// core/spectrum.h:234-240
inline Spec spec_clamp0(const Spec &a) { Spec r; for (int i = 0; i < NB; ++i) r.c[i] = a.c[i] < 0.f ? 0.f : a.c[i]; return r; }

// core/spectrum.h:266-279 -- integer wavelength of the single positive bin, else -1.
inline int extract_lambda(const Spec &a) {
    bool first = true;
    int l = -1;
    const int step = (700 - 400) / (NB - 1);  // integer division: 10
    for (int i = 0; i < NB; ++i) {
        if (a.c[i] > 0.f && !first) return -1;
        if (a.c[i] > 0.f && first) { l = 400 + i * step; first = false; }
    }
    return l;
}

// The colour-matching weights travel with the scene (pvol_scene.cie_*): SampledSpectrum::X/Y/Z.
struct Cie {
    Spec x, y, z;
    float scale;  // core/spectrum.h:427-428, 436-437
};
// core/spectrum.h:433-439
inline float spec_y(const Cie &cie, const Spec &a) {
    float yy = 0.f;
    for (int i = 0; i < NB; ++i) yy += cie.y.c[i] * a.c[i];
    return yy * float(700 - 400) / float(106.856895f * NB);
}
// core/spectrum.h:420-432
inline void spec_xyz(const Cie &cie, const Spec &a, float xyz[3]) {
    xyz[0] = xyz[1] = xyz[2] = 0.f;
    for (int i = 0; i < NB; ++i) {
        xyz[0] += cie.x.c[i] * a.c[i];
        xyz[1] += cie.y.c[i] * a.c[i];
        xyz[2] += cie.z.c[i] * a.c[i];
    }
    float scale = float(700 - 400) / float(106.856895f * NB);
    xyz[0] *= scale; xyz[1] *= scale; xyz[2] *= scale;
}

// ---------------------------------------------------------------- geometry (core/geometry.h)
struct V3 { float x, y, z; };
inline V3 v3(float x, float y, float z) { V3 v = {x, y, z}; return v; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, float f) { return v3(a.x * f, a.y * f, a.z * f); }
inline V3 operator*(float f, V3 a) { return a * f; }
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
// geometry.h:94-98: multiply by the reciprocal
inline V3 operator/(V3 a, float f) { float inv = 1.f / f; return v3(a.x * inv, a.y * inv, a.z * inv); }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float length_sq(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
inline float length(V3 a) { return sqrtf(length_sq(a)); }
inline V3 normalize(V3 a) { return a / length(a); }
inline float comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
// geometry.h:477-484: products and differences in double, rounded once to float
inline V3 cross(V3 a, V3 b) {
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return v3(float((ay * bz) - (az * by)), float((az * bx) - (ax * bz)), float((ax * by) - (ay * bx)));
}
// geometry.h:508-519
inline void coordinate_system(V3 v1, V3 *v2, V3 *v3o) {
    if (fabsf(v1.x) > fabsf(v1.y)) {
        float invLen = 1.f / sqrtf(v1.x * v1.x + v1.z * v1.z);
        *v2 = v3(-v1.z * invLen, 0.f, v1.x * invLen);
    } else {
        float invLen = 1.f / sqrtf(v1.y * v1.y + v1.z * v1.z);
        *v2 = v3(0.f, v1.z * invLen, -v1.y * invLen);
    }
    *v3o = cross(v1, *v2);
}

struct Ray {
    V3 o, d;
    float mint, maxt, time;
};
inline Ray make_ray(V3 o, V3 d, float mint, float maxt, float time) { Ray r; r.o = o; r.d = d; r.mint = mint; r.maxt = maxt; r.time = time; return r; }
inline V3 ray_at(const Ray &r, float t) { return r.o + r.d * t; }

// core/transform.h:187-217 (point, with the w divide) and :220-226 (vector)
inline V3 xform_point(const float *m, V3 p) {
    float x = p.x, y = p.y, z = p.z;
    float xp = m[0] * x + m[1] * y + m[2] * z + m[3];
    float yp = m[4] * x + m[5] * y + m[6] * z + m[7];
    float zp = m[8] * x + m[9] * y + m[10] * z + m[11];
    float wp = m[12] * x + m[13] * y + m[14] * z + m[15];
    if (wp == 1.f) return v3(xp, yp, zp);
    float inv = 1.f / wp;  // Point::operator/ (geometry.h:193-196)
    return v3(inv * xp, inv * yp, inv * zp);
}
inline V3 xform_vector(const float *m, V3 v) {
    float x = v.x, y = v.y, z = v.z;
    return v3(m[0] * x + m[1] * y + m[2] * z, m[4] * x + m[5] * y + m[6] * z, m[8] * x + m[9] * y + m[10] * z);
}

struct Box { V3 lo, hi; };
// core/geometry.h:404-408
inline bool box_inside(const Box &b, V3 p) {
    return p.x >= b.lo.x && p.x <= b.hi.x && p.y >= b.lo.y && p.y <= b.hi.y && p.z >= b.lo.z && p.z <= b.hi.z;
}
// core/geometry.cpp:68-86
inline bool box_intersect(const Box &b, const Ray &ray, float *hitt0, float *hitt1) {
    float t0 = ray.mint, t1 = ray.maxt;
    for (int i = 0; i < 3; ++i) {
        float invRayDir = 1.f / comp(ray.d, i);
        float tNear = (comp(b.lo, i) - comp(ray.o, i)) * invRayDir;
        float tFar = (comp(b.hi, i) - comp(ray.o, i)) * invRayDir;
        if (tNear > tFar) std::swap(tNear, tFar);
        t0 = tNear > t0 ? tNear : t0;
        t1 = tFar < t1 ? tFar : t1;
        if (t0 > t1) return false;
    }
    *hitt0 = t0;
    *hitt1 = t1;
    return true;
}
// core/geometry.cpp:60-63
inline void box_bounding_sphere(const Box &b, V3 *c, float *rad) {
    *c = b.lo * .5f + b.hi * .5f;
    *rad = box_inside(b, *c) ? length(*c - b.hi) : 0.f;
}

// ---------------------------------------------------------------- RNG (core/rng.cpp:43-107)
struct Rng {
    uint32_t mt[PVOL_MT_N];
    int mti;
    uint64_t draws;  // RandomUInt calls since construction (bookkeeping only)
    explicit Rng(uint32_t seed = 5489u) { this->seed(seed); }
    void seed(uint32_t s) {
        mt[0] = s;
        for (mti = 1; mti < PVOL_MT_N; mti++) mt[mti] = 1812433253u * (mt[mti - 1] ^ (mt[mti - 1] >> 30)) + (uint32_t)mti;
        draws = 0;
    }
    void regenerate() {
        const int N = PVOL_MT_N, M = 397;
        const uint32_t A = 0x9908b0dfu, UP = 0x80000000u, LO = 0x7fffffffu;
        int kk;
        uint32_t y;
        for (kk = 0; kk < N - M; kk++) { y = (mt[kk] & UP) | (mt[kk + 1] & LO); mt[kk] = mt[kk + M] ^ (y >> 1) ^ ((y & 1u) ? A : 0u); }
        for (; kk < N - 1; kk++) { y = (mt[kk] & UP) | (mt[kk + 1] & LO); mt[kk] = mt[kk + (M - N)] ^ (y >> 1) ^ ((y & 1u) ? A : 0u); }
        y = (mt[N - 1] & UP) | (mt[0] & LO);
        mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        mti = 0;
    }
    uint32_t random_uint() {
        if (mti >= PVOL_MT_N) regenerate();
        uint32_t y = mt[mti++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        ++draws;
        return y;
    }
    // core/rng.cpp:59-65
    float random_float() { return (random_uint() & 0xffffff) / float(1 << 24); }
    void skip(uint64_t n) { for (uint64_t i = 0; i < n; ++i) random_uint(); }
};

// ---------------------------------------------------------------- LD sampling (core/montecarlo.h)
// montecarlo.h:277-286
inline float van_der_corput(uint32_t n, uint32_t scramble) {
    n = (n << 16) | (n >> 16);
    n = ((n & 0x00ff00ff) << 8) | ((n & 0xff00ff00) >> 8);
    n = ((n & 0x0f0f0f0f) << 4) | ((n & 0xf0f0f0f0) >> 4);
    n = ((n & 0x33333333) << 2) | ((n & 0xcccccccc) >> 2);
    n = ((n & 0x55555555) << 1) | ((n & 0xaaaaaaaa) >> 1);
    n ^= scramble;
    return std::min(((n >> 8) & 0xffffff) / float(1 << 24), kOneMinusEpsilon);
}
// montecarlo.h:289-293
inline float sobol2(uint32_t n, uint32_t scramble) {
    for (uint32_t v = 1u << 31; n != 0; n >>= 1, v ^= v >> 1)
        if (n & 0x1) scramble ^= v;
    return std::min(((scramble >> 8) & 0xffffff) / float(1 << 24), kOneMinusEpsilon);
}
// montecarlo.h:174-181
template <typename T> inline void shuffle(T *samp, uint32_t count, uint32_t dims, Rng &rng) {
    for (uint32_t i = 0; i < count; ++i) {
        uint32_t other = i + (rng.random_uint() % (count - i));
        for (uint32_t j = 0; j < dims; ++j) std::swap(samp[dims * i + j], samp[dims * other + j]);
    }
}
// montecarlo.h:304-312
inline void ld_shuffle_scrambled_1d(int nSamples, int nPixel, float *samples, Rng &rng) {
    uint32_t scramble = rng.random_uint();
    for (int i = 0; i < nSamples * nPixel; ++i) samples[i] = van_der_corput(i, scramble);
    for (int i = 0; i < nPixel; ++i) shuffle(samples + i * nSamples, nSamples, 1, rng);
    shuffle(samples, nPixel, nSamples, rng);
}
// montecarlo.h:315-323
inline void ld_shuffle_scrambled_2d(int nSamples, int nPixel, float *samples, Rng &rng) {
    uint32_t scramble[2];
    scramble[0] = rng.random_uint();
    scramble[1] = rng.random_uint();
    for (int i = 0; i < nSamples * nPixel; ++i) {
        samples[2 * i] = van_der_corput(i, scramble[0]);
        samples[2 * i + 1] = sobol2(i, scramble[1]);
    }
    for (int i = 0; i < nPixel; ++i) shuffle(samples + 2 * i * nSamples, nSamples, 2, rng);
    shuffle(samples, nPixel, 2 * nSamples, rng);
}

// montecarlo.h:206-218 -- digit loop in double, `n *= invBase` truncates through double.
inline double permuted_radical_inverse(uint32_t n, uint32_t base, const uint32_t *p) {
    double val = 0;
    double invBase = 1. / base, invBi = invBase;
    while (n > 0) {
        uint32_t d_i = p[n % base];
        val += d_i * invBi;
        n = (uint32_t)(n * invBase);
        invBi *= invBase;
    }
    return val;
}
// montecarlo.h:221-243, montecarlo.cpp:380-397 (bases = the first `dims` primes)
struct PermutedHalton {
    uint32_t dims;
    uint32_t b[8];
    uint32_t permute[64];
    PermutedHalton(uint32_t d, Rng &rng) {
        static const uint32_t primes[8] = {2, 3, 5, 7, 11, 13, 17, 19};
        dims = d;
        uint32_t *p = permute;
        for (uint32_t i = 0; i < dims; ++i) {
            b[i] = primes[i];
            for (uint32_t j = 0; j < b[i]; ++j) p[j] = j;  // GeneratePermutation, montecarlo.h:199-203
            shuffle(p, b[i], 1, rng);
            p += b[i];
        }
    }
    void sample(uint32_t n, float *out) const {
        const uint32_t *p = permute;
        for (uint32_t i = 0; i < dims; ++i) {
            out[i] = std::min(float(permuted_radical_inverse(n, b[i], p)), kOneMinusEpsilon);
            p += b[i];
        }
    }
};

// montecarlo.cpp:283-290
inline V3 uniform_sample_sphere(float u1, float u2) {
    float z = 1.f - 2.f * u1;
    float r = sqrtf(std::max(0.f, 1.f - z * z));
    float phi = 2.f * kPi * u2;
    return v3(r * cosf(phi), r * sinf(phi), z);
}
// montecarlo.cpp:405-410
inline V3 uniform_sample_cone(float u1, float u2, float costhetamax) {
    float costheta = (1.f - u1) + u1 * costhetamax;
    float sintheta = sqrtf(1.f - costheta * costheta);
    float phi = u2 * 2.f * kPi;
    return v3(cosf(phi) * sintheta, sinf(phi) * sintheta, costheta);
}
// montecarlo.cpp:306-348
inline void concentric_sample_disk(float u1, float u2, float *dx, float *dy) {
    float r, theta;
    float sx = 2 * u1 - 1;
    float sy = 2 * u2 - 1;
    if (sx == 0.0 && sy == 0.0) { *dx = 0.0; *dy = 0.0; return; }
    if (sx >= -sy) {
        if (sx > sy) { r = sx; if (sy > 0.0) theta = sy / r; else theta = 8.0f + sy / r; }
        else { r = sy; theta = 2.0f - sx / r; }
    } else {
        if (sx <= sy) { r = -sx; theta = 4.0f - sy / r; }
        else { r = -sy; theta = 6.0f + sx / r; }
    }
    theta *= kPi / 4.f;
    *dx = r * cosf(theta);
    *dy = r * sinf(theta);
}
// montecarlo.cpp:351-357 (CosineSampleHemisphere, used by BxDF::Sample_f reflection.cpp:436-444)
inline V3 cosine_sample_hemisphere(float u1, float u2) {
    V3 ret;
    concentric_sample_disk(u1, u2, &ret.x, &ret.y);
    ret.z = sqrtf(std::max(0.f, 1.f - ret.x * ret.x - ret.y * ret.y));
    return ret;
}

// core/volume.cpp:150-154
inline float phase_hg(V3 w, V3 wp, float g) {
    float costheta = dot(w, wp);
    return 1.f / (4.f * kPi) * (1.f - g * g) / powf(1.f + g * g - 2.f * g * costheta, 1.5f);
}
// core/volume.cpp:138-141
inline float phase_mie_hazy(V3 w, V3 wp) {
    float costheta = dot(w, wp);
    return (0.5f + 4.5f * powf(0.5 * (1.f + costheta), 8.f)) / (4.f * kPi);
}

}  // namespace orc
#endif
