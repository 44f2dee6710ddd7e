// pvol_rng_dev.h -- MT19937 (core/rng.cpp:43-107) held in LDS by ONE wave: shared by the march kernels (one stream per
// render tile) and the photon shooter (one stream per PhotonShootingTask).
#ifndef PVOL_RNG_DEV_H
#define PVOL_RNG_DEV_H
#include "pvol_math.h"

// ------------------------------------------------------------------------------------------ RNG
// SEQ: MT19937 state in LDS (core/rng.cpp:43-107).  !SEQ: only the number of draws is tracked.
struct Rng {
    uint32_t *mt;   // LDS, 624 words (SEQ only)
    int mti;        // wave-uniform
    unsigned long long draws;
};
__device__ __forceinline__ uint32_t mt_twist(uint32_t a, uint32_t b) {
    uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
// core/rng.cpp:80-92, 64 words per step.  Every kernel that owns an MT19937 state runs ONE wave per workgroup, and
// the LDS operations of a wave execute in program order: within a step all lanes read before any lane writes (one
// load instruction, then one store instruction), and a later step's reads see the earlier steps' writes -- exactly the
// "old" and "new" words of the serial loop, without barriers (they cost ~4x in this latency-bound routine).
__device__ void mt_regenerate(uint32_t *mt, int lane) {
    for (int base = 0; base < MT_N - MT_M; base += LANES) {
        int kk = base + lane;
        if (kk < MT_N - MT_M) {
            uint32_t v = mt[kk + MT_M] ^ mt_twist(mt[kk], mt[kk + 1]);
            mt[kk] = v;
        }
    }
    for (int base = MT_N - MT_M; base < MT_N - 1; base += LANES) {
        int kk = base + lane;
        if (kk < MT_N - 1) {
            uint32_t v = mt[kk + (MT_M - MT_N)] ^ mt_twist(mt[kk], mt[kk + 1]);
            mt[kk] = v;
        }
    }
    if (lane == 0) mt[MT_N - 1] = mt[MT_M - 1] ^ mt_twist(mt[MT_N - 1], mt[0]);
    __syncthreads();
}
__device__ void mt_seed(uint32_t *mt, uint32_t seed, int lane) {  // core/rng.cpp:43-55
    uint32_t x = seed;
    if (lane == 0) mt[0] = x;
    for (int i = 1; i < MT_N; ++i) {
        x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
        if (lane == (i & 63)) mt[i] = x;
    }
    __syncthreads();
}
template <bool SEQ> __device__ __forceinline__ uint32_t rng_uint(Rng &r, int lane) {
    ++r.draws;
    if (!SEQ) return 0u;
    if (r.mti >= MT_N) { mt_regenerate(r.mt, lane); r.mti = 0; }
    uint32_t y = r.mt[r.mti++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
template <bool SEQ> __device__ __forceinline__ float rng_float(Rng &r, int lane) {
    return (rng_uint<SEQ>(r, lane) & 0xffffff) / float(1 << 24);   // core/rng.cpp:59-65
}
template <bool SEQ> __device__ __forceinline__ void rng_skip(Rng &r, unsigned long long n, int lane) {
    r.draws += n;
    if (!SEQ) return;
    while (n > 0) {
        if (r.mti >= MT_N) { mt_regenerate(r.mt, lane); r.mti = 0; }
        unsigned long long avail = (unsigned long long)(MT_N - r.mti);
        unsigned long long take = n < avail ? n : avail;
        r.mti += (int)take;
        n -= take;
    }
}


// draw j of the next `cnt` (<= 64) draws lands in lane j; the stream advances by cnt
__device__ __forceinline__ uint32_t rng_bulk(Rng &r, int cnt, int lane) {
    uint32_t out = 0u;
    int done = 0;
    r.draws += (unsigned long long)cnt;
    while (done < cnt) {
        if (r.mti >= MT_N) { mt_regenerate(r.mt, lane); r.mti = 0; }
        const int take = min(cnt - done, MT_N - r.mti);
        if (lane >= done && lane < done + take) {
            uint32_t y = r.mt[r.mti + lane - done];
            y ^= (y >> 11);
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= (y >> 18);
            out = y;
        }
        r.mti += take;
        done += take;
    }
    return out;
}
#endif
