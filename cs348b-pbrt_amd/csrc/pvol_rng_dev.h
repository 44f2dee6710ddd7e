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
// core/rng.cpp:80-92 by ONE wave, 64 words per instruction.  The serial loop's word kk needs the OLD words kk, kk+1 and -- for
// kk < 227 -- the old word kk+397, else the NEW word kk-227.  The LDS operations of a wave execute in program order, so no
// workgroup barrier is needed (they cost ~4x in this latency-bound routine) -- but the COMPILER must not move a load above a
// store of another lane that feeds it: to the compiler every thread's own addresses are distinct, so after unrolling it may hoist
// "mt[kk-227]" of a late step above the early step's store that (through another lane) produces that word.  (Round 3 found this:
// the same source gave a correct schedule inside tile_kernel and a wrong one inside tile_mw_kernel.)  Hence four groups of steps
// whose inputs are all complete before the group starts -- A: words 0..226 (old words only), B: 227..418 (new 0..191),
// C: 419..610 (new 192..383), D: 611..623 (new 384..396, and new word 0 for the last) -- every group loads everything into
// registers, then stores, with compiler barriers in between (no instruction, no wait: the hardware order does the rest).
// WG1: the workgroup IS the wave that owns the state (every march kernel, the shooter, the one-wave tile kernel): __syncthreads.
// !WG1: the owning wave is one of several in its workgroup (the multi-wave tile pre-pass) and must not touch the workgroup
// barrier on its own: the same release / acquire fences as __syncthreads() without the s_barrier -- the LDS operations of
// ONE wave execute in program order, so its own later reads see its earlier writes once the fence has drained the queue.
template <bool WG1> __device__ __forceinline__ void rng_sync() {
    if (WG1) __syncthreads();
    else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}
// Two forms, both independent of the compiler's schedule.  Which kernel gets which was decided by measurement: the tile pre-pass
// is ~13 k instructions (more than the instruction cache holds) with the routine at a dozen call sites, and its time moved
// between 181 and 373 ms (one-wave kernel, C2 frame) with the form, the inlining and the register allocation around it --
// forced inlining and forced calls were both slower than leaving the choice to the compiler.
//  * one-wave workgroups (march kernels, shooter, the one-wave tile pre-pass): the serial loop's own order, 64 words per step, every
//    access of the state VOLATILE -- program order among volatile accesses is the one thing the compiler may not change.
//    (One-wave tile pre-pass of the C2 frame: 226 ms; the unsafe round-2 form 181 ms; barrier forms 335 ms.)
//  * the multi-wave tile pre-pass: two LDS round trips.  The OLD inputs of words 0..610 are loaded first; words 0..226 (A)
//    follow from those alone, words 227..418 (B) need the NEW words 0..191, which THIS lane has just computed (kk - 227 =
//    64 j + lane): registers, no load.  After A and B are stored, words 419..623 take the new words 192..396 (and the new word 0)
//    -- other lanes' results -- through LDS.  Compiler barriers between "last old load | first store" and "stores of A, B |
//    loads of the new words"; every load unconditional (clamped index).  (8 waves per task, 512 tasks: 98 ms; volatile loop 217.)
#define PVOL_COMPILER_BARRIER() asm volatile("" ::: "memory")
template <bool WG1 = true, bool TWO_RT = !WG1> __device__ void mt_regenerate(uint32_t *mt_, int lane) {
    if (!TWO_RT) {
        volatile uint32_t *mt = mt_;
        for (int base = 0; base < MT_N - MT_M; base += LANES) {
            const int kk = base + lane;
            if (kk < MT_N - MT_M) { const uint32_t v = mt[kk + MT_M] ^ mt_twist(mt[kk], mt[kk + 1]); mt[kk] = v; }
        }
        for (int base = MT_N - MT_M; base < MT_N - 1; base += LANES) {
            const int kk = base + lane;
            if (kk < MT_N - 1) { const uint32_t v = mt[kk + (MT_M - MT_N)] ^ mt_twist(mt[kk], mt[kk + 1]); mt[kk] = v; }
        }
        if (lane == 0) mt[MT_N - 1] = mt[MT_M - 1] ^ mt_twist(mt[MT_N - 1], mt[0]);
    } else {
        uint32_t *mt = mt_;
        constexpr int NA = MT_N - MT_M;   // 227
        uint32_t a0[4], a1[4], am[4], b0[3], b1[3], c0[3], c1[3];
        const int k3 = min(3 * LANES + lane, NA - 1);   // group A, j = 3: words 192 .. 226 (lanes 0 .. 34)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kk = j < 3 ? LANES * j + lane : k3;
            a0[j] = mt[kk]; a1[j] = mt[kk + 1]; am[j] = mt[kk + MT_M];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int kb = NA + LANES * j + lane, kc = NA + 3 * LANES + LANES * j + lane;
            b0[j] = mt[kb]; b1[j] = mt[kb + 1];
            c0[j] = mt[kc]; c1[j] = mt[kc + 1];
        }
        const int kdRaw = NA + 6 * LANES + lane;   // 611 + lane, lanes 0 .. 12
        const bool dOn = kdRaw < MT_N;
        const int kd = min(kdRaw, MT_N - 1);
        const uint32_t d0 = mt[kd], d1 = mt[min(kd + 1, MT_N - 1)];
        PVOL_COMPILER_BARRIER();
        uint32_t va[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) va[j] = am[j] ^ mt_twist(a0[j], a1[j]);
#pragma unroll
        for (int j = 0; j < 3; ++j) mt[LANES * j + lane] = va[j];
        if (3 * LANES + lane < NA) mt[3 * LANES + lane] = va[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) mt[NA + LANES * j + lane] = va[j] ^ mt_twist(b0[j], b1[j]);   // new word kk - 227 is this lane's va[j]
        PVOL_COMPILER_BARRIER();
        uint32_t cn[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) cn[j] = mt[3 * LANES + LANES * j + lane];   // new words 192 .. 383
        const uint32_t dn = mt[min(6 * LANES + lane, MT_N - 1)];                   // new words 384 .. 396 (lanes 0 .. 12)
        const uint32_t n0 = mt[0];
        PVOL_COMPILER_BARRIER();
#pragma unroll
        for (int j = 0; j < 3; ++j) mt[NA + 3 * LANES + LANES * j + lane] = cn[j] ^ mt_twist(c0[j], c1[j]);
        if (dOn) mt[kd] = dn ^ mt_twist(d0, kdRaw + 1 < MT_N ? d1 : n0);
    }
    rng_sync<WG1>();
}
template <bool WG1 = true> __device__ void mt_seed(uint32_t *mt, uint32_t seed, int lane) {  // core/rng.cpp:43-55
    uint32_t x = seed;
    if (lane == 0) mt[0] = x;
    for (int i = 1; i < MT_N; ++i) {
        x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
        if (lane == (i & 63)) mt[i] = x;
    }
    rng_sync<WG1>();
}
template <bool SEQ, bool WG1 = true> __device__ __forceinline__ uint32_t rng_uint(Rng &r, int lane) {
    ++r.draws;
    if (!SEQ) return 0u;
    if (r.mti >= MT_N) { mt_regenerate<WG1>(r.mt, lane); r.mti = 0; }
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
// TWO_RT: the two-round-trip regeneration at THIS site whatever the workgroup's shape (the tile pre-pass's per-trip advance: ~31
// regenerations back to back, 0.33 us each alone against 1.4 us for the volatile loop, tools/mt_probe.hip)
template <bool SEQ, bool WG1 = true, bool TWO_RT = !WG1> __device__ __forceinline__ void rng_skip(Rng &r, unsigned long long n, int lane) {
    r.draws += n;
    if (!SEQ) return;
    while (n > 0) {
        if (r.mti >= MT_N) { mt_regenerate<WG1, TWO_RT>(r.mt, lane); r.mti = 0; }
        unsigned long long avail = (unsigned long long)(MT_N - r.mti);
        unsigned long long take = n < avail ? n : avail;
        r.mti += (int)take;
        n -= take;
    }
}


// draw j of the next `cnt` (<= 64) draws lands in lane j; the stream advances by cnt
template <bool WG1 = true> __device__ __forceinline__ uint32_t rng_bulk(Rng &r, int cnt, int lane) {
    uint32_t out = 0u;
    int done = 0;
    r.draws += (unsigned long long)cnt;
    while (done < cnt) {
        if (r.mti >= MT_N) { mt_regenerate<WG1>(r.mt, lane); r.mti = 0; }
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
