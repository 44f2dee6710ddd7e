// pvol_liargs.h -- the argument block of the Li() kernels, ONE definition for the kernels (pvol_march.hip) and the host code that
// fills it (pvol_api.hip, pvol_tile.hip through pvol_host.h)
#ifndef PVOL_LIARGS_H
#define PVOL_LIARGS_H
#include "pvol_dev.h"
struct LiArgs {
    const DevScene *scene;
    const pvol_ray *rays;
    pvol_stream *streams;
    uint32_t nStreams;
    uint32_t nRays;
    int outputKind;
    float *out;
    uint32_t *draws;
    const uint32_t *initState;  // optional: nStreams x 625 words (mt[624], mti) instead of seeding
    uint32_t *finalState;       // optional: same layout, written back
    DevCounters *counters;
    int transmittanceOnly;
    uint32_t *chunkCounter;     // li_par_kernel: next chunk of CHUNK_RAYS rays
    uint32_t *needSeq;          // li_par_kernel sets it when a ray reaches the roulette; li_seq_kernel runs only if set (gate)
    int gated;                  // li_seq_kernel: 1 = return at once unless *needSeq
    // resolve + replay (scenes where drawn values matter): rays are processed in slices, slice k of a stream =
    // its rays [k*sliceM, (k+1)*sliceM); the MT state of every stream persists in `state` between slices
    unsigned char *records;     // [nStreams * sliceM] slots of recStride bytes
    uint32_t recStride, sliceM, sliceK;
    uint32_t *state;            // [nStreams][625]
    float grpGuess;             // li_group_kernel: search radius^2 = this x the guessed k-th distance^2
    int liteResolve;            // no march step can reach the roulette: geometry pre-pass + RNG-only resolve
    DeferRec *defer;            // li_group_kernel: lookups handed to li_fixup_kernel
    uint32_t *deferCount;
    uint32_t deferCap;
    float *tauOut;              // optional: per ray the optical length of Li()'s last march step (T = exp(-sigma_t * tau))
    int32_t fixGroup;           // nused beyond the bucket plan: the hand-over list is padded to 64-slot runs for li_fixup_group_kernel
    float fxgWiden, fxgAim;     // li_fixup_group_kernel's radius policy (0 = the defaults 1.3 / 1.4): first radius^2 = widen x the probe's, the probe aims at aim x nused photons
};
#endif
