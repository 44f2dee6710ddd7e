// pvol_group_dev.h -- included by pvol_march.hip.  li_group_kernel: PhotonVolumeIntegrator::Li for 64 camera rays
// at a time, ONE RAY PER LANE, for homogeneous isotropic media under the conditions of li_par_kernel (no drawn
// value reaches the result).  The k-NN gathers of the 64 lanes at march step j are neighbours in space -- the
// rays of a chunk are first ordered by their scatter offset, so a group's query points lie within a few
// hundredths of a unit of each other -- and share one candidate search:
//   1. STAGE   the photons within sqrt(T) + rho of the group's centre (T = guessed radius^2, rho = spread of the
//              query points) are found cooperatively in the cell grid and parked in LDS ("LDS-staged bucket");
//   2. SELECT  every lane finds ITS exact k-th smallest DistanceSquared over the bucket.  Pass 1 histograms the
//              values over [0, T) into 64 four-bit counters per lane (LDS, one column per lane) and a prefix scan
//              finds the bin b* of the k-th.  Pass 2 leaves, per lane, one BIT per bucket slot: "bin < b*" (a member
//              for sure) and "bin <= b*".  The few photons of bin b* itself are then looked at individually: their
//              values sorted in registers give the exact k-th distance, ties are taken in bucket order, and their
//              member bits are added.
//   3. FLUX    for every bucket slot that is a member for SOME lane, the 128-byte alpha row comes through the scalar
//              cache into SGPRs (wave-uniform address) and is added with v_pk_fma_f32(row, bit, acc): the exact
//              addition for members, a no-op for the others.  No distance is evaluated a third time.
// Both passes evaluate kdtree.h:180's DistanceSquared per lane in the reference's operation order, so the k-NN sets
// are the reference's; only the order of the flux additions differs (as in lphoton).
//
// Radiance recurrence without a 30-bin carry.  The reference marches Lv = w_j + Tr_j * Lv (photonvolume.cpp:150-218)
// with Tr_j = exp(-sigma_t * len_j) ASSIGNED per step, so the result is  Lv = sum_j w_j * exp(-sigma_t * R_j),
// R_j = sum_{m > j} len_m.  A cheap geometric pre-loop gives every lane its total length; the march then adds each
// step's contribution straight into X, Y, Z (or into the 30 bins for spectral output) -- 30 fewer live registers per
// lane -- and makes every lookup's contribution INDEPENDENT of the others: a lane whose lookup does not fit the plan
// (cold start, radius guess off twice, crowded bin, bucket overflow) appends {ray, point, R_j, step} to a list
// instead of running the exact wave-cooperative lookup inline, and li_fixup_kernel adds those terms afterwards
// (lphoton(), always exact).  The hot kernel therefore carries neither lphoton's registers nor its LDS: three waves
// per SIMD.
#define GRP_CH 512  // rays per chunk (ordered by scatter_u, then cut into groups of 64)
#define GRP_CAP 256   // bucket capacity (photons)
#define GRP_NW (GRP_CAP / 32)     // 32-bit mask words per lane
#define GRP_BINS 64   // histogram bins over [0, T), 4-bit counters (eight per LDS word) + one overflow word
#define GRP_PITCH (GRP_CAP + 4)   // floats per bucket component (x | y | z | photon index), padded for the 4-wide passes
#define GRP_WIDEN 1.7f   // a lane's search radius^2 may grow to this multiple of its guess where the bucket covers it
#define GRP_MINI 8    // photons of the k-th's bin a lane can rank
#ifndef GRP_CORE_MIN
#define GRP_CORE_MIN 8 // slots that are members for every served lane of a group-step are summed once for the wave when there are at least this many (GRP_CAP switches it off)
#endif
#define GRP_TRI_ROWS 8   // scenes with a distant light and at most this many triangles test shadow rays against precomputed rows
#define GRP_WPE 3     // waves per SIMD the register allocation must leave room for
#define GRP_U_BYTES 4096   // shared scratch: stage paint list | histogram | chunk ordering | mini lists (never live together)
#define GRP_PLAN_KMAX 100  // nused above this: the bucket (GRP_CAP photons) cannot hold a k-NN ball, only the "all photons inside maxDist" case
// Two more template switches widen the kernel beyond C2:
//   REPLAY  scenes where drawn VALUES reach the result (> 1 light: the per-step light choice; VolumeGrid: the tau() offsets): the
//           rays come slice by slice with the per-step records of the RNG pre-pass (pvol_march.hip: li_geo / li_resolve_lite /
//           li_resolve / the fused tile pre-pass), one record slot per ray, and the kernel neither counts draws nor touches streams;
//   GRID    VolumeGridDensity (volumes/volumegrid.cpp:39-57): trilinear density per lane, DensityRegion::tau (core/volume.cpp:
//           296-310) stepped per lane with the recorded offsets.  tau_b = sigma_t_b x (a scalar), so the forward formulation holds.
// nused > GRP_PLAN_KMAX (C3: 500) keeps only the fixed-radius plan: when everything within maxDist fits the bucket the lanes
// with fewer than nused photons are served here (most of C3's lookups: < 10 photons -> 0), the dense ones go to li_fixup_kernel.

struct GroupLds {
    float *pos;             // bucket, SoA: x[GRP_PITCH] | y | z | photon index bits
    uint32_t *hist;         // [GRP_BINS / 8 + 1][64] packed 4-bit counters, one column per lane (U region)
    float *miniD;           // [GRP_MINI][64] values of the k-th's bin, one column per lane (U region)
    float *ubuf;            // GRP_CH scatter offsets (U region)
    unsigned short *order;  // GRP_CH: chunk-local ray index by rank of scatter offset
    float *cst;             // 12 x 32 floats: sigA, sigS, le, albedo, light-0 intensity, 1/sigS, CIE X, Y, Z weights, sigT, sigA le, albedo / sigS
    float *trows;           // GRP_TRI_ROWS x 16 floats: per-triangle shadow-ray precomputation for a distant light
    float *lint;            // PVOL_MAX_LIGHTS x 32 floats: every light's intensity spectrum (REPLAY: the light differs per lane and step)
};

// DensityRegion::tau (core/volume.cpp:296-310) of o + t d, t in [mint, maxt], for ONE lane: the scalar L with tau_b = sigma_t_b * L
// (sum of densities x step; the reference adds sigma_t * density per sample, the same up to rounding)
__device__ float grid_tau_lane(const DevScene &S, V3 o, V3 d, float mint, float maxt, float stepSize, float u) {
    const float length = len(d);
    if (length == 0.f) return 0.f;
    RayD rn;
    rn.o = o; rn.d = vdiv(d, length); rn.mint = mint * length; rn.maxt = maxt * length;
    float t0, t1;
    if (!vol_intersect(S, rn, &t0, &t1)) return 0.f;
    float dsum = 0.f;
    t0 += u * stepSize;
    while (t0 < t1) {
        dsum += grid_density(S, xform_point(S.w2v, rn.o + rn.d * t0));
        t0 += stepSize;
    }
    return dsum * stepSize;
}

// Photons within Rs of c -> LDS bucket of CAP slots (SoA, pitch CAP + 4).  Returns the count, or -1 if the bucket would overflow.
// IDX_ONLY: the bucket is the list of photon indices alone (CAP + 8 words, padded with index 0 to a multiple of eight).
template <int CAP, bool FINE = false, bool IDX_ONLY = false>
__device__ int stage_bucket_g(const GridView &S, Gather &G, float *bucket, V3 c, float Rs, int lane, unsigned long long &tested) {
    constexpr int GRP_PITCH_ = CAP + 4;
    float *bX = bucket, *bY = bucket + GRP_PITCH_, *bZ = bucket + 2 * GRP_PITCH_, *bI = IDX_ONLY ? bucket : bucket + 3 * GRP_PITCH_;
    const float T = Rs * Rs;
    const GridRows rows = grid_rows<FINE>(S, c, Rs, 0);
    int count = 0;
    for (int rb = 0; rb < rows.nrows; rb += LANES) {
        uint32_t start, rlen;
        grid_row_range<FINE>(S, rows, rb + lane, c, T, &start, &rlen);
        const uint32_t lenS = rlen <= PAINT_ROW ? rlen : 0u;
        uint32_t incl = lenS;
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, true);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, true);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, true);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, true);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xa, 0xf, false);
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xc, 0xf, false);
        const uint32_t off = incl - lenS;
        const uint32_t total = (uint32_t)lane_i((int)incl, LANES - 1);
        tested += total;
        uint64_t longRows = __ballot(rlen > PAINT_ROW);
        if (total) {
            __syncthreads();
            for (uint32_t it = 0; it < PAINT_ROW; ++it) {
                if (!wave_any(it < lenS)) break;
                if (it < lenS) G.paint[off + it] = start + it;
            }
            __syncthreads();
        }
        uint32_t segBase = 0u, segLen = total;
        bool painted = true;
        for (;;) {
            for (uint32_t cb = 0; cb < segLen; cb += LANES * NCH) {
                f4 P[NCH];
                uint32_t I[NCH];
                bool on[NCH];
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    on[k] = false;
                    I[k] = 0u;
                    P[k] = mk4(0.f);
                    if (cb + k * LANES < segLen) {
                        uint32_t g = cb + k * LANES + lane;
                        on[k] = g < segLen;
                        if (on[k]) {
                            I[k] = painted ? G.paint[g] : segBase + g;
                            P[k] = S.pos4[I[k]];
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < NCH; ++k) {
                    if (cb + k * LANES >= segLen) continue;
                    float dx = P[k].x - c.x, dyy = P[k].y - c.y, dzz = P[k].z - c.z;
                    float d2 = dx * dx + dyy * dyy + dzz * dzz;
                    bool acc = on[k] && d2 < T;
                    uint64_t m = __ballot(acc);
                    if (m) {
                        const int add = __popcll(m);
                        if (count + add > CAP) return -1;
                        if (acc) {
                            const int at = count + (int)lanes_below(m, lane);
                            if (!IDX_ONLY) { bX[at] = P[k].x; bY[at] = P[k].y; bZ[at] = P[k].z; }
                            bI[at] = __uint_as_float(I[k]);
                        }
                        count += add;
                    }
                }
            }
            if (!longRows) break;
            const int j = __ffsll((unsigned long long)longRows) - 1;
            longRows &= longRows - 1;
            painted = false;
            segBase = (uint32_t)lane_i((int)start, j);
            segLen = (uint32_t)lane_i((int)rlen, j);
            tested += segLen;
        }
    }
    if (IDX_ONLY) { if (lane < 8) bI[count + lane] = 0.f; }
    else if (lane < 4) { bX[count + lane] = 3.0e18f; bY[count + lane] = 3.0e18f; bZ[count + lane] = 3.0e18f; bI[count + lane] = 0.f; }   // pad to a multiple of four: never inside any radius
    __syncthreads();
    return count;
}
__device__ __forceinline__ int stage_bucket(const DevScene &S, Gather &G, float *bucket, V3 c, float Rs, int lane, unsigned long long &tested) {
    const GridView g = volume_grid(S);
    return stage_bucket_g<GRP_CAP>(g, G, bucket, c, Rs, lane, tested);
}

typedef float nf4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef GRP_PHASE_DIAG
#define GRP_PHASE_DIAG 0   // 1 (stats build, timing only): diag0 / diag1 / diag3 / diag4 become the cycles of the prefix scan / pass 1 / pass 2 / bin b* ranking
#endif

// OR over the wave, every lane gets it (DPP + permlane swaps, no LDS)
__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_QUAD_XOR1, 0xf, 0xf, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_QUAD_XOR2, 0xf, 0xf, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_HALF_MIRROR, 0xf, 0xf, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, DPP_ROW_ROR8, 0xf, 0xf, true);
    { auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); v = r[0] | r[1]; }
    { auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); v = r[0] | r[1]; }
    return v;
}

// Histogram bin of a DistanceSquared value: trunc(d2 * scale) clamped to GRP_BINS (= "not below T").  EXACT: lanes that
// search the full radius must agree with the reference's `dist2 < maxDistSquared` (kdtree.h:180) to the last bit, so the
// comparison itself decides in-range and the multiplication only places the value.
template <bool EXACT>
__device__ __forceinline__ uint32_t grp_bin(float d2, float scale, float Tl) {
    uint32_t bin = (uint32_t)fminf(d2 * scale, (float)GRP_BINS);   // fminf also absorbs the sentinels' inf / NaN
    if (EXACT) bin = d2 < Tl ? min(bin, (uint32_t)(GRP_BINS - 1)) : (uint32_t)GRP_BINS;
    return bin;
}

typedef float nf2 __attribute__((ext_vector_type(2)));
// two coordinates minus one per-lane value: ONE v_pk_add_f32 with the subtrahend negated and taken from the low (x) or high (y)
// half of the pair pp for both results -- a + (-b) is a - b bit for bit; the compiler itself writes two v_sub_f32 here
__device__ __forceinline__ nf2 pk_sub_lo(nf2 a, nf2 pp) {
    nf2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(pp));
    return r;
}
__device__ __forceinline__ nf2 pk_sub_hi(nf2 a, nf2 pp) {
    nf2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(pp));
    return r;
}
// DistanceSquared(photon.p, p) (kdtree.h:180) of four bucket photons: same operations in the same order per element
__device__ __forceinline__ nf4 grp_dist4(nf4 X, nf4 Y, nf4 Z, nf2 pxy, nf2 pz_) {
    const nf2 x0 = pk_sub_lo(__builtin_shufflevector(X, X, 0, 1), pxy), x1 = pk_sub_lo(__builtin_shufflevector(X, X, 2, 3), pxy);
    const nf2 y0 = pk_sub_hi(__builtin_shufflevector(Y, Y, 0, 1), pxy), y1 = pk_sub_hi(__builtin_shufflevector(Y, Y, 2, 3), pxy);
    const nf2 z0 = pk_sub_lo(__builtin_shufflevector(Z, Z, 0, 1), pz_), z1 = pk_sub_lo(__builtin_shufflevector(Z, Z, 2, 3), pz_);
    const nf2 d0 = x0 * x0 + y0 * y0 + z0 * z0, d1 = x1 * x1 + y1 * y1 + z1 * z1;
    return nf4{d0.x, d0.y, d1.x, d1.y};
}

// Pass 1: per-lane histogram of DistanceSquared over the bucket (padded to a multiple of four with far-away sentinels).
template <bool EXACT>
__device__ __forceinline__ void grp_pass1(const float *bX, const float *bY, const float *bZ, int Mb, nf2 pxy, nf2 pz_, float scale,
                                          float Tl, uint32_t *histLane) {
    nf4 nX = *reinterpret_cast<const nf4 *>(bX), nY = *reinterpret_cast<const nf4 *>(bY), nZ = *reinterpret_cast<const nf4 *>(bZ);
    for (int c0 = 0; c0 < Mb; c0 += 4) {
        // the next quartet's coordinates are requested before this one's arithmetic (the padded bucket makes the read past
        // the end harmless)
        const nf4 dd = grp_dist4(nX, nY, nZ, pxy, pz_);
        nX = *reinterpret_cast<const nf4 *>(bX + c0 + 4); nY = *reinterpret_cast<const nf4 *>(bY + c0 + 4); nZ = *reinterpret_cast<const nf4 *>(bZ + c0 + 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t bin = grp_bin<EXACT>(dd[u], scale, Tl);
            // word = bits 3..6 of the bin as ONE v_bfe_u32 (the plain shift-and-mask form costs the compiler a shift, a mask and an add)
            uint32_t word;
            asm("v_bfe_u32 %0, %1, 3, 4" : "=v"(word) : "v"(bin));   // (an intrinsic is folded back into shift + mask)
            atomicAdd(&histLane[word * LANES], 1u << ((bin & 7u) << 2));   // ds_add_u32, no return
        }
    }
}

// The smallest float x >= 0 with grp_bin(x) >= b.  grp_bin is monotone, so "bin < b" IS "d2 < x": pass 2 compares distances with two
// such thresholds and never forms a bin.  *okp is cleared if the few steps around b / scale did not land on it (never seen; the
// lookup would go to the hand-over list).
template <bool EXACT>
__device__ __forceinline__ float grp_theta(uint32_t b, float scale, float Tl, bool *okp) {
    if (b == 0u) return 0.f;
    if (b > (uint32_t)GRP_BINS) return INFINITY;
    float x = (float)b / scale;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float xm = __uint_as_float(__float_as_uint(x) - 1u);
        if (x > 0.f && grp_bin<EXACT>(xm, scale, Tl) >= b) x = xm;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (grp_bin<EXACT>(x, scale, Tl) < b) x = __uint_as_float(__float_as_uint(x) + 1u);
    const bool good = grp_bin<EXACT>(x, scale, Tl) >= b && (!(x > 0.f) || grp_bin<EXACT>(__uint_as_float(__float_as_uint(x) - 1u), scale, Tl) < b);
    if (!good) *okp = false;
    return x;
}

// m = 2 m + (d2 < th): one compare and one add-with-carry per slot; the slots of a word arrive most significant first
__device__ __forceinline__ void grp_bit_lt(uint32_t &m, float d2, float th) {
    asm("v_cmp_lt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(d2), "v"(th) : "vcc");
}

// Pass 2: per-lane bit masks over the bucket slots: lt = "bin < bstar" = "d2 < th1", le = "bin <= bstar" = "d2 < th2".
template <bool EXACT>
__device__ __forceinline__ void grp_pass2(const float *bX, const float *bY, const float *bZ, int Mb, nf2 pxy, nf2 pz_, float th1, float th2,
                                          uint32_t (&lt)[GRP_NW], uint32_t (&le)[GRP_NW], float &dmax) {
#pragma unroll
    for (int wd = 0; wd < GRP_NW; ++wd) {
        uint32_t mlt = 0u, mle = 0u;
        if (wd * 32 < Mb) {   // wave-uniform
            const int cend = min(Mb, wd * 32 + 32);
            nf4 nX = *reinterpret_cast<const nf4 *>(bX + wd * 32), nY = *reinterpret_cast<const nf4 *>(bY + wd * 32), nZ = *reinterpret_cast<const nf4 *>(bZ + wd * 32);
            int n = 0;
            for (int c0 = wd * 32; c0 < cend; c0 += 4) {
                const nf4 dd = grp_dist4(nX, nY, nZ, pxy, pz_);
                nX = *reinterpret_cast<const nf4 *>(bX + c0 + 4); nY = *reinterpret_cast<const nf4 *>(bY + c0 + 4); nZ = *reinterpret_cast<const nf4 *>(bZ + c0 + 4);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    grp_bit_lt(mlt, dd[u], th1);
                    grp_bit_lt(mle, dd[u], th2);
                    if (EXACT) dmax = fmaxf(dmax, dd[u] < th1 ? dd[u] : 0.f);
                }
                n += 4;
            }
            // slot i of the n processed sits at bit n - 1 - i: reverse, then drop the 32 - n empty places
            mlt = __builtin_bitreverse32(mlt) >> (32 - n);
            mle = __builtin_bitreverse32(mle) >> (32 - n);
        }
        lt[wd] = mlt;
        le[wd] = mle;
    }
}

__device__ __forceinline__ uint32_t nib_sum(uint32_t w) {   // sum of the eight 4-bit counters of a word
    const uint32_t t = (w & 0x0f0f0f0fu) + ((w >> 4) & 0x0f0f0f0fu);
    return (t * 0x01010101u) >> 24;
}

template <bool STATS, bool SPECTRAL, bool REPLAY, bool GRID>
__global__ __launch_bounds__(LANES, GRP_WPE) void li_group_kernel(LiArgs A) {
    extern __shared__ __align__(16) unsigned char lds[];
    const DevScene &S = *A.scene;
    const int lane = threadIdx.x;
    // LDS plan (11.7 KB: twelve waves per CU): prevRk | order | bucket | U | cst | trows
    float *prevRk = reinterpret_cast<float *>(lds);
    GroupLds L;
    L.order = reinterpret_cast<unsigned short *>(prevRk + PREV_N);
    L.pos = reinterpret_cast<float *>(L.order + GRP_CH);
    unsigned char *U = reinterpret_cast<unsigned char *>(L.pos + 4 * GRP_PITCH);
    Gather G;
    G.cap = 0; G.cd = 0; G.ci = 0;
    G.paint = reinterpret_cast<uint32_t *>(U);
    L.hist = reinterpret_cast<uint32_t *>(U);
    L.miniD = reinterpret_cast<float *>(U);
    L.ubuf = reinterpret_cast<float *>(U);
    L.cst = reinterpret_cast<float *>(U + GRP_U_BYTES);
    L.trows = L.cst + 12 * 32;
    L.lint = L.trows + GRP_TRI_ROWS * 16;
    for (int i = lane; i < PREV_N; i += LANES) prevRk[i] = 0.f;
    const int q = lane & 7;
    const f4 sigA4 = ld4(S.sigA, q), sigS4 = ld4(S.sigS, q);
    const f4 sigT4 = sigA4 + sigS4;
    const f4 Y4 = ld4(S.cieY, q);
    const int nLights = S.nLights;
    const float ySa1 = spec_y(sigA4, Y4), ySs1 = spec_y(sigS4, Y4);
    const bool blackS1 = spec_is_black(sigS4);
    unsigned int lightBlackMask = 0u;   // lights whose intensity spectrum is black
    for (int l = 0; l < nLights; ++l) if (spec_is_black(ld4(S.lights[l].intensity, q))) lightBlackMask |= 1u << l;
    if (REPLAY) for (int i = lane; i < PVOL_MAX_LIGHTS * 32; i += LANES) L.lint[i] = ((i >> 5) < nLights && (i & 31) < 30) ? S.lights[i >> 5].intensity[i & 31] : 0.f;
    float sigTmax = fmaxf(fmaxf(sigT4.x, sigT4.y), fmaxf(sigT4.z, sigT4.w));
    sigTmax = wave_max(sigTmax);
    if (lane < 32) {
        const float a = S.sigA[lane], s = S.sigS[lane];
        L.cst[lane] = a;
        L.cst[32 + lane] = s;
        L.cst[64 + lane] = S.le[lane];
        L.cst[96 + lane] = lane < 30 ? __fdividef(s, a + s) : 0.f;
        L.cst[128 + lane] = nLights > 0 ? S.lights[0].intensity[lane] : 0.f;
        L.cst[160 + lane] = lane < 30 ? __builtin_amdgcn_rcpf(s) : 0.f;
        L.cst[192 + lane] = lane < 30 ? S.cieX[lane] : 0.f;
        L.cst[224 + lane] = lane < 30 ? S.cieY[lane] : 0.f;
        L.cst[256 + lane] = lane < 30 ? S.cieZ[lane] : 0.f;
        L.cst[288 + lane] = a + s;
        L.cst[320 + lane] = a * S.le[lane];
        L.cst[352 + lane] = lane < 30 ? __fdividef(s, a + s) * __builtin_amdgcn_rcpf(s) : 0.f;
    }
    __syncthreads();
    const f4 *cA = reinterpret_cast<const f4 *>(L.cst), *cS = cA + 8, *cI = cA + 32;
    const f4 *cX = cA + 48, *cY = cA + 56, *cZ = cA + 64, *cT = cA + 72, *cAL = cA + 80, *cAR = cA + 88;
    const bool rowsOK = !REPLAY && nLights > 0 && S.lights[0].kind == PVOL_LIGHT_DISTANT && S.nTris <= GRP_TRI_ROWS && !S.bvhNodes && !S.nSpheres;
    if (rowsOK) tri_rows_prepare(S, v3(S.lights[0].dir[0], S.lights[0].dir[1], S.lights[0].dir[2]), L.trows, lane);
    const int k = S.nUsed;
    const float wIso = 1.f / (4.f * K_PI);
    const bool useLiiAny = (ySa1 != 0.0 || ySs1 != 0.0) && !blackS1;   // L_ii reaches the result at all (photonvolume.cpp:208-211)
    const bool fixedR = k > GRP_PLAN_KMAX;   // only the all-photons-inside-maxDist plan (see the header)
    const uint32_t chunksPerSlice = REPLAY ? (A.sliceM + GRP_CH - 1) / GRP_CH : 0u;
    const unsigned long long nChunks = REPLAY ? (unsigned long long)chunksPerSlice * A.nStreams : 0ull;
    const bool gridVol = GRID;
    WaveCounters wc = {};
    unsigned long long tk0 = STATS ? stamp() : 0ull;
    const float *bX = L.pos, *bY = L.pos + GRP_PITCH, *bZ = L.pos + 2 * GRP_PITCH, *bI = L.pos + 3 * GRP_PITCH;
    uint32_t *histLane = L.hist + lane;
    for (;;) {
        uint32_t chunk = 0;
        if (lane == 0) chunk = atomicAdd(A.chunkCounter, 1u);
        chunk = (uint32_t)lane_i((int)chunk, 0);
        unsigned long long r0 = (unsigned long long)chunk * GRP_CH;
        int nIn;
        size_t slot0 = 0;   // REPLAY: record slot of the chunk's first ray
        if (!REPLAY) {
            if (r0 >= A.nRays) break;
            nIn = (int)min((unsigned long long)GRP_CH, (unsigned long long)A.nRays - r0);
        } else {   // chunk = GRP_CH consecutive rays of one stream's slice (as li_replay_kernel cuts them, eight times as long)
            if (chunk >= nChunks) break;
            const uint32_t sidx = chunk / chunksPerSlice, jc = chunk - sidx * chunksPerSlice;
            const uint32_t nr = A.streams[sidx].n_rays, first = A.streams[sidx].first_ray;
            const uint32_t begin = A.sliceK * A.sliceM;
            if (begin >= nr) continue;
            const uint32_t sliceLen = min(nr - begin, A.sliceM);
            const uint32_t l0 = jc * GRP_CH;
            if (l0 >= sliceLen) continue;
            nIn = (int)min(sliceLen - l0, (uint32_t)GRP_CH);
            r0 = (unsigned long long)first + begin + l0;
            slot0 = (size_t)sidx * A.sliceM + l0;
        }
        // ---- order the chunk's rays by scatter offset: lanes of a group then march in near lock-step positions
        __syncthreads();
        // key = step count + scatter offset: rays of one pixel whose lengths straddle a step-count boundary march with
        // different step lengths and drift apart by up to a step; keeping equal counts together keeps the groups compact
        for (int i = lane; i < GRP_CH; i += LANES) {
            float key = 3.0e9f;
            if (i < nIn) {
                const pvol_ray kr = A.rays[r0 + i];
                RayD rr;
                rr.o = v3(kr.o[0], kr.o[1], kr.o[2]); rr.d = v3(kr.d[0], kr.d[1], kr.d[2]); rr.mint = kr.mint; rr.maxt = kr.maxt;
                float ka = 0.f, kb = 0.f;
                const bool kh = S.volKind != PVOL_VOLUME_NONE && vol_intersect(S, rr, &ka, &kb) && (kb - ka) != 0.f;
                key = (kh ? ceilf((kb - ka) / S.stepSize) : 0.f) + kr.scatter_u;
            }
            L.ubuf[i] = key;
        }
        __syncthreads();
        for (int i = lane; i < nIn; i += LANES) {
            const float ui = L.ubuf[i];
            int rank = 0;
            for (int j = 0; j < nIn; ++j) {
                const float uj = L.ubuf[j];
                rank += (uj < ui || (uj == ui && j < i)) ? 1 : 0;
            }
            L.order[rank] = (unsigned short)i;
        }
        __syncthreads();
#pragma unroll 1
        for (int g0 = 0; g0 < nIn; g0 += LANES) {
            const bool have = g0 + lane < nIn;
            const uint32_t local = have ? (uint32_t)L.order[g0 + lane] : 0u;
            const size_t ri = (size_t)r0 + local;
            const pvol_ray pr = A.rays[ri];
            RayRec rec = {0, 0, 0};
            if (REPLAY) rec = ray_rec(A.records + (slot0 + local) * A.recStride, S.maxSteps, gridVol);
            const V3 o = v3(pr.o[0], pr.o[1], pr.o[2]), d = v3(pr.d[0], pr.d[1], pr.d[2]);
            RayD ray;
            ray.o = o; ray.d = d; ray.mint = pr.mint; ray.maxt = pr.maxt;
            float t0 = 0.f, t1 = 0.f;
            const bool hit = have && S.volKind != PVOL_VOLUME_NONE && vol_intersect(S, ray, &t0, &t1) && (t1 - t0) != 0.f;
            // REPLAY: the pre-pass's count (0 for a ray the record plan could not hold: it was reported there)
            const int nS = hit ? (REPLAY ? min((int)rec.hdr[0], (int)ceilf((t1 - t0) / S.stepSize)) : (int)ceilf((t1 - t0) / S.stepSize)) : 0;
            const float step = (hit && nS > 0) ? (t1 - t0) / (int)ceilf((t1 - t0) / S.stepSize) : 0.f;
            const V3 pEntry = o + d * t0;
            const bool inEntry = !GRID && hit && box_inside(S.extLo, S.extHi, xform_point(S.w2v, pEntry));
            const float tStart = t0 + pr.scatter_u * step;
            const int maxN = (int)wave_max((float)nS);   // step counts are far below 2^24
            // ---- pre-loop: this ray's total optical length sum_j len_j (the geometry of the march, nothing else)
            float totalLen = 0.f;
            {
                float tc = tStart;
                V3 pP = pEntry;
                bool inP0 = inEntry;
                for (int j = 0; j < maxN; ++j) {
                    if (j < nS) {
                        const V3 p = o + d * tc;
                        tc += step;
                        const V3 dseg = p - pP;
                        float lenStep;
                        if (GRID) {
                            lenStep = grid_tau_lane(S, pP, dseg, 0.f, 1.f, .5f * S.stepSize, rec.stepU[2 * j]);   // photonvolume.cpp:154
                        } else {
                            const bool inP = box_inside(S.extLo, S.extHi, xform_point(S.w2v, p));
                            if (inP0 && inP) { const V3 a = pP + dseg * 0.f, b = pP + dseg * 1.f; lenStep = len(a - b); }
                            else lenStep = analytic_tau_length(S, pP, dseg, 0.f, 1.f);
                            inP0 = inP;
                        }
                        totalLen += lenStep;
                        pP = p;
                    }
                }
            }
            V3 pPrev = pEntry;
            bool inPrev = inEntry;
            float tcur = tStart;
            float Lv[SPECTRAL ? 32 : 1];
#pragma unroll
            for (int b = 0; b < (SPECTRAL ? 32 : 1); ++b) Lv[b] = 0.f;
            float accX = 0.f, accY = 0.f, accZ = 0.f;
            float lenLast = 0.f, lastRk = 0.f, cumLen = 0.f;
            uint32_t uCount = 0;
            bool bad = false;
            wc.rays += __popcll(__ballot(have));
            for (int j = 0; j < maxN; ++j) {
                const unsigned long long tq0 = (STATS && GRP_PHASE_DIAG == 2) ? stamp() : 0ull;
                const bool act = j < nS;
                const V3 p = o + d * tcur;
                if (act) tcur += step;
                const V3 pv = xform_point(S.w2v, p);
                // density factor of sigma_a / sigma_s / Le at p (homogeneous.h:64-75: inside ? 1 : 0; volume.h:81-92: Density(p))
                const float dens = !act ? 0.f : (GRID ? grid_density(S, pv) : (box_inside(S.extLo, S.extHi, pv) ? 1.f : 0.f));
                const bool inP = dens != 0.f;
                float lenStep = 0.f;   // tau of the step = sigma_t x lenStep
                if (act) {
                    const V3 dseg = p - pPrev;
                    if (GRID) {
                        lenStep = grid_tau_lane(S, pPrev, dseg, 0.f, 1.f, .5f * S.stepSize, rec.stepU[2 * j]);
                    } else if (inPrev && inP) {
                        const V3 a = pPrev + dseg * 0.f, b = pPrev + dseg * 1.f;
                        lenStep = len(a - b);
                    } else {
                        lenStep = analytic_tau_length(S, pPrev, dseg, 0.f, 1.f);
                    }
                    if (!REPLAY && !(lenStep * sigTmax < 6.8f)) bad = true;   // the roulette could fire: sequential kernel (photonvolume.cpp:156-161)
                    cumLen += lenStep;
                }
                // ---- direct lighting geometry (photonvolume.cpp:178-203), light 0 (at most one light here)
                float fallReg = 1.f, d2Reg = 1.f, exitLen = 0.f, ph = 0.f;
                bool lit = false, distant = true;
                const int ln = REPLAY ? (act ? (int)(rec.stepByte[j] & 7u) : 0) : 0;   // the light this step samples (photonvolume.cpp:181-183)
                if (inP && !blackS1 && nLights > 0) {
                    const DevLight &light = S.lights[ln];
                    RayD vis;
                    V3 wo;
                    distant = light.kind == PVOL_LIGHT_DISTANT;
                    if (distant) {
                        wo = v3(light.dir[0], light.dir[1], light.dir[2]);
                        vis.o = p; vis.d = wo; vis.mint = 0.f; vis.maxt = INFINITY;
                    } else {
                        V3 lp = v3(light.pos[0], light.pos[1], light.pos[2]);
                        wo = normalize(lp - p);
                        float dist = len(p - lp);
                        vis.o = p; vis.d = vdiv(lp - p, dist); vis.mint = 0.f; vis.maxt = dist * (1.f - 0.f);
                        d2Reg = len_sq(lp - p);
                        if (light.kind == PVOL_LIGHT_SPOT) {
                            V3 wl = normalize(v3(light.w2l[0] * -wo.x + light.w2l[1] * -wo.y + light.w2l[2] * -wo.z,
                                                 light.w2l[4] * -wo.x + light.w2l[5] * -wo.y + light.w2l[6] * -wo.z,
                                                 light.w2l[8] * -wo.x + light.w2l[9] * -wo.y + light.w2l[10] * -wo.z));
                            float costheta = wl.z;
                            if (costheta < light.cosTotalWidth) fallReg = 0.f;
                            else if (costheta > light.cosFalloffStart) fallReg = 1.f;
                            else {
                                float delta = (costheta - light.cosTotalWidth) / (light.cosFalloffStart - light.cosTotalWidth);
                                fallReg = delta * delta * delta * delta;
                            }
                        }
                    }
                    const bool black = (fallReg == 0.f) || ((lightBlackMask >> ln) & 1u);
                    if (!black && !(rowsOK ? tri_rows_occluded(L.trows, S.nTris, vis.o, vis.d, vis.mint, vis.maxt) : lane_occluded(S, vis))) {
                        lit = true;
                        V3 dv = xform_vector(S.w2v, vis.d);
                        V3 dvInv = v3(1.f / dv.x, 1.f / dv.y, 1.f / dv.z);
                        exitLen = GRID ? grid_tau_lane(S, vis.o, vis.d, vis.mint, vis.maxt, 4.f * S.stepSize, rec.stepU[2 * j + 1])   // photonvolume.cpp:24-27
                                       : inside_exit_length(S, vis.o, vis.d, pv, dvInv, vis.maxt);
                        ph = phase_hg(-d, -wo, S.g);
                        ++uCount;
                    }
                }
                wc.steps += __popcll(__ballot(act));
                if (STATS) wc.unocc += __popcll(__ballot(lit));
                const float kRem = -1.442695041f * (totalLen - cumLen);   // exp(-sigma_t R_j) = exp2(sigma_t * kRem)
                const float stepD = step * dens;
                const unsigned long long tq1 = (STATS && GRP_PHASE_DIAG == 2) ? stamp() : 0ull;
                if (STATS && GRP_PHASE_DIAG == 2) wc.diag0 += tq1 - tq0;
                // ---- k-NN gather of the group
                // Raw flux sums of the step: two 32 x 32 fp32 matrix accumulators (v_mfma_f32_32x32x2_f32, bit for bit an
                // fmaf chain): rows = bins, columns = rays.  Lane l holds column (ray) l & 31 of CA (rays 0..31) and of CB (rays 32..63),
                // register i = bin 8 (i >> 2) + (i & 3) + 4 (l >> 5).  After the attempts 16 v_permlane32_swap leave every lane its OWN
                // ray's 32 bins: CA[i] = bin 8 (i >> 2) + (i & 3), CB[i] = the same + 4.
                f32x16 CA, CB;
#pragma unroll
                for (int b = 0; b < 16; ++b) { CA[b] = 0.f; CB[b] = 0.f; }
                bool anyFlux = false;   // wave-uniform
                float rk = 0.f;
                int nFoundLane = k;
                const bool need = inP && S.nPhotons > 0u && useLiiAny;
                bool done = !need;      // lanes whose L_ii is settled (served by the plan, or nothing to look up)
                if (__ballot(need)) {
                    // The bucket plan, up to three times.  Every attempt serves the lanes whose wanted radius^2 is within 3x of the
                    // smallest one (a common bucket sized for an outlier would overflow for everybody); a lane that found too few
                    // photons asks again with a radius scaled from the count it saw, a bucket overflow halves everybody's radius.
                    // What is still open afterwards is handed to li_fixup_kernel.
                    float guessBase = lastRk;
                    if (j < PREV_N) guessBase = fmaxf(guessBase, prevRk[j]);
                    {   // cold: what the other lanes found at their previous step, else the map-wide estimate
                        const float nbr = wave_max(lastRk);
                        if (!(guessBase > 0.f)) guessBase = nbr > 0.f ? nbr : S.rkEstimate;
                    }
                    float Twant = need ? ((!fixedR && guessBase * A.grpGuess < S.maxDistSq) ? guessBase * A.grpGuess : S.maxDistSq) : 0.f;
                    int lastFail = 0;   // stats: 1 = bucket overflow, 2 = too few photons inside the radius, 3 = other
                    for (int attempt = 0; attempt < 3; ++attempt) {
                        bool needP = need && !done && Twant > 0.f;
                        if (!__ballot(needP)) break;
                        float Tl = Twant;
                        if (attempt < 2) {
                            const float Tmin = -wave_max(needP ? -Tl : -3.0e38f);
                            needP = needP && !(Tl > 3.f * Tmin);   // the others wait for the next attempt with their Twant
                        }
                        const float gbl = Tl;
                        bool fullR = !(Tl < S.maxDistSq);
                        float T = needP ? Tl : 0.f;
                        T = wave_max(T);
                        // centre and spread of the query points
                        const float big = 3.0e38f;
                        float lx = needP ? p.x : big, ly = needP ? p.y : big, lz = needP ? p.z : big;
                        float hx = needP ? p.x : -big, hy = needP ? p.y : -big, hz = needP ? p.z : -big;
                        lx = -wave_max(-lx); ly = -wave_max(-ly); lz = -wave_max(-lz);
                        hx = wave_max(hx); hy = wave_max(hy); hz = wave_max(hz);
                        V3 c = v3(0.5f * (lx + hx), 0.5f * (ly + hy), 0.5f * (lz + hz));
                        float rho = needP ? len(p - c) : 0.f;
                        rho = wave_max(rho);
                        if (attempt < 2 && rho > 0.3f * sqrtf(T)) {
                            // the query points are spread over a good part of the search radius (rays that march with different step
                            // lengths): serve the half below the centre along the widest axis now, the rest in the next attempt
                            const float ex = hx - lx, ey = hy - ly, ez = hz - lz;
                            const float side = (ex >= ey && ex >= ez) ? p.x - c.x : (ey >= ez ? p.y - c.y : p.z - c.z);
                            needP = needP && !(side > 0.f);
                            T = wave_max(needP ? Tl : 0.f);
                            lx = needP ? p.x : big; ly = needP ? p.y : big; lz = needP ? p.z : big;
                            hx = needP ? p.x : -big; hy = needP ? p.y : -big; hz = needP ? p.z : -big;
                            lx = -wave_max(-lx); ly = -wave_max(-ly); lz = -wave_max(-lz);
                            hx = wave_max(hx); hy = wave_max(hy); hz = wave_max(hz);
                            c = v3(0.5f * (lx + hx), 0.5f * (ly + hy), 0.5f * (lz + hz));
                            rho = wave_max(needP ? len(p - c) : 0.f);
                        }
                        const float Rs = (sqrtf(T) + rho) * 1.0001f + 1e-6f;   // superset by the triangle inequality, with rounding slack
                        unsigned long long tst = 0, ts0 = STATS ? stamp() : 0ull;
                        __syncthreads();   // the U region changes hands: mini lists -> paint list
                        const int Mb = stage_bucket(S, G, L.pos, c, Rs, lane, tst);
                        if (STATS) { wc.tested += (unsigned long long)max(Mb, 0); wc.cySearch += stamp() - ts0; }
                        if (STATS) wc.diag5 += 1;
                        if (Mb < 0) {   // bucket overflow
                            if (STATS && !GRP_PHASE_DIAG) wc.diag1 += __popcll(__ballot(needP));
                            if (needP) { Twant = fixedR ? 0.f : 0.4f * Tl; lastFail = 1; }   // fixed radius: nothing smaller to try
                            continue;
                        }
                        // the bucket covers more than this lane asked for when other lanes asked for more: take it (up to
                        // GRP_WIDEN x) -- a wider search ball costs this lane nothing and spares it a failed guess
                        if (needP) {
                            const float cover = (sqrtf(T) + rho) - len(p - c);
                            Tl = fminf(S.maxDistSq, fmaxf(Tl, fminf(cover * cover, gbl * GRP_WIDEN)));
                            fullR = !(Tl < S.maxDistSq);
                        }
                        const unsigned long long tp1 = STATS ? stamp() : 0ull;
                        const nf2 pxy = {p.x, p.y}, pz_ = {p.z, p.z};
                        // ---- pass 1: histogram over [0, Tl)
                        const float scale = needP ? (float)GRP_BINS / Tl : 0.f;   // lanes without a lookup put everything in the overflow word
                        const bool exact = __ballot(needP && fullR) != 0ull;
#pragma unroll
                        for (int wd = 0; wd <= GRP_BINS / 8; ++wd) histLane[wd * LANES] = 0u;
                        if (exact) grp_pass1<true>(bX, bY, bZ, Mb, pxy, pz_, scale, Tl, histLane);
                        else grp_pass1<false>(bX, bY, bZ, Mb, pxy, pz_, scale, Tl, histLane);
                        const unsigned long long tpa = (STATS && GRP_PHASE_DIAG == 1) ? stamp() : 0ull;
                        if (STATS && GRP_PHASE_DIAG == 1) wc.diag1 += tpa - tp1;
                        // ---- prefix scan: word of the k-th, then its nibble
                        int cum = 0, wsel = -1, cumWord = 0;
                        uint32_t selBits = 0u;
#pragma unroll
                        for (int wd = 0; wd < GRP_BINS / 8; ++wd) {
                            const uint32_t w = histLane[wd * LANES];
                            const int t = (int)nib_sum(w);
                            const bool here = wsel < 0 && cum + t >= k;
                            wsel = here ? wd : wsel;
                            selBits = here ? w : selBits;
                            cumWord = here ? cum : cumWord;
                            cum += t;
                        }
                        const int inRange = ((Mb + 3) & ~3) - (int)histLane[(GRP_BINS / 8) * LANES];
                        int bstarI = -1, cumBelow = 0, binCount = 0;
                        {
                            int c2 = cumWord;
#pragma unroll
                            for (int s = 0; s < 8; ++s) {
                                const int cb = (int)((selBits >> (4 * s)) & 15u);
                                const bool here = bstarI < 0 && c2 + cb >= k;
                                bstarI = here ? 8 * wsel + s : bstarI;
                                cumBelow = here ? c2 : cumBelow;
                                binCount = here ? cb : binCount;
                                c2 += cb;
                            }
                        }
                        // a 4-bit counter that wrapped makes the counters' total fall short of the in-range count: not trusted
                        const bool sane = needP && cum == inRange;
                        bool ok = sane && wsel >= 0 && bstarI >= 0 && binCount <= GRP_MINI;
                        // the full radius holds fewer than k photons: all of them count, r^2 = the farthest (photonvolume.cpp:76-105)
                        const bool shortSet = sane && fullR && inRange < k;
                        const bool tooFew = needP && !fullR && cum < k;
                        if (STATS && !GRP_PHASE_DIAG && attempt == 0) wc.diag0 += __popcll(__ballot(needP && !ok && !shortSet));
                        if (needP) {   // what to ask for next time (0 = nothing: crowded bin or wrapped counter, the exact lookup takes it)
                            Twant = 0.f;
                            lastFail = tooFew ? 2 : 3;
                            // crowded bin (more values than the mini list ranks) or a wrapped counter: a radius that ends just behind
                            // the k-th's bin still holds k photons and spreads them over finer bins
                            if (!tooFew && !ok && !shortSet) Twant = (sane && bstarI >= 0) ? Tl * ((float)(bstarI + 1) * (1.f / GRP_BINS)) * 1.001f : 0.6f * Tl;
                            if (tooFew) {   // the count seen inside Tl gives the local density: k photons need ~ (k / count)^(2/3) x Tl
                                const float grow = cum > 0 ? 1.35f * __builtin_amdgcn_exp2f(0.6666667f * __builtin_amdgcn_logf((float)k / (float)cum)) : 1.0e9f;
                                Twant = fminf(S.maxDistSq, Tl * fmaxf(1.5f, grow));
                            }
                        }
                        const bool planLane = ok || shortSet;
                        const unsigned long long tpb = (STATS && GRP_PHASE_DIAG == 1) ? stamp() : 0ull;
                        if (STATS && GRP_PHASE_DIAG == 1) wc.diag0 += tpb - tpa;
                        if (__ballot(planLane)) {
                            // ---- pass 2: member bits
                            uint32_t mem[GRP_NW], le[GRP_NW];
                            float dmax = 0.f;
                            const uint32_t bstar = shortSet ? (uint32_t)GRP_BINS : (ok ? (uint32_t)bstarI : 0u);
                            float th1, th2;
                            bool thOK = true;
                            if (exact) { th1 = grp_theta<true>(bstar, scale, Tl, &thOK); th2 = grp_theta<true>(bstar + 1u, scale, Tl, &thOK); }
                            else { th1 = grp_theta<false>(bstar, scale, Tl, &thOK); th2 = grp_theta<false>(bstar + 1u, scale, Tl, &thOK); }
                            ok = ok && thOK;
                            __syncthreads();
                            if (exact) grp_pass2<true>(bX, bY, bZ, Mb, pxy, pz_, th1, th2, mem, le, dmax);
                            else grp_pass2<false>(bX, bY, bZ, Mb, pxy, pz_, th1, th2, mem, le, dmax);
                            const unsigned long long tpc = (STATS && GRP_PHASE_DIAG == 1) ? stamp() : 0ull;
                            if (STATS && GRP_PHASE_DIAG == 1) wc.diag3 += tpc - tpb;
                            // ---- the photons of bin b*: their exact values into the lane's mini list (slot order)
                            int nb = 0;
#pragma unroll
                            for (int wd = 0; wd < GRP_NW; ++wd) {
                                if (wd * 32 >= Mb) continue;   // wave-uniform
                                uint32_t ib = ok ? (le[wd] & ~mem[wd]) : 0u;
                                while (__ballot(ib != 0u)) {
                                    const bool v = ib != 0u;
                                    const int slot = wd * 32 + (v ? __builtin_ctz(ib) : 0);
                                    const float dx = bX[slot] - p.x, dy = bY[slot] - p.y, dz = bZ[slot] - p.z;
                                    const float d2 = dx * dx + dy * dy + dz * dz;
                                    if (v && nb < GRP_MINI) L.miniD[nb * LANES + lane] = d2;
                                    nb += v ? 1 : 0;
                                    ib &= ib - 1u;
                                }
                            }
                            __syncthreads();
                            // sort (<= GRP_MINI values; absent slots are +inf): odd-even transposition network
                            float sv[GRP_MINI];
#pragma unroll
                            for (int t = 0; t < GRP_MINI; ++t) sv[t] = t < nb ? L.miniD[t * LANES + lane] : INFINITY;
#pragma unroll
                            for (int rnd = 0; rnd < GRP_MINI; ++rnd) {
#pragma unroll
                                for (int t = rnd & 1; t + 1 < GRP_MINI; t += 2) {
                                    const float lo = fminf(sv[t], sv[t + 1]), hi = fmaxf(sv[t], sv[t + 1]);
                                    sv[t] = lo; sv[t + 1] = hi;
                                }
                            }
                            const int m = k - cumBelow;   // 1-based rank of the k-th inside its bin
                            float rkC = sv[0];
#pragma unroll
                            for (int t = 1; t < GRP_MINI; ++t) rkC = (m == t + 1) ? sv[t] : rkC;
                            int nLessIn = 0;
#pragma unroll
                            for (int t = 0; t < GRP_MINI; ++t) nLessIn += (sv[t] < rkC) ? 1 : 0;
                            int quota = m - nLessIn;   // ties at rkC are taken in bucket order
                            ok = ok && nb == binCount && rkC < INFINITY;
                            // member bits of bin b*: d2 < r_k, or d2 == r_k while the tie quota lasts
                            {
                                int t2 = 0;
#pragma unroll
                                for (int wd = 0; wd < GRP_NW; ++wd) {
                                    if (wd * 32 >= Mb) continue;
                                    uint32_t ib = ok ? (le[wd] & ~mem[wd]) : 0u;
                                    while (__ballot(ib != 0u)) {
                                        const bool v = ib != 0u;
                                        const uint32_t bit = v ? (ib & (0u - ib)) : 0u;
                                        const float dv = L.miniD[min(t2, GRP_MINI - 1) * LANES + lane];
                                        const bool tie = v && dv == rkC && quota > 0;
                                        if (tie) --quota;
                                        if (v && (dv < rkC || tie)) mem[wd] |= bit;
                                        t2 += v ? 1 : 0;
                                        ib &= ib - 1u;
                                    }
                                }
                            }
                            if (shortSet) { ok = thOK; rkC = dmax; }
                            const unsigned long long tp3 = STATS ? stamp() : 0ull;
                            if (STATS) wc.cySelect += tp3 - tp1;
                            if (STATS && GRP_PHASE_DIAG == 1) wc.diag4 += tp3 - tpc;
                            // ---- flux on the matrix pipe.  A = alpha rows (lane l: bin l & 31 of the slot of its half-wave: one coalesced
                            // 128-B row per half-wave and load), B = member bits of ray l & 31 for that slot as 0.f / 1.f (the upper
                            // half-wave's copy of a ray's word comes from ONE v_permlane32_swap per 32 slots): D = A B + C adds a row to exactly
                            // the rays it is a member for.  Two slots per instruction, rays 0..31 into CA, rays 32..63 into CB.
                            const int half = lane >> 5, binL = lane & 31;
                            typedef const __attribute__((address_space(1))) float gfloat;   // global_load, not flat_load
                            gfloat *alphaF = (gfloat *)(S.alpha4);
                            if (GRP_CORE_MIN < GRP_CAP) {
                                const bool anyOk = __ballot(ok) != 0ull;
                                int nCore = 0;
                                uint32_t coreW[GRP_NW];
#pragma unroll
                                for (int wd = 0; wd < GRP_NW; ++wd) {
                                    coreW[wd] = 0u;
                                    if (wd * 32 >= Mb || !anyOk) continue;
                                    coreW[wd] = (uint32_t)__builtin_amdgcn_readfirstlane((int)~wave_or(ok ? ~mem[wd] : 0u));
                                    nCore += __builtin_popcount(coreW[wd]);
                                }
                                if (nCore >= GRP_CORE_MIN) {
                                    // slots that are members for EVERY served lane: their rows are summed once (two rows per load, lane = bin)
                                    // and the sum enters as one more "row" whose member bits are the served lanes
                                    uint32_t *clist = reinterpret_cast<uint32_t *>(L.miniD);
                                    int basePos = 0;
#pragma unroll
                                    for (int r2 = 0; r2 < GRP_NW / 2; ++r2) {
                                        if (r2 * 64 >= Mb) continue;
                                        const unsigned long long c64 = (unsigned long long)coreW[2 * r2] | ((unsigned long long)coreW[2 * r2 + 1] << 32);
                                        if ((c64 >> lane) & 1ull) clist[basePos + (int)lanes_below(c64, lane)] = __float_as_uint(bI[r2 * 64 + lane]);
                                        basePos += __popcll(c64);
                                    }
                                    __syncthreads();
                                    float cs = 0.f;
                                    for (int c0 = 0; c0 < nCore; c0 += 16) {   // up to 16 rows in flight per trip
                                        float v[8];
#pragma unroll
                                        for (int q2 = 0; q2 < 8; ++q2) {
                                            const int at = c0 + 2 * q2 + half;
                                            const bool on = at < nCore;
                                            const uint32_t idx = clist[on ? at : 0];
                                            const float w = alphaF[(size_t)idx * 32 + binL];
                                            v[q2] = on ? w : 0.f;
                                        }
                                        cs += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                                    }
                                    { auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(cs), __float_as_uint(cs), false, false); cs = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
                                    const uint32_t okW = ok ? 1u : 0u;
                                    auto ro = __builtin_amdgcn_permlane32_swap(okW, okW, false, false);
                                    const float fA = half ? 0.f : (float)ro[0], fB = half ? 0.f : (float)ro[1];
                                    CA = __builtin_amdgcn_mfma_f32_32x32x2f32(cs, fA, CA, 0, 0, 0);
                                    CB = __builtin_amdgcn_mfma_f32_32x32x2f32(cs, fB, CB, 0, 0, 0);
                                    anyFlux = true;
#pragma unroll
                                    for (int wd = 0; wd < GRP_NW; ++wd) mem[wd] &= ~coreW[wd];
                                    __syncthreads();
                                }
                            }
#pragma unroll
                            for (int wd = 0; wd < GRP_NW; ++wd) {
                                if (wd * 32 >= Mb) continue;
                                const uint32_t mine = ok ? mem[wd] : 0u;
                                auto sw = __builtin_amdgcn_permlane32_swap(mine, mine, false, false);
                                const uint32_t W1 = sw[0], W2 = sw[1];   // every lane: the word of ray (l & 31), of ray 32 + (l & 31)
                                uint32_t any = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wave_or(W1 | W2)));
                                if (!any) continue;
                                anyFlux = true;
                                const int idxW = (int)__float_as_uint(bI[wd * 32 + binL]);   // this word's photon indices, one per lane
                                while (any) {   // four slot pairs per trip: their rows are requested before the first is used
                                    int sel[4];
                                    float av[4];
                                    bool on[4], two[4];
#pragma unroll
                                    for (int t = 0; t < 4; ++t) {
                                        on[t] = any != 0u;
                                        const int b0 = on[t] ? __builtin_ctz(any) : 0;
                                        any &= any - 1u;
                                        two[t] = any != 0u;
                                        const int b1 = two[t] ? __builtin_ctz(any) : b0;
                                        any &= any - 1u;   // 0 & anything stays 0
                                        sel[t] = half ? b1 : b0;
                                        av[t] = 0.f;
                                        if (on[t]) {   // wave-uniform
                                            const uint32_t i0 = (uint32_t)__builtin_amdgcn_readlane(idxW, b0), i1 = (uint32_t)__builtin_amdgcn_readlane(idxW, b1);
                                            av[t] = alphaF[(size_t)(half ? i1 : i0) * 32 + binL];
                                        }
                                    }
#pragma unroll
                                    for (int t = 0; t < 4; ++t) {
                                        if (!on[t]) continue;   // wave-uniform
                                        const bool live = two[t] || !half;   // an odd slot out: the upper half-wave contributes nothing
                                        const float f1 = live ? (float)((W1 >> sel[t]) & 1u) : 0.f, f2 = live ? (float)((W2 >> sel[t]) & 1u) : 0.f;
                                        CA = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], f1, CA, 0, 0, 0);
                                        CB = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], f2, CB, 0, 0, 0);
                                    }
                                }
                            }
                            if (ok) { done = true; rk = rkC; if (shortSet) nFoundLane = inRange; }
                            if (STATS) { wc.kept += (unsigned long long)k * __popcll(__ballot(ok)); wc.cyFlux += stamp() - tp3; }
                        }
                    }   // attempt
                    // ---- lanes the plan did not serve: their term is added by li_fixup_kernel (exact lookup), nothing else waits for it
                    const uint64_t todo = __ballot(need && !done);
                    if (todo) {
                        // A.fixGroup: whole 64-slot runs, one per group-step (li_fixup_group_kernel shares a bucket per run);
                        // slots of lanes with nothing to hand over are marked empty
                        const bool padded = A.fixGroup != 0;
                        const int nfb = padded ? LANES : __popcll(todo);
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(A.deferCount, (uint32_t)nfb);
                        base = (uint32_t)lane_i((int)base, 0);
                        if (padded || (need && !done)) {
                            const uint32_t at = base + (padded ? (uint32_t)lane : lanes_below(todo, lane));
                            if (at < A.deferCap) {
                                DeferRec r;
                                r.ray = (need && !done) ? (uint32_t)ri : 0xffffffffu; r.px = p.x; r.py = p.y; r.pz = p.z; r.kRem = kRem; r.stepD = stepD;
                                // the radius^2 the next attempt would have asked for is the best guess there is (lphoton widens it by 1.3)
                                r.guess = fixedR ? 0.f : (Twant > 0.f ? Twant * (1.f / PVOL_GUESS_SCALE) : (guessBase > 0.f ? guessBase : 0.f)); r.dens = dens;
                                A.defer[at] = r;
                            } else {
                                atomicOr(A.needSeq, 1u);   // list full: the batch is redone by the sequential kernel, never dropped
                            }
                        }
                        if (STATS && !GRP_PHASE_DIAG) { wc.retries += nfb; wc.diag3 += __popcll(__ballot(need && !done && lastFail == 1)); wc.diag4 += __popcll(__ballot(need && !done && lastFail == 2)); }
                    }
                    // a k-th distance exists only for full sets; a short set says "search the full radius here"
                    const float rkGuess = (need && done) ? (nFoundLane >= k ? rk : S.maxDistSq) : 0.f;
                    {   // mean k-th distance^2 of the group at this step -> guess of the next group
                        const float sr = wave_sum(rkGuess), sn = (float)__popcll(__ballot(rkGuess > 0.f));
                        if (j < PREV_N && lane == 0 && sn > 0.f) prevRk[j] = sr / sn;
                    }
                    if (need) lastRk = rkGuess;
                }
                // ---- this step's term of  Lv = sum_j w_j exp(-sigma_t R_j)  (photonvolume.cpp:150-218), 30 bins per lane:
                //   w_j  = sigma_a Le step + sigma_s step (L_d + albedo L_ii)
                //   L_ii = Sum(alpha) * phase / (4/3 pi r^3 sigma_s)                        (:99-105; acc holds the raw sum)
                //   L_d  = I * [falloff / d^2] * exp(-sigma_t * exit) * phase * nLights      (:178-203)
                const unsigned long long tq2 = (STATS && GRP_PHASE_DIAG == 2) ? stamp() : 0ull;
                if (STATS && GRP_PHASE_DIAG == 2) wc.diag1 += tq2 - tq1;
                float liiScale = 0.f;   // raw sum -> L_ii * sigma_s
                if (need && done) {
                    const float dV = rk * sqrtf(rk);
                    if (dV != 0.f && nFoundLane >= 10) liiScale = wIso * __builtin_amdgcn_rcpf(float(4.0 / 3.0 * (double)K_PI * (double)dV));
                    if (GRID) liiScale *= __builtin_amdgcn_rcpf(dens);   // LPhoton divides by sigma_s(p) = sigma_s x density (photonvolume.cpp:103-105)
                }
                const float stepE = GRID ? stepD * dens : stepD;   // emission: sigma_a(p) x Lve(p), both carry the density (photonvolume.cpp:215)
                const bool useLii = inP && (ySa1 != 0.0 || ySs1 != 0.0);
                float ldScale = 0.f;
                if (lit) ldScale = (distant ? 1.f : fallReg * __builtin_amdgcn_rcpf(d2Reg)) * ph * float(nLights);
                const float kExit = -1.442695041f * exitLen;   // exp(-x) = exp2(-x log2 e)
                if (anyFlux) {   // every lane takes its own ray's column: CA <- bins 8g + c, CB <- bins 8g + 4 + c
#pragma unroll
                    for (int b = 0; b < 16; ++b) {
                        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(CA[b]), __float_as_uint(CB[b]), false, false);
                        CA[b] = __uint_as_float(r[0]); CB[b] = __uint_as_float(r[1]);
                    }
                }
                if (act) {
                    // per bin:  t = exp2(sT kRem) [ (sigA Le) stepE + (sigS stepD) (Ld + (albedo / sigS) acc liiScale) ],  fused multiply-adds
                    // (values only: no decision hangs on them)
                    const float liiU = useLii ? liiScale : 0.f;
#pragma unroll
                    for (int qq = 0; qq < 8; ++qq) {
                        __builtin_amdgcn_sched_barrier(0);   // keep the constants of one bin quartet live at a time
                        const f4 t4 = cT[qq], s4 = cS[qq], al4 = cAL[qq], ar4 = cAR[qq];
                        const f4 i4 = REPLAY ? *reinterpret_cast<const f4 *>(L.lint + ln * 32 + 4 * qq) : cI[qq];
                        const f4 x4 = cX[qq], y4 = cY[qq], z4 = cZ[qq];
                        const float tv[4] = {t4.x, t4.y, t4.z, t4.w}, sv[4] = {s4.x, s4.y, s4.z, s4.w}, alv[4] = {al4.x, al4.y, al4.z, al4.w};
                        const float arv[4] = {ar4.x, ar4.y, ar4.z, ar4.w}, iv[4] = {i4.x, i4.y, i4.z, i4.w};
                        const float xv[4] = {x4.x, x4.y, x4.z, x4.w}, yv[4] = {y4.x, y4.y, y4.z, y4.w}, zv[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) {
                            const int b = 4 * qq + cc;
                            if (b >= 30) continue;
                            const float Pj = __builtin_amdgcn_exp2f(tv[cc] * kRem);
                            const float Ld = iv[cc] * (ldScale * __builtin_amdgcn_exp2f(tv[cc] * kExit));
                            const float accB = (qq & 1) ? CB[4 * (qq >> 1) + cc] : CA[4 * (qq >> 1) + cc];
                            const float Li = __builtin_fmaf(arv[cc], accB * liiU, Ld);
                            const float w = __builtin_fmaf(sv[cc] * stepD, Li, alv[cc] * stepE);
                            const float t = Pj * w;
                            if (SPECTRAL) Lv[b] += t;
                            else { accX = __builtin_fmaf(xv[cc], t, accX); accY = __builtin_fmaf(yv[cc], t, accY); accZ = __builtin_fmaf(zv[cc], t, accZ); }
                        }
                    }
                    pPrev = p;
                    inPrev = inP;
                    lenLast = lenStep;
                }
                if (STATS && GRP_PHASE_DIAG == 2) wc.diag3 += stamp() - tq2;
            }
            // ---- outputs
            if (have) {
                if (bad) {
                    atomicOr(A.needSeq, 1u);
                } else {
                    const uint32_t draws = hit ? 4u + 7u * (uint32_t)nS + uCount : 0u;
                    const float kLast = -1.442695041f * lenLast;
                    if (SPECTRAL) {
                        float *op = A.out + ri * 60;
#pragma unroll
                        for (int b = 0; b < 30; ++b) {
                            op[b] = Lv[b];
                            op[30 + b] = hit ? __builtin_amdgcn_exp2f((L.cst[b] + L.cst[32 + b]) * kLast) : 1.f;
                        }
                    } else {
                        float ty = 0.f;
#pragma unroll
                        for (int qq = 0; qq < 8; ++qq) {
                            const f4 a4 = cA[qq], s4 = cS[qq], y4 = cY[qq];
                            const float av[4] = {a4.x, a4.y, a4.z, a4.w}, sv[4] = {s4.x, s4.y, s4.z, s4.w}, yv[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
                            for (int cc = 0; cc < 4; ++cc) {
                                if (4 * qq + cc >= 30) continue;
                                const float trb = hit ? __builtin_amdgcn_exp2f((av[cc] + sv[cc]) * kLast) : 1.f;
                                ty += yv[cc] * trb;
                            }
                        }
                        const float scale = float(700 - 400) / float(106.856895f * 30);
                        *reinterpret_cast<f4 *>(A.out + ri * 4) = make_float4(accX * scale, accY * scale, accZ * scale, ty * 300.f / (106.856895f * 30));
                    }
                    if (!REPLAY && A.draws) A.draws[ri] = draws;
                    if (A.tauOut) A.tauOut[ri] = hit ? lenLast : 0.f;   // T = exp(-sigma_t * this): what the surface term is attenuated by
                }
            }
            if (!REPLAY) {   // stream positions: one atomic per group when all its rays belong to one stream (the usual case)
                const bool cntd = have && !bad;
                const uint32_t sidx = cntd ? stream_of(A.streams, A.nStreams, (uint32_t)ri) : 0u;
                const uint32_t mine = cntd ? (hit ? 4u + 7u * (uint32_t)nS + uCount : 0u) + pr.rng_skip : 0u;
                const uint64_t cm = __ballot(cntd);
                if (cm) {
                    const uint32_t s0 = (uint32_t)lane_i((int)sidx, __ffsll((unsigned long long)cm) - 1);
                    if (!__ballot(cntd && sidx != s0)) {
                        unsigned long long tot = mine;
                        for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
                        if (lane == 0) atomicAdd((unsigned long long *)&A.streams[s0].end_draw, tot);
                    } else if (cntd) {
                        atomicAdd((unsigned long long *)&A.streams[sidx].end_draw, (unsigned long long)mine);
                    }
                }
            }
        }
    }
    flush_counters<STATS>(A.counters, wc, tk0, lane);
}

#include "pvol_fixgrp_dev.h"   // li_fixup_group_kernel: the hand-over list of nused beyond the bucket plan, 64 lookups per staged bucket

// The lookups li_group_kernel handed over: one wave per entry runs the exact wave-cooperative lphoton() and adds the
// entry's term  exp(-sigma_t R_j) sigma_s step albedo L_ii  to the ray's output (atomic: a ray can have several entries).
template <bool STATS, bool SPECTRAL, int NREG>
__global__ __launch_bounds__(LANES, (NREG > 4 ? PVOL_WPE_BIG : PVOL_WPE)) void li_fixup_kernel(LiArgs A) {
    extern __shared__ __align__(16) unsigned char lds[];
    const DevScene &S = *A.scene;
    const int lane = threadIdx.x;
    if (*A.needSeq != 0u) return;   // the whole batch is redone sequentially
    const uint32_t n = min(*A.deferCount, A.deferCap);
    Gather G;
    G.cap = S.candCap;
    G.cd = reinterpret_cast<float *>(lds);
    G.ci = reinterpret_cast<uint32_t *>(lds + (size_t)G.cap * 4);
    G.paint = reinterpret_cast<uint32_t *>(lds + (size_t)G.cap * 8);
    const int q = lane & 7;
    const f4 sigA4 = ld4(S.sigA, q), sigS4 = ld4(S.sigS, q);
    const f4 sigT4 = sigA4 + sigS4;
    const f4 albedo4 = clean4(fdiv4(sigS4, sigT4), q);
    const f4 X4 = ld4(S.cieX, q), Y4 = ld4(S.cieY, q), Z4 = ld4(S.cieZ, q);
    WaveCounters wc = {};
    unsigned long long tk0 = STATS ? stamp() : 0ull;
    // Runs of 64 consecutive entries per wave: the list is filled group by group, so neighbours in the list are neighbours in
    // space (the 64 rays of a group at one march step) and the k-th distance^2 just found is the first radius of the next lookup.
    // An entry without a guess of its own (nused beyond the bucket plan hands everything over unseen) would otherwise scan the
    // whole maxdist ball -- tens of thousands of photons inside pinkfloyd's beams.  A wrong first radius costs a retry, never
    // the result: lphoton() widens it until the k nearest are inside.
    // With A.fixGroup the list comes in padded 64-slot runs and li_fixup_group_kernel serves the compact ones (fxg_compact).
    float carry = 0.f;
    const GridView gv = volume_grid(S);
    // Eight CONSECUTIVE runs per wave at a time: the list is filled group-step by group-step, so neighbouring runs are the same rays one march
    // step apart -- their photon sets overlap almost entirely, and dealt round-robin over the waves (as until round 3) every XCD fetched them
    // into its own L2 (hit rate 9 %, the kernel HBM-bound at 3.2 TB/s on C3)
    const uint32_t nRuns = (n + 63u) / 64u;
    for (uint32_t runI = blockIdx.x * 8u; runI < nRuns; runI = ((runI & 7u) == 7u) ? runI + 1u + (gridDim.x - 1u) * 8u : runI + 1u) {
    const uint32_t e0 = runI * 64u;
    DeferRec mine;
    mine.ray = 0xffffffffu; mine.px = mine.py = mine.pz = 0.f; mine.kRem = 0.f; mine.stepD = 0.f; mine.guess = 0.f; mine.dens = 1.f;
    if (e0 + (uint32_t)lane < n) mine = A.defer[e0 + lane];
    unsigned long long todo = __ballot(mine.ray != 0xffffffffu);
    if (A.fixGroup) {
        float Tprobe;
        if (fxg_compact(gv, S, mine.ray != 0xffffffffu, v3(mine.px, mine.py, mine.pz), lane, &Tprobe, A.fxgAim > 0.f ? A.fxgAim : 1.4f)) continue;
        if (Tprobe > 0.f) carry = Tprobe * (1.f / PVOL_GUESS_SCALE);
    }
    while (todo) {
        const int j = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        DeferRec r;
        r.ray = (uint32_t)lane_i((int)mine.ray, j);
        r.px = lane_f(mine.px, j); r.py = lane_f(mine.py, j); r.pz = lane_f(mine.pz, j);
        r.kRem = lane_f(mine.kRem, j); r.stepD = lane_f(mine.stepD, j); r.guess = lane_f(mine.guess, j); r.dens = lane_f(mine.dens, j);
        float rk;
        const float first = r.guess > 0.f ? r.guess : (carry > 0.f ? carry : S.rkEstimate);
        const f4 Lii = lphoton<STATS, NREG, true>(S, G, v3(0.f, 0.f, 1.f), v3(r.px, r.py, r.pz), sigS4 * r.dens, lane, wc, first, &rk);   // g == 0: the direction is not read
        carry = rk > 0.f ? rk : S.maxDistSq;   // fewer than nused within maxdist: sparse here, the full ball is the cheap radius
        const f4 kk = sigT4 * r.kRem;
        f4 c = make_float4(__builtin_amdgcn_exp2f(kk.x), __builtin_amdgcn_exp2f(kk.y), __builtin_amdgcn_exp2f(kk.z), __builtin_amdgcn_exp2f(kk.w));
        c = clean4(c * (sigS4 * (albedo4 * Lii) * r.stepD), q);
        if (SPECTRAL) {
            if (lane < 8) {
                float *op = A.out + (size_t)r.ray * 60 + 4 * q;
                const float cv[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) if (4 * q + cc < 30 && cv[cc] != 0.f) atomicAdd(op + cc, cv[cc]);
            }
        } else {
            const float scale = float(700 - 400) / float(106.856895f * 30);
            const float x = group8_sum(X4.x * c.x + X4.y * c.y + X4.z * c.z + X4.w * c.w) * scale;
            const float y = group8_sum(Y4.x * c.x + Y4.y * c.y + Y4.z * c.z + Y4.w * c.w) * scale;
            const float z = group8_sum(Z4.x * c.x + Z4.y * c.y + Z4.z * c.z + Z4.w * c.w) * scale;
            if (lane == 0) {
                float *op = A.out + (size_t)r.ray * 4;
                atomicAdd(op, x); atomicAdd(op + 1, y); atomicAdd(op + 2, z);
            }
        }
    }
    }
    wc.rays = 0; wc.steps = 0;   // the march steps were counted by li_group_kernel
    if (STATS) { wc.diag2 = stamp() - tk0; wc.cySearch = wc.cySelect = wc.cyFlux = 0; tk0 = stamp(); }   // this kernel's cycles are reported on their own
    flush_counters<STATS>(A.counters, wc, tk0, lane);
}

extern "C" size_t pvol_group_lds_bytes(int candCap) {
    (void)candCap;
    return (size_t)PREV_N * 4 + GRP_CH * 2 + (size_t)GRP_PITCH * 16 + GRP_U_BYTES + 12 * 32 * 4 + GRP_TRI_ROWS * 64 + PVOL_MAX_LIGHTS * 32 * 4;
}

// replay: 0 = ray-parallel scenes (li_par_kernel's conditions), 1 = records of the RNG pre-pass, homogeneous, 2 = records, VolumeGrid
extern "C" hipError_t pvol_launch_li_group(const LiArgs *args, size_t ldsBytes, int candCap, bool stats, uint32_t nWaves, uint32_t nFixWaves,
                                           int replay, hipStream_t stream) {
    if (!replay) hipLaunchKernelGGL(stream_begin_kernel, dim3((args->nStreams + 255) / 256), dim3(256), 0, stream, args->streams, args->nStreams);
    dim3 grid(nWaves), block(LANES), fgrid(nFixWaves);
    const bool spectral = args->outputKind == PVOL_OUT_SPECTRAL;
    const size_t fixLds = (size_t)candCap * 8 + PAINT_CAP * 4;
    const bool big = candCap > 4 * LANES;   // select_k registers of the exact lookup: 4 cover nused <= 64, 12 cover nused <= 576
    const bool fixGroup = args->fixGroup != 0 && big && !stats;
    const size_t fixGrpLds = pvol_fixgrp_lds_bytes(candCap);
#define GRP_LAUNCH(ST, SP, RP, GR) hipLaunchKernelGGL((li_group_kernel<ST, SP, RP, GR>), grid, block, ldsBytes, stream, *args)
#define FIX_LAUNCH(ST, SP) do { if (fixGroup) hipLaunchKernelGGL((li_fixup_group_kernel<SP>), fgrid, block, fixGrpLds, stream, *args);   /* compact runs */ \
                                if (big) hipLaunchKernelGGL((li_fixup_kernel<ST, SP, 12>), fgrid, block, fixLds, stream, *args); \
                                else hipLaunchKernelGGL((li_fixup_kernel<ST, SP, 4>), fgrid, block, fixLds, stream, *args); } while (0)
    if (replay == 0) {
        if (stats) { if (spectral) { GRP_LAUNCH(true, true, false, false); FIX_LAUNCH(true, true); } else { GRP_LAUNCH(true, false, false, false); FIX_LAUNCH(true, false); } }
        else { if (spectral) { GRP_LAUNCH(false, true, false, false); FIX_LAUNCH(false, true); } else { GRP_LAUNCH(false, false, false, false); FIX_LAUNCH(false, false); } }
    } else if (replay == 1) {
        if (spectral) { GRP_LAUNCH(false, true, true, false); FIX_LAUNCH(false, true); } else { GRP_LAUNCH(false, false, true, false); FIX_LAUNCH(false, false); }
    } else {
        if (spectral) { GRP_LAUNCH(false, true, true, true); FIX_LAUNCH(false, true); } else { GRP_LAUNCH(false, false, true, true); FIX_LAUNCH(false, false); }
    }
#undef GRP_LAUNCH
#undef FIX_LAUNCH
    return hipGetLastError();
}
