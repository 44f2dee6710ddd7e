// pvol_group_dev.h -- included by pvol_march.hip.  li_group_kernel: PhotonVolumeIntegrator::Li for 64 camera rays
// at a time, ONE RAY PER LANE, for homogeneous isotropic media under the conditions of li_par_kernel (no drawn
// value reaches the result).  The k-NN gathers of the 64 lanes at march step j are neighbours in space -- the
// rays of a chunk are first ordered by their scatter offset, so a group's query points lie within a few
// hundredths of a unit of each other -- and share one candidate search:
//   1. STAGE   the photons within sqrt(T) + rho of the group's centre (T = guessed radius^2, rho = spread of the
//              query points) are found cooperatively in the cell grid and parked in LDS ("LDS-staged bucket");
//   2. SELECT  every lane finds ITS exact k-th smallest DistanceSquared over the bucket: a histogram pass over
//              [T/4, T) (byte counters in LDS, one column per lane) locates the bin of the k-th, a second pass keeps
//              that bin's few values sorted in registers;
//   3. FLUX    one more pass adds the alpha row of every bucket photon (wave-uniform address: broadcast loads)
//              into the per-lane 30-bin accumulators of the lanes whose k-NN set contains it.
// All three passes evaluate kdtree.h:180's DistanceSquared per lane in the reference's operation order, so the
// k-NN sets are the reference's; only the order of the flux additions differs (as in lphoton).  A lane whose
// lookup does not fit the plan (guess too small or too large, crowded bin, bucket overflow) is redone by the
// wave-cooperative lphoton(), which is always exact.  Spectra live as 30 registers per lane.
#define GRP_CH 512    // rays per chunk (ordered by scatter_u, then cut into groups of 64)
#define GRP_CAP 512   // bucket capacity (photons)
#define GRP_BINS 64   // histogram bins over [T/4, T), 4-bit counters (eight per LDS word)
#define GRP_PITCH (GRP_CAP + 4)   // floats per bucket component (x | y | z | photon index), padded for the 4-wide passes
#define GRP_WIDEN 1.7f   // a lane's search radius^2 may grow to this multiple of its guess where the bucket covers it
#define GRP_MINI 8    // values of the k-th's bin a lane can sort
#define GRP_TRI_ROWS 8   // scenes with a distant light and at most this many triangles test shadow rays against precomputed rows
#define GRP_WPE 2

struct GroupLds {
    float *pos;             // bucket, SoA: x[GRP_PITCH] | y | z | photon index bits
    float *mini;            // [GRP_MINI][64] values of the k-th's bin, one column per lane
    uint32_t *hist;         // [GRP_BINS / 8][64] packed 4-bit counters, one column per lane
    float *ubuf;            // GRP_CH scatter offsets
    unsigned short *order;  // GRP_CH: chunk-local ray index by rank of scatter offset
    float *cst;             // 9 x 32 floats: sigA, sigS, le, albedo, light-0 intensity, 1/sigS, CIE X, Y, Z weights
    unsigned short *clist;  // GRP_CAP bucket slots whose photon belongs to some lane's k-NN set
    float *trows;           // GRP_TRI_ROWS x 16 floats: per-triangle shadow-ray precomputation for a distant light
};

// Photons within Rs of c -> LDS bucket.  Returns the count, or -1 if the bucket would overflow.
__device__ int stage_bucket(const DevScene &S, Gather &G, float *bucket, V3 c, float Rs, int lane, unsigned long long &tested) {
    float *bX = bucket, *bY = bucket + GRP_PITCH, *bZ = bucket + 2 * GRP_PITCH, *bI = bucket + 3 * GRP_PITCH;
    const float cell = S.cellSize, inv = S.invCell;
    const float eps = cell * 1e-4f;
    const float T = Rs * Rs;
    const int cy = (int)floorf((c.y - S.gridLo[1]) * inv), cz = (int)floorf((c.z - S.gridLo[2]) * inv);
    const int Rt = (int)ceilf(Rs * inv + 1e-3f);
    const int side = 2 * Rt + 1, nrows = side * side;
    int count = 0;
    for (int rb = 0; rb < nrows; rb += LANES) {
        int r = rb + lane;
        int iy = r / side;
        int dy = iy - Rt, dz = (r - iy * side) - Rt;
        int y = cy + dy, z = cz + dz;
        bool rowOn = r < nrows && y >= 0 && y < S.gdim[1] && z >= 0 && z < S.gdim[2];
        float ylo = S.gridLo[1] + y * cell, zlo = S.gridLo[2] + z * cell;
        float ddy = fmaxf(0.f, fmaxf(ylo - c.y, c.y - (ylo + cell)) - eps);
        float ddz = fmaxf(0.f, fmaxf(zlo - c.z, c.z - (zlo + cell)) - eps);
        float rd2 = ddy * ddy + ddz * ddz;
        rowOn = rowOn && rd2 < T;
        float hw = sqrtf(fmaxf(0.f, T - rd2)) + eps;
        int x0 = (int)floorf((c.x - hw - S.gridLo[0]) * inv), x1 = (int)floorf((c.x + hw - S.gridLo[0]) * inv);
        x0 = max(x0, 0);
        x1 = min(x1, S.gdim[0] - 1);
        rowOn = rowOn && x0 <= x1;
        uint32_t start = 0u, rlen = 0u;
        if (rowOn) {
            size_t base = ((size_t)z * S.gdim[1] + y) * S.gdim[0];
            start = S.cellStart[base + x0];
            rlen = S.cellStart[base + x1 + 1] - start;
        }
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
                        if (count + add > GRP_CAP) return -1;
                        if (acc) {
                            const int at = count + (int)lanes_below(m, lane);
                            bX[at] = P[k].x; bY[at] = P[k].y; bZ[at] = P[k].z; bI[at] = __uint_as_float(I[k]);
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
    if (lane < 4) { bX[count + lane] = 3.0e18f; bY[count + lane] = 3.0e18f; bZ[count + lane] = 3.0e18f; bI[count + lane] = 0.f; }   // pad to a multiple of four: never inside any radius
    __syncthreads();
    return count;
}

__device__ __forceinline__ float dist2_ref(f4 P, V3 pt) {   // DistanceSquared(photon.p, p), kdtree.h:180
    float dx = P.x - pt.x, dy = P.y - pt.y, dz = P.z - pt.z;
    return dx * dx + dy * dy + dz * dz;
}

template <bool STATS, int NREG>
__global__ __launch_bounds__(LANES, GRP_WPE) void li_group_kernel(LiArgs A) {
    extern __shared__ __align__(16) unsigned char lds[];
    const DevScene &S = *A.scene;
    const int lane = threadIdx.x;
    MarchLds M;
    // LDS plan (19.9 KB with candCap 256: eight waves per CU): prevRk | paint | bucket | mini | hist | cst | clist.
    // The fallback lookup's candidate arrays alias the bucket (dead by then), the chunk-ordering scratch aliases mini + hist
    // (used only between chunks; the groups' ray indices are taken into registers first).
    M.G.cap = S.candCap;
    M.lightNum = 0;
    M.prevRk = reinterpret_cast<float *>(lds);
    M.G.paint = reinterpret_cast<uint32_t *>(M.prevRk + PREV_N);
    GroupLds L;
    L.pos = reinterpret_cast<float *>(M.G.paint + PAINT_CAP);
    M.G.cd = L.pos;                                                  // candCap * 8 <= 4 * GRP_PITCH * 4 (checked on the host)
    M.G.ci = reinterpret_cast<uint32_t *>(L.pos + M.G.cap);
    L.mini = L.pos + 4 * GRP_PITCH;
    L.hist = reinterpret_cast<uint32_t *>(L.mini + (GRP_MINI + 1) * LANES);
    L.ubuf = L.mini;                                                 // GRP_CH floats
    L.order = reinterpret_cast<unsigned short *>(L.ubuf + GRP_CH);   // GRP_CH shorts; both fit in mini + hist
    L.cst = reinterpret_cast<float *>(L.hist + (GRP_BINS / 8) * LANES);
    L.clist = reinterpret_cast<unsigned short *>(L.cst + 9 * 32);
    L.trows = reinterpret_cast<float *>(lds + ((PREV_N * 4 + PAINT_CAP * 4 + (GRP_CAP + 4) * 16 + (GRP_MINI + 1) * LANES * 4 + (GRP_BINS / 8) * LANES * 4 + 9 * 32 * 4 + (GRP_CAP + 4) * 2 + 15) & ~15));
    for (int i = lane; i < PREV_N; i += LANES) M.prevRk[i] = 0.f;
    for (int i = lane; i < GRP_CAP + 4; i += LANES) L.clist[i] = 0;   // entries are read up to three past the list: keep them valid slots
    const int q = lane & 7;
    const f4 sigA4 = ld4(S.sigA, q), sigS4 = ld4(S.sigS, q);
    const f4 sigT4 = sigA4 + sigS4;
    const f4 Y4 = ld4(S.cieY, q);
    const int nLights = S.nLights;
    const float ySa1 = spec_y(sigA4, Y4), ySs1 = spec_y(sigS4, Y4);
    const bool blackS1 = spec_is_black(sigS4);
    const bool lightBlack = nLights > 0 ? spec_is_black(ld4(S.lights[0].intensity, q)) : true;
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
    }
    __syncthreads();
    const f4 *cA = reinterpret_cast<const f4 *>(L.cst), *cS = cA + 8, *cLe = cA + 16, *cAl = cA + 24, *cI = cA + 32, *cRs = cA + 40;
    const f4 *cX = cA + 48, *cY = cA + 56, *cZ = cA + 64;
    const bool rowsOK = nLights > 0 && S.lights[0].kind == PVOL_LIGHT_DISTANT && S.nTris <= GRP_TRI_ROWS;
    if (rowsOK) tri_rows_prepare(S, v3(S.lights[0].dir[0], S.lights[0].dir[1], S.lights[0].dir[2]), L.trows, lane);
    const int k = S.nUsed;
    const float wIso = 1.f / (4.f * K_PI);
    Rng rngNone;
    rngNone.mt = 0; rngNone.mti = 0; rngNone.draws = 0;
    WaveCounters wc = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    unsigned long long tk0 = STATS ? stamp() : 0ull;
    for (;;) {
        uint32_t chunk = 0;
        if (lane == 0) chunk = atomicAdd(A.chunkCounter, 1u);
        chunk = (uint32_t)lane_i((int)chunk, 0);
        const unsigned long long r0 = (unsigned long long)chunk * GRP_CH;
        if (r0 >= A.nRays) break;
        const int nIn = (int)min((unsigned long long)GRP_CH, (unsigned long long)A.nRays - r0);
        // ---- order the chunk's rays by scatter offset: lanes of a group then march in near lock-step positions
        __syncthreads();
        for (int i = lane; i < GRP_CH; i += LANES) L.ubuf[i] = i < nIn ? A.rays[r0 + i].scatter_u : 3.f;
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
        unsigned short ordReg[GRP_CH / LANES];
#pragma unroll
        for (int g = 0; g < GRP_CH / LANES; ++g) ordReg[g] = (g * LANES + lane < nIn) ? L.order[g * LANES + lane] : (unsigned short)0;
        __syncthreads();
#pragma unroll 1
        for (int g0 = 0; g0 < nIn; g0 += LANES) {
            const bool have = g0 + lane < nIn;
            unsigned short ordMine = ordReg[0];
#pragma unroll
            for (int g = 1; g < GRP_CH / LANES; ++g) ordMine = (g0 == g * LANES) ? ordReg[g] : ordMine;
            const size_t ri = (size_t)r0 + (have ? (size_t)ordMine : 0u);
            pvol_ray pr = A.rays[ri];
            V3 o = v3(pr.o[0], pr.o[1], pr.o[2]), d = v3(pr.d[0], pr.d[1], pr.d[2]);
            RayD ray;
            ray.o = o; ray.d = d; ray.mint = pr.mint; ray.maxt = pr.maxt;
            float t0 = 0.f, t1 = 0.f;
            const bool hit = have && S.volKind != PVOL_VOLUME_NONE && vol_intersect(S, ray, &t0, &t1) && (t1 - t0) != 0.f;
            const int nS = hit ? (int)ceilf((t1 - t0) / S.stepSize) : 0;
            const float step = hit ? (t1 - t0) / nS : 0.f;
            V3 pPrev = o + d * t0;
            bool inPrev = hit && box_inside(S.extLo, S.extHi, xform_point(S.w2v, pPrev));
            float tcur = t0 + pr.scatter_u * step;
            const V3 w = -d;
            float Lv[32];
#pragma unroll
            for (int b = 0; b < 32; ++b) Lv[b] = 0.f;
            float lenLast = 0.f, lastRk = 0.f;
            uint32_t uCount = 0;
            bool bad = false;
            int maxN = nS;
            for (int off = 32; off > 0; off >>= 1) maxN = max(maxN, __shfl_xor(maxN, off));
            wc.rays += __popcll(__ballot(have));
            for (int j = 0; j < maxN; ++j) {
                const bool act = j < nS;
                const V3 p = o + d * tcur;
                if (act) tcur += step;
                const V3 pv = xform_point(S.w2v, p);
                const bool inP = act && box_inside(S.extLo, S.extHi, pv);
                float lenStep = 0.f;
                if (act) {
                    V3 dseg = p - pPrev;
                    if (inPrev && inP) {
                        V3 a = pPrev + dseg * 0.f, b = pPrev + dseg * 1.f;
                        lenStep = len(a - b);
                    } else {
                        lenStep = analytic_tau_length(S, pPrev, dseg, 0.f, 1.f);
                    }
                    if (!(lenStep * sigTmax < 6.8f)) bad = true;   // the roulette could fire: sequential kernel (photonvolume.cpp:156-161)
                }
                // ---- direct lighting geometry (photonvolume.cpp:178-203), light 0 (at most one light here)
                float fallReg = 1.f, d2Reg = 1.f, exitLen = 0.f, ph = 0.f;
                bool lit = false, distant = true;
                if (inP && !blackS1 && nLights > 0) {
                    const DevLight &light = S.lights[0];
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
                    const bool black = (fallReg == 0.f) || lightBlack;
                    if (!black && !(rowsOK ? tri_rows_occluded(L.trows, S.nTris, vis.o, vis.d, vis.mint, vis.maxt) : lane_occluded(S, vis))) {
                        lit = true;
                        V3 dv = xform_vector(S.w2v, vis.d);
                        V3 dvInv = v3(1.f / dv.x, 1.f / dv.y, 1.f / dv.z);
                        exitLen = inside_exit_length(S, vis.o, vis.d, pv, dvInv, vis.maxt);
                        ph = phase_hg(w, -wo, S.g);
                        ++uCount;
                    }
                }
                wc.steps += __popcll(__ballot(act));
                if (STATS) wc.unocc += __popcll(__ballot(lit));
                // ---- k-NN gather of the group
                float acc[32];
#pragma unroll
                for (int b = 0; b < 32; ++b) acc[b] = 0.f;
                float rk = 0.f;
                int nFoundLane = k;
                const bool need = inP && S.nPhotons > 0u;
                bool done = !need;      // lanes whose L_ii is final
                bool viaPlan = false;   // acc holds a raw flux sum of exactly k photons (scaled below)
                const uint64_t needMask = __ballot(need);
                if (needMask) {
                    // The bucket plan, at most twice: lanes whose guessed radius turned out too small or too large get one more
                    // shared attempt with a corrected radius (they come in clusters: ~11 lanes per affected step on C2) before
                    // the per-lane exact lookup takes whatever is left.
                    float Tretry = 0.f;
                    for (int attempt = 0; attempt < 2; ++attempt) {
                    const bool needP = need && !done && (attempt == 0 || Tretry > 0.f);
                    if (!__ballot(needP)) break;
                    // per-lane search radius^2: 1.3 x the larger of this ray's previous step and the previous group's
                    // mean at this step; the full radius when neither exists.  The bucket covers the largest of them.
                    float gbl = lastRk;
                    if (j < PREV_N) gbl = fmaxf(gbl, M.prevRk[j]);
                    float Tl = (gbl > 0.f && gbl * A.grpGuess < S.maxDistSq) ? gbl * A.grpGuess : S.maxDistSq;
                    if (attempt == 1) { Tl = Tretry; gbl = 0.f; }   // second chance: the radius the first attempt asked for, no widening
                    bool fullR = !(Tl < S.maxDistSq);
                    float T = needP ? Tl : 0.f;
                    T = wave_max(T);
                    int Mb = -1;
                    if (k >= 10 && k <= 64) {
                        // centre and spread of the query points
                        const float big = 3.0e38f;
                        float lx = needP ? p.x : big, ly = needP ? p.y : big, lz = needP ? p.z : big;
                        float hx = needP ? p.x : -big, hy = needP ? p.y : -big, hz = needP ? p.z : -big;
                        lx = -wave_max(-lx); ly = -wave_max(-ly); lz = -wave_max(-lz);
                        hx = wave_max(hx); hy = wave_max(hy); hz = wave_max(hz);
                        const V3 c = v3(0.5f * (lx + hx), 0.5f * (ly + hy), 0.5f * (lz + hz));
                        float rho = needP ? len(p - c) : 0.f;
                        rho = wave_max(rho);
                        const float Rs = (sqrtf(T) + rho) * 1.0001f + 1e-6f;   // superset by the triangle inequality, with rounding slack
                        unsigned long long tst = 0, ts0 = STATS ? stamp() : 0ull;
                        Mb = stage_bucket(S, M.G, L.pos, c, Rs, lane, tst);
                        // the bucket covers more than this lane asked for when other lanes guessed larger: take it (up to
                        // GRP_WIDEN x the guess) -- a wider search ball costs this lane nothing and spares it a failed guess
                        if (needP && gbl > 0.f) {
                            const float cover = (sqrtf(T) + rho) - len(p - c);
                            Tl = fminf(S.maxDistSq, fmaxf(Tl, fminf(cover * cover, gbl * GRP_WIDEN)));
                            fullR = !(Tl < S.maxDistSq);
                        }
                        if (STATS) { wc.tested += (unsigned long long)max(Mb, 0); wc.cySearch += stamp() - ts0; }
                    }
                    if (STATS && Mb < 0) wc.diag1 += __popcll(needMask);   // bucket plan skipped (overflow, k out of range)
                    if (Mb >= 0) {
                        const unsigned long long tp1 = STATS ? stamp() : 0ull;
                        typedef float nf4 __attribute__((ext_vector_type(4)));
                        const float *bX = L.pos, *bY = L.pos + GRP_PITCH, *bZ = L.pos + 2 * GRP_PITCH, *bI = L.pos + 3 * GRP_PITCH;
                        const nf4 px4 = {p.x, p.x, p.x, p.x}, py4 = {p.y, p.y, p.y, p.y}, pz4 = {p.z, p.z, p.z, p.z};
                        // DistanceSquared(photon.p, p) (kdtree.h:180) of four bucket photons: same operations in the same order per
                        // element, two elements per packed instruction
#define GRP_D2X4(c0) ({ const nf4 dx_ = *reinterpret_cast<const nf4 *>(bX + (c0)) - px4, dy_ = *reinterpret_cast<const nf4 *>(bY + (c0)) - py4, \
                                  dz_ = *reinterpret_cast<const nf4 *>(bZ + (c0)) - pz4; dx_ * dx_ + dy_ * dy_ + dz_ * dz_; })
                        // ---- pass 1: per-lane histogram of DistanceSquared over [Tl/4, Tl).  The bucket is padded to a
                        // multiple of four with far-away sentinels.
                        const float Tlo = 0.25f * Tl;
                        const float scale = (float)GRP_BINS / (Tl - Tlo);
#pragma unroll
                        for (int wd = 0; wd < GRP_BINS / 8; ++wd) L.hist[wd * LANES + lane] = 0u;
                        int below = 0, cnt = 0;
                        float dmax = 0.f;
                        const float TlEff = needP ? Tl : -1.f;   // lanes without a lookup accept nothing
                        const uint32_t histBase = (uint32_t)lane;
                        nf4 nX = *reinterpret_cast<const nf4 *>(bX), nY = *reinterpret_cast<const nf4 *>(bY), nZ = *reinterpret_cast<const nf4 *>(bZ);
                        for (int c0 = 0; c0 < Mb; c0 += 4) {
                            // the next quartet's coordinates are requested before this one's arithmetic (the padded bucket makes the
                            // read past the end harmless)
                            const nf4 dx_ = nX - px4, dy_ = nY - py4, dz_ = nZ - pz4;
                            nX = *reinterpret_cast<const nf4 *>(bX + c0 + 4); nY = *reinterpret_cast<const nf4 *>(bY + c0 + 4); nZ = *reinterpret_cast<const nf4 *>(bZ + c0 + 4);
                            const nf4 dd = dx_ * dx_ + dy_ * dy_ + dz_ * dz_;
#pragma unroll
                            for (int u = 0; u < 4; ++u) {   // branch-free: predicates are 0/1 integers, the histogram add is +0 when not counted
                                const float d2 = dd[u];
                                const int inT = d2 < TlEff ? 1 : 0;
                                const int low = d2 < Tlo ? 1 : 0;
                                const int bin = min(GRP_BINS - 1, (int)((d2 - Tlo) * scale));
                                cnt += inT;
                                below += inT & low;
                                dmax = fmaxf(dmax, inT ? d2 : 0.f);
                                const int hb = (inT & (low ^ 1)) ? bin : 0;
                                const uint32_t inc = (uint32_t)(inT & (low ^ 1)) << ((hb & 7) * 4);
                                atomicAdd(&L.hist[(uint32_t)(hb >> 3) * LANES + histBase], inc);
                            }
                        }
                        bool ok = needP && cnt >= k && below < k;
                        // the full radius holds fewer than k photons: all of them count, r^2 = the farthest (photonvolume.cpp:76-105)
                        const bool shortSet = needP && fullR && cnt < k && cnt < 250;
                        if (STATS && attempt == 0) wc.diag0 += __popcll(__ballot(needP && ((cnt < k && !shortSet) || (cnt >= k && below >= k))));   // radius guess too small / too large
                        if (attempt == 0 && needP) {   // what a second attempt should search: 2.2 x (too few found) or 0.3 x (k-th below the histogram)
                            if (cnt < k && !shortSet) Tretry = fminf(S.maxDistSq, 2.2f * Tl);
                            else if (cnt >= k && below >= k) Tretry = 0.3f * Tl;
                        }
                        int bstar = -1, cumBelow = 0, binCount = 0;
                        int cumAll = below;
                        {
#pragma unroll
                            for (int wd = 0; wd < GRP_BINS / 8; ++wd) {
                                const uint32_t word = L.hist[wd * LANES + lane];
#pragma unroll
                                for (int s = 0; s < 8; ++s) {
                                    const int cb = (int)((word >> (4 * s)) & 15u);
                                    if (bstar < 0 && cumAll + cb >= k) { bstar = 8 * wd + s; cumBelow = cumAll; binCount = cb; }
                                    cumAll += cb;
                                }
                            }
                        }
                        // a 4-bit counter that wrapped (> 15 values in one bin) makes the counters' total fall short of cnt: not trusted
                        ok = ok && bstar >= 0 && binCount <= GRP_MINI && cumAll == cnt;
                        // ---- pass 2: (a) the values of bin bstar go to the lane's mini list; (b) every bucket photon that can
                        // belong to SOME lane's k-NN set (it lies in a bin <= that lane's bstar) goes to the compact list, in bucket order
                        int nb = 0, nC = 0;
                        const bool planLane = ok || shortSet;
                        if (__ballot(planLane)) {
                            const float TlPlan = planLane ? Tl : -1.f;
                            const int bEq = ok ? bstar : -1000;                    // bin whose values are kept (none for short sets)
                            const int bLe = shortSet ? 1000 : (ok ? bstar : -1000);   // last bin that can hold a member
                            nf4 mX = *reinterpret_cast<const nf4 *>(bX), mY = *reinterpret_cast<const nf4 *>(bY), mZ = *reinterpret_cast<const nf4 *>(bZ);
                            for (int c0 = 0; c0 < Mb; c0 += 4) {
                                const nf4 dx_ = mX - px4, dy_ = mY - py4, dz_ = mZ - pz4;
                                mX = *reinterpret_cast<const nf4 *>(bX + c0 + 4); mY = *reinterpret_cast<const nf4 *>(bY + c0 + 4); mZ = *reinterpret_cast<const nf4 *>(bZ + c0 + 4);
                                const nf4 dd = dx_ * dx_ + dy_ * dy_ + dz_ * dz_;
#pragma unroll
                                for (int u = 0; u < 4; ++u) {   // branch-free; bins below the histogram range count as -1
                                    const float d2 = dd[u];
                                    const int inT = d2 < TlPlan ? 1 : 0;
                                    const int binH = min(GRP_BINS - 1, (int)((d2 - Tlo) * scale));
                                    const int bin = d2 < Tlo ? -1 : binH;
                                    const int inBin = inT & (bin == bEq ? 1 : 0);
                                    L.mini[nb * LANES + lane] = d2;   // kept only if inBin: the slot is overwritten otherwise
                                    nb += inBin;
                                    const int possible = inT & (bin <= bLe ? 1 : 0);
                                    L.clist[nC] = (unsigned short)(c0 + u);   // every lane stores the same value; kept only if some lane may needP it
                                    nC += __ballot(possible != 0) != 0ull ? 1 : 0;
                                }
                            }
                        }
                        // sort the mini list (<= GRP_MINI values; absent slots are +inf): odd-even transposition network
                        float sv[GRP_MINI];
#pragma unroll
                        for (int t = 0; t < GRP_MINI; ++t) sv[t] = t < nb ? L.mini[t * LANES + lane] : INFINITY;
#pragma unroll
                        for (int rnd = 0; rnd < GRP_MINI; ++rnd) {
#pragma unroll
                            for (int t = rnd & 1; t + 1 < GRP_MINI; t += 2) {
                                const float lo = fminf(sv[t], sv[t + 1]), hi = fmaxf(sv[t], sv[t + 1]);
                                sv[t] = lo; sv[t + 1] = hi;
                            }
                        }
                        const int m = k - cumBelow;   // 1-based rank of the k-th inside its bin
                        float rkC = sv[0];   // this attempt's k-th distance^2 (rk itself belongs to the lanes already served)
                        int nLessIn = 0;
#pragma unroll
                        for (int t = 1; t < GRP_MINI; ++t) rkC = (m == t + 1) ? sv[t] : rkC;
#pragma unroll
                        for (int t = 0; t < GRP_MINI; ++t) nLessIn += (sv[t] < rkC) ? 1 : 0;
                        int quota0 = k - cumBelow - nLessIn;   // ties at rkC are taken in bucket order
                        ok = ok && rkC < Tl;
                        if (shortSet) { ok = true; rkC = dmax; quota0 = 256; }
                        const unsigned long long tp3 = STATS ? stamp() : 0ull;
                        if (STATS) wc.cySelect += tp3 - tp1;
                        // ---- pass 3: flux.  Row addresses are wave-uniform: the rows come through the scalar cache into SGPRs
                        // (constant address space: the map is read-only while this kernel runs) and cost no vector-memory
                        // bandwidth.  Two list entries per iteration; acc = fma(row, 0 or 1, acc) is the exact addition for
                        // members and a no-op for the others, without a branch.
                        nC = __builtin_amdgcn_readfirstlane(nC);
                        if (lane == 0) L.clist[nC] = (unsigned short)Mb;   // odd count: pair the last entry with a sentinel slot (never a member)
                        __syncthreads();
                        if (nC && __ballot(ok)) {
                            typedef const __attribute__((address_space(4))) nf4 cf4;
                            int quota = quota0;
                            // the LDS side of the next pair (list entries, coordinates, photon indices) is read one iteration ahead
                            int nca = (int)L.clist[0], ncb = (int)L.clist[1];
                            float nax = bX[nca], nay = bY[nca], naz = bZ[nca], nbx = bX[ncb], nby = bY[ncb], nbz = bZ[ncb];
                            float nia = bI[nca], nib = bI[ncb];
                            for (int i = 0; i < nC; i += 2) {
                                const float ax = nax, ay = nay, az = naz, bx = nbx, by = nby, bz = nbz;
                                cf4 *ra = (cf4 *)(S.alpha4 + (size_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(nia)) * 8);
                                cf4 *rb = (cf4 *)(S.alpha4 + (size_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(nib)) * 8);
                                nca = (int)L.clist[i + 2]; ncb = (int)L.clist[i + 3];   // past the end: stale but in-range slots, never used
                                nax = bX[nca]; nay = bY[nca]; naz = bZ[nca]; nbx = bX[ncb]; nby = bY[ncb]; nbz = bZ[ncb];
                                nia = bI[nca]; nib = bI[ncb];
                                nf4 rowA[8], rowB[8];
#pragma unroll
                                for (int qq = 0; qq < 8; ++qq) { rowA[qq] = ra[qq]; rowB[qq] = rb[qq]; }
                                const float dA = dist2_ref(make_float4(ax, ay, az, 0.f), p), dB = dist2_ref(make_float4(bx, by, bz, 0.f), p);
                                bool mA = ok && dA < rkC;
                                if (ok && dA == rkC && quota > 0) { mA = true; --quota; }
                                bool mB = ok && dB < rkC;
                                if (ok && dB == rkC && quota > 0) { mB = true; --quota; }
                                const float fA = mA ? 1.f : 0.f, fB = mB ? 1.f : 0.f;
#pragma unroll
                                for (int qq = 0; qq < 8; ++qq) {
                                    acc[4 * qq] = __builtin_fmaf(rowA[qq].x, fA, acc[4 * qq]); acc[4 * qq + 1] = __builtin_fmaf(rowA[qq].y, fA, acc[4 * qq + 1]);
                                    acc[4 * qq + 2] = __builtin_fmaf(rowA[qq].z, fA, acc[4 * qq + 2]); acc[4 * qq + 3] = __builtin_fmaf(rowA[qq].w, fA, acc[4 * qq + 3]);
                                }
#pragma unroll
                                for (int qq = 0; qq < 8; ++qq) {
                                    acc[4 * qq] = __builtin_fmaf(rowB[qq].x, fB, acc[4 * qq]); acc[4 * qq + 1] = __builtin_fmaf(rowB[qq].y, fB, acc[4 * qq + 1]);
                                    acc[4 * qq + 2] = __builtin_fmaf(rowB[qq].z, fB, acc[4 * qq + 2]); acc[4 * qq + 3] = __builtin_fmaf(rowB[qq].w, fB, acc[4 * qq + 3]);
                                }
                            }
                        }
#undef GRP_D2X4
                        if (ok) { done = true; viaPlan = true; rk = rkC; if (shortSet) nFoundLane = cnt; }
                        if (STATS) { wc.kept += (unsigned long long)k * __popcll(__ballot(ok)); wc.cyFlux += stamp() - tp3; }
                    }
                    }   // attempt
                    // ---- lanes the plan did not serve: the wave-cooperative exact lookup, one lane at a time
                    uint64_t todo = __ballot(need && !done);
                    WaveCounters wsave = wc;
                    const unsigned long long tfb = STATS ? stamp() : 0ull;
                    const int nfb = __popcll(todo);
                    while (todo) {
                        const int l = __ffsll((unsigned long long)todo) - 1;
                        todo &= todo - 1;
                        const V3 pl = v3(lane_f(p.x, l), lane_f(p.y, l), lane_f(p.z, l));
                        const V3 wl = v3(lane_f(w.x, l), lane_f(w.y, l), lane_f(w.z, l));
                        const float gl = lane_f(lastRk, l);
                        float guess = gl;
                        if (j < PREV_N) guess = fmaxf(guess, M.prevRk[j]);
                        float rkl;
                        const f4 Lf = lphoton<STATS, NREG>(S, M.G, wl, pl, sigS4, lane, wc, guess, &rkl);   // inside the extent: sigma_s * 1
                        const float lf[4] = {Lf.x, Lf.y, Lf.z, Lf.w};
#pragma unroll
                        for (int qq = 0; qq < 8; ++qq) {
#pragma unroll
                            for (int cc = 0; cc < 4; ++cc) {
                                const float v = lane_f(lf[cc], qq);
                                if (lane == l) acc[4 * qq + cc] = v;
                            }
                        }
                        if (lane == l) rk = rkl;
                    }
                    if (STATS) { wc = wsave; wc.retries += nfb; wc.diag2 += stamp() - tfb; }   // the exact lookups are accounted as retries + their cycles, not in the phase counters
                    const float rkGuess = (need && nFoundLane >= k) ? rk : 0.f;   // as lphoton's rkOut: a k-th distance exists only for full sets
                    {   // mean k-th distance^2 of the group at this step -> guess of the next group
                        float sr = rkGuess, sn = rkGuess > 0.f ? 1.f : 0.f;
                        for (int off = 32; off > 0; off >>= 1) { sr += __shfl_xor(sr, off); sn += __shfl_xor(sn, off); }
                        if (j < PREV_N && lane == 0) M.prevRk[j] = sn > 0.f ? sr / sn : 0.f;
                    }
                    if (need) lastRk = rkGuess;
                }
                // ---- the recurrence of photonvolume.cpp:150-218, 30 bins per lane.  Per-lane scalars first:
                //   L_ii = Sum(alpha) * phase / (4/3 pi r^3 sigma_s)     (photonvolume.cpp:99-105; planned lanes hold the raw sum)
                //   L_d  = I * [falloff / d^2] * exp(-sigma_t * exit) * phase * nLights          (:178-203)
                float liiScale = 0.f;   // raw sum -> L_ii * sigma_s
                if (viaPlan) {
                    const float dV = rk * sqrtf(rk);
                    if (dV != 0.f && !blackS1 && nFoundLane >= 10) liiScale = wIso * __builtin_amdgcn_rcpf(float(4.0 / 3.0 * (double)K_PI * (double)dV));
                }
                const float dens = inP ? 1.f : 0.f;
                const bool useLii = inP && (ySa1 != 0.0 || ySs1 != 0.0);
                float ldScale = 0.f;
                if (lit) ldScale = (distant ? 1.f : fallReg * __builtin_amdgcn_rcpf(d2Reg)) * ph * float(nLights);
                const float kLen = -1.442695041f * lenStep, kExit = -1.442695041f * exitLen;   // exp(-x) = exp2(-x log2 e)
                const float stepD = step * dens;
                if (act) {
#pragma unroll
                    for (int qq = 0; qq < 8; ++qq) {
                        __builtin_amdgcn_sched_barrier(0);   // keep the constants of one bin quartet live at a time
                        const f4 a4 = cA[qq], s4 = cS[qq], le4 = cLe[qq], al4 = cAl[qq], i4 = cI[qq], r4 = cRs[qq];
                        const float av[4] = {a4.x, a4.y, a4.z, a4.w}, sv[4] = {s4.x, s4.y, s4.z, s4.w}, lev[4] = {le4.x, le4.y, le4.z, le4.w};
                        const float alv[4] = {al4.x, al4.y, al4.z, al4.w}, iv[4] = {i4.x, i4.y, i4.z, i4.w}, rv[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) {
                            const int b = 4 * qq + cc;
                            if (b >= 30) continue;
                            const float sT = av[cc] + sv[cc];
                            const float Trb = __builtin_amdgcn_exp2f(sT * kLen);
                            const float Ld = (iv[cc] * ldScale) * __builtin_amdgcn_exp2f(sT * kExit);
                            float Lii = acc[b];
                            if (viaPlan) Lii = acc[b] * liiScale * rv[cc];
                            const float Li = useLii ? Ld + alv[cc] * Lii : Ld;
                            Lv[b] = (av[cc] * lev[cc] * stepD) + (sv[cc] * Li * stepD) + (Trb * Lv[b]);
                        }
                    }
                    pPrev = p;
                    inPrev = inP;
                    lenLast = lenStep;
                }
            }
            // ---- outputs
            if (have) {
                if (bad) {
                    atomicOr(A.needSeq, 1u);
                } else {
                    const uint32_t draws = hit ? 4u + 7u * (uint32_t)nS + uCount : 0u;
                    const float kLast = -1.442695041f * lenLast;
                    if (A.outputKind == PVOL_OUT_SPECTRAL) {
                        float *op = A.out + ri * 60;
#pragma unroll
                        for (int b = 0; b < 30; ++b) {
                            op[b] = Lv[b];
                            op[30 + b] = hit ? __builtin_amdgcn_exp2f((L.cst[b] + L.cst[32 + b]) * kLast) : 1.f;
                        }
                    } else {
                        float x = 0.f, y = 0.f, z = 0.f, ty = 0.f;
#pragma unroll
                        for (int qq = 0; qq < 8; ++qq) {
                            __builtin_amdgcn_sched_barrier(0);
                            const f4 a4 = cA[qq], s4 = cS[qq], x4 = cX[qq], y4 = cY[qq], z4 = cZ[qq];
                            const float av[4] = {a4.x, a4.y, a4.z, a4.w}, sv[4] = {s4.x, s4.y, s4.z, s4.w};
                            const float xv[4] = {x4.x, x4.y, x4.z, x4.w}, yv[4] = {y4.x, y4.y, y4.z, y4.w}, zv[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
                            for (int cc = 0; cc < 4; ++cc) {
                                const int b = 4 * qq + cc;
                                if (b >= 30) continue;
                                x += xv[cc] * Lv[b];
                                y += yv[cc] * Lv[b];
                                z += zv[cc] * Lv[b];
                                const float trb = hit ? __builtin_amdgcn_exp2f((av[cc] + sv[cc]) * kLast) : 1.f;
                                ty += yv[cc] * trb;
                            }
                        }
                        const float scale = float(700 - 400) / float(106.856895f * 30);
                        *reinterpret_cast<f4 *>(A.out + ri * 4) = make_float4(x * scale, y * scale, z * scale, ty * 300.f / (106.856895f * 30));
                    }
                    if (A.draws) A.draws[ri] = draws;
                }
            }
            {   // stream positions: one atomic per group when all its rays belong to one stream (the usual case)
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

extern "C" size_t pvol_group_lds_bytes(int candCap) {
    (void)candCap;   // the fallback candidate arrays alias the bucket
    return (size_t)PREV_N * 4 + PAINT_CAP * 4 + (GRP_CAP + 4) * 16 + (GRP_MINI + 1) * LANES * 4 + (GRP_BINS / 8) * LANES * 4 + 9 * 32 * 4 + (((GRP_CAP + 4) * 2 + 15) & ~15) + GRP_TRI_ROWS * 64;
}

extern "C" hipError_t pvol_launch_li_group(const LiArgs *args, size_t ldsBytes, int candCap, bool stats, uint32_t nWaves, hipStream_t stream) {
    hipLaunchKernelGGL(stream_begin_kernel, dim3((args->nStreams + 255) / 256), dim3(256), 0, stream, args->streams, args->nStreams);
    dim3 grid(nWaves), block(LANES);
    if (stats) hipLaunchKernelGGL((li_group_kernel<true, 4>), grid, block, ldsBytes, stream, *args);
    else hipLaunchKernelGGL((li_group_kernel<false, 4>), grid, block, ldsBytes, stream, *args);
    return hipGetLastError();
}
