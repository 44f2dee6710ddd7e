// oracle/orc_integrator.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see orc_core.h).
//
// CPU restatement of integrators/photonvolume.cpp (Li, LPhoton, Transmittance) and of the
// kd-tree it searches (core/kdtree.h), including the reference's quirks.
#ifndef ORC_INTEGRATOR_H
#define ORC_INTEGRATOR_H

#include "orc_scene.h"

namespace orc {

// core/photonshooter.h:20-27 without the two tag floats.
struct Photon {
    V3 p;
    Spec alpha;
    V3 wi;
};

// core/kdtree.h:44-60
struct KdNode {
    float splitPos;
    uint32_t splitAxis : 2;
    uint32_t hasLeftChild : 1, rightChild : 29;
};

struct Counters {
    uint64_t n_rays, n_steps, n_lookups, n_nodes_visited, n_heap_offers, n_kept, n_lookups_lt10,
        n_shadow_unoccluded, n_density_evals, n_draws;
    Counters() { memset(this, 0, sizeof(*this)); }
    void add(const Counters &o) {
        n_rays += o.n_rays; n_steps += o.n_steps; n_lookups += o.n_lookups; n_nodes_visited += o.n_nodes_visited;
        n_heap_offers += o.n_heap_offers; n_kept += o.n_kept; n_lookups_lt10 += o.n_lookups_lt10;
        n_shadow_unoccluded += o.n_shadow_unoccluded; n_density_evals += o.n_density_evals; n_draws += o.n_draws;
    }
};

// core/photonshooter.h:40-50
struct ClosePhoton {
    uint32_t photon;  // index into KdTree::data (stands for the pointer; same ordering)
    float distanceSquared;
    bool operator<(const ClosePhoton &o) const {
        return distanceSquared == o.distanceSquared ? (photon < o.photon) : (distanceSquared < o.distanceSquared);
    }
};

// core/kdtree.h:63-183.  Left-balanced: left child is nodeNum+1.
struct KdTree {
    std::vector<KdNode> nodes;
    std::vector<Photon> data;
    uint32_t nNodes, nextFreeNode;

    explicit KdTree(const std::vector<Photon> &d) {
        nNodes = (uint32_t)d.size();
        nextFreeNode = 1;
        nodes.resize(nNodes);
        data.resize(nNodes);
        std::vector<const Photon *> build(nNodes);
        for (uint32_t i = 0; i < nNodes; ++i) build[i] = &d[i];
        if (nNodes) recursive_build(0, 0, (int)nNodes, &build[0]);
    }
    void recursive_build(uint32_t nodeNum, int start, int end, const Photon **b) {
        if (start + 1 == end) {  // kdtree.h:117-121
            nodes[nodeNum].splitAxis = 3;
            nodes[nodeNum].rightChild = (1 << 29) - 1;
            nodes[nodeNum].hasLeftChild = 0;
            data[nodeNum] = *b[start];
            return;
        }
        // kdtree.h:126-132: bounds -> max extent axis -> nth_element at the median
        V3 lo = v3(kInfinity, kInfinity, kInfinity), hi = v3(-kInfinity, -kInfinity, -kInfinity);
        for (int i = start; i < end; ++i) {
            lo = v3(std::min(lo.x, b[i]->p.x), std::min(lo.y, b[i]->p.y), std::min(lo.z, b[i]->p.z));
            hi = v3(std::max(hi.x, b[i]->p.x), std::max(hi.y, b[i]->p.y), std::max(hi.z, b[i]->p.z));
        }
        V3 diag = hi - lo;
        int axis = (diag.x > diag.y && diag.x > diag.z) ? 0 : (diag.y > diag.z ? 1 : 2);  // geometry.h:422-430
        int splitPos = (start + end) / 2;
        std::nth_element(&b[start], &b[splitPos], &b[end], [axis](const Photon *d1, const Photon *d2) {
            float a1 = comp(d1->p, axis), a2 = comp(d2->p, axis);
            return a1 == a2 ? (d1 < d2) : a1 < a2;
        });
        nodes[nodeNum].splitPos = comp(b[splitPos]->p, axis);
        nodes[nodeNum].splitAxis = axis;
        nodes[nodeNum].rightChild = (1 << 29) - 1;
        nodes[nodeNum].hasLeftChild = 0;
        data[nodeNum] = *b[splitPos];
        if (start < splitPos) {
            nodes[nodeNum].hasLeftChild = 1;
            uint32_t childNum = nextFreeNode++;
            recursive_build(childNum, start, splitPos, b);
        }
        if (splitPos + 1 < end) {
            nodes[nodeNum].rightChild = nextFreeNode++;
            recursive_build(nodes[nodeNum].rightChild, splitPos + 1, end, b);
        }
    }
};

// PhotonProcess (core/photonshooter.h:53-60,186-203): bounded max-heap of the nLookup nearest.
struct PhotonProcess {
    ClosePhoton *photons;
    uint32_t nLookup, nFound;
    Counters *ctr;
    void operator()(uint32_t photon, float distSquared, float &maxDistSquared) {
        if (ctr) ++ctr->n_heap_offers;
        if (nFound < nLookup) {
            photons[nFound].photon = photon;
            photons[nFound].distanceSquared = distSquared;
            ++nFound;
            if (nFound == nLookup) {
                std::make_heap(&photons[0], &photons[nLookup]);
                maxDistSquared = photons[0].distanceSquared;
            }
        } else {
            std::pop_heap(&photons[0], &photons[nLookup]);
            photons[nLookup - 1].photon = photon;
            photons[nLookup - 1].distanceSquared = distSquared;
            std::push_heap(&photons[0], &photons[nLookup]);
            maxDistSquared = photons[0].distanceSquared;
        }
    }
};

// core/kdtree.h:157-183
inline void kd_lookup(const KdTree &t, uint32_t nodeNum, V3 p, PhotonProcess &proc, float &maxDistSquared) {
    const KdNode *node = &t.nodes[nodeNum];
    if (proc.ctr) ++proc.ctr->n_nodes_visited;
    int axis = node->splitAxis;
    if (axis != 3) {
        float pa = comp(p, axis);
        float dist2 = (pa - node->splitPos) * (pa - node->splitPos);
        if (pa <= node->splitPos) {
            if (node->hasLeftChild) kd_lookup(t, nodeNum + 1, p, proc, maxDistSquared);
            if (dist2 < maxDistSquared && node->rightChild < t.nNodes) kd_lookup(t, node->rightChild, p, proc, maxDistSquared);
        } else {
            if (node->rightChild < t.nNodes) kd_lookup(t, node->rightChild, p, proc, maxDistSquared);
            if (dist2 < maxDistSquared && node->hasLeftChild) kd_lookup(t, nodeNum + 1, p, proc, maxDistSquared);
        }
    }
    float dist2 = length_sq(t.data[nodeNum].p - p);
    if (dist2 < maxDistSquared) proc(nodeNum, dist2, maxDistSquared);
}

struct Integrator {
    // PhotonVolumeIntegrator members (integrators/photonvolume.h:17-33)
    float stepSize, maxDist, maxDistSquared;
    int nUsed;
    const Scene *scene;
    const KdTree *volumeMap;  // may be NULL (photonvolume.cpp:69)
};

// PhotonVolumeIntegrator::Transmittance, photonvolume.cpp:15-30.  `sample_offset` < 0 means
// sample == NULL (every call on the hot path): step = 4*stepSize, one RandomFloat.
inline Spec transmittance(const Integrator &I, const Ray &ray, Rng &rng, Counters *ctr) {
    if (I.scene->vol.kind == PVOL_VOLUME_NONE) return spec_const(1.f);
    float step = 4.f * I.stepSize;
    float offset = rng.random_float();
    VolCounters vc = {0};
    Spec tau = vol_tau(I.scene->vol, ray, step, offset, &vc);
    if (ctr) ctr->n_density_evals += vc.density_evals;
    return spec_exp(-tau);
}

// volumes/rainbow.cpp:41-78 (+ LerpOrZero :6-20, LerpTransfer :22-37, filter core/spectrum.h:300-320)
inline float lerp_or_zero(float theta, float minTheta, float maxTheta, float startW, float endW) {
    if (theta < minTheta || maxTheta < theta) return 0;
    const float thetaRange = maxTheta - minTheta;
    const float wavelengthRange = endW - startW;
    return startW + (theta - minTheta) * wavelengthRange / thetaRange;
}
inline float lerp_transfer(float x, float xMin, float xMax, float y0, float y1) {
    if (x < xMin) return y0;
    if (xMax < x) return y1;
    const float thetaRange = xMax - xMin;
    const float range = y1 - y0;
    return y0 + (x - xMin) * range / thetaRange;
}
inline Spec spec_filter(const Spec &s, float lambda) {
    float deltaLambda = float(700 - 400) / NB;
    float indexWithDecimals = (lambda - 400) / deltaLambda;
    int index = int(indexWithDecimals);
    float t = indexWithDecimals - index;
    Spec ret = spec_const(0.f);
    if (index < 0 || index >= NB) return ret;  // the reference writes out of bounds at lambda == 700 exactly
    ret.c[index] = s.c[index] * t;
    if (index + 1 < NB) ret.c[index + 1] = s.c[index + 1] * (1 - t);
    return ret;
}
inline Spec rainbow_reflection(const Spec &spectrum, V3 w, V3 wi) {
    float cosTheta = dot(wi, -w);
    const float radToDeg = 57.2957;
    float theta = radToDeg * acosf(cosTheta);
    float I = phase_mie_hazy(wi, -w);
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
    if (!lambda) return I * mistI * spectrum;
    Spec rainbow = spec_filter(spectrum, lambda);
    return I * (mistI * spectrum + rainbowI * rainbow);
}

// PhotonVolumeIntegrator::LPhoton, photonvolume.cpp:65-108
inline Spec lphoton(const Integrator &I, ClosePhoton *buf, V3 w, V3 pt, Counters *ctr) {
    Spec L = spec_const(0.f);
    if (!I.volumeMap || I.volumeMap->nNodes == 0) return L;
    const KdTree &map = *I.volumeMap;
    PhotonProcess proc;
    proc.photons = buf;
    proc.nLookup = (uint32_t)I.nUsed;
    proc.nFound = 0;
    proc.ctr = ctr;
    float md2 = I.maxDistSquared;  // passed by value: the caller's bound is not shrunk (photonvolume.cpp:66,76)
    if (ctr) ++ctr->n_lookups;
    kd_lookup(map, 0, pt, proc, md2);
    int nFound = (int)proc.nFound;
    if (nFound < 10) { if (ctr) ++ctr->n_lookups_lt10; return L; }
    if (ctr) ctr->n_kept += nFound;
    Spec totalFlux = spec_const(0.f);
    float maxmd = 0.0;
    for (int i = 0; i < nFound; ++i) {
        const Photon *p = &map.data[buf[i].photon];
        float distSq = buf[i].distanceSquared;
        if (distSq > maxmd) maxmd = distSq;
        totalFlux += p->alpha * vol_phase(I.scene->vol, p->p, p->wi, -w);
    }
    float distSq = maxmd;
    float dV = distSq * sqrtf(distSq);  // `sqrt(float)` resolves to the float overload under <math.h> in C++
    Spec scale = vol_sigma_s(I.scene->vol, pt);
    if (dV != 0.0 && !is_black(scale)) {
        // `4.0/3.0*M_PI*dV` is a double expression converted to float by operator*(float, Spectrum)
        float f = float(4.0 / 3.0 * kPi * dV);
        L += totalFlux / (f * scale);
    }
    return L;
}

// PhotonVolumeIntegrator::Li, photonvolume.cpp:112-222
inline Spec li(const Integrator &I, const Ray &ray, float scatterU, Rng &rng, Spec *T, Counters *ctr,
               std::vector<float> &scratch, std::vector<ClosePhoton> &lookupBuf) {
    const Scene &sc = *I.scene;
    const Volume &vr = sc.vol;
    bool rv = (vr.kind == PVOL_VOLUME_RAINBOW);
    if (ctr) ++ctr->n_rays;
    float t0, t1;
    if (vr.kind == PVOL_VOLUME_NONE || !vol_intersect(vr, ray, &t0, &t1) || (t1 - t0) == 0.f) {
        *T = spec_const(1.f);
        return spec_const(0.f);
    }
    Spec Lv = spec_const(0.f);
    int nSamples = (int)ceilf((t1 - t0) / I.stepSize);
    float step = (t1 - t0) / nSamples;
    Spec Tr = spec_const(1.f);
    V3 p = ray_at(ray, t0), pPrev;
    V3 w = -ray.d;
    t0 += scatterU * step;

    scratch.resize((size_t)4 * nSamples);
    float *lightNum = &scratch[0];
    ld_shuffle_scrambled_1d(1, nSamples, lightNum, rng);
    float *lightComp = lightNum + nSamples;
    ld_shuffle_scrambled_1d(1, nSamples, lightComp, rng);
    float *lightPos = lightComp + nSamples;
    ld_shuffle_scrambled_2d(1, nSamples, lightPos, rng);
    int sampOffset = 0;
    if ((int)lookupBuf.size() < I.nUsed) lookupBuf.resize(I.nUsed);

    for (int i = 0; i < nSamples; ++i, t0 += step) {
        if (ctr) ++ctr->n_steps;
        pPrev = p;
        p = ray_at(ray, t0);
        Ray tauRay = make_ray(pPrev, p - pPrev, 0.f, 1.f, ray.time);
        VolCounters vc = {0};
        Spec stepTau = vol_tau(vr, tauRay, .5f * I.stepSize, rng.random_float(), &vc);
        Tr = spec_exp(-stepTau);  // assigned, not accumulated (photonvolume.cpp:155)

        if (spec_y(sc.cie, Tr) < 1e-3) {
            const float continueProb = .5f;
            if (rng.random_float() > continueProb) {
                Tr = spec_const(0.f);
                break;
            }
            Tr /= continueProb;
        }

        Spec L_i = spec_const(0.f), L_d = spec_const(0.f), L_ii = spec_const(0.f);
        Spec ss = vol_sigma_s(vr, p, &vc);
        Spec sa = vol_sigma_a(vr, p, &vc);

        if (!is_black(ss) && sc.lights.size() > 0) {
            int nLights = (int)sc.lights.size();
            int ln = std::min((int)floorf(lightNum[sampOffset] * nLights), nLights - 1);
            const Light &light = sc.lights[ln];
            float pdf;
            Ray vis;
            V3 wo;
            Spec L = light_sample_L(light, p, 0.f, ray.time, &wo, &pdf, &vis);
            if (!is_black(L) && pdf > 0.f && !scene_intersect_p(sc, vis)) {
                if (ctr) ++ctr->n_shadow_unoccluded;
                Spec Ld = L * transmittance(I, vis, rng, ctr);
                if (rv) L_d = rainbow_reflection(Ld, ray.d, wo);
                else L_d = vol_phase(vr, p, w, -wo) * Ld * float(nLights) / pdf;
            }
        }
        if (!rv) L_ii += lphoton(I, &lookupBuf[0], w, p, ctr);

        if (spec_y(sc.cie, sa) != 0.0 || spec_y(sc.cie, ss) != 0.0) L_i = L_d + (ss / (sa + ss)) * L_ii;
        else L_i = L_d;

        Spec nLv = (sa * vol_lve(vr, p, &vc) * step) + (ss * L_i * step) + (Tr * Lv);
        Lv = nLv;
        sampOffset++;
        if (ctr) ctr->n_density_evals += vc.density_evals;
    }
    *T = Tr;
    return Lv;
}

}  // namespace orc
#endif
