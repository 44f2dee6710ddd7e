// pvol_bvh_dev.h -- included by pvol_math.h.  SURVEY 8(f)-4: Scene::Intersect / IntersectP (core/scene.h:50-61) over a
// bounding-volume hierarchy for scenes with more triangles than the DevScene's embedded array holds (PVOL_MAX_TRIS);
// replaces BVHAccel::Intersect / IntersectP (accelerators/bvh.cpp:585-690).  The hierarchy is a device-built LBVH
// (pvol_bvh.hip): leaves are single triangles in Morton order of their centroids.
//
// Layout in HBM:
//   bvhNodes [n-1][4] float4: {left box lo, left child} {left box hi, right child} {right box lo, first slot} {right box hi,
//                             last slot}; a child >= 0 is an inner node, < 0 is the leaf ~child (a slot of bvhTris).
//                             One 64-B fetch per visited node tests both children.
//   bvhTris  [n][3]  float4: {p1, original index} {p2, material} {p3, flip_normal} in Morton order (int fields as bits).
//
// Results are those of the linear scan the small scenes use: the per-triangle arithmetic is the same
// function, the smallest t wins and, of several triangles at the same t, the one with the highest ORIGINAL index -- the
// serial scan rejects `t > maxt` only, so a later triangle at the same t replaces an earlier one.  A subtree is left out
// only when its box (padded at build time) starts behind the best t so far, so the order of the traversal cannot show.
// One ray per lane; the wave-per-ray kernels call these with wave-uniform arguments (every lane walks the same path).
#ifndef PVOL_BVH_DEV_H
#define PVOL_BVH_DEV_H

#define PVOL_BVH_STACK 64   // the keys are 62 bits (30 Morton + 32 index), so no path of the radix tree is deeper

// shapes/trianglemesh.cpp:116-160 for one triangle given by its vertices (closest hit: t reported)
__device__ __forceinline__ bool tri_closest_v(V3 p1, V3 p2, V3 p3, V3 o, V3 d, float mint, float maxt, float *tHit) {
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

// entry distance of the ray into a box, +inf when it misses [mint, maxt]; fminf/fmaxf drop the NaN of 0 * inf
__device__ __forceinline__ float bvh_slab(const float4 lo, const float4 hi, V3 o, V3 inv, float mint, float maxt) {
    const float ax = (lo.x - o.x) * inv.x, bx = (hi.x - o.x) * inv.x;
    const float ay = (lo.y - o.y) * inv.y, by = (hi.y - o.y) * inv.y;
    const float az = (lo.z - o.z) * inv.z, bz = (hi.z - o.z) * inv.z;
    const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), mint));
    const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), maxt));
    return tn <= tf ? tn : INFINITY;
}

// Closest hit.  Returns the slot of the triangle in bvhTris (-1: none) and its t.
__device__ int bvh_closest(const DevScene &S, V3 o, V3 d, float mint, float maxt, float *tHit) {
    const float4 *N = S.bvhNodes, *T = S.bvhTris;
    const V3 inv = v3(1.f / d.x, 1.f / d.y, 1.f / d.z);
    float best = maxt;
    int bestSlot = -1, bestOrig = -1;
    int stack[PVOL_BVH_STACK];
    int sp = 0, node = 0;
    for (;;) {
        const float4 a = N[4 * node], b = N[4 * node + 1], c = N[4 * node + 2], e = N[4 * node + 3];
        float nl = bvh_slab(a, b, o, inv, mint, best), nr = bvh_slab(c, e, o, inv, mint, best);
        int cl = __float_as_int(a.w), cr = __float_as_int(b.w);
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int ch = side ? cr : cl;
            const float nn = side ? nr : nl;
            if (ch < 0 && nn != INFINITY) {
                const int slot = ~ch;
                const float4 q1 = T[3 * slot], q2 = T[3 * slot + 1], q3 = T[3 * slot + 2];
                float t;
                if (tri_closest_v(v3(q1.x, q1.y, q1.z), v3(q2.x, q2.y, q2.z), v3(q3.x, q3.y, q3.z), o, d, mint, best, &t)) {
                    const int orig = __float_as_int(q1.w);
                    if (t < best || orig > bestOrig) { best = t; bestSlot = slot; bestOrig = orig; }
                }
            }
        }
        const bool goL = cl >= 0 && nl != INFINITY, goR = cr >= 0 && nr != INFINITY;
        if (goL && goR) {
            const bool swap = nr < nl;
            if (sp < PVOL_BVH_STACK) stack[sp++] = swap ? cl : cr;
            node = swap ? cr : cl;
        } else if (goL) node = cl;
        else if (goR) node = cr;
        else {
            if (sp == 0) break;
            node = stack[--sp];
        }
    }
    *tHit = best;
    return bestSlot;
}

// Any hit (shapes/trianglemesh.cpp:211-243 per triangle)
__device__ bool bvh_occluded(const DevScene &S, V3 o, V3 d, float mint, float maxt) {
    const float4 *N = S.bvhNodes, *T = S.bvhTris;
    const V3 inv = v3(1.f / d.x, 1.f / d.y, 1.f / d.z);
    int stack[PVOL_BVH_STACK];
    int sp = 0, node = 0;
    for (;;) {
        const float4 a = N[4 * node], b = N[4 * node + 1], c = N[4 * node + 2], e = N[4 * node + 3];
        const float nl = bvh_slab(a, b, o, inv, mint, maxt), nr = bvh_slab(c, e, o, inv, mint, maxt);
        const int cl = __float_as_int(a.w), cr = __float_as_int(b.w);
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int ch = side ? cr : cl;
            const float nn = side ? nr : nl;
            if (ch < 0 && nn != INFINITY) {
                const int slot = ~ch;
                const float4 q1 = T[3 * slot], q2 = T[3 * slot + 1], q3 = T[3 * slot + 2];
                float t;
                if (tri_closest_v(v3(q1.x, q1.y, q1.z), v3(q2.x, q2.y, q2.z), v3(q3.x, q3.y, q3.z), o, d, mint, maxt, &t)) return true;
            }
        }
        const bool goL = cl >= 0 && nl != INFINITY, goR = cr >= 0 && nr != INFINITY;
        if (goL && goR) {
            if (sp < PVOL_BVH_STACK) stack[sp++] = cr;
            node = cl;
        } else if (goL) node = cl;
        else if (goR) node = cr;
        else {
            if (sp == 0) break;
            node = stack[--sp];
        }
    }
    return false;
}

#endif
