// pvol_sphere_dev.h -- included by pvol_math.h.  Shape "sphere" (SURVEY 8(f)-3): Sphere::Intersect / IntersectP
// (shapes/sphere.cpp:59-157, 160-216) in object space, operation for operation (Quadratic: core/pbrt.h:309-323; the ray goes
// to object space as Transform::operator()(const Ray&, Ray*) does, core/transform.h:260-269), and the DifferentialGeometry
// a hit hands to the BSDF (sphere.cpp:106-149 -> core/diffgeom.cpp:46-54): p = ObjectToWorld(phit), dpdu, dpdv taken to world
// space, nn = Normalize(Cross(dpdu, dpdv)) flipped by ReverseOrientation ^ TransformSwapsHandedness.
// A scene holds at most PVOL_MAX_SPHERES of them in the DevScene (scalar loads); they are tested after the triangles, so of
// a triangle and a sphere at the same t the sphere wins (the later primitive, as everywhere in this library).
#ifndef PVOL_SPHERE_DEV_H
#define PVOL_SPHERE_DEV_H

__device__ __forceinline__ bool sphere_clipped(const DevSphere &sp, V3 phit, float phi) {
    return (sp.zmin > -sp.radius && phit.z < sp.zmin) || (sp.zmax < sp.radius && phit.z > sp.zmax) || phi > sp.phiMax;
}
__device__ __forceinline__ float sphere_phi(const DevSphere &sp, V3 &phit) {
    if (phit.x == 0.f && phit.y == 0.f) phit.x = 1e-5f * sp.radius;
    float phi = (float)atan2((double)phit.y, (double)phit.x);   // the double routine, rounded: what a correctly rounded atan2f returns
    if (phi < 0.f) phi += 2.f * K_PI;
    return phi;
}
// thit and the object-space hit point; IntersectP is the same test without the differential geometry
__device__ bool sphere_hit(const DevSphere &sp, V3 ow, V3 dw, float mint, float maxt, float *tHit, V3 *phitOut) {
    const V3 o = xform_point(sp.w2o, ow), d = xform_vector(sp.w2o, dw);
    const float A = d.x * d.x + d.y * d.y + d.z * d.z;
    const float B = 2 * (d.x * o.x + d.y * o.y + d.z * o.z);
    const float C = o.x * o.x + o.y * o.y + o.z * o.z - sp.radius * sp.radius;
    const float discrim = B * B - 4.f * A * C;
    if (discrim < 0.f) return false;
    const float rootDiscrim = sqrtf(discrim);
    const float q = (B < 0) ? -.5f * (B - rootDiscrim) : -.5f * (B + rootDiscrim);
    float t0 = q / A, t1 = C / q;
    if (t0 > t1) { const float t = t0; t0 = t1; t1 = t; }
    if (t0 > maxt || t1 < mint) return false;
    float thit = t0;
    if (t0 < mint) {
        thit = t1;
        if (thit > maxt) return false;
    }
    V3 phit = o + d * thit;
    float phi = sphere_phi(sp, phit);
    if (sphere_clipped(sp, phit, phi)) {
        if (thit == t1) return false;
        if (t1 > maxt) return false;
        thit = t1;
        phit = o + d * thit;
        phi = sphere_phi(sp, phit);
        if (sphere_clipped(sp, phit, phi)) return false;
    }
    *tHit = thit;
    *phitOut = phit;
    return true;
}
__device__ void sphere_dg(const DevSphere &sp, V3 phit, V3 *p, V3 *dpduW, V3 *nn) {
    // acosf / sinf through the double routines, rounded once: the values of a correctly rounded libm (the host's is, within
    // its last bit), where the device's float routines may be 1-2 ulp off -- the normal feeds Fresnel terms near grazing angles
    const float theta = (float)acos((double)fminf(fmaxf(phit.z / sp.radius, -1.f), 1.f));
    const float zradius = sqrtf(phit.x * phit.x + phit.y * phit.y);
    const float invzradius = 1.f / zradius;
    const float cosphi = phit.x * invzradius, sinphi = phit.y * invzradius;
    const V3 dpdu = v3(-sp.phiMax * phit.y, sp.phiMax * phit.x, 0.f);
    const V3 dpdv = v3(phit.z * cosphi, phit.z * sinphi, -sp.radius * (float)sin((double)theta)) * (sp.thetaMax - sp.thetaMin);
    *p = xform_point(sp.o2w, phit);
    *dpduW = xform_vector(sp.o2w, dpdu);
    const V3 dpdvW = xform_vector(sp.o2w, dpdv);
    *nn = normalize(cross(*dpduW, dpdvW));
    if (sp.flip) *nn = *nn * -1.f;
}
// Scene::IntersectP over the spheres
__device__ __forceinline__ bool spheres_occluded(const DevScene &S, V3 o, V3 d, float mint, float maxt) {
    for (int i = 0; i < S.nSpheres; ++i) {
        float t; V3 ph;
        if (sphere_hit(S.spheres[i], o, d, mint, maxt, &t, &ph)) return true;
    }
    return false;
}
// Scene::Intersect over the spheres with the ray already shortened to *t by the triangles: index of the sphere that wins, or -1
__device__ __forceinline__ int spheres_closest(const DevScene &S, V3 o, V3 d, float mint, float *t, V3 *phit) {
    int best = -1;
    for (int i = 0; i < S.nSpheres; ++i) {
        float th; V3 ph;
        if (sphere_hit(S.spheres[i], o, d, mint, *t, &th, &ph)) { *t = th; *phit = ph; best = i; }
    }
    return best;
}
#endif
