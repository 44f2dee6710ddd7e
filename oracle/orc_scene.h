// oracle/orc_scene.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see orc_core.h).
//
// CPU restatement of what the hot path reads through `const Scene *`: the volume region,
// the three delta lights, closest/any hit over the world-space triangles.
#ifndef ORC_SCENE_H
#define ORC_SCENE_H

#include "orc_core.h"

namespace orc {

struct Volume {
    int kind;
    Box extent;
    float w2v[16], v2w[16];
    Spec sig_a, sig_s, le;
    float g;
    int nx, ny, nz;
    std::vector<float> density;
};

struct Light {
    int kind;
    V3 pos, dir;
    float l2w[16], w2l[16];
    Spec intensity;
    float cosTotalWidth, cosFalloffStart;
};

struct Material {
    int kind;
    Spec kd, kr, kt;
    float ior, vn;
};

struct Triangle {
    V3 p1, p2, p3;
    int material;
    int flip;
};

struct Sphere {   // shapes/sphere.cpp:41-49
    float o2w[16], w2o[16];
    float radius, zmin, zmax, thetaMin, thetaMax, phiMax;
    int material;
    int flip;
};

struct Scene {
    std::vector<Sphere> spheres;
    Volume vol;
    std::vector<Light> lights;
    std::vector<Triangle> tris;
    std::vector<Material> mats;
    Box world;
    Cie cie;
};

inline void scene_from_pod(const pvol_scene *s, Scene *out) {
    Volume &v = out->vol;
    v.kind = s->volume.kind;
    v.extent.lo = v3(s->volume.extent_min[0], s->volume.extent_min[1], s->volume.extent_min[2]);
    v.extent.hi = v3(s->volume.extent_max[0], s->volume.extent_max[1], s->volume.extent_max[2]);
    memcpy(v.w2v, s->volume.world_to_volume, sizeof(v.w2v));
    memcpy(v.v2w, s->volume.volume_to_world, sizeof(v.v2w));
    v.sig_a = spec_from(s->volume.sigma_a);
    v.sig_s = spec_from(s->volume.sigma_s);
    v.le = spec_from(s->volume.le);
    v.g = s->volume.g;
    v.nx = s->volume.nx; v.ny = s->volume.ny; v.nz = s->volume.nz;
    v.density.clear();
    if (v.kind == PVOL_VOLUME_GRID && s->volume.density)
        v.density.assign(s->volume.density, s->volume.density + (size_t)v.nx * v.ny * v.nz);
    out->lights.resize(s->n_lights);
    for (uint32_t i = 0; i < s->n_lights; ++i) {
        const pvol_light &pl = s->lights[i];
        Light &l = out->lights[i];
        l.kind = pl.kind;
        l.pos = v3(pl.pos[0], pl.pos[1], pl.pos[2]);
        l.dir = v3(pl.dir[0], pl.dir[1], pl.dir[2]);
        memcpy(l.l2w, pl.light_to_world, sizeof(l.l2w));
        memcpy(l.w2l, pl.world_to_light, sizeof(l.w2l));
        l.intensity = spec_from(pl.intensity);
        l.cosTotalWidth = pl.cos_total_width;
        l.cosFalloffStart = pl.cos_falloff_start;
    }
    out->tris.resize(s->n_triangles);
    for (uint32_t i = 0; i < s->n_triangles; ++i) {
        const pvol_triangle &pt = s->triangles[i];
        Triangle &t = out->tris[i];
        t.p1 = v3(pt.p[0][0], pt.p[0][1], pt.p[0][2]);
        t.p2 = v3(pt.p[1][0], pt.p[1][1], pt.p[1][2]);
        t.p3 = v3(pt.p[2][0], pt.p[2][1], pt.p[2][2]);
        t.material = pt.material;
        t.flip = pt.flip_normal;
    }
    out->spheres.resize(s->n_spheres);
    for (uint32_t i = 0; i < s->n_spheres; ++i) {
        const pvol_sphere &ps = s->spheres[i];
        Sphere &q = out->spheres[i];
        memcpy(q.o2w, ps.object_to_world, sizeof(q.o2w));
        memcpy(q.w2o, ps.world_to_object, sizeof(q.w2o));
        q.radius = ps.radius; q.zmin = ps.z_min; q.zmax = ps.z_max; q.thetaMin = ps.theta_min; q.thetaMax = ps.theta_max; q.phiMax = ps.phi_max;
        q.material = ps.material; q.flip = ps.flip_normal;
    }
    out->mats.resize(s->n_materials);
    for (uint32_t i = 0; i < s->n_materials; ++i) {
        const pvol_material &pm = s->materials[i];
        Material &m = out->mats[i];
        m.kind = pm.kind;
        m.kd = spec_from(pm.kd); m.kr = spec_from(pm.kr); m.kt = spec_from(pm.kt);
        m.ior = pm.ior; m.vn = pm.vn;
    }
    out->world.lo = v3(s->world_min[0], s->world_min[1], s->world_min[2]);
    out->world.hi = v3(s->world_max[0], s->world_max[1], s->world_max[2]);
    out->cie.x = spec_from(s->cie_x);
    out->cie.y = spec_from(s->cie_y);
    out->cie.z = spec_from(s->cie_z);
    out->cie.scale = s->xyz_scale;
}

// ------------------------------------------------------------------ volume region
// volumes/homogeneous.h:60-63, volumes/volumegrid.h:52-55: transform the ray, slab test.
inline bool vol_intersect(const Volume &v, const Ray &r, float *t0, float *t1) {
    Ray ray = r;
    ray.o = xform_point(v.w2v, r.o);
    ray.d = xform_vector(v.w2v, r.d);
    return box_intersect(v.extent, ray, t0, t1);
}

// volumes/volumegrid.h:60-65
inline float grid_D(const Volume &v, int x, int y, int z) {
    x = std::min(std::max(x, 0), v.nx - 1);
    y = std::min(std::max(y, 0), v.ny - 1);
    z = std::min(std::max(z, 0), v.nz - 1);
    return v.density[(size_t)z * v.nx * v.ny + (size_t)y * v.nx + x];
}
inline float lerpf(float t, float a, float b) { return (1.f - t) * a + t * b; }  // core/pbrt.h:218-220
// volumes/volumegrid.cpp:39-57
inline float grid_density(const Volume &v, V3 Pobj, uint64_t *evals) {
    if (evals) ++*evals;
    if (!box_inside(v.extent, Pobj)) return 0;
    // BBox::Offset, core/geometry.h:436-440
    float vx_ = (Pobj.x - v.extent.lo.x) / (v.extent.hi.x - v.extent.lo.x);
    float vy_ = (Pobj.y - v.extent.lo.y) / (v.extent.hi.y - v.extent.lo.y);
    float vz_ = (Pobj.z - v.extent.lo.z) / (v.extent.hi.z - v.extent.lo.z);
    vx_ = vx_ * v.nx - .5f;
    vy_ = vy_ * v.ny - .5f;
    vz_ = vz_ * v.nz - .5f;
    int vx = (int)floorf(vx_), vy = (int)floorf(vy_), vz = (int)floorf(vz_);
    float dx = vx_ - vx, dy = vy_ - vy, dz = vz_ - vz;
    float d00 = lerpf(dx, grid_D(v, vx, vy, vz), grid_D(v, vx + 1, vy, vz));
    float d10 = lerpf(dx, grid_D(v, vx, vy + 1, vz), grid_D(v, vx + 1, vy + 1, vz));
    float d01 = lerpf(dx, grid_D(v, vx, vy, vz + 1), grid_D(v, vx + 1, vy, vz + 1));
    float d11 = lerpf(dx, grid_D(v, vx, vy + 1, vz + 1), grid_D(v, vx + 1, vy + 1, vz + 1));
    float d0 = lerpf(dy, d00, d10);
    float d1 = lerpf(dy, d01, d11);
    return lerpf(dz, d0, d1);
}

struct VolCounters { uint64_t density_evals; };

// sigma_a / sigma_s / sigma_t / Lve: homogeneous.h:64-75 (Inside test), volume.h:81-92 (density *)
inline Spec vol_sigma_a(const Volume &v, V3 p, VolCounters *vc = 0) {
    if (v.kind == PVOL_VOLUME_GRID) return grid_density(v, xform_point(v.w2v, p), vc ? &vc->density_evals : 0) * v.sig_a;
    return box_inside(v.extent, xform_point(v.w2v, p)) ? v.sig_a : spec_const(0.f);
}
inline Spec vol_sigma_s(const Volume &v, V3 p, VolCounters *vc = 0) {
    if (v.kind == PVOL_VOLUME_GRID) return grid_density(v, xform_point(v.w2v, p), vc ? &vc->density_evals : 0) * v.sig_s;
    return box_inside(v.extent, xform_point(v.w2v, p)) ? v.sig_s : spec_const(0.f);
}
inline Spec vol_sigma_t(const Volume &v, V3 p, VolCounters *vc = 0) {
    if (v.kind == PVOL_VOLUME_GRID) return grid_density(v, xform_point(v.w2v, p), vc ? &vc->density_evals : 0) * (v.sig_a + v.sig_s);
    return box_inside(v.extent, xform_point(v.w2v, p)) ? (v.sig_a + v.sig_s) : spec_const(0.f);
}
inline Spec vol_lve(const Volume &v, V3 p, VolCounters *vc = 0) {
    if (v.kind == PVOL_VOLUME_GRID) return grid_density(v, xform_point(v.w2v, p), vc ? &vc->density_evals : 0) * v.le;
    return box_inside(v.extent, xform_point(v.w2v, p)) ? v.le : spec_const(0.f);
}
// p(): homogeneous.h:76-79 tests Inside; DensityRegion::p (volume.h:93-95) does not.
inline float vol_phase(const Volume &v, V3 p, V3 wi, V3 wo) {
    if (v.kind != PVOL_VOLUME_GRID && !box_inside(v.extent, xform_point(v.w2v, p))) return 0.f;
    return phase_hg(wi, wo, v.g);
}
// tau(): homogeneous.h:80-84 (analytic) / DensityRegion::tau core/volume.cpp:296-310 (stepped)
inline Spec vol_tau(const Volume &v, const Ray &r, float stepSize, float u, VolCounters *vc = 0) {
    if (v.kind != PVOL_VOLUME_GRID) {
        float t0, t1;
        if (!vol_intersect(v, r, &t0, &t1)) return spec_const(0.f);
        return length(ray_at(r, t0) - ray_at(r, t1)) * (v.sig_a + v.sig_s);
    }
    float t0, t1;
    float len = length(r.d);
    if (len == 0.f) return spec_const(0.f);
    Ray rn = make_ray(r.o, r.d / len, r.mint * len, r.maxt * len, r.time);
    if (!vol_intersect(v, rn, &t0, &t1)) return spec_const(0.f);
    Spec tau = spec_const(0.f);
    t0 += u * stepSize;
    while (t0 < t1) {
        tau += vol_sigma_t(v, ray_at(rn, t0), vc);
        t0 += stepSize;
    }
    return tau * stepSize;
}

// ------------------------------------------------------------------ triangles
// shapes/trianglemesh.cpp:127-160 (closest hit) / :211-243 (any hit): identical t test.
inline bool tri_hit(const Triangle &tr, const Ray &ray, float *tHit, float *b1o, float *b2o) {
    V3 e1 = tr.p2 - tr.p1;
    V3 e2 = tr.p3 - tr.p1;
    V3 s1 = cross(ray.d, e2);
    float divisor = dot(s1, e1);
    if (divisor == 0.f) return false;
    float invDivisor = 1.f / divisor;
    V3 s = ray.o - tr.p1;
    float b1 = dot(s, s1) * invDivisor;
    if (b1 < 0.f || b1 > 1.f) return false;
    V3 s2 = cross(s, e1);
    float b2 = dot(ray.d, s2) * invDivisor;
    if (b2 < 0.f || b1 + b2 > 1.f) return false;
    float t = dot(e2, s2) * invDivisor;
    if (t < ray.mint || t > ray.maxt) return false;
    *tHit = t;
    if (b1o) *b1o = b1;
    if (b2o) *b2o = b2;
    return true;
}

// ------------------------------------------------------------------ spheres
// Sphere::Intersect / IntersectP (shapes/sphere.cpp:59-104, 160-216): the ray goes to object space (core/transform.h:260-269),
// Quadratic (core/pbrt.h:309-323), nearest root inside [mint, maxt], z / phi clipping with the second root as fallback.
inline float sphere_phi(const Sphere &sp, V3 *phit) {
    if (phit->x == 0.f && phit->y == 0.f) phit->x = 1e-5f * sp.radius;
    float phi = atan2f(phit->y, phit->x);
    if (phi < 0.) phi += 2.f * kPi;
    return phi;
}
inline bool sphere_clipped(const Sphere &sp, V3 phit, float phi) {
    return (sp.zmin > -sp.radius && phit.z < sp.zmin) || (sp.zmax < sp.radius && phit.z > sp.zmax) || phi > sp.phiMax;
}
inline bool sphere_hit(const Sphere &sp, const Ray &r, float *tHit, V3 *phitOut) {
    V3 o = xform_point(sp.w2o, r.o), d = xform_vector(sp.w2o, r.d);
    float A = d.x * d.x + d.y * d.y + d.z * d.z;
    float B = 2 * (d.x * o.x + d.y * o.y + d.z * o.z);
    float C = o.x * o.x + o.y * o.y + o.z * o.z - sp.radius * sp.radius;
    float discrim = B * B - 4.f * A * C;
    if (discrim < 0.) return false;
    float rootDiscrim = sqrtf(discrim);
    float q;
    if (B < 0) q = -.5f * (B - rootDiscrim);
    else q = -.5f * (B + rootDiscrim);
    float t0 = q / A, t1 = C / q;
    if (t0 > t1) std::swap(t0, t1);
    if (t0 > r.maxt || t1 < r.mint) return false;
    float thit = t0;
    if (t0 < r.mint) {
        thit = t1;
        if (thit > r.maxt) return false;
    }
    V3 phit = o + d * thit;
    float phi = sphere_phi(sp, &phit);
    if (sphere_clipped(sp, phit, phi)) {
        if (thit == t1) return false;
        if (t1 > r.maxt) return false;
        thit = t1;
        phit = o + d * thit;
        phi = sphere_phi(sp, &phit);
        if (sphere_clipped(sp, phit, phi)) return false;
    }
    *tHit = thit;
    *phitOut = phit;
    return true;
}

inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }   // Clamp, core/pbrt.h:223-227
// material of a primitive: index >= 0 is a triangle, -1 - i the i-th sphere
inline int prim_material(const Scene &sc, int prim) { return prim >= 0 ? sc.tris[prim].material : sc.spheres[-1 - prim].material; }

struct Hit {
    int tri;
    float t, rayEpsilon;
    V3 p;           // dg.p = ray(t)
    V3 dpdu, dpdv;  // trianglemesh.cpp:163-181 with the default uvs (0,0),(1,0),(1,1)
    V3 nn;          // core/diffgeom.cpp:46-54
};

// Scene::Intersect (core/scene.h:50-56) -> closest hit; sets ray->maxt like the BVH does
// (accelerators/bvh.cpp:585-637 via GeometricPrimitive::Intersect core/primitive.cpp:97-110).
// Order of the linear scan only matters for exact ties in t.
inline bool scene_intersect(const Scene &sc, Ray *ray, Hit *hit) {
    bool any = false;
    for (size_t i = 0; i < sc.tris.size(); ++i) {
        float t;
        if (!tri_hit(sc.tris[i], *ray, &t, 0, 0)) continue;
        any = true;
        ray->maxt = t;
        hit->tri = (int)i;
        hit->t = t;
    }
    // spheres after the triangles (GeometricPrimitive::Intersect shortens the ray after every hit, core/primitive.cpp:97-110)
    int sph = -1;
    V3 phit = v3(0.f, 0.f, 0.f);
    for (size_t i = 0; i < sc.spheres.size(); ++i) {
        float t;
        V3 ph;
        if (!sphere_hit(sc.spheres[i], *ray, &t, &ph)) continue;
        any = true;
        ray->maxt = t;
        sph = (int)i;
        phit = ph;
        hit->tri = -1 - (int)i;
        hit->t = t;
    }
    if (!any) return false;
    if (sph >= 0) {   // sphere.cpp:106-155: parametric derivatives in object space, taken to world space; diffgeom.cpp:46-54
        const Sphere &sp = sc.spheres[sph];
        float theta = acosf(clampf(phit.z / sp.radius, -1.f, 1.f));
        float zradius = sqrtf(phit.x * phit.x + phit.y * phit.y);
        float invzradius = 1.f / zradius;
        float cosphi = phit.x * invzradius;
        float sinphi = phit.y * invzradius;
        V3 dpdu = v3(-sp.phiMax * phit.y, sp.phiMax * phit.x, 0);
        V3 dpdv = (sp.thetaMax - sp.thetaMin) * v3(phit.z * cosphi, phit.z * sinphi, -sp.radius * sinf(theta));
        hit->p = xform_point(sp.o2w, phit);
        hit->dpdu = xform_vector(sp.o2w, dpdu);
        hit->dpdv = xform_vector(sp.o2w, dpdv);
        hit->rayEpsilon = 5e-4f * hit->t;
        hit->nn = normalize(cross(hit->dpdu, hit->dpdv));
        if (sp.flip) hit->nn = hit->nn * -1.f;
        return true;
    }
    const Triangle &tr = sc.tris[hit->tri];
    // trianglemesh.cpp:163-181: uvs = {(0,0),(1,0),(1,1)} (trianglemesh.h:86-100)
    float du1 = 0.f - 1.f, du2 = 1.f - 1.f, dv1 = 0.f - 1.f, dv2 = 0.f - 1.f;
    V3 dp1 = tr.p1 - tr.p3, dp2 = tr.p2 - tr.p3;
    float determinant = du1 * dv2 - dv1 * du2;
    float invdet = 1.f / determinant;
    hit->dpdu = (dv2 * dp1 - dv1 * dp2) * invdet;
    hit->dpdv = (-du2 * dp1 + du1 * dp2) * invdet;
    hit->p = ray_at(*ray, hit->t);
    hit->rayEpsilon = 1e-3f * hit->t;
    hit->nn = normalize(cross(hit->dpdu, hit->dpdv));
    if (tr.flip) hit->nn = hit->nn * -1.f;
    return true;
}
// Scene::IntersectP (core/scene.h:57-61)
inline bool scene_intersect_p(const Scene &sc, const Ray &ray) {
    for (size_t i = 0; i < sc.tris.size(); ++i) {
        float t;
        if (tri_hit(sc.tris[i], ray, &t, 0, 0)) return true;
    }
    for (size_t i = 0; i < sc.spheres.size(); ++i) {
        float t;
        V3 ph;
        if (sphere_hit(sc.spheres[i], ray, &t, &ph)) return true;
    }
    return false;
}

// ------------------------------------------------------------------ lights
// lights/spot.cpp:60-69
inline float spot_falloff(const Light &l, V3 w) {
    V3 wl = normalize(xform_vector(l.w2l, w));
    float costheta = wl.z;
    if (costheta < l.cosTotalWidth) return 0.f;
    if (costheta > l.cosFalloffStart) return 1.f;
    float delta = (costheta - l.cosTotalWidth) / (l.cosFalloffStart - l.cosTotalWidth);
    return delta * delta * delta * delta;
}

// Light::Sample_L(p, pEpsilon, ls, time, &wi, &pdf, &vis): spot.cpp:50-57, point.cpp:50-57,
// distant.cpp:48-55.  `vis` is the shadow ray (core/light.h:87-95).
inline Spec light_sample_L(const Light &l, V3 p, float pEpsilon, float time, V3 *wi, float *pdf, Ray *vis) {
    if (l.kind == PVOL_LIGHT_DISTANT) {
        *wi = l.dir;
        *pdf = 1.f;
        *vis = make_ray(p, *wi, pEpsilon, kInfinity, time);
        return l.intensity;
    }
    *wi = normalize(l.pos - p);
    *pdf = 1.f;
    float dist = length(p - l.pos);                       // Distance(p1, p2), light.h:88
    *vis = make_ray(p, (l.pos - p) / dist, pEpsilon, dist * (1.f - 0.f), time);
    float d2 = length_sq(l.pos - p);                      // DistanceSquared(lightPos, p)
    if (l.kind == PVOL_LIGHT_SPOT) return l.intensity * spot_falloff(l, -*wi) / d2;
    return l.intensity / d2;
}

// Light::Power: spot.cpp:72-75, point.cpp:60-62, distant.cpp:58-63
inline Spec light_power(const Scene &sc, const Light &l) {
    if (l.kind == PVOL_LIGHT_SPOT) return l.intensity * 2.f * kPi * (1.f - .5f * (l.cosFalloffStart + l.cosTotalWidth));
    if (l.kind == PVOL_LIGHT_POINT) return 4.f * kPi * l.intensity;
    V3 c; float rad;
    box_bounding_sphere(sc.world, &c, &rad);
    return l.intensity * kPi * rad * rad;
}

// Light::Sample_L(scene, ls, u1, u2, time, &ray, &Ns, &pdf): spot.cpp:106-114, point.cpp:80-88,
// distant.cpp:82-102.  Only ls.uPos[0..1] are read by these three lights.
inline Spec light_sample_emit(const Scene &sc, const Light &l, float uPos0, float uPos1, float time, Ray *ray, V3 *Ns, float *pdf) {
    if (l.kind == PVOL_LIGHT_SPOT) {
        V3 v = uniform_sample_cone(uPos0, uPos1, l.cosTotalWidth);
        *ray = make_ray(l.pos, xform_vector(l.l2w, v), 0.f, kInfinity, time);
        *Ns = ray->d;
        *pdf = 1.f / (2.f * kPi * (1.f - l.cosTotalWidth));  // UniformConePdf, montecarlo.cpp:400-402
        return l.intensity * spot_falloff(l, ray->d);
    }
    if (l.kind == PVOL_LIGHT_POINT) {
        *ray = make_ray(l.pos, uniform_sample_sphere(uPos0, uPos1), 0.f, kInfinity, time);
        *Ns = ray->d;
        *pdf = 1.f / (4.f * kPi);                            // UniformSpherePdf, montecarlo.cpp:293-295
        return l.intensity;
    }
    V3 worldCenter; float worldRadius;
    box_bounding_sphere(sc.world, &worldCenter, &worldRadius);
    V3 v1, v2;
    coordinate_system(l.dir, &v1, &v2);
    float d1, d2;
    concentric_sample_disk(uPos0, uPos1, &d1, &d2);
    V3 Pdisk = worldCenter + worldRadius * (d1 * v1 + d2 * v2);
    *ray = make_ray(Pdisk + worldRadius * l.dir, -l.dir, 0.f, kInfinity, time);
    *Ns = ray->d;
    *pdf = 1.f / (kPi * worldRadius * worldRadius);
    return l.intensity;
}

}  // namespace orc
#endif
