// integration/hip_flatten.h -- reference-side half of the drop-in: turns a live pbrt `Scene` into the POD
// `pvol_scene` of include/pvol.h.  Included by integration/hip_photonvolume.cpp (the plugin) and by the test
// tools under oracle/ that link it against the reference's own objects (oracle/ref_capture.cpp `shimscene`,
// oracle/shim_drive.cpp), so the flattening that ships is the flattening that is tested.
//
// Expects the reference headers to be included already WITH `private`/`protected` opened (a real patch adds
// `friend` declarations instead): scene.h, primitive.h, shape.h, light.h, volume.h, accelerators/{bvh,grid,
// kdtreeaccel}.h, shapes/{trianglemesh,sphere}.h, lights/{distant,point,spot}.h, materials/{glass,matte}.h,
// volumes/{homogeneous,rainbow,volumegrid}.h, textures/constant.h.
//
// ORDER.  The device answers closest-hit ties (two triangles at the same fp32 t) with the triangle LATEST in
// pvol_scene order, which is what a linear scan over the scene's shapes in creation order returns
// (shapes/trianglemesh.cpp:127-160 rejects `t > ray.maxt` only).  The reference's BVHAccel keeps its refined
// primitives in *leaf* order (accelerators/bvh.cpp:226-237 swaps in `orderedPrims`), so the walk below restores
// creation order through Shape::shapeId (core/shape.cpp:45: a counter bumped by every Shape constructor;
// TriangleMesh::Refine creates a mesh's triangles in index order, shapes/trianglemesh.cpp:100-105).  The
// flattened arrays are then exactly those of a scene flattened before the accelerator was built.
#ifndef HIP_FLATTEN_H
#define HIP_FLATTEN_H

#include <algorithm>
#include <map>
#include <vector>
#include <string.h>
#include "pvol.h"

struct HipFlatScene {
    pvol_scene scene;   // pointers aim into the vectors below
    std::vector<pvol_light> lights;
    std::vector<pvol_triangle> tris;
    std::vector<pvol_sphere> spheres;
    std::vector<pvol_material> mats;
};

namespace hipflat {

inline void putSpec(pvol_spectrum *d, const Spectrum &s) { for (int i = 0; i < PVOL_NBINS; ++i) d->c[i] = s.c[i]; }
inline void putMat(float *d, const Matrix4x4 &m) { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) d[4 * r + c] = m.m[r][c]; }

// The intersectable primitives below `agg`.  Primitive::FullyRefine (core/primitive.cpp:55-67) stops at anything whose
// CanIntersect() is true -- and every accelerator's is (accelerators/bvh.h:59) -- so on the scene's aggregate it returns the
// aggregate itself; the accelerators' own fully refined lists are read instead.
inline const char *collectLeaves(const Primitive *agg, std::vector<Reference<Primitive> > &leaves) {
    if (!agg) return NULL;
    if (const BVHAccel *bvh = dynamic_cast<const BVHAccel *>(agg)) { leaves = bvh->primitives; return NULL; }
    if (const GridAccel *grid = dynamic_cast<const GridAccel *>(agg)) { leaves = grid->primitives; return NULL; }
    if (const KdTreeAccel *kd = dynamic_cast<const KdTreeAccel *>(agg)) { leaves = kd->primitives; return NULL; }
    if (!agg->CanIntersect()) { agg->FullyRefine(leaves); return NULL; }
    if (dynamic_cast<const GeometricPrimitive *>(agg)) { leaves.push_back(const_cast<Primitive *>(agg)); return NULL; }
    return "the scene's aggregate is not a BVHAccel / GridAccel / KdTreeAccel";
}

struct ByShapeId {
    bool operator()(const Reference<Primitive> &a, const Reference<Primitive> &b) const {
        return static_cast<const GeometricPrimitive *>(a.GetPtr())->shape->shapeId <
               static_cast<const GeometricPrimitive *>(b.GetPtr())->shape->shapeId;
    }
};

template <typename T> inline bool isConstant(const Reference<Texture<T> > &t) {
    return dynamic_cast<const ConstantTexture<T> *>(t.GetPtr()) != NULL;
}

inline const char *flattenVolume(const VolumeRegion *vr, pvol_volume *v) {
    memset(v, 0, sizeof(*v));
    if (!vr) { v->kind = PVOL_VOLUME_NONE; return NULL; }
    const BBox *e = NULL;
    const Transform *w2v = NULL;
    if (const VolumeGridDensity *g = dynamic_cast<const VolumeGridDensity *>(vr)) {
        v->kind = PVOL_VOLUME_GRID; e = &g->extent; w2v = &g->WorldToVolume;
        putSpec(&v->sigma_a, g->sig_a); putSpec(&v->sigma_s, g->sig_s); putSpec(&v->le, g->le); v->g = g->g;
        v->nx = g->nx; v->ny = g->ny; v->nz = g->nz; v->density = g->density;
    } else if (const HomogeneousVolumeDensity *h = dynamic_cast<const HomogeneousVolumeDensity *>(vr)) {
        v->kind = dynamic_cast<const RainbowVolume *>(vr) ? PVOL_VOLUME_RAINBOW : PVOL_VOLUME_HOMOGENEOUS;
        e = &h->extent; w2v = &h->WorldToVolume;
        putSpec(&v->sigma_a, h->sig_a); putSpec(&v->sigma_s, h->sig_s); putSpec(&v->le, h->le); v->g = h->g;
    } else {
        return "volume kind not supported (homogeneous, rainbow, volumegrid)";
    }
    v->extent_min[0] = e->pMin.x; v->extent_min[1] = e->pMin.y; v->extent_min[2] = e->pMin.z;
    v->extent_max[0] = e->pMax.x; v->extent_max[1] = e->pMax.y; v->extent_max[2] = e->pMax.z;
    putMat(v->world_to_volume, w2v->m);
    putMat(v->volume_to_world, w2v->mInv);
    return NULL;
}

inline const char *flattenLight(const Light *L, pvol_light *l) {
    memset(l, 0, sizeof(*l));
    putMat(l->light_to_world, L->LightToWorld.m);
    putMat(l->world_to_light, L->WorldToLight.m);
    if (const DistantLight *d = dynamic_cast<const DistantLight *>(L)) {
        l->kind = PVOL_LIGHT_DISTANT; l->dir[0] = d->lightDir.x; l->dir[1] = d->lightDir.y; l->dir[2] = d->lightDir.z;
        putSpec(&l->intensity, d->L);
    } else if (const SpotLight *s = dynamic_cast<const SpotLight *>(L)) {
        l->kind = PVOL_LIGHT_SPOT; l->pos[0] = s->lightPos.x; l->pos[1] = s->lightPos.y; l->pos[2] = s->lightPos.z;
        putSpec(&l->intensity, s->Intensity); l->cos_total_width = s->cosTotalWidth; l->cos_falloff_start = s->cosFalloffStart;
    } else if (const PointLight *p = dynamic_cast<const PointLight *>(L)) {
        l->kind = PVOL_LIGHT_POINT; l->pos[0] = p->lightPos.x; l->pos[1] = p->lightPos.y; l->pos[2] = p->lightPos.z;
        putSpec(&l->intensity, p->Intensity);
    } else {
        return "light kind not supported (point, spot, distant)";
    }
    return NULL;
}

inline const char *flattenMaterial(const Material *m, pvol_material *o) {
    memset(o, 0, sizeof(*o));
    DifferentialGeometry dg;   // constant textures ignore it
    if (const MatteMaterial *mm = dynamic_cast<const MatteMaterial *>(m)) {
        if (!isConstant(mm->Kd) || !isConstant(mm->sigma) || mm->bumpMap) return "matte: only constant Kd / sigma without a bump map";
        if (mm->sigma->Evaluate(dg) != 0.f) return "matte: sigma != 0 (Oren-Nayar) is outside the photon path's BSDFs";
        o->kind = PVOL_MATERIAL_MATTE;
        o->ior = 1.f;
        putSpec(&o->kd, mm->Kd->Evaluate(dg).Clamp());   // materials/matte.cpp:55
    } else if (const GlassMaterial *gm = dynamic_cast<const GlassMaterial *>(m)) {
        if (!isConstant(gm->Kr) || !isConstant(gm->Kt) || !isConstant(gm->index) || gm->bumpMap) return "glass: only constant Kr / Kt / index without a bump map";
        o->kind = PVOL_MATERIAL_GLASS;
        putSpec(&o->kr, gm->Kr->Evaluate(dg).Clamp());   // materials/glass.cpp:50-52
        putSpec(&o->kt, gm->Kt->Evaluate(dg).Clamp());
        o->ior = gm->index->Evaluate(dg);
        o->vn = gm->Vn;
    } else {
        return "material not supported on the photon path (matte, glass)";
    }
    return NULL;
}

}  // namespace hipflat

// Fills *out from the live scene; returns NULL, or a static message naming what the path does not cover.
inline const char *HipFlattenScene(const Scene *scene, HipFlatScene *out) {
    using namespace hipflat;
    out->lights.clear(); out->tris.clear(); out->spheres.clear(); out->mats.clear();
    pvol_scene &s = out->scene;
    memset(&s, 0, sizeof(s));
    const char *err = flattenVolume(scene->volumeRegion, &s.volume);
    if (err) return err;
    for (size_t i = 0; i < scene->lights.size(); ++i) {
        pvol_light l;
        if ((err = flattenLight(scene->lights[i], &l)) != NULL) return err;
        out->lights.push_back(l);
    }
    std::vector<Reference<Primitive> > leaves;
    if ((err = collectLeaves(scene->aggregate, leaves)) != NULL) return err;
    for (size_t i = 0; i < leaves.size(); ++i)
        if (!dynamic_cast<const GeometricPrimitive *>(leaves[i].GetPtr()))
            return "only GeometricPrimitives (no object instances / animated transforms) are supported on the photon path";
    std::stable_sort(leaves.begin(), leaves.end(), ByShapeId());   // creation order == the scene file's order (see ORDER above)
    std::map<const Material *, int> matIndex;
    for (size_t i = 0; i < leaves.size(); ++i) {
        const GeometricPrimitive *gp = static_cast<const GeometricPrimitive *>(leaves[i].GetPtr());
        if (gp->areaLight) return "area lights are outside the photon path's light kinds";
        const Triangle *tri = dynamic_cast<const Triangle *>(gp->shape.GetPtr());
        const Sphere *sph = dynamic_cast<const Sphere *>(gp->shape.GetPtr());
        if (!tri && !sph) return "only triangle meshes and spheres are supported on the photon path";
        const Material *m = gp->material.GetPtr();
        if (!matIndex.count(m)) {
            pvol_material pm;
            if ((err = flattenMaterial(m, &pm)) != NULL) return err;
            matIndex[m] = (int)out->mats.size();
            out->mats.push_back(pm);
        }
        if (sph) {   // shapes/sphere.cpp:41-49: what the constructor stored
            pvol_sphere q;
            memset(&q, 0, sizeof(q));
            putMat(q.object_to_world, sph->ObjectToWorld->m);
            putMat(q.world_to_object, sph->WorldToObject->m);
            q.radius = sph->radius; q.z_min = sph->zmin; q.z_max = sph->zmax;
            q.theta_min = sph->thetaMin; q.theta_max = sph->thetaMax; q.phi_max = sph->phiMax;
            q.material = matIndex[m];
            q.flip_normal = (sph->ReverseOrientation ^ sph->TransformSwapsHandedness) ? 1 : 0;
            out->spheres.push_back(q);
            continue;
        }
        if (tri->mesh->alphaTexture) return "alpha-textured meshes are outside the photon path";
        pvol_triangle t;
        memset(&t, 0, sizeof(t));
        for (int k = 0; k < 3; ++k) {
            const Point &p = tri->mesh->p[tri->v[k]];   // world space (shapes/trianglemesh.cpp:70-71)
            t.p[k][0] = p.x; t.p[k][1] = p.y; t.p[k][2] = p.z;
        }
        t.material = matIndex[m];
        t.flip_normal = (tri->ReverseOrientation ^ tri->TransformSwapsHandedness) ? 1 : 0;
        out->tris.push_back(t);
    }
    s.n_lights = (uint32_t)out->lights.size(); s.lights = out->lights.empty() ? NULL : &out->lights[0];
    s.n_triangles = (uint32_t)out->tris.size(); s.triangles = out->tris.empty() ? NULL : &out->tris[0];
    s.n_materials = (uint32_t)out->mats.size(); s.materials = out->mats.empty() ? NULL : &out->mats[0];
    s.n_spheres = (uint32_t)out->spheres.size(); s.spheres = out->spheres.empty() ? NULL : &out->spheres[0];
    const BBox &wb = scene->WorldBound();   // geometry U volume (core/scene.cpp:59-60)
    s.world_min[0] = wb.pMin.x; s.world_min[1] = wb.pMin.y; s.world_min[2] = wb.pMin.z;
    s.world_max[0] = wb.pMax.x; s.world_max[1] = wb.pMax.y; s.world_max[2] = wb.pMax.z;
    putSpec(&s.cie_x, SampledSpectrum::X); putSpec(&s.cie_y, SampledSpectrum::Y); putSpec(&s.cie_z, SampledSpectrum::Z);
    s.xyz_scale = float(sampledLambdaEnd - sampledLambdaStart) / float(CIE_Y_integral * nSpectralSamples);
    return NULL;
}

#endif  // HIP_FLATTEN_H
