"""ctypes mirror of include/pvol.h (the C-ABI boundary) and helpers that turn a scene blob
(see blob.py) into the POD structs.  No compute happens here."""
import ctypes as C

import math

import numpy as np

NBINS = 30
MT_N = 624

PVOL_OK = 0
PVOL_E_INVALID = -1
PVOL_E_NO_DEVICE = -2
PVOL_E_NO_SCENE = -3
PVOL_E_NO_MEMORY = -4
PVOL_E_UNSUPPORTED = -5
PVOL_E_LIMIT = -6
PVOL_E_SHOOT_FAILED = -7

VOLUME_NONE, VOLUME_HOMOGENEOUS, VOLUME_GRID, VOLUME_RAINBOW = 0, 1, 2, 3
LIGHT_POINT, LIGHT_SPOT, LIGHT_DISTANT = 0, 1, 2
MATERIAL_MATTE, MATERIAL_GLASS = 0, 1
OUT_SPECTRAL, OUT_XYZ = 0, 1


class Spectrum(C.Structure):
    _fields_ = [("c", C.c_float * NBINS)]


class Volume(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("extent_min", C.c_float * 3), ("extent_max", C.c_float * 3),
        ("world_to_volume", C.c_float * 16), ("volume_to_world", C.c_float * 16),
        ("sigma_a", Spectrum), ("sigma_s", Spectrum), ("le", Spectrum),
        ("g", C.c_float),
        ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32),
        ("density", C.POINTER(C.c_float)),
    ]


class Light(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("pos", C.c_float * 3), ("dir", C.c_float * 3),
        ("light_to_world", C.c_float * 16), ("world_to_light", C.c_float * 16),
        ("intensity", Spectrum),
        ("cos_total_width", C.c_float), ("cos_falloff_start", C.c_float),
    ]


class Material(C.Structure):
    _fields_ = [("kind", C.c_int32), ("kd", Spectrum), ("kr", Spectrum), ("kt", Spectrum),
                ("ior", C.c_float), ("vn", C.c_float)]


class Triangle(C.Structure):
    _fields_ = [("p", (C.c_float * 3) * 3), ("material", C.c_int32), ("flip_normal", C.c_int32)]


class Sphere(C.Structure):
    _fields_ = [("object_to_world", C.c_float * 16), ("world_to_object", C.c_float * 16), ("radius", C.c_float),
                ("z_min", C.c_float), ("z_max", C.c_float), ("theta_min", C.c_float), ("theta_max", C.c_float), ("phi_max", C.c_float),
                ("material", C.c_int32), ("flip_normal", C.c_int32)]


class Scene(C.Structure):
    _fields_ = [
        ("volume", Volume),
        ("n_lights", C.c_uint32), ("lights", C.POINTER(Light)),
        ("n_triangles", C.c_uint32), ("triangles", C.POINTER(Triangle)),
        ("n_materials", C.c_uint32), ("materials", C.POINTER(Material)),
        ("world_min", C.c_float * 3), ("world_max", C.c_float * 3),
        ("cie_x", Spectrum), ("cie_y", Spectrum), ("cie_z", Spectrum),
        ("xyz_scale", C.c_float),
        ("n_spheres", C.c_uint32), ("spheres", C.POINTER(Sphere)),
    ]


class Params(C.Structure):
    _fields_ = [
        ("step_size", C.c_float), ("n_used", C.c_int32), ("max_dist", C.c_float),
        ("n_volume_photons", C.c_uint32), ("shooter_step_size", C.c_float),
        ("max_photon_depth", C.c_int32), ("n_caustic_photons", C.c_uint32),
        ("n_indirect_photons", C.c_uint32), ("final_gather", C.c_int32),
        ("device", C.c_int32), ("grid_cell_scale", C.c_float), ("keep_surface_photons", C.c_uint32), ("reserved", C.c_uint32 * 6),
    ]


class Stats(C.Structure):
    _fields_ = [("n_rays", C.c_uint64), ("n_steps", C.c_uint64), ("n_tested", C.c_uint64),
                ("n_kept", C.c_uint64), ("n_lookups_lt10", C.c_uint64),
                ("n_shadow_unoccluded", C.c_uint64), ("n_guess_retries", C.c_uint64),
                ("cy_search", C.c_uint64), ("cy_select", C.c_uint64), ("cy_flux", C.c_uint64),
                ("cy_total", C.c_uint64), ("group_guess_failed", C.c_uint64), ("group_plan_skipped", C.c_uint64),
                ("cy_fallback", C.c_uint64), ("group_deferred_overflow", C.c_uint64), ("group_deferred_too_few", C.c_uint64),
                ("group_attempts", C.c_uint64)]


# numpy views of the two array-of-struct inputs (sizes checked against pvol.h in the tests)
RAY_DTYPE = np.dtype([("o", "<f4", 3), ("mint", "<f4"), ("d", "<f4", 3), ("maxt", "<f4"),
                      ("time", "<f4"), ("scatter_u", "<f4"), ("rng_skip", "<u4"), ("flags", "<u4")])
STREAM_DTYPE = np.dtype([("seed", "<u4"), ("first_ray", "<u4"), ("n_rays", "<u4"), ("reserved", "<u4"),
                         ("start_draw", "<u8"), ("end_draw", "<u8")])
TRI_DTYPE = np.dtype([("p", "<f4", (3, 3)), ("material", "<i4"), ("flip_normal", "<i4")])
assert RAY_DTYPE.itemsize == 48 and STREAM_DTYPE.itemsize == 32 and TRI_DTYPE.itemsize == C.sizeof(Triangle)


def _spec(dst, src):
    for i in range(NBINS):
        dst.c[i] = float(src[i])


def _fill(dst, src):
    for i, v in enumerate(src):
        dst[i] = float(v)


class SceneHolder:
    """Owns the ctypes arrays a pvol_scene points into."""

    def __init__(self, b):
        s = Scene()
        v = s.volume
        v.kind = int(b["vol.kind"][0])
        _fill(v.extent_min, b["vol.extent"][:3])
        _fill(v.extent_max, b["vol.extent"][3:])
        _fill(v.world_to_volume, b["vol.w2v"])
        _fill(v.volume_to_world, b["vol.v2w"])
        _spec(v.sigma_a, b["vol.sigma_a"])
        _spec(v.sigma_s, b["vol.sigma_s"])
        _spec(v.le, b["vol.le"])
        v.g = float(b["vol.g"][0])
        v.nx, v.ny, v.nz = [int(x) for x in b["vol.dims"]]
        self.density = None
        if v.kind == VOLUME_GRID:
            self.density = np.ascontiguousarray(b["vol.density"], dtype=np.float32)
            assert self.density.size == v.nx * v.ny * v.nz
            v.density = self.density.ctypes.data_as(C.POINTER(C.c_float))
        nl = len(b["lights.kind"])
        self.lights = (Light * max(nl, 1))()
        for i in range(nl):
            L = self.lights[i]
            L.kind = int(b["lights.kind"][i])
            _fill(L.pos, b["lights.pos"][3 * i:3 * i + 3])
            _fill(L.dir, b["lights.dir"][3 * i:3 * i + 3])
            _fill(L.light_to_world, b["lights.l2w"][16 * i:16 * i + 16])
            _fill(L.world_to_light, b["lights.w2l"][16 * i:16 * i + 16])
            _spec(L.intensity, b["lights.intensity"][NBINS * i:NBINS * (i + 1)])
            L.cos_total_width = float(b["lights.cos"][2 * i])
            L.cos_falloff_start = float(b["lights.cos"][2 * i + 1])
        s.n_lights = nl
        s.lights = C.cast(self.lights, C.POINTER(Light))
        nt = len(b["tris.material"])
        self.tris = np.zeros(max(nt, 1), TRI_DTYPE)   # a million triangles are one array copy, not nine million assignments
        self.tris["p"][:nt] = np.asarray(b["tris.p"], np.float32).reshape(-1, 3, 3)
        self.tris["material"][:nt] = b["tris.material"]
        self.tris["flip_normal"][:nt] = b["tris.flip"]
        s.n_triangles = nt
        s.triangles = self.tris.ctypes.data_as(C.POINTER(Triangle))
        nm = len(b["mats.kind"])
        self.mats = (Material * max(nm, 1))()
        for i in range(nm):
            M = self.mats[i]
            M.kind = int(b["mats.kind"][i])
            _spec(M.kd, b["mats.kd"][NBINS * i:NBINS * (i + 1)])
            _spec(M.kr, b["mats.kr"][NBINS * i:NBINS * (i + 1)])
            _spec(M.kt, b["mats.kt"][NBINS * i:NBINS * (i + 1)])
            M.ior = float(b["mats.ior"][i])
            M.vn = float(b["mats.vn"][i])
        s.n_materials = nm
        s.materials = C.cast(self.mats, C.POINTER(Material))
        _fill(s.world_min, b["world"][:3])
        _fill(s.world_max, b["world"][3:])
        _spec(s.cie_x, b["cie.x"])
        _spec(s.cie_y, b["cie.y"])
        _spec(s.cie_z, b["cie.z"])
        s.xyz_scale = float(b["xyz_scale"][0])
        # Shape "sphere" (optional keys): o2w / w2o 16 floats each, f = radius, z_min, z_max, theta_min, theta_max, phi_max
        ns = len(b["spheres.material"]) if "spheres.material" in b else 0
        self.spheres = (Sphere * max(ns, 1))()
        for i in range(ns):
            P = self.spheres[i]
            _fill(P.object_to_world, b["spheres.o2w"][16 * i:16 * i + 16])
            _fill(P.world_to_object, b["spheres.w2o"][16 * i:16 * i + 16])
            f = b["spheres.f"][6 * i:6 * i + 6]
            P.radius, P.z_min, P.z_max, P.theta_min, P.theta_max, P.phi_max = [float(x) for x in f]
            P.material = int(b["spheres.material"][i])
            P.flip_normal = int(b["spheres.flip"][i])
        s.n_spheres = ns
        s.spheres = C.cast(self.spheres, C.POINTER(Sphere))
        self.scene = s
        self.blob = b


def params_from_blob(b, **over):
    """pvol_params from a scene blob's params.f / params.i, with keyword overrides."""
    p = Params()
    p.step_size, p.max_dist, p.shooter_step_size = [float(x) for x in b["params.f"]]
    (p.n_used, p.n_volume_photons, p.max_photon_depth, p.n_caustic_photons,
     p.n_indirect_photons, p.final_gather) = [int(x) for x in b["params.i"]]
    p.device = 0
    p.grid_cell_scale = 0.0
    for k, v in over.items():
        setattr(p, k, v)
    return p


def make_rays(o, d, mint, maxt, scatter_u, time=0.0, rng_skip=0):
    n = len(o)
    r = np.zeros(n, dtype=RAY_DTYPE)
    r["o"] = o
    r["d"] = d
    r["mint"] = mint
    r["maxt"] = maxt
    r["time"] = time
    r["scatter_u"] = scatter_u
    r["rng_skip"] = rng_skip
    return r


def make_streams(seeds, counts, start_draw=0):
    s = np.zeros(len(seeds), dtype=STREAM_DTYPE)
    s["seed"] = seeds
    s["n_rays"] = counts
    first = np.concatenate([[0], np.cumsum(counts)[:-1]]) if len(counts) else []
    s["first_ray"] = first
    s["start_draw"] = start_draw
    return s


# ---- tile driver (include/pvol.h: pvol_camera, pvol_film, pvol_sampler, pvol_render_debug)
MAX_SAMPLE_ARRAYS = 16
FILTER_TABLE_SIZE = 16


class Camera(C.Structure):
    _fields_ = [("raster_to_camera", C.c_float * 16), ("camera_to_world", C.c_float * 16),
                ("shutter_open", C.c_float), ("shutter_close", C.c_float),
                ("lens_radius", C.c_float), ("focal_distance", C.c_float)]


class Film(C.Structure):
    _fields_ = [("x_resolution", C.c_int32), ("y_resolution", C.c_int32),
                ("filter_xwidth", C.c_float), ("filter_ywidth", C.c_float),
                ("filter_table", C.c_float * (FILTER_TABLE_SIZE * FILTER_TABLE_SIZE))]


class Sampler(C.Structure):
    _fields_ = [("x_start", C.c_int32), ("x_end", C.c_int32), ("y_start", C.c_int32), ("y_end", C.c_int32),
                ("pixel_samples", C.c_uint32), ("n_tasks", C.c_uint32),
                ("n1d_count", C.c_uint32), ("n2d_count", C.c_uint32),
                ("n1d", C.c_uint32 * MAX_SAMPLE_ARRAYS), ("n2d", C.c_uint32 * MAX_SAMPLE_ARRAYS),
                ("tau_index", C.c_uint32), ("scatter_index", C.c_uint32)]


class RenderDebug(C.Structure):
    _fields_ = [("d_rays", C.c_void_p), ("d_image_xy", C.c_void_p), ("d_xyz", C.c_void_p), ("d_streams", C.c_void_p),
                ("d_surf_xyz", C.c_void_p)]


class SurfaceParams(C.Structure):   # pvol_surface_params
    _fields_ = [("n_used", C.c_int32), ("max_dist", C.c_float), ("max_specular_depth", C.c_int32), ("final_gather", C.c_int32),
                ("n_caustic_paths", C.c_uint32), ("use_preprocess_store", C.c_int32), ("n_indirect_photons", C.c_uint32),
                ("reserved", C.c_uint32 * 1)]


def make_camera(raster_to_camera, camera_to_world, shutter_open=0.0, shutter_close=1.0, lens_radius=0.0, focal_distance=1e30):
    c = Camera()
    for i, v in enumerate(np.asarray(raster_to_camera, np.float32).reshape(16)):
        c.raster_to_camera[i] = float(v)
    for i, v in enumerate(np.asarray(camera_to_world, np.float32).reshape(16)):
        c.camera_to_world[i] = float(v)
    c.shutter_open, c.shutter_close, c.lens_radius, c.focal_distance = shutter_open, shutter_close, lens_radius, focal_distance
    return c


def make_film(xres, yres, filter_table, xwidth=2.0, ywidth=2.0):
    f = Film()
    f.x_resolution, f.y_resolution, f.filter_xwidth, f.filter_ywidth = int(xres), int(yres), xwidth, ywidth
    for i, v in enumerate(np.asarray(filter_table, np.float32).reshape(256)):
        f.filter_table[i] = float(v)
    return f


def make_sampler(xres, yres, spp, n_tasks, xwidth=2.0, ywidth=2.0, n1d=(1, 1), n2d=(), tau_index=0, scatter_index=1):
    """LDSampler over Film::GetSampleExtent (film/image.cpp:157-166) with the Sample layout
    PhotonVolumeIntegrator::RequestSamples leaves when it is the only requester (photonvolume.cpp:9-13)."""
    s = Sampler()
    s.x_start = int(math.floor(0 + 0.5 - xwidth))
    s.x_end = int(math.ceil(0 + 0.5 + xres + xwidth))
    s.y_start = int(math.floor(0 + 0.5 - ywidth))
    s.y_end = int(math.ceil(0 + 0.5 + yres + ywidth))
    s.pixel_samples, s.n_tasks = int(spp), int(n_tasks)
    s.n1d_count, s.n2d_count = len(n1d), len(n2d)
    for i, v in enumerate(n1d):
        s.n1d[i] = v
    for i, v in enumerate(n2d):
        s.n2d[i] = v
    s.tau_index, s.scatter_index = tau_index, scatter_index
    return s


def perspective_camera(fov, xres, yres, camera_to_world):
    """RasterToCamera of PerspectiveCamera (core/camera.cpp:83-102, core/transform.cpp:313-323) in fp32, for
    bench/tests that have no reference camera object at hand; the shim copies the reference's own matrices."""
    f32 = np.float32
    frame = f32(xres) / f32(yres)
    sw = [-frame, frame, f32(-1), f32(1)] if frame > 1 else [f32(-1), f32(1), f32(-1) / frame, f32(1) / frame]
    n, f = f32(1e-2), f32(1000.0)
    persp = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, f / (f - n), -f * n / (f - n)], [0, 0, 1, 0]], np.float32)
    inv_tan = f32(1) / f32(math.tan(math.radians(fov) / 2))
    cam_to_screen = (np.diag(np.array([inv_tan, inv_tan, 1, 1], np.float32)) @ persp).astype(np.float32)

    def scale(x, y, z):
        return np.diag(np.array([x, y, z, 1], np.float32))

    def translate(x, y, z):
        m = np.eye(4, dtype=np.float32)
        m[:3, 3] = [x, y, z]
        return m
    screen_to_raster = (scale(f32(xres), f32(yres), 1) @ scale(f32(1) / (sw[1] - sw[0]), f32(1) / (sw[2] - sw[3]), 1)
                        @ translate(-sw[0], -sw[3], 0)).astype(np.float32)
    raster_to_screen = np.linalg.inv(screen_to_raster.astype(np.float64))
    raster_to_camera = (np.linalg.inv(cam_to_screen.astype(np.float64)) @ raster_to_screen).astype(np.float32)
    return make_camera(raster_to_camera, camera_to_world)
