"""ctypes binding of libpvol.so, the C ABI declared in include/pvol.h.

This is plumbing only: every compute call goes to the HIP library, and loading fails loudly when
the library has not been built (there is no Python or CPU fallback).
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PVOL_LIB") or os.path.join(_HERE, "libpvol.so")  # PVOL_LIB: A/B builds of the same ABI

_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)


class PvolError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        RuntimeError.__init__(self, "%s: %s (%d)" % (where, lib().pvol_strerror(status).decode(), status))


_lib = None


def lib():
    """Load libpvol.so (built by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); there is no fallback path" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.pvol_abi_version.restype = C.c_int
        L.pvol_strerror.restype = C.c_char_p
        L.pvol_strerror.argtypes = [C.c_int]
        L.pvol_device_count.restype = C.c_int
        L.pvol_default_params.argtypes = [C.POINTER(abi.Params)]
        L.pvol_default_params.restype = None
        L.pvol_create.argtypes = [C.POINTER(abi.Params), C.POINTER(C.c_void_p)]
        L.pvol_destroy.argtypes = [C.c_void_p]
        L.pvol_destroy.restype = None
        L.pvol_set_scene.argtypes = [C.c_void_p, C.POINTER(abi.Scene)]
        L.pvol_upload_photons.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, C.c_uint32]
        L.pvol_preprocess.argtypes = [C.c_void_p, C.c_uint32]
        L.pvol_preprocess_blocks.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.pvol_photon_count.argtypes = [C.c_void_p, _u32p]
        L.pvol_download_photons.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, C.c_uint32]
        L.pvol_li_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, _f32p, _u32p]
        L.pvol_li_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p]
        L.pvol_li.argtypes = [C.c_void_p, C.c_void_p, _u32p, C.POINTER(C.c_int32), _f32p, _f32p]
        L.pvol_transmittance_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, _f32p]
        L.pvol_get_stats.argtypes = [C.c_void_p, C.POINTER(abi.Stats), C.c_int]
        L.pvol_enable_stats.argtypes = [C.c_void_p, C.c_int]
        L.pvol_kernel_time_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]
        L.pvol_get_shoot_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.pvol_march_kernel_name.argtypes = [C.c_void_p]
        L.pvol_check_errors.argtypes = [C.c_void_p]
        L.pvol_get_preprocess_seconds.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.pvol_get_accel_info.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.pvol_surface_photon_count.argtypes = [C.c_void_p, C.c_int, _u32p, _u32p]
        L.pvol_download_surface_photons.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p, _f32p, C.c_uint32]
        L.pvol_radiance_photon_count.argtypes = [C.c_void_p, _u32p]
        L.pvol_download_radiance_photons.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, _f32p, C.c_uint32]
        L.pvol_set_surface_integrator.argtypes = [C.c_void_p, C.POINTER(abi.SurfaceParams), _f32p, _f32p, _f32p, C.c_uint32]
        L.pvol_march_kernel_name.restype = C.c_char_p
        L.pvol_gaussian_filter_table.argtypes = [C.c_float, C.c_float, C.c_float, _f32p]
        L.pvol_gaussian_filter_table.restype = None
        L.pvol_compute_sub_window.argtypes = [C.POINTER(abi.Sampler), C.c_uint32, C.POINTER(C.c_int32)]
        L.pvol_compute_sub_window.restype = None
        L.pvol_render_sample_count.argtypes = [C.POINTER(abi.Sampler), _u32p, C.c_uint32]
        L.pvol_render_sample_count.restype = C.c_uint64
        L.pvol_render_tasks_device.argtypes = [C.c_void_p, C.POINTER(abi.Camera), C.POINTER(abi.Film), C.POINTER(abi.Sampler), _u32p, C.c_uint32,
                                               C.c_void_p, C.POINTER(abi.RenderDebug), C.c_void_p]
        L.pvol_film_add_samples_device.argtypes = [C.c_void_p, C.POINTER(abi.Film), C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]
        L.pvol_film_resolve_device.argtypes = [C.c_void_p, C.POINTER(abi.Film), C.c_void_p, C.c_void_p, C.c_void_p]
        L.pvol_partition_tasks.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, _u32p, C.c_uint32, _u32p]
        L.pvol_render_frame_ranks.argtypes = [C.c_void_p, C.POINTER(abi.Camera), C.POINTER(abi.Film), C.POINTER(abi.Sampler), C.c_uint32, C.c_uint32,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pvol_enable_phase_timing.argtypes = [C.c_void_p, C.c_int]
        L.pvol_get_phase_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
        _lib = L
    return _lib


EXPORTS = ["pvol_abi_version", "pvol_strerror", "pvol_device_count", "pvol_default_params", "pvol_create",
           "pvol_destroy", "pvol_set_scene", "pvol_upload_photons", "pvol_preprocess", "pvol_photon_count",
           "pvol_download_photons", "pvol_li_batch", "pvol_li_batch_device", "pvol_li", "pvol_transmittance_batch",
           "pvol_get_stats", "pvol_enable_stats", "pvol_kernel_time_ms", "pvol_get_shoot_stats",
           "pvol_gaussian_filter_table", "pvol_compute_sub_window", "pvol_render_sample_count", "pvol_render_tasks_device",
           "pvol_film_add_samples_device", "pvol_film_resolve_device", "pvol_march_kernel_name", "pvol_check_errors", "pvol_get_preprocess_seconds", "pvol_get_accel_info", "pvol_surface_photon_count",
           "pvol_download_surface_photons", "pvol_radiance_photon_count", "pvol_download_radiance_photons",
           "pvol_set_surface_integrator", "pvol_enable_phase_timing", "pvol_get_phase_ms",
           "pvol_partition_tasks", "pvol_render_frame_ranks", "pvol_preprocess_blocks"]

SHOOT_STAT_NAMES = ["paths", "follow_calls", "no_hit", "march_steps", "interactions", "absorbed", "stored_volume",
                    "stored_caustic", "stored_direct", "stored_indirect", "split_children", "nshot"]


def partition_tasks(n_tasks, rank, n_ranks):
    """pvol_partition_tasks: the task numbers rank `rank` of `n_ranks` renders (pure host code: needs no GPU)."""
    n = C.c_uint32()
    _check(lib().pvol_partition_tasks(n_tasks, rank, n_ranks, None, 0, C.byref(n)), "pvol_partition_tasks")
    ids = np.zeros(n.value, np.uint32)
    _check(lib().pvol_partition_tasks(n_tasks, rank, n_ranks, ids.ctypes.data_as(_u32p), n.value, C.byref(n)), "pvol_partition_tasks")
    return ids


def _check(rc, where):
    if rc != abi.PVOL_OK:
        raise PvolError(rc, where)


class PhotonVolume:
    """Host-side mirror of the reference's plugin surface for this path: construct from the
    integrator/shooter parameters (CreatePhotonVolumeIntegrator / CreatePhotonShooter), hand it the
    scene, `preprocess()` or `upload_photons()`, then `li()` / `transmittance()`."""

    def __init__(self, params):
        self._h = C.c_void_p()
        self.params = params
        _check(lib().pvol_create(C.byref(params), C.byref(self._h)), "pvol_create")
        self._scene = None

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().pvol_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_scene(self, holder):
        self._scene = holder  # keeps the ctypes arrays alive
        _check(lib().pvol_set_scene(self._h, C.byref(holder.scene)), "pvol_set_scene")

    def upload_photons(self, p, wi, alpha):
        p = np.ascontiguousarray(p, np.float32).reshape(-1)
        wi = np.ascontiguousarray(wi, np.float32).reshape(-1)
        alpha = np.ascontiguousarray(alpha, np.float32).reshape(-1)
        n = p.size // 3
        if wi.size != 3 * n or alpha.size != 30 * n:
            raise ValueError("photon arrays disagree: %d positions, %d directions, %d weights" % (n, wi.size // 3, alpha.size // 30))
        _check(lib().pvol_upload_photons(self._h, p.ctypes.data_as(_f32p), wi.ctypes.data_as(_f32p), alpha.ctypes.data_as(_f32p), n),
               "pvol_upload_photons")

    def preprocess(self, n_tasks=1, block_paths=4096):
        """PhotonShooter::Preprocess on the device; block_paths < 4096: many small blocks (pvol_preprocess_blocks)."""
        if block_paths == 4096:
            _check(lib().pvol_preprocess(self._h, n_tasks), "pvol_preprocess")
        else:
            _check(lib().pvol_preprocess_blocks(self._h, n_tasks, block_paths), "pvol_preprocess_blocks")

    def preprocess_times(self):
        """(shoot seconds, search-structure build seconds) of the last preprocess()."""
        v = (C.c_double * 2)()
        _check(lib().pvol_get_preprocess_seconds(self._h, v), "pvol_get_preprocess_seconds")
        return float(v[0]), float(v[1])

    def accel_info(self):
        """(triangles in the device-built hierarchy -- 0 when the scene is scanned linearly --, build milliseconds)."""
        v = (C.c_double * 2)()
        _check(lib().pvol_get_accel_info(self._h, v), "pvol_get_accel_info")
        return int(v[0]), float(v[1])

    def surface_photons(self, kind):
        """(p, wo, alpha, n_paths) of the caustic (0) / direct (1) / indirect (2) store kept by the last preprocess()."""
        n, npaths = C.c_uint32(), C.c_uint32()
        _check(lib().pvol_surface_photon_count(self._h, kind, C.byref(n), C.byref(npaths)), "pvol_surface_photon_count")
        p, wo, a = np.zeros((n.value, 3), np.float32), np.zeros((n.value, 3), np.float32), np.zeros((n.value, 30), np.float32)
        _check(lib().pvol_download_surface_photons(self._h, kind, p.ctypes.data_as(_f32p), wo.ctypes.data_as(_f32p), a.ctypes.data_as(_f32p), n.value),
               "pvol_download_surface_photons")
        return p, wo, a, int(npaths.value)

    def radiance_photons(self):
        """(p, n, rho_r, rho_t) of the radiance photons kept by the last preprocess()."""
        n = C.c_uint32()
        _check(lib().pvol_radiance_photon_count(self._h, C.byref(n)), "pvol_radiance_photon_count")
        p, nn = np.zeros((n.value, 3), np.float32), np.zeros((n.value, 3), np.float32)
        rr, rt = np.zeros((n.value, 30), np.float32), np.zeros((n.value, 30), np.float32)
        _check(lib().pvol_download_radiance_photons(self._h, p.ctypes.data_as(_f32p), nn.ctypes.data_as(_f32p), rr.ctypes.data_as(_f32p),
                                                    rt.ctypes.data_as(_f32p), n.value), "pvol_download_radiance_photons")
        return p, nn, rr, rt

    def check_errors(self):
        """Raises PvolError(PVOL_E_LIMIT) if a batch enqueued through a device entry point hit a kernel limit."""
        _check(lib().pvol_check_errors(self._h), "pvol_check_errors")

    def render_frame_ranks(self, cam, film, smp, rank, n_ranks, nccl_comm, d_pixels, d_rgb=0, hip_stream=0):
        """One rank of an N-GPU frame behind the C ABI: partition, render, ncclReduce of the film, resolve on rank 0."""
        _check(lib().pvol_render_frame_ranks(self._h, C.byref(cam), C.byref(film), C.byref(smp), rank, n_ranks, nccl_comm, d_pixels, d_rgb, hip_stream),
               "pvol_render_frame_ranks")

    def enable_phase_timing(self, on=True):
        _check(lib().pvol_enable_phase_timing(self._h, int(bool(on))), "pvol_enable_phase_timing")

    def phase_ms(self, reset=False):
        """Device milliseconds per phase of render_tasks since the last reset: tile pre-pass, march + gather, surface, film."""
        v = (C.c_double * 6)()
        _check(lib().pvol_get_phase_ms(self._h, v, int(bool(reset))), "pvol_get_phase_ms")
        return {"tile_prepass": v[0], "march_gather": v[2], "surface": v[3], "film_add": v[4]}

    def shoot_stats(self):
        v = (C.c_uint64 * 12)()
        _check(lib().pvol_get_shoot_stats(self._h, v), "pvol_get_shoot_stats")
        return dict(zip(SHOOT_STAT_NAMES, [int(x) for x in v]))

    def photon_count(self):
        n = C.c_uint32()
        _check(lib().pvol_photon_count(self._h, C.byref(n)), "pvol_photon_count")
        return n.value

    def download_photons(self):
        n = self.photon_count()
        p = np.zeros((n, 3), np.float32)
        wi = np.zeros((n, 3), np.float32)
        alpha = np.zeros((n, 30), np.float32)
        _check(lib().pvol_download_photons(self._h, p.ctypes.data_as(_f32p), wi.ctypes.data_as(_f32p), alpha.ctypes.data_as(_f32p), n),
               "pvol_download_photons")
        return p, wi, alpha

    def li(self, rays, streams, output_kind=abi.OUT_SPECTRAL):
        """PhotonVolumeIntegrator::Li for a batch; returns (out[n, 60|4], draws[n]); streams['end_draw'] is updated."""
        rays = np.ascontiguousarray(rays)
        assert rays.dtype == abi.RAY_DTYPE and streams.dtype == abi.STREAM_DTYPE and streams.flags["C_CONTIGUOUS"]
        n = len(rays)
        out = np.zeros((n, 60 if output_kind == abi.OUT_SPECTRAL else 4), np.float32)
        draws = np.zeros(n, np.uint32)
        _check(lib().pvol_li_batch(self._h, rays.ctypes.data, n, streams.ctypes.data, len(streams), output_kind,
                                   out.ctypes.data_as(_f32p), draws.ctypes.data_as(_u32p)), "pvol_li_batch")
        return out, draws

    def li_device(self, d_rays, n_rays, d_streams, n_streams, output_kind, d_out, d_draws=0, hip_stream=0):
        """Device-pointer variant (integers): enqueue only, no synchronisation."""
        _check(lib().pvol_li_batch_device(self._h, d_rays, n_rays, d_streams, n_streams, output_kind, d_out, d_draws, hip_stream),
               "pvol_li_batch_device")

    # ---- tile driver (device pointers are integers, e.g. torch.Tensor.data_ptr())
    def render_tasks(self, camera, film, sampler, task_ids, d_pixels, debug=None, hip_stream=0):
        ids = np.ascontiguousarray(task_ids, np.uint32)
        _check(lib().pvol_render_tasks_device(self._h, C.byref(camera), C.byref(film), C.byref(sampler), ids.ctypes.data_as(_u32p), len(ids),
                                              d_pixels, C.byref(debug) if debug is not None else None, hip_stream), "pvol_render_tasks_device")

    def set_surface_integrator(self, n_used=50, max_dist=0.1, max_specular_depth=5, final_gather=False, caustic=None, n_paths=0,
                               from_preprocess=False, off=False, n_indirect=0):
        """PhotonIntegrator in front of the volume term (CreatePhotonMapSurfaceIntegrator, photonmap.cpp:336-363: nused 50,
        maxdist .1, maxspeculardepth 5).  `caustic` = (p[n,3], wo[n,3], alpha[n,30]) with `n_paths`, or from_preprocess=True to
        take the caustic photons the last preprocess() kept; off=True disables."""
        if off:
            _check(lib().pvol_set_surface_integrator(self._h, None, None, None, None, 0), "pvol_set_surface_integrator")
            return
        sp = abi.SurfaceParams()
        sp.n_used, sp.max_dist, sp.max_specular_depth, sp.final_gather = int(n_used), float(max_dist), int(max_specular_depth), int(bool(final_gather))
        sp.n_caustic_paths, sp.use_preprocess_store = int(n_paths), int(bool(from_preprocess))
        sp.n_indirect_photons = int(n_indirect)   # the integrator's indirect map, if it has one: refused (PVOL_E_UNSUPPORTED)
        if caustic is None or from_preprocess:
            _check(lib().pvol_set_surface_integrator(self._h, C.byref(sp), None, None, None, 0), "pvol_set_surface_integrator")
            return
        p, w, a = (np.ascontiguousarray(x, np.float32) for x in caustic)
        n = p.size // 3
        assert w.size == 3 * n and a.size == 30 * n
        _check(lib().pvol_set_surface_integrator(self._h, C.byref(sp), p.ctypes.data_as(_f32p), w.ctypes.data_as(_f32p), a.ctypes.data_as(_f32p), n),
               "pvol_set_surface_integrator")

    def film_add_samples(self, film, d_image_xy, d_xyz, stride, n, d_pixels, hip_stream=0):
        _check(lib().pvol_film_add_samples_device(self._h, C.byref(film), d_image_xy, d_xyz, stride, n, d_pixels, hip_stream),
               "pvol_film_add_samples_device")

    def film_resolve(self, film, d_pixels, d_rgb, hip_stream=0):
        _check(lib().pvol_film_resolve_device(self._h, C.byref(film), d_pixels, d_rgb, hip_stream), "pvol_film_resolve_device")

    def li_single(self, ray, mt, mti):
        """Per-sample shim: ray is a length-1 RAY_DTYPE array, mt a uint32[624] array (updated in place)."""
        Lv = np.zeros(30, np.float32)
        T = np.zeros(30, np.float32)
        i = C.c_int32(int(mti))
        ray = np.ascontiguousarray(ray)
        _check(lib().pvol_li(self._h, ray.ctypes.data, mt.ctypes.data_as(_u32p), C.byref(i), Lv.ctypes.data_as(_f32p), T.ctypes.data_as(_f32p)),
               "pvol_li")
        return Lv, T, i.value

    def transmittance(self, rays, streams):
        rays = np.ascontiguousarray(rays)
        n = len(rays)
        out = np.zeros((n, 30), np.float32)
        _check(lib().pvol_transmittance_batch(self._h, rays.ctypes.data, n, streams.ctypes.data, len(streams), out.ctypes.data_as(_f32p)),
               "pvol_transmittance_batch")
        return out

    def enable_stats(self, on=True):
        _check(lib().pvol_enable_stats(self._h, int(on)), "pvol_enable_stats")

    def stats(self, reset=False):
        s = abi.Stats()
        _check(lib().pvol_get_stats(self._h, C.byref(s), int(reset)), "pvol_get_stats")
        return {k: int(getattr(s, k)) for k, _ in abi.Stats._fields_}

    def march_kernel_name(self):
        return lib().pvol_march_kernel_name(self._h).decode()

    def kernel_time_ms(self, reset=False):
        avg = C.c_double()
        n = C.c_uint64()
        _check(lib().pvol_kernel_time_ms(self._h, C.byref(avg), C.byref(n), int(reset)), "pvol_kernel_time_ms")
        return avg.value, n.value


def gaussian_filter_table(xwidth=2.0, ywidth=2.0, alpha=2.0):
    t = np.zeros(256, np.float32)
    lib().pvol_gaussian_filter_table(xwidth, ywidth, alpha, t.ctypes.data_as(_f32p))
    return t


def sub_window(sampler, task):
    w = (C.c_int32 * 4)()
    lib().pvol_compute_sub_window(C.byref(sampler), task, w)
    return list(w)


def render_sample_count(sampler, task_ids):
    ids = np.ascontiguousarray(task_ids, np.uint32)
    return int(lib().pvol_render_sample_count(C.byref(sampler), ids.ctypes.data_as(_u32p), len(ids)))
