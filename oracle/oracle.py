"""ctypes front-end of the CPU oracle (oracle/pt_oracle.c). TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
PARITY STATUS: parity unpinned (no reference golden vectors exist; MathLib absent) -- see pt_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libpt_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("pt_oracle.c", "pt_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libpt_oracle.so"])
    return so


class GeometryDesc(C.Structure):
    _fields_ = [("Vertices", C.c_void_p), ("VertexCount", C.c_uint32), ("VertexStride", C.c_uint32),
                ("Indices", C.c_void_p), ("IndexCount", C.c_uint32), ("IndexStride", C.c_uint32),
                ("Flags", C.c_uint32), ("_pad", C.c_uint32)]


class BlasDesc(C.Structure):
    _fields_ = [("FirstGeometry", C.c_uint32), ("GeometryCount", C.c_uint32)]


class InstanceDesc(C.Structure):
    _fields_ = [("Transform", C.c_float * 12), ("InstanceID", C.c_uint32), ("InstanceMask", C.c_uint32),
                ("Blas", C.c_uint32), ("_pad", C.c_uint32)]


class HeapEntry(C.Structure):
    _fields_ = [("Ptr", C.c_void_p), ("Bytes", C.c_uint64), ("Stride", C.c_uint32), ("Kind", C.c_uint32)]


GB_NAMES = ["Position", "FlatNormal", "GeometricNormal", "LinearDepth", "NormalizedDepth", "MotionVector",
            "BaseColorMetalness", "DiffuseAlbedo", "SpecularAlbedo", "NormalRoughness", "IOR", "Transmission",
            "Radiance", "RadianceF32", "Diffuse", "Specular", "SpecularHitDistance"]


class GBufferTextures(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in GB_NAMES]


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.or_scene_create.restype = C.c_void_p
        L.or_scene_create.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                      C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
        L.or_scene_destroy.argtypes = [C.c_void_p]
        for f in (L.or_gbuffer_render, L.or_raytrace_render):
            f.restype = C.c_uint64
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]
        L.or_rng_init.restype = C.c_uint32
        L.or_rng_init.argtypes = [C.c_uint32] * 3
        L.or_rng_float.restype = C.c_float
        L.or_rng_float.argtypes = [C.POINTER(C.c_uint32)]
        L.or_sincos_2pi.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.or_f32_to_f16.restype = C.c_uint16; L.or_f32_to_f16.argtypes = [C.c_float]
        L.or_f16_to_f32.restype = C.c_float; L.or_f16_to_f32.argtypes = [C.c_uint16]
        L.or_f32_to_snorm16.restype = C.c_int16; L.or_f32_to_snorm16.argtypes = [C.c_float]
        L.or_snorm16_to_f32.restype = C.c_float; L.or_snorm16_to_f32.argtypes = [C.c_int16]
        L.or_f32_to_unorm8.restype = C.c_uint8; L.or_f32_to_unorm8.argtypes = [C.c_float]
        fp = C.POINTER(C.c_float)
        L.or_oct_encode.argtypes = [fp, fp]; L.or_oct_decode.argtypes = [fp, fp]
        L.or_ray_triangle.restype = C.c_int
        L.or_ray_triangle.argtypes = [fp, fp, C.c_float, C.c_float, fp, fp, fp, fp, fp, fp]
        L.or_bsdf_sample.restype = C.c_int
        L.or_bsdf_sample.argtypes = [fp, C.c_int, fp, fp, fp, fp, C.c_uint32, fp, C.POINTER(C.c_int), fp, fp, fp]
        L.or_env_term_rtg.argtypes = [fp, C.c_float, C.c_float, fp]
        L.or_safe_spawn.argtypes = [fp, fp, fp, fp, fp, fp, fp, fp, fp]
        L.or_invert_3x4.argtypes = [fp, fp]
        L.or_bsdf_evaluate.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.or_trace_visibility.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.or_skin_mesh.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.or_texture_sample.argtypes = [C.c_void_p, C.c_float, C.c_float, fp]
        L.or_cube_sample.argtypes = [C.c_void_p, fp, fp]
        _LIB = L
    return _LIB


def _fa(a):
    return np.ascontiguousarray(a, np.float32)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class OracleScene:
    """Holds the host arrays of a scenes.Scene alive and the C-side OrScene built from them."""

    def __init__(self, scene, accel_mode=0):
        L = lib()
        self.scene = scene
        self._geoms = (GeometryDesc * max(1, len(scene.geometry)))()
        for i, (mesh, hv, hi) in enumerate(scene.geometry):
            g = self._geoms[i]
            g.Vertices = mesh.vertices.ctypes.data
            g.VertexCount, g.VertexStride = len(mesh.vertices), mesh.vertices.dtype.itemsize
            g.Indices = mesh.indices.ctypes.data
            g.IndexCount, g.IndexStride = mesh.indices.size, mesh.indices.dtype.itemsize
            alpha_mode = int(mesh.material["AlphaMode"]) if mesh.material is not None else 0
            g.Flags = 1 if alpha_mode == 0 else 0       # OPAQUE iff AlphaMode::Opaque or no material (Scene.ixx:320-324)
        self._blas = (BlasDesc * max(1, len(scene.blas)))()
        for i, (f, c) in enumerate(scene.blas):
            self._blas[i].FirstGeometry, self._blas[i].GeometryCount = f, c
        n = len(scene.objects)
        self._inst = (InstanceDesc * max(1, n))()
        for i in range(n):
            t = np.ascontiguousarray(scene.instance_data[i]["ObjectToWorld"], np.float32).reshape(-1)
            self._inst[i].Transform = (C.c_float * 12)(*t.tolist())
            self._inst[i].InstanceID = int(scene.instance_ids[i])
            self._inst[i].InstanceMask = int(scene.instance_masks[i])
            self._inst[i].Blas = int(scene.instance_blas[i])
        self._heap = (HeapEntry * max(1, len(scene.heap)))()
        for i, item in enumerate(scene.heap):
            e = self._heap[i]
            e.Ptr, e.Kind = item.array.ctypes.data, item.kind
            if item.kind == 0:
                e.Bytes, e.Stride = item.array.nbytes, item.stride
            else:                                                   # texture: Bytes = width | height << 32, Stride = format
                e.Bytes, e.Stride = item.width | (item.height << 32), item.fmt
        self._od = np.ascontiguousarray(scene.object_data)
        self._id = np.ascontiguousarray(scene.instance_data)
        self.handle = L.or_scene_create(C.addressof(self._geoms), len(scene.geometry), C.addressof(self._blas), len(scene.blas),
                                        C.addressof(self._inst), n, self._od.ctypes.data, len(self._od),
                                        self._id.ctypes.data, C.addressof(self._heap), len(scene.heap), accel_mode)

    def close(self):
        if self.handle:
            lib().or_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        self.close()

    @staticmethod
    def _textures(gb, extra_f32=None):
        t = GBufferTextures()
        for n in GB_NAMES:
            if n != "RadianceF32":
                setattr(t, n, gb[n].ctypes.data if n in gb and gb[n] is not None else None)
        t.RadianceF32 = extra_f32.ctypes.data if extra_f32 is not None else None
        return t

    def gbuffer(self, consts, gb, camera=None, scene_data=None, rows=None, threads=0):
        cam = np.array(camera if camera is not None else self.scene.camera)
        sd = np.array(scene_data if scene_data is not None else self.scene.scene_data)
        k = np.array(consts).reshape(())
        t = self._textures(gb)
        y0, y1 = rows if rows else (0, int(k["RenderSize"][1]))
        return lib().or_gbuffer_render(self.handle, cam.ctypes.data, sd.ctypes.data, k.ctypes.data, C.addressof(t), y0, y1, threads)

    def raytrace(self, settings, gb, camera=None, scene_data=None, rows=None, threads=0, radiance_f32=None):
        cam = np.array(camera if camera is not None else self.scene.camera)
        sd = np.array(scene_data if scene_data is not None else self.scene.scene_data)
        gs = np.array(settings).reshape(())
        t = self._textures(gb, radiance_f32)
        y0, y1 = rows if rows else (0, int(gs["RenderSize"][1]))
        return lib().or_raytrace_render(self.handle, cam.ctypes.data, sd.ctypes.data, gs.ctypes.data, C.addressof(t), y0, y1, threads)


def render(scene, settings, gbuffer_flags=0xFFFFFFFF & ~0xC0, accel_mode=0, threads=0, want_f32=False, layouts=None):
    """G-buffer pass then path tracer, as App::RenderScene does (Source/App.cpp:1196-1328).

    Returns (gbuffer dict incl. final Radiance, rays_traced, radiance_f32 or None)."""
    W, H = int(settings["RenderSize"][0]), int(settings["RenderSize"][1])
    gb = {k: np.zeros((H, W, c), dt) for k, (dt, c) in layouts.GBUFFER_FORMATS.items()}
    gb.update({k: np.zeros((H, W, c), dt) for k, (dt, c) in layouts.DENOISER_FORMATS.items()})
    consts = np.zeros((), layouts.GBUFFER_CONSTANTS)
    consts["RenderSize"] = (W, H); consts["Flags"] = gbuffer_flags
    osc = OracleScene(scene, accel_mode)
    rays = osc.gbuffer(consts, gb, threads=threads)
    f32 = np.zeros((H, W, 4), np.float32) if want_f32 else None
    if int(settings["Bounces"]) > 0:       # App.cpp:1277
        rays += osc.raytrace(settings, gb, threads=threads, radiance_f32=f32)
    osc.close()
    return gb, rays, f32
