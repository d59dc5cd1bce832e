"""ctypes binding of libptamd.so (include/ptamd.h) + the Python mirror of the reference's operators.

Mirror of the reference interface for this path (same names, argument meaning, error behaviour):
  Scene.CreateAccelerationStructures / GetTopLevelAccelerationStructure   Source/Scene.ixx:282-380
  GBufferGeneration{GPUBuffers, Textures, Render(constants)}              Source/GBufferGeneration.ixx:27-122
  Raytracing{GPUBuffers, Textures, SetConstants, Render}                  Source/Raytracing.ixx:29-250
PyTorch is used only as plumbing: device memory (uint8 tensors), streams, torch.distributed.
There is NO CPU fallback: if libptamd.so is missing or no GPU is visible this module raises.
"""
import ctypes as C
import os

import numpy as np

from . import layouts as L

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libptamd.so")
_LIB = None

EXPORTS = [
    "pt_abi_version", "pt_create", "pt_destroy", "pt_last_error", "pt_set_stream", "pt_sync", "pt_set_frames_in_flight", "pt_set_round_chains",
    "pt_heap_resize", "pt_heap_set_buffer", "pt_heap_set_texture", "pt_build_bottom_level", "pt_update_bottom_level", "pt_release_bottom_level", "pt_skin_mesh",
    "pt_build_top_level", "pt_get_accel_stats", "pt_share_scene", "pt_set_camera", "pt_set_scene_data", "pt_set_object_data", "pt_invalidate_object_data",
    "pt_set_instance_data", "pt_set_sharding", "pt_local_rows", "pt_deinterleave_bands", "pt_gbuffer_render",
    "pt_comm_get_unique_id", "pt_comm_init", "pt_comm_adopt", "pt_comm_destroy", "pt_gather_bands", "pt_gather_plan",
    "pt_raytrace_set_constants", "pt_raytrace_render", "pt_trace_visibility", "pt_bsdf_evaluate", "pt_reset_counters", "pt_get_counters",
    "pt_set_debug_flags", "pt_debug_read_mismatch", "pt_debug_download_blob", "pt_debug_trace_ray", "pt_enable_kernel_timing", "pt_get_kernel_timing", "pt_get_round_timing",
]


class PtError(RuntimeError):
    """reference: std::system_error from ThrowIfFailed (Source/ErrorHelpers.ixx:16-32)."""


class PtInvalidArgument(ValueError):
    """reference: Throw<std::invalid_argument> (Source/RaytracingHelpers.ixx:83-88)."""


class GeometryDesc(C.Structure):
    _fields_ = [("VertexBuffer", C.c_void_p), ("VertexCount", C.c_uint32), ("VertexStride", C.c_uint32),
                ("IndexBuffer", C.c_void_p), ("IndexCount", C.c_uint32), ("IndexStride", C.c_uint32),
                ("Flags", C.c_uint32), ("_pad", C.c_uint32)]


class InstanceDesc(C.Structure):
    _fields_ = [("Transform", C.c_float * 12), ("InstanceID", C.c_uint32), ("InstanceMask", C.c_uint32),
                ("AccelerationStructure", C.c_uint64)]


TEXTURE_NAMES = L.GBUFFER_ORDER + ["RadianceF32", "Diffuse", "Specular", "SpecularHitDistance"]


class Textures(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in TEXTURE_NAMES]


class Sharding(C.Structure):
    _fields_ = [("RankIndex", C.c_uint32), ("RankCount", C.c_uint32), ("BandHeight", C.c_uint32), ("_pad", C.c_uint32)]


class BandMessage(C.Structure):
    _fields_ = [("Peer", C.c_uint32), ("IsSend", C.c_uint32), ("Band", C.c_uint32), ("_pad", C.c_uint32),
                ("LocalOffset", C.c_uint64), ("FullOffset", C.c_uint64), ("Bytes", C.c_uint64)]


class Counters(C.Structure):
    _fields_ = [("PrimaryRays", C.c_uint64), ("SecondaryRays", C.c_uint64), ("NodesVisited", C.c_uint64),
                ("TrianglesTested", C.c_uint64), ("WavefrontIterations", C.c_uint64), ("BvhMismatches", C.c_uint64),
                ("StackOverflows", C.c_uint64), ("MaxNodesPerRay", C.c_uint64)]


class BlobLayout(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("InstanceOffset16", "NodeOffset16", "TriangleOffset16", "LeafInstanceOffset16",
                                          "InstanceCount", "NodeCount", "TriangleCount", "Bytes")]


class AccelStats(C.Structure):
    _fields_ = [("InstanceCount", C.c_uint32), ("BottomLevelCount", C.c_uint32), ("TriangleCount", C.c_uint64),
                ("NodeBytes", C.c_uint64), ("TriangleBytes", C.c_uint64), ("NodeSizeBytes", C.c_uint32),
                ("TriangleSizeBytes", C.c_uint32), ("MaxBottomLevelDepth", C.c_uint32), ("TopLevelDepth", C.c_uint32),
                ("BlobBytes", C.c_uint64), ("SharedScene", C.c_uint32), ("NormalRecords", C.c_uint32),
                ("OwnedBottomLevelBytes", C.c_uint64), ("RoundObjectsInLds", C.c_uint32), ("RoundRecordsInLds", C.c_uint32)]


def load_library():
    """Load libptamd.so; fail loudly when the HIP extension has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise PtError(f"{LIB_PATH} is missing: run __graft_entry__.build() (make -C .../csrc). "
                          "The path tracer has no CPU fallback.")
        # torch bundles its own libamdhip64.so.7; import it first so libptamd.so binds to the SAME HIP
        # runtime (two runtimes in one process cannot both open the device, and tensors would not be shared)
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        lib.pt_last_error.restype = C.c_char_p
        lib.pt_last_error.argtypes = [C.c_void_p]
        lib.pt_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        lib.pt_destroy.argtypes = [C.c_void_p]; lib.pt_destroy.restype = None
        lib.pt_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        lib.pt_sync.argtypes = [C.c_void_p]
        lib.pt_set_frames_in_flight.argtypes = [C.c_void_p, C.c_uint32]
        lib.pt_set_round_chains.argtypes = [C.c_void_p, C.c_uint32]
        lib.pt_heap_resize.argtypes = [C.c_void_p, C.c_uint32]
        lib.pt_heap_set_buffer.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32]
        lib.pt_heap_set_texture.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        lib.pt_build_bottom_level.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
        lib.pt_release_bottom_level.argtypes = [C.c_void_p, C.c_uint64]
        lib.pt_update_bottom_level.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32]
        lib.pt_skin_mesh.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        lib.pt_build_top_level.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        lib.pt_get_accel_stats.argtypes = [C.c_void_p, C.c_void_p]
        lib.pt_share_scene.argtypes = [C.c_void_p, C.c_void_p]
        for f in (lib.pt_set_camera, lib.pt_set_scene_data, lib.pt_set_sharding, lib.pt_raytrace_set_constants,
                  lib.pt_raytrace_render, lib.pt_get_counters):
            f.argtypes = [C.c_void_p, C.c_void_p]
        lib.pt_set_object_data.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        lib.pt_set_instance_data.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        lib.pt_invalidate_object_data.argtypes = [C.c_void_p]
        lib.pt_local_rows.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        lib.pt_deinterleave_bands.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                              C.c_uint32, C.c_uint32, C.c_uint32]
        lib.pt_comm_get_unique_id.argtypes = [C.c_void_p]
        lib.pt_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        lib.pt_comm_adopt.argtypes = [C.c_void_p, C.c_void_p]
        lib.pt_comm_destroy.argtypes = [C.c_void_p]
        lib.pt_gather_bands.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        lib.pt_gather_plan.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        lib.pt_gbuffer_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.pt_trace_visibility.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        lib.pt_bsdf_evaluate.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        lib.pt_reset_counters.argtypes = [C.c_void_p]
        lib.pt_set_debug_flags.argtypes = [C.c_void_p, C.c_uint32]
        lib.pt_debug_read_mismatch.argtypes = [C.c_void_p, C.c_void_p]
        lib.pt_debug_download_blob.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        lib.pt_debug_trace_ray.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        lib.pt_enable_kernel_timing.argtypes = [C.c_void_p, C.c_int]
        lib.pt_get_round_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
        lib.pt_get_kernel_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                             C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        _LIB = lib
    return _LIB


def local_rows(height, rank=0, world=1, band=16):
    s = Sharding(rank, world, band, 0)
    out = C.c_uint32(0)
    if load_library().pt_local_rows(C.byref(s), height, C.byref(out)) != 0:
        raise PtInvalidArgument("invalid sharding")
    return out.value


def gather_plan(height, row_bytes, rank, world, band=16, root=0):
    """pt_gather_plan: the (peer, is_send, band, local_offset, full_offset, bytes) messages rank `rank` issues in a gather. No GPU."""
    lib = load_library()
    s = Sharding(rank, world, band, 0)
    n = C.c_uint32(0)
    if lib.pt_gather_plan(C.byref(s), height, row_bytes, root, None, 0, C.byref(n)) != 0:
        raise PtInvalidArgument("invalid sharding / root")
    msgs = (BandMessage * max(1, n.value))()
    if lib.pt_gather_plan(C.byref(s), height, row_bytes, root, msgs, n.value, C.byref(n)) != 0:
        raise PtInvalidArgument("invalid sharding / root")
    return [(m.Peer, m.IsSend, m.Band, m.LocalOffset, m.FullOffset, m.Bytes) for m in msgs[:n.value]]


class DeviceContext:
    """One per GPU (reference: DeviceContext/CommandList pair, one D3D12 device)."""

    def __init__(self, device_ordinal=0, stream=None):
        self.lib = load_library()
        h = C.c_void_p()
        st = self.lib.pt_create(device_ordinal, C.byref(h))
        if st != 0:
            raise PtError(f"pt_create failed ({st}): {self.lib.pt_last_error(None).decode()}")
        self.handle = h
        self.device_ordinal = device_ordinal
        if stream is not None:
            self.check(self.lib.pt_set_stream(self.handle, C.c_void_p(stream)))

    def check(self, status):
        if status == 0:
            return
        msg = self.lib.pt_last_error(self.handle).decode()
        if status == -1:
            raise PtInvalidArgument(msg)
        raise PtError(f"status {status}: {msg}")

    def sync(self):
        self.check(self.lib.pt_sync(self.handle))

    def set_frames_in_flight(self, n):
        self.check(self.lib.pt_set_frames_in_flight(self.handle, n))

    def set_round_chains(self, n):
        """0 = the library's choice; n independent chains of launches per frame (pt_set_round_chains)."""
        self.check(self.lib.pt_set_round_chains(self.handle, n))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.pt_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # measurement ---------------------------------------------------------------------------
    def reset_counters(self):
        self.check(self.lib.pt_reset_counters(self.handle))

    def counters(self):
        c = Counters()
        self.check(self.lib.pt_get_counters(self.handle, C.byref(c)))
        return c

    def set_debug_flags(self, flags):
        self.check(self.lib.pt_set_debug_flags(self.handle, flags))

    def invalidate_object_data(self):
        """VertexDesc / MeshDescriptors of the bound ObjectData were rewritten in place: resolve and check them again at the next render."""
        self.check(self.lib.pt_invalidate_object_data(self.handle))

    def enable_kernel_timing(self, on=True):
        self.check(self.lib.pt_enable_kernel_timing(self.handle, 1 if on else 0))

    def kernel_timing(self):
        e, s, ne, ns = C.c_float(), C.c_float(), C.c_uint32(), C.c_uint32()
        self.check(self.lib.pt_get_kernel_timing(self.handle, C.byref(e), C.byref(s), C.byref(ne), C.byref(ns)))
        r, nr = C.c_float(), C.c_uint32()
        self.check(self.lib.pt_get_round_timing(self.handle, C.byref(r), C.byref(nr)))
        return {"extend_ms": e.value, "shade_ms": s.value, "extend_launches": ne.value, "shade_launches": ns.value,
                "round_ms": r.value, "round_launches": nr.value}

    def download_blob(self):
        """(layout, bytes) of the traversal copy of the scene -- for the structural checks in tests/."""
        lay = BlobLayout()
        self.check(self.lib.pt_debug_download_blob(self.handle, None, 0, C.byref(lay)))
        buf = np.zeros(lay.Bytes, np.uint8)
        self.check(self.lib.pt_debug_download_blob(self.handle, C.c_void_p(buf.ctypes.data), buf.nbytes, C.byref(lay)))
        return lay, buf

    def accel_stats(self):
        s = AccelStats()
        self.check(self.lib.pt_get_accel_stats(self.handle, C.byref(s)))
        return s

    # multi-GPU gather (pt_comm.hip) ------------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        """ncclGetUniqueId through the library: 128 bytes, made by one rank, handed to the others by the caller."""
        lib = load_library()
        buf = (C.c_uint8 * 128)()
        st = lib.pt_comm_get_unique_id(buf)
        if st != 0:
            raise PtError(f"pt_comm_get_unique_id failed ({st}): {lib.pt_last_error(None).decode()}")
        return bytes(buf)

    def comm_init(self, unique_id, rank, world):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self.check(self.lib.pt_comm_init(self.handle, buf, rank, world))

    def gather_bands(self, local, dst_full, width, height, pixel_bytes, root=0):
        """local / dst_full: CUDA tensors (dst_full may be None on a non-root rank)."""
        self.check(self.lib.pt_gather_bands(self.handle, C.c_void_p(local.data_ptr()),
                                            C.c_void_p(dst_full.data_ptr() if dst_full is not None else None), width, height, pixel_bytes, root))

    def set_sharding(self, rank, world, band=16):
        s = Sharding(rank, world, band, 0)
        self.check(self.lib.pt_set_sharding(self.handle, C.byref(s)))
        self.sharding = (rank, world, band)


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise PtError("no GPU visible to torch: the path tracer has no CPU fallback")
    return torch


def to_device(arr, device):
    """numpy array (any dtype, incl. structured) -> uint8 CUDA tensor holding the same bytes."""
    torch = _torch()
    raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
    if raw.size == 0:
        raw = np.zeros(16, np.uint8)
    return torch.from_numpy(raw.copy()).to(device)


class SharedScene:
    """A second context's view of a Scene built on another context of the same GPU (pt_share_scene): frames in flight on separate
    streams render one copy of the scene. Holds a reference to the owning Scene so that it outlives the view."""

    def __init__(self, ctx, owner):
        self.ctx, self.owner, self.desc, self.device = ctx, owner, owner.desc, owner.device
        ctx.check(ctx.lib.pt_share_scene(ctx.handle, owner.ctx.handle))

    def GetTopLevelAccelerationStructure(self):
        return self.ctx.handle

    def close(self):
        pass


class Scene:
    """Device-side scene: the part of Scene (Source/Scene.ixx:75-403) and App::UpdateScene
    (Source/App.cpp:1016-1074) that feeds the hot path: vertex/index buffers, descriptor heap,
    ObjectData / InstanceData buffers, BLAS per mesh node, TLAS over instances."""
    SKIN_STAGING_RING = 4          # pinned staging buffers of the joint matrices, taken in turn (SkinSkeletalMeshes)

    def __init__(self, ctx, scene, device=None):
        torch = _torch()
        self.ctx, self.desc = ctx, scene
        self.device = device or torch.device("cuda", ctx.device_ordinal)
        lib = ctx.lib
        self._buffers = []
        ctx.check(lib.pt_heap_resize(ctx.handle, len(scene.heap)))
        heap_dev = []
        for i, item in enumerate(scene.heap):
            t = to_device(item.array, self.device)
            self._buffers.append(t); heap_dev.append(t)
            if item.kind == 0:
                ctx.check(lib.pt_heap_set_buffer(ctx.handle, i, C.c_void_p(t.data_ptr()), item.array.nbytes, item.stride))
            else:
                ctx.check(lib.pt_heap_set_texture(ctx.handle, i, C.c_void_p(t.data_ptr()), item.width, item.height, item.fmt,
                                                  1 if item.kind == 2 else 0))
        self.object_data = to_device(scene.object_data, self.device)
        self.instance_data = to_device(scene.instance_data, self.device)
        ctx.check(lib.pt_set_object_data(ctx.handle, C.c_void_p(self.object_data.data_ptr()), len(scene.object_data)))
        ctx.check(lib.pt_set_instance_data(ctx.handle, C.c_void_p(self.instance_data.data_ptr()), len(scene.instance_data)))
        self._heap_dev = heap_dev
        self.blas_ids = []
        self._skin_cache = {}
        self.CreateAccelerationStructures()

    def close(self):
        """~Scene (Source/Scene.ixx:108-123): the bottom levels this scene built go back to the context. The live top level
        dies with them: a render before the next scene's build answers PT_ERROR_NOT_READY."""
        ctx = self.ctx
        if getattr(ctx, "handle", None):
            for bid in self.blas_ids:
                ctx.lib.pt_release_bottom_level(ctx.handle, bid)
        self.blas_ids = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _geometry_descs(self, node_index):
        """D3D12_RAYTRACING_GEOMETRY_DESC per mesh of a mesh node + the build flags of Scene.ixx:320-329: OPAQUE iff the mesh
        has no material or AlphaMode::Opaque; skeletal mesh nodes are built PREFER_FAST_BUILD | ALLOW_UPDATE, static ones
        PREFER_FAST_TRACE."""
        scene = self.desc
        first, count = scene.blas[node_index]
        geoms = (GeometryDesc * max(1, count))()
        skeletal = False
        for g in range(count):
            mesh, hv, hi = scene.geometry[first + g]
            d = geoms[g]
            d.VertexBuffer = self._heap_dev[hv].data_ptr()
            d.VertexCount, d.VertexStride = len(mesh.vertices), mesh.vertices.dtype.itemsize
            d.IndexBuffer = self._heap_dev[hi].data_ptr()
            d.IndexCount, d.IndexStride = mesh.indices.size, mesh.indices.dtype.itemsize
            alpha_mode = int(mesh.material["AlphaMode"]) if mesh.material is not None else 0
            d.Flags = 1 if alpha_mode == 0 else 0
            skeletal = skeletal or getattr(mesh, "skeletal_vertices", None) is not None
        return geoms, count, (0x8 | 0x1) if skeletal else 0x4

    def SkinSkeletalMeshes(self, mesh, skeletal_transforms):
        """Scene::SkinSkeletalMeshes (Source/Scene.ixx:233-280) for one mesh: joint transforms -> SkeletalMeshSkinning."""
        ctx, lib = self.ctx, self.ctx.lib
        hv = next(h for m, h, _ in self.desc.geometry if m is mesh)
        hm = int(self.desc._motion_heap[id(mesh)])
        torch = _torch()
        if id(mesh) not in self._skin_cache:                     # skeletal vertices once; joint matrices: a ring of pinned staging tensors
            n = int(np.asarray(skeletal_transforms).size)         # + a device tensor that live as long as the scene (no allocation per frame)
            ring = [[torch.zeros(n, dtype=torch.float32).pin_memory(), None] for _ in range(self.SKIN_STAGING_RING)]
            self._skin_cache[id(mesh)] = [to_device(mesh.skeletal_vertices, self.device), ring,
                                          torch.zeros(n, dtype=torch.float32, device=self.device), 0]
        entry = self._skin_cache[id(mesh)]
        sk, ring, tr = entry[0], entry[1], entry[2]
        slot = ring[entry[3] % len(ring)]; entry[3] += 1
        # The host may run frames ahead of the stream (nothing on the dynamic path synchronises): a pinned buffer is rewritten only after
        # the H2D copy that last read it has run -- the event recorded behind that copy. With the ring that wait is over before it starts
        # unless the host is a whole ring ahead. The device tensor needs no ring: copy N+1 is stream-ordered behind skin kernel N.
        staging = slot[0]
        if slot[1] is not None:
            slot[1].synchronize()
        staging.copy_(torch.from_numpy(np.ascontiguousarray(skeletal_transforms, np.float32).reshape(-1)))
        tr.copy_(staging, non_blocking=True)
        if slot[1] is None:
            slot[1] = torch.cuda.Event()
        slot[1].record(torch.cuda.current_stream(self.device))
        ctx.check(lib.pt_skin_mesh(ctx.handle, C.c_void_p(sk.data_ptr()), C.c_void_p(tr.data_ptr()),
                                   C.c_void_p(self._heap_dev[hv].data_ptr()), C.c_void_p(self._heap_dev[hm].data_ptr()), len(mesh.vertices)))

    def UpdateAccelerationStructures(self, node_index):
        """the PERFORM_UPDATE branch of Scene::CreateAccelerationStructures (Source/Scene.ixx:327-345) + TLAS rebuild."""
        ctx, lib = self.ctx, self.ctx.lib
        geoms, count, flags = self._geometry_descs(node_index)
        ctx.check(lib.pt_update_bottom_level(ctx.handle, self.blas_ids[node_index], C.addressof(geoms), count, flags))
        self._build_top_level()

    def download(self, heap_index, dtype):
        return self._heap_dev[heap_index].cpu().numpy().view(dtype)

    def CreateAccelerationStructures(self):
        """Scene::CreateAccelerationStructures (Source/Scene.ixx:286-380)."""
        ctx, scene, lib = self.ctx, self.desc, self.ctx.lib
        for bid in self.blas_ids:
            ctx.check(lib.pt_release_bottom_level(ctx.handle, bid))
        self.blas_ids = []
        for node_index in range(len(scene.blas)):             # one BLAS per MeshNode, one geometry per Mesh
            geoms, count, flags = self._geometry_descs(node_index)
            bid = C.c_uint64(0)
            ctx.check(lib.pt_build_bottom_level(ctx.handle, C.addressof(geoms), count, flags, C.byref(bid)))
            self.blas_ids.append(bid.value)
        self._build_top_level()

    def _build_top_level(self):
        """instance descs as Scene::CreateAccelerationStructures fills them (Scene.ixx:365-377). The array is kept: a dynamic frame
        re-submits it as it is (transforms of the static instances do not move; update_instance_transform() for those that do)."""
        ctx, scene, lib = self.ctx, self.desc, self.ctx.lib
        n = len(scene.objects)
        if getattr(self, "_descs", None) is None or len(self._descs) != max(1, n) or self._descs_ids != list(self.blas_ids):
            descs = (InstanceDesc * max(1, n))()
            tr = np.ascontiguousarray(scene.instance_data["ObjectToWorld"], np.float32).reshape(n, 12) if n else np.zeros((0, 12), np.float32)
            for i in range(n):
                descs[i].Transform = (C.c_float * 12)(*tr[i].tolist())
                descs[i].InstanceID = int(scene.instance_ids[i])
                descs[i].InstanceMask = int(scene.instance_masks[i])
                descs[i].AccelerationStructure = self.blas_ids[int(scene.instance_blas[i])]
            self._descs, self._descs_ids = descs, list(self.blas_ids)
        ctx.check(lib.pt_build_top_level(ctx.handle, C.addressof(self._descs), n, 0x4))

    def GetTopLevelAccelerationStructure(self):
        return self.ctx.handle        # the context owns the single TLAS


def alloc_textures(width, local_rows_, device, with_f32=False, with_denoiser_outputs=False):
    """The G-buffer textures of App::CreateWindowSizeDependentResources (Source/App.cpp:438-455) as
    linear CUDA tensors in the same DXGI formats."""
    torch = _torch()
    out = {}
    tmap = {"<f4": torch.float32, "<i2": torch.int16, "<u2": torch.int16, "u1": torch.uint8}
    for name, (dt, ch) in L.GBUFFER_FORMATS.items():
        out[name] = torch.zeros((local_rows_, width, ch), dtype=tmap[dt], device=device)
    if with_f32:
        out["RadianceF32"] = torch.zeros((local_rows_, width, 4), dtype=torch.float32, device=device)
    if with_denoiser_outputs:
        for name, (dt, ch) in L.DENOISER_FORMATS.items():
            out[name] = torch.zeros((local_rows_, width, ch), dtype=tmap[dt], device=device)
    return out


def _pack_textures(tex):
    t = Textures()
    for n in TEXTURE_NAMES:
        v = tex.get(n) if tex else None
        setattr(t, n, v.data_ptr() if v is not None else None)
    return t


def textures_to_numpy(tex):
    out = {}
    for name, t in tex.items():
        a = t.detach().cpu().numpy()
        if name in L.GBUFFER_FORMATS:
            a = a.view(np.dtype(L.GBUFFER_FORMATS[name][0]))
        elif name in L.DENOISER_FORMATS:
            a = a.view(np.dtype(L.DENOISER_FORMATS[name][0]))
        out[name] = a
    return out


class GBufferGeneration:
    """Mirror of `struct GBufferGeneration` (Source/GBufferGeneration.ixx:27-122)."""
    Flags = L.GBufferFlags

    def __init__(self, ctx):
        self.ctx = ctx
        self.GPUBuffers = {"SceneData": None, "Camera": None, "InstanceData": None, "ObjectData": None}
        self.Textures = {}

    def Render(self, topLevelAccelerationStructure, constants):
        """constants: numpy GBUFFER_CONSTANTS {RenderSize, Flags}. GPUBuffers.Camera / SceneData are the
        host structs (numpy CAMERA / SCENE_DATA) the reference copies into its constant buffers."""
        ctx, lib = self.ctx, self.ctx.lib
        cam = np.array(self.GPUBuffers["Camera"]); sd = np.array(self.GPUBuffers["SceneData"])
        ctx.check(lib.pt_set_camera(ctx.handle, C.c_void_p(cam.ctypes.data)))
        ctx.check(lib.pt_set_scene_data(ctx.handle, C.c_void_p(sd.ctypes.data)))
        k = np.array(constants).reshape(())
        t = _pack_textures(self.Textures)
        ctx.check(lib.pt_gbuffer_render(ctx.handle, C.c_void_p(k.ctypes.data), C.addressof(t)))


class Raytracing:
    """Mirror of `struct Raytracing` (Source/Raytracing.ixx:29-250), DEFAULT permutation."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.GPUBuffers = {"SceneData": None, "Camera": None, "ObjectData": None}
        self.Textures = {}
        self._settings = None

    def SetConstants(self, graphicsSettings):
        self._settings = np.array(graphicsSettings).reshape(())
        self.ctx.check(self.ctx.lib.pt_raytrace_set_constants(self.ctx.handle, C.c_void_p(self._settings.ctypes.data)))

    def Render(self, topLevelAccelerationStructure):
        ctx, lib = self.ctx, self.ctx.lib
        cam = np.array(self.GPUBuffers["Camera"]); sd = np.array(self.GPUBuffers["SceneData"])
        ctx.check(lib.pt_set_camera(ctx.handle, C.c_void_p(cam.ctypes.data)))
        ctx.check(lib.pt_set_scene_data(ctx.handle, C.c_void_p(sd.ctypes.data)))
        t = _pack_textures(self.Textures)
        ctx.check(lib.pt_raytrace_render(ctx.handle, C.addressof(t)))


class Renderer:
    """App::RenderScene for this path (Source/App.cpp:1157-1329): G-buffer pass, then the path tracer."""

    def __init__(self, ctx, scene_gpu, width, height, with_f32=False, with_denoiser_outputs=False):
        self.ctx, self.scene, self.width, self.height = ctx, scene_gpu, width, height
        rank, world, band = getattr(ctx, "sharding", (0, 1, 16))
        self.local_rows = local_rows(height, rank, world, band)
        self.textures = alloc_textures(width, self.local_rows, scene_gpu.device, with_f32, with_denoiser_outputs)
        self.gbuffer = GBufferGeneration(ctx)
        self.raytracing = Raytracing(ctx)
        d = scene_gpu.desc
        for op in (self.gbuffer, self.raytracing):
            op.GPUBuffers["Camera"] = d.camera
            op.GPUBuffers["SceneData"] = d.scene_data
            op.Textures = self.textures
        self.constants = np.zeros((), L.GBUFFER_CONSTANTS)
        self.constants["RenderSize"] = (width, height)
        self.constants["Flags"] = L.GBufferFlags.DefaultNoDenoiser        # App.cpp:1224 with Denoiser::None

    def render(self, settings):
        tlas = self.scene.GetTopLevelAccelerationStructure()
        denoiser = int(np.array(settings).reshape(())["Denoiser"])
        self.constants["Flags"] = 0xFFFFFFFF if denoiser != L.DENOISER_NONE else L.GBufferFlags.DefaultNoDenoiser   # App.cpp:1223
        self.gbuffer.Render(tlas, self.constants)
        if int(np.array(settings).reshape(())["Bounces"]) > 0:            # App.cpp:1277
            self.raytracing.SetConstants(settings)
            self.raytracing.Render(tlas)
