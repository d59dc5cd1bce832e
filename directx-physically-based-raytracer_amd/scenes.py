"""Procedural scenes in the reference's Mesh / MeshNode / ObjectData / InstanceData layout.

The reference's Cornell Box asset (Assets/Scenes/Default.json, Source/App.cpp:129-131) is not in
its tree, so the BASELINE.json configs are generated here (SURVEY.md 8d "Synthetic inputs"):
  cornell_box()      C1 / C2 / C4   (~36-60 triangles, 8-9 mesh-node instances)
  sponza_scale()     C3             (~250k triangles, one BLAS, 24 materials)
  instanced_grid()   C5             (10k instances of one mesh, two-level BVH)
What the host does with them mirrors Scene::Refresh (Source/Scene.ixx:195-231: InstanceData,
FirstGeometryIndex) and App::UpdateScene (Source/App.cpp:1016-1074: ObjectData fill).
"""
import math
from dataclasses import dataclass, field

import numpy as np

from . import layouts as L


# ----------------------------------------------------------------------------------------------
# host-side encoders (Source/Vertex.ixx:17-26; MathLib float2_to_snorm_16_16 / float2_to_float16_t2)
# ----------------------------------------------------------------------------------------------
def encode_snorm16(v):
    v = np.clip(np.asarray(v, np.float64), -1.0, 1.0) * 32767.0
    return np.where(v >= 0, np.floor(v + 0.5), np.ceil(v - 0.5)).astype(np.int16)


def make_vertices(positions, normals=None, uvs=None, tangents=None, uvs1=None):
    vb = np.zeros(len(positions), L.VERTEX)
    vb["Position"] = np.asarray(positions, np.float32)
    if normals is not None:
        vb["Normal"] = encode_snorm16(normals)
    if tangents is not None:
        vb["Tangent"] = encode_snorm16(tangents)
    if uvs is not None:
        vb["TexCoord0"] = np.asarray(uvs, np.float16)
    if uvs1 is not None:
        vb["TexCoord1"] = np.asarray(uvs1, np.float16)
    return vb


def make_indices(idx):
    """u16 if index count <= 65535 else u32 (Source/GLTFHelpers.ixx:183-188)."""
    idx = np.asarray(idx).reshape(-1)
    return idx.astype(np.uint16 if idx.size <= 65535 else np.uint32)


FMT_RGBA8_UNORM, FMT_RGBA8_UNORM_SRGB, FMT_RGBA32_FLOAT = 0, 1, 2
KIND_BUFFER, KIND_TEXTURE2D, KIND_TEXTURECUBE = 0, 1, 2
TEX_SLOTS = ["BaseColor", "EmissiveColor", "Metallic", "Roughness", "MetallicRoughness", "Transmission", "Normal"]   # Material.ixx:22-33


@dataclass
class Texture:
    """Mip 0 of a texture. data: [H,W,4] (2D) or [6,H,W,4] (cube, faces +X,-X,+Y,-Y,+Z,-Z); uint8 or float32.
    srgb: base-colour / emissive textures are created as *_UNORM_SRGB (Source/GLTFHelpers.ixx:375-391)."""
    data: np.ndarray
    srgb: bool = False

    @property
    def fmt(self):
        if self.data.dtype == np.float32:
            return FMT_RGBA32_FLOAT
        return FMT_RGBA8_UNORM_SRGB if self.srgb else FMT_RGBA8_UNORM


@dataclass
class HeapItem:                  # one descriptor: a buffer (stride = typed element size) or a texture
    array: np.ndarray
    stride: int = 0
    kind: int = KIND_BUFFER
    width: int = 0
    height: int = 0
    fmt: int = 0


@dataclass
class Mesh:                      # Source/Model.ixx:26-47
    vertices: np.ndarray         # L.VERTEX
    indices: np.ndarray          # uint16 / uint32
    has_normals: bool = True
    material: np.ndarray = None  # L.MATERIAL scalar, None => Material() (App.cpp:1044)
    has_tangents: bool = False
    has_uv: tuple = (False, False)
    textures: dict = None        # slot name (TEX_SLOTS) -> (Texture, TextureCoordinateIndex)
    motion_vectors: np.ndarray = None   # skinned meshes: [n,4] half bits, written by the skinning pass (Model.ixx:33)
    skeletal_vertices: np.ndarray = None  # L.SKELETAL_VERTEX, Mesh::SkeletalVertices


@dataclass
class MeshNode:                  # Source/Model.ixx:58-70: one BLAS per mesh node
    meshes: list


@dataclass
class RenderObject:              # one instance of a mesh node with its world transform (3x4, column-vector [R|t])
    node: int
    transform: np.ndarray
    visible: bool = True


def trs(translation=(0, 0, 0), yaw_deg=0.0, scale=(1, 1, 1), pitch_deg=0.0):
    """3x4 object-to-world = T * R_y(yaw) * R_x(pitch) * S (column-vector convention)."""
    cy, sy = math.cos(math.radians(yaw_deg)), math.sin(math.radians(yaw_deg))
    cp, sp = math.cos(math.radians(pitch_deg)), math.sin(math.radians(pitch_deg))
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    m = np.zeros((3, 4))
    m[:, :3] = ry @ rx @ np.diag(scale)
    m[:, 3] = translation
    return m.astype(np.float32)


@dataclass
class Scene:
    nodes: list
    objects: list
    camera: np.ndarray = None
    scene_data: np.ndarray = None
    name: str = "scene"
    # flattened, filled by finalize()
    env_texture: Texture = None                       # EnvironmentLight.Texture (Scene.ixx:56-60): 2D lat-long or cube
    heap: list = field(default_factory=list)          # list of HeapItem
    geometry: list = field(default_factory=list)      # per BLAS geometry: (mesh, heap_v, heap_i)
    blas: list = field(default_factory=list)          # (first_geometry, geometry_count) per node
    object_data: np.ndarray = None
    instance_data: np.ndarray = None
    instance_ids: np.ndarray = None
    instance_masks: np.ndarray = None
    instance_blas: np.ndarray = None

    def finalize(self):
        """Scene::Refresh + App::UpdateScene: InstanceData / ObjectData / descriptor heap."""
        self.heap, self.geometry, self.blas = [], [], []
        mesh_heap, tex_heap = {}, {}
        self._motion_heap = {}

        def tex_descriptor(tex):
            if id(tex) not in tex_heap:
                cube = tex.data.ndim == 4 and tex.data.shape[0] == 6 and tex.data.shape[-1] == 4 and tex.data.shape[1] == tex.data.shape[2]
                h, w = tex.data.shape[-3], tex.data.shape[-2]
                tex_heap[id(tex)] = len(self.heap)
                self.heap.append(HeapItem(np.ascontiguousarray(tex.data), 0, KIND_TEXTURECUBE if cube else KIND_TEXTURE2D, w, h, tex.fmt))
            return tex_heap[id(tex)]

        for node in self.nodes:
            first = len(self.geometry)
            for mesh in node.meshes:
                key = id(mesh)
                if key not in mesh_heap:
                    hv = len(self.heap); self.heap.append(HeapItem(mesh.vertices, 0))
                    hi = len(self.heap); self.heap.append(HeapItem(mesh.indices, mesh.indices.dtype.itemsize))
                    hm = L.NONE
                    if mesh.motion_vectors is not None:
                        hm = len(self.heap); self.heap.append(HeapItem(mesh.motion_vectors, 8))
                    mesh_heap[key] = (hv, hi, hm)
                self.geometry.append((mesh,) + mesh_heap[key][:2])
                self._motion_heap[key] = mesh_heap[key][2]
            self.blas.append((first, len(node.meshes)))
        if self.env_texture is not None and self.scene_data is not None:
            self.scene_data["EnvironmentLightTextureDescriptor"] = tex_descriptor(self.env_texture)
            self.scene_data["IsEnvironmentLightTextureCubeMap"] = 1 if self.heap[tex_heap[id(self.env_texture)]].kind == KIND_TEXTURECUBE else 0
        n_inst = len(self.objects)
        self.instance_data = np.zeros(n_inst, L.INSTANCE_DATA)
        self.instance_ids = np.zeros(n_inst, np.uint32)
        self.instance_masks = np.zeros(n_inst, np.uint32)
        self.instance_blas = np.zeros(n_inst, np.uint32)
        objs = []
        object_index = 0
        for i, ro in enumerate(self.objects):
            node = self.nodes[ro.node]
            self.instance_data[i]["FirstGeometryIndex"] = object_index
            self.instance_data[i]["PreviousObjectToWorld"] = ro.transform
            self.instance_data[i]["ObjectToWorld"] = ro.transform
            self.instance_ids[i] = object_index          # InstanceID = FirstGeometryIndex (Scene.ixx:371)
            self.instance_masks[i] = 0xFF if ro.visible else 0
            self.instance_blas[i] = ro.node
            first, count = self.blas[ro.node]
            for g in range(count):
                mesh, hv, hi = self.geometry[first + g]
                od = np.zeros((), L.OBJECT_DATA)
                od["VertexDesc"]["Stride"] = L.VERTEX.itemsize
                od["VertexDesc"]["Normal"] = 12 if mesh.has_normals else L.NONE
                od["VertexDesc"]["Tangent"] = 18 if mesh.has_tangents else L.NONE
                od["VertexDesc"]["TexCoord"] = (24 if mesh.has_uv[0] else L.NONE, 28 if mesh.has_uv[1] else L.NONE)
                od["MeshDescriptors"]["Vertices"] = hv
                od["MeshDescriptors"]["Indices"] = hi
                od["MeshDescriptors"]["MotionVectors"] = self._motion_heap.get(id(mesh), L.NONE)
                od["Material"] = mesh.material if mesh.material is not None else L.default_material()
                od["TextureMapInfoArray"]["Descriptor"] = L.NONE
                for slot, (tex, uv_index) in (mesh.textures or {}).items():       # App.cpp:1052-1063
                    k = TEX_SLOTS.index(slot)
                    od["TextureMapInfoArray"][k]["Descriptor"] = tex_descriptor(tex)
                    od["TextureMapInfoArray"][k]["TextureCoordinateIndex"] = uv_index
                objs.append(od)
            object_index += count
        self.object_data = np.array(objs, L.OBJECT_DATA) if objs else np.zeros(0, L.OBJECT_DATA)
        return self

    @property
    def triangle_count(self):
        return sum(sum(m.indices.size // 3 for m in self.nodes[o.node].meshes) for o in self.objects)


# ----------------------------------------------------------------------------------------------
# camera (Source/Camera.ixx:38-177 CameraController, Source/App.cpp:540-561)
# ----------------------------------------------------------------------------------------------
def make_camera(position, forward=(0, 0, 1), up=(0, 1, 0), hfov_deg=90.0, aspect=16 / 9,
                near=0.01, far=float("inf"), jitter=(0.0, 0.0)):
    pos = np.asarray(position, np.float64)
    f = np.asarray(forward, np.float64); f /= np.linalg.norm(f)
    r = np.cross(np.asarray(up, np.float64), f); r /= np.linalg.norm(r)   # Right = up x forward (LH)
    u = np.cross(f, r)                                                      # Up = forward x right
    right_len = math.tan(math.radians(hfov_deg) / 2)                        # |Forward| = focus distance = 1
    up_len = right_len / aspect
    cam = np.zeros((), L.CAMERA)
    cam["IsNormalizedDepthReversed"] = 1
    cam["PreviousPosition"] = cam["Position"] = pos
    cam["RightDirection"] = r * right_len
    cam["UpDirection"] = u * up_len
    cam["ForwardDirection"] = f
    cam["NearDepth"], cam["FarDepth"] = near, far
    cam["Jitter"] = jitter
    # XMMatrixLookToLH, row-vector convention
    w2v = np.eye(4)
    w2v[:3, 0], w2v[:3, 1], w2v[:3, 2] = r, u, f
    w2v[3, :3] = [-pos @ r, -pos @ u, -pos @ f]
    # [MathLib spec] float4x4::SetupByHalfFovxInf, LH reversed-Z infinite far plane
    v2p = np.zeros((4, 4))
    v2p[0, 0] = 1 / right_len
    v2p[1, 1] = aspect / right_len
    v2p[2, 3] = 1
    v2p[3, 2] = near
    if math.isfinite(far):   # SetupByHalfFovx, reversed Z
        v2p[2, 2] = -near / (far - near)
        v2p[3, 2] = far * near / (far - near)
    v2w = np.eye(4)
    v2w[0, :3], v2w[1, :3], v2w[2, :3], v2w[3, :3] = r, u, f, pos
    w2p = w2v @ v2p
    p2v = np.linalg.inv(v2p)
    for prefix in ("Previous", ""):
        cam[prefix + "WorldToProjection"] = w2p
        cam[prefix + "ProjectionToView"] = p2v
        cam[prefix + "ViewToWorld"] = v2w
    cam["PreviousWorldToView"] = w2v
    cam["PreviousViewToProjection"] = v2p
    return cam


def make_scene_data(env_color=(0, 0, 0, 1), is_static=True):
    """EnvironmentLightColor.a < 0 selects the procedural sky (Scene.ixx:58 default (0,0,0,-1))."""
    sd = np.zeros((), L.SCENE_DATA)
    sd["IsStatic"] = 1 if is_static else 0
    sd["EnvironmentLightTextureDescriptor"] = L.NONE
    sd["EnvironmentLightColor"] = env_color
    sd["EnvironmentLightTransform"] = np.eye(3, 4)
    return sd


# ----------------------------------------------------------------------------------------------
# mesh builders
# ----------------------------------------------------------------------------------------------
def material(base=(0.73, 0.73, 0.73), emissive=(0, 0, 0), strength=1.0, metallic=0.0, roughness=0.5,
             ior=1.5, transmission=0.0):
    m = L.default_material()
    m["BaseColor"] = tuple(base) + (1.0,)
    m["EmissiveColor"] = emissive
    m["EmissiveStrength"] = strength
    m["Metallic"], m["Roughness"], m["IOR"], m["Transmission"] = metallic, roughness, ior, transmission
    return m


def quad_mesh(p0, p1, p2, p3, normal, mat, has_normals=True, uv_scale=None, textures=None, uv1_scale=None):
    """Two triangles (p0,p1,p2), (p0,p2,p3). uv_scale adds TexCoord0 = corner * scale and a tangent along p0->p1."""
    pos = np.array([p0, p1, p2, p3], np.float32)
    nrm = np.tile(np.asarray(normal, np.float32), (4, 1))
    if uv_scale is None:
        return Mesh(make_vertices(pos, nrm), make_indices([0, 1, 2, 0, 2, 3]), has_normals, mat)
    corners = np.array([(0, 0), (1, 0), (1, 1), (0, 1)], np.float32)
    tan = pos[1] - pos[0]
    tan = np.tile(tan / np.linalg.norm(tan), (4, 1))
    uvs1 = corners * uv1_scale if uv1_scale is not None else None
    return Mesh(make_vertices(pos, nrm, corners * uv_scale, tan, uvs1), make_indices([0, 1, 2, 0, 2, 3]), has_normals, mat,
                has_tangents=True, has_uv=(True, uvs1 is not None), textures=textures)


def box_mesh(mat, has_normals=True):
    """Unit cube [-0.5,0.5]^3, 24 vertices with per-face normals, 12 triangles."""
    pos, nrm, idx = [], [], []
    for axis in range(3):
        for sgn in (-1.0, 1.0):
            n = np.zeros(3); n[axis] = sgn
            a, b = (axis + 1) % 3, (axis + 2) % 3
            corners = []
            for sa, sb in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
                p = np.zeros(3); p[axis] = 0.5 * sgn; p[a] = 0.5 * sa; p[b] = 0.5 * sb
                corners.append(p)
            if sgn < 0:
                corners = corners[::-1]
            base = len(pos)
            pos += corners; nrm += [n] * 4
            idx += [base, base + 1, base + 2, base, base + 2, base + 3]
    return Mesh(make_vertices(np.array(pos), np.array(nrm)), make_indices(idx), has_normals, mat)


def icosphere_mesh(subdiv, mat):
    """Smooth-normal unit sphere (radius 1)."""
    t = (1 + 5 ** 0.5) / 2
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2),
         (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11),
         (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(subdiv):
        cache, nf = {}, []
        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = v[a] + v[b]; v.append(m / np.linalg.norm(m)); cache[k] = len(v) - 1
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    pos = np.array(v)
    return Mesh(make_vertices(pos, pos), make_indices(np.array(f).reshape(-1)), True, mat)


# ----------------------------------------------------------------------------------------------
# BASELINE.json scenes
# ----------------------------------------------------------------------------------------------
def cornell_box(aspect=16 / 9, variant="ggx", glass_sphere=False, has_normals=True, jitter=(0.0, 0.0)):
    """Cornell Box, LH, +Z forward, box = [-1,1]^3 open at z=-1, camera in front of the opening.

    variant "diffuse": every surface uses the reference's default Roughness 0.5 / Metallic 0 / IOR 1.5.
    variant "ggx" (config C2 "full GGX metallic-roughness"): tall box Metallic 1 Roughness 0.05,
    short box Roughness 0.2. glass_sphere adds a Transmission 1 sphere (exercises the third lobe).
    """
    white, red, green = (0.73, 0.73, 0.73), (0.65, 0.05, 0.05), (0.12, 0.45, 0.15)
    hn = has_normals
    nodes = [
        MeshNode([quad_mesh((-1, -1, -1), (-1, -1, 1), (1, -1, 1), (1, -1, -1), (0, 1, 0), material(white), hn)]),   # floor
        MeshNode([quad_mesh((-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1), (0, -1, 0), material(white), hn)]),      # ceiling
        MeshNode([quad_mesh((-1, -1, 1), (-1, 1, 1), (1, 1, 1), (1, -1, 1), (0, 0, -1), material(white), hn)]),      # back
        MeshNode([quad_mesh((-1, -1, -1), (-1, 1, -1), (-1, 1, 1), (-1, -1, 1), (1, 0, 0), material(red), hn)]),     # left
        MeshNode([quad_mesh((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1), (-1, 0, 0), material(green), hn)]),      # right
        MeshNode([quad_mesh((-0.25, 0, -0.25), (0.25, 0, -0.25), (0.25, 0, 0.25), (-0.25, 0, 0.25), (0, -1, 0),
                            material((0.78, 0.78, 0.78), emissive=(1, 1, 1), strength=15.0), hn)]),                # light
    ]
    if variant == "ggx":
        tall = material((0.95, 0.93, 0.88), metallic=1.0, roughness=0.05)
        short = material(white, roughness=0.2)
    else:
        tall, short = material(white), material(white)
    nodes.append(MeshNode([box_mesh(tall, hn)]))
    nodes.append(MeshNode([box_mesh(short, hn)]))
    ident = trs()
    objects = [RenderObject(i, ident) for i in range(5)]
    objects.append(RenderObject(5, trs((0, 0.998, 0.1))))
    objects.append(RenderObject(6, trs((-0.35, -0.4, 0.35), -18.0, (0.6, 1.2, 0.6))))
    objects.append(RenderObject(7, trs((0.35, -0.7, -0.25), 15.0, (0.6, 0.6, 0.6))))
    if glass_sphere:
        nodes.append(MeshNode([icosphere_mesh(2, material((0.98, 0.98, 1.0), roughness=0.05, transmission=1.0))]))
        objects.append(RenderObject(8, trs((0.35, -0.15, -0.25), 0.0, (0.25, 0.25, 0.25))))
    cam = make_camera((0, 0, -1.95), hfov_deg=90.0, aspect=aspect, jitter=jitter)
    return Scene(nodes, objects, cam, make_scene_data((0, 0, 0, 1)), name="cornell_" + variant).finalize()


def _checker(n, cells, c0, c1, alpha0=255, alpha1=255):
    y, x = np.mgrid[0:n, 0:n]
    m = (((x * cells) // n + (y * cells) // n) % 2).astype(bool)
    img = np.zeros((n, n, 4), np.uint8)
    img[~m] = tuple(c0) + (alpha0,)
    img[m] = tuple(c1) + (alpha1,)
    return img


def cornell_box_textured(aspect=16 / 9, env="latlong", seed=7):
    """Cornell box exercising every texture slot of Material.ixx:22-33 plus alpha-tested geometry and an
    environment texture: checker base colour (sRGB) + normal map + packed metallic-roughness on the floor,
    separate metallic / roughness maps on the back wall, emissive texture on a panel, transmission texture on a pane,
    an alpha-masked (AlphaMode Mask) lattice in front of the light, lat-long or cube HDR environment."""
    rng = np.random.default_rng(seed)
    white, red, green = (0.73, 0.73, 0.73), (0.65, 0.05, 0.05), (0.12, 0.45, 0.15)
    n = 32
    base = Texture(_checker(n, 8, (230, 230, 230), (120, 60, 30)), srgb=True)
    nm = np.zeros((n, n, 4), np.uint8)
    ang = rng.random((n, n)) * 2 * np.pi; tilt = rng.random((n, n)) * 0.35
    nm[..., 0] = np.clip((np.cos(ang) * tilt * 0.5 + 0.5) * 255, 0, 255)
    nm[..., 1] = np.clip((np.sin(ang) * tilt * 0.5 + 0.5) * 255, 0, 255)
    nm[..., 2] = 255; nm[..., 3] = 255
    normal_map = Texture(nm)
    mr = np.zeros((n, n, 4), np.uint8)
    mr[..., 1] = rng.integers(40, 255, (n, n)); mr[..., 2] = (rng.random((n, n)) > 0.7) * 255; mr[..., 3] = 255
    metal_rough = Texture(mr)
    single = Texture(np.repeat(rng.integers(30, 255, (n, n, 1)), 4, -1).astype(np.uint8))
    single2 = Texture(np.repeat(rng.integers(0, 255, (n, n, 1)), 4, -1).astype(np.uint8))
    emis = Texture(_checker(n, 4, (255, 180, 60), (20, 20, 80)), srgb=True)
    lattice = Texture(_checker(n, 6, (200, 200, 40), (10, 10, 10), alpha0=255, alpha1=0), srgb=True)
    trans = Texture(_checker(n, 2, (255, 255, 255), (60, 60, 60)))
    fl = material(white, metallic=1.0, roughness=1.0)
    bw = material(white, metallic=0.8, roughness=0.9)
    panel = material((0.2, 0.2, 0.2), emissive=(1, 1, 1), strength=4.0)
    mask = material((1, 1, 1)); mask["AlphaMode"] = 1; mask["AlphaCutoff"] = 0.5
    pane = material((0.95, 0.98, 1.0), roughness=0.08, transmission=1.0)
    nodes = [
        MeshNode([quad_mesh((-1, -1, -1), (-1, -1, 1), (1, -1, 1), (1, -1, -1), (0, 1, 0), fl, uv_scale=3.0, uv1_scale=1.0,
                            textures={"BaseColor": (base, 0), "Normal": (normal_map, 0), "MetallicRoughness": (metal_rough, 1)})]),
        MeshNode([quad_mesh((-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1), (0, -1, 0), material(white))]),
        MeshNode([quad_mesh((-1, -1, 1), (-1, 1, 1), (1, 1, 1), (1, -1, 1), (0, 0, -1), bw, uv_scale=2.0,
                            textures={"Metallic": (single, 0), "Roughness": (single2, 0), "BaseColor": (base, 0)})]),
        MeshNode([quad_mesh((-1, -1, -1), (-1, 1, -1), (-1, 1, 1), (-1, -1, 1), (1, 0, 0), material(red))]),
        MeshNode([quad_mesh((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1), (-1, 0, 0), material(green))]),
        MeshNode([quad_mesh((-0.25, 0, -0.25), (0.25, 0, -0.25), (0.25, 0, 0.25), (-0.25, 0, 0.25), (0, -1, 0),
                            material((0.78, 0.78, 0.78), emissive=(1, 1, 1), strength=15.0))]),
        MeshNode([quad_mesh((-0.5, -0.5, 0), (0.5, -0.5, 0), (0.5, 0.5, 0), (-0.5, 0.5, 0), (0, 0, -1), panel, uv_scale=1.0,
                            textures={"EmissiveColor": (emis, 0)})]),
        MeshNode([quad_mesh((-0.5, 0, -0.5), (0.5, 0, -0.5), (0.5, 0, 0.5), (-0.5, 0, 0.5), (0, -1, 0), mask, uv_scale=1.0,
                            textures={"BaseColor": (lattice, 0)})]),
        MeshNode([quad_mesh((-0.5, -0.5, 0), (0.5, -0.5, 0), (0.5, 0.5, 0), (-0.5, 0.5, 0), (0, 0, -1), pane, uv_scale=1.0,
                            textures={"Transmission": (trans, 0)})]),
    ]
    ident = trs()
    objects = [RenderObject(i, ident) for i in range(5)]
    objects.append(RenderObject(5, trs((0, 0.998, 0.1))))
    objects.append(RenderObject(6, trs((-0.4, -0.2, 0.95), 0.0, (0.8, 0.8, 1))))
    objects.append(RenderObject(7, trs((0, 0.6, 0.1), 0.0, (1.2, 1, 1.2))))
    objects.append(RenderObject(8, trs((0.45, -0.45, -0.2), 25.0, (0.9, 1.0, 1))))
    sc = Scene(nodes, objects, make_camera((0, 0, -1.95), hfov_deg=90.0, aspect=aspect), make_scene_data((0, 0, 0, 1)),
               name="cornell_textured_" + str(env))
    if env == "latlong":
        h, w = 16, 32
        e = np.zeros((h, w, 4), np.float32)
        e[..., :3] = (0.2 + rng.random((h, w, 3)) * 0.6) * np.linspace(2.0, 0.2, h)[:, None, None]
        e[2, 5, :3] = 40.0                                      # a small "sun"
        sc.env_texture = Texture(e)
    elif env == "cube":
        e = (0.1 + rng.random((6, 8, 8, 4)) * 0.9).astype(np.float32)
        e[2] *= 3.0
        sc.env_texture = Texture(e)
    return sc.finalize()


def skinned_bar(segments=12, seed=3):
    """A square column along +Y bound to two joints (bottom, top) with height-blended weights, plus its skeletal vertex
    buffer (Model.ixx:31-34, Vertex.ixx:52-57) and a motion-vector buffer: the input of SkeletalMeshSkinning."""
    pos, nrm, tan, idx = [], [], [], []
    for side in range(4):
        a = side * math.pi / 2
        n = np.array([math.cos(a), 0.0, math.sin(a)]); t = np.array([-math.sin(a), 0.0, math.cos(a)])
        base = len(pos)
        for k in range(segments + 1):
            y = k / segments
            for e in (-1, 1):
                pos.append(n * 0.12 + t * 0.12 * e + np.array([0, y, 0])); nrm.append(n); tan.append(t)
        for k in range(segments):
            i0 = base + 2 * k
            idx += [i0, i0 + 2, i0 + 1, i0 + 1, i0 + 2, i0 + 3]
    pos = np.array(pos, np.float32)
    vb = make_vertices(pos, np.array(nrm), None, np.array(tan))
    sk = np.zeros(len(pos), L.SKELETAL_VERTEX)
    sk["Position"] = pos; sk["Normal"] = vb["Normal"]; sk["Tangent"] = vb["Tangent"]
    sk["Joints"] = (0, 1, 0, 0)
    h = pos[:, 1]
    sk["Weights"] = np.stack([1 - h, h, np.zeros_like(h), np.zeros_like(h)], 1)     # 4th weight is implied: 1 - w0 - w1 - w2
    mesh = Mesh(vb, make_indices(idx), True, material((0.8, 0.5, 0.2), roughness=0.4), has_tangents=True,
                motion_vectors=np.zeros((len(pos), 4), np.uint16), skeletal_vertices=sk)
    return mesh


def bar_pose(angle_deg, lift=0.0):
    """joint transforms (row-major 3x4): joint 0 fixed, joint 1 rotated about Z around the bar's base and lifted."""
    c, s_ = math.cos(math.radians(angle_deg)), math.sin(math.radians(angle_deg))
    j0 = np.eye(3, 4)
    j1 = np.array([[c, -s_, 0, 0], [s_, c, 0, lift], [0, 0, 1, 0]])
    return np.stack([j0, j1]).astype(np.float32)


def dynamic_scene(aspect=16 / 9):
    """floor + light + a skinned bar: the smallest scene that exercises skinning, BLAS update, TLAS rebuild and
    per-vertex motion vectors (SceneData.IsStatic = 0)."""
    white = (0.73, 0.73, 0.73)
    bar = skinned_bar()
    nodes = [MeshNode([quad_mesh((-2, 0, -2), (-2, 0, 2), (2, 0, 2), (2, 0, -2), (0, 1, 0), material(white))]),
             MeshNode([quad_mesh((-0.5, 0, -0.5), (0.5, 0, -0.5), (0.5, 0, 0.5), (-0.5, 0, 0.5), (0, -1, 0),
                                 material((0.8, 0.8, 0.8), emissive=(1, 1, 1), strength=10.0))]),
             MeshNode([bar])]
    objects = [RenderObject(0, trs()), RenderObject(1, trs((0, 2.2, 0))), RenderObject(2, trs((0.1, 0, 0.2), 20.0, (1, 1.2, 1)))]
    cam = make_camera((0, 0.9, -2.2), forward=(0, -0.1, 1), hfov_deg=70.0, aspect=aspect)
    return Scene(nodes, objects, cam, make_scene_data((0.05, 0.06, 0.08, 1), is_static=False), name="dynamic").finalize()


def _material_textures(rng, size, masked):
    """Procedural RGBA8 texture set of one glTF-style material (Source/GLTFHelpers.ixx:370-428): base colour (sRGB; alpha is a lattice with
    a quarter of its cells cut out when the material is alpha-masked), tangent-space normal map, packed metallic-roughness (G = roughness,
    B = metallic, as GLTFHelpers.ixx reads them). Smooth low-frequency patterns plus texel noise, so neighbouring hits fetch neighbouring texels."""
    y, x = np.mgrid[0:size, 0:size].astype(np.float32) / size
    f = rng.integers(2, 9, 4)
    ph = rng.random(4) * 6.28
    wave = 0.5 + 0.5 * np.sin(2 * np.pi * f[0] * x + ph[0]) * np.cos(2 * np.pi * f[1] * y + ph[1])
    tint = 0.35 + 0.6 * rng.random(3)
    base = np.zeros((size, size, 4), np.uint8)
    noise = rng.integers(0, 24, (size, size), dtype=np.uint8)
    for c in range(3):
        base[..., c] = np.clip((0.35 + 0.65 * wave) * tint[c] * 255.0, 0, 231).astype(np.uint8) + noise
    base[..., 3] = 255
    if masked:
        cells = 48
        cut = (((x * cells).astype(np.int32) % 2 == 0) & ((y * cells).astype(np.int32) % 2 == 0))
        base[..., 3] = np.where(cut, 0, 255)
    nm = np.zeros((size, size, 4), np.uint8)
    nm[..., 0] = np.clip((0.5 + 0.18 * np.sin(2 * np.pi * f[2] * 4 * x + ph[2])) * 255.0, 0, 255)
    nm[..., 1] = np.clip((0.5 + 0.18 * np.cos(2 * np.pi * f[3] * 4 * y + ph[3])) * 255.0, 0, 255)
    nm[..., 2] = 255; nm[..., 3] = 255
    mr = np.zeros((size, size, 4), np.uint8)
    mr[..., 1] = np.clip((0.15 + 0.7 * wave) * 255.0, 0, 255)
    mr[..., 2] = np.where(np.sin(2 * np.pi * f[0] * 0.5 * (x + y) + ph[1]) > 0.6, 255, 0)
    mr[..., 3] = 255
    return Texture(base, srgb=True), Texture(nm), Texture(mr)


def sponza_scale(n_side=354, seed=1234, aspect=16 / 9, n_materials=24, textured=False, texture_size=1024, masked_strips=(3, 12, 20)):
    """~250k-triangle displaced terrain + column grid in ONE BLAS (config C3); materials vary per strip.
    textured=True ("Sponza-scale glTF": the same mesh as a glTF import would deliver it): every vertex carries TexCoord0 and a tangent, every
    one of the 24 materials a base-colour (sRGB), a normal and a metallic-roughness texture of texture_size^2 RGBA8 texels; the strips
    masked_strips (3 of 24: 11.9 % of the triangles) are AlphaMode Mask with a lattice in its base-colour alpha: not FLAG_OPAQUE in the
    bottom level, so every candidate hit on it runs the alpha test inside the traversal (RaytracingHelpers.hlsli:19-44)."""
    rng = np.random.default_rng(seed)
    trng = np.random.default_rng(seed + 17)                  # textures draw from their own stream: the geometry is that of the untextured scene
    strips = n_materials
    rows_per = max(1, n_side // strips)
    meshes = []
    xs = np.linspace(-4, 4, n_side + 1)
    zs = np.linspace(-1, 9, n_side + 1)
    gx, gz = np.meshgrid(xs, zs, indexing="xy")
    h = (0.15 * np.sin(gx * 2.1) * np.cos(gz * 1.7) + 0.05 * rng.standard_normal(gx.shape)).astype(np.float64)
    # arcade: raise two side walls and periodic columns
    h += 2.5 * (np.abs(gx) > 3.2)
    h += 1.8 * ((np.abs(np.abs(gx) - 2.0) < 0.12) & (np.mod(gz, 1.0) < 0.24))
    h -= 1.0
    r0 = 0
    for s in range(strips):
        r1 = n_side if s == strips - 1 else r0 + rows_per
        if r1 <= r0:
            break
        sub = np.s_[r0:r1 + 1, :]
        px, pz, py = gx[sub], gz[sub], h[sub]
        pos = np.stack([px, py, pz], -1).reshape(-1, 3)
        # normals from central differences
        dy_dx = np.gradient(h, xs, axis=1)[sub]; dy_dz = np.gradient(h, zs, axis=0)[sub]
        nrm = np.stack([-dy_dx, np.ones_like(dy_dx), -dy_dz], -1).reshape(-1, 3)
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        nr, nc = r1 - r0 + 1, n_side + 1
        i0 = (np.arange(nr - 1)[:, None] * nc + np.arange(nc - 1)[None, :]).reshape(-1)
        idx = np.stack([i0, i0 + nc, i0 + 1, i0 + 1, i0 + nc, i0 + nc + 1], -1).reshape(-1)
        col = tuple(0.25 + 0.6 * rng.random(3))
        mat = material(col, metallic=float(s % 5 == 0), roughness=float(0.08 + 0.8 * rng.random()))
        if not textured:
            meshes.append(Mesh(make_vertices(pos, nrm), make_indices(idx), True, mat))
        else:
            masked = s in masked_strips
            tex = _material_textures(trng, texture_size, masked)
            mat = material((1.0, 1.0, 1.0), metallic=1.0, roughness=1.0)     # glTF factors of 1: the textures carry the values
            if masked:
                mat["AlphaMode"] = 1; mat["AlphaCutoff"] = 0.5
            uv = np.stack([px * 1.5, pz * 1.5], -1).reshape(-1, 2)
            tan = np.stack([np.ones_like(dy_dx), dy_dx, np.zeros_like(dy_dx)], -1).reshape(-1, 3)
            tan /= np.linalg.norm(tan, axis=1, keepdims=True)
            meshes.append(Mesh(make_vertices(pos, nrm, uv, tan), make_indices(idx), True, mat, has_tangents=True, has_uv=(True, False),
                               textures={"BaseColor": (tex[0], 0), "Normal": (tex[1], 0), "MetallicRoughness": (tex[2], 0)}))
        r0 = r1
    light = quad_mesh((-1.5, 0, -1.5), (1.5, 0, -1.5), (1.5, 0, 1.5), (-1.5, 0, 1.5), (0, -1, 0),
                      material((0.8, 0.8, 0.8), emissive=(1, 0.95, 0.9), strength=20.0))
    nodes = [MeshNode(meshes), MeshNode([light])]
    objects = [RenderObject(0, trs()), RenderObject(1, trs((0, 3.0, 4.0)))]
    cam = make_camera((0, 0.4, -0.5), forward=(0, -0.12, 1), hfov_deg=90.0, aspect=aspect)
    return Scene(nodes, objects, cam, make_scene_data((0, 0, 0, -1)), name="sponza_scale_textured" if textured else "sponza_scale").finalize()


def instanced_grid(n=100, seed=42, aspect=16 / 9, subdiv=2):
    """n*n instances of one icosphere-derived mesh on a jittered grid + ground + light (config C5)."""
    rng = np.random.default_rng(seed)
    blob = icosphere_mesh(subdiv, material((0.8, 0.6, 0.3), metallic=1.0, roughness=0.25))
    ground = quad_mesh((-1, 0, -1), (-1, 0, 1), (1, 0, 1), (1, 0, -1), (0, 1, 0), material((0.6, 0.6, 0.6)))
    light = quad_mesh((-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1), (0, -1, 0),
                      material((0.8, 0.8, 0.8), emissive=(1, 1, 1), strength=12.0))
    nodes = [MeshNode([blob]), MeshNode([ground]), MeshNode([light])]
    objects = []
    span = 20.0
    for iz in range(n):
        for ix in range(n):
            x = (ix + 0.5) / n * span - span / 2 + (rng.random() - 0.5) * 0.08
            z = (iz + 0.5) / n * span + (rng.random() - 0.5) * 0.08
            s = 0.04 + 0.04 * rng.random()
            objects.append(RenderObject(0, trs((x, s * 0.9, z), rng.random() * 360.0, (s, s * (0.7 + 0.6 * rng.random()), s),
                                                pitch_deg=rng.random() * 40 - 20)))
    objects.append(RenderObject(1, trs((0, 0, span / 2), 0, (span, 1, span))))
    objects.append(RenderObject(2, trs((0, 6.0, span / 2), 0, (4, 1, 4))))
    cam = make_camera((0, 1.2, -0.5), forward=(0, -0.25, 1), hfov_deg=90.0, aspect=aspect)
    return Scene(nodes, objects, cam, make_scene_data((0, 0, 0, -1)), name="instanced_grid").finalize()


def dynamic_instanced(n=32, aspect=16 / 9, segments=256):
    """n*n static instances + ground + light (instanced_grid) and one skinned column in front of the camera: the per-frame work of a
    dynamic scene -- skinning, bottom-level update of the skinned mesh node, top-level rebuild over ~1 k instances (bench.py
    --workload dynamic)."""
    sc = instanced_grid(n=n, aspect=aspect)
    bar = skinned_bar(segments=segments)
    sc.nodes.append(MeshNode([bar]))
    sc.objects.append(RenderObject(len(sc.nodes) - 1, trs((0.0, 0.0, 1.2), 15.0, (1.0, 1.6, 1.0))))
    sc.scene_data = make_scene_data((0.05, 0.06, 0.08, 1), is_static=False)
    sc.name = "dynamic_instanced"
    return sc.finalize()


def graphics_settings(width, height, spp=1, bounces=8, frame_index=0, russian_roulette=True, ext_flags=0,
                      throughput_threshold=1e-3):
    """Raytracing::GraphicsSettings with the reference defaults (MyAppData.h:182-188, Raytracing.ixx:33)."""
    gs = np.zeros((), L.GRAPHICS_SETTINGS)
    gs["RenderSize"] = (width, height)
    gs["FrameIndex"], gs["Bounces"], gs["SamplesPerPixel"] = frame_index, bounces, spp
    gs["ThroughputThreshold"] = throughput_threshold
    gs["IsRussianRouletteEnabled"] = 1 if russian_roulette else 0
    gs["ExtFlags"] = ext_flags
    return gs


def alloc_gbuffer(width, height):
    """Host G-buffer in the reference's texture formats (Source/App.cpp:438-455)."""
    return {k: np.zeros((height, width, c), dt) for k, (dt, c) in L.GBUFFER_FORMATS.items()}
