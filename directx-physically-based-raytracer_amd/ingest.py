"""Scene-JSON + glTF 2.0 ingest in the reference's schema and conventions.

What it mirrors (host side of the path, upstream of the acceleration-structure build):
  scene descriptor   Source/MyScene.ixx:33-90, Source/JSONConverters.ixx:12-33, Source/Scene.ixx:33-73
                     {Camera{Position,Rotation}, EnvironmentLight{Color,Rotation,Texture}, Models{name:path},
                      RenderObjects[{Name,Transform{Translation,Rotation,Scale},IsVisible,Model}]}
  glTF loader        Source/GLTFHelpers.ixx:142-537 (ProcessPrimitive, LoadModel): triangles only, index array written
                     BACKWARDS when flipWindingOrder (always, Scene.ixx:90), u16 indices iff count <= 65535,
                     tangents ALWAYS recomputed when NORMAL + TEXCOORD_0 exist (the loader asks for an attribute called
                     "Tangent", which glTF never has, :193), materials incl. KHR_materials_emissive_strength / ior /
                     transmission, base colour + emissive textures forced to sRGB, normal texture only with tangents
  instance transform Source/Scene.ixx:195-231: world = GlobalTransform * Scale(1,1,-1) * RenderObject.Transform()
                     (SimpleMath row-vector convention), AffineTransform = Scale * Rotation * Translation (Math.ixx:17-19)
Texture *files* are decoded with PIL (PNG/JPEG); DDS/EXR/HDR codecs stay out of scope (TextureHelpers.ixx).
[DirectXMesh spec] ComputeTangentFrame is an un-vendored dependency: restated as Lengyel's per-vertex accumulation with
Gram-Schmidt against the normal.
"""
import base64
import io
import json
import math
import os
import struct

import numpy as np

from . import layouts as L
from . import scenes as S

_COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


# ----------------------------------------------------------------------------------------------
# SimpleMath / DirectXMath conventions (row vectors: v' = v M)
# ----------------------------------------------------------------------------------------------
def rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0, 0], [0, c, s, 0], [0, -s, c, 0], [0, 0, 0, 1]], np.float64)


def rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, -s, 0], [0, 1, 0, 0], [s, 0, c, 0], [0, 0, 0, 1]], np.float64)


def rot_z(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, s, 0, 0], [-s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float64)


def matrix_from_quaternion(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y + z * w), 2 * (x * z - y * w), 0],
                     [2 * (x * y - z * w), 1 - 2 * (x * x + z * z), 2 * (y * z + x * w), 0],
                     [2 * (x * z + y * w), 2 * (y * z - x * w), 1 - 2 * (x * x + y * y), 0],
                     [0, 0, 0, 1]], np.float64)


def rotation_from_json(j):
    """JSONConverters.ixx:18-26: {Yaw,Pitch,Roll} in degrees -> CreateFromYawPitchRoll(yaw, -pitch, -roll) (roll about Z
    first, then pitch about X, then yaw about Y); all three zero -> raw quaternion {X,Y,Z,W} (default identity)."""
    if j is None:
        return np.eye(4)
    yaw, pitch, roll = float(j.get("Yaw", 0)), float(j.get("Pitch", 0)), float(j.get("Roll", 0))
    if yaw == 0 and pitch == 0 and roll == 0:
        q = (float(j.get("X", 0)), float(j.get("Y", 0)), float(j.get("Z", 0)), float(j.get("W", 1)))
        n = math.sqrt(sum(c * c for c in q)) or 1.0
        return matrix_from_quaternion(tuple(c / n for c in q))
    return rot_z(math.radians(-roll)) @ rot_x(math.radians(-pitch)) @ rot_y(math.radians(yaw))


def vec3_from_json(j, default):
    if j is None:
        return np.array(default, np.float64)
    return np.array([float(j.get("X", default[0])), float(j.get("Y", default[1])), float(j.get("Z", default[2]))], np.float64)


def affine_from_json(j):
    """Math::AffineTransform::operator(): Scale * Rotation * Translation (row-vector)."""
    j = j or {}
    s = vec3_from_json(j.get("Scale"), (1, 1, 1)); t = vec3_from_json(j.get("Translation"), (0, 0, 0))
    m = np.diag([s[0], s[1], s[2], 1.0]) @ rotation_from_json(j.get("Rotation"))
    tr = np.eye(4); tr[3, :3] = t
    return m @ tr


def store_float3x4(m_row):
    """XMStoreFloat3x4: the column-vector affine [R|t] = top 3 rows of the transpose."""
    return np.ascontiguousarray(m_row.T[:3, :], np.float32)


# ----------------------------------------------------------------------------------------------
# scene descriptor
# ----------------------------------------------------------------------------------------------
def load_scene_desc(path):
    with open(path) as f:
        j = json.load(f)
    base = os.path.dirname(os.path.abspath(path))

    def resolve(p):
        return p if (not p or os.path.isabs(p)) else os.path.join(base, p)

    env = j.get("EnvironmentLight", {}) or {}
    col = env.get("Color") or {}
    desc = {
        "Camera": {"Position": vec3_from_json((j.get("Camera") or {}).get("Position"), (0, 0, 0)),
                   "Rotation": rotation_from_json((j.get("Camera") or {}).get("Rotation"))},
        "EnvironmentLight": {"Color": (float(col.get("R", 0)), float(col.get("G", 0)), float(col.get("B", 0)), float(col.get("A", -1))),
                             "Rotation": rotation_from_json(env.get("Rotation")), "Texture": resolve(env.get("Texture", ""))},
        "Models": {k: resolve(v) for k, v in (j.get("Models") or {}).items()},
        "RenderObjects": [],
    }
    for ro in j.get("RenderObjects") or []:
        model = ro.get("Model", "")
        if model and model not in desc["Models"]:                      # MyScene.ixx:57-70
            name = ("RenderObject " + ro["Name"]) if ro.get("Name") else "Unnamed RenderObject"
            raise RuntimeError(f"{path}: {name}: Models {model} not found")
        desc["RenderObjects"].append({"Name": ro.get("Name", ""), "Transform": affine_from_json(ro.get("Transform")),
                                      "IsVisible": bool(ro.get("IsVisible", True)), "Model": model})
    return desc


# ----------------------------------------------------------------------------------------------
# glTF
# ----------------------------------------------------------------------------------------------
class _Asset:
    def __init__(self, path):
        self.dir = os.path.dirname(os.path.abspath(path))
        raw = open(path, "rb").read()
        self.bin_chunk = None
        if raw[:4] == b"glTF":
            _, _, length = struct.unpack_from("<III", raw, 0)
            off = 12
            while off < length:
                clen, ctype = struct.unpack_from("<II", raw, off)
                data = raw[off + 8: off + 8 + clen]
                if ctype == 0x4E4F534A:
                    self.j = json.loads(data.decode("utf-8"))
                elif ctype == 0x004E4942:
                    self.bin_chunk = data
                off += 8 + clen
        else:
            self.j = json.loads(raw.decode("utf-8"))
        self.buffers = {}

    def buffer(self, i):
        if i not in self.buffers:
            b = self.j["buffers"][i]
            uri = b.get("uri")
            if uri is None:
                self.buffers[i] = self.bin_chunk
            elif uri.startswith("data:"):
                self.buffers[i] = base64.b64decode(uri.split(",", 1)[1])
            else:
                self.buffers[i] = open(os.path.join(self.dir, uri), "rb").read()
        return self.buffers[i]

    def view_bytes(self, vi):
        v = self.j["bufferViews"][vi]
        b = self.buffer(v["buffer"])
        off = v.get("byteOffset", 0)
        return b[off: off + v["byteLength"]], v.get("byteStride", 0)

    def accessor(self, ai):
        a = self.j["accessors"][ai]
        dt = np.dtype(_COMPONENT[a["componentType"]]); nc = _NCOMP[a["type"]]
        count = a["count"]
        data, stride = self.view_bytes(a["bufferView"])
        off = a.get("byteOffset", 0)
        elem = dt.itemsize * nc
        if stride in (0, elem):
            arr = np.frombuffer(data, dt, count * nc, off).reshape(count, nc)
        else:
            arr = np.stack([np.frombuffer(data, dt, nc, off + i * stride) for i in range(count)])
        if a.get("normalized") and dt.kind in "iu":
            arr = arr.astype(np.float32) / float(np.iinfo(dt).max)
        return arr

    def image(self, ii):
        from PIL import Image
        im = self.j["images"][ii]
        if "uri" in im:
            uri = im["uri"]
            data = base64.b64decode(uri.split(",", 1)[1]) if uri.startswith("data:") else open(os.path.join(self.dir, uri), "rb").read()
        else:
            data, _ = self.view_bytes(im["bufferView"])
        return np.ascontiguousarray(np.asarray(Image.open(io.BytesIO(data)).convert("RGBA"), np.uint8))


def compute_tangents(positions, normals, uvs, indices):
    """[DirectXMesh spec] ComputeTangentFrame (tangent output only)."""
    tan = np.zeros((len(positions), 3), np.float64)
    tri = np.asarray(indices, np.int64).reshape(-1, 3)
    p = positions.astype(np.float64); t = uvs.astype(np.float64)
    for a, b, c in tri:
        e1, e2 = p[b] - p[a], p[c] - p[a]
        du1, dv1 = t[b] - t[a]; du2, dv2 = t[c] - t[a]
        det = du1 * dv2 - du2 * dv1
        if abs(det) < 1e-20:
            continue
        sdir = (e1 * dv2 - e2 * dv1) / det
        tan[a] += sdir; tan[b] += sdir; tan[c] += sdir
    n = normals.astype(np.float64)
    tan = tan - n * (n * tan).sum(1, keepdims=True)
    ln = np.linalg.norm(tan, axis=1, keepdims=True)
    fallback = np.cross(n, np.array([0.0, 1.0, 0.0]))
    fl = np.linalg.norm(fallback, axis=1, keepdims=True)
    fallback = np.where(fl > 1e-8, fallback / np.maximum(fl, 1e-30), np.array([1.0, 0.0, 0.0]))
    return np.where(ln > 1e-12, tan / np.maximum(ln, 1e-30), fallback)


def _node_matrix(node):
    """glTF column-vector local matrix."""
    if "matrix" in node:
        return np.array(node["matrix"], np.float64).reshape(4, 4).T
    t = np.eye(4); r = np.eye(4); s = np.eye(4)
    if "translation" in node:
        t[:3, 3] = node["translation"]
    if "rotation" in node:
        r = matrix_from_quaternion(node["rotation"]).T          # column-vector form
    if "scale" in node:
        s = np.diag(list(node["scale"]) + [1.0])
    return t @ r @ s


def load_model(path, flip_winding_order=True):
    """GLTFHelpers::LoadModel. Returns a list of (MeshNode, GlobalTransform row-vector 4x4)."""
    asset = _Asset(path)
    j = asset.j
    tex_cache = {}

    def texture_for(info, force_srgb):
        tex = j["textures"][info["index"]]
        src = tex.get("source")
        key = (src, force_srgb)
        if key not in tex_cache:
            tex_cache[key] = S.Texture(asset.image(src), srgb=force_srgb)
        return tex_cache[key]

    def process_primitive(prim):
        if prim.get("mode", 4) != 4 or "POSITION" not in prim["attributes"] or "indices" not in prim:
            return None                                                   # :150-152,169-171,191-193
        attrs = prim["attributes"]
        pos = asset.accessor(attrs["POSITION"]).astype(np.float32)
        idx = asset.accessor(prim["indices"]).reshape(-1).astype(np.int64)
        if flip_winding_order:
            idx = idx[::-1].copy()                                        # slot count-1-i <- index i (:179)
        indices = idx.astype(np.uint16 if idx.size <= 65535 else np.uint32)
        has_uv = [False, False]; uvs = [None, None]
        for i in (0, 1):
            if f"TEXCOORD_{i}" in attrs:
                uvs[i] = asset.accessor(attrs[f"TEXCOORD_{i}"]).astype(np.float32); has_uv[i] = True
        nrm = tan = None
        if "NORMAL" in attrs:
            nrm = asset.accessor(attrs["NORMAL"]).astype(np.float32)
            if uvs[0] is not None:                                        # "Tangent" is never found -> always recomputed (:251-275)
                tan = compute_tangents(pos, nrm, uvs[0], indices)
        vb = S.make_vertices(pos, nrm, uvs[0], tan, uvs[1])
        mesh = S.Mesh(vb, indices, has_normals=nrm is not None, material=None, has_tangents=tan is not None, has_uv=tuple(has_uv))
        if "material" in prim:
            m = j["materials"][prim["material"]]
            pbr = m.get("pbrMetallicRoughness", {})
            ext = m.get("extensions", {})
            mat = L.default_material()
            mat["BaseColor"] = tuple(pbr.get("baseColorFactor", (1, 1, 1, 1)))
            mat["EmissiveStrength"] = ext.get("KHR_materials_emissive_strength", {}).get("emissiveStrength", 1.0)
            mat["EmissiveColor"] = tuple(m.get("emissiveFactor", (0, 0, 0)))
            mat["Metallic"] = pbr.get("metallicFactor", 1.0)
            mat["Roughness"] = pbr.get("roughnessFactor", 1.0)
            mat["IOR"] = ext.get("KHR_materials_ior", {}).get("ior", 1.5)
            mat["AlphaMode"] = {"OPAQUE": 0, "MASK": 1, "BLEND": 2}[m.get("alphaMode", "OPAQUE")]
            mat["AlphaCutoff"] = m.get("alphaCutoff", 0.5)
            tr = ext.get("KHR_materials_transmission")
            if tr:
                mat["Transmission"] = tr.get("transmissionFactor", 0.0)
            mesh.material = mat
            if has_uv[0] or has_uv[1]:                                    # :370-428
                slots = {"BaseColor": (pbr.get("baseColorTexture"), True), "EmissiveColor": (m.get("emissiveTexture"), True),
                         "MetallicRoughness": (pbr.get("metallicRoughnessTexture"), False),
                         "Transmission": ((tr or {}).get("transmissionTexture"), False),
                         "Normal": (m.get("normalTexture") if tan is not None else None, False)}
                textures = {}
                for slot, (info, srgb) in slots.items():
                    if info is not None and info.get("texCoord", 0) < 2 and has_uv[info.get("texCoord", 0)]:
                        textures[slot] = (texture_for(info, srgb), info.get("texCoord", 0))
                mesh.textures = textures or None
        return mesh

    out = []
    scene = j["scenes"][j.get("scene", 0)]

    def visit(ni, parent):
        node = j["nodes"][ni]
        m = parent @ _node_matrix(node)
        if "mesh" in node:
            prims = j["meshes"][node["mesh"]].get("primitives", [])
            if prims:
                meshes = [x for x in (process_primitive(p) for p in prims) if x is not None]
                out.append((S.MeshNode(meshes), m.T.copy()))            # GlobalTransform reinterpreted as a row-vector Matrix
        for c in node.get("children", []):
            visit(c, m)

    for ni in scene.get("nodes", []):
        visit(ni, np.eye(4))
    return out


# ----------------------------------------------------------------------------------------------
# Scene::Load + Refresh
# ----------------------------------------------------------------------------------------------
def camera_from_desc(desc, aspect, hfov_deg=90.0, near=0.01):
    """App::ResetCamera: CameraController::SetPosition / SetRotation (Source/Camera.ixx:84-97), default lens."""
    r = desc["Camera"]["Rotation"]
    fwd = np.array([0, 0, 1, 0]) @ r
    right = np.array([1, 0, 0, 0]) @ r
    up = np.cross(fwd[:3], right[:3])
    return S.make_camera(desc["Camera"]["Position"], forward=fwd[:3], up=up, hfov_deg=hfov_deg, aspect=aspect, near=near)


def load_scene(path, aspect=16 / 9):
    desc = load_scene_desc(path)
    models = {}
    nodes, objects = [], []
    zflip = np.diag([1.0, 1.0, -1.0, 1.0])
    for ro in desc["RenderObjects"]:
        if not ro["Model"]:
            continue
        if ro["Model"] not in models:
            models[ro["Model"]] = load_model(desc["Models"][ro["Model"]], flip_winding_order=True)    # Scene.ixx:90
        for mesh_node, g in models[ro["Model"]]:
            key = id(mesh_node)
            if key not in [id(n) for n in nodes]:
                nodes.append(mesh_node)
            world = g @ zflip @ ro["Transform"]                           # Scene.ixx:199-214
            objects.append(S.RenderObject([id(n) for n in nodes].index(key), store_float3x4(world), ro["IsVisible"]))
    sd = S.make_scene_data(desc["EnvironmentLight"]["Color"])
    sd["EnvironmentLightTransform"] = store_float3x4(desc["EnvironmentLight"]["Rotation"])     # App.cpp:1019
    sc = S.Scene(nodes, objects, camera_from_desc(desc, aspect), sd, name=os.path.basename(path))
    tex = desc["EnvironmentLight"]["Texture"]
    if tex:
        from PIL import Image
        img = np.asarray(Image.open(tex).convert("RGBA"), np.float32) / 255.0
        sc.env_texture = S.Texture(np.ascontiguousarray(img))
    return sc.finalize()


# ----------------------------------------------------------------------------------------------
# writer (tests / fixtures): a scenes.Scene -> .gltf + .bin + scene .json that load_scene() maps back onto it
# ----------------------------------------------------------------------------------------------
def export_scene(scene, directory, name="scene", yaw_pitch_roll=None):
    """Every mesh node becomes one glTF file node (identity node transform, vertices z-negated so that the loader's
    Scale(1,1,-1) restores them); every RenderObject keeps its transform as a raw matrix decomposition is avoided by
    storing Translation / Rotation(quaternion) / Scale recovered from the 3x4."""
    from PIL import Image
    os.makedirs(directory, exist_ok=True)
    blob = bytearray(); views = []; accessors = []; images = []; textures = []; materials = []; meshes = []; gl_nodes = []
    img_index = {}

    def add_view(data, target=None):
        while len(blob) % 4:
            blob.append(0)
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": len(data), **({"target": target} if target else {})})
        blob.extend(data)
        return len(views) - 1

    def add_accessor(arr, ctype, atype, target=None, minmax=False):
        v = add_view(np.ascontiguousarray(arr).tobytes(), target)
        a = {"bufferView": v, "componentType": ctype, "count": len(arr), "type": atype}
        if minmax:
            a["min"] = [float(x) for x in arr.min(0)]; a["max"] = [float(x) for x in arr.max(0)]
        accessors.append(a)
        return len(accessors) - 1

    def add_texture(tex):
        if id(tex) not in img_index:
            buf = io.BytesIO(); Image.fromarray(tex.data).save(buf, "PNG")
            images.append({"bufferView": add_view(buf.getvalue()), "mimeType": "image/png"})
            textures.append({"source": len(images) - 1})
            img_index[id(tex)] = len(textures) - 1
        return img_index[id(tex)]

    for node in scene.nodes:
        prims = []
        for mesh in node.meshes:
            vb = mesh.vertices
            pos = vb["Position"].astype(np.float32) * np.array([1, 1, -1], np.float32)
            attrs = {"POSITION": add_accessor(pos, 5126, "VEC3", 34962, True)}
            if mesh.has_normals:
                n = np.maximum(vb["Normal"].astype(np.float32) / 32767.0, -1.0) * np.array([1, 1, -1], np.float32)
                attrs["NORMAL"] = add_accessor(n, 5126, "VEC3", 34962)
            if mesh.has_uv[0]:
                attrs["TEXCOORD_0"] = add_accessor(vb["TexCoord0"].astype(np.float32), 5126, "VEC2", 34962)
            if mesh.has_uv[1]:
                attrs["TEXCOORD_1"] = add_accessor(vb["TexCoord1"].astype(np.float32), 5126, "VEC2", 34962)
            idx = mesh.indices[::-1].astype(np.uint32)                  # the loader writes them back to front
            prim = {"attributes": attrs, "indices": add_accessor(idx, 5125, "SCALAR", 34963), "mode": 4}
            if mesh.material is not None:
                m = mesh.material
                pbr = {"baseColorFactor": [float(x) for x in m["BaseColor"]], "metallicFactor": float(m["Metallic"]), "roughnessFactor": float(m["Roughness"])}
                gm = {"pbrMetallicRoughness": pbr, "emissiveFactor": [float(x) for x in m["EmissiveColor"]],
                      "alphaMode": ["OPAQUE", "MASK", "BLEND"][int(m["AlphaMode"])], "alphaCutoff": float(m["AlphaCutoff"]),
                      "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": float(m["EmissiveStrength"])},
                                     "KHR_materials_ior": {"ior": float(m["IOR"])}}}
                if float(m["Transmission"]) > 0 or (mesh.textures and "Transmission" in mesh.textures):
                    gm["extensions"]["KHR_materials_transmission"] = {"transmissionFactor": float(m["Transmission"])}
                for slot, (tex, uvi) in (mesh.textures or {}).items():
                    info = {"index": add_texture(tex), "texCoord": uvi}
                    if slot == "BaseColor":
                        pbr["baseColorTexture"] = info
                    elif slot == "MetallicRoughness":
                        pbr["metallicRoughnessTexture"] = info
                    elif slot == "EmissiveColor":
                        gm["emissiveTexture"] = info
                    elif slot == "Normal":
                        gm["normalTexture"] = info
                    elif slot == "Transmission":
                        gm["extensions"]["KHR_materials_transmission"]["transmissionTexture"] = info
                materials.append(gm); prim["material"] = len(materials) - 1
            prims.append(prim)
        meshes.append({"primitives": prims})
        gl_nodes.append({"mesh": len(meshes) - 1, "name": f"node{len(gl_nodes)}"})
    models, render_objects = {}, []
    for ni in range(len(scene.nodes)):
        g = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [gl_nodes[ni]],
             "meshes": meshes, "materials": materials, "accessors": accessors, "bufferViews": views, "images": images, "textures": textures,
             "buffers": [{"uri": name + ".bin", "byteLength": len(blob)}],
             "extensionsUsed": ["KHR_materials_emissive_strength", "KHR_materials_ior", "KHR_materials_transmission"]}
        g["nodes"] = [dict(gl_nodes[ni])]
        with open(os.path.join(directory, f"{name}_node{ni}.gltf"), "w") as f:
            json.dump(g, f)
        models[f"node{ni}"] = f"{name}_node{ni}.gltf"
    open(os.path.join(directory, name + ".bin"), "wb").write(bytes(blob))
    for i, ro in enumerate(scene.objects):
        m = np.asarray(ro.transform, np.float64)                          # column-vector [R|t]; R = Rot * diag(scale)
        scale = np.linalg.norm(m[:, :3], axis=0)
        rot_col = m[:, :3] / scale
        if np.linalg.det(rot_col) < 0:
            scale[2] = -scale[2]; rot_col[:, 2] = -rot_col[:, 2]
        r = rot_col.T                                                      # row-vector rotation matrix
        w = math.sqrt(max(0.0, 1 + r[0, 0] + r[1, 1] + r[2, 2])) / 2
        if w > 1e-6:
            q = ((r[1, 2] - r[2, 1]) / (4 * w), (r[2, 0] - r[0, 2]) / (4 * w), (r[0, 1] - r[1, 0]) / (4 * w), w)
        else:
            x = math.sqrt(max(0.0, 1 + r[0, 0] - r[1, 1] - r[2, 2])) / 2
            y = math.sqrt(max(0.0, 1 - r[0, 0] + r[1, 1] - r[2, 2])) / 2
            z = math.sqrt(max(0.0, 1 - r[0, 0] - r[1, 1] + r[2, 2])) / 2
            q = (x, math.copysign(y, r[0, 1] + r[1, 0]), math.copysign(z, r[0, 2] + r[2, 0]), 0.0)
        render_objects.append({"Name": f"object{i}", "IsVisible": bool(ro.visible), "Model": f"node{ro.node}",
                               "Transform": {"Translation": dict(zip("XYZ", map(float, m[:, 3]))),
                                             "Rotation": dict(zip("XYZW", map(float, q))),
                                             "Scale": dict(zip("XYZ", map(float, scale)))}})
    cam = scene.camera
    desc = {"Camera": {"Position": dict(zip("XYZ", map(float, cam["Position"]))), "Rotation": yaw_pitch_roll or {"X": 0, "Y": 0, "Z": 0, "W": 1}},
            "EnvironmentLight": {"Color": dict(zip("RGBA", map(float, scene.scene_data["EnvironmentLightColor"])))},
            "Models": models, "RenderObjects": render_objects}
    out = os.path.join(directory, name + ".json")
    with open(out, "w") as f:
        json.dump(desc, f, indent=1)
    return out
