"""Writes tests/golden/ingest/fixture.gltf + scene.json, and the same model once more as a binary container: fixture.glb + scene_glb.json
(fastgltf::Parser::loadGltf takes both, Source/GLTFHelpers.ixx:53-57). Uses struct, base64 and json ONLY -- nothing of ingest.py, so that the
fixture cannot share a misreading with the code it checks (the expected arrays in tests/test_ingest.py are written out by hand from
Source/GLTFHelpers.ixx:169-192, Source/Scene.ixx:199-214, Source/JSONConverters.ixx:18-26). The two files are committed; this script
documents how the base64 payload of the .gltf was made."""
import base64, json, os, struct

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ingest")
quad_pos = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0)]
quad_nrm = [(0, 0, 1)] * 4
quad_idx = [0, 1, 2, 0, 2, 3]                 # UNSIGNED_SHORT in the file
tri_pos = [(0, 0, 0), (2, 0, 0), (0, 0, 2)]
tri_idx = [0, 1, 2]                           # UNSIGNED_INT in the file
blob = b"".join(struct.pack("<3f", *p) for p in quad_pos)            # 0   .. 48
blob += b"".join(struct.pack("<3f", *n) for n in quad_nrm)           # 48  .. 96
blob += struct.pack("<6H", *quad_idx)                                # 96  .. 108
blob += b"".join(struct.pack("<3f", *p) for p in tri_pos)            # 108 .. 144
blob += struct.pack("<3I", *tri_idx)                                 # 144 .. 156
gltf = {
    "asset": {"version": "2.0"},
    "extensionsUsed": ["KHR_materials_emissive_strength", "KHR_materials_ior", "KHR_materials_transmission"],
    "scene": 0, "scenes": [{"nodes": [0, 2]}],
    "nodes": [
        {"name": "parent", "translation": [1, 2, 3], "children": [1]},
        {"name": "child", "rotation": [0, 0.7071067811865476, 0, 0.7071067811865476], "scale": [2, 2, 2], "mesh": 0},
        {"name": "loose", "matrix": [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, -1, 0, 0.5, 1], "mesh": 1},
    ],
    "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1}, "indices": 2, "material": 0}]},
               {"primitives": [{"attributes": {"POSITION": 3}, "indices": 4}]}],
    "materials": [{"pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.2, 0.1, 1.0], "metallicFactor": 0.25, "roughnessFactor": 0.5},
                   "emissiveFactor": [1.0, 0.5, 0.25], "alphaMode": "MASK", "alphaCutoff": 0.3,
                   "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 3.0}, "KHR_materials_ior": {"ior": 1.33},
                                  "KHR_materials_transmission": {"transmissionFactor": 0.75}}}],
    "accessors": [
        {"bufferView": 0, "componentType": 5126, "count": 4, "type": "VEC3", "min": [0, 0, 0], "max": [1, 1, 0]},
        {"bufferView": 1, "componentType": 5126, "count": 4, "type": "VEC3"},
        {"bufferView": 2, "componentType": 5123, "count": 6, "type": "SCALAR"},
        {"bufferView": 3, "componentType": 5126, "count": 3, "type": "VEC3", "min": [0, 0, 0], "max": [2, 0, 2]},
        {"bufferView": 4, "componentType": 5125, "count": 3, "type": "SCALAR"},
    ],
    "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 48}, {"buffer": 0, "byteOffset": 48, "byteLength": 48},
                    {"buffer": 0, "byteOffset": 96, "byteLength": 12}, {"buffer": 0, "byteOffset": 108, "byteLength": 36},
                    {"buffer": 0, "byteOffset": 144, "byteLength": 12}],
    "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}],
}
scene = {
    "Camera": {"Position": {"X": 0, "Y": 1, "Z": -5}, "Rotation": {"Yaw": 90, "Pitch": 0, "Roll": 0}},
    "EnvironmentLight": {"Color": {"R": 0.1, "G": 0.2, "B": 0.3, "A": 1.0}},
    "Models": {"m": "fixture.gltf"},
    "RenderObjects": [
        {"Name": "a", "Model": "m", "Transform": {"Translation": {"X": 10, "Y": 0, "Z": 0}, "Rotation": {"Yaw": 90}, "Scale": {"X": 1, "Y": 1, "Z": 1}}},
        {"Name": "b", "Model": "m", "IsVisible": False,
         "Transform": {"Translation": {"X": 0, "Y": 5, "Z": 0}, "Rotation": {"X": 0, "Y": 0, "Z": 0.7071067811865476, "W": 0.7071067811865476}}},
    ],
}
os.makedirs(HERE, exist_ok=True)
json.dump(gltf, open(os.path.join(HERE, "fixture.gltf"), "w"), indent=1)
json.dump(scene, open(os.path.join(HERE, "scene.json"), "w"), indent=1)

# ---- the same asset as .glb (glTF 2.0 binary container): 12-byte header "glTF" | version 2 | total length, then chunks of
# length | type | payload, each padded to 4 bytes: JSON (0x4E4F534A, padded with spaces) and BIN (0x004E4942, padded with zeros).
# The one buffer has no uri: it is the BIN chunk.
glb_json = dict(gltf)
glb_json["buffers"] = [{"byteLength": len(blob)}]
jbytes = json.dumps(glb_json, separators=(",", ":")).encode("utf-8")
jbytes += b" " * (-len(jbytes) % 4)
bbytes = blob + b"\0" * (-len(blob) % 4)
total = 12 + 8 + len(jbytes) + 8 + len(bbytes)
with open(os.path.join(HERE, "fixture.glb"), "wb") as f:
    f.write(struct.pack("<4sII", b"glTF", 2, total))
    f.write(struct.pack("<II", len(jbytes), 0x4E4F534A) + jbytes)
    f.write(struct.pack("<II", len(bbytes), 0x004E4942) + bbytes)
scene_glb = dict(scene)
scene_glb["Models"] = {"m": "fixture.glb"}
json.dump(scene_glb, open(os.path.join(HERE, "scene_glb.json"), "w"), indent=1)
