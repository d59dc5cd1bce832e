"""Generates tests/golden/*.npz from the CPU oracle (oracle/pt_oracle.c).

The reference has no tests, fixtures or golden vectors for this path and cannot be
built or run here, so these are SELF-CONSISTENCY pins of the oracle (parity unpinned w.r.t. the reference):
they freeze the oracle's arithmetic so that neither the oracle nor the HIP path can drift silently.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
import dxpbrt_amd.layouts as L  # noqa: E402
import dxpbrt_amd.scenes as S  # noqa: E402

CASES = {
    # name: (scene factory, W, H, spp, bounces, ext flags)
    "cornell_ggx_64x36_s2_b4": (lambda: S.cornell_box(aspect=64 / 36, variant="ggx", glass_sphere=True), 64, 36, 2, 4, 0),
    "cornell_lambert_64x64_s1_b2": (lambda: S.cornell_box(aspect=1.0, variant="diffuse", has_normals=False), 64, 64, 1, 2, 1),
    "cornell_ggx_jitter_48x27_s3_b8": (lambda: S.cornell_box(aspect=48 / 27, variant="ggx", jitter=(0.25, -0.375)), 48, 27, 3, 8, 0),
    # every texture slot, normal map, alpha-masked geometry, constant environment (bit-pinned arithmetic: no libm call)
    "cornell_textured_64x36_s2_b6": (lambda: S.cornell_box_textured(aspect=64 / 36, env=None), 64, 36, 2, 6, 0),
}


def render_case(name, oracle):
    make, W, H, spp, bounces, ext = CASES[name]
    scene = make()
    gs = S.graphics_settings(W, H, spp=spp, bounces=bounces, frame_index=7, ext_flags=ext)
    gb, rays, f32 = oracle.render(scene, gs, accel_mode=0, want_f32=True, layouts=L)
    return scene, gs, gb, rays, f32


def main():
    oracle = ge.load_oracle()
    for name in CASES:
        scene, gs, gb, rays, f32 = render_case(name, oracle)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), rays=np.uint64(rays), radiance_f32=f32,
                            radiance_f16=gb["Radiance"], position=gb["Position"], flat_normal=gb["FlatNormal"],
                            normal_roughness=gb["NormalRoughness"], base_color_metalness=gb["BaseColorMetalness"])
        print(name, "rays", rays, "mean radiance", f32[..., :3].mean())
    # RNG / math known-answer vectors
    lib = oracle.lib()
    import ctypes as C
    st = C.c_uint32(lib.or_rng_init(3, 5, 11))
    seq = [st.value] + [lib.or_rng_float(C.byref(st)) for _ in range(16)]
    us = np.linspace(0, 1, 33, dtype=np.float32)
    sc = []
    for u in us:
        s, c = C.c_float(), C.c_float()
        lib.or_sincos_2pi(float(u), C.byref(s), C.byref(c))
        sc.append((s.value, c.value))
    np.savez(os.path.join(HERE, "math_vectors.npz"), rng_seed=np.uint32(seq[0]), rng_floats=np.array(seq[1:], np.float32),
             sincos_u=us, sincos=np.array(sc, np.float32))


if __name__ == "__main__":
    main()
