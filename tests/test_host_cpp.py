"""The C++ host (host/pt_demo.cpp over host/ptamd.hpp): same bytes in, same radiance out as the Python-driven path
and the oracle. The reference's host side is compiled C++, so this is the host a maintainer would actually link."""
import json
import os
import subprocess

import numpy as np
import pytest

import __graft_entry__ as ge

DEMO = os.path.join(ge.PKG_DIR, "pt_demo")


def test_cpp_mirror_header_compiles_standalone(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text('#include "directx-physically-based-raytracer_amd/host/ptamd.hpp"\nint main() { ptamd::Raytracing::GraphicsSettings g; return (int)g.Bounces; }\n')
    subprocess.check_call(["g++", "-std=c++20", "-fsyntax-only", "-I", ge.ROOT, str(src)])


@pytest.mark.gpu
def test_cpp_host_matches_oracle(tmp_path, oracle, pkg):
    assert os.path.exists(DEMO), "pt_demo is not built: run __graft_entry__.build()"
    W, H, spp, bounces = 160, 90, 3, 6
    out = str(tmp_path / "radiance.bin")
    line = subprocess.check_output([DEMO, "--width", str(W), "--height", str(H), "--spp", str(spp), "--bounces", str(bounces),
                                    "--frames", "2", "--out", out], text=True)
    info = json.loads(line.strip().splitlines()[-1])
    got = np.fromfile(out, np.float32).reshape(H, W, 4)
    S, L = pkg.scenes, pkg.layouts
    scene = S.cornell_box(aspect=W / H, variant="ggx")
    gs = S.graphics_settings(W, H, spp=spp, bounces=bounces, frame_index=0)
    ref_gb, ref_rays, ref_f32 = oracle.render(scene, gs, accel_mode=0, want_f32=True, layouts=L)
    assert np.array_equal(got.view(np.uint32), ref_f32.view(np.uint32))          # C++ host builds byte-identical inputs
    assert info["host"] == "c++" and info["rays"] > 2 * W * H


@pytest.mark.gpu
def test_cpp_host_multi_rank_path_assembles_the_same_frame(tmp_path):
    """pt_demo --ranks 1: the C++ host's N > 1 code path (child process per rank, RCCL unique id through a file, pt_comm_init,
    pt_gather_bands into the full frame) on the one GPU a test box has; the assembled frame equals the plain run's."""
    W, H = 160, 90
    common = ["--width", str(W), "--height", str(H), "--spp", "2", "--bounces", "4", "--frames", "2"]
    a, b = str(tmp_path / "a.bin"), str(tmp_path / "b.bin")
    subprocess.check_call([DEMO] + common + ["--out", a])
    line = subprocess.check_output([DEMO] + common + ["--out", b, "--ranks", "1"], text=True)
    info = json.loads(line.strip().splitlines()[-1])
    assert info["world"] == 1 and info["rank"] == 0 and info["local_rows"] == H
    assert np.array_equal(np.fromfile(a, np.uint32), np.fromfile(b, np.uint32))
