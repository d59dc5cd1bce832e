"""bench.py --gpus N without torchrun's environment must start its own ranks (the driver runs it unwrapped) -- as a child
process, never by re-exec -- relay rank 0's single JSON line and return the child's status. --launch-check stops every rank
once the process group is up (gloo, so no GPU is needed) and reports the world size it saw."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=600)


def test_gpus_2_self_launches_two_ranks():
    p = _run(["--gpus", "2", "--launch-check"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout                      # the contract: ONE JSON line on stdout
    out = json.loads(lines[0])
    assert out == {"launch_check": True, "world_size": 2, "ranks_seen": 2}


def test_gpus_must_match_world_size_under_torchrun():
    p = _run(["--gpus", "2", "--launch-check"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr
    assert p.stdout.strip() == ""


def test_no_exec_family_call_in_bench():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "os.exec" not in src and "execv" not in src


def test_cpu_baseline_states_the_cores_it_really_uses():
    """`cores` used to be os.cpu_count() (256 on the GPU box, whose cgroup grants the job 16). The baseline now reports the
    affinity mask, the cgroup quota and OpenMP's default next to the thread count it uses, and a one-thread figure beside the all-thread one."""
    sys.path.insert(0, ROOT)
    import numpy as np
    import bench
    import __graft_entry__ as ge
    share = bench.host_cpu_share()
    assert share["os_cpu_count"] >= share["sched_affinity"] >= 1
    assert share["cgroup_cpus"] is None or share["cgroup_cpus"] > 0
    ge.load_package()
    import dxpbrt_amd.layouts as L
    import dxpbrt_amd.scenes as S
    W, H = 96, 54
    scene = S.cornell_box(aspect=W / H, variant="ggx")
    gs = S.graphics_settings(W, H, spp=1, bounces=3)
    out = bench.cpu_baseline(scene, gs, W, H, L, budget_s=0.5)
    usable = share["sched_affinity"] if not share["cgroup_cpus"] else max(1, min(share["sched_affinity"], int(share["cgroup_cpus"] + 0.5)))
    assert out["cores"] == out["threads"] == usable and out["kind"] == "port" and out["value"] > 0
    assert out["one_thread"]["cores"] == 1 and out["one_thread"]["value"] > 0
    assert 0 < out["parallel_efficiency"] <= 1.5 and out["host"]["sched_affinity"] == share["sched_affinity"]
    assert np.isfinite(out["value"])
