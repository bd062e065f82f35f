"""The C++ boundary: an ecsm-style headless frame loop (Manager::update -> Update -> Render -> PreDeferredRender)
with the CPU reference-path system (oracle) and, on the GPU tier, the drop-in GpuVisibilitySystem over the C-ABI,
compared bit for bit on what they leave for the render phase (isVisible, combinedMeshes, counters)."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "build", "headless_tick")


@pytest.fixture(scope="module")
def tick():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")], check=True)

    def run(*args, check=True):
        p = subprocess.run([BIN, *args], capture_output=True, text=True, timeout=600)
        out = json.loads(p.stdout.strip().splitlines()[-1])
        if check:
            assert p.returncode == 0 and out["ok"], out
        return p.returncode, out
    return run


def test_cfg1_cpu_reference_path_headless_tick(tick):
    """BASELINE.json configs[0]: 10k entities, flat hierarchy, frustum-only cull on the CPU path, no Vulkan."""
    _, one = tick("--mode", "cpu", "--entities", "10000", "--ticks", "20")
    assert 0 < one["draw_count"] < 10000 and one["is_visible_set"] == one["draw_count"]
    _, many = tick("--mode", "cpu", "--entities", "10000", "--ticks", "20", "--threads", "4")
    assert many["draw_count"] == one["draw_count"]  # range split does not change the set (thread-pool.cpp:173-200)
    # the AVX2+FMA SoA path (the reference builds for -march=haswell): same visible set, 1 thread and several
    _, avx = tick("--mode", "cpu", "--entities", "10000", "--ticks", "20", "--avx2")
    _, avx4 = tick("--mode", "cpu", "--entities", "10000", "--ticks", "20", "--avx2", "--threads", "4")
    assert avx["draw_count"] == one["draw_count"] == avx4["draw_count"] and avx["is_visible_set"] == one["is_visible_set"]


def test_cpu_hierarchy_and_mutations(tick):
    _, out = tick("--mode", "cpu", "--entities", "5000", "--ticks", "3", "--hier", "--mutate")
    assert out["draw_count"] > 0


def test_cpu_mixed_mesh_systems_sorted_and_unsorted_buffers(tick):
    """prepareMeshes' classification (mesh.cpp:341-546): Opaque + OIT unsorted buffers, two Translucent systems sharing
    transSortedMeshes, one UI system with its own frustum and 2D key; plus two shadow passes."""
    _, out = tick("--mode", "cpu", "--entities", "8000", "--ticks", "2", "--mixed")
    assert out["draw_count"] > 0 and out["sorted_draw_count"] > 0
    _, out = tick("--mode", "cpu", "--entities", "8000", "--ticks", "2", "--mixed", "--hier", "--mutate", "--threads", "3")
    assert out["draw_count"] > 0 and out["sorted_draw_count"] > 0


def test_cpu_entity_churn(tick):
    """Entities destroyed and created between frames (incl. re-parented orphans, re-used slots, pool growth)."""
    _, out = tick("--mode", "cpu", "--entities", "6000", "--ticks", "2", "--hier", "--mixed", "--churn", "4")
    assert out["draw_count"] > 0


def test_csm_cascades_on_the_cpu_reference_path(tick):
    """csm_lite.hpp restates CsmRenderSystem's calcLightViewProj / slice selection (csm.cpp:262-325): the binary checks
    that every slice corner lies in its light box; here the three cascades see nested, growing parts of the scene."""
    _, out = tick("--mode", "cpu", "--entities", "200000", "--ticks", "2", "--csm")
    a, b, c = out["shadow_draw_counts"]
    assert 0 < a < b < c < 200000 and out["draw_count"] > 0
    _, out = tick("--mode", "cpu", "--entities", "20000", "--ticks", "2", "--csm", "--mixed", "--hier")
    assert all(n > 0 for n in out["shadow_draw_counts"]) and out["sorted_draw_count"] > 0


def test_gpu_system_fails_loudly_without_device(tick):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    rc, out = tick("--mode", "gpu", "--entities", "100", check=False)
    assert rc == 1 and "no CPU fallback" in out["why"]


@pytest.mark.gpu
@pytest.mark.parametrize("args", [
    ["--entities", "10000"],
    ["--entities", "10000", "--threads", "8"],
    ["--entities", "30000", "--hier", "--mutate", "--avx2"],
    ["--entities", "50000", "--hier"],
    ["--entities", "50000", "--hier", "--mutate"],
    ["--entities", "1000", "--mutate"],
    ["--entities", "50000", "--hier", "--toggle"],
    ["--entities", "20000", "--toggle"],
    ["--entities", "40000", "--mixed"],
    ["--entities", "40000", "--mixed", "--hier", "--mutate"],
    ["--entities", "60000", "--hier", "--toggle", "--bounds", "--ticks", "4"],
    ["--entities", "30000", "--churn", "6"],
    ["--entities", "30000", "--hier", "--mutate", "--churn", "8"],
    ["--entities", "20000", "--mixed", "--hier", "--churn", "5", "--bounds"],
    ["--entities", "40000", "--mixed", "--bounds"],
    # a moving scene, compared tick by tick: dense and sparse dirty ranges, small pools (published results) and large
    ["--entities", "20000", "--animate", "3", "--hier", "--ticks", "6"],
    ["--entities", "9000", "--animate", "1", "--mixed", "--ticks", "5"],
    ["--entities", "150000", "--animate", "64", "--ticks", "4"],
    ["--entities", "12000", "--animate", "5", "--hier", "--mixed", "--bounds", "--ticks", "6"],
    # shadow passes from calcLightViewProj (csm_lite.hpp): three cascades batched with the main camera
    ["--entities", "150000", "--csm"],
    ["--entities", "30000", "--csm", "--mixed", "--hier", "--mutate"],
    ["--entities", "14000", "--csm", "--animate", "2", "--ticks", "4"],
    # the kept world-matrix cache (GV_SWEEP_INCREMENTAL), checked against the oracle's chain walk every tick: itemised moves
    # (subtree-scoped sweeps), whole-pool marks (full sweeps), re-parenting / destruction / creation in between
    ["--entities", "40000", "--hier", "--animate", "7", "--itemised", "--world", "--ticks", "6"],
    ["--entities", "40000", "--hier", "--animate", "3", "--world", "--ticks", "4"],
    ["--entities", "30000", "--hier", "--animate", "11", "--itemised", "--world", "--mutate", "--ticks", "4"],
    ["--entities", "20000", "--hier", "--mixed", "--animate", "5", "--itemised", "--world", "--churn", "3", "--ticks", "3"],
    ["--entities", "9000", "--animate", "4", "--itemised", "--world", "--ticks", "5"],
])
def test_gpu_dropin_matches_cpu_system(tick, args):
    _, out = tick("--mode", "both", *(["--ticks", "3"] if "--ticks" not in args else []), *args)
    assert out["draw_count"] > 0


@pytest.mark.gpu
def test_native_exchange_driver_one_process_per_gpu():
    """tests/cpp/exchange_ranks: the exchange through the C-ABI alone (fork per rank, unique id over pipes, RCCL bound at
    run time). The GPU test tier runs on 1-GPU boxes, so this is the 1-rank communicator; on a multi-GPU node run
    `tests/cpp/build/exchange_ranks --ranks N` by hand (one rank per GPU: RCCL refuses two ranks on one device)."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")], check=True)
    ranks = 1
    p = subprocess.run([os.path.join(ROOT, "tests", "cpp", "build", "exchange_ranks"), "--ranks", str(ranks), "--entities", "200000"],
                       capture_output=True, text=True, timeout=600)
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == ranks and all(l["ok"] and l["visible"] > 0 for l in lines), (p.stdout, p.stderr[-2000:])
    assert all(l["gathered"] == sum(x["visible"] for x in lines) for l in lines)
