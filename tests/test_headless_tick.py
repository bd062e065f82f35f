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


@pytest.mark.parametrize("gate", ["never", "shadow", "reverse", "empty"])
def test_cpu_system_gate_of_prepare_meshes_against_the_reference_text(tick, gate):
    """mesh.cpp:426 / :482 on the CPU reference-path system (the comparator of the GPU tier): a system with no components, or one
    whose isDrawReady(shadowPass) says no, has its counters at 0 for that pass, contributes no record and — light pass — keeps
    the isVisible bytes the tick started from; hasAnyRefr / hasAnyOIT / hasAnyTD follow :339,488-490 (headless_tick gateHolds)."""
    _, out = tick("--mode", "cpu", "--entities", "8000", "--ticks", "2", "--mixed", "--gate", gate, "--hier")
    assert out["ok"] and out["draw_count"] > 0, out
    # ... and behind prepareSystems' isNonTranslucent filter (mesh.cpp:89-101): Color / Opaque / UI systems only
    _, kept = tick("--mode", "cpu", "--entities", "8000", "--ticks", "2", "--mixed", "--gate", gate, "--hier", "--non-translucent")
    assert kept["ok"] and kept["draw_count"] == out["draw_count"] and kept["sorted_draw_count"] < out["sorted_draw_count"], (out, kept)


def test_cpu_shadow_pass_left_out_by_prepare_shadow_render_keeps_the_numbering(tick):
    """renderShadows (mesh.cpp:809-815): a pass whose prepareShadowRender says no is not prepared, and the passes behind it keep
    their numbers — isDrawReady(shadowPass) is asked with the NUMBER. --gate shadow makes two systems not ready for pass 1 only;
    with pass 0 left out, pass 1 is the first (and only) entry of the list and they must not be drawn in it."""
    _, out = tick("--mode", "cpu", "--entities", "20000", "--ticks", "2", "--mixed", "--gate", "shadow", "--skip-pass", "0")
    assert out["ok"] and out["shadow_draw_counts"] == [0], out
    _, out = tick("--mode", "cpu", "--entities", "20000", "--ticks", "2", "--mixed", "--gate", "shadow", "--skip-pass", "1")
    assert out["ok"] and out["shadow_draw_counts"][0] > 0, out


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
    # record spans: the render passes read the library's page-locked result buffer (UnsortedBuffer::meshes()), nothing is copied
    ["--entities", "100000", "--span-records", "--ticks", "4"],
    ["--entities", "40000", "--mixed", "--csm", "--span-records", "--mutate"],
    ["--entities", "9000", "--animate", "2", "--span-records", "--churn", "4", "--ticks", "5"],
    ["--entities", "30000", "--csm", "--mixed", "--hier", "--mutate"],
    ["--entities", "14000", "--csm", "--animate", "2", "--ticks", "4"],
    # the kept world-matrix cache (GV_SWEEP_INCREMENTAL), checked against the oracle's chain walk every tick: itemised moves
    # (subtree-scoped sweeps), whole-pool marks (full sweeps), re-parenting / destruction / creation in between
    ["--entities", "40000", "--hier", "--animate", "7", "--itemised", "--world", "--ticks", "6"],
    ["--entities", "40000", "--hier", "--animate", "3", "--world", "--ticks", "4"],
    ["--entities", "30000", "--hier", "--animate", "11", "--itemised", "--world", "--mutate", "--ticks", "4"],
    ["--entities", "20000", "--hier", "--mixed", "--animate", "5", "--itemised", "--world", "--churn", "3", "--ticks", "3"],
    ["--entities", "9000", "--animate", "4", "--itemised", "--world", "--ticks", "5"],
    # the per-system gate of prepareMeshes (mesh.cpp:426 / :482: `componentCount == 0 || !isDrawReady(shadowPass)`) and the
    # hasAnyRefr / hasAnyOIT / hasAnyTD outputs (:339,488-490): systems that are never ready, ready for some passes only
    # (light pass + shadow pass 0 but not shadow pass 1, as InstanceRenderSystem::isDrawReady can answer; shadow passes but not
    # the light pass), pools that have lost all their components. Both systems are ALSO checked against the reference text:
    # counters 0, no record, and isVisible of a system that is not drawn keeps the bytes the tick started from
    ["--entities", "30000", "--mixed", "--gate", "never"],
    ["--entities", "30000", "--mixed", "--gate", "shadow", "--hier"],
    ["--entities", "30000", "--mixed", "--gate", "reverse", "--mutate"],
    ["--entities", "30000", "--mixed", "--gate", "empty"],
    ["--entities", "20000", "--mixed", "--gate", "never", "--hier", "--mutate", "--span-records"],
    ["--entities", "20000", "--mixed", "--gate", "shadow", "--animate", "3", "--ticks", "4", "--copy-records"],
    ["--entities", "12000", "--mixed", "--gate", "empty", "--churn", "2", "--soa-records"],
    # MeshRenderSystem::isNonTranslucent (mesh.hpp:275): prepareSystems keeps the Color / Opaque / UI systems only (mesh.cpp:89-101);
    # the others get no buffer and keep the isVisible bytes the tick started from
    ["--entities", "30000", "--mixed", "--non-translucent", "--hier", "--mutate"],
    ["--entities", "20000", "--mixed", "--non-translucent", "--gate", "shadow", "--csm"],
    # a depth image is handed over (setHizDepth, where HizRenderSystem::preHdrRender runs, hiz.cpp:170-174): the light pass of the
    # non-UI systems runs the per-AABB occlusion query behind the frustum test; shadow passes and the UI pass do not
    ["--entities", "60000", "--hiz"],
    ["--entities", "40000", "--hiz", "--mixed", "--hier", "--mutate"],
    ["--entities", "30000", "--hiz", "--mixed", "--csm", "--animate", "4", "--ticks", "4", "--span-records"],
    # a shadow pass left out by prepareShadowRender (mesh.cpp:812-813): the others keep their numbers, isDrawReady is asked with those
    ["--entities", "30000", "--mixed", "--gate", "shadow", "--skip-pass", "0"],
    ["--entities", "30000", "--mixed", "--csm", "--gate", "shadow", "--skip-pass", "1", "--hier", "--mutate"],
    ["--entities", "30000", "--mixed", "--csm", "--gate", "reverse", "--skip-pass", "0", "--churn", "3"],
    # both systems in the SAME frame: entities destroyed since the last frame are still in their pools (wiped at the end of the frame,
    # docs/ECS/Entities.md:52-54) while Manager::tryGet no longer finds them — what an engine presents on the frame after a destroy
    ["--entities", "30000", "--hier", "--mutate", "--churn", "5", "--same-frame"],
    ["--entities", "24000", "--mixed", "--hier", "--csm", "--churn", "4", "--same-frame", "--bounds"],
    ["--entities", "20000", "--mixed", "--gate", "empty", "--churn", "3", "--same-frame", "--animate", "4", "--ticks", "4"],
])
def test_gpu_dropin_matches_cpu_system(tick, args):
    _, out = tick("--mode", "both", *(["--ticks", "3"] if "--ticks" not in args else []), *args)
    assert out["draw_count"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("args", [
    ["--entities", "40000", "--ranks", "4"],
    ["--entities", "30000", "--ranks", "4", "--mixed", "--hier"],
    ["--entities", "30000", "--ranks", "3", "--mixed", "--gate", "shadow"],
    ["--entities", "30000", "--ranks", "2", "--hier", "--mutate"],
    ["--entities", "20000", "--ranks", "4", "--hier", "--animate", "3", "--itemised", "--ticks", "4"],
    ["--entities", "20000", "--ranks", "3", "--mixed", "--hier", "--churn", "3"],
    ["--entities", "60000", "--ranks", "8", "--csm"],
    ["--entities", "20000", "--ranks", "2", "--mixed", "--gate", "never", "--toggle", "--hier"],
    ["--entities", "50000", "--ranks", "4", "--hiz", "--mixed", "--hier"],  # the pyramid is built on every rank
    ["--entities", "30000", "--ranks", "3", "--mixed", "--csm", "--gate", "shadow", "--skip-pass", "0"],
    # mesh systems without change counters (every one of the reference's): what changed is found by comparing with the ranks' copies
    ["--entities", "30000", "--ranks", "3", "--mixed", "--unversioned", "--hier", "--mutate"],
    ["--entities", "30000", "--ranks", "4", "--mixed", "--csm", "--unversioned", "--animate", "3", "--ticks", "5"],
    # roots that cross cells take their trees to another rank (SURVEY.md §8e "re-bin only roots whose position crosses a cell")
    ["--entities", "20000", "--ranks", "4", "--hier", "--animate", "2", "--animate-step", "157.5", "--itemised", "--ticks", "8", "--expect-moved-trees"],
    ["--entities", "20000", "--ranks", "4", "--hier", "--mixed", "--animate", "3", "--animate-step", "211", "--ticks", "6", "--expect-moved-trees"],
    ["--entities", "20000", "--ranks", "3", "--hier", "--mixed", "--animate", "3", "--animate-step", "211", "--ticks", "6", "--no-rebin"],
    ["--entities", "20000", "--ranks", "4", "--mixed", "--csm", "--probe-exchange", "--communicator"],
    # the same mode with the lists travelling through a communicator (gv_exchange_init_all; here the tests' transport) instead of the
    # peer stores one process uses by default: predicted rooms, rows completed by a second exchange
    ["--entities", "40000", "--ranks", "4", "--communicator"],
    ["--entities", "30000", "--ranks", "3", "--mixed", "--hier", "--communicator"],
    ["--entities", "20000", "--ranks", "3", "--mixed", "--hier", "--churn", "3", "--communicator"],
    ["--entities", "60000", "--ranks", "8", "--csm", "--communicator"],
    ["--entities", "50000", "--ranks", "4", "--hiz", "--mixed", "--hier", "--communicator"],
    ["--entities", "30000", "--ranks", "4", "--mixed", "--csm", "--unversioned", "--animate", "3", "--ticks", "5", "--communicator"],
    ["--entities", "20000", "--ranks", "4", "--hier", "--mixed", "--animate", "3", "--animate-step", "211", "--ticks", "6", "--expect-moved-trees", "--communicator"],
    ["--entities", "2000000", "--ranks", "8", "--mixed", "--csm", "--ticks", "2", "--threads", "16", "--communicator"],
    ["--entities", "20000", "--ranks", "2", "--soa-records", "--mixed", "--hier"],  # records through the three-array fetch
    # the frame after a destroy as the engine presents it (--same-frame): components still in their pools, entities gone
    ["--entities", "30000", "--ranks", "4", "--hier", "--mutate", "--churn", "6", "--same-frame"],
    ["--entities", "24000", "--ranks", "3", "--mixed", "--hier", "--csm", "--churn", "5", "--same-frame", "--unversioned"],
    ["--entities", "60000", "--ranks", "8", "--hier", "--churn", "12", "--same-frame", "--ticks", "2"],
    # BASELINE sizes through the drop-in's own multi-GPU mode: 10 M entities dealt to 8 / 4 contexts (hierarchies follow their roots;
    # the occlusion query on every rank), five mesh systems + three cascades at 4 M — every buffer and isVisible byte of the whole
    # pools == the CPU system's
    ["--entities", "10000000", "--ranks", "8", "--hier", "--ticks", "2", "--threads", "16"],
    ["--entities", "10000000", "--ranks", "4", "--hiz", "--ticks", "2", "--threads", "16"],
    ["--entities", "4000000", "--ranks", "8", "--mixed", "--csm", "--ticks", "2", "--threads", "16"],
])
def test_gpu_dropin_multi_gpu_mode_one_process_one_thread(tick, args):
    """The drop-in's own multi-GPU mode — ONE process, ONE thread, N contexts (the reference is one process with one Manager,
    source/editor/entry.cpp:135): the pools are dealt to the ranks (rank_shares.hpp: roots by position, descendants follow), every
    rank culls its share, ALL the frame's lists are gathered on the devices by ONE gv_exchange_views_all / _acquire_all — by default
    with NO communicator: every rank's scatter kernel stores its lists into its row of every rank's rows (gv_exchange_init_peers; here
    all ranks share the box's GPU, on a node the stores cross xGMI); with --communicator through gv_exchange_init_all and the test
    transport — and the engine's buffers are filled from the ranks' results. Checked: every buffer and isVisible of the whole pools == the CPU system's (headless_tick --mode both), every rank
    holds the same gathered rows, and their union is the set of world slots the pass's buffer holds."""
    stub = os.path.join(ROOT, "tests", "cpp", "build", "librccl_stub.so")
    env_before = os.environ.get("GV_RCCL_LIBRARY")
    os.environ["GV_RCCL_LIBRARY"] = stub
    expect_moved = "--expect-moved-trees" in args
    args = [a for a in args if a != "--expect-moved-trees"]
    try:
        _, out = tick("--mode", "both", *(["--ticks", "3"] if "--ticks" not in args else []), *args)
    finally:
        if env_before is None:
            os.environ.pop("GV_RCCL_LIBRARY", None)
        else:
            os.environ["GV_RCCL_LIBRARY"] = env_before
    assert out["ok"] and out["draw_count"] > 0, out
    assert (out["exchange_mode"] == 3) == ("--communicator" not in args), out  # GV_EXCHANGE_PEER unless a communicator was asked for
    # ONE exchange per frame (the reference waits once per prepareMeshes, mesh.cpp:548), seen by the consumer's callback too
    assert out["exchanges"] == out["rank_frames"] == out["exchanges_seen"] or "--probe-exchange" in args, out
    if "--probe-exchange" in args:
        assert out["exchanges"] == out["rank_frames"] and all(ms > 0 for ms in out["exchange_mode_probe_ms"]), out
        assert out["exchange_mode"] == min(range(3), key=lambda m: out["exchange_mode_probe_ms"][m]), out
    # the pools are dealt ONCE: entities and components that come or go (--mutate, --churn, --gate empty), parent links that move
    # (--toggle --hier), edits and moves are all followed slot by slot (rank_shares.hpp followEntities)
    assert out["deals"] == 1, out
    if expect_moved:
        assert out["moved_trees"] > 0 and out["deals"] == 1, out
    if "--no-rebin" in args:
        assert out["moved_trees"] == 0, out


def _exchange_ranks(ranks, entities, env=None, extra=()):
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")], check=True)
    p = subprocess.run([os.path.join(ROOT, "tests", "cpp", "build", "exchange_ranks"), "--ranks", str(ranks), "--entities", str(entities), *extra],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, **(env or {})))
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1 and lines[0]["ok"], (p.stdout, p.stderr[-3000:])
    return lines[0]


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,extra", [(1, ()), (2, ("--random-camera", "7")), (3, ("--batched",)), (4, ("--random-camera", "11", "--batched")),
                                         (8, ("--random-camera", "5")), (8, ("--check-oracle", "--frames", "8"))])
def test_native_exchange_by_peer_stores_one_process(ranks, extra):
    """tests/cpp/exchange_ranks --peers: ONE process, ONE thread, N contexts and NO communicator (gv_exchange_init_peers /
    GV_EXCHANGE_PEER) — every rank's scatter kernel stores its list(s) into its row of every rank's rows (here within the box's GPU; on a
    node over xGMI). Frames behind a camera that turns and cuts (or a new lens and direction every frame), every third frame acquired
    only after the next has been sent; every row of every rank is compared word for word with its owner's own fetched list(s); no
    frame is ever short; a member destroyed without a shutdown dissolves the group."""
    out = _exchange_ranks(ranks, 60000, extra=("--peers", *extra))
    assert out["ranks"] == ranks and out["mismatches"] == 0 and out["transport"].startswith("peer stores") and out["gathered_last_frame"] > 0
    assert out["frames_with_a_second_exchange"] == 0 and abs(out["words_on_links_over_list_words"] - 1.0) < 1e-9  # exactly the lists travel
    if "--check-oracle" in extra:
        assert out["oracle_checked_frames"] >= 2


@pytest.mark.gpu
def test_cfg5_in_its_shape_by_peer_stores_against_the_oracle():
    """BASELINE cfg5 in its shape — 8 ranks of 12.5 M entities — through the one-process peer path: frame 0 and the cut against the CPU
    oracle on every rank (oracle_checked_frames >= 2), every row of every rank word for word its owner's list."""
    out = _exchange_ranks(8, 12_500_000, extra=("--peers", "--check-oracle", "--frames", "6"))
    assert out["mismatches"] == 0 and out["oracle_checked_frames"] >= 2 and out["gathered_last_frame"] > 1_000_000


@pytest.mark.gpu
def test_native_exchange_driver_one_process_per_gpu():
    """tests/cpp/exchange_ranks: the exchange through the C-ABI alone (fork per rank, unique id over pipes, RCCL bound at
    run time) with one rank per GPU of the box — `--ranks auto` = min(GPUs, 8): a 1-rank communicator on the 1-GPU boxes of the
    GPU test tier, real RCCL traffic between ranks wherever the tier runs on a multi-GPU node, without anyone editing the test.
    Frames of gv_exchange_visible + gv_exchange_acquire (rows owned and sized by the library) behind a camera that turns and cuts, then
    gv_exchange_shards with per-rank capacities; every rank's rows are checked against every owner's list — WHOLE in every
    frame the library sized, the one right after the cut included."""
    out = _exchange_ranks("auto", 200000)
    assert out["ranks"] >= 1 and out["frames_with_a_second_exchange"] >= 1 and out["gathered_last_frame"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [("--batched",), ("--batched", "--random-camera", "3", "--mode", "p2p"), ("--random-camera", "9", "--mode", "broadcast"),
                                   ("--peers", "--ranks", "auto", "--batched")])
def test_native_exchange_driver_real_rccl_other_forms(extra):
    """The same driver, one rank per GPU of the box over REAL RCCL (no test transport), in its other forms: ONE exchange for two lists
    per frame (gv_exchange_views: count table + lists, completed by a second exchange after the cut), the direct travel patterns
    behind a new lens every frame — and, last, the one-process peer path with one context per GPU of the box (on a multi-GPU node
    its stores cross xGMI without anyone editing the test)."""
    if "--peers" in extra:
        rest = tuple(a for a in extra if a not in ("--ranks", "auto"))
        out = _exchange_ranks("auto", 200000, extra=rest)
        assert out["mismatches"] == 0 and out["transport"].startswith("peer stores")
        if "unavailable" in out["transport"]:
            pytest.skip(out["transport"])  # (GV_E_STATE: the box's devices cannot reach each other's memory)
    else:
        out = _exchange_ranks("auto", 200000, extra=extra)
        assert out["ranks"] >= 1 and out["mismatches"] == 0 and out["gathered_last_frame"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 4, 8])
def test_native_exchange_logic_with_several_ranks_on_one_gpu(ranks):
    """The same driver with N ranks SHARING the GPU(s) of the box: RCCL refuses two ranks on one device, so the rows travel
    through tests/cpp/rccl_stub (RCCL's entry points over shared memory, named with GV_RCCL_LIBRARY; one rank per process: its
    DEVICE transport — a call only enqueues a kernel on the caller's stream, as RCCL does, so the events and the two streams the
    library orders a frame with are exercised for real). Everything but RCCL's own wire is the product path: per-rank room
    predicted from the previous frame's headers, predictions that the camera cut leaves short completed by a second exchange
    INSIDE the frame, the three travel patterns, frames acquired at once or a frame late, per-rank capacities in the
    caller-sized form. exchange_ranks fails unless every rank holds every owner's whole list in every frame."""
    out = _exchange_ranks(ranks, 60000, env={"GV_RCCL_LIBRARY": os.path.join(ROOT, "tests", "cpp", "build", "librccl_stub.so")})
    assert out["ranks"] == ranks and out["mismatches"] == 0
    # the camera cut swaps which slabs are in view: rows came up short and were completed (a counter now, not a caveat)
    assert out["short_rows_completed"] >= 1 and out["tail_words"] > 0, out
    # the direct patterns move little more than the lists themselves (head-room 1/8 + up to 2 x 1024 words per row, tails included)
    assert out["words_on_links_over_list_words"] < 3.0, out


@pytest.mark.gpu
def test_a_stalled_peer_is_a_status_code_not_a_hang():
    """One of three ranks stops calling half way. The others' next gv_exchange_visible / gv_exchange_acquire waits for headers
    that never come: GV_E_TIMEOUT after the bound (2 s here, gv_exchange_set_timeout), the communicator aborted by
    gv_exchange_shutdown — every frame acquired before that was whole."""
    out = _exchange_ranks(3, 30000, env={"GV_RCCL_LIBRARY": os.path.join(ROOT, "tests", "cpp", "build", "librccl_stub.so")},
                          extra=["--stall-rank", "1", "--frames", "10"])
    assert out["timed_out_ranks"] == 2 and out["mismatches"] == 0, out


@pytest.mark.gpu
def test_shutdown_behind_a_frame_a_stalled_peer_never_joined_is_bounded_too():
    """The other ranks SEND the frame the stalled peer never joins and shut down without acquiring it: gv_exchange_shutdown drains
    the exchange stream with the same bounded wait (GV_E_TIMEOUT, communicator aborted, everything released) instead of
    synchronising a stream that never drains; exchange_ranks fails if it takes longer than 10 s or returns GV_OK."""
    out = _exchange_ranks(3, 30000, env={"GV_RCCL_LIBRARY": os.path.join(ROOT, "tests", "cpp", "build", "librccl_stub.so")},
                          extra=["--stall-rank", "2", "--abandon", "--frames", "10"])
    assert out["timed_out_ranks"] == 2 and out["mismatches"] == 0, out
    # ... and gv_destroy straight away (no shutdown call): bounded the same way
    out = _exchange_ranks(3, 30000, env={"GV_RCCL_LIBRARY": os.path.join(ROOT, "tests", "cpp", "build", "librccl_stub.so")},
                          extra=["--stall-rank", "0", "--abandon-by-destroy", "--frames", "10"])
    assert out["timed_out_ranks"] == 2 and out["mismatches"] == 0, out


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3, 8])
def test_native_batched_exchange_one_frame_carries_two_lists(ranks):
    """gv_exchange_views, one process per rank (the per-rank form of the drop-in's one exchange per frame): every frame culls two
    views and sends both lists in ONE exchange — row = [2 + total, c_0, c_1, list 0, list 1] — through the test transport's device
    form; a camera that turns and cuts (short predictions completed by a second exchange) and a random-lens camera. exchange_ranks
    fails unless every rank holds every owner's whole row — table and both lists — in every frame."""
    env = {"GV_RCCL_LIBRARY": os.path.join(ROOT, "tests", "cpp", "build", "librccl_stub.so")}
    out = _exchange_ranks(ranks, 60000, env=env, extra=["--batched", "--frames", "12"])
    assert out["ranks"] == ranks and out["mismatches"] == 0 and out["short_rows_completed"] >= 1, out
    out = _exchange_ranks(ranks, 60000, env=env, extra=["--batched", "--frames", "16", "--random-camera", "7", "--check-oracle"])
    assert out["ranks"] == ranks and out["mismatches"] == 0 and out["oracle_checked_frames"] >= 2, out
    # a direct travel pattern from the communicator's first frame on (no history: the count tables still arrive with the headers)
    for mode in ("p2p", "broadcast"):
        out = _exchange_ranks(ranks, 30000, env=env, extra=["--batched", "--frames", "6", "--mode", mode])
        assert out["ranks"] == ranks and out["mismatches"] == 0, out


@pytest.mark.gpu
def test_cfg5_in_its_shape_eight_ranks_of_twelve_and_a_half_million():
    """BASELINE.json configs[4] (cfg5) is 10^8 entities over 8 ranks with an all-gatherv of the visible lists. No 8-GPU node is
    available to the test tier, so the SHAPE runs on the one GPU there is: 8 rank processes of 12.5 M entities each (10^8 in
    all), every one a full product context through the C-ABI, the exchange over the test transport (RCCL refuses two ranks on a
    device). 6 frames behind a camera that turns and cuts: every rank must hold every owner's WHOLE list in every frame — about
    2.6 M entries per rank, 2 * 10^7 gathered per frame, tails of millions of words on the cut. Needs ~15 GB of host memory for the
    eight scenes (the GPU boxes have terabytes); ~30 s. --check-oracle: on frame 0 and on the cut every rank also runs the CPU
    oracle over its own 12.5 M entities and compares its own list with it; with every rank holding every owner's whole list, the
    union over the ranks is the oracle's visible set of the 10^8-entity world."""
    out = _exchange_ranks(8, 12_500_000, env={"GV_RCCL_LIBRARY": os.path.join(ROOT, "tests", "cpp", "build", "librccl_stub.so")},
                          extra=["--frames", "6", "--check-oracle"])
    assert out["ranks"] == 8 and out["mismatches"] == 0 and out["timed_out_ranks"] == 0, out
    assert out["oracle_checked_frames"] >= 2, out
    assert out["gathered_last_frame"] > 10_000_000 and out["short_rows_completed"] >= 1 and out["tail_words"] > 1_000_000, out
