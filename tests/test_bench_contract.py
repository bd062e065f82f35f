"""bench.py's driver contract on a real GPU: exactly ONE line on stdout, a JSON object with the agreed keys (library
banners and warnings go to stderr), also when the exchange step runs over RCCL."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", [False, True])
def test_bench_prints_one_json_line(exchange):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    if exchange:
        env["GV_BENCH_EXCHANGE"] = "1"  # 1-rank RCCL group: the communicator's version banner must not reach stdout
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg3", "--entities", "200000", "--steps", "5",
                        "--warmup", "2"] + (["--no-cpu-baseline"] if exchange else []),  # the CPU leg (12 s) once is enough
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.splitlines()
    assert len(lines) == 1, p.stdout[:2000]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "parity"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["vs_baseline"] is None and d["value"] > 0
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "measured_stream_peak")) <= set(d["roofline"])
    assert d["roofline"]["measured_stream_peak"] > 100.0  # GB/s, measured in this run
    assert d["ms_per_step_median"] > 0 and d["ms_per_step_min"] <= d["ms_per_step_median"] <= d["ms_per_step_max"]
    if not exchange:
        assert set(("value", "unit", "cores", "kind", "sample", "pyramid_ms", "cull_ms", "cull_culls_per_s")) <= set(d["cpu_baseline"])
    assert d["parity"]["visible_set_bit_identical"] and d["config"]["workload"].startswith("cfg3")
    assert (d["config"]["exchange"] is not None) == exchange
    # every kernel of a frame, measured outside the timed region; the timed region brackets the dominant kernel only
    assert set(("cull", "hiz", "emit")) <= set(d["config"]["frame_kernel_ms"]) and set(d["config"]["kernel_ms"]) == {"cull"}
    assert d["value_with_block_bounds"] == d["config"]["block_bounds_variant"]["value"]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["allgather", "p2p", "broadcast"])
def test_bench_gpus_2_starts_its_own_ranks(mode):
    """`python bench.py --gpus 2` with no torch.distributed environment: bench.py starts the two ranks itself (fresh
    children, before anything touches the GPU), they share this box's one GPU over gloo (a functional check, never a
    measurement), the workload defaults to cfg5's shape, and rank 0 prints the one JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["GV_BENCH_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--entities", "300000", "--steps", "4",
                        "--warmup", "1", "--exchange", mode], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[:2000]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["workload"].startswith("cfg5") and d["config"]["entities_total"] == 600000
    assert d["config"]["exchange"] is not None and d["config"]["exchange_mode"] == mode and d["config"]["exchange_mode_probe_ms"] is None  # (forced: no probe)
    # the timed exchange is the library's own C-ABI step (here over the shared-memory transport: two ranks share the GPU);
    # the torch.distributed form and the other travel patterns are timed beside it, each checked against the exact lists
    assert d["config"]["exchange_path"] == "c-abi" and "GV_RCCL_LIBRARY" in d["config"]["exchange_transport"]
    assert d["config"]["torch_variant"]["checked_against_exact_allgatherv"] and d["config"]["torch_variant"]["ms_per_step"] > 0
    assert set(d["config"]["exchange_mode_variants"]) == {"allgather", "p2p", "broadcast"} - {mode}
    assert all(v["checked_against_exact_allgatherv"] for v in d["config"]["exchange_mode_variants"].values())
    assert d["parity"]["visible_set_bit_identical"] and d["parity"]["baked_model_bit_identical"]
    assert d["config"]["same_frames_without_exchange"]["value"] > 0
    # the compacted index list is what travels by default (BASELINE.json's north_star); the bit shards are a timed variant
    assert d["config"]["exchange_payload"] == "indices" and "uint32 indices" in d["config"]["exchange"]
    assert d["config"]["mask_variant"]["checked_against_exact_allgatherv"] and d["config"]["mask_variant"]["ms_per_step"] > 0
    assert d["parity"]["checked_ranks"] == 2 and len(d["parity"]["visible_by_rank"]) == 2


def _run_bench(argv, env_extra, timeout=1500):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[:2000]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_gpus_8_every_field_of_the_scaling_line():
    """The line the first 8-GPU run will print, with 8 ranks sharing this box's one GPU over gloo (functional, never a
    measurement): the default workload is cfg5's shape, the compacted index lists travel, every rank's tile is checked against
    the oracle, and the line carries n1_same_workload / scaling_efficiency / exchange_ms / shard bytes / the bit-shard variant."""
    d = _run_bench(["--gpus", "8", "--entities", "200000", "--steps", "4", "--warmup", "1"], {"GV_BENCH_BACKEND": "gloo"})
    c = d["config"]
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and c["workload"].startswith("cfg5") and c["entities_total"] == 1_600_000
    assert c["exchange_payload"] == "indices" and c["exchange_path"] == "c-abi"
    # no --exchange: five frames of each travel pattern were timed before the warm-up (slowest rank's clock, all-reduced) and the
    # fastest one carried the timed frames — the first real 8-GPU run chooses its pattern instead of inheriting one chosen blind
    probe = c["exchange_mode_probe_ms"]
    assert set(probe) == {"allgather", "p2p", "broadcast"} and all(ms > 0 for ms in probe.values()), probe
    assert c["exchange_mode"] == min(probe, key=probe.get) and c["exchange_mode"] in c["exchange"], c["exchange_mode"]
    assert set(c["exchange_mode_variants"]) == {"allgather", "p2p", "broadcast"} - {c["exchange_mode"]}
    # every rank owns a share of every region (cells dealt in Morton order): all of them have work, none waits long for another,
    # and the gather moves little more than the lists ([.., 0, 0, 0, 0], 3.6 x)
    vis_by_rank = d["parity"]["visible_by_rank"]
    assert min(vis_by_rank) > 0 and max(vis_by_rank) / (sum(vis_by_rank) / 8.0) <= 1.5 and c["visible_max_over_mean_by_rank"] <= 1.5
    assert c["gathered_over_list_bytes"] <= 1.3, c["gathered_over_list_bytes"]
    assert c["torch_variant"]["checked_against_exact_allgatherv"] and len(c["exchange_mode_variants"]) == 2
    assert d["n1_same_workload"]["value"] > 0 and d["n1_same_workload"]["entities"] == 200000
    assert 0 < d["scaling_efficiency"] < 1.5
    assert abs(d["scaling_efficiency"] - d["value"] / (8 * d["n1_same_workload"]["value"])) < 1e-9
    assert c["exchange_ms"] > 0 and c["exchange_overhead_ms_per_step"] is not None
    # every frame of the run was acquired complete; with a static camera only the frames without a usable history (the first of
    # the communicator, the first after the travel pattern's buffers changed) needed the second exchange — the steady state has none
    assert c["exchange_frames_acquired"] > 50 and c["exchange_frames_completed_by_a_second_exchange"] <= 3, c
    assert len(c["shard_bytes_per_rank"]) == 8 and len(c["list_bytes_per_rank"]) == 8 and c["gathered_bytes_per_rank"] == sum(c["shard_bytes_per_rank"])
    assert all(s >= l for s, l in zip(c["shard_bytes_per_rank"], c["list_bytes_per_rank"]))  # padded shards hold the lists
    assert len(c["same_frames_without_exchange"]["ms_per_step_by_rank"]) == 8
    assert c["mask_variant"]["checked_against_exact_allgatherv"] and len(c["mask_variant"]["shard_bytes_per_rank"]) == 8
    par = d["parity"]
    assert par["checked_ranks"] == 8 and par["checked_entities"] == 1_600_000 and len(par["visible_by_rank"]) == 8
    assert par["visible_set_bit_identical"] and par["is_visible_identical"] and par["baked_model_bit_identical"]
    assert sum(par["visible_by_rank"]) == par["visible"] > 0
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [1, 2])
def test_bench_strong_scaling_cuts_one_world(ranks):
    """--scaling strong: --entities-total is the WORLD; N ranks take total / N each (N = 1: all of it, same workload name)."""
    argv = ["--gpus", str(ranks), "--scaling", "strong", "--entities-total", "600000", "--steps", "4", "--warmup", "1", "--no-cpu-baseline"]
    d = _run_bench(argv, {"GV_BENCH_BACKEND": "gloo"} if ranks > 1 else {})
    assert d["scaling"] == "strong" and d["n_gpus"] == ranks
    assert d["config"]["entities_total"] == 600000 and d["config"]["entities_per_gpu"] == 600000 // ranks
    assert d["config"]["workload"].startswith("cfg5-strong") and d["parity"]["visible_set_bit_identical"]
    assert (d["n1_same_workload"] is not None) == (ranks > 1)


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [1, 2])
def test_bench_mask_payload(ranks):
    """`--payload mask`: shards carry one bit per mirror entry (the entry -> slot tables travel once); the gathered sets are checked against the exact all-gatherv
    inside the bench. One rank over RCCL (GV_BENCH_EXCHANGE=1), two ranks sharing this box's GPU over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    if ranks == 1:
        env.update(GV_BENCH_EXCHANGE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29581")
        cmd = ["--workload", "cfg5", "--entities", "300000"]
    else:
        env["GV_BENCH_BACKEND"] = "gloo"
        cmd = ["--gpus", "2", "--entities", "300000"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *cmd, "--steps", "4", "--warmup", "1", "--payload", "mask",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[:2000]
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["config"]["exchange_payload"] == "mask" and "one bit per mirror entry" in d["config"]["exchange"]
    assert d["config"]["mask_variant"] is None  # the variant is only timed beside the index lists
    assert d["config"]["exchange_path"] == "c-abi" and "gv_exchange_masks" in d["config"]["exchange"]
    assert d["parity"]["visible_set_bit_identical"] and "error" not in d


@pytest.mark.gpu
def test_bench_torch_exchange_path_still_runs():
    """--exchange-path torch: the round-3 form (torch.distributed over the script's buffers) as the timed exchange."""
    d = _run_bench(["--gpus", "2", "--entities", "200000", "--steps", "3", "--warmup", "1", "--exchange-path", "torch", "--no-mask-variant"],
                   {"GV_BENCH_BACKEND": "gloo"})
    assert d["config"]["exchange_path"] == "torch" and d["config"]["torch_variant"] is None and d["config"]["exchange_mode_variants"] is None
    assert d["parity"]["visible_set_bit_identical"]


@pytest.mark.gpu
def test_bench_falls_back_to_the_torch_path_loudly_when_the_library_exchange_cannot_come_up():
    """The library's exchange cannot load its transport (GV_RCCL_LIBRARY names nothing): the line is still measured — through
    torch.distributed — and says so: exchange_path "torch", exchange_path_fallback = what went wrong; parity on every rank."""
    d = _run_bench(["--gpus", "2", "--entities", "100000", "--steps", "3", "--warmup", "1", "--no-mask-variant"],
                   {"GV_BENCH_BACKEND": "gloo", "GV_RCCL_LIBRARY": "/nonexistent/librccl_nowhere.so"})
    c = d["config"]
    assert c["exchange_path"] == "torch" and "gv_exchange_unique_id" in c["exchange_path_fallback"]
    assert c["torch_variant"] is None and c["exchange_mode_variants"] is None and d["parity"]["visible_set_bit_identical"]


@pytest.mark.gpu
def test_bench_hands_over_to_a_child_run_when_a_collective_of_the_library_exchange_never_returns():
    """The library's exchange has never met real RCCL with more than one rank: if one of its collectives hangs on the first real
    node, the run must still print a line. Here the test transport stops completing at its 8th transfer on every rank (among the
    first frames): after GV_BENCH_EXCHANGE_WATCHDOG_S seconds every rank starts the torch.distributed form as a child, which
    prints the line — exchange_path "torch", exchange_path_fallback says what happened — and the ranks leave with its exit code.
    (The library's own bounded waits would end the run with GV_E_TIMEOUT and no line: bench.py sets them behind its watchdog.)"""
    d = _run_bench(["--gpus", "2", "--entities", "100000", "--steps", "3", "--warmup", "1", "--no-mask-variant"],
                   {"GV_BENCH_BACKEND": "gloo", "RCCL_STUB_HANG_AT": "8", "GV_BENCH_EXCHANGE_WATCHDOG_S": "20"})
    c = d["config"]
    assert c["exchange_path"] == "torch" and "no progress" in c["exchange_path_fallback"] and "shared this GPU" in c["exchange_path_fallback"]
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["parity"]["visible_set_bit_identical"]


@pytest.mark.gpu
def test_bench_hands_over_when_the_library_exchange_returns_an_error_in_the_middle_of_the_run():
    """A STATUS CODE from the library's exchange, not a hang: one of the transport's transfers stops completing and the library's
    own bounded wait (3 s here, GV_BENCH_EXCHANGE_TIMEOUT_MS) ends long before the watchdog (60 s): the ranks get GV_E_TIMEOUT
    (or GV_E_RCCL from ncclCommGetAsyncError once a peer has aborted) from gv_exchange_visible / _acquire and hand the line to a
    child run through torch.distributed at once — exchange_path "torch", exchange_path_fallback = the library's error text."""
    d = _run_bench(["--gpus", "2", "--entities", "100000", "--steps", "3", "--warmup", "1", "--no-mask-variant"],
                   {"GV_BENCH_BACKEND": "gloo", "RCCL_STUB_HANG_AT": "8", "GV_BENCH_EXCHANGE_WATCHDOG_S": "60", "GV_BENCH_EXCHANGE_TIMEOUT_MS": "3000"})
    c = d["config"]
    assert c["exchange_path"] == "torch" and "failed after" in c["exchange_path_fallback"] and "libgarden_vis error" in c["exchange_path_fallback"], c["exchange_path_fallback"]
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["parity"]["visible_set_bit_identical"]
