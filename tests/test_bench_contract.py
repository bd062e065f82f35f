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
    assert d["config"]["exchange"] is not None and d["config"]["exchange_mode"] == mode
    assert d["parity"]["visible_set_bit_identical"] and d["parity"]["baked_model_bit_identical"]
    assert d["config"]["same_frames_without_exchange"]["value"] > 0


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
    assert d["parity"]["visible_set_bit_identical"] and "error" not in d
