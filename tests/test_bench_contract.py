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
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    if not exchange:
        assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"])
    assert d["parity"]["visible_set_bit_identical"] and d["config"]["workload"].startswith("cfg3")
    assert (d["config"]["exchange"] is not None) == exchange
