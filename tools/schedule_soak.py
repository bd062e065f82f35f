#!/usr/bin/env python3
"""GPU box: many more random call schedules than the test tier runs (tests/schedules.py; replayer and oracle checks of
tests/test_gpu_fuzz.py::ScheduleReplay) — the hunt for orders of calls over the held-back mechanisms that nobody thought of.
    python tools/schedule_soak.py [--first 200] [--count 2000] [--ops 80]
Prints one line per failing schedule (its text goes to gpurun_out/schedule_<seed>.txt) and a summary."""
import argparse
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import schedules  # noqa: E402
from test_gpu_fuzz import ScheduleReplay  # noqa: E402
from garden_amd.lib import GpuVisibility  # noqa: E402
from oracle import oracle_py  # noqa: E402  (checker)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=200)
    ap.add_argument("--count", type=int, default=2000)
    ap.add_argument("--ops", type=int, default=80)
    args = ap.parse_args()
    oracle_py.load()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    failed, readers, t0 = [], 0, time.time()
    for seed in range(args.first, args.first + args.count):
        schedule = schedules.generate(seed, ops=args.ops)
        try:
            rg16f, linear = seed % 7 == 1, seed % 11 == 2
            with GpuVisibility(device=0, keep_slot_order=bool(seed & 1), block_bounds=bool(seed % 5 == 3) and not linear, hiz_rg16f=rg16f,
                               linear_scan=linear) as vis:
                replay = ScheduleReplay(vis, oracle_py, schedule, seed, rg16f=rg16f)
                replay.run(schedule)
                readers += replay.readers
        except Exception as e:  # noqa: BLE001
            failed.append(seed)
            with open(os.path.join(ROOT, "gpurun_out", f"schedule_{seed}.txt"), "w") as f:
                f.write(schedules.to_text(schedule) + "\n# " + "".join(traceback.format_exception_only(type(e), e))[:2000])
            print(f"schedule {seed}: {type(e).__name__}: {str(e)[:300]}", flush=True)
    print(f"{args.count} schedules of ~{args.ops} operations (seeds {args.first}..{args.first + args.count - 1}), {readers} readers checked against the "
          f"oracle, {len(failed)} failed {failed[:20]}, {time.time() - t0:.0f} s")
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
