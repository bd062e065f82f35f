import sys, os, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import schedules
from test_gpu_fuzz import ScheduleReplay
from garden_amd.lib import GpuVisibility
from oracle import oracle_py
oracle_py.load()
for seed in [int(a) for a in sys.argv[1:]]:
    schedule = schedules.generate(seed, ops=80)
    with GpuVisibility(device=0, keep_slot_order=bool(seed & 1), block_bounds=bool(seed % 5 == 3)) as vis:
        r = ScheduleReplay(vis, oracle_py, schedule, seed)
        k = 1
        try:
            for k in range(1, len(schedule)):
                r.run([schedule[0], schedule[k]])
            print("seed", seed, "ok")
        except Exception as e:
            print("seed", seed, "world", schedule[0], "keep_slot_order", bool(seed & 1), "bounds", seed % 5 == 3)
            print("  failed op", k, schedule[k], repr(e)[:300])
            p = schedule[k][1] if len(schedule[k]) > 1 and isinstance(schedule[k][1], int) else r.last_pool
            hist = [(j, op) for j, op in enumerate(schedule[:k + 1]) if j and (op[0] in ("begin", "end", "move_xf", "rebuild", "reparent", "dirty_xf", "sweep", "hiz", "hiz_rebuild", "sync", "wait") or (len(op) > 1 and op[1] == p))]
            print("  history for pool", p, hist[-25:])
