import sys, os, traceback
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
import schedules
from test_gpu_fuzz import ScheduleReplay
from garden_amd.lib import GpuVisibility
from oracle import oracle_py
oracle_py.load()
for seed in [int(a) for a in sys.argv[1:]]:
    schedule = schedules.generate(seed, ops=80)
    try:
        with GpuVisibility(device=0, keep_slot_order=bool(seed & 1), block_bounds=bool(seed % 5 == 3)) as vis:
            r = ScheduleReplay(vis, oracle_py, schedule, seed)
            # trace ops
            done = []
            orig_run = r.run
            k = 1
            try:
                for k in range(1, len(schedule)):
                    r.run([schedule[0], schedule[k]])
            except AssertionError:
                import numpy as np, torch
                if schedule[k][0] == "mask":
                    p_ = r.last_pool; exp = r.expected(p_, 0); n = r.pools[p_].shape[0]; words = (n + 31) // 32
                    buf = torch.zeros(words + 1, dtype=torch.int32, device="cuda:0"); torch.cuda.synchronize()
                    vis.copy_mask_device(0, buf.data_ptr(), words); vis.wait()
                    host = buf.cpu().numpy().view(np.uint32)
                    bits = np.unpackbits(host[1:].view(np.uint8), bitorder="little")[:n]
                    slots = np.sort(vis.mirror_slots(p_, n)[np.flatnonzero(bits)])
                    print("mask: header", host[0], "expected", exp["count"], "bits set", int(bits.sum()), "missing", len(np.setdiff1d(exp["idx"], slots)), "extra", len(np.setdiff1d(slots, exp["idx"])), "views", [v.get("use_hiz") for v in r.culls[p_]["views"]], "n", n)
                print("seed", seed, "op", k, schedule[k], "last_pool(model)", r.last_pool, "recent ops:", schedule[max(1, k - 12):k + 1])
                raise
        print("seed", seed, "ok")
    except Exception as e:
        tb = traceback.extract_tb(sys.exc_info()[2])
        print("seed", seed, "FAILED at", [(f.lineno, f.name) for f in tb][-3:], repr(e)[:500])
