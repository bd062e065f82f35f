"""The spatial re-order after entity churn: a pool of N entities grows by N/8 + 1 created entities (re-bound with the
larger occupancy, no rebuild request), which puts more than 1/8 of it into the mirror's unsorted tail — the next sync re-orders.
Times that sync (growth + re-order) on the device (the full host rebuild it replaced: profiles/r03_reorder.txt), and the
frame after it; checks the visible set against the oracle's both times. Dev tool / evidence (profiles/r03_reorder.txt).
   python tools/reorder_bench.py [entities after growth = 10_000_000] [flat|hier]"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))


def one(n_total, kind):
    import numpy as np
    from garden_amd import scene
    from garden_amd.lib import GpuVisibility
    from oracle import oracle_py
    full = scene.hierarchy_scene(n_total) if kind == "hier" else scene.flat_scene(n_total)
    n0 = n_total - (n_total // 8 + 1)  # the tail is then just over 1/8 of the grown pool
    view = scene.main_camera_view()

    def cut(n):
        e2t = full.entity_to_transform.copy()
        e2t[e2t >= n] = 0xFFFFFFFF
        tr = full.transforms[:n].copy()
        return scene.Scene(full.meshes[:n].copy(), tr, e2t)

    with GpuVisibility() as vis:
        a = cut(n0)
        vis.bind_transforms(a.transforms, a.entity_to_transform)
        vis.bind_pool(0, a.meshes)
        vis.hierarchy_rebuild()
        vis.cull(0, [view])
        vis.wait()
        b = cut(n_total)
        t0 = time.perf_counter()
        vis.bind_transforms(b.transforms, b.entity_to_transform)
        vis.bind_pool(0, b.meshes)
        vis.sync()
        vis.wait()
        t_sync = time.perf_counter() - t0
        reorders = vis.stats()["mirror_reorders"]
        for _ in range(3):
            vis.cull(0, [view])
        vis.wait()
        t1 = time.perf_counter()
        for _ in range(20):
            vis.cull(0, [view])
        vis.wait()
        t_frame = (time.perf_counter() - t1) / 20
        got = vis.fetch(0, write_back=False, occupancy=n_total)
        exp = oracle_py.prepare_meshes(b.meshes.copy(), b.transforms, b.entity_to_transform, view, threads=os.cpu_count())
        ok = np.array_equal(got["visible_idx"], np.sort(exp["visible_idx"]))
        mode = "device re-order"
        print(f"{kind} {n0} -> {n_total} entities, {mode}: sync (append {n_total - n0} + re-order) {t_sync * 1e3:8.1f} ms, "
              f"device re-orders {reorders}, frame afterwards {t_frame * 1e3:.3f} ms, visible set == oracle: {ok}", flush=True)


if __name__ == "__main__":
    if os.environ.get("GV_REORDER_BENCH_CHILD"):
        one(int(sys.argv[1]), sys.argv[2])
        sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    kinds = [sys.argv[2]] if len(sys.argv) > 2 else ["flat", "hier"]
    for kind in kinds:
        subprocess.run([sys.executable, __file__, str(n), kind], env=dict(os.environ, GV_REORDER_BENCH_CHILD="1", GV_DEBUG_TIMING="1"), check=False)
