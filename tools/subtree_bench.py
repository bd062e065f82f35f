"""DEV TOOL: GV_SWEEP_INCREMENTAL vs a full sweep as a function of the dirty fraction (10 M transforms, 4-deep forest).
Dirty sets: scattered single slots over all levels / whole subtrees under moved roots. Times are hipEvent kernel times
of the sweep launch(es) only (the re-mirror of the dirty slots is reported beside them)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from garden_amd import scene
from garden_amd.lib import GpuVisibility, GV_DIRTY_TRANSFORM, GV_SWEEP_INCREMENTAL, GV_SWEEP_VALU, GV_SWEEP_MFMA
n = 10_000_000
sc = scene.hierarchy_scene(n)
tr = sc.transforms
rng = np.random.Generator(np.random.PCG64(5))
with GpuVisibility(profile_events=True) as vis:
    vis.bind_transforms(tr, sc.entity_to_transform); vis.bind_pool(0, sc.meshes); vis.hierarchy_rebuild()
    for mode, name in ((GV_SWEEP_VALU, "full VALU"), (GV_SWEEP_MFMA, "full MFMA")):
        for _ in range(3): vis.sweep(mode)
        vis.wait(); vis.stats_reset()
        for _ in range(10): vis.sweep(mode)
        vis.wait(); print(f"{name} sweep: {vis.stats()['device_ms']['sweep'] / 10 * 1e3:.1f} us", flush=True)
    vis.sweep(GV_SWEEP_INCREMENTAL)
    for label, picks in (("scattered slots", [10, 1000, 10_000, 100_000]), ("moved roots (whole subtrees)", [1, 100, 2000])):
        for k in picks:
            times, remirror = [], []
            for rep in range(5):
                if label.startswith("scattered"):
                    slots = np.sort(rng.choice(n, size=k, replace=False))
                else:
                    slots = np.sort(rng.choice(9600, size=k, replace=False))  # level 0 = the first ~9.6 k slots
                tr["position"][slots, :3] += np.float32(1.0)
                # itemised dirty marks, one per moved transform (as TransformSystem reports setPosition calls)
                t0 = time.perf_counter()
                for s_ in slots.tolist():
                    vis.mark_dirty(GV_DIRTY_TRANSFORM, s_, 1)
                vis.sync(); vis.wait()
                remirror.append(time.perf_counter() - t0)
                vis.stats_reset()
                vis.sweep(GV_SWEEP_INCREMENTAL); vis.wait()
                times.append(vis.stats()["device_ms"]["sweep"] * 1e3)
            print(f"{label}: {k} dirty -> incremental sweep {np.median(times):.1f} us (itemised marks + re-mirror {np.median(remirror) * 1e3:.2f} ms)", flush=True)
