"""A 10 M-entity flat scene in which K scattered entities move EVERY frame (round 3: the block bounds and emit seeds of the pool are
kept current by re-deriving only the 256-entry blocks that hold a moved entity). Per frame: K single-slot dirty marks + cull (+ emit)
with a Hi-Z pyramid; wall clock per frame and the device time per kernel kind. (The round-2 behaviour — a pool that changes every
frame is culled without boxes — was measured beside it in profiles/r03_moving.txt and is in the history.)
   python tools/moving_bench.py [entities = 10_000_000]"""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))


def one(n):
    import numpy as np
    from garden_amd import scene
    from garden_amd.lib import GpuVisibility
    sc = scene.flat_scene(n)
    view = dict(scene.main_camera_view(), use_hiz=1)
    depth = scene.synthetic_depth(4096, 4096)
    rng = np.random.Generator(np.random.PCG64(3))
    mode = "blocks patched"
    with GpuVisibility(profile_events=True) as vis:
        vis.hiz_build(depth)
        vis.bind_transforms(sc.transforms, sc.entity_to_transform)
        vis.bind_pool(0, sc.meshes)
        vis.hierarchy_rebuild()
        for _ in range(3):
            vis.cull(0, [view])
        vis.wait()
        for movers in (0, 10, 1000, 100_000):
            for timed in (False, True):
                if timed:
                    vis.wait()
                    vis.stats_reset()
                    t0 = time.perf_counter()
                frames = 20
                for _ in range(frames):
                    for s in rng.integers(0, n, movers) if movers <= 1000 else ():
                        sc.transforms["position"][s, 0] += np.float32(0.25)
                        vis.mark_dirty(0, int(s), 1)
                    if movers > 1000:  # one contiguous range (the device-side gather)
                        lo = int(rng.integers(0, n - movers))
                        sc.transforms["position"][lo:lo + movers, 0] += np.float32(0.25)
                        vis.mark_dirty(0, lo, movers)
                    vis.cull(0, [view])
                vis.wait()
            dt = (time.perf_counter() - t0) / frames
            st = vis.stats()
            ms = {k: round(v / frames * 1e3, 1) for k, v in st["device_ms"].items() if v > 0}
            print(f"{n} entities, {movers:>6} moved per frame, {mode}: {dt * 1e3:7.3f} ms/frame; device us/frame {ms}; "
                  f"blocks examined {st['bounds_blocks_examined']} of {st['bounds_blocks_total']}", flush=True)


if __name__ == "__main__":
    if os.environ.get("GV_MOVING_BENCH_CHILD"):
        one(int(sys.argv[1]))
        sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    subprocess.run([sys.executable, __file__, str(n)], env=dict(os.environ, GV_MOVING_BENCH_CHILD="1"), check=False)
