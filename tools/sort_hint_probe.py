import sys, time
sys.path.insert(0, '.')
import numpy as np
from garden_amd import scene
from garden_amd.lib import GpuVisibility
for n in (40_000, 150_000):
    sc = scene.flat_scene(n, seed=5 + n)
    small = scene.cascade_view(size=700.0 if n == 40_000 else 1100.0, depth=60000.0)
    wide = scene.cascade_view(size=30000.0, depth=60000.0)
    with GpuVisibility(device=0) as vis:
        vis.bind_transforms(sc.transforms, sc.entity_to_transform); vis.bind_pool(0, sc.meshes); vis.hierarchy_rebuild()
        def frame(v):
            vis.cull(0, [v]); vis.wait()
            t0 = time.perf_counter(); vis.sort(0); r = vis.fetch(0, write_back=False, occupancy=n, order="raw"); dt = time.perf_counter() - t0
            return r["draw_count"], dt * 1e6
        for v, name in [(small, "small"), (small, "small"), (wide, "wide after small (rank sort alone, keys from memory)"), (wide, "wide again (radix)"), (wide, "wide again (radix)"), (small, "small after wide (both enqueued)"), (small, "small again (rank sort alone)")]:
            c, us = frame(v)
            print(f"{n} slots: {name:55s} {c:7d} records  sort + fetch {us:7.1f} us")
