"""Per-frame cost of re-mirroring transforms (host AoS -> SoA gather + PCIe + device scatter) for dirty ranges of
different sizes on a 10 M pool; the cull itself is ~0.1 ms. Dev tool."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from garden_amd import scene
from garden_amd.lib import GpuVisibility
n = 10_000_000
sc = scene.flat_scene(n)
view = scene.main_camera_view()
with GpuVisibility() as vis:
    vis.bind_transforms(sc.transforms, sc.entity_to_transform); vis.bind_pool(0, sc.meshes); vis.hierarchy_rebuild()
    vis.cull(0, [view]); vis.wait()
    for count in (10_000, 100_000, 1_000_000, 5_000_000, 10_000_000):
        ts = []
        for it in range(5):
            t0 = time.perf_counter()
            vis.mark_dirty(0, (it * 1234567) % (n - count + 1), count)
            vis.cull(0, [view]); vis.wait()
            ts.append(time.perf_counter() - t0)
        t = sorted(ts)[2]
        print(f"dirty {count:>9} transforms: {t*1e3:8.2f} ms/frame  {count/t/1e6:8.1f} M transforms/s re-mirrored  {count*45/t/1e9:6.2f} GB/s of mirror bytes")
