import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from garden_amd import scene
from garden_amd.lib import GpuVisibility
n = 10_000_000
sc = scene.flat_scene(n)
views = [scene.main_camera_view()] + [scene.cascade_view(index=k, size=3000.0 + 1000 * k) for k in range(3)]
for emit, bounds in (((1, False), (0, False)) if os.environ.get('MULTIVIEW_QUICK') else ((1, False), (0, False), (1, True), (0, True))):
    vs = [dict(v, emit_records=emit) for v in views]
    with GpuVisibility(profile_events=True, block_bounds=bounds) as vis:
        vis.bind_transforms(sc.transforms, sc.entity_to_transform); vis.bind_pool(0, sc.meshes); vis.hierarchy_rebuild()
        for label, runs in (("batched (1 pass, 4 views)", lambda: vis.cull(0, vs)), ("separate (4 passes)", lambda: [vis.cull(0, [v]) for v in vs])):
            for _ in range(5): runs()
            vis.wait(); vis.stats_reset()
            t0 = time.perf_counter()
            for _ in range(20): runs()
            vis.wait(); dt = (time.perf_counter() - t0) / 20
            st = vis.stats()
            print(f"emit={emit} block_bounds={int(bounds)} {label}: {dt*1e3:.3f} ms/frame; cull kernel {st['device_ms']['cull']/20*1e3:.1f} us/frame, scan {st['device_ms']['scan']/20*1e3:.1f}, emit {st['device_ms']['emit']/20*1e3:.1f}")
