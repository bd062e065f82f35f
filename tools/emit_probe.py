"""Where the emit launch's fixed time goes: emit_kernel time (hipEvents, every launch) over a 10 M-entry pool for views that
see nothing / little / a fifth of it, as main pass (isVisible bytes written) and as shadow pass (not written).
    python tools/emit_probe.py"""
import sys
sys.path.insert(0, '.')
import numpy as np
from garden_amd import scene
from garden_amd.lib import GpuVisibility

n = 10_000_000
sc = scene.flat_scene(n)
main = scene.main_camera_view()
away = scene.main_camera_view(camera_position=(1e7, 1e7, 1e7))          # the world is behind / beside it: nothing visible
tiny = scene.cascade_view(size=1500.0, depth=3000.0)                    # a small box
with GpuVisibility(device=0, profile_events=True) as vis:
    vis.bind_transforms(sc.transforms, sc.entity_to_transform)
    vis.bind_pool(0, sc.meshes)
    vis.hierarchy_rebuild()
    vis.profile_sampling(1)
    for name, v in [("main camera, 21 %", main), ("main camera looking at nothing", away),
                    ("small ortho box as main pass", dict(tiny, shadow_pass=-1)), ("small ortho box as shadow pass", tiny),
                    ("main camera as shadow pass", dict(main, shadow_pass=0))]:
        for _ in range(5):
            vis.cull(0, [v])
        vis.wait()
        vis.stats_reset()
        for _ in range(40):
            vis.cull(0, [v])
        vis.wait()
        st, ns = vis.stats(), vis.profile_samples()
        count = vis.result_count(0)
        per = {k: st['device_ms'][k] / max(1, ns[k]) * 1e3 for k in ('cull', 'emit')}
        print(f"{name:36s} records {count:8d}  cull {per['cull']:6.1f} us  emit {per['emit']:6.1f} us")
