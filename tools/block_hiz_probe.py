"""How many 256-entry workgroups the bounded cull examines: frustum test alone vs frustum + block-level Hi-Z, per depth image.
    python tools/block_hiz_probe.py [entities]"""
import sys
sys.path.insert(0, '.')
import numpy as np
from garden_amd import scene
from garden_amd.lib import GpuVisibility

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
sc = scene.flat_scene(n)
depth = scene.synthetic_depth(4096, 4096)
with GpuVisibility(device=0, block_bounds=True, profile_events=True) as vis:
    vis.bind_transforms(sc.transforms, sc.entity_to_transform)
    vis.bind_pool(0, sc.meshes)
    vis.hierarchy_rebuild()
    for name, d in (("synthetic walls", depth), ("all wall 0.75", np.full_like(depth, 0.75)), ("no wall", np.zeros_like(depth))):
        vis.hiz_build(d)
        for hiz in (0, 1):
            v = scene.main_camera_view(use_hiz=hiz)
            for _ in range(3):
                vis.cull(0, [v])
            vis.wait()
            vis.stats_reset()
            for _ in range(10):
                vis.cull(0, [v])
            vis.wait()
            st, ns = vis.stats(), vis.profile_samples()
            print(f"{name:16s} hiz={hiz} examined {st['bounds_blocks_examined']:6d} of {st['bounds_blocks_total']:6d}  visible {vis.result_count(0):8d}  "
                  f"cull {st['device_ms']['cull'] / max(1, ns['cull']) * 1e3:6.1f} us")
