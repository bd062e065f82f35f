import sys, time
sys.path.insert(0, '.')
from garden_amd import scene
from garden_amd.lib import GpuVisibility
n = 10_000_000
sc = scene.flat_scene(n)
with GpuVisibility(profile_events=True) as vis:
    vis.bind_transforms(sc.transforms, sc.entity_to_transform); vis.bind_pool(0, sc.meshes); vis.hierarchy_rebuild()
    for label, v in (("main camera (21 % visible)", scene.main_camera_view()), ("ortho cascade", scene.cascade_view(size=4000.0))):
        for _ in range(3):
            vis.cull(0, [v]); vis.sort(0)
        vis.wait(); vis.stats_reset()
        for _ in range(50):
            vis.cull(0, [v]); vis.sort(0)
        vis.wait(); st = vis.stats()
        print(f"{label}: {vis.result_count(0)} records; gv_sort {st['device_ms']['sort']/50*1e3:.1f} us (cull {st['device_ms']['cull']/50*1e3:.1f}, emit {st['device_ms']['emit']/50*1e3:.1f})")
