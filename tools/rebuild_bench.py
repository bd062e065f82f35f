"""Host phases of a full mirror build (GV_DEBUG_TIMING=1) at 10 M, flat and 4-deep. Dev tool."""
import os, sys, time
os.environ["GV_DEBUG_TIMING"] = "1"
sys.path.insert(0, '.')
from garden_amd import scene
from garden_amd.lib import GpuVisibility
for name, sc in (("flat", scene.flat_scene(10_000_000)), ("hier", scene.hierarchy_scene(10_000_000))):
    with GpuVisibility() as vis:
        for it in range(2):
            t0 = time.perf_counter()
            vis.bind_transforms(sc.transforms, sc.entity_to_transform); vis.bind_pool(0, sc.meshes); vis.hierarchy_rebuild(); vis.wait()
            print(f"{name} full rebuild #{it}: {(time.perf_counter()-t0)*1e3:.1f} ms", file=sys.stderr)
