"""DEV TOOL: per-kernel times (hipEvents) of cull / emit / sort for the cfg2@10M and cfg4 shapes; run once per
environment setting (GV_DEBUG_EMIT_DIRECT_STORES, GV_DEBUG_EMIT_CHAIN ...) for same-box A/Bs.
  python tools/emit_bench.py [label]"""
import sys
sys.path.insert(0, '.')
from garden_amd import scene
from garden_amd.lib import GpuVisibility, GV_SWEEP_WITH_CULL, GV_SWEEP_WITH_CULL_VALU, GV_SWEEP_MFMA
label = sys.argv[1] if len(sys.argv) > 1 else ""
n = 10_000_000
view = scene.main_camera_view()
for name, sc, sweeps in (("flat 10M (cfg2 shape)", scene.flat_scene(n), [None]),
                         ("hier 10M (cfg4)", scene.hierarchy_scene(n), [None, GV_SWEEP_WITH_CULL_VALU, GV_SWEEP_WITH_CULL, GV_SWEEP_MFMA])):
    with GpuVisibility(profile_events=True) as vis:
        vis.bind_transforms(sc.transforms, sc.entity_to_transform); vis.bind_pool(0, sc.meshes); vis.hierarchy_rebuild()
        for sweep in sweeps:
            def frame():
                if sweep is not None:
                    vis.sweep(sweep)
                vis.cull(0, [view])
            for _ in range(5):
                frame()
            vis.wait(); vis.stats_reset()
            for _ in range(20):
                frame()
            vis.wait(); st = vis.stats()
            ms = {k: v / 20 * 1e3 for k, v in st["device_ms"].items() if v > 0}
            print(f"[{label}] {name} sweep={sweep}: {vis.result_count(0)} records; " + ", ".join(f"{k} {v:.1f} us" for k, v in ms.items()), flush=True)
