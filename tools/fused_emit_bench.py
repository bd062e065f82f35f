"""DEV TOOL: frame time (wall clock over many frames, no profiling events) of cull + emit as two launches vs the fused
look-back kernel, by pool size. Run once per GV_DEBUG_FUSED_EMIT_MAX setting.  python tools/fused_emit_bench.py label"""
import sys, time
sys.path.insert(0, '.')
from garden_amd import scene
from garden_amd.lib import GpuVisibility
label = sys.argv[1] if len(sys.argv) > 1 else ""
view = scene.main_camera_view()
for n, hier in ((10_000, False), (100_000, False), (1_000_000, False), (1_000_000, True), (4_000_000, False), (10_000_000, False), (10_000_000, True)):
    sc = scene.hierarchy_scene(n) if hier else scene.flat_scene(n)
    with GpuVisibility() as vis:
        vis.bind_transforms(sc.transforms, sc.entity_to_transform); vis.bind_pool(0, sc.meshes); vis.hierarchy_rebuild()
        for _ in range(20):
            vis.cull(0, [view])
        vis.wait()
        frames = 300 if n <= 1_000_000 else 60
        t0 = time.perf_counter()
        for _ in range(frames):
            vis.cull(0, [view])
        vis.wait()
        dt = (time.perf_counter() - t0) / frames
        print(f"[{label}] n={n} hier={hier}: {dt * 1e6:.1f} us/frame, {vis.result_count(0)} records", flush=True)
