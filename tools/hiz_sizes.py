"""Pyramid rebuild time by frame size (wall clock over 300 back-to-back rebuilds, no events): python tools/hiz_sizes.py"""
import sys, time
sys.path.insert(0, '.')
from garden_amd import scene
from garden_amd.lib import GpuVisibility
for (w, h) in [(4096, 4096), (3840, 2160), (2560, 1440), (1920, 1080), (2048, 1024), (1600, 900), (1366, 768), (1280, 720)]:
    depth = scene.synthetic_depth(w, h)
    with GpuVisibility(device=0) as vis:
        vis.hiz_build(depth)
        best = 1e9
        for rep in range(3):
            for _ in range(20): vis.hiz_rebuild()
            vis.wait()
            t0 = time.perf_counter()
            for _ in range(300): vis.hiz_rebuild()
            vis.wait()
            best = min(best, (time.perf_counter() - t0) / 300)
        print(f"{w}x{h}: pyramid rebuild {best * 1e6:.1f} us ({vis.hiz_mip_count()} mips)")
