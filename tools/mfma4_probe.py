"""Stand-alone world-matrix sweeps at 10 M transforms, 4-deep forest: VALU chain, MFMA chain (1 lane per slot on the memory side,
LDS hand-over) and — with GV_DEBUG_SWEEP_MFMA4=1 in the environment — the MFMA chain with four lanes per slot end to end.
    python tools/mfma4_probe.py            # the two shipped forms
    GV_DEBUG_SWEEP_MFMA4=1 python tools/mfma4_probe.py"""
import os
import sys
sys.path.insert(0, '.')
import numpy as np
from garden_amd import scene
from garden_amd.lib import GpuVisibility, GV_SWEEP_MFMA, GV_SWEEP_VALU

n = 10_000_000
sc = scene.hierarchy_scene(n)
with GpuVisibility(device=0, profile_events=True) as vis:
    vis.bind_transforms(sc.transforms, sc.entity_to_transform)
    vis.bind_pool(0, sc.meshes)
    vis.hierarchy_rebuild()
    out = {}
    for name, mode in (("valu", GV_SWEEP_VALU), ("mfma4" if os.environ.get("GV_DEBUG_SWEEP_MFMA4") else "mfma", GV_SWEEP_MFMA)):
        for _ in range(3):
            vis.sweep(mode)
        vis.wait()
        vis.stats_reset()
        for _ in range(20):
            vis.sweep(mode)
        vis.wait()
        st, ns = vis.stats(), vis.profile_samples()
        out[name] = vis.get_world(0, 2_000_000)
        print(f"{name:6s} sweep {st['device_ms']['sweep'] / max(1, ns['sweep']) * 1e3:7.1f} us")
    a, b = out.values()
    print("bit-identical to the VALU chain:", bool(np.array_equal(a.view(np.uint32), b.view(np.uint32))))
