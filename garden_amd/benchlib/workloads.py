"""The bench's workloads (BASELINE.json configs[1..4]), their synthetic scenes and their algorithmic bytes."""
import hashlib
import json
import os

import numpy as np

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
# sources whose hash identifies the dominant kernel's code: profiles/traffic.json records it at PMC-collection time and
# roofline.traffic is only reported while it still matches (a stale counter figure is worse than none)
KERNEL_SOURCES = ["garden_amd/csrc/gv_cull.hip", "garden_amd/csrc/gv_device.hpp", "garden_amd/csrc/gv_device_math.hpp",
                  "garden_amd/csrc/gv_sweep.hip", "garden_amd/csrc/gv_kernels.hpp"]
WORKLOADS = {
    "cfg2": dict(entities=1_000_000, hier=False, hiz=False, sweep=False,
                 name="cfg2: 1M static entities, flat hierarchy, frustum-only AABB cull, fp32"),
    "cfg3": dict(entities=10_000_000, hier=False, hiz=True, sweep=False,
                 name="cfg3: 10M entities, frustum + Hi-Z occlusion vs synthetic 4096^2 depth pyramid (rebuilt per frame)"),
    "cfg5": dict(entities=12_500_000, hier=False, hiz=False, sweep=False,
                 name="cfg5: 100M entities over 8 spatial tiles (12.5M per GPU), frustum-only cull per tile + all-gather of the visible lists"),
    "cfg4": dict(entities=10_000_000, hier=True, hiz=False, sweep=True,
                 name="cfg4: 10M entities, 4-deep transform hierarchy recomputed each frame (MFMA 4x4 chain sweep) + cull"),
}
HIZ_SIZE = 4096


def kernel_source_sha(root):
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(root, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def make_tile_scene(wl, n_local, rank, world):
    """What rank `rank` owns of the world cube (side 100 * N_total^(1/3); camera at the world centre): the cube is cut into
    cell_grid(world) cells (8 x 8 x 8 for 8 GPUs), the cells are dealt to the ranks round-robin in Morton order
    (garden_amd/multi.py::cell_owners — the rule of gv_scene_extract_rank) and the rank's n_local roots are spread evenly over
    ITS cells, concatenated into one pool: every rank holds a share of every region, so every rank has its share of whatever
    the camera looks at (round 3 gave each rank one octant: half the ranks had nothing in view)."""
    from garden_amd import scene
    sc = scene.hierarchy_scene(n_local, seed=scene.SEED + rank) if wl["hier"] else scene.flat_scene(n_local, seed=scene.SEED + rank)
    if world > 1:
        from garden_amd.multi import cell_grid, cell_owners
        side = 100.0 * (n_local * world) ** (1.0 / 3.0)
        local_side = 100.0 * n_local ** (1.0 / 3.0)
        g = cell_grid(world)
        mine = np.nonzero(cell_owners(g, world) == rank)[0]  # linear cell ids x + y * gx + z * gx * gy
        k = mine.shape[0]
        roots = sc.transforms["parent"] == 0
        pos = sc.transforms["position"]
        # roots were drawn uniform in [-local_side/2, local_side/2)^3: x picks the cell (k equal slabs of the local cube) and the
        # place inside it, y and z the place inside the cell
        u = (pos[roots, :3].astype(np.float64) / local_side + 0.5).clip(0.0, np.nextafter(1.0, 0.0))
        j = np.minimum((u[:, 0] * k).astype(np.int64), k - 1)
        u[:, 0] = u[:, 0] * k - j
        cell = mine[j]
        cxyz = np.stack([cell % g[0], (cell // g[0]) % g[1], cell // (g[0] * g[1])], axis=1).astype(np.float64)
        ext = side / np.array(g, dtype=np.float64)
        pos[roots, :3] = (-0.5 * side + (cxyz + u) * ext).astype(np.float32)
    return sc


def algorithmic_bytes(wl, n, frustum_survivors, visible, depth, fused=False, examined=1.0):
    """Minimal SoA stream bytes per launch for the cull kernel, and
    for the whole step (for information). `examined`: fraction of the 256-entry workgroups whose streams are read
    (1 without block bounds; with them the rest only write their outputs and read a 32-byte box)."""
    # TRS 40 + AABB 24 + flags 1 read; ballot word 1/8 written (the isVisible bytes are expanded from those words by the emit
    # kernel: 1 B per entity there, not here)
    cull = n * examined * 65.0 + n * 0.125
    if examined < 1.0:
        cull += (n / 256.0) * 32.0
    if wl["hier"]:
        cull += n * examined * 4.0  # parent index
    if wl["hiz"]:
        cull += frustum_survivors * 32.0  # 4 texels x (min,max) fp32 per frustum-surviving entity
    emit = visible * (40.0 + 4.0 + 4.0 + 48.0 + 4.0) + n * 0.125 + n * 1.0
    # level 1 is not stored: depth read + levels 2..12 written
    hiz = (HIZ_SIZE * HIZ_SIZE * 4 + sum(max(HIZ_SIZE >> k, 1) ** 2 * 8 for k in range(2, 13))) if wl["hiz"] else 0.0
    sweep = n * (40.0 + 4.0 + 48.0) if wl["sweep"] else 0.0
    if fused and wl["sweep"]:  # one pass: the TRS streams are read once, the world matrices (48 B) written beside the cull outputs
        cull += n * 48.0
        sweep = 0.0
    return dict(cull=cull, emit=emit, hiz=float(hiz), sweep=sweep)


def counter_traffic(root, args, n):
    """roofline.traffic: PMC bytes of the dominant kernel from profiles/traffic.json — only while the kernel sources still hash to
    what they were when the counters were collected. Returns (bytes per launch or None, where it came from or None)."""
    tpath = os.path.join(root, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None, None
    try:
        tj = json.load(open(tpath))
        key = {"cfg5": "cfg2_at_10M", "cfg2": "cfg2_at_10M"}.get(args.workload, args.workload)
        if args.workload == "cfg3" and args.depth == "noise":
            key = "cfg3_hard_depth"
        entry = tj.get(key, {})
        now = kernel_source_sha(root)
        source = {"file": "profiles/traffic.json", "entry": key, "collected": tj.get("_collected"),
                  "kernel_source_sha_at_collection": tj.get("_kernel_source_sha"), "kernel_source_sha_now": now, "per_entity_scaled": False}
        traffic = None
        if tj.get("_kernel_source_sha") == now and not args.block_bounds and not args.hiz_rg16f and "cull_kernel_hbm_bytes_per_launch" in entry:
            traffic = entry["cull_kernel_hbm_bytes_per_launch"]
            measured_n = entry.get("entities", 10_000_000)
            if measured_n != n:  # counters were taken at another pool size of the same streaming kernel
                traffic = traffic * n / measured_n
                source["per_entity_scaled"] = True
        return traffic, source
    except Exception:  # noqa: BLE001
        return None, None
