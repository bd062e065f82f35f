"""gv_pool_set_record_target through the bare C-ABI (no shim): the engine's combinedMeshes as an anonymous mapping this script
controls. Cases: `grow` (valid: clear the target, unmap, map a larger array, set it again, every frame), `replace` (valid: set
the new array while the old one is still mapped, unmap afterwards), `early_free` (INVALID: unmap the registered array, then set
another one: the library should say so — GV_E_STATE — and carry on). Exit code 0 = behaved; the caller counts aborts.
    python tools/record_target_probe.py grow|replace|early_free [frames]"""
import ctypes as C
import mmap
import sys

sys.path.insert(0, '.')
import numpy as np

from garden_amd import scene
from garden_amd.lib import GpuVisibility, GvError, GV_E_STATE

case, frames = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12
n = 60_000
dt = np.dtype([("componentOffset", "<u8"), ("bakedModel", "<f4", (12,)), ("distanceSq", "<f4"), ("pad", "<u4")])  # 64 bytes
assert dt.itemsize == 64
sc = scene.flat_scene(n, seed=3)
view = scene.main_camera_view()


class Mapping:
    def __init__(self, records):
        self.bytes = records * dt.itemsize
        self.m = mmap.mmap(-1, self.bytes)
        self.address = C.addressof(C.c_char.from_buffer(self.m))

    def array(self):
        return np.frombuffer(self.m, dtype=dt)

    def unmap(self):
        self.m.close()  # (raises if a numpy view of it is still alive: the callers below drop theirs first)


def set_target(vis, mapping):
    return vis.lib.gv_pool_set_record_target(vis.ctx, 0, 0, C.c_void_p(mapping.address if mapping else None), mapping.bytes if mapping else 0)


with GpuVisibility(device=0) as vis:
    vis.bind_transforms(sc.transforms, sc.entity_to_transform)
    vis.bind_pool(0, sc.meshes)
    vis.hierarchy_rebuild()
    vis.set_record_layout(0, dt, component_stride=int(sc.meshes.dtype.itemsize))
    vis.cull(0, [view])
    vis.sort(0, pool_id=0)
    vis.fetch(0, write_back=False, occupancy=n, pool_id=0)
    ref = vis.records(0, 0, dt).copy()
    assert ref.shape[0] > 1000

    def frame_into(mapping):
        vis.cull(0, [view])
        vis.sort(0, pool_id=0)
        got = vis.fetch(0, write_back=False, occupancy=n, pool_id=0)
        a = mapping.array()
        ok = got["draw_count"] == ref.shape[0] and np.array_equal(a[:ref.shape[0]].view(np.uint8), ref.view(np.uint8))
        del a
        return ok

    cur = Mapping(n)
    assert set_target(vis, cur) == 0
    detected = 0
    for f in range(frames):
        assert frame_into(cur), f"frame {f}: records differ"
        nxt = Mapping(n + 4096 * (f + 1))  # the vector grows
        if case == "grow":
            assert set_target(vis, None) == 0
            cur.unmap()
            assert set_target(vis, nxt) == 0
        elif case == "replace":
            assert set_target(vis, nxt) == 0
            cur.unmap()
        elif case == "early_free":
            vis.wait()
            cur.unmap()  # the registered range goes away under the library
            before = vis.stats()["record_targets_lost"]
            assert set_target(vis, nxt) == 0  # the new target is in place: the lost range is counted, not an error return
            detected += vis.stats()["record_targets_lost"] - before
        cur = nxt
    assert frame_into(cur)
    set_target(vis, None)
print(f"{case}: {frames} frames ok" + (f", lost registration reported {detected} times" if case == "early_free" else ""))
