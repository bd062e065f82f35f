"""ctypes wrapper of the CPU oracle (oracle/gv_oracle.c). TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; nothing in
garden_amd/ does. Parity unpinned: see the header of gv_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "build", "libgv_oracle.so")
GVO_NONE = 0xFFFFFFFF
RULE_REFERENCE, RULE_CONSERVATIVE = 0, 1
FORMAT_RG16F = 0x100  # OR-ed into the rule: levels >= 1 hold RG16F-representable values, rounded outward


class GvoMeshPool(C.Structure):
    _fields_ = [("base", C.c_void_p), ("stride", C.c_size_t), ("occupancy", C.c_uint32), ("off_entity", C.c_uint32),
                ("off_is_enabled", C.c_uint32), ("off_is_visible", C.c_uint32), ("off_aabb_min", C.c_uint32),
                ("off_aabb_max", C.c_uint32), ("ready_base", C.c_void_p), ("ready_stride", C.c_size_t),
                ("ready_width", C.c_uint32)]


class GvoTransformPool(C.Structure):
    _fields_ = [("base", C.c_void_p), ("stride", C.c_size_t), ("occupancy", C.c_uint32), ("off_entity", C.c_uint32),
                ("off_parent", C.c_uint32), ("off_position", C.c_uint32), ("off_scale", C.c_uint32),
                ("off_rotation", C.c_uint32), ("off_self_active", C.c_uint32), ("off_ancestors_active", C.c_uint32),
                ("off_model_with_ancestors", C.c_uint32), ("entity_to_transform", C.c_void_p),
                ("entity_capacity", C.c_uint32)]


class GvoFrustum(C.Structure):
    _fields_ = [("planes", (C.c_float * 4) * 6), ("count", C.c_uint32)]


class GvoHiz(C.Structure):
    _fields_ = [("depth", C.c_void_p), ("mips", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32),
                ("mip_count", C.c_uint32), ("mip_w", C.c_uint32 * 16), ("mip_h", C.c_uint32 * 16),
                ("mip_offset", C.c_uint64 * 16)]


class GvoView(C.Structure):
    _fields_ = [("view_proj", C.c_float * 16), ("camera_position", C.c_float * 4), ("camera_offset", C.c_float * 4),
                ("shadow_pass", C.c_int8), ("use_hiz", C.c_uint8), ("distance_2d", C.c_uint8), ("reserved", C.c_uint8)]


class GvoCullOut(C.Structure):
    _fields_ = [("visible_idx", C.c_void_p), ("baked_model", C.c_void_p), ("distance_sq", C.c_void_p),
                ("draw_count", C.c_uint32), ("instance_count", C.c_uint32)]


_lib = None


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    lib.gvo_calc_model.argtypes = [C.c_void_p] * 4
    lib.gvo_mul4x4.argtypes = [C.c_void_p] * 3
    lib.gvo_frustum_from_view_proj.argtypes = [C.c_void_p, C.POINTER(GvoFrustum)]
    lib.gvo_is_behind_frustum.argtypes = [C.POINTER(GvoFrustum), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gvo_is_behind_frustum.restype = C.c_int
    lib.gvo_transform_calc_model.argtypes = [C.POINTER(GvoTransformPool), C.c_uint32, C.c_void_p, C.c_void_p]
    lib.gvo_world_matrices.argtypes = [C.POINTER(GvoTransformPool), C.c_uint32, C.c_uint32, C.c_void_p]
    lib.gvo_calc_mip_count.argtypes = [C.c_uint32, C.c_uint32]
    lib.gvo_calc_mip_count.restype = C.c_uint32
    lib.gvo_half_directed.argtypes = [C.c_float, C.c_int]
    lib.gvo_half_directed.restype = C.c_uint16
    lib.gvo_half_to_float.argtypes = [C.c_uint16]
    lib.gvo_half_to_float.restype = C.c_float
    lib.gvo_hiz_layout.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(GvoHiz)]
    lib.gvo_hiz_layout.restype = C.c_uint64
    lib.gvo_hiz_build.argtypes = [C.POINTER(GvoHiz), C.c_void_p, C.c_int]
    lib.gvo_hiz_build_mt.argtypes = [C.POINTER(GvoHiz), C.c_void_p, C.c_int, C.c_uint32]
    lib.gvo_world_matrices_mt.argtypes = [C.POINTER(GvoTransformPool), C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
    lib.gvo_hiz_occluded.argtypes = [C.POINTER(GvoHiz), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gvo_hiz_occluded.restype = C.c_int
    lib.gvo_prepare_meshes.argtypes = [C.POINTER(GvoMeshPool), C.POINTER(GvoTransformPool), C.POINTER(GvoView),
                                       C.POINTER(GvoHiz), C.c_uint32, C.POINTER(GvoCullOut)]
    lib.gvo_sort_records.argtypes = [C.POINTER(GvoCullOut), C.c_int]
    lib.gvo_soa_build.argtypes = [C.POINTER(GvoMeshPool), C.POINTER(GvoTransformPool)]
    lib.gvo_soa_build.restype = C.c_void_p
    lib.gvo_soa_build_threads.argtypes = [C.POINTER(GvoMeshPool), C.POINTER(GvoTransformPool), C.c_uint32]
    lib.gvo_soa_build_threads.restype = C.c_void_p
    lib.gvo_soa_free.argtypes = [C.c_void_p]
    lib.gvo_prepare_meshes_avx2.argtypes = [C.c_void_p, C.POINTER(GvoMeshPool), C.POINTER(GvoView), C.POINTER(GvoHiz),
                                            C.c_uint32, C.POINTER(GvoCullOut)]
    _lib = lib
    return lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def calc_model(pos, rot, scale):
    out = np.empty(16, np.float32)
    p, r, s = _f32(pos), _f32(rot), _f32(scale)
    load().gvo_calc_model(p.ctypes.data, r.ctypes.data, s.ctypes.data, out.ctypes.data)
    return out


def mul4x4(a, b):
    out = np.empty(16, np.float32)
    a, b = _f32(a), _f32(b)
    load().gvo_mul4x4(a.ctypes.data, b.ctypes.data, out.ctypes.data)
    return out


def frustum(view_proj):
    f = GvoFrustum()
    vp = _f32(view_proj)
    load().gvo_frustum_from_view_proj(vp.ctypes.data, C.byref(f))
    return np.array([[f.planes[i][c] for c in range(4)] for i in range(f.count)], dtype=np.float32)


def is_behind_frustum(view_proj, aabb_min, aabb_max, model):
    f = GvoFrustum()
    vp, mn, mx, m = _f32(view_proj), _f32(aabb_min), _f32(aabb_max), _f32(model)
    lib = load()
    lib.gvo_frustum_from_view_proj(vp.ctypes.data, C.byref(f))
    return bool(lib.gvo_is_behind_frustum(C.byref(f), mn.ctypes.data, mx.ctypes.data, m.ctypes.data))


def mesh_pool(meshes, ready=None):
    """ready: optional per-slot ready counts (numpy u8 / u32, kept alive by the caller): the derived predicate's result"""
    f = meshes.dtype.fields
    mp = GvoMeshPool(meshes.ctypes.data, meshes.dtype.itemsize, meshes.shape[0], f["entity"][1], f["isEnabled"][1],
                     f["isVisible"][1], f["aabbMin"][1], f["aabbMax"][1], None, 0, 0)
    if ready is not None:
        mp.ready_base, mp.ready_stride, mp.ready_width = ready.ctypes.data, ready.strides[0], ready.dtype.itemsize
    return mp


def transform_pool(transforms, e2t):
    f = transforms.dtype.fields
    return GvoTransformPool(transforms.ctypes.data, transforms.dtype.itemsize, transforms.shape[0], f["entity"][1],
                            f["parent"][1], f["position"][1], f["scale"][1], f["rotation"][1], f["selfActive"][1],
                            f["ancestorsActive"][1], f["modelWithAncestors"][1], e2t.ctypes.data, e2t.shape[0])


def to_view(v):
    out = GvoView()
    out.view_proj[:] = [float(x) for x in v["view_proj"]]
    out.camera_position[:] = [float(x) for x in v["camera_position"]]
    out.camera_offset[:] = [float(x) for x in v["camera_offset"]]
    out.shadow_pass = v.get("shadow_pass", -1)
    out.use_hiz = v.get("use_hiz", 0)
    out.distance_2d = v.get("distance_2d", 0)
    return out


class Hiz:
    """Pyramid built by the oracle (hiz.frag:23-63)."""

    def __init__(self, depth, rule=RULE_REFERENCE, threads=1, rg16f=False):
        lib = load()
        self.depth = _f32(depth)
        h, w = self.depth.shape
        self.c = GvoHiz()
        pairs = lib.gvo_hiz_layout(w, h, C.byref(self.c))
        self.mips = np.zeros((max(int(pairs), 1), 2), dtype=np.float32)
        self.c.depth = self.depth.ctypes.data
        self.rule = rule | (FORMAT_RG16F if rg16f else 0)
        self.rebuild(threads)
        self.mip_count = self.c.mip_count

    def rebuild(self, threads=1):
        """Re-runs the reduction into the same storage (one pass per mip, rows split over `threads` threads)."""
        if threads > 1:
            load().gvo_hiz_build_mt(C.byref(self.c), self.mips.ctypes.data, self.rule, threads)
        else:
            load().gvo_hiz_build(C.byref(self.c), self.mips.ctypes.data, self.rule)

    def level(self, k):
        w, h, off = self.c.mip_w[k], self.c.mip_h[k], self.c.mip_offset[k]
        if k == 0:
            return np.stack([self.depth, self.depth], axis=-1)
        return self.mips[off:off + w * h].reshape(h, w, 2)

    def occluded(self, view_proj, aabb_min, aabb_max, model):
        vp, mn, mx, m = _f32(view_proj), _f32(aabb_min), _f32(aabb_max), _f32(model)
        return bool(load().gvo_hiz_occluded(C.byref(self.c), vp.ctypes.data, mn.ctypes.data, mx.ctypes.data,
                                            m.ctypes.data))


def world_matrices(transforms, e2t, first=0, count=None, threads=1, out=None):
    count = transforms.shape[0] - first if count is None else count
    e2t = np.ascontiguousarray(e2t, dtype=np.uint32)
    tp = transform_pool(transforms, e2t)
    if out is None:
        out = np.empty((count, 12), dtype=np.float32)
    if threads > 1:
        load().gvo_world_matrices_mt(C.byref(tp), first, count, out.ctypes.data, threads)
    else:
        load().gvo_world_matrices(C.byref(tp), first, count, out.ctypes.data)
    return out


def transform_calc_model(transforms, e2t, slot, camera_position=(0, 0, 0)):
    e2t = np.ascontiguousarray(e2t, dtype=np.uint32)
    tp = transform_pool(transforms, e2t)
    cam = _f32(camera_position)
    out = np.empty(16, np.float32)
    load().gvo_transform_calc_model(C.byref(tp), slot, cam.ctypes.data, out.ctypes.data)
    return out


def prepare_meshes(meshes, transforms, e2t, view, hiz=None, threads=1, sort=None, ready=None):
    """MeshRenderSystem::prepareMeshes for one pool and one view (mesh.cpp:331-553 -> :111-184).
    Writes isVisible into `meshes` in place on a main pass, exactly as the reference does.
    Returns dict(visible_idx, baked_model[n,12], distance_sq, draw_count, instance_count)."""
    lib = load()
    e2t = np.ascontiguousarray(e2t, dtype=np.uint32)
    mp, tp, gv = mesh_pool(meshes, ready), transform_pool(transforms, e2t), to_view(view)
    n = meshes.shape[0]
    idx = np.empty(max(n, 1), np.uint32)
    bm = np.empty((max(n, 1), 12), np.float32)
    ds = np.empty(max(n, 1), np.float32)
    out = GvoCullOut(idx.ctypes.data, bm.ctypes.data, ds.ctypes.data, 0, 0)
    lib.gvo_prepare_meshes(C.byref(mp), C.byref(tp), C.byref(gv), C.byref(hiz.c) if hiz is not None else None,
                           threads, C.byref(out))
    if sort is not None:
        lib.gvo_sort_records(C.byref(out), 1 if sort == "descending" else 0)
    k = out.draw_count
    return dict(visible_idx=idx[:k].copy(), baked_model=bm[:k].copy(), distance_sq=ds[:k].copy(), draw_count=k,
                instance_count=out.instance_count)


class Avx2Scene:
    """SoA copy of the pools for the AVX2 path (built once, like the GPU mirror)."""

    def __init__(self, meshes, transforms, e2t, ready=None, threads=1):
        """threads > 1: the arrays are filled by that many workers over the ranges prepare_meshes(threads=...) will hand them
        (first touch by the thread that culls the range: the pages land on its NUMA node)."""
        self.lib = load()
        self.meshes, self.transforms, self.ready = meshes, transforms, ready
        self.e2t = np.ascontiguousarray(e2t, dtype=np.uint32)
        self.mp, self.tp = mesh_pool(meshes, ready), transform_pool(transforms, self.e2t)
        self.soa = self.lib.gvo_soa_build_threads(C.byref(self.mp), C.byref(self.tp), max(1, int(threads)))
        n = max(meshes.shape[0], 1)
        self.idx = np.empty(n, np.uint32)
        self.bm = np.empty((n, 12), np.float32)
        self.ds = np.empty(n, np.float32)

    def prepare_meshes(self, view, hiz=None, threads=1):
        gv = to_view(view)
        out = GvoCullOut(self.idx.ctypes.data, self.bm.ctypes.data, self.ds.ctypes.data, 0, 0)
        self.lib.gvo_prepare_meshes_avx2(self.soa, C.byref(self.mp), C.byref(gv),
                                         C.byref(hiz.c) if hiz is not None else None, threads, C.byref(out))
        k = out.draw_count
        return dict(visible_idx=self.idx[:k], baked_model=self.bm[:k], distance_sq=self.ds[:k], draw_count=k,
                    instance_count=out.instance_count)

    def close(self):
        if self.soa:
            self.lib.gvo_soa_free(self.soa)
            self.soa = None

    __del__ = close
