"""ctypes binding of libgarden_vis.so (C-ABI: include/garden_vis.h).

Plumbing for tests and bench.py. Loading fails loudly when the HIP library is missing; creating a
context fails loudly (GvError) when there is no gfx950 device — there is no CPU fallback anywhere in
this package.
"""
import ctypes as C
import os

import numpy as np

from .pools import mesh_layout_offsets, transform_layout_offsets

_HERE = os.path.dirname(os.path.abspath(__file__))
# GV_LIB_PATH: dev A/Bs against another build of the same library (e.g. a previous round's kernels); never a fallback
LIB_PATH = os.environ.get("GV_LIB_PATH") or os.path.join(_HERE, "lib", "libgarden_vis.so")

GV_MAX_POOLS, GV_MAX_VIEWS, GV_MAX_MIPS, GV_K_COUNT = 16, 8, 16, 6
GV_OK, GV_E_ARG, GV_E_HIP, GV_E_OOM, GV_E_RCCL, GV_E_STATE, GV_E_NODEVICE, GV_E_TIMEOUT = 0, -1, -2, -3, -4, -5, -6, -7
GV_HIZ_RULE_REFERENCE, GV_HIZ_RULE_CONSERVATIVE = 0, 1
GV_CONFIG_PROFILE_EVENTS = 1
GV_CONFIG_PROFILE_CULL_ONLY = 2
GV_CONFIG_KEEP_SLOT_ORDER = 4
GV_CONFIG_BLOCK_BOUNDS = 8
GV_CONFIG_HIZ_RG16F = 16
GV_CONFIG_LINEAR_SCAN = 32
GV_DIRTY_TRANSFORM, GV_DIRTY_HIERARCHY, GV_DIRTY_MESH = 0, 1, 2
GV_SWEEP_VALU, GV_SWEEP_MFMA, GV_SWEEP_WITH_CULL, GV_SWEEP_WITH_CULL_VALU, GV_SWEEP_INCREMENTAL = 0, 1, 2, 3, 4
GV_MEM_HOST, GV_MEM_DEVICE = 0, 1
GV_EXCHANGE_ALLGATHER, GV_EXCHANGE_P2P, GV_EXCHANGE_BROADCAST, GV_EXCHANGE_PEER = 0, 1, 2, 3
KERNEL_NAMES = ["cull", "scan", "emit", "hiz", "sweep", "sort"]


class GvConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("hiz_rule", C.c_uint32), ("flags", C.c_uint32)]


class GvTransformLayout(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("entity", "parent", "position", "scale", "rotation", "self_active",
                                          "ancestors_active", "model_with_ancestors")]


class GvMeshLayout(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("entity", "is_enabled", "is_visible", "aabb_min", "aabb_max")]


class GvView(C.Structure):
    _fields_ = [("view_proj", C.c_float * 16), ("camera_position", C.c_float * 4), ("camera_offset", C.c_float * 4),
                ("shadow_pass", C.c_int8), ("use_hiz", C.c_uint8), ("distance_2d", C.c_uint8),
                ("emit_records", C.c_uint8)]


class GvResult(C.Structure):
    _fields_ = [("visible_idx", C.POINTER(C.c_uint32)), ("baked_model", C.POINTER(C.c_float)),
                ("distance_sq", C.POINTER(C.c_float)), ("is_visible", C.POINTER(C.c_uint8)),
                ("draw_count", C.c_uint32), ("instance_count", C.c_uint32)]


class GvRecordLayout(C.Structure):
    _fields_ = [("stride", C.c_uint32), ("component_offset", C.c_uint32), ("baked_model", C.c_uint32), ("distance_sq", C.c_uint32),
                ("buffer_index", C.c_uint32), ("component_stride", C.c_uint32), ("buffer_index_value", C.c_uint32)]


class GvDeviceResult(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("visible_idx", "baked_model", "distance_sq", "is_visible", "draw_count")]


class GvStats(C.Structure):
    _fields_ = [("launches", C.c_uint64 * GV_K_COUNT), ("device_ms", C.c_double * GV_K_COUNT),
                ("upload_bytes", C.c_uint64), ("max_depth", C.c_uint32), ("transform_count", C.c_uint32),
                ("mesh_count", C.c_uint32 * GV_MAX_POOLS), ("bounds_blocks_total", C.c_uint64),
                ("bounds_blocks_examined", C.c_uint64), ("mirror_reorders", C.c_uint64), ("record_targets_lost", C.c_uint64),
                ("exchanges", C.c_uint64), ("exchange_tail_rounds", C.c_uint64)]


GV_EXCHANGE_MAX_RANKS = 64


class GvExchangeFrame(C.Structure):
    _fields_ = [("gathered_device", C.c_void_p), ("row_words", C.c_uint32), ("world_size", C.c_uint32), ("frame", C.c_uint64),
                ("room", C.c_uint32 * GV_EXCHANGE_MAX_RANKS), ("travelled_words", C.c_uint32 * GV_EXCHANGE_MAX_RANKS),
                ("counts", C.c_uint32 * GV_EXCHANGE_MAX_RANKS), ("tail_words", C.c_uint32 * GV_EXCHANGE_MAX_RANKS), ("cut_ranks", C.c_uint64),
                ("complete", C.c_uint32), ("mode", C.c_uint32), ("ready_event", C.c_void_p),
                ("items", C.c_uint32), ("item_counts", C.POINTER(C.c_uint32))]


GV_EXCHANGE_MAX_ITEMS = 128
GV_RESULTS_MAP_RECORDS, GV_RESULTS_MAP_VISIBLE = 1, 2


class GvExchangeItem(C.Structure):
    _fields_ = [("pool_id", C.c_uint32), ("view_index", C.c_uint32), ("index_base", C.c_uint32)]


class GvColumn(C.Structure):
    _fields_ = [("data", C.c_void_p), ("stride", C.c_uint32)]


class GvTransformColumns(C.Structure):
    _fields_ = [(n, GvColumn) for n in ("entity", "parent", "position", "scale", "rotation", "self_active",
                                        "ancestors_active", "model_with_ancestors")]


class GvMeshColumns(C.Structure):
    _fields_ = [("entity", GvColumn), ("is_enabled", GvColumn), ("aabb_min", GvColumn), ("aabb_max", GvColumn),
                ("is_visible", C.c_void_p), ("is_visible_stride", C.c_uint32)]


class GvScenePool(C.Structure):
    _fields_ = [("component_type", C.c_char_p), ("pool_id", C.c_uint32)]


class GvSceneInfo(C.Structure):
    _fields_ = [("entity_count", C.c_uint32), ("transform_count", C.c_uint32), ("mesh_count", C.c_uint32 * GV_MAX_POOLS),
                ("skipped_entities", C.c_uint32), ("other_components", C.c_uint32), ("duplicate_uids", C.c_uint32),
                ("self_parents", C.c_uint32), ("unresolved_parents", C.c_uint32)]


class GvError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"libgarden_vis error {code}: {text}")
        self.code = code


# every symbol include/garden_vis.h declares; tests/test_abi.py checks the .so exports all of them
EXPORTS = [
    "gv_abi_version", "gv_create", "gv_destroy", "gv_last_error", "gv_transform_bind", "gv_pool_bind",
    "gv_transform_bind_columns", "gv_pool_bind_columns", "gv_pool_bind_ready",
    "gv_mark_dirty", "gv_hierarchy_rebuild", "gv_sync", "gv_cull", "gv_wait", "gv_results_fetch",
    "gv_result_count", "gv_results_device", "gv_results_copy_idx_device", "gv_results_copy_shard_device", "gv_results_copy_mask_device", "gv_pool_mirror_slots", "gv_pool_mirror_epoch", "gv_sort", "gv_sweep", "gv_get_world",
    "gv_hiz_build", "gv_hiz_rebuild", "gv_hiz_read_level", "gv_hiz_mip_count", "gv_stats", "gv_stats_reset",
    "gv_stream", "gv_debug_stream_peak",
    "gv_scene_parse_json", "gv_scene_parse_bson", "gv_scene_destroy", "gv_scene_info", "gv_scene_transform_columns", "gv_scene_mesh_columns",
    "gv_scene_bind", "gv_scene_extract_tile", "gv_scene_extract_rank", "gv_cell_owner", "gv_scene_tile_maps",
    "gv_exchange_unique_id", "gv_exchange_init", "gv_exchange_init_all", "gv_exchange_init_peers", "gv_exchange_shards", "gv_exchange_visible", "gv_exchange_visible_all", "gv_pool_exchange_visible", "gv_pool_exchange_visible_all", "gv_exchange_acquire", "gv_exchange_acquire_all", "gv_exchange_set_timeout", "gv_exchange_masks", "gv_exchange_shutdown", "gv_exchange_set_mode", "gv_pool_set_index_map",
    "gv_exchange_views", "gv_exchange_views_all", "gv_pool_update_index_map", "gv_pool_set_result_mapping", "gv_host_parallel_ranges", "gv_host_parallel_tasks",
    "gv_pool_results_fetch", "gv_pool_result_count", "gv_pool_results_device", "gv_pool_sort",
    "gv_cull_batch_begin", "gv_cull_batch_end", "gv_pool_set_record_layout", "gv_pool_results_records", "gv_pool_set_record_target",
    "gv_pool_results_instance_bases", "gv_profile_sampling", "gv_profile_samples", "gv_profile_kernels",
]

_lib = None


def load():
    """dlopen the HIP library (no compute). Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: build it with `make -C garden_amd/csrc` (or __graft_entry__.build()); "
            "there is no CPU fallback")
    # One HIP runtime per process: PyTorch bundles its own libamdhip64, and a process that loads the system one first
    # (through this library) and torch's second ends up with two runtimes, the second of which finds no GPU. Loading
    # torch's first makes this library bind to it as well (same SONAME). torch stays optional.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    P, u32, sz = C.c_void_p, C.c_uint32, C.c_size_t
    lib.gv_abi_version.restype = u32
    lib.gv_create.argtypes = [C.POINTER(GvConfig), C.POINTER(P)]
    lib.gv_destroy.argtypes = [P]
    lib.gv_destroy.restype = None
    lib.gv_last_error.argtypes = [P]
    lib.gv_last_error.restype = C.c_char_p
    lib.gv_transform_bind.argtypes = [P, P, sz, u32, C.POINTER(GvTransformLayout), P, u32]
    lib.gv_pool_bind.argtypes = [P, u32, P, sz, u32, C.POINTER(GvMeshLayout)]
    lib.gv_transform_bind_columns.argtypes = [P, C.POINTER(GvTransformColumns), u32, P, u32]
    lib.gv_pool_bind_columns.argtypes = [P, u32, C.POINTER(GvMeshColumns), u32]
    lib.gv_pool_bind_ready.argtypes = [P, u32, P, u32, u32]
    lib.gv_mark_dirty.argtypes = [P, u32, u32, u32]
    lib.gv_hierarchy_rebuild.argtypes = [P]
    lib.gv_sync.argtypes = [P]
    lib.gv_cull.argtypes = [P, u32, C.POINTER(GvView), u32]
    lib.gv_wait.argtypes = [P]
    lib.gv_results_fetch.argtypes = [P, u32, C.c_int, C.POINTER(GvResult)]
    lib.gv_result_count.argtypes = [P, u32, C.POINTER(u32)]
    lib.gv_results_device.argtypes = [P, u32, C.POINTER(GvDeviceResult)]
    lib.gv_results_copy_idx_device.argtypes = [P, u32, P, u32, u32]
    lib.gv_results_copy_shard_device.argtypes = [P, u32, P, u32, u32]
    lib.gv_results_copy_mask_device.argtypes = [P, u32, P, u32]
    lib.gv_pool_mirror_slots.argtypes = [P, u32, P, u32]
    lib.gv_pool_mirror_epoch.argtypes = [P, u32, P]
    lib.gv_sort.argtypes = [P, u32, C.c_int]
    lib.gv_sweep.argtypes = [P, u32]
    lib.gv_get_world.argtypes = [P, u32, u32, P]
    lib.gv_hiz_build.argtypes = [P, P, u32, u32, u32]
    lib.gv_hiz_rebuild.argtypes = [P]
    lib.gv_hiz_read_level.argtypes = [P, u32, P, C.POINTER(u32), C.POINTER(u32)]
    lib.gv_hiz_mip_count.argtypes = [P, C.POINTER(u32)]
    lib.gv_stats.argtypes = [P, C.POINTER(GvStats)]
    lib.gv_stats_reset.argtypes = [P]
    lib.gv_debug_stream_peak.argtypes = [P, u32, u32, C.POINTER(C.c_double)]
    lib.gv_stream.argtypes = [P]
    lib.gv_stream.restype = P
    lib.gv_scene_parse_json.argtypes = [C.c_char_p, sz, C.POINTER(GvScenePool), u32, u32, C.POINTER(P), C.c_char_p, sz]
    lib.gv_scene_parse_bson.argtypes = [C.c_char_p, sz, C.POINTER(GvScenePool), u32, u32, C.POINTER(P), C.c_char_p, sz]
    lib.gv_scene_destroy.argtypes = [P]
    lib.gv_scene_destroy.restype = None
    lib.gv_scene_info.argtypes = [P, C.POINTER(GvSceneInfo)]
    lib.gv_scene_transform_columns.argtypes = [P, C.POINTER(GvTransformColumns), C.POINTER(u32), C.POINTER(P),
                                               C.POINTER(u32), C.POINTER(P)]
    lib.gv_scene_mesh_columns.argtypes = [P, u32, C.POINTER(GvMeshColumns), C.POINTER(u32)]
    lib.gv_scene_bind.argtypes = [P, P]
    lib.gv_scene_extract_tile.argtypes = [P, C.POINTER(C.c_uint32 * 3), C.c_double, u32, C.POINTER(P)]
    lib.gv_scene_extract_rank.argtypes = [P, C.POINTER(C.c_uint32 * 3), C.c_double, u32, u32, C.POINTER(P)]
    lib.gv_cell_owner.argtypes = [C.POINTER(C.c_uint32 * 3), C.c_double, u32, P, u32, u32, P]
    lib.gv_scene_tile_maps.argtypes = [P, u32, C.POINTER(P), C.POINTER(u32), C.POINTER(P), C.POINTER(u32)]
    lib.gv_exchange_unique_id.argtypes = [P]
    lib.gv_exchange_init.argtypes = [P, P, C.c_int, C.c_int]
    lib.gv_exchange_shards.argtypes = [P, u32, u32, P, u32, P]
    lib.gv_exchange_visible.argtypes = [P, u32, u32, u32, C.POINTER(GvExchangeFrame)]
    lib.gv_exchange_acquire.argtypes = [P, C.c_uint64, C.POINTER(GvExchangeFrame)]
    lib.gv_exchange_init_all.argtypes = [C.POINTER(P), C.c_int]
    lib.gv_exchange_init_peers.argtypes = [C.POINTER(P), C.c_int]
    lib.gv_exchange_visible_all.argtypes = [C.POINTER(P), C.c_int, C.POINTER(u32), C.POINTER(u32), u32, C.POINTER(GvExchangeFrame)]
    lib.gv_pool_exchange_visible.argtypes = [P, u32, u32, u32, u32, C.POINTER(GvExchangeFrame)]
    lib.gv_pool_exchange_visible_all.argtypes = [C.POINTER(P), C.c_int, u32, C.POINTER(u32), C.POINTER(u32), u32, C.POINTER(GvExchangeFrame)]
    lib.gv_exchange_acquire_all.argtypes = [C.POINTER(P), C.c_int, C.c_uint64, C.POINTER(GvExchangeFrame)]
    lib.gv_exchange_views.argtypes = [P, C.POINTER(GvExchangeItem), u32, u32, C.POINTER(GvExchangeFrame)]
    lib.gv_exchange_views_all.argtypes = [C.POINTER(P), C.c_int, C.POINTER(GvExchangeItem), u32, u32, C.POINTER(GvExchangeFrame)]
    lib.gv_pool_update_index_map.argtypes = [P, u32, u32, C.POINTER(u32), u32]
    lib.gv_pool_set_result_mapping.argtypes = [P, u32, u32, C.c_void_p, C.c_size_t, u32]
    lib.gv_host_parallel_ranges.argtypes = [u32, u32, C.c_void_p, C.c_void_p]
    lib.gv_host_parallel_ranges.restype = None
    lib.gv_host_parallel_tasks.argtypes = [u32, C.c_void_p, C.c_void_p]
    lib.gv_host_parallel_tasks.restype = None
    lib.gv_exchange_set_timeout.argtypes = [P, u32]
    lib.gv_exchange_masks.argtypes = [P, u32, u32, P]
    lib.gv_exchange_shutdown.argtypes = [P]
    lib.gv_exchange_set_mode.argtypes = [P, u32]
    lib.gv_pool_set_index_map.argtypes = [P, u32, P, u32]
    lib.gv_pool_results_fetch.argtypes = [P, u32, u32, C.c_int, C.POINTER(GvResult)]
    lib.gv_pool_result_count.argtypes = [P, u32, u32, C.POINTER(u32)]
    lib.gv_pool_results_device.argtypes = [P, u32, u32, C.POINTER(GvDeviceResult)]
    lib.gv_pool_sort.argtypes = [P, u32, u32, C.c_int]
    lib.gv_pool_set_record_layout.argtypes = [P, u32, C.POINTER(GvRecordLayout)]
    lib.gv_pool_results_records.argtypes = [P, u32, u32, C.POINTER(C.c_void_p), C.POINTER(u32)]
    lib.gv_pool_set_record_target.argtypes = [P, u32, u32, C.c_void_p, C.c_size_t]
    lib.gv_profile_sampling.argtypes = [P, u32]
    lib.gv_profile_samples.argtypes = [P, C.POINTER(C.c_uint64 * GV_K_COUNT)]
    lib.gv_profile_kernels.argtypes = [P, u32]
    lib.gv_pool_results_instance_bases.argtypes = [P, u32, u32, C.POINTER(C.POINTER(u32)), C.POINTER(u32)]
    lib.gv_cull_batch_begin.argtypes = [P]
    lib.gv_cull_batch_end.argtypes = [P]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("gv_abi_version", "gv_destroy", "gv_last_error", "gv_stream", "gv_scene_destroy"):
            fn.restype = C.c_int
    _lib = lib
    return lib


def to_gv_view(v):
    out = GvView()
    out.view_proj[:] = [float(x) for x in v["view_proj"]]
    out.camera_position[:] = [float(x) for x in v["camera_position"]]
    out.camera_offset[:] = [float(x) for x in v["camera_offset"]]
    out.shadow_pass = v.get("shadow_pass", -1)
    out.use_hiz = v.get("use_hiz", 0)
    out.distance_2d = v.get("distance_2d", 0)
    out.emit_records = v.get("emit_records", 1)
    return out


class GpuVisibility:
    """One libgarden_vis context (one per process per GPU). Thin: every method is one C-ABI call."""

    def __init__(self, device=0, hiz_rule=GV_HIZ_RULE_REFERENCE, profile_events=False, profile_cull_only=False,
                 keep_slot_order=False, block_bounds=False, hiz_rg16f=False, linear_scan=False):
        self.lib = load()
        flags = GV_CONFIG_PROFILE_EVENTS if (profile_events or profile_cull_only) else 0
        if profile_cull_only:
            flags |= GV_CONFIG_PROFILE_CULL_ONLY
        if keep_slot_order:
            flags |= GV_CONFIG_KEEP_SLOT_ORDER
        if block_bounds:
            flags |= GV_CONFIG_BLOCK_BOUNDS
        if hiz_rg16f:
            flags |= GV_CONFIG_HIZ_RG16F
        if linear_scan:
            flags |= GV_CONFIG_LINEAR_SCAN
        cfg = GvConfig(C.sizeof(GvConfig), device, hiz_rule, flags)
        self.ctx = C.c_void_p()
        rc = self.lib.gv_create(C.byref(cfg), C.byref(self.ctx))
        if rc != GV_OK:
            self.ctx = C.c_void_p()
            raise GvError(rc, self.lib.gv_last_error(None).decode())
        self._keep = {}

    def close(self):
        if getattr(self, "ctx", None) and self.ctx.value:
            self.lib.gv_destroy(self.ctx)
            self.ctx = C.c_void_p()
            self._keep = {}  # bound pools and record targets are let go only after the context (and its page locks) is gone

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != GV_OK:
            raise GvError(rc, self.lib.gv_last_error(self.ctx).decode())

    # ---- binding ----
    def bind_transforms(self, transforms, entity_to_transform):
        lay = GvTransformLayout(**transform_layout_offsets(transforms.dtype))
        e2t = np.ascontiguousarray(entity_to_transform, dtype=np.uint32)
        self._keep["xf"] = (transforms, e2t)
        self._check(self.lib.gv_transform_bind(self.ctx, transforms.ctypes.data, transforms.dtype.itemsize,
                                               transforms.shape[0], C.byref(lay), e2t.ctypes.data, e2t.shape[0]))

    def bind_pool(self, pool_id, meshes):
        lay = GvMeshLayout(**mesh_layout_offsets(meshes.dtype))
        self._keep[("pool", pool_id)] = meshes
        self._check(self.lib.gv_pool_bind(self.ctx, pool_id, meshes.ctypes.data, meshes.dtype.itemsize,
                                          meshes.shape[0], C.byref(lay)))

    @staticmethod
    def _column(a):
        """GvColumn of a numpy array whose rows are the elements (C-contiguous rows; any row stride)."""
        assert a.ndim in (1, 2) and (a.ndim == 1 or a.strides[1] == a.itemsize)
        return GvColumn(a.ctypes.data, a.strides[0])

    def bind_transform_columns(self, columns, entity_to_transform):
        """columns: dict of numpy arrays entity/parent (u32[n]), position/scale (f32[n,3+]), rotation (f32[n,4]),
        self_active/ancestors_active/model_with_ancestors (u8[n]). The caller keeps them alive and unmoved."""
        self._xf_columns, self._e2t = columns, np.ascontiguousarray(entity_to_transform, dtype=np.uint32)
        c = GvTransformColumns(*[self._column(columns[n]) for n, _ in GvTransformColumns._fields_])
        self._check(self.lib.gv_transform_bind_columns(self.ctx, C.byref(c), columns["entity"].shape[0],
                                                       self._e2t.ctypes.data, self._e2t.shape[0]))

    def bind_pool_columns(self, pool_id, columns):
        """columns: dict of numpy arrays entity (u32[n]), is_enabled (u8[n]), aabb_min/aabb_max (f32[n,3+]) and
        optionally is_visible (u8[n], written by fetch(write_back=True))."""
        self._pool_columns = getattr(self, "_pool_columns", {})
        self._pool_columns[pool_id] = columns
        vis = columns.get("is_visible")
        c = GvMeshColumns(self._column(columns["entity"]), self._column(columns["is_enabled"]),
                          self._column(columns["aabb_min"]), self._column(columns["aabb_max"]),
                          vis.ctypes.data if vis is not None else None, vis.strides[0] if vis is not None else 0)
        self._check(self.lib.gv_pool_bind_columns(self.ctx, pool_id, C.byref(c), columns["entity"].shape[0]))

    def bind_ready(self, pool_id, ready):
        """Per-slot ready counts of `pool_id` (numpy u8 or u32 array, one element per slot; None removes the column)."""
        if ready is None:
            self._keep.pop(("ready", pool_id), None)
            self._check(self.lib.gv_pool_bind_ready(self.ctx, pool_id, None, 0, 0))
            return
        assert ready.dtype in (np.uint8, np.uint32) and ready.ndim == 1
        self._keep[("ready", pool_id)] = ready
        self._check(self.lib.gv_pool_bind_ready(self.ctx, pool_id, ready.ctypes.data, ready.strides[0], ready.dtype.itemsize))

    def mark_dirty(self, kind, first, count, pool_id=0):
        if kind == GV_DIRTY_MESH:
            first |= pool_id << 28
        self._check(self.lib.gv_mark_dirty(self.ctx, kind, first, count))

    def hierarchy_rebuild(self):
        self._check(self.lib.gv_hierarchy_rebuild(self.ctx))

    def sync(self):
        self._check(self.lib.gv_sync(self.ctx))

    # ---- per frame ----
    def cull(self, pool_id, views):
        """views: a list of view dicts (scene.make_view), or the array views_array() made of one (a frame loop that
        re-submits the same views need not rebuild the ctypes structs every frame)."""
        arr = views if isinstance(views, C.Array) else self.views_array(views)
        self._check(self.lib.gv_cull(self.ctx, pool_id, arr, len(arr)))

    @staticmethod
    def views_array(views):
        return (GvView * len(views))(*[to_gv_view(v) for v in views])

    def cull_batch_begin(self):
        """Culls of small pools are recorded until the first read and then launched together (one tick, four launches)."""
        self._check(self.lib.gv_cull_batch_begin(self.ctx))

    def cull_batch_end(self):
        self._check(self.lib.gv_cull_batch_end(self.ctx))

    def wait(self):
        self._check(self.lib.gv_wait(self.ctx))

    def result_count(self, view_index=0):
        n = C.c_uint32()
        self._check(self.lib.gv_result_count(self.ctx, view_index, C.byref(n)))
        return n.value

    def fetch(self, view_index=0, write_back=True, occupancy=None, order="slot", pool_id=None):
        """Returns copies: dict(visible_idx, baked_model[n,12], distance_sq, is_visible or None, draw_count).
        order="slot": records re-ordered here by pool slot (what the oracle's single-thread loop produces);
        order="raw": as the library emitted them (mirror order, or distance order after sort()).
        pool_id: the pool whose results are read (default: the pool of the most recent cull)."""
        r = GvResult()
        if pool_id is None:
            self._check(self.lib.gv_results_fetch(self.ctx, view_index, 1 if write_back else 0, C.byref(r)))
        else:
            self._check(self.lib.gv_pool_results_fetch(self.ctx, pool_id, view_index, 1 if write_back else 0, C.byref(r)))
        n = r.draw_count
        out = dict(draw_count=n, instance_count=r.instance_count, visible_idx=None, baked_model=None,
                   distance_sq=None, is_visible=None)
        if n and r.visible_idx:
            out["visible_idx"] = np.ctypeslib.as_array(r.visible_idx, shape=(n,)).copy()
            out["baked_model"] = np.ctypeslib.as_array(r.baked_model, shape=(n, 12)).copy()
            out["distance_sq"] = np.ctypeslib.as_array(r.distance_sq, shape=(n,)).copy()
            if order == "slot":
                o = np.argsort(out["visible_idx"], kind="stable")
                out["visible_idx"], out["baked_model"], out["distance_sq"] = \
                    out["visible_idx"][o], out["baked_model"][o], out["distance_sq"][o]
        elif n == 0:
            out["visible_idx"] = np.zeros(0, np.uint32)
            out["baked_model"] = np.zeros((0, 12), np.float32)
            out["distance_sq"] = np.zeros(0, np.float32)
        if r.is_visible and occupancy:
            out["is_visible"] = np.ctypeslib.as_array(r.is_visible, shape=(occupancy,)).copy()
        return out

    def set_record_layout(self, pool_id, dtype=None, component_stride=1, buffer_index_value=0):
        """Deliver pool `pool_id`'s records as an array of structs described by the numpy structured `dtype` with fields
        componentOffset (u8), bakedModel (12 x f4), distanceSq (f4) and optionally bufferIndex (u4); None removes it."""
        if dtype is None:
            self._check(self.lib.gv_pool_set_record_layout(self.ctx, pool_id, None))
            return
        dtype = np.dtype(dtype)
        at = {name: dtype.fields[name][1] for name in dtype.names}
        layout = GvRecordLayout(dtype.itemsize, at["componentOffset"], at["bakedModel"], at["distanceSq"],
                                at.get("bufferIndex", 0xFFFFFFFF), component_stride, buffer_index_value)
        self._check(self.lib.gv_pool_set_record_layout(self.ctx, pool_id, C.byref(layout)))

    def records(self, pool_id, view_index, dtype):
        """The fetched records of (pool, view) as a copy of the library's struct array, viewed as `dtype`."""
        ptr, n = C.c_void_p(), C.c_uint32()
        self._check(self.lib.gv_pool_results_records(self.ctx, pool_id, view_index, C.byref(ptr), C.byref(n)))
        dtype = np.dtype(dtype)
        if n.value == 0:
            return np.zeros(0, dtype)
        raw = (C.c_uint8 * (n.value * dtype.itemsize)).from_address(ptr.value)
        return np.frombuffer(raw, dtype=np.uint8).copy().view(dtype)  # bytewise: a structured copy would skip the padding

    def set_record_target(self, pool_id, view_index, array):
        """The records of (pool, view) are written straight into `array` (a C-contiguous numpy array the caller keeps alive
        and in place; None removes the target): the engine's own combinedMeshes instead of the library's buffer."""
        if array is None:
            self._check(self.lib.gv_pool_set_record_target(self.ctx, pool_id, view_index, None, 0))
            self._keep.pop(("target", pool_id, view_index), None)  # (only now: the call above synchronises and un-registers)
            return
        assert array.flags["C_CONTIGUOUS"] and array.flags["WRITEABLE"]
        self._check(self.lib.gv_pool_set_record_target(self.ctx, pool_id, view_index, array.ctypes.data, array.nbytes))
        self._keep[("target", pool_id, view_index)] = array  # page-locked and written by the device until replaced / removed / close()

    def instance_bases(self, pool_id=0, view_index=0):
        """First instance index of every fetched record ([count + 1] words; the last one is instance_count)."""
        ptr, n = C.POINTER(C.c_uint32)(), C.c_uint32()
        self._check(self.lib.gv_pool_results_instance_bases(self.ctx, pool_id, view_index, C.byref(ptr), C.byref(n)))
        return np.ctypeslib.as_array(ptr, shape=(n.value + 1,)).copy()

    def results_device(self, view_index=0):
        d = GvDeviceResult()
        self._check(self.lib.gv_results_device(self.ctx, view_index, C.byref(d)))
        return d

    def set_index_map(self, pool_id, global_ids):
        """Pool slot -> global id table for the exchange shards of `pool_id` (None / empty removes it)."""
        if global_ids is None or len(global_ids) == 0:
            self._check(self.lib.gv_pool_set_index_map(self.ctx, pool_id, None, 0))
            return
        g = np.ascontiguousarray(global_ids, dtype=np.uint32)
        self._check(self.lib.gv_pool_set_index_map(self.ctx, pool_id, g.ctypes.data, g.shape[0]))

    def copy_idx_device(self, view_index, dst_ptr, capacity, index_base=0):
        self._check(self.lib.gv_results_copy_idx_device(self.ctx, view_index, dst_ptr, capacity, index_base))

    def copy_shard_device(self, view_index, dst_ptr, capacity, index_base=0):
        """dst[0] = draw_count, dst[1:1+min(count, capacity)] = visible_idx + index_base (device memory, no sync)."""
        self._check(self.lib.gv_results_copy_shard_device(self.ctx, view_index, dst_ptr, capacity, index_base))

    def copy_mask_device(self, view_index, dst_ptr, word_count):
        """Exchange shard as a bit per MIRROR entry: [draw_count, ceil(occupancy / 32) words] into device memory at dst_ptr
        (entry e is pool slot mirror_slots()[e])."""
        self._check(self.lib.gv_results_copy_mask_device(self.ctx, view_index, dst_ptr, word_count))

    def mirror_epoch(self, pool_id=0):
        """Changes whenever mirror_slots() does (rebuild, appended slots, re-order after entity churn)."""
        e = C.c_uint64()
        self._check(self.lib.gv_pool_mirror_epoch(self.ctx, pool_id, C.byref(e)))
        return int(e.value)

    def mirror_slots(self, pool_id, occupancy):
        """entry -> pool slot table of the pool's device mirror (changes only when the mirror is rebuilt)."""
        out = np.empty(occupancy, dtype=np.uint32)
        self._check(self.lib.gv_pool_mirror_slots(self.ctx, pool_id, out.ctypes.data, occupancy))
        return out

    # ---- native RCCL exchange (what a C++ engine calls; bench.py --gpus N times it: config.exchange_path "c-abi") ----
    @staticmethod
    def exchange_unique_id():
        buf = C.create_string_buffer(128)
        rc = load().gv_exchange_unique_id(buf)
        if rc != GV_OK:
            raise GvError(rc, "gv_exchange_unique_id failed (RCCL not loadable?)")
        return buf.raw

    def exchange_init(self, unique_id, rank, world_size):
        self._check(self.lib.gv_exchange_init(self.ctx, unique_id, rank, world_size))

    def exchange_shards(self, view_index, capacity, index_base, gathered_ptr, capacities=None, world=None):
        """Caller-owned rows [world, 1 + capacity]; capacities (one per rank, the same list on every rank): what of each
        rank's row travels under the direct patterns. (gv_exchange_shards reads one capacity per rank of the communicator: pass
        `world` to have the list's length checked here.)"""
        caps = None
        if capacities is not None:
            if world is not None and len(capacities) != world:
                raise ValueError(f"exchange_shards: {len(capacities)} capacities for {world} ranks")
            caps = (C.c_uint32 * max(len(capacities), GV_EXCHANGE_MAX_RANKS))(*[int(c) for c in capacities])
        self._check(self.lib.gv_exchange_shards(self.ctx, view_index, capacity, caps, index_base, gathered_ptr))

    @staticmethod
    def _exchange_frame(f):
        w = f.world_size
        return dict(ptr=f.gathered_device, row_words=f.row_words, world=w, frame=int(f.frame), complete=bool(f.complete),
                    room=[int(f.room[r]) for r in range(w)], travelled_words=[int(f.travelled_words[r]) for r in range(w)],
                    counts=[int(f.counts[r]) for r in range(w)], tail_words=[int(f.tail_words[r]) for r in range(w)],
                    cut_ranks=[r for r in range(w) if (f.cut_ranks >> r) & 1], mode=int(f.mode), ready_event=f.ready_event,
                    # gv_exchange_views: lists per rank in the frame (0: a single-list frame) and, once acquired, item_counts[r][i]
                    items=int(f.items), item_counts=([[int(f.item_counts[r * f.items + i]) for i in range(f.items)] for r in range(w)]
                                                     if f.items and f.item_counts else None))

    def exchange_visible(self, view_index=0, index_base=0, pool_id=None):
        """Sends the frame (gv_exchange_visible; pool_id: gv_pool_exchange_visible for a named pool): library-owned rows, sized from
        the previous frame's headers. Returns a dict: frame, row_words, world, room, travelled_words, mode — the rows themselves are
        handed out by exchange_acquire."""
        f = GvExchangeFrame()
        if pool_id is None:
            self._check(self.lib.gv_exchange_visible(self.ctx, view_index, index_base, 0, C.byref(f)))
        else:
            self._check(self.lib.gv_pool_exchange_visible(self.ctx, pool_id, view_index, index_base, 0, C.byref(f)))
        return self._exchange_frame(f)

    def exchange_views(self, items):
        """ONE exchange for ALL the lists of a frame (gv_exchange_views): items = [(pool_id, view_index, index_base), ...], the same
        list on every rank. Row r of the acquired frame = [len(items) + total, c_0 .. c_k, list 0, list 1 ...]."""
        arr = (GvExchangeItem * len(items))(*[GvExchangeItem(int(p), int(v), int(b)) for p, v, b in items])
        f = GvExchangeFrame()
        self._check(self.lib.gv_exchange_views(self.ctx, arr, len(items), 0, C.byref(f)))
        return self._exchange_frame(f)

    def update_index_map(self, pool_id, first, global_ids):
        ids = np.ascontiguousarray(global_ids, dtype=np.uint32)
        self._check(self.lib.gv_pool_update_index_map(self.ctx, pool_id, int(first), ids.ctypes.data_as(C.POINTER(C.c_uint32)), ids.shape[0]))

    def set_result_mapping(self, pool_id, flags, visible=None, stride=1):
        """gv_pool_set_result_mapping: records in world slots (GV_RESULTS_MAP_RECORDS) and / or the write-back of isVisible through the
        index map into `visible` (a uint8 array the caller keeps alive; element i at visible + i * stride)."""
        base = visible.ctypes.data if visible is not None else None
        count = (visible.nbytes // stride) if visible is not None else 0
        self._check(self.lib.gv_pool_set_result_mapping(self.ctx, pool_id, int(flags), base, int(stride), int(count)))

    def exchange_acquire(self, frame):
        """Frame `frame` complete — every rank's WHOLE list, short predictions made good by a second exchange — and the context's
        stream ordered behind it (gv_exchange_acquire). Returns the frame's dict: ptr (device address of the uint32 rows
        [world, row_words]), counts, cut_ranks / tail_words (statistics), ready_event."""
        f = GvExchangeFrame()
        self._check(self.lib.gv_exchange_acquire(self.ctx, frame, C.byref(f)))
        return self._exchange_frame(f)

    def exchange_set_timeout(self, milliseconds):
        self._check(self.lib.gv_exchange_set_timeout(self.ctx, milliseconds))

    def exchange_masks(self, view_index, word_count, gathered_ptr):
        """All ranks' [draw_count, one bit per mirror entry] shards into gathered_ptr ([world, 1 + word_count] uint32, device)."""
        self._check(self.lib.gv_exchange_masks(self.ctx, view_index, word_count, gathered_ptr))

    def exchange_set_mode(self, mode):
        """0 all-gather, 1 grouped send/recv with every peer, 2 one broadcast per root (GvExchangeMode)."""
        self._check(self.lib.gv_exchange_set_mode(self.ctx, mode))

    def exchange_shutdown(self):
        self._check(self.lib.gv_exchange_shutdown(self.ctx))

    def stream(self):
        """The context's hipStream_t as an integer (e.g. for torch.cuda.ExternalStream)."""
        return self.lib.gv_stream(self.ctx)

    def sort(self, view_index=0, descending=False, pool_id=None):
        if pool_id is None:
            self._check(self.lib.gv_sort(self.ctx, view_index, 1 if descending else 0))
        else:
            self._check(self.lib.gv_pool_sort(self.ctx, pool_id, view_index, 1 if descending else 0))

    # ---- world matrices ----
    def sweep(self, mode=GV_SWEEP_VALU):
        self._check(self.lib.gv_sweep(self.ctx, mode))

    def get_world(self, first, count):
        out = np.empty((count, 12), dtype=np.float32)
        self._check(self.lib.gv_get_world(self.ctx, first, count, out.ctypes.data))
        return out

    # ---- Hi-Z ----
    def hiz_build(self, depth):
        d = np.ascontiguousarray(depth, dtype=np.float32)
        self._check(self.lib.gv_hiz_build(self.ctx, d.ctypes.data, d.shape[1], d.shape[0], GV_MEM_HOST))

    def hiz_rebuild(self):
        self._check(self.lib.gv_hiz_rebuild(self.ctx))

    def hiz_mip_count(self):
        n = C.c_uint32()
        self._check(self.lib.gv_hiz_mip_count(self.ctx, C.byref(n)))
        return n.value

    def hiz_read_level(self, level, w, h):
        out = np.empty((h, w, 2), dtype=np.float32)
        rw, rh = C.c_uint32(), C.c_uint32()
        self._check(self.lib.gv_hiz_read_level(self.ctx, level, out.ctypes.data, C.byref(rw), C.byref(rh)))
        assert (rw.value, rh.value) == (w, h), (rw.value, rh.value, w, h)
        return out

    # ---- metrics ----
    def stats(self):
        s = GvStats()
        self._check(self.lib.gv_stats(self.ctx, C.byref(s)))
        return dict(launches={k: int(s.launches[i]) for i, k in enumerate(KERNEL_NAMES)},
                    device_ms={k: float(s.device_ms[i]) for i, k in enumerate(KERNEL_NAMES)},
                    upload_bytes=int(s.upload_bytes), mirror_reorders=int(s.mirror_reorders), record_targets_lost=int(s.record_targets_lost),
                    max_depth=int(s.max_depth),
                    transform_count=int(s.transform_count), bounds_blocks_total=int(s.bounds_blocks_total),
                    bounds_blocks_examined=int(s.bounds_blocks_examined))

    def stream_peak(self, pool_id=0, launches=20):
        """GB/s of a read-only pass over the cull kernel's input streams of `pool_id` (median of `launches`)."""
        g = C.c_double()
        self._check(self.lib.gv_debug_stream_peak(self.ctx, pool_id, launches, C.byref(g)))
        return g.value

    def stats_reset(self):
        self._check(self.lib.gv_stats_reset(self.ctx))

    def profile_sampling(self, every):
        """Bracket only every `every`-th launch of each kernel kind with events (each bracket costs ~5 us of stream time)."""
        self._check(self.lib.gv_profile_sampling(self.ctx, every))

    def profile_kernels(self, names):
        """Bracket the launches of these kernel kinds (KERNEL_NAMES) from now on; [] = none."""
        self._check(self.lib.gv_profile_kernels(self.ctx, sum(1 << KERNEL_NAMES.index(n) for n in names)))

    def profile_samples(self):
        """Bracketed launches per kernel kind since stats_reset: the divisor for stats()['device_ms']."""
        a = (C.c_uint64 * GV_K_COUNT)()
        self._check(self.lib.gv_profile_samples(self.ctx, C.byref(a)))
        return {k: int(a[i]) for i, k in enumerate(KERNEL_NAMES)}


def cell_owner(grid, side, world, positions):
    """Rank that owns each position (float32 [n, >= 3]) under the dealing rule of gv_scene_extract_rank (gv_cell_owner)."""
    pos = np.ascontiguousarray(positions, dtype=np.float32)
    out = np.empty(pos.shape[0], dtype=np.uint32)
    g = (C.c_uint32 * 3)(*[int(x) for x in grid])
    rc = load().gv_cell_owner(C.byref(g), float(side), int(world), pos.ctypes.data, pos.strides[0], pos.shape[0], out.ctypes.data)
    if rc != GV_OK:
        raise GvError(rc, "gv_cell_owner: bad argument")
    return out


class Scene:
    """A Garden scene file ingested straight into column pools (gv_scene_*; no device needed until bind)."""

    def __init__(self, text, pools, add_root_entity=False, bson=False):
        """text: the scene JSON (str or bytes), or with bson=True the BSON document packed builds ship (json2bson);
        pools: {component ".type": pool id}; add_root_entity: loadScene's addRootEntity (a default transform as
        entity 1, parent of everything that names no other parent)."""
        self.lib = load()
        raw = text.encode() if isinstance(text, str) else bytes(text)
        arr = (GvScenePool * max(len(pools), 1))(*[GvScenePool(k.encode(), v) for k, v in pools.items()])
        handle, err = C.c_void_p(), C.create_string_buffer(512)
        parse = self.lib.gv_scene_parse_bson if bson else self.lib.gv_scene_parse_json
        rc = parse(raw, len(raw), arr, len(pools), 1 if add_root_entity else 0, C.byref(handle), err, len(err))
        if rc != 0:
            raise GvError(rc, err.value.decode(errors="replace"))
        self.handle, self.pools = handle, dict(pools)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.gv_scene_destroy(self.handle)
            self.handle = None

    __del__ = close

    def info(self):
        i = GvSceneInfo()
        self.lib.gv_scene_info(self.handle, C.byref(i))
        out = {n: getattr(i, n) for n, _ in GvSceneInfo._fields_ if n != "mesh_count"}
        out["mesh_count"] = {pid: i.mesh_count[pid] for pid in self.pools.values()}
        return out

    @staticmethod
    def _view(col, n, dtype, width):
        if n == 0:
            return np.zeros((0, width) if width > 1 else 0, dtype)
        a = np.ctypeslib.as_array(C.cast(col.data, C.POINTER(np.ctypeslib.as_ctypes_type(dtype))), shape=(n * width,))
        return a.reshape(n, width).copy() if width > 1 else a.copy()

    def transform_columns(self):
        """Copies of the transform columns + entity_to_transform + uids, as numpy arrays."""
        c, n, cap, e2t, uids = GvTransformColumns(), C.c_uint32(), C.c_uint32(), C.c_void_p(), C.c_void_p()
        self.lib.gv_scene_transform_columns(self.handle, C.byref(c), C.byref(n), C.byref(e2t), C.byref(cap), C.byref(uids))
        n = n.value
        out = dict(entity=self._view(c.entity, n, np.uint32, 1), parent=self._view(c.parent, n, np.uint32, 1),
                   position=self._view(c.position, n, np.float32, 3), scale=self._view(c.scale, n, np.float32, 3),
                   rotation=self._view(c.rotation, n, np.float32, 4), self_active=self._view(c.self_active, n, np.uint8, 1),
                   ancestors_active=self._view(c.ancestors_active, n, np.uint8, 1),
                   model_with_ancestors=self._view(c.model_with_ancestors, n, np.uint8, 1))
        out["entity_to_transform"] = self._view(GvColumn(e2t.value, 4), cap.value, np.uint32, 1)
        out["uid"] = self._view(GvColumn(uids.value, 8), n, np.uint64, 1)
        return out

    def mesh_columns(self, pool_id):
        c, n = GvMeshColumns(), C.c_uint32()
        self.lib.gv_scene_mesh_columns(self.handle, pool_id, C.byref(c), C.byref(n))
        n = n.value
        return dict(entity=self._view(c.entity, n, np.uint32, 1), is_enabled=self._view(c.is_enabled, n, np.uint8, 1),
                    aabb_min=self._view(c.aabb_min, n, np.float32, 3), aabb_max=self._view(c.aabb_max, n, np.float32, 3),
                    is_visible=self._view(GvColumn(c.is_visible, 1), n, np.uint8, 1))

    def extract_tile(self, grid, side, tile):
        """One spatial tile of this scene as a Scene of its own (gv_scene_extract_tile)."""
        handle = C.c_void_p()
        g = (C.c_uint32 * 3)(*[int(x) for x in grid])
        rc = self.lib.gv_scene_extract_tile(self.handle, C.byref(g), float(side), int(tile), C.byref(handle))
        if rc != 0:
            raise GvError(rc, "gv_scene_extract_tile failed")
        t = Scene.__new__(Scene)
        t.lib, t.handle, t.pools = self.lib, handle, dict(self.pools)
        return t

    def extract_rank(self, grid, side, rank, world):
        """Everything rank `rank` of `world` owns (the grid's cells in Morton order, dealt round-robin) as one Scene
        (gv_scene_extract_rank; multi.py::cell_owners is the same table)."""
        handle = C.c_void_p()
        g = (C.c_uint32 * 3)(*[int(x) for x in grid])
        rc = self.lib.gv_scene_extract_rank(self.handle, C.byref(g), float(side), int(rank), int(world), C.byref(handle))
        if rc != 0:
            raise GvError(rc, "gv_scene_extract_rank failed")
        t = Scene.__new__(Scene)
        t.lib, t.handle, t.pools = self.lib, handle, dict(self.pools)
        return t

    def tile_maps(self, pool_id):
        """(transform_global, mesh_global) of a tile: local slot -> slot in the scene it was cut from."""
        tg, tn, mg, mn = C.c_void_p(), C.c_uint32(), C.c_void_p(), C.c_uint32()
        rc = self.lib.gv_scene_tile_maps(self.handle, pool_id, C.byref(tg), C.byref(tn), C.byref(mg), C.byref(mn))
        if rc != 0:
            raise GvError(rc, "gv_scene_tile_maps: not a tile")
        return (self._view(GvColumn(tg.value, 4), tn.value, np.uint32, 1), self._view(GvColumn(mg.value, 4), mn.value, np.uint32, 1))

    def bind(self, vis):
        """Binds the scene's columns to a GpuVisibility context (the scene must stay alive while bound)."""
        vis._scene = self
        vis._check(self.lib.gv_scene_bind(vis.ctx, self.handle))
