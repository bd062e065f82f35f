"""Byte layouts of the reference's component pools as numpy structured dtypes.

MeshRenderComponent  include/garden/system/render/mesh.hpp:45-55   (48 B: Component{entity} + 3 reserved
                     + isEnabled + isVisible pad the header to 16 B, then Aabb = 2 x f32x4)
TransformComponent   include/garden/system/transform.hpp:31-61     (80 B release layout)
"""
import numpy as np

GV_NONE = 0xFFFFFFFF

MESH_DTYPE = np.dtype({
    "names": ["entity", "reserved0", "reserved1", "reserved2", "isEnabled", "isVisible", "aabbMin", "aabbMax"],
    "formats": ["<u4", "<u4", "<u4", "<u2", "u1", "u1", ("<f4", 4), ("<f4", 4)],
    "offsets": [0, 4, 8, 12, 14, 15, 16, 32],
    "itemsize": 48,
})

TRANSFORM_DTYPE = np.dtype({
    "names": ["entity", "parent", "uid", "position", "scale", "rotation", "childs",
              "selfActive", "ancestorsActive", "modelWithAncestors"],
    "formats": ["<u4", "<u4", "<u8", ("<f4", 4), ("<f4", 4), ("<f4", 4), "<u8", "u1", "u1", "u1"],
    "offsets": [0, 4, 8, 16, 32, 48, 64, 72, 73, 74],
    "itemsize": 80,
})


def mesh_layout_offsets(dtype=MESH_DTYPE):
    f = dtype.fields
    return dict(entity=f["entity"][1], is_enabled=f["isEnabled"][1], is_visible=f["isVisible"][1],
                aabb_min=f["aabbMin"][1], aabb_max=f["aabbMax"][1])


def transform_layout_offsets(dtype=TRANSFORM_DTYPE):
    f = dtype.fields
    return dict(entity=f["entity"][1], parent=f["parent"][1], position=f["position"][1], scale=f["scale"][1],
                rotation=f["rotation"][1], self_active=f["selfActive"][1],
                ancestors_active=f["ancestorsActive"][1], model_with_ancestors=f["modelWithAncestors"][1])


def derived_mesh_dtype(extra_bytes):
    """A MeshRenderComponent-derived struct (e.g. SpriteRenderComponent, sprite.hpp:29-43): same header,
    larger stride — the reason the reference walks the pool by getMeshComponentSize() (mesh.cpp:119,139)."""
    d = dict(names=list(MESH_DTYPE.names), formats=[MESH_DTYPE.fields[n][0] for n in MESH_DTYPE.names],
             offsets=[MESH_DTYPE.fields[n][1] for n in MESH_DTYPE.names], itemsize=48 + int(extra_bytes))
    return np.dtype(d)
