"""Synthetic scenes for tests and bench.py.

Positions uniform in a cube of side 100*N^(1/3); random unit quaternions (normalised in fp32);
scale in [0.5, 2]^3; local AABB = [-h, +h], h in [0.25, 1]^3; 1 % each of disabled meshes,
zero-size AABBs, selfActive=false transforms and free slots; camera at the origin with a random
orientation, perspective FOV 90 deg, 16:9, near 0.01, infinite reversed-Z
(include/garden/system/camera.hpp:31-39,111-121). PRNG: numpy PCG64 seeded with 0x6A7D3E11.
"""
import math

import numpy as np

from .pools import GV_NONE, MESH_DTYPE, TRANSFORM_DTYPE

SEED = 0x6A7D3E11
F32 = np.float32


def _unit_quats(rng, n):
    q = rng.standard_normal((n, 4), dtype=np.float32)
    q[np.all(q == 0, axis=1)] = (0, 0, 0, 1)
    inv = (F32(1.0) / np.sqrt(np.sum(q * q, axis=1, dtype=np.float32), dtype=np.float32))
    return (q * inv[:, None]).astype(np.float32)


def quat_to_mat3(q):
    x, y, z, w = [F32(v) for v in q]
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]], dtype=np.float32)


def persp_inf_rev_z(fov_y, aspect, near):
    """calcPerspProjInfRevZ (camera.hpp:115-116): depth = near / z_view, 1 at the near plane, 0 at infinity.
    Column-major float[16]."""
    f = F32(1.0 / math.tan(fov_y * 0.5))
    m = np.zeros((4, 4), dtype=np.float32)  # m[c][r]
    m[0][0] = f / F32(aspect)
    m[1][1] = -f
    m[2][3] = 1.0
    m[3][2] = near
    return m.reshape(16)


def ortho_rev_z(width, height, near, far):
    """calcOrthoProjRevZ (camera.hpp:119-120): depth 1 at near, 0 at far. Column-major float[16]."""
    m = np.zeros((4, 4), dtype=np.float32)
    m[0][0] = F32(2.0 / width)
    m[1][1] = F32(-2.0 / height)
    m[2][2] = F32(-1.0 / (far - near))
    m[3][2] = F32(far / (far - near))
    m[3][3] = 1.0
    return m.reshape(16)


def mul_cm(a, b):
    """Column-major 4x4 product a*b in fp32 (host-side camera maths only; not the canonical kernel order)."""
    A = np.asarray(a, dtype=np.float32).reshape(4, 4).T
    B = np.asarray(b, dtype=np.float32).reshape(4, 4).T
    return (A @ B).astype(np.float32).T.reshape(16).copy()


def view_from_quat(q):
    """View matrix of a camera with orientation q, translation zeroed (graphics.cpp:201): rotation^T."""
    r = quat_to_mat3(q)
    m = np.zeros((4, 4), dtype=np.float32)  # m[c][r]
    m[:3, :3] = r  # column-major storage of r^T == row-major r
    m[3][3] = 1.0
    return m.reshape(16)


def make_view(view_proj, camera_position=(0, 0, 0), camera_offset=(0, 0, 0), shadow_pass=-1, use_hiz=0,
              distance_2d=0, emit_records=1):
    return dict(view_proj=np.asarray(view_proj, dtype=np.float32).reshape(16).copy(),
                camera_position=np.asarray(list(camera_position) + [0], dtype=np.float32)[:4].copy(),
                camera_offset=np.asarray(list(camera_offset) + [0], dtype=np.float32)[:4].copy(),
                shadow_pass=int(shadow_pass), use_hiz=int(use_hiz), distance_2d=int(distance_2d),
                emit_records=int(emit_records))


def main_camera_view(seed=SEED, use_hiz=0, camera_position=(0.0, 0.0, 0.0)):
    rng = np.random.Generator(np.random.PCG64(seed ^ 0xC0FFEE))
    q = _unit_quats(rng, 1)[0]
    proj = persp_inf_rev_z(math.radians(90.0), 16.0 / 9.0, 0.01)
    return make_view(mul_cm(proj, view_from_quat(q)), camera_position=camera_position, use_hiz=use_hiz)


def cascade_view(seed=SEED, size=4000.0, depth=20000.0, index=0):
    """One orthographic shadow-cascade-like view (csm.cpp:260-343 produces viewProj + cameraOffset)."""
    rng = np.random.Generator(np.random.PCG64(seed ^ (0x5AD0 + index)))
    q = _unit_quats(rng, 1)[0]
    proj = ortho_rev_z(size, size, -depth * 0.5, depth * 0.5)
    off = rng.uniform(-10, 10, 3).astype(np.float32)
    return make_view(mul_cm(proj, view_from_quat(q)), camera_offset=off, shadow_pass=index)


class Scene:
    """AoS pools in the reference's byte layouts + the entity -> transform-slot map."""

    def __init__(self, meshes, transforms, entity_to_transform):
        self.meshes = meshes
        self.transforms = transforms
        self.entity_to_transform = entity_to_transform

    @property
    def count(self):
        return int(self.meshes.shape[0])


def _fill_common(rng, n, meshes, transforms, side, defects=True):
    transforms["position"][:, :3] = rng.uniform(-0.5 * side, 0.5 * side, (n, 3)).astype(np.float32)
    transforms["scale"][:, :3] = rng.uniform(0.5, 2.0, (n, 3)).astype(np.float32)
    transforms["rotation"] = _unit_quats(rng, n)
    transforms["selfActive"] = 1
    transforms["ancestorsActive"] = 1
    transforms["modelWithAncestors"] = 1
    h = rng.uniform(0.25, 1.0, (n, 3)).astype(np.float32)
    meshes["aabbMin"][:, :3] = -h
    meshes["aabbMax"][:, :3] = h
    meshes["isEnabled"] = 1
    meshes["isVisible"] = 0
    ids = np.arange(1, n + 1, dtype=np.uint32)
    meshes["entity"] = ids
    transforms["entity"] = ids
    transforms["uid"] = ids.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
    if defects and n >= 100:
        r = rng.random(n)
        meshes["isEnabled"][r < 0.01] = 0
        zs = (r >= 0.01) & (r < 0.02)
        meshes["aabbMax"][zs] = meshes["aabbMin"][zs]
        transforms["selfActive"][(r >= 0.02) & (r < 0.03)] = 0
        free = (r >= 0.03) & (r < 0.04)
        meshes["entity"][free] = 0
        transforms["entity"][free] = 0


def flat_scene(n, seed=SEED, defects=True, stride_extra=0):
    """cfg1/cfg2/cfg3: flat hierarchy, one transform and one mesh per entity, same slot order."""
    from .pools import derived_mesh_dtype
    rng = np.random.Generator(np.random.PCG64(seed))
    mesh_dtype = MESH_DTYPE if not stride_extra else derived_mesh_dtype(stride_extra)
    meshes = np.zeros(n, dtype=mesh_dtype)
    transforms = np.zeros(n, dtype=TRANSFORM_DTYPE)
    side = 100.0 * n ** (1.0 / 3.0)
    _fill_common(rng, n, meshes, transforms, side, defects)
    e2t = np.full(n + 1, GV_NONE, dtype=np.uint32)
    live = transforms["entity"] != 0
    e2t[transforms["entity"][live]] = np.nonzero(live)[0].astype(np.uint32)
    return Scene(meshes, transforms, e2t)


def hierarchy_scene(n, depth=4, fanout=10, seed=SEED, defects=True):
    """cfg4: forest of `depth` levels, level l+1 has `fanout` children per node of level l until the last
    level takes the remainder (10^7: 10^4 roots -> 10^5 -> 10^6 -> 8.89*10^6 leaves). Level-ordered
    slots, children of one parent contiguous; every level carries meshes."""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x4EE))
    denom = sum(fanout ** l for l in range(depth))
    roots = max(1, n // denom) if depth > 1 else n
    counts, total = [], 0
    for l in range(depth):
        c = roots * fanout ** l if l < depth - 1 else n - total
        c = max(0, min(c, n - total))
        counts.append(c)
        total += c
    meshes = np.zeros(n, dtype=MESH_DTYPE)
    transforms = np.zeros(n, dtype=TRANSFORM_DTYPE)
    side = 100.0 * n ** (1.0 / 3.0)
    _fill_common(rng, n, meshes, transforms, side, defects)
    start = 0
    prev_start, prev_count = 0, 0
    for l, c in enumerate(counts):
        if l > 0 and c > 0:
            sl = slice(start, start + c)
            parent_slot = prev_start + (np.arange(c, dtype=np.int64) * prev_count // c)
            parent_entity = transforms["entity"][parent_slot]
            transforms["parent"][sl] = parent_entity  # 0 when the parent slot is free: chain ends there
            transforms["position"][sl, :3] = rng.uniform(-40.0, 40.0, (c, 3)).astype(np.float32)
            transforms["scale"][sl, :3] = rng.uniform(0.7, 1.3, (c, 3)).astype(np.float32)
            pa = transforms["selfActive"][parent_slot] & transforms["ancestorsActive"][parent_slot]
            transforms["ancestorsActive"][sl] = np.where(parent_entity != 0, pa, 1)
        prev_start, prev_count = start, c
        start += c
    e2t = np.full(n + 1, GV_NONE, dtype=np.uint32)
    live = transforms["entity"] != 0
    e2t[transforms["entity"][live]] = np.nonzero(live)[0].astype(np.uint32)
    return Scene(meshes, transforms, e2t)


def synthetic_depth(width, height, seed=SEED, rects=256):
    """cfg3 depth image: background 0.0 (far, reversed-Z) + `rects` screen-space walls with depth in
    [0.02, 0.5]; nearer walls overwrite farther ones (max, since larger = nearer)."""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0xD3F7))
    d = np.zeros((height, width), dtype=np.float32)
    for _ in range(rects):
        w = int(rng.integers(max(1, width // 64), max(2, width // 6)))
        h = int(rng.integers(max(1, height // 64), max(2, height // 6)))
        x = int(rng.integers(0, max(1, width - w)))
        y = int(rng.integers(0, max(1, height - h)))
        z = np.float32(rng.uniform(0.02, 0.5))
        np.maximum(d[y:y + h, x:x + w], z, out=d[y:y + h, x:x + w])
    return d


def noise_depth(width, height, seed=SEED, block=8, near=0.01, nearest=50.0, farthest=20000.0):
    """A HARD depth image for the occlusion query: every `block` x `block` pixel block holds an occluder at
    its own distance, log-uniform in [nearest, farthest] metres — reversed-Z infinite projection: depth = near / distance — i.e.
    right among the entities of the 10 M world (cube of 21.5 km), not centimetres from the camera like synthetic_depth's walls.
    A coarse pyramid texel then spans many occluder distances, its (min, max) straddles nearly every box's zNear, the early
    accept / reject of hiz_occluded (gv_device.hpp) decides almost nothing and the queries go down to levels 0-2."""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x9E15E))
    bw, bh = (width + block - 1) // block, (height + block - 1) // block
    dist = np.exp(rng.uniform(np.log(nearest), np.log(farthest), size=(bh, bw)))
    d = (np.float32(near) / dist.astype(np.float32)).astype(np.float32)
    return np.ascontiguousarray(np.repeat(np.repeat(d, block, axis=0), block, axis=1)[:height, :width])


def shuffled_scene(sc, fraction=1.0, seed=SEED, drop_transforms=0.0):
    """Same entities, but the transform pool is permuted relative to the mesh pool (ECS pools are independent:
    an entity's mesh slot and transform slot need not match) for `fraction` of the slots; `drop_transforms`
    removes that share of transforms altogether (mesh without a TransformComponent: mesh.cpp:149-155)."""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x5F1E))
    n = sc.count
    perm = np.arange(n)
    k = int(n * fraction)
    if k > 1:
        chosen = rng.choice(n, size=k, replace=False)
        perm[chosen] = chosen[rng.permutation(k)]
    transforms = sc.transforms[perm].copy()  # new transform slot j holds the old slot perm[j]
    if drop_transforms > 0:
        gone = rng.random(n) < drop_transforms
        has_children = np.isin(transforms["entity"], sc.transforms["parent"][sc.transforms["parent"] != 0])
        gone &= ~has_children
        transforms["entity"][gone] = 0
    e2t = np.full(n + 1, GV_NONE, dtype=np.uint32)
    live = transforms["entity"] != 0
    e2t[transforms["entity"][live]] = np.nonzero(live)[0].astype(np.uint32)
    return Scene(sc.meshes.copy(), transforms, e2t)
