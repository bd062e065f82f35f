"""GPU parity: libgarden_vis (HIP, through the C-ABI) vs the CPU oracle on the same seeded AoS pools.

Bar: bit-exact visible-index set, isVisible bytes, bakedModel and distanceSq (the kernels and the oracle
use the same written operation order, so the matrices agree to the bit, not just to 1e-5)."""
import numpy as np
import pytest

from garden_amd import scene

pytestmark = pytest.mark.gpu


def run_both(gpu, oracle, sc, views, hiz_depth=None):
    gpu.bind_transforms(sc.transforms, sc.entity_to_transform)
    gpu.bind_pool(0, sc.meshes)
    gpu.hierarchy_rebuild()
    hz = None
    if hiz_depth is not None:
        gpu.hiz_build(hiz_depth)
        hz = oracle.Hiz(hiz_depth)
    gpu.cull(0, views)
    out = []
    for vi, v in enumerate(views):
        sc.meshes["isVisible"] = 7  # poison: the main pass must overwrite every slot
        got = gpu.fetch(vi, write_back=True, occupancy=sc.count)
        got_vis = sc.meshes["isVisible"].copy()
        sc.meshes["isVisible"] = 7
        exp = oracle.prepare_meshes(sc.meshes, sc.transforms, sc.entity_to_transform, v, hiz=hz if v.get("use_hiz") else None)
        exp_vis = sc.meshes["isVisible"].copy()
        out.append((got, got_vis, exp, exp_vis))
    return out


def assert_same(got, got_vis, exp, exp_vis, main_pass=True):
    assert got["draw_count"] == exp["draw_count"]
    # the oracle's single-thread order is ascending slot order; the library's is too (stable compaction)
    assert np.array_equal(got["visible_idx"], exp["visible_idx"])
    assert np.array_equal(got["baked_model"].view(np.uint32), exp["baked_model"].view(np.uint32))
    assert np.array_equal(got["distance_sq"].view(np.uint32), exp["distance_sq"].view(np.uint32))
    if main_pass:
        assert np.array_equal(got_vis, exp_vis)
        assert got["is_visible"] is not None and np.array_equal(got["is_visible"], exp_vis)
    else:
        assert np.all(got_vis == 7) and np.all(exp_vis == 7)  # shadow passes leave isVisible alone (mesh.cpp:144)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 10_000, 100_003])
def test_flat_frustum_parity(gpu, oracle, n):
    sc = scene.flat_scene(n, seed=scene.SEED + n)
    (got, gv, exp, ev), = run_both(gpu, oracle, sc, [scene.main_camera_view()])
    assert_same(got, gv, exp, ev)


def test_hierarchy_parity(gpu, oracle):
    sc = scene.hierarchy_scene(50_000, depth=4, fanout=10)
    (got, gv, exp, ev), = run_both(gpu, oracle, sc, [scene.main_camera_view()])
    assert exp["draw_count"] > 0
    assert_same(got, gv, exp, ev)
    assert gpu.stats()["max_depth"] == 3


def test_multi_view_and_shadow_pass(gpu, oracle):
    sc = scene.flat_scene(20_000)
    views = [scene.main_camera_view(), scene.cascade_view(index=0), scene.cascade_view(index=1)]
    res = run_both(gpu, oracle, sc, views)
    assert_same(*res[0], main_pass=True)
    assert_same(*res[1], main_pass=False)
    assert_same(*res[2], main_pass=False)


def test_hiz_occlusion_parity(gpu, oracle):
    sc = scene.flat_scene(50_000)
    depth = scene.synthetic_depth(512, 256)
    v = scene.main_camera_view(use_hiz=1)
    (got, gv, exp, ev), = run_both(gpu, oracle, sc, [v], hiz_depth=depth)
    frustum_only = oracle.prepare_meshes(sc.meshes, sc.transforms, sc.entity_to_transform, scene.main_camera_view())
    assert 0 < exp["draw_count"] < frustum_only["draw_count"]  # the query does cull something
    assert_same(got, gv, exp, ev)


@pytest.mark.parametrize("size", [(64, 64), (5, 3), (7, 7), (135, 77), (1920, 1080), (4096, 4096), (1024, 512),
                                  # sizes not divisible by 64 take the any-size fused kernel (three levels per launch, rims recomputed):
                                  # odd at every level / thin / two such launches in a row (the second reads pairs) / partial edge tiles
                                  (257, 131), (515, 389), (1283, 719), (1000, 37), (2560, 1440), (3840, 2160), (2049, 1025),
                                  (20000, 3), (3, 20000), (16400, 2), (1600, 900)])  # one-texel-high / -wide levels inside the fused kernel
@pytest.mark.parametrize("rule", [0, 1])
@pytest.mark.parametrize("rg16f", [False, True])
def test_hiz_pyramid_parity(oracle, size, rule, rg16f):
    """rg16f: GV_CONFIG_HIZ_RG16F — the pyramid in the reference's image format (hiz.hpp:41), min rounded toward -inf and
    max toward +inf; gv_hiz_read_level returns the stored halfs widened, equal to the oracle's values bit for bit."""
    from garden_amd.lib import GpuVisibility
    w, h = size
    depth = scene.synthetic_depth(w, h, rects=37)
    rng = np.random.default_rng(w * 131 + h)
    depth = np.maximum(depth, (rng.random((h, w)) * 0.01).astype(np.float32))
    # texels the comparisons treat specially: +0 / -0 (equal, different bits: which one a min / max keeps depends on the
    # operand order), NaN (every comparison false: a NaN is kept or dropped depending on which side it arrives) and
    # infinities — the reduction order of hiz.frag:29-60 is part of the contract, so these pin it
    k = max(1, (w * h) // 23)
    flat = depth.reshape(-1)
    for value in (0.0, -0.0, np.nan, np.inf, -np.inf, 3e-6, -3e-6, 1e-9, 7e4, -7e4):  # incl. half subnormals, underflow, overflow
        flat[rng.choice(flat.size, k, replace=False)] = np.float32(value)
    with GpuVisibility(device=0, hiz_rule=rule, hiz_rg16f=rg16f) as vis:
        vis.hiz_build(depth)
        exp = oracle.Hiz(depth, rule=rule, rg16f=rg16f)
        assert vis.hiz_mip_count() == exp.mip_count
        for k in range(1, exp.mip_count):
            e = exp.level(k)
            g = vis.hiz_read_level(k, e.shape[1], e.shape[0])
            assert np.array_equal(g.view(np.uint32), e.view(np.uint32)), f"mip {k} differs"


@pytest.mark.parametrize("mode", [0, 1])
def test_world_matrices(gpu, oracle, mode):
    sc = scene.hierarchy_scene(30_000, depth=5, fanout=6)
    gpu.bind_transforms(sc.transforms, sc.entity_to_transform)
    gpu.hierarchy_rebuild()
    gpu.sweep(mode)
    got = gpu.get_world(0, sc.count)
    exp = oracle.world_matrices(sc.transforms, sc.entity_to_transform)
    assert np.max(np.abs(got - exp)) <= 1e-5  # the north-star tolerance
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))  # and in fact bit-exact


def test_keep_slot_order_gives_ascending_records(gpu_slot_order, oracle):
    """GV_CONFIG_KEEP_SLOT_ORDER: the mirror stays in pool order and the order-stable compaction returns records
    in ascending slot order with no host-side sort; an ortho view that sees most of the cube (many full chunks)."""
    gpu = gpu_slot_order
    sc = scene.flat_scene(200_000, seed=5)
    v = scene.cascade_view(size=20000.0, depth=40000.0)
    gpu.bind_transforms(sc.transforms, sc.entity_to_transform)
    gpu.bind_pool(0, sc.meshes)
    gpu.hierarchy_rebuild()
    gpu.cull(0, [v])
    got = gpu.fetch(0, write_back=False, occupancy=sc.count, order="raw")
    exp = oracle.prepare_meshes(sc.meshes, sc.transforms, sc.entity_to_transform, v)
    assert got["draw_count"] == exp["draw_count"] > 100_000
    assert np.array_equal(got["visible_idx"], exp["visible_idx"])
    assert np.array_equal(got["baked_model"].view(np.uint32), exp["baked_model"].view(np.uint32))
    assert np.array_equal(got["distance_sq"].view(np.uint32), exp["distance_sq"].view(np.uint32))


def test_spatial_mirror_order_is_deterministic_and_complete(gpu, oracle):
    """Default: mirror entries are Morton-ordered by root position; records come out in that order (the same
    permutation every time), every pool slot exactly once, and isVisible / gv_get_world are indexed by pool slot."""
    sc = scene.shuffled_scene(scene.hierarchy_scene(80_000, depth=4, fanout=9), fraction=0.2)
    v = scene.cascade_view(size=9000.0, depth=40000.0)
    gpu.bind_transforms(sc.transforms, sc.entity_to_transform)
    gpu.bind_pool(0, sc.meshes)
    gpu.hierarchy_rebuild()
    raws = []
    for _ in range(2):
        gpu.cull(0, [v])
        raws.append(gpu.fetch(0, write_back=False, occupancy=sc.count, order="raw"))
    exp = oracle.prepare_meshes(sc.meshes.copy(), sc.transforms, sc.entity_to_transform, v)
    assert np.array_equal(raws[0]["visible_idx"], raws[1]["visible_idx"])            # reproducible
    assert not np.array_equal(raws[0]["visible_idx"], exp["visible_idx"])            # and not pool order
    assert np.array_equal(np.sort(raws[0]["visible_idx"]), exp["visible_idx"])       # same set, each slot once
    gpu.sweep(0)
    assert np.array_equal(gpu.get_world(1000, 5000).view(np.uint32),
                          oracle.world_matrices(sc.transforms, sc.entity_to_transform, 1000, 5000).view(np.uint32))


@pytest.mark.parametrize("hier", [False, True])
def test_slot_order_context_full_parity(gpu_slot_order, oracle, hier):
    sc = scene.hierarchy_scene(50_000) if hier else scene.flat_scene(50_000, seed=8)
    depth = scene.synthetic_depth(256, 128)
    res = run_both(gpu_slot_order, oracle, sc, [scene.main_camera_view(use_hiz=1), scene.cascade_view()], hiz_depth=depth)
    assert_same(*res[0], main_pass=True)
    assert_same(*res[1], main_pass=False)


def test_count_only_view(gpu, oracle):
    """emit_records = 0: isVisible + drawCount only (editor statistics path, mesh.cpp:540-544)."""
    sc = scene.flat_scene(30_000, seed=9)
    v = dict(scene.main_camera_view(), emit_records=0)
    gpu.bind_transforms(sc.transforms, sc.entity_to_transform)
    gpu.bind_pool(0, sc.meshes)
    gpu.hierarchy_rebuild()
    gpu.cull(0, [v])
    got = gpu.fetch(0, write_back=False, occupancy=sc.count)
    exp = oracle.prepare_meshes(sc.meshes, sc.transforms, sc.entity_to_transform, v)
    assert got["draw_count"] == exp["draw_count"] == gpu.result_count(0)
    assert np.array_equal(got["is_visible"], sc.meshes["isVisible"])


def test_batched_views_equal_separate_passes(gpu, oracle):
    """Main camera (+ Hi-Z) and 4 shadow cascades in one pass over the streams == each view on its own; and views
    with different cameraPosition (not batchable) still go through the per-view path."""
    sc = scene.hierarchy_scene(60_000, depth=4, fanout=10)
    depth = scene.synthetic_depth(512, 256)
    hz = oracle.Hiz(depth)
    views = [scene.main_camera_view(use_hiz=1)] + [scene.cascade_view(index=k, size=3000.0 + 500 * k) for k in range(4)]
    gpu.bind_transforms(sc.transforms, sc.entity_to_transform)
    gpu.bind_pool(0, sc.meshes)
    gpu.hierarchy_rebuild()
    gpu.hiz_build(depth)
    gpu.stats_reset()
    gpu.cull(0, views)
    assert gpu.stats()["launches"]["cull"] == 1  # one batched launch for the 5 views
    for vi, v in enumerate(views):
        sc.meshes["isVisible"] = 7
        got = gpu.fetch(vi, write_back=True, occupancy=sc.count)
        got_vis = sc.meshes["isVisible"].copy()
        sc.meshes["isVisible"] = 7
        exp = oracle.prepare_meshes(sc.meshes, sc.transforms, sc.entity_to_transform, v, hiz=hz if v["use_hiz"] else None)
        assert_same(got, got_vis, exp, sc.meshes["isVisible"].copy(), main_pass=v["shadow_pass"] < 0)
    moved = [scene.main_camera_view(), scene.main_camera_view(camera_position=(100.0, 5.0, -40.0))]
    gpu.stats_reset()
    gpu.cull(0, moved)
    assert gpu.stats()["launches"]["cull"] == 2  # different cameraPosition: corners differ, one pass per view
    for vi, v in enumerate(moved):
        got = gpu.fetch(vi, write_back=False, occupancy=sc.count)
        exp = oracle.prepare_meshes(sc.meshes, sc.transforms, sc.entity_to_transform, v)
        assert np.array_equal(got["visible_idx"], exp["visible_idx"])
        assert np.array_equal(got["baked_model"].view(np.uint32), exp["baked_model"].view(np.uint32))


@pytest.mark.parametrize("n,descending,d2", [(100_000, False, 0), (100_000, True, 0), (300_000, False, 0), (5_000, True, 1), (70, False, 0),
                                            (4096, False, 0), (4000, True, 0), (513, False, 1), (3, True, 0), (40, False, 0),
                                            # occupancy <= 16384: one-launch LDS sort, network sized by the count on the device
                                            (16_384, False, 0), (16_384, True, 1), (12_000, True, 0), (16_385, False, 0), (9_000, False, 0),
                                            # radix sort with the short tiles (up to 524288 records, chosen on the device) and just beyond
                                            (13_500, False, 0), (21_000, True, 0), (50_000, False, 1), (68_000, True, 0), (540_000, False, 0),
                                            (560_000, True, 0)])
def test_gpu_sort_matches_sort_meshes(gpu, oracle, n, descending, d2):
    """gv_sort == sortMeshes (mesh.cpp:265-328): ascending distanceSq for unsorted buffers, descending for the
    sorted ones; the oracle breaks ties by slot, and so does the stable radix sort."""
    sc = scene.flat_scene(n, seed=3 + n)
    wide = n >= 13_500 or n == 12_000  # nearly everything visible: large record counts
    v = scene.cascade_view(size=30000.0, depth=60000.0) if wide else scene.main_camera_view()
    v = dict(v, distance_2d=d2)
    gpu.bind_transforms(sc.transforms, sc.entity_to_transform)
    gpu.bind_pool(0, sc.meshes)
    gpu.hierarchy_rebuild()
    gpu.cull(0, [v])
    gpu.sort(0, descending=descending)
    got = gpu.fetch(0, write_back=False, occupancy=sc.count, order="raw")
    exp = oracle.prepare_meshes(sc.meshes, sc.transforms, sc.entity_to_transform, v, sort="descending" if descending else "ascending")
    assert got["draw_count"] == exp["draw_count"] > 0
    d = got["distance_sq"]
    assert np.all(d[:-1] >= d[1:]) if descending else np.all(d[:-1] <= d[1:])
    # equal keys: the stable sort keeps the emitted (mirror) order, the oracle breaks ties by slot, std::sort in the
    # reference leaves them unspecified -> canonicalise ties by slot before comparing
    o = np.lexsort((got["visible_idx"], -d if descending else d))
    got = {k: (v[o] if isinstance(v, np.ndarray) and v.shape[:1] == d.shape else v) for k, v in got.items()}
    assert np.array_equal(got["visible_idx"], exp["visible_idx"])
    assert np.array_equal(got["baked_model"].view(np.uint32), exp["baked_model"].view(np.uint32))
    assert np.array_equal(got["distance_sq"].view(np.uint32), exp["distance_sq"].view(np.uint32))


@pytest.mark.parametrize("kind", ["spread", "few_values", "two_clusters", "all_equal", "one_outlier"])
def test_large_sorts_when_the_keys_bunch_up(gpu_slot_order, oracle, kind):
    """Long lists whose keys bunch up — a handful of distinct distances, two tight clusters, all equal, one far outlier that
    stretches the range (round 3; written for a value-bucket sort that was measured and withdrawn, kept for the radix passes) —
    come out in sortMeshes' order (mesh.cpp:265-328) with ties in emission order, which in a slot-order mirror is the oracle's
    tie order: compared element for element."""
    gpu = gpu_slot_order
    n = 600_000
    sc = scene.flat_scene(n, seed=21, defects=False)
    rng = np.random.Generator(np.random.PCG64(4))
    z = sc.transforms["position"][:, 2]
    if kind == "few_values":
        z[:] = rng.integers(0, 5, n).astype(np.float32) * np.float32(250.0)
    elif kind == "two_clusters":
        z[:] = np.where(rng.random(n) < 0.5, 0.0, 9000.0).astype(np.float32) + rng.uniform(-1e-3, 1e-3, n).astype(np.float32)
    elif kind == "all_equal":
        z[:] = np.float32(42.0)
    elif kind == "one_outlier":
        z[:] = rng.uniform(0, 1000, n).astype(np.float32)
        z[12345] = np.float32(3e8)
    view = dict(scene.cascade_view(size=1e6, depth=2e9), distance_2d=1, shadow_pass=-1)
    view["view_proj"] = [2e-6, 0, 0, 0, 0, 2e-6, 0, 0, 0, 0, 5e-10, 0, 0, 0, 0.5, 1]  # an axis-aligned box around everything
    view["camera_offset"] = [0, 0, 0, 0]
    gpu.bind_transforms(sc.transforms, sc.entity_to_transform)
    gpu.bind_pool(0, sc.meshes)
    gpu.hierarchy_rebuild()
    for descending in (False, True):
        gpu.cull(0, [view])
        gpu.sort(0, descending=descending)
        got = gpu.fetch(0, write_back=False, occupancy=n, order="raw")
        exp = oracle.prepare_meshes(sc.meshes.copy(), sc.transforms, sc.entity_to_transform, view, sort="descending" if descending else "ascending")
        assert got["draw_count"] == exp["draw_count"] > 0.9 * n
        assert np.array_equal(got["distance_sq"].view(np.uint32), exp["distance_sq"].view(np.uint32))
        assert np.array_equal(got["visible_idx"], exp["visible_idx"])  # ties: ascending slot on both sides
        assert np.array_equal(got["baked_model"].view(np.uint32), exp["baked_model"].view(np.uint32))


@pytest.mark.parametrize("n", [40_000, 65_000, 150_000])  # (150 000: beyond the pools the hint applies to)
def test_sort_of_a_mid_sized_pool_follows_the_previous_count(gpu, oracle, n):
    """A pool between the one-launch batch and 65 536 slots is sorted by the rank sort ALONE when the view's previous fetch
    saw a short list (no idle radix launches); a list that outgrows the rank sort's key table within one frame is still
    sorted by that launch (keys from memory), and the frame after it takes the radix passes again."""
    sc = scene.flat_scene(n, seed=5 + n)
    gpu.bind_transforms(sc.transforms, sc.entity_to_transform)
    gpu.bind_pool(0, sc.meshes)
    gpu.hierarchy_rebuild()
    small = scene.cascade_view(size=700.0 if n == 40_000 else (800.0 if n == 65_000 else 1100.0), depth=60000.0)   # a few thousand records
    wide = scene.cascade_view(size=30000.0, depth=60000.0)                             # nearly everything
    counts = []
    for v, descending in [(small, False), (small, True), (wide, False), (wide, True), (small, False), (scene.main_camera_view(), True)]:
        gpu.cull(0, [v])
        gpu.sort(0, descending=descending)
        got = gpu.fetch(0, write_back=False, occupancy=sc.count, order="raw")
        exp = oracle.prepare_meshes(sc.meshes, sc.transforms, sc.entity_to_transform, v, sort="descending" if descending else "ascending")
        assert got["draw_count"] == exp["draw_count"] > 0
        d = got["distance_sq"]
        assert np.all(d[:-1] >= d[1:]) if descending else np.all(d[:-1] <= d[1:])
        o = np.lexsort((got["visible_idx"], -d if descending else d))  # ties: canonical by slot (see test_gpu_sort_matches_sort_meshes)
        assert np.array_equal(got["visible_idx"][o], exp["visible_idx"])
        assert np.array_equal(got["baked_model"][o].view(np.uint32), exp["baked_model"].view(np.uint32))
        assert np.array_equal(got["distance_sq"][o].view(np.uint32), exp["distance_sq"].view(np.uint32))
        counts.append(int(got["draw_count"]))
    assert counts[0] <= 10240 < 16384 < counts[2], counts  # short list -> rank sort alone; then a list beyond its key table


@pytest.mark.parametrize("fraction,hier", [(1.0, False), (1.0, True), (0.05, False), (0.3, True)])
def test_mesh_and_transform_pools_in_different_order(gpu, oracle, fraction, hier):
    """ECS pools are independent: mesh slot i need not map to transform slot i. fraction = 1 exercises the
    non-speculating kernels, small fractions the mis-speculation reload inside the speculating ones; 2 % of the
    meshes have no TransformComponent at all (mesh.cpp:149-155)."""
    base = scene.hierarchy_scene(40_000, depth=4, fanout=7) if hier else scene.flat_scene(40_000, seed=21)
    sc = scene.shuffled_scene(base, fraction=fraction, drop_transforms=0.02)
    depth = scene.synthetic_depth(256, 256)
    views = [scene.main_camera_view(use_hiz=1), scene.cascade_view(index=1)]
    res = run_both(gpu, oracle, sc, views, hiz_depth=depth)
    assert res[0][2]["draw_count"] > 0
    assert_same(*res[0], main_pass=True)
    assert_same(*res[1], main_pass=False)
    gpu.sweep(1)
    world = gpu.get_world(0, sc.count)
    assert np.array_equal(world.view(np.uint32), oracle.world_matrices(sc.transforms, sc.entity_to_transform).view(np.uint32))


def test_dirty_ranges_and_rebinding(gpu, oracle):
    """gv_mark_dirty: only the marked transform / mesh ranges are re-gathered; unmarked edits must NOT show up
    (the mirror is authoritative until told otherwise), a hierarchy mark rebuilds everything."""
    sc = scene.hierarchy_scene(20_000, depth=3, fanout=8)
    v = scene.main_camera_view()
    from garden_amd.lib import GV_DIRTY_HIERARCHY, GV_DIRTY_MESH, GV_DIRTY_TRANSFORM
    gpu.bind_transforms(sc.transforms, sc.entity_to_transform)
    gpu.bind_pool(0, sc.meshes)
    gpu.hierarchy_rebuild()

    def cull_and_compare(expect_equal=True):
        gpu.cull(0, [v])
        got = gpu.fetch(0, write_back=False, occupancy=sc.count)
        exp = oracle.prepare_meshes(sc.meshes.copy(), sc.transforms, sc.entity_to_transform, v)
        same = got["draw_count"] == exp["draw_count"] and np.array_equal(got["visible_idx"], exp["visible_idx"]) and \
            np.array_equal(got["baked_model"].view(np.uint32), exp["baked_model"].view(np.uint32))
        assert same == expect_equal
    cull_and_compare()
    rng = np.random.default_rng(4)
    sc.transforms["position"][5000:9000, :3] += rng.uniform(-500, 500, (4000, 3)).astype(np.float32)
    cull_and_compare(expect_equal=False)          # edited but not marked: the mirror still holds the old TRS
    gpu.mark_dirty(GV_DIRTY_TRANSFORM, 5000, 4000)
    cull_and_compare()
    sc.meshes["aabbMax"][100:300, :3] *= 3
    sc.meshes["isEnabled"][300:350] = 0
    gpu.mark_dirty(GV_DIRTY_MESH, 100, 250, pool_id=0)
    cull_and_compare()
    sc.transforms["parent"][15000:15100] = sc.transforms["entity"][10:110]  # re-parent (setParent, transform.cpp:130-195)
    gpu.mark_dirty(GV_DIRTY_HIERARCHY, 0, 0)          # count == 0: rebuild (and re-order) everything
    cull_and_compare()
    assert gpu.stats()["max_depth"] >= 2
    # ranged hierarchy mark: only these slots' links are re-gathered, the mirror order is kept, depth re-validated
    sc.transforms["parent"][16000:16050] = sc.transforms["entity"][15000:15050]   # hang them under the re-parented ones
    gpu.mark_dirty(GV_DIRTY_HIERARCHY, 16000, 50)
    cull_and_compare()
    assert gpu.stats()["max_depth"] >= 2
    from garden_amd.lib import GV_E_ARG, GvError
    sc.transforms["parent"][10] = sc.transforms["entity"][16000]                  # closes a cycle 10 -> 16000 -> 15000 -> 10
    gpu.mark_dirty(GV_DIRTY_HIERARCHY, 10, 1)
    with pytest.raises(GvError) as e:
        gpu.cull(0, [v])
    assert e.value.code == GV_E_ARG and "cycle" in str(e.value)
    sc.transforms["parent"][10] = 0                                              # caller fixes it: next sync rebuilds
    cull_and_compare()


def test_empty_and_tiny_pools_and_derived_stride(gpu, oracle):
    from garden_amd.pools import MESH_DTYPE, TRANSFORM_DTYPE
    v = scene.main_camera_view()
    empty = scene.Scene(np.zeros(0, MESH_DTYPE), np.zeros(0, TRANSFORM_DTYPE), np.full(1, 0xFFFFFFFF, np.uint32))
    gpu.bind_transforms(empty.transforms, empty.entity_to_transform)
    gpu.bind_pool(0, empty.meshes)
    gpu.hierarchy_rebuild()
    gpu.cull(0, [v])
    assert gpu.fetch(0, occupancy=0)["draw_count"] == 0
    wide = scene.flat_scene(5000, seed=77, stride_extra=32)  # SpriteRenderComponent-like derived struct, 80-byte stride
    assert wide.meshes.dtype.itemsize == 80
    (got, gv, exp, ev), = run_both(gpu, oracle, wide, [v])
    assert_same(got, gv, exp, ev)


def test_a_rank_that_owns_nothing_still_takes_part_in_the_exchange():
    """More ranks than cells (or an empty region of the world): a rank with EMPTY pools culls nothing and still makes its
    gv_exchange_visible calls — header 0, no entries — frame after frame, in every travel pattern; a pool that then gets entities
    outgrows the room its empty past predicted: the frame is completed by a second exchange before it is handed out, the next
    ones are sized from its headers."""
    import torch
    from garden_amd.lib import GpuVisibility
    from garden_amd.pools import MESH_DTYPE, TRANSFORM_DTYPE
    v = scene.main_camera_view()
    empty = scene.Scene(np.zeros(0, MESH_DTYPE), np.zeros(0, TRANSFORM_DTYPE), np.full(1, 0xFFFFFFFF, np.uint32))
    full = scene.flat_scene(30_000, seed=3)

    class _Span:
        pass

    with GpuVisibility(device=0) as vis:
        vis.exchange_init(GpuVisibility.exchange_unique_id(), 0, 1)
        vis.bind_transforms(empty.transforms, empty.entity_to_transform)
        vis.bind_pool(0, empty.meshes)
        vis.hierarchy_rebuild()
        for frame in range(4):
            vis.exchange_set_mode(frame % 3)
            vis.cull(0, [v])
            f = vis.exchange_acquire(vis.exchange_visible(0, index_base=5)["frame"])
            assert f["complete"] and f["counts"] == [0] and not f["cut_ranks"] and f["row_words"] >= 4 and f["row_words"] % 4 == 0
        vis.bind_transforms(full.transforms, full.entity_to_transform)
        vis.bind_pool(0, full.meshes)
        vis.hierarchy_rebuild()
        completed = 0
        for frame in range(4):
            vis.cull(0, [v])
            sent = vis.exchange_visible(0, index_base=5)
            f = vis.exchange_acquire(sent["frame"])
            got = vis.fetch(0, write_back=False, occupancy=full.count)
            assert f["complete"] and f["counts"] == [got["draw_count"]] and got["draw_count"] > 1024
            completed += bool(f["cut_ranks"])
            # whatever room the prediction gave the row, the acquired frame holds the WHOLE list
            span = _Span()
            span.__cuda_array_interface__ = {"shape": (f["row_words"],), "typestr": "<i4", "data": (int(f["ptr"]), False), "version": 2}
            vis.wait()
            row = torch.as_tensor(span, device="cuda:0").cpu().numpy().view(np.uint32)
            assert row[0] == got["draw_count"] and np.array_equal(np.sort(row[1:1 + row[0]]), got["visible_idx"] + 5)
            assert f["tail_words"] == [max(0, got["draw_count"] - sent["room"][0])]
        # the first frame with entities outgrew the empty pool's room and was completed inside the frame; the next ones were predicted
        assert completed == 1 and not f["cut_ranks"]
        vis.exchange_shutdown()


def test_two_contexts_alive_in_one_process_share_no_state(oracle):
    """One process, several GPUs = several contexts in one address space (the drop-in's multi-GPU mode: gpu_visibility_system.hpp).
    Two REAL contexts on the same device, different worlds, different configurations, their calls interleaved — binds, pyramid
    builds, culls, sorts, sweeps, fetches, a rebind with another size — each against the oracle after every step: nothing one context
    does may show in the other (no static buffer, no shared stream, no cached launch state)."""
    from garden_amd.lib import GpuVisibility
    a_scene = scene.hierarchy_scene(150_000, depth=4, fanout=8, seed=5)
    b_scene = scene.flat_scene(400_000, seed=6)
    depth = scene.synthetic_depth(512, 256)
    va, vb = scene.main_camera_view(seed=11, use_hiz=1), scene.main_camera_view(seed=12)

    def check(vis, sc, view, hiz=None):
        got = vis.fetch(0, write_back=False, occupancy=sc.count)
        exp = oracle.prepare_meshes(sc.meshes.copy(), sc.transforms, sc.entity_to_transform, view, hiz=hiz)
        assert got["draw_count"] == exp["draw_count"] > 0
        order = np.argsort(got["visible_idx"], kind="stable")
        assert np.array_equal(got["visible_idx"][order], exp["visible_idx"])
        assert np.array_equal(got["baked_model"][order].view(np.uint32), exp["baked_model"].view(np.uint32))

    with GpuVisibility(device=0, block_bounds=True) as a, GpuVisibility(device=0, keep_slot_order=True) as b:
        assert a.stream() != b.stream()
        a.bind_transforms(a_scene.transforms, a_scene.entity_to_transform)
        b.bind_transforms(b_scene.transforms, b_scene.entity_to_transform)
        a.bind_pool(0, a_scene.meshes)
        a.hiz_build(depth)
        b.bind_pool(0, b_scene.meshes)
        hz = oracle.Hiz(depth)
        for frame in range(3):
            a.cull(0, [va])
            b.cull(0, [vb])       # (between a's cull and a's fetch)
            a.sort(0)
            check(b, b_scene, vb)
            b.sweep(1)
            check(a, a_scene, va, hiz=hz)
            world_b = b.get_world(0, b_scene.count)
            a.hiz_rebuild()
        assert np.array_equal(world_b.view(np.uint32), oracle.world_matrices(b_scene.transforms, b_scene.entity_to_transform).view(np.uint32))
        # b is re-bound to a smaller world while a keeps going
        c_scene = scene.flat_scene(90_000, seed=7)
        b.bind_transforms(c_scene.transforms, c_scene.entity_to_transform)
        b.bind_pool(0, c_scene.meshes)
        b.hierarchy_rebuild()
        a.cull(0, [va])
        b.cull(0, [vb])
        check(a, a_scene, va, hiz=hz)
        check(b, c_scene, vb)


def test_hierarchy_cycle_is_rejected(gpu):
    from garden_amd.lib import GV_E_ARG, GvError
    sc = scene.hierarchy_scene(100, depth=3, fanout=3, defects=False)
    sc.transforms["parent"][0] = sc.transforms["entity"][50]  # root now hangs under its own descendant
    chain = []
    s = 50
    while sc.transforms["parent"][s]:
        s = int(sc.entity_to_transform[sc.transforms["parent"][s]])
        chain.append(s)
        if len(chain) > 200:
            break
    gpu.bind_transforms(sc.transforms, sc.entity_to_transform)
    gpu.bind_pool(0, sc.meshes)
    if 0 in chain:  # only a real cycle must be rejected (reference asserts, transform.cpp:137-143)
        with pytest.raises(GvError) as e:
            gpu.hierarchy_rebuild()
        assert e.value.code == GV_E_ARG and "cycle" in str(e.value)


def test_several_mesh_pools_share_the_transform_pool(gpu, oracle):
    """Several IMeshRenderSystems (mesh.cpp:69-108) each own a pool; all resolve entities through the one
    TransformComponent pool. Pool 0: plain 48-byte components for every entity; pool 1: an 80-byte derived
    component for every third entity, in a different slot order; pool 2: empty."""
    from garden_amd.pools import MESH_DTYPE, derived_mesh_dtype
    sc = scene.hierarchy_scene(30_000, depth=3, fanout=9)
    rng = np.random.default_rng(12)
    ents = sc.transforms["entity"][sc.transforms["entity"] != 0]
    chosen = rng.permutation(ents[::3])
    pool1 = np.zeros(chosen.shape[0] + 50, dtype=derived_mesh_dtype(32))
    pool1["entity"][:chosen.shape[0]] = chosen          # the last 50 slots stay free (entity 0)
    pool1["isEnabled"] = 1
    h = rng.uniform(0.5, 3.0, (pool1.shape[0], 3)).astype(np.float32)
    pool1["aabbMin"][:, :3] = -h
    pool1["aabbMax"][:, :3] = h
    pool2 = np.zeros(0, dtype=MESH_DTYPE)
    pools = [sc.meshes, pool1, pool2]
    views = [scene.main_camera_view(), scene.cascade_view(index=3)]
    gpu.bind_transforms(sc.transforms, sc.entity_to_transform)
    for pid, pool in enumerate(pools):
        gpu.bind_pool(pid, pool)
    gpu.hierarchy_rebuild()
    for pid, pool in enumerate(pools):
        gpu.cull(pid, views)
        if pool.shape[0] == 0:
            assert all(gpu.fetch(vi, occupancy=0)["draw_count"] == 0 for vi in range(len(views)))
            continue
        for vi, v in enumerate(views):
            pool["isVisible"] = 7
            got = gpu.fetch(vi, write_back=True, occupancy=pool.shape[0])
            got_vis = pool["isVisible"].copy()
            pool["isVisible"] = 7
            exp = oracle.prepare_meshes(pool, sc.transforms, sc.entity_to_transform, v)
            assert_same(got, got_vis, exp, pool["isVisible"].copy(), main_pass=v["shadow_pass"] < 0)
        assert got["draw_count"] > 0


def same_bits_or_both_nan(a, b):
    """Bit-identical, except that a NaN may carry a different sign/payload (x86 makes 0xFFC00000, gfx950 0x7FC00000)."""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return a.shape == b.shape and bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def test_non_finite_and_denormal_inputs_agree_with_the_oracle(gpu, oracle):
    """Garbage in, the SAME garbage out: NaN / Inf / denormal / -0 / huge values in TRS and AABBs must lead to the
    same visibility decision and the same record bits as the oracle (NaN-propagating plane test, minNum/maxNum
    reductions in the Hi-Z query, no flush-to-zero)."""
    sc = scene.hierarchy_scene(20_000, depth=3, fanout=6)
    rng = np.random.default_rng(99)
    t, m = sc.transforms, sc.meshes
    specials = np.array([np.nan, np.inf, -np.inf, 1e-42, -1e-42, -0.0, 3e38, -3e38, 1e-30], dtype=np.float32)
    for field, width in (("position", 3), ("scale", 3), ("rotation", 4)):
        rows = rng.choice(sc.count, 400, replace=False)
        t[field][rows, rng.integers(0, width, 400)] = specials[rng.integers(0, len(specials), 400)]
    for field in ("aabbMin", "aabbMax"):
        rows = rng.choice(sc.count, 300, replace=False)
        m[field][rows, rng.integers(0, 3, 300)] = specials[rng.integers(0, len(specials), 300)]
    t["scale"][rng.choice(sc.count, 100, replace=False), :3] = 0.0          # degenerate models
    t["position"][rng.choice(sc.count, 100, replace=False), :3] = 0.0       # entities exactly at the camera
    depth = scene.synthetic_depth(256, 256)
    views = [scene.main_camera_view(use_hiz=1), scene.cascade_view(index=0)]
    res = run_both(gpu, oracle, sc, views, hiz_depth=depth)
    for (got, gv, exp, ev), main in zip(res, (True, False)):
        assert got["draw_count"] == exp["draw_count"] and np.array_equal(got["visible_idx"], exp["visible_idx"])
        assert same_bits_or_both_nan(got["baked_model"], exp["baked_model"])
        assert same_bits_or_both_nan(got["distance_sq"], exp["distance_sq"])
        if main:
            assert np.array_equal(gv, ev)
    for mode in (0, 1):
        gpu.sweep(mode)
        assert same_bits_or_both_nan(gpu.get_world(0, sc.count), oracle.world_matrices(sc.transforms, sc.entity_to_transform)), f"sweep mode {mode}"


@pytest.mark.parametrize("size,rule", [((301, 171), 1), ((640, 360), 1), ((1000, 1000), 1), ((301, 171), 0), ((1024, 512), 0)])
@pytest.mark.parametrize("rg16f", [False, True])
def test_hiz_query_early_accept_is_exact(oracle, size, rule, rg16f):
    """The coarse-level early-accept must never change a decision: NPOT pyramids under the conservative rule
    (nested, incl. the clamped last texels), NPOT under the reference rule (not nested: shortcut off), and an
    all-even pyramid under the reference rule (nested). rg16f: the same queries against the RG16F pyramid (incl. the
    virtual level 1 of the 1024x512 case: its texels are rounded like stored ones); outward rounding can only keep more."""
    from garden_amd.lib import GpuVisibility
    w, h = size
    sc = scene.flat_scene(120_000, seed=w + h + rule)
    depth = scene.synthetic_depth(w, h, rects=90)
    v = scene.main_camera_view(use_hiz=1)
    with GpuVisibility(device=0, hiz_rule=rule, hiz_rg16f=rg16f) as vis:
        vis.bind_transforms(sc.transforms, sc.entity_to_transform)
        vis.bind_pool(0, sc.meshes)
        vis.hiz_build(depth)
        vis.cull(0, [v])
        got = vis.fetch(0, write_back=False, occupancy=sc.count)
    exp = oracle.prepare_meshes(sc.meshes.copy(), sc.transforms, sc.entity_to_transform, v, hiz=oracle.Hiz(depth, rule=rule, rg16f=rg16f))
    frustum_only = oracle.prepare_meshes(sc.meshes.copy(), sc.transforms, sc.entity_to_transform, dict(v, use_hiz=0))
    assert 0 < exp["draw_count"] < frustum_only["draw_count"]
    assert np.array_equal(got["visible_idx"], exp["visible_idx"])
    if rg16f:  # conservative w.r.t. the fp32 pyramid: everything that one keeps is kept
        fp32 = oracle.prepare_meshes(sc.meshes.copy(), sc.transforms, sc.entity_to_transform, v, hiz=oracle.Hiz(depth, rule=rule))
        assert np.isin(fp32["visible_idx"], exp["visible_idx"]).all() and exp["draw_count"] >= fp32["draw_count"]


def test_device_side_result_accessors(gpu, oracle):
    """gv_results_device / gv_results_copy_idx_device / gv_results_copy_shard_device: the compact list as device
    consumers (the multi-GPU exchange, an indirect-draw builder) see it, incl. the capacity clamp."""
    import torch
    sc = scene.flat_scene(50_000)
    view = scene.main_camera_view()
    bind_and_cull = lambda: (gpu.bind_transforms(sc.transforms, sc.entity_to_transform), gpu.bind_pool(0, sc.meshes),
                             gpu.hierarchy_rebuild(), gpu.cull(0, [view]))
    bind_and_cull()
    raw = gpu.fetch(0, write_back=False, occupancy=sc.count, order="raw")
    k = raw["draw_count"]
    assert k > 100
    base = 7_000_000
    dst = torch.full((sc.count,), -1, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()  # (the fill runs on torch's stream; the library's stream is non-blocking: no implicit order between them)
    gpu.copy_idx_device(0, dst.data_ptr(), sc.count, index_base=base)
    assert gpu.result_count(0) == k  # readback on the library's stream: fences the copy
    got = dst.cpu().numpy()
    assert np.array_equal(got[:k].astype(np.int64), raw["visible_idx"].astype(np.int64) + base) and np.all(got[k:] == -1)

    shard = torch.full((1 + sc.count,), -1, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()  # (the fill runs on torch's stream; the library's stream is non-blocking: no implicit order between them)
    gpu.copy_shard_device(0, shard.data_ptr(), sc.count, index_base=base)
    gpu.wait()
    got = shard.cpu().numpy()
    assert got[0] == k and np.array_equal(got[1:1 + k].astype(np.int64), raw["visible_idx"].astype(np.int64) + base)
    assert np.all(got[1 + k:] == -1)

    cap = 64  # too small: header still carries the true count, body is clamped
    small = torch.full((1 + cap + 8,), -1, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()  # (the fill runs on torch's stream; the library's stream is non-blocking: no implicit order between them)
    gpu.copy_shard_device(0, small.data_ptr(), cap, index_base=0)
    gpu.wait()
    got = small.cpu().numpy()
    assert got[0] == k and np.array_equal(got[1:1 + cap], raw["visible_idx"][:cap].astype(np.int32)) and np.all(got[1 + cap:] == -1)

    d = gpu.results_device(0)
    assert d.visible_idx and d.baked_model and d.distance_sq and d.draw_count


@pytest.mark.parametrize("fused_mode", [2, 3])  # GV_SWEEP_WITH_CULL (MFMA chain), GV_SWEEP_WITH_CULL_VALU
@pytest.mark.parametrize("hier,hiz", [(True, False), (True, True), (False, False)])
def test_sweep_with_cull_is_the_same_as_sweep_then_cull(gpu, oracle, hier, hiz, fused_mode):
    """GV_SWEEP_WITH_CULL: one pass produces the world matrices and the cull outputs of an exactly paired pool;
    bit-identical to gv_sweep(MFMA) + gv_cull and to the oracle. Unpaired pools and batched views fall back to the
    two launches with the same results."""
    from garden_amd.lib import GV_SWEEP_MFMA
    GV_SWEEP_WITH_CULL = fused_mode
    sc = scene.hierarchy_scene(70_001, depth=5, fanout=5) if hier else scene.flat_scene(70_001)
    depth = scene.synthetic_depth(512, 256) if hiz else None
    view = scene.main_camera_view(use_hiz=1 if hiz else 0)
    gpu.bind_transforms(sc.transforms, sc.entity_to_transform)
    gpu.bind_pool(0, sc.meshes)
    gpu.hierarchy_rebuild()
    if hiz:
        gpu.hiz_build(depth)
    gpu.sweep(GV_SWEEP_MFMA)
    world_ref = gpu.get_world(0, sc.count)
    gpu.cull(0, [view])
    ref = gpu.fetch(0, write_back=False, occupancy=sc.count)

    gpu.mark_dirty(0, 0, sc.count)  # GV_DIRTY_TRANSFORM: invalidates the world cache
    gpu.sweep(GV_SWEEP_WITH_CULL)
    gpu.cull(0, [view])
    got = gpu.fetch(0, write_back=False, occupancy=sc.count)
    world = gpu.get_world(0, sc.count)
    assert np.array_equal(world.view(np.uint32), world_ref.view(np.uint32))
    for k in ("visible_idx", "baked_model", "distance_sq", "is_visible"):
        assert np.array_equal(got[k].view(np.uint8), ref[k].view(np.uint8)), k
    exp_w = oracle.world_matrices(sc.transforms, sc.entity_to_transform)
    assert np.array_equal(world.view(np.uint32), exp_w.view(np.uint32))
    m2 = sc.meshes.copy()
    exp = oracle.prepare_meshes(m2, sc.transforms, sc.entity_to_transform, view, hiz=oracle.Hiz(depth) if hiz else None)
    assert np.array_equal(got["visible_idx"], exp["visible_idx"]) and np.array_equal(got["is_visible"], m2["isVisible"])

    # fallbacks: two views sharing the camera (batched cull) and a shuffled (not exactly paired) pool
    views = [view, scene.cascade_view(index=0)]
    gpu.mark_dirty(0, 0, sc.count)
    gpu.sweep(GV_SWEEP_WITH_CULL)
    gpu.cull(0, views)
    assert np.array_equal(gpu.get_world(0, sc.count).view(np.uint32), world_ref.view(np.uint32))
    assert np.array_equal(gpu.fetch(0, write_back=False, occupancy=sc.count)["visible_idx"], ref["visible_idx"])
    sh = scene.shuffled_scene(sc, fraction=1.0)
    gpu.bind_transforms(sh.transforms, sh.entity_to_transform)
    gpu.bind_pool(0, sh.meshes)
    gpu.hierarchy_rebuild()
    gpu.sweep(GV_SWEEP_WITH_CULL)
    gpu.cull(0, [view])
    got = gpu.fetch(0, write_back=False, occupancy=sh.count)
    m2 = sh.meshes.copy()
    exp = oracle.prepare_meshes(m2, sh.transforms, sh.entity_to_transform, view, hiz=oracle.Hiz(depth) if hiz else None)
    assert np.array_equal(got["visible_idx"], np.sort(exp["visible_idx"])) and np.array_equal(got["is_visible"], m2["isVisible"])
    assert np.array_equal(gpu.get_world(0, sh.count).view(np.uint32),
                          oracle.world_matrices(sh.transforms, sh.entity_to_transform).view(np.uint32))


def soa_columns(sc):
    """The scene's pools as separately allocated column arrays (what an SoA engine or the scene loader holds)."""
    t, m = sc.transforms, sc.meshes
    xf = dict(entity=t["entity"].copy(), parent=t["parent"].copy(), position=np.ascontiguousarray(t["position"][:, :3]),
              scale=np.ascontiguousarray(t["scale"][:, :3]), rotation=t["rotation"].copy(),
              self_active=t["selfActive"].copy(), ancestors_active=t["ancestorsActive"].copy(),
              model_with_ancestors=t["modelWithAncestors"].copy())
    mesh = dict(entity=m["entity"].copy(), is_enabled=m["isEnabled"].copy(),
                aabb_min=np.ascontiguousarray(m["aabbMin"][:, :3]), aabb_max=np.ascontiguousarray(m["aabbMax"][:, :3]),
                is_visible=np.full(m.shape[0], 7, np.uint8))
    return xf, mesh


@pytest.mark.parametrize("hier", [False, True])
def test_column_binds_give_the_same_results_as_aos_binds(gpu, oracle, hier):
    """gv_transform_bind_columns / gv_pool_bind_columns: tightly packed SoA columns (12-byte positions, 1-byte flags)
    instead of the 80/48-byte components; incl. a ranged dirty update and the isVisible write-back column."""
    sc = scene.hierarchy_scene(30_011, depth=4, fanout=6) if hier else scene.flat_scene(30_011)
    view = scene.main_camera_view()
    xf, mesh = soa_columns(sc)
    gpu.bind_transform_columns(xf, sc.entity_to_transform)
    gpu.bind_pool_columns(0, mesh)
    gpu.hierarchy_rebuild()
    gpu.cull(0, [view])
    got = gpu.fetch(0, write_back=True, occupancy=sc.count)
    m2 = sc.meshes.copy()
    exp = oracle.prepare_meshes(m2, sc.transforms, sc.entity_to_transform, view)
    assert np.array_equal(got["visible_idx"], exp["visible_idx"])
    assert np.array_equal(got["baked_model"].view(np.uint32), exp["baked_model"].view(np.uint32))
    assert np.array_equal(mesh["is_visible"], m2["isVisible"])  # written through the column

    # move a slice of entities and toggle some flags: only that range is re-mirrored
    lo, hi = 5_000, 9_000
    xf["position"][lo:hi] += np.float32(3.5)
    xf["self_active"][lo:hi:7] ^= 1
    sc.transforms["position"][lo:hi, :3] = xf["position"][lo:hi]
    sc.transforms["selfActive"][lo:hi] = xf["self_active"][lo:hi]
    if not hier:  # (with a hierarchy the host would also have to push ancestorsActive down; flat pools have no children)
        gpu.mark_dirty(0, lo, hi - lo)
        gpu.cull(0, [view])
        got = gpu.fetch(0, write_back=True, occupancy=sc.count)
        m2 = sc.meshes.copy()
        exp = oracle.prepare_meshes(m2, sc.transforms, sc.entity_to_transform, view)
        assert np.array_equal(got["visible_idx"], exp["visible_idx"]) and np.array_equal(mesh["is_visible"], m2["isVisible"])


def test_error_conventions_of_the_c_abi(gpu):
    """Every gv_* returns a status and leaves a message; nothing aborts or throws across the boundary."""
    import ctypes as C
    from garden_amd import lib as L
    raw, ctx = gpu.lib, gpu.ctx
    sc = scene.flat_scene(100)
    gpu.bind_transforms(sc.transforms, sc.entity_to_transform)
    gpu.bind_pool(0, sc.meshes)
    gpu.hierarchy_rebuild()

    def failed(rc, needle):
        assert rc == L.GV_E_ARG or rc == L.GV_E_STATE, rc
        assert needle in raw.gv_last_error(ctx).decode(), raw.gv_last_error(ctx)

    layout_t = L.GvTransformLayout(*[0] * 8)
    layout_m = L.GvMeshLayout(0, 14, 15, 16, 32)
    big = 1 << 28  # slot ids are 28-bit
    failed(raw.gv_transform_bind(ctx, sc.transforms.ctypes.data, 80, big, C.byref(layout_t), None, 0), "28-bit")
    failed(raw.gv_pool_bind(ctx, 0, sc.meshes.ctypes.data, 48, big, C.byref(layout_m)), "28-bit")
    failed(raw.gv_pool_bind(ctx, 16, sc.meshes.ctypes.data, 48, 10, C.byref(layout_m)), "pool_id 16")
    failed(raw.gv_pool_bind(ctx, 0, sc.meshes.ctypes.data, 20, 10, C.byref(layout_m)), "stride")
    failed(raw.gv_pool_bind(ctx, 0, None, 48, 10, C.byref(layout_m)), "bad argument")
    failed(raw.gv_mark_dirty(ctx, 9, 0, 1), "unknown kind")
    failed(raw.gv_sweep(ctx, 7), "unknown mode")
    views = (L.GvView * 9)()
    failed(raw.gv_cull(ctx, 0, views, 9), "view")          # more than GV_MAX_VIEWS
    failed(raw.gv_cull(ctx, 5, views, 1), "pool")          # never bound
    n = C.c_uint32()
    failed(raw.gv_result_count(ctx, 7, C.byref(n)), "no results")
    res = L.GvResult()
    failed(raw.gv_results_fetch(ctx, 7, 0, C.byref(res)), "view")
    failed(raw.gv_sort(ctx, 6, 0), "no emitted records")
    gpu.mark_dirty(0, 0, 1)  # transforms changed: the world cache is stale
    gpu.sync()
    out = np.zeros((1, 12), np.float32)
    failed(raw.gv_get_world(ctx, 0, 1, out.ctypes.data), "gv_sweep has not run")
    gpu.sweep(0)
    failed(raw.gv_get_world(ctx, 90, 20, out.ctypes.data), "outside the pool")
    assert raw.gv_cull(None, 0, views, 1) == L.GV_E_ARG    # NULL context: status only
    # the context is still usable afterwards
    gpu.cull(0, [scene.main_camera_view()])
    assert gpu.result_count(0) >= 0


@pytest.mark.parametrize("hier", [False, True])
@pytest.mark.parametrize("ctx_name", ["gpu", "gpu_slot_order", "gpu_bounds"])
def test_pools_that_grow_are_appended_to_the_mirror(request, oracle, hier, ctx_name):
    """Entities created after the first bind: the pools are re-bound with a larger occupancy (and a moved base, as
    LinearPool::create reallocates) and NO rebuild request — the new slots are appended to the mirror; once the
    unsorted tail passes 1/8 of the pool the library re-orders by itself. Also shrinking (a rebuild) and a mesh pool
    that grows alone."""
    vis = request.getfixturevalue(ctx_name)
    full = scene.hierarchy_scene(40_000, depth=4, fanout=5) if hier else scene.flat_scene(40_000)
    view = scene.main_camera_view()

    def cut(n_t, n_m):
        """The first n_t transforms / n_m meshes in freshly allocated arrays; links to slots beyond the cut dropped."""
        tr = full.transforms[:n_t].copy()
        e2t = full.entity_to_transform.copy()
        e2t[e2t >= n_t] = 0xFFFFFFFF  # entities whose transform does not exist yet
        return scene.Scene(full.meshes[:n_m].copy(), tr, e2t)

    def check(sc, dirty_meshes=None):
        vis.bind_transforms(sc.transforms, sc.entity_to_transform)
        vis.bind_pool(0, sc.meshes)
        if dirty_meshes:
            vis.mark_dirty(2, dirty_meshes[0], dirty_meshes[1], pool_id=0)
        vis.cull(0, [view])
        got = vis.fetch(0, write_back=False, occupancy=sc.count)
        m2 = sc.meshes.copy()
        exp = oracle.prepare_meshes(m2, sc.transforms, sc.entity_to_transform, view)
        assert np.array_equal(got["visible_idx"], np.sort(exp["visible_idx"]))
        o = np.argsort(exp["visible_idx"], kind="stable")
        assert np.array_equal(got["baked_model"].view(np.uint32), exp["baked_model"][o].view(np.uint32))
        assert np.array_equal(got["is_visible"], m2["isVisible"])
        vis.sweep(1)
        assert np.array_equal(vis.get_world(0, sc.transforms.shape[0]).view(np.uint32),
                              oracle.world_matrices(sc.transforms, sc.entity_to_transform).view(np.uint32))

    sc = cut(30_000, 30_000)
    vis.bind_transforms(sc.transforms, sc.entity_to_transform)
    vis.bind_pool(0, sc.meshes)
    vis.hierarchy_rebuild()
    check(sc)
    check(cut(30_500, 30_500))     # +500 entities: appended
    check(cut(31_700, 31_700))     # appended again
    check(cut(31_700, 32_000))     # meshes of entities whose transforms do not exist yet (no transform: filtered out)
    # ... which arrive now: those meshes gained a TransformComponent, which the caller reports like any other change
    check(cut(32_000, 32_000), dirty_meshes=(31_700, 300))
    before = vis.stats()["mirror_reorders"]  # (counted since gv_create; the context is shared by the module's tests)
    check(cut(36_000, 36_000))     # the tail passes 1/8 of the pool: the library re-orders (on the device: gv_reorder.hip)
    assert vis.stats()["mirror_reorders"] - before == (0 if ctx_name == "gpu_slot_order" else 1)
    check(cut(40_000, 40_000))
    check(cut(25_000, 25_000))     # shrink: rebuild
    check(cut(25_100, 25_100))


@pytest.mark.parametrize("kind", ["flat", "hier", "shuffled"])
def test_the_mirror_is_reordered_on_the_device_after_entity_churn(gpu, oracle, kind, monkeypatch):
    """entities created in batches until the unsorted tail of the mirror passes 1/8 of the pools, twice over.
    The library then puts the mirror back into spatial order on the device (Morton codes of the roots, the radix kernels on
    bare keys, one permuting pass per stream; gv_mirror.cpp reorder_*_device) instead of rebuilding it on the host. After every
    step: visible set, records, isVisible and every world matrix against the oracle; then dirty marks of every kind (scattered
    slots, a large range through the device-side gather, re-parenting) must land on the re-ordered entries, and a second mesh
    pool bound to the same transforms must have followed them. 'shuffled': mesh slots do not pair with transform slots."""
    n = 260_000
    full = scene.hierarchy_scene(n, depth=4, fanout=5) if kind == "hier" else scene.flat_scene(n)
    rng = np.random.Generator(np.random.PCG64(4))
    if kind == "shuffled":
        full = scene.Scene(full.meshes[rng.permutation(n)].copy(), full.transforms, full.entity_to_transform)
    view = scene.main_camera_view()
    second = full.meshes[::3].copy()  # another mesh system over the same entities

    def cut(k):
        e2t = full.entity_to_transform.copy()
        e2t[e2t >= k] = 0xFFFFFFFF
        return scene.Scene(full.meshes[:k].copy(), full.transforms[:k].copy(), e2t)

    def check(sc, rebind=True):
        if rebind:
            gpu.bind_transforms(sc.transforms, sc.entity_to_transform)
            gpu.bind_pool(0, sc.meshes)
            gpu.bind_pool(1, second)
            # meshes whose entity has only now received its transform: reported like any other change of a mesh
            gpu.mark_dirty(2, 0, second.shape[0], pool_id=1)
            if kind == "shuffled":
                gpu.mark_dirty(2, 0, sc.meshes.shape[0], pool_id=0)
        for pool, meshes in ((0, sc.meshes), (1, second)):
            gpu.cull(pool, [view])
            got = gpu.fetch(0, write_back=False, occupancy=meshes.shape[0])
            m2 = meshes.copy()
            exp = oracle.prepare_meshes(m2, sc.transforms, sc.entity_to_transform, view)
            assert np.array_equal(got["visible_idx"], np.sort(exp["visible_idx"]))
            o = np.argsort(exp["visible_idx"], kind="stable")
            assert np.array_equal(got["baked_model"].view(np.uint32), exp["baked_model"][o].view(np.uint32))
            assert np.array_equal(got["is_visible"], m2["isVisible"])
        gpu.sweep(1)
        assert np.array_equal(gpu.get_world(0, sc.transforms.shape[0]).view(np.uint32),
                              oracle.world_matrices(sc.transforms, sc.entity_to_transform).view(np.uint32))

    import torch
    from garden_amd.multi import expand_mask_rows, mask_words

    def check_bits(sc):
        """The visible list as one bit per MIRROR entry decodes to the same set through the entry -> slot table as it is now."""
        k = sc.meshes.shape[0]
        table = gpu.mirror_slots(0, k)
        assert np.array_equal(np.sort(table), np.arange(k))
        gpu.cull(0, [view])
        shard = torch.full((1 + mask_words(k),), -1, dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()  # (the fill runs on torch's stream; the library's stream is non-blocking: no implicit order between them)
        gpu.copy_mask_device(0, shard.data_ptr(), mask_words(k))
        gpu.wait()
        exp = oracle.prepare_meshes(sc.meshes.copy(), sc.transforms, sc.entity_to_transform, view)
        slots, counts = expand_mask_rows(shard.view(1, -1), k, entry_tables=[table], index_bases=[0])
        assert counts.tolist() == [exp["draw_count"]] and np.array_equal(slots, np.sort(exp["visible_idx"]).astype(np.int64))

    sizes = [150_000, 160_000, 172_000, 200_000, 215_000, 236_000, 260_000]
    before = gpu.stats()["mirror_reorders"]
    sc = cut(sizes[0])
    gpu.bind_transforms(sc.transforms, sc.entity_to_transform)
    gpu.bind_pool(0, sc.meshes)
    gpu.bind_pool(1, second)
    gpu.hierarchy_rebuild()
    check(sc, rebind=False)
    epochs = [gpu.mirror_epoch(0)]
    for k in sizes[1:]:
        sc = cut(k)
        check(sc)
        epochs.append(gpu.mirror_epoch(0))  # the table changed (appended slots, and twice a re-order): consumers can tell
        assert epochs[-1] > epochs[-2]
        check_bits(sc)
    assert gpu.stats()["mirror_reorders"] - before >= 2
    assert gpu.mirror_epoch(0) == epochs[-1] and gpu.mirror_epoch(1) > 0  # ... and stays put while it does not
    # changes after the re-order land where the entries now are
    lo = 3_000
    sc.transforms["position"][lo:lo + 5_000, :3] += rng.normal(0, 25, (5_000, 3)).astype(np.float32)
    gpu.mark_dirty(0, lo, 5_000)  # GV_DIRTY_TRANSFORM, a large range: device-side gather
    for s in rng.integers(0, n, 40):
        sc.transforms["position"][s, :3] += np.float32(11)
        sc.transforms["selfActive"][s] ^= 1
        gpu.mark_dirty(0, int(s), 1)
    sc.meshes["isEnabled"][100:3_000] ^= 1
    gpu.mark_dirty(2, 100, 2_900, pool_id=0)  # GV_DIRTY_MESH
    check(sc, rebind=False)
    if kind == "hier":
        kids = np.flatnonzero(sc.transforms["parent"] != 0)[:50]
        sc.transforms["parent"][kids] = 0  # re-parented to the root: links changed
        for s in kids:
            gpu.mark_dirty(1, int(s), 1)  # GV_DIRTY_HIERARCHY
        check(sc, rebind=False)
    sc.transforms["rotation"][:] = sc.transforms["rotation"][::-1].copy()
    gpu.mark_dirty(0, 0, n)  # the whole pool: dense path
    check(sc, rebind=False)


@pytest.mark.parametrize("hier", [False, True])
@pytest.mark.parametrize("ctx_name", ["gpu", "gpu_slot_order"])
def test_large_dirty_ranges_take_the_device_side_gather(request, oracle, hier, ctx_name):
    """GV_DIRTY_TRANSFORM over >= 2 Ki slots of an AoS pool: the raw components are copied and gathered on the device
    (the host staging of those slots goes stale); small ranges, link changes and dense host re-mirrors afterwards must
    still see coherent data."""
    vis = request.getfixturevalue(ctx_name)
    n = 150_000
    sc = scene.hierarchy_scene(n, depth=4, fanout=6) if hier else scene.flat_scene(n)
    view = scene.main_camera_view()
    rng = np.random.Generator(np.random.PCG64(99))
    vis.bind_transforms(sc.transforms, sc.entity_to_transform)
    vis.bind_pool(0, sc.meshes)
    vis.hierarchy_rebuild()

    def check():
        vis.cull(0, [view])
        got = vis.fetch(0, write_back=False, occupancy=n)
        m2 = sc.meshes.copy()
        exp = oracle.prepare_meshes(m2, sc.transforms, sc.entity_to_transform, view)
        assert np.array_equal(got["visible_idx"], np.sort(exp["visible_idx"]))
        o = np.argsort(exp["visible_idx"], kind="stable")
        assert np.array_equal(got["baked_model"].view(np.uint32), exp["baked_model"][o].view(np.uint32))
        assert np.array_equal(got["is_visible"], m2["isVisible"])
        vis.sweep(1)
        assert np.array_equal(vis.get_world(0, n).view(np.uint32),
                              oracle.world_matrices(sc.transforms, sc.entity_to_transform).view(np.uint32))

    def move(lo, cnt):
        sc.transforms["position"][lo:lo + cnt, :3] += rng.normal(0, 30, (cnt, 3)).astype(np.float32)
        sc.transforms["rotation"][lo:lo + cnt] = sc.transforms["rotation"][lo:lo + cnt][::-1]
        sc.transforms["selfActive"][lo:lo + cnt] ^= (rng.random(cnt) < 0.05).astype(np.uint8)
        vis.mark_dirty(0, lo, cnt)

    check()
    up0 = vis.stats()["upload_bytes"]
    move(20_000, 60_000)            # device-side gather: 80 raw bytes per slot
    check()
    assert vis.stats()["upload_bytes"] - up0 == 59_999 * 80 + 75  # whole strides + the last element up to its last bound field
    move(70_000, 300)               # small range inside the stale region: host gather + scatter
    check()
    move(0, n)                      # the whole pool
    check()
    if hier:                        # link changes go through the host (depth / cycle validation), staging refreshed as needed
        for s_ in range(100_000, 100_400):
            sc.transforms["parent"][s_] = sc.transforms["entity"][s_ - 90_000]
        vis.mark_dirty(1, 100_000, 400)
        check()
        sc.transforms["parent"][50_000:130_000:7] = 0
        vis.mark_dirty(1, 50_000, 80_000)   # a large ranged hierarchy change: dense host path over stale staging
        check()
    move(10_000, 20_000)
    check()


@pytest.mark.parametrize("hier", [False, True])
@pytest.mark.parametrize("ctx_name", ["gpu", "gpu_slot_order"])
def test_large_dirty_mesh_ranges_take_the_device_side_gather(request, oracle, hier, ctx_name):
    """GV_DIRTY_MESH over large ranges of an AoS pool (round 3): the raw components are copied and gathered on the device —
    boxes, enable flags, and the entity -> transform lookup (through a device copy of the entity map), including meshes that
    change hands (an exactly paired pool is demoted) — while the host staging of those slots goes stale; small ranges and dense
    host re-mirrors afterwards must still see coherent data."""
    vis = request.getfixturevalue(ctx_name)
    n = 150_000
    sc = scene.hierarchy_scene(n, depth=4, fanout=6) if hier else scene.flat_scene(n)
    view = scene.main_camera_view()
    rng = np.random.Generator(np.random.PCG64(7))
    vis.bind_transforms(sc.transforms, sc.entity_to_transform)
    vis.bind_pool(0, sc.meshes)
    vis.hierarchy_rebuild()

    def check():
        vis.cull(0, [view])
        got = vis.fetch(0, write_back=False, occupancy=n)
        m2 = sc.meshes.copy()
        exp = oracle.prepare_meshes(m2, sc.transforms, sc.entity_to_transform, view)
        assert np.array_equal(got["visible_idx"], np.sort(exp["visible_idx"]))
        o = np.argsort(exp["visible_idx"], kind="stable")
        assert np.array_equal(got["baked_model"].view(np.uint32), exp["baked_model"][o].view(np.uint32))
        assert np.array_equal(got["is_visible"], m2["isVisible"])
        return got["draw_count"]

    def edit(lo, cnt):
        sc.meshes["aabbMax"][lo:lo + cnt, :3] *= rng.uniform(0.5, 3.0, (cnt, 3)).astype(np.float32)
        sc.meshes["isEnabled"][lo:lo + cnt] ^= (rng.random(cnt) < 0.05).astype(np.uint8)
        sc.meshes["aabbMin"][lo:lo + cnt:97, :3] = sc.meshes["aabbMax"][lo:lo + cnt:97, :3]  # some boxes collapse (the size <= 0 rule)
        vis.mark_dirty(2, lo, cnt, pool_id=0)

    first = check()
    up0 = vis.stats()["upload_bytes"]
    edit(20_000, 60_000)                      # device-side gather: 48 raw bytes per slot + the entity map
    assert check() != first
    stride, cap = sc.meshes.dtype.itemsize, len(sc.entity_to_transform)
    assert vis.stats()["upload_bytes"] - up0 >= 59_999 * stride + cap * 4
    edit(70_000, 300)                         # a small range inside the stale region: host gather
    check()
    # meshes change hands: the entities of two stretches are swapped (no mesh pairs with its own transform index there any more)
    a, b = slice(30_000, 50_000), slice(90_000, 110_000)
    sc.meshes["entity"][a], sc.meshes["entity"][b] = sc.meshes["entity"][b].copy(), sc.meshes["entity"][a].copy()
    sc.meshes["entity"][52_000:52_500] = 0    # ... and some meshes are destroyed
    vis.mark_dirty(2, 30_000, 80_000, pool_id=0)
    check()
    edit(0, n)                                # the whole pool
    check()
    for lo in range(10, n - 10, 6007):        # many small ranges: the host paths over (formerly) stale staging
        sc.meshes["isEnabled"][lo:lo + 5] = 1
        sc.meshes["aabbMax"][lo:lo + 5, :3] += np.float32(0.25)
        vis.mark_dirty(2, lo, 5, pool_id=0)
    check()


def test_dirty_ranges_from_pools_in_pinned_memory(gpu, oracle):
    """Pools that already live in page-locked memory (an engine allocating its ECS pools with hipHostMalloc, a torch
    pinned tensor): the device-side gather copies straight from them."""
    import torch
    n = 50_000
    sc = scene.flat_scene(n)
    pinned = torch.empty(sc.transforms.nbytes, dtype=torch.uint8).pin_memory()
    tr = np.frombuffer(pinned.numpy(), dtype=sc.transforms.dtype)
    tr[:] = sc.transforms
    view = scene.main_camera_view()
    gpu.bind_transforms(tr, sc.entity_to_transform)
    gpu.bind_pool(0, sc.meshes)
    gpu.hierarchy_rebuild()
    for lo, cnt in ((0, n), (10_000, 5_000), (100, 3_000)):
        tr["position"][lo:lo + cnt, 1] += np.float32(2.5)
        gpu.mark_dirty(0, lo, cnt)
        gpu.cull(0, [view])
        got = gpu.fetch(0, write_back=False, occupancy=n)
        m2 = sc.meshes.copy()
        exp = oracle.prepare_meshes(m2, tr, sc.entity_to_transform, view)
        assert np.array_equal(got["visible_idx"], exp["visible_idx"])
        assert np.array_equal(got["baked_model"].view(np.uint32), exp["baked_model"].view(np.uint32))


def test_pools_of_different_sizes_alternate_on_the_same_view(gpu, oracle):
    """The per-view chunk-total buffers alternate from cull to cull (the self-prefixing emit clears the other one):
    a small pool after a large one must not leave stale totals behind, nor must count-only culls (scan path) in between."""
    big, small = scene.flat_scene(100_000), scene.flat_scene(3_000, seed=scene.SEED + 5)
    view = scene.main_camera_view()
    count_only = dict(view, emit_records=0)
    gpu.bind_transforms(big.transforms, big.entity_to_transform)
    gpu.bind_pool(0, big.meshes)
    gpu.bind_pool(1, small.meshes)  # entities 1..3000 of the big scene's transform pool: a second mesh system
    gpu.hierarchy_rebuild()
    exp = {}
    for pid, sc in ((0, big), (1, small)):
        m2 = sc.meshes.copy()
        e = oracle.prepare_meshes(m2, big.transforms, big.entity_to_transform, view)
        exp[pid] = (np.sort(e["visible_idx"]), e["draw_count"])
    for pid, v in ((0, view), (1, view), (0, count_only), (1, view), (1, count_only), (0, view), (0, view), (1, count_only),
                   (1, view), (0, view)):
        gpu.cull(pid, [v])
        if v["emit_records"]:
            got = gpu.fetch(0, write_back=False, occupancy=(big if pid == 0 else small).count)
            assert np.array_equal(got["visible_idx"], exp[pid][0]), pid
        assert gpu.result_count(0) == exp[pid][1], pid


def test_native_rccl_exchange_single_rank(oracle):
    """gv_exchange_*: the C-ABI's own all-gather of the visible-list shards (RCCL bound at run time). One rank is all a
    1-GPU box can run; the multi-rank wire format is the one test_host_logic.py checks through gloo."""
    import torch
    from garden_amd.lib import GpuVisibility
    sc = scene.flat_scene(40_000)
    view = scene.main_camera_view()
    with GpuVisibility(device=0) as vis:
        vis.bind_transforms(sc.transforms, sc.entity_to_transform)
        vis.bind_pool(0, sc.meshes)
        vis.hierarchy_rebuild()
        vis.exchange_init(GpuVisibility.exchange_unique_id(), 0, 1)
        cap = 16_384
        gathered = torch.full((1 * (1 + cap),), -1, dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()  # (the fill runs on torch's stream; the library's stream is non-blocking: no implicit order between them)
        for frame in range(3):
            vis.cull(0, [view])
            vis.exchange_shards(0, cap, 5_000_000, gathered.data_ptr())
        vis.wait()
        raw = vis.fetch(0, write_back=False, occupancy=sc.count, order="raw")
        g = gathered.cpu().numpy()
        k = raw["draw_count"]
        assert g[0] == k and 0 < k <= cap
        assert np.array_equal(g[1:1 + k].astype(np.int64), raw["visible_idx"].astype(np.int64) + 5_000_000)
        # the same frame as bit shards (gv_exchange_masks), through every transport pattern
        from garden_amd.multi import expand_mask_rows, mask_words
        words = mask_words(sc.count)
        table = vis.mirror_slots(0, sc.count)
        for mode in (0, 1, 2):
            vis.exchange_set_mode(mode)
            rows = torch.full((1, 1 + words), -1, dtype=torch.int32, device="cuda:0")
            torch.cuda.synchronize()  # (the fill runs on torch's stream; the library's stream is non-blocking: no implicit order between them)
            vis.cull(0, [view])
            vis.exchange_masks(0, words, rows.data_ptr())
            vis.wait()
            slots, counts = expand_mask_rows(rows, sc.count, entry_tables=[table], index_bases=[0])
            assert counts.tolist() == [k] and np.array_equal(slots, np.sort(raw["visible_idx"]).astype(np.int64)), mode
        vis.exchange_set_mode(0)
        vis.exchange_shutdown()
        with pytest.raises(Exception):
            vis.exchange_shards(0, cap, 0, gathered.data_ptr())  # GV_E_STATE after shutdown
        with pytest.raises(Exception):
            vis.exchange_masks(0, words, gathered.data_ptr())


@pytest.mark.gpu
def test_exchange_of_a_named_pool_after_another_system_was_culled(oracle):
    """Two mesh systems culled one after the other, THEN exchanged (an engine that batches its culls): gv_pool_exchange_visible
    carries the named pool's list, the view-indexed gv_exchange_visible the list of the pool culled last. Real RCCL, one rank;
    every travel pattern; rows == the oracle's visible set of that pool."""
    import torch
    from garden_amd.lib import GpuVisibility
    big, small = scene.flat_scene(100_000), scene.flat_scene(3_000, seed=scene.SEED + 5)
    view = scene.main_camera_view()
    class _Span:
        pass

    exp = {}
    for pid, sc in ((0, big), (1, small)):
        e = oracle.prepare_meshes(sc.meshes.copy(), big.transforms, big.entity_to_transform, view)
        exp[pid] = np.sort(e["visible_idx"]).astype(np.int64)
    assert len(exp[0]) > 4 * len(exp[1]) > 0
    with GpuVisibility(device=0) as vis:
        vis.bind_transforms(big.transforms, big.entity_to_transform)
        vis.bind_pool(0, big.meshes)
        vis.bind_pool(1, small.meshes)
        vis.hierarchy_rebuild()
        vis.exchange_init(GpuVisibility.exchange_unique_id(), 0, 1)

        for mode in (0, 1, 2):
            vis.exchange_set_mode(mode)
            vis.cull(0, [view])
            vis.cull(1, [view])
            for pool_id, want in ((0, exp[0]), (None, exp[1]), (1, exp[1])):
                sent = vis.exchange_visible(0, index_base=7, pool_id=pool_id)
                f = vis.exchange_acquire(sent["frame"])
                vis.wait()
                assert f["complete"] and int(f["counts"][0]) == len(want), (mode, pool_id)
                span = _Span()
                span.__cuda_array_interface__ = {"shape": (f["row_words"],), "typestr": "<i4", "data": (int(f["ptr"]), False), "version": 2}
                row = torch.as_tensor(span, device="cuda:0").cpu().numpy().view(np.uint32)[:1 + len(want)]
                assert row[0] == len(want) and np.array_equal(np.sort(row[1:].astype(np.int64)), want + 7), (mode, pool_id)
        with pytest.raises(Exception):
            vis.exchange_visible(0, pool_id=5)  # never bound
        vis.exchange_shutdown()


@pytest.mark.gpu
@pytest.mark.parametrize("hier", [False, True])
def test_sphere_pretest_agrees_with_the_exact_test_around_every_plane(gpu, oracle, hier):
    """The cull kernels decide entries away from the planes by a sphere bound and run the exact 8-corner test only in
    the band around a plane (gv_device.hpp classify_sphere). Here most entities sit IN that band: positions are random
    points on the planes of a perspective and an orthographic frustum, pushed along the normal by up to +-12 m (boxes
    reach ~3.5 m), near and far from the camera (1 m .. 90 km: the slack scales with the magnitude), with non-finite
    and extreme values sprinkled over AABBs and transforms. Single views, a batched pair, Hi-Z on: same visible sets."""
    n = 120_000
    sc = scene.hierarchy_scene(n, depth=2, fanout=4) if hier else scene.flat_scene(n, seed=77)
    n = sc.count
    rng = np.random.default_rng(2024)
    main, cascade = scene.main_camera_view(use_hiz=0), scene.cascade_view(index=0, size=6000.0, depth=30000.0)
    t, m = sc.transforms, sc.meshes
    roots = np.flatnonzero(t["parent"] == 0) if hier else np.arange(n)
    planes = np.concatenate([oracle.frustum(main["view_proj"]), oracle.frustum(cascade["view_proj"])])
    k = rng.integers(0, len(planes), roots.size)
    nrm, w = planes[k, :3].astype(np.float64), planes[k, 3].astype(np.float64)
    # a random point, projected onto its plane, then offset along the normal
    p = rng.normal(size=(roots.size, 3)) * (10.0 ** rng.uniform(0, 4.95, (roots.size, 1)))
    p -= ((p * nrm).sum(1) + w)[:, None] * nrm
    p += nrm * rng.uniform(-12, 12, (roots.size, 1))
    t["position"][roots, :3] = p.astype(np.float32)
    specials = np.array([np.nan, np.inf, -np.inf, 1e-42, -0.0, 3e38, -3e38, 1e-30], dtype=np.float32)
    for arr, field, width in ((m, "aabbMin", 3), (m, "aabbMax", 3), (t, "position", 3), (t, "scale", 3), (t, "rotation", 4)):
        rows = rng.choice(n, 600, replace=False)
        arr[field][rows, rng.integers(0, width, 600)] = specials[rng.integers(0, len(specials), 600)]
    lop = rng.choice(n, 3000, replace=False)  # lopsided boxes: the sphere is centred on the translation, not on the box
    m["aabbMin"][lop, :3] += rng.uniform(-3, 3, (3000, 3)).astype(np.float32)
    depth = scene.synthetic_depth(512, 256)
    for views in ([main], [cascade], [dict(main, use_hiz=1), cascade], [dict(main, camera_position=np.array([40.0, -7.0, 3.0, 0.0], np.float32))]):
        res = run_both(gpu, oracle, sc, views, hiz_depth=depth)
        for (got, gv, exp, ev), v in zip(res, views):
            assert exp["draw_count"] > 1000
            assert got["draw_count"] == exp["draw_count"] and np.array_equal(got["visible_idx"], exp["visible_idx"])
            assert same_bits_or_both_nan(got["baked_model"], exp["baked_model"])
            if v["shadow_pass"] < 0:
                assert np.array_equal(gv, ev)


@pytest.mark.gpu
def test_deferred_sorts_of_a_small_pool_reach_every_reader(gpu, oracle):
    """gv_sort on a pool of <= 16384 slots is deferred and batched over the views of the cull. Whoever reads the records
    first — a device accessor, gv_wait, a fetch of ANOTHER view — must see them sorted; a sort nobody read before the
    next cull is dropped; re-sorting in the other direction after a read works on the published results."""
    import torch
    sc = scene.flat_scene(12_000, seed=41)
    views = [scene.main_camera_view(), dict(scene.cascade_view(index=0, size=9000.0, depth=30000.0)),
             dict(scene.cascade_view(index=1, size=5000.0, depth=30000.0))]
    gpu.bind_transforms(sc.transforms, sc.entity_to_transform)
    gpu.bind_pool(0, sc.meshes)
    gpu.hierarchy_rebuild()
    exp = [oracle.prepare_meshes(sc.meshes.copy(), sc.transforms, sc.entity_to_transform, v, sort="descending" if k == 1 else "ascending")
           for k, v in enumerate(views)]

    def canon(got, descending):  # ties: by slot, as the oracle breaks them
        d = got["distance_sq"]
        o = np.lexsort((got["visible_idx"], -d if descending else d))
        return got["visible_idx"][o]

    gpu.cull(0, views)
    for k in range(3):
        gpu.sort(k, descending=(k == 1))
    # 1. a device accessor of view 1 is the first reader
    dst = torch.full((sc.count,), -1, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()  # (the fill runs on torch's stream; the library's stream is non-blocking: no implicit order between them)
    gpu.copy_idx_device(1, dst.data_ptr(), sc.count)
    n1 = gpu.result_count(1)
    got1 = gpu.fetch(1, write_back=False, occupancy=sc.count, order="raw")
    assert n1 == exp[1]["draw_count"] and np.array_equal(dst[:n1].cpu().numpy().astype(np.uint32), got1["visible_idx"])
    d = got1["distance_sq"]
    assert np.all(d[:-1] >= d[1:]) and np.array_equal(canon(got1, True), exp[1]["visible_idx"])
    # 2. the other views were sorted by the same flush
    for k in (0, 2):
        g = gpu.fetch(k, write_back=False, occupancy=sc.count, order="raw")
        assert np.all(g["distance_sq"][:-1] <= g["distance_sq"][1:]) and np.array_equal(canon(g, False), exp[k]["visible_idx"])
    # 3. re-sort one view the other way after it has been read
    gpu.sort(0, descending=True)
    g = gpu.fetch(0, write_back=False, occupancy=sc.count, order="raw")
    assert np.all(g["distance_sq"][:-1] >= g["distance_sq"][1:]) and set(g["visible_idx"].tolist()) == set(exp[0]["visible_idx"].tolist())
    # 4. a sort nobody read is dropped by the next cull: slot order again
    gpu.sort(2, descending=True)
    gpu.cull(0, views)
    g = gpu.fetch(2, write_back=False, occupancy=sc.count)
    assert np.array_equal(g["visible_idx"], np.sort(exp[2]["visible_idx"]))
    # 5. gv_wait flushes too
    gpu.sort(1, descending=True)
    gpu.wait()
    dev = gpu.results_device(1)
    assert dev.visible_idx
    g = gpu.fetch(1, write_back=False, occupancy=sc.count, order="raw")
    assert np.array_equal(canon(g, True), exp[1]["visible_idx"]) and np.all(g["distance_sq"][:-1] >= g["distance_sq"][1:])


def test_device_gather_never_reads_past_the_callers_pool(oracle):
    """ADVICE r1 (gv_mirror.cpp): a component whose first bound field sits above offset 0 (a header in front of `entity`),
    bound so that the pool ENDS at the last byte before an inaccessible page, with a dirty range that reaches the last
    slot: the device-side AoS gather must read (count - 1) strides + the extent of the bound fields, not count strides
    from the lowest field (which would run 16 bytes into the guard page and fault). Also: a dirty mark whose
    first + count wraps around uint32 must not be dropped."""
    import ctypes
    import mmap

    from garden_amd.lib import GpuVisibility
    from garden_amd.pools import TRANSFORM_DTYPE
    n = 6000
    sc = scene.flat_scene(n)
    names = list(TRANSFORM_DTYPE.names)
    dt = np.dtype(dict(names=names, formats=[TRANSFORM_DTYPE.fields[k][0] for k in names],
                       offsets=[TRANSFORM_DTYPE.fields[k][1] + 16 for k in names], itemsize=96))
    page = mmap.PAGESIZE
    size = n * dt.itemsize
    pages = (size + page - 1) // page + 1
    mm = mmap.mmap(-1, pages * page)
    base = ctypes.addressof(ctypes.c_char.from_buffer(mm))
    libc = ctypes.CDLL(None, use_errno=True)
    libc.mprotect.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    assert libc.mprotect(base + (pages - 1) * page, page, 0) == 0  # PROT_NONE guard page after the pool
    start = (pages - 1) * page - size  # the pool's last byte is the last accessible byte
    tr = np.frombuffer(mm, dtype=dt, count=n, offset=start)
    for k in names:
        tr[k] = sc.transforms[k]
    view = scene.main_camera_view()
    with GpuVisibility(device=0) as vis:
        vis.bind_transforms(tr, sc.entity_to_transform)
        vis.bind_pool(0, sc.meshes)
        vis.hierarchy_rebuild()
        tr["position"][1000:, :3] *= np.float32(0.5)
        sc.transforms["position"][1000:, :3] *= np.float32(0.5)
        vis.mark_dirty(0, 1000, n - 1000)  # >= 2048 slots, up to the last one: the device-side gather
        up0 = vis.stats()["upload_bytes"]
        vis.cull(0, [view])
        got = vis.fetch(0, write_back=False, occupancy=n)
        assert vis.stats()["upload_bytes"] - up0 == (n - 1000 - 1) * 96 + 75
        m2 = sc.meshes.copy()
        exp = oracle.prepare_meshes(m2, sc.transforms, sc.entity_to_transform, view)
        assert np.array_equal(got["visible_idx"], np.sort(exp["visible_idx"])) and np.array_equal(got["is_visible"], m2["isVisible"])
        # first + count overflows uint32: saturates to "up to the end of the pool" instead of wrapping to nothing
        tr["position"][3000:, :3] *= np.float32(0.5)
        sc.transforms["position"][3000:, :3] *= np.float32(0.5)
        vis.mark_dirty(0, 3000, 0xFFFFFFFF)
        vis.cull(0, [view])
        got = vis.fetch(0, write_back=False, occupancy=n)
        m2 = sc.meshes.copy()
        exp = oracle.prepare_meshes(m2, sc.transforms, sc.entity_to_transform, view)
        assert np.array_equal(got["visible_idx"], np.sort(exp["visible_idx"])) and np.array_equal(got["is_visible"], m2["isVisible"])
    del tr
    assert libc.mprotect(base + (pages - 1) * page, page, 3) == 0


@pytest.mark.parametrize("ctx_name", ["gpu", "gpu_slot_order"])
def test_incremental_sweep_recomputes_exactly_the_dirty_subtrees(request, oracle, ctx_name):
    """GV_SWEEP_INCREMENTAL: after dirty ranges only chains through re-mirrored transforms are
    recomputed — every tick the whole cache must equal the oracle's world matrices bit for bit, whatever path the
    dirty range took (device-side AoS gather, scattered host packet, contiguous host upload, re-parenting), and a tick
    without changes must not launch anything."""
    from garden_amd.lib import GV_DIRTY_HIERARCHY, GV_DIRTY_TRANSFORM, GV_SWEEP_INCREMENTAL, GV_SWEEP_VALU
    vis = request.getfixturevalue(ctx_name)
    n = 120_000
    sc = scene.hierarchy_scene(n, depth=4, fanout=6)
    tr = sc.transforms
    rng = np.random.Generator(np.random.PCG64(4242))
    vis.bind_transforms(tr, sc.entity_to_transform)
    vis.bind_pool(0, sc.meshes)
    vis.hierarchy_rebuild()

    def check():
        exp = oracle.world_matrices(tr, sc.entity_to_transform)
        got = vis.get_world(0, n)
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))

    def launches():
        return vis.stats()["launches"]["sweep"]

    vis.sweep(GV_SWEEP_INCREMENTAL)  # first use: a full sweep
    check()
    l0 = launches()
    vis.sweep(GV_SWEEP_INCREMENTAL)  # nothing changed: nothing launched
    assert launches() == l0
    check()

    def move(lo, cnt):
        tr["position"][lo:lo + cnt, :3] += rng.normal(0, 25, (cnt, 3)).astype(np.float32)
        tr["scale"][lo:lo + cnt, :3] *= np.float32(1.01)
        vis.mark_dirty(GV_DIRTY_TRANSFORM, lo, cnt)

    # a few roots (level 0 is the first n / 259 slots): their whole subtrees follow
    move(3, 40)
    vis.sweep(GV_SWEEP_INCREMENTAL)
    check()
    # a large range in the leaves: device-side gather path (>= 2048 slots)
    move(60_000, 30_000)
    vis.sweep(GV_SWEEP_INCREMENTAL)
    check()
    # several ticks of small scattered changes, one incremental sweep per tick
    for _ in range(4):
        for lo in rng.integers(0, n - 50, 6):
            move(int(lo), int(rng.integers(1, 50)))
        vis.sweep(GV_SWEEP_INCREMENTAL)
        check()
    # re-parenting (links through the host path) + moving in the same tick; a cull in between must not disturb the cache
    for s_ in range(100_000, 100_300):
        tr["parent"][s_] = tr["entity"][s_ - 99_000]
    vis.mark_dirty(GV_DIRTY_HIERARCHY, 100_000, 300)
    move(500, 100)
    vis.cull(0, [scene.main_camera_view()])
    vis.sweep(GV_SWEEP_INCREMENTAL)
    check()
    # most of the pool: falls back to a full sweep; then a full-mode sweep and an incremental no-op agree
    move(0, n)
    vis.sweep(GV_SWEEP_INCREMENTAL)
    check()
    vis.sweep(GV_SWEEP_VALU)
    l1 = launches()
    vis.sweep(GV_SWEEP_INCREMENTAL)
    assert launches() == l1
    check()
    # records emitted from the cache (emit reads world[] when it is current) are the oracle's
    move(7, 20)
    vis.sweep(GV_SWEEP_INCREMENTAL)
    view = scene.main_camera_view()
    vis.cull(0, [view])
    got = vis.fetch(0, write_back=False, occupancy=n)
    m2 = sc.meshes.copy()
    exp = oracle.prepare_meshes(m2, tr, sc.entity_to_transform, view)
    o = np.argsort(exp["visible_idx"], kind="stable")
    assert np.array_equal(got["visible_idx"], exp["visible_idx"][o])
    assert np.array_equal(got["baked_model"].view(np.uint32), exp["baked_model"][o].view(np.uint32))


@pytest.mark.parametrize("dtype", [np.uint8, np.uint32])
def test_ready_counts_filter_like_the_derived_predicates(gpu, oracle, dtype):
    """gv_pool_bind_ready: a derived system's getReadyMeshesAsync result per slot (sprite.cpp:90-97: 0 until the
    descriptor set exists; counts above 1 = instances). Count 0 ends like readyCount == 0 in mesh.cpp:158-165 — not
    drawn, isVisible = false — and instance_count is the sum of the drawn meshes' counts (mesh.cpp:174)."""
    from garden_amd.lib import GV_DIRTY_MESH
    n = 50_000
    sc = scene.hierarchy_scene(n, depth=3, fanout=5)
    rng = np.random.default_rng(11)
    ready = rng.choice(np.array([0, 1, 1, 1, 2, 7], dtype=dtype), n)
    view = scene.main_camera_view()
    shadow = scene.cascade_view(index=0, size=5000.0)
    gpu.bind_transforms(sc.transforms, sc.entity_to_transform)
    gpu.bind_pool(0, sc.meshes)
    gpu.bind_ready(0, ready)
    gpu.hierarchy_rebuild()

    def check(views):
        gpu.cull(0, views)
        for vi, v in enumerate(views):
            got = gpu.fetch(vi, write_back=False, occupancy=n)
            m2 = sc.meshes.copy()
            exp = oracle.prepare_meshes(m2, sc.transforms, sc.entity_to_transform, v, ready=ready)
            assert exp["draw_count"] > 0 and got["draw_count"] == exp["draw_count"]
            if v.get("emit_records", 1):
                assert np.array_equal(got["visible_idx"], exp["visible_idx"])
                assert got["instance_count"] == exp["instance_count"] != exp["draw_count"]
            elif v["shadow_pass"] < 0:
                assert got["instance_count"] == exp["instance_count"]
            if v["shadow_pass"] < 0:
                assert np.array_equal(got["is_visible"], m2["isVisible"])
                assert not np.any(got["is_visible"][ready == 0])

    check([view])
    check([view, shadow])
    check([dict(view, emit_records=0)])
    # resources finish loading / get evicted: counts change, reported as mesh dirt
    ready[1000:3000] = 1
    ready[20_000:20_500] = 0
    gpu.mark_dirty(GV_DIRTY_MESH, 1000, 2000)
    gpu.mark_dirty(GV_DIRTY_MESH, 20_000, 500)
    check([view])
    # the AVX2 form of the oracle applies the same rule
    soa = oracle.Avx2Scene(sc.meshes.copy(), sc.transforms, sc.entity_to_transform, ready=ready)
    a = soa.prepare_meshes(view)
    b = oracle.prepare_meshes(sc.meshes.copy(), sc.transforms, sc.entity_to_transform, view, ready=ready)
    assert a["draw_count"] == b["draw_count"] and a["instance_count"] == b["instance_count"]
    assert np.array_equal(np.sort(a["visible_idx"]), np.sort(b["visible_idx"]))
    soa.close()
    # without the column every slot counts 1 again
    gpu.bind_ready(0, None)
    gpu.cull(0, [view])
    got = gpu.fetch(0, write_back=False, occupancy=n)
    exp = oracle.prepare_meshes(sc.meshes.copy(), sc.transforms, sc.entity_to_transform, view)
    assert got["draw_count"] == exp["draw_count"] == got["instance_count"]


def test_a_tick_of_small_pools_is_culled_emitted_sorted_and_published_together(oracle):
    """gv_cull_batch_begin: the culls of several engine-sized pools (different sizes, mappings, hierarchy; main camera
    + shadow passes, Hi-Z on the main view) are recorded and launched as ONE cull, ONE emit, ONE sort and ONE publish
    kernel at the first read — results per (pool, view) equal the oracle's, a pool too large for the batch and a
    count-only view take the ordinary path inside the same tick, and a second tick re-uses everything."""
    from garden_amd.lib import GpuVisibility
    rng = np.random.default_rng(5)
    flat = scene.flat_scene(9_000, seed=3)
    hier = scene.hierarchy_scene(12_000, depth=4, fanout=5)
    # three mesh pools over ONE transform pool: slots of `hier`; pool 1 = a shuffled subset, pool 2 = too large for the batch
    tr, e2t = hier.transforms, hier.entity_to_transform
    pool0 = hier.meshes
    pick = rng.permutation(12_000)[:5_000]
    pool1 = hier.meshes[pick].copy()
    big = scene.flat_scene(40_000, seed=9)  # own entities: none has a transform in `tr` beyond slot range -> mostly filtered
    big.meshes["entity"] = (rng.integers(1, 12_000, 40_000)).astype(np.uint32)
    pool2 = big.meshes
    depth = scene.synthetic_depth(512, 256)
    main = scene.main_camera_view(use_hiz=1)
    shadows = [scene.cascade_view(index=k, size=3000.0 + 800 * k) for k in range(3)]
    hz = oracle.Hiz(depth)
    del flat

    def expect(meshes, view):
        m2 = meshes.copy()
        r = oracle.prepare_meshes(m2, tr, e2t, view, hiz=hz if view.get("use_hiz") else None)
        return r, m2["isVisible"]

    with GpuVisibility(device=0, profile_events=True) as vis:
        vis.bind_transforms(tr, e2t)
        for pid, meshes in enumerate((pool0, pool1, pool2)):
            vis.bind_pool(pid, meshes)
        vis.bind_pool(3, pool0[:0])  # an empty pool inside the same tick
        vis.hierarchy_rebuild()
        vis.hiz_build(depth)
        for tick in range(2):
            vis.stats_reset()
            vis.cull_batch_begin()
            vis.cull(0, [main] + shadows)          # recorded: 4 views
            vis.cull(1, [dict(main, use_hiz=0)])   # recorded: 1 view
            vis.cull(2, [dict(main, use_hiz=0)])   # 40 k slots: launched at once
            vis.cull(3, [dict(main, use_hiz=0)])   # nothing to cull: draw count 0
            vis.sort(0, descending=False, pool_id=0)
            vis.sort(2, descending=True, pool_id=0)
            vis.sort(0, descending=True, pool_id=1)
            assert vis.stats()["launches"]["cull"] == 1  # only pool 2 so far
            got0 = [vis.fetch(v, write_back=False, occupancy=12_000, pool_id=0, order="raw") for v in range(4)]
            st = vis.stats()["launches"]
            assert st["cull"] == 2 and st["emit"] == 2 and st["sort"] == 1, st  # pool 2's pair + ONE table launch each
            got1 = vis.fetch(0, write_back=False, occupancy=5_000, pool_id=1, order="raw")
            got2 = vis.fetch(0, write_back=False, occupancy=40_000, pool_id=2)
            assert vis.stats()["launches"] == st  # nothing more was needed
            assert vis.fetch(0, write_back=False, occupancy=0, pool_id=3)["draw_count"] == 0
            for v, view in enumerate([main] + shadows):
                exp, vis_bytes = expect(pool0, view)
                g = got0[v]
                assert g["draw_count"] == exp["draw_count"] > 0
                o = np.argsort(g["visible_idx"], kind="stable")
                e = np.argsort(exp["visible_idx"], kind="stable")
                assert np.array_equal(g["visible_idx"][o], exp["visible_idx"][e])
                assert np.array_equal(g["baked_model"][o].view(np.uint32), exp["baked_model"][e].view(np.uint32))
                if v == 0:
                    assert np.array_equal(g["is_visible"], vis_bytes)
                    assert np.all(np.diff(g["distance_sq"]) >= 0)  # sorted front to back
                if v == 2:
                    assert np.all(np.diff(g["distance_sq"]) <= 0)
            exp, vis_bytes = expect(pool1, dict(main, use_hiz=0))
            o = np.argsort(got1["visible_idx"], kind="stable")
            assert np.array_equal(got1["visible_idx"][o], np.sort(exp["visible_idx"])) and np.array_equal(got1["is_visible"], vis_bytes)
            assert np.all(np.diff(got1["distance_sq"]) <= 0)
            exp, vis_bytes = expect(pool2, dict(main, use_hiz=0))
            assert np.array_equal(got2["visible_idx"], np.sort(exp["visible_idx"])) and np.array_equal(got2["is_visible"], vis_bytes)
            # something moves between the ticks
            tr["position"][:3000, :3] += np.float32(3.0)
            vis.mark_dirty(0, 0, 3000)


RECORD_DTYPES = {
    # render/mesh.hpp:191-205 under an 8-byte-aligned float4x3 (64-byte structs; SortedMesh's bufferIndex fills the tail) ...
    "packed": (np.dtype({"names": ["componentOffset", "bakedModel", "distanceSq"], "formats": ["<u8", ("<f4", 12), "<f4"],
                         "offsets": [0, 8, 56], "itemsize": 64}),
               np.dtype({"names": ["componentOffset", "bakedModel", "distanceSq", "bufferIndex"],
                         "formats": ["<u8", ("<f4", 12), "<f4", "<u4"], "offsets": [0, 8, 56, 60], "itemsize": 64})),
    # ... and under a 16-byte-aligned SIMD float4x3 (80-byte structs with padding after componentOffset and at the end)
    "simd": (np.dtype({"names": ["componentOffset", "bakedModel", "distanceSq"], "formats": ["<u8", ("<f4", 12), "<f4"],
                       "offsets": [0, 16, 64], "itemsize": 80}),
             np.dtype({"names": ["componentOffset", "bakedModel", "distanceSq", "bufferIndex"],
                       "formats": ["<u8", ("<f4", 12), "<f4", "<u4"], "offsets": [0, 16, 64, 68], "itemsize": 80})),
}


@pytest.mark.parametrize("n", [3_000, 20_000, 300_000])  # in-LDS publish / publish / device pack + one copy
@pytest.mark.parametrize("kind", ["packed", "simd"])
def test_records_in_the_engines_own_struct_layout(oracle, n, kind):
    """gv_pool_set_record_layout: the results of a pool arrive as an array of the caller's UnsortedMesh / SortedMesh
    structs (componentOffset = slot * component size, bakedModel, distanceSq, bufferIndex, padding zero) — the same
    values, bit for bit, as the three-array fetch; sorted, batched, with ready counts, and removable again."""
    from garden_amd.lib import GpuVisibility, GV_E_ARG, GV_E_STATE
    unsorted_dt, sorted_dt = RECORD_DTYPES[kind]
    sc = scene.flat_scene(n, seed=n + 5)
    main = scene.main_camera_view()
    shadow = scene.cascade_view(index=0)
    component = int(sc.meshes.dtype.itemsize)
    ready = np.random.default_rng(n).choice(np.array([0, 1, 1, 3], np.uint32), n)
    with GpuVisibility(device=0) as vis:
        vis.bind_transforms(sc.transforms, sc.entity_to_transform)
        vis.bind_pool(0, sc.meshes)
        vis.bind_pool(1, sc.meshes)
        vis.hierarchy_rebuild()
        # reference: three arrays
        vis.cull(0, [main, shadow])
        vis.sort(0, descending=False, pool_id=0)
        soa = [vis.fetch(v, write_back=False, occupancy=n, pool_id=0, order="raw") for v in range(2)]
        assert soa[0]["draw_count"] > 0 and soa[1]["draw_count"] > 0

        def check(rec, ref, dt, buffer_index, order=None):
            assert rec.shape[0] == ref["draw_count"]
            covered = np.zeros(dt.itemsize, bool)  # every byte no field owns is zero
            for name in dt.names:
                sub, at = dt.fields[name][:2]
                covered[at:at + sub.itemsize] = True
            raw = rec.view(np.uint8).reshape(-1, dt.itemsize)
            assert not raw[:, ~covered].any()
            if order is not None:
                rec = rec[order]
            assert np.array_equal(rec["componentOffset"], ref["visible_idx"].astype(np.uint64) * component)
            assert np.array_equal(rec["bakedModel"].view(np.uint32), ref["baked_model"].view(np.uint32))
            assert np.array_equal(rec["distanceSq"].view(np.uint32), ref["distance_sq"].view(np.uint32))
            if buffer_index is not None:
                assert np.all(rec["bufferIndex"] == buffer_index)

        vis.set_record_layout(0, unsorted_dt, component_stride=component)
        vis.set_record_layout(1, sorted_dt, component_stride=component, buffer_index_value=5)
        for tick in range(2):  # the second tick runs inside a batch
            if tick:
                vis.cull_batch_begin()
            vis.cull(0, [main, shadow])
            vis.cull(1, [main])
            vis.sort(0, descending=False, pool_id=0)
            vis.sort(0, descending=True, pool_id=1)
            got = vis.fetch(0, write_back=False, occupancy=n, pool_id=0, order="raw")
            assert got["draw_count"] == soa[0]["draw_count"] and got["visible_idx"] is None  # delivered as structs instead
            assert np.array_equal(got["is_visible"], soa[0]["is_visible"])
            check(vis.records(0, 0, unsorted_dt), soa[0], unsorted_dt, None)
            check(vis.records(0, 1, unsorted_dt), soa[1], unsorted_dt, None)  # without a fetch of its own first
            back = vis.records(1, 0, sorted_dt)
            assert np.all(np.diff(back["distanceSq"]) <= 0)
            order = np.argsort(back["componentOffset"], kind="stable")
            ref = {k: (soa[0][k][np.argsort(soa[0]["visible_idx"], kind="stable")] if k in ("visible_idx", "baked_model", "distance_sq")
                       else soa[0][k]) for k in soa[0]}
            check(back, ref, sorted_dt, 5, order=order)
        # instanceCount from the records (mesh.cpp:174) when ready counts are bound
        vis.bind_ready(0, ready)
        vis.cull(0, [main])
        got = vis.fetch(0, write_back=False, occupancy=n, pool_id=0)
        rec = vis.records(0, 0, unsorted_dt)
        slots = (rec["componentOffset"] // component).astype(np.int64)
        assert np.all(ready[slots] > 0) and got["instance_count"] == int(ready[slots].sum()) and got["draw_count"] == rec.shape[0]
        vis.bind_ready(0, None)
        # a count-only view has no records; a pool without layout says so; a bad layout is refused
        vis.cull(0, [dict(main, emit_records=0)])
        with pytest.raises(RuntimeError) as e:
            vis.records(0, 0, unsorted_dt)
        assert "count-only" in str(e.value)
        vis.set_record_layout(0, None)
        vis.cull(0, [main, shadow])
        vis.sort(0, descending=False, pool_id=0)
        again = vis.fetch(0, write_back=False, occupancy=n, pool_id=0, order="raw")
        assert np.array_equal(again["visible_idx"], soa[0]["visible_idx"])
        with pytest.raises(RuntimeError) as e:
            vis.records(0, 0, unsorted_dt)
        assert "no record layout" in str(e.value)
        for bad in (np.dtype({"names": ["componentOffset", "bakedModel", "distanceSq"], "formats": ["<u8", ("<f4", 12), "<f4"],
                              "offsets": [0, 8, 56], "itemsize": 72}),     # stride not a multiple of 16
                    np.dtype({"names": ["componentOffset", "bakedModel", "distanceSq"], "formats": ["<u8", ("<f4", 12), "<f4"],
                              "offsets": [0, 4, 56], "itemsize": 64}),     # overlapping fields
                    np.dtype({"names": ["componentOffset", "bakedModel", "distanceSq"], "formats": ["<u8", ("<f4", 12), "<f4"],
                              "offsets": [0, 8, 140], "itemsize": 144})):  # too large
            with pytest.raises(RuntimeError):
                vis.set_record_layout(0, bad, component_stride=component)


@pytest.mark.parametrize("keep_slot_order", [False, True])  # isVisible un-permuted in LDS / copied as it is
@pytest.mark.parametrize("n", [3_000, 20_000, 300_000])  # the sort publishes / publish launch / device pack + one copy
def test_record_target_receives_the_records_in_place(oracle, n, keep_slot_order):
    """gv_pool_set_record_target: the device writes a view's records straight into the caller's own array (the
    engine's combinedMeshes) — the same bytes as the library-buffer delivery, gv_pool_results_records hands back the
    caller's address, slots past draw_count are left alone, an array smaller than occupancy * stride is refused at the
    fetch, the target can be moved and removed."""
    from garden_amd.lib import GpuVisibility
    import ctypes as C
    unsorted_dt, _ = RECORD_DTYPES["packed"]
    sc = scene.flat_scene(n, seed=n + 11)
    main = scene.main_camera_view()
    shadow = scene.cascade_view(index=0)
    component = int(sc.meshes.dtype.itemsize)
    with GpuVisibility(device=0, keep_slot_order=keep_slot_order) as vis:
        vis.bind_transforms(sc.transforms, sc.entity_to_transform)
        vis.bind_pool(0, sc.meshes)
        vis.hierarchy_rebuild()
        vis.set_record_layout(0, unsorted_dt, component_stride=component)
        vis.cull(0, [main, shadow])
        vis.sort(0, descending=False, pool_id=0)
        vis.sort(1, descending=False, pool_id=0)
        first = vis.fetch(0, write_back=False, occupancy=n, pool_id=0)
        ref = [vis.records(0, v, unsorted_dt) for v in range(2)]
        assert ref[0].shape[0] > 0 and ref[1].shape[0] > 0
        # the main pass's isVisible bytes (pool-slot order) say what the records say
        seen = np.zeros(n, np.uint8)
        seen[(ref[0]["componentOffset"] // component).astype(np.int64)] = 1
        assert np.array_equal(first["is_visible"], seen)
        assert np.all(np.diff(ref[0]["distanceSq"]) >= 0)

        def address(pool_id, view):
            ptr, count = C.c_void_p(), C.c_uint32()
            vis._check(vis.lib.gv_pool_results_records(vis.ctx, pool_id, view, C.byref(ptr), C.byref(count)))
            return ptr.value, count.value

        targets = [np.full(n, 0xAB, np.uint8).repeat(unsorted_dt.itemsize).view(unsorted_dt) for _ in range(2)]
        for tick in range(3):  # the same arrays every frame, as the engine's vectors are
            for v in range(2):
                vis.set_record_target(0, v, targets[v])
            vis.cull(0, [main, shadow])
            vis.sort(0, descending=False, pool_id=0)
            vis.sort(1, descending=False, pool_id=0)
            got = vis.fetch(0, write_back=False, occupancy=n, pool_id=0)
            assert got["draw_count"] == ref[0].shape[0] and np.array_equal(got["is_visible"], seen)
            for v in range(2):
                at, count = address(0, v)
                assert at == targets[v].ctypes.data and count == ref[v].shape[0]
                assert np.array_equal(targets[v][:count].view(np.uint8), ref[v].view(np.uint8))
                assert np.all(targets[v][count:].view(np.uint8) == 0xAB)  # nothing past the records is touched
                targets[v][:count].view(np.uint8)[:] = 0xAB
        # instance-count prefix sums read the records where they were delivered
        bases = vis.instance_bases(0, 0)
        assert bases.shape[0] == ref[0].shape[0] + 1 and bases[-1] == ref[0].shape[0]
        # moved to another (larger) array; view 1 back to the library's buffer
        moved = np.zeros(n + 100, unsorted_dt)
        vis.set_record_target(0, 0, moved)
        vis.set_record_target(0, 1, None)
        vis.cull(0, [main, shadow])
        vis.sort(0, descending=False, pool_id=0)
        vis.sort(1, descending=False, pool_id=0)
        vis.fetch(0, write_back=False, occupancy=n, pool_id=0)
        at, count = address(0, 0)
        assert at == moved.ctypes.data and np.array_equal(moved[:count].view(np.uint8), ref[0].view(np.uint8))
        at1, count1 = address(0, 1)
        assert at1 != targets[1].ctypes.data and np.array_equal(vis.records(0, 1, unsorted_dt).view(np.uint8), ref[1].view(np.uint8))
        assert np.all(targets[1].view(np.uint8) == 0xAB)
        # too small for the pool: refused at the fetch, nothing written
        small = np.full((n - 1) * unsorted_dt.itemsize, 0xCD, np.uint8).view(unsorted_dt)
        vis.set_record_target(0, 0, small)
        vis.cull(0, [main])
        with pytest.raises(RuntimeError) as e:
            vis.fetch(0, write_back=False, occupancy=n, pool_id=0)
        assert "record target" in str(e.value)
        assert np.all(small.view(np.uint8) == 0xCD)
        vis.set_record_target(0, 0, None)
        vis.cull(0, [main])
        vis.sort(0, descending=False, pool_id=0)
        vis.fetch(0, write_back=False, occupancy=n, pool_id=0)
        assert np.array_equal(vis.records(0, 0, unsorted_dt).view(np.uint8), ref[0].view(np.uint8))
        # misaligned arrays are refused when they are set
        with pytest.raises(RuntimeError):
            vis._check(vis.lib.gv_pool_set_record_target(vis.ctx, 0, 0, moved.ctypes.data + 8, 4096))


def test_rg16f_conversion_on_every_half_and_its_float_neighbours(oracle):
    """The device's directed float -> binary16 conversion (hardware nearest + one corrective step) against the oracle's
    integer one (pinned on numpy's float16 table in the CPU tier): a depth image whose 2x2 blocks are constant makes
    level 1 of the RG16F pyramid the pair (toward -inf, toward +inf) of each value — every finite half, the floats just
    above and below each (incl. across 0, the subnormal range and both overflow edges), infinities, NaNs with payloads."""
    from garden_amd.lib import GpuVisibility
    h = np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float32)
    h = h[np.isfinite(h)]
    rng = np.random.default_rng(3)
    extra = np.array([np.inf, -np.inf, 65519.99, 65520.0, 65536.0, 1e30, -1e30, 3e38, -3e38, 1e-45, -1e-45, 1e-40, 2.9e-8, 2.99e-8, -2.99e-8],
                     dtype=np.float32)
    nans = np.array([0x7FC00000, 0xFFC00000, 0x7F800001, 0xFFBFFFFF, 0x7FC12345], dtype=np.uint32).view(np.float32)
    rand = rng.integers(0, 2**32, 300_000, dtype=np.uint64).astype(np.uint32).view(np.float32)  # any bit pattern
    vals = np.concatenate([h, np.nextafter(h, np.float32(np.inf)), np.nextafter(h, np.float32(-np.inf)), extra, nans, rand]).astype(np.float32)
    width = 2048  # blocks per row = 1024
    rows = -(-vals.size // (width // 2))
    padded = np.zeros(rows * (width // 2), np.float32)
    padded[:vals.size] = vals
    depth = np.repeat(np.repeat(padded.reshape(rows, width // 2), 2, axis=0), 2, axis=1)
    assert depth.shape == (2 * rows, width)
    with GpuVisibility(device=0, hiz_rg16f=True) as vis:
        vis.hiz_build(depth)
        got = vis.hiz_read_level(1, width // 2, rows)
    exp = oracle.Hiz(depth, rg16f=True).level(1)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    lib = oracle.load()
    sample = np.concatenate([np.arange(0, vals.size, 97), np.arange(h.size * 3, h.size * 3 + extra.size + nans.size)])
    flat = got.reshape(-1, 2)
    for i in sample:  # and the level really is the per-value conversion
        lo, hi = lib.gvo_half_to_float(lib.gvo_half_directed(float(vals[i]), 0)), lib.gvo_half_to_float(lib.gvo_half_directed(float(vals[i]), 1))
        assert (flat[i, 0] == lo or (np.isnan(lo) and np.isnan(flat[i, 0]))) and (flat[i, 1] == hi or (np.isnan(hi) and np.isnan(flat[i, 1])))


@pytest.mark.parametrize("n", [6_000, 200_000])  # published small pool / large pool (chunked prefix, radix sort)
def test_instance_bases_are_the_reference_draw_loops_fetch_adds(oracle, n):
    """gv_pool_results_instance_bases: bases[k] = what `instanceCount.fetch_add(getInstancesAsync(view))` returns for draw k
    in the reference's single-threaded draw loop (mesh.cpp:617-631) — the running sum of the records' ready counts in record
    order (sorted order after gv_sort), total = instance_count; identity without ready counts; same from the three arrays
    and from the struct records; a count-only view has none."""
    from garden_amd.lib import GpuVisibility
    sc = scene.flat_scene(n, seed=n + 1)
    main = scene.main_camera_view()
    ready = np.random.default_rng(n).choice(np.array([0, 1, 1, 2, 5], np.uint32), n)
    component = int(sc.meshes.dtype.itemsize)
    with GpuVisibility(device=0) as vis:
        vis.bind_transforms(sc.transforms, sc.entity_to_transform)
        vis.bind_pool(0, sc.meshes)
        vis.hierarchy_rebuild()
        vis.cull(0, [main])
        got = vis.fetch(0, write_back=False, occupancy=n, pool_id=0, order="raw")
        bases = vis.instance_bases(0, 0)
        assert bases.shape[0] == got["draw_count"] + 1 and np.array_equal(bases, np.arange(got["draw_count"] + 1, dtype=np.uint32))
        vis.bind_ready(0, ready)
        for layout in (None, RECORD_DTYPES["packed"][0]):
            vis.set_record_layout(0, layout, component_stride=component)
            vis.cull(0, [main])
            vis.sort(0, descending=True, pool_id=0)
            bases = vis.instance_bases(0, 0)  # before any fetch: it fetches
            got = vis.fetch(0, write_back=False, occupancy=n, pool_id=0, order="raw")
            if layout is None:
                slots, dist = got["visible_idx"].astype(np.int64), got["distance_sq"]
            else:
                rec = vis.records(0, 0, layout)
                slots, dist = (rec["componentOffset"] // component).astype(np.int64), rec["distanceSq"]
            assert slots.shape[0] == got["draw_count"] > 0 and np.all(np.diff(dist) <= 0)
            counts = ready[slots]
            assert np.all(counts > 0)
            expect = np.concatenate([[0], np.cumsum(counts, dtype=np.uint64)]).astype(np.uint32)
            assert np.array_equal(bases, expect) and int(bases[-1]) == got["instance_count"]
            exp = oracle.prepare_meshes(sc.meshes.copy(), sc.transforms, sc.entity_to_transform, main, ready=ready)
            assert exp["instance_count"] == got["instance_count"] and exp["draw_count"] == got["draw_count"]
        vis.cull(0, [dict(main, emit_records=0)])
        with pytest.raises(RuntimeError) as e:
            vis.instance_bases(0, 0)
        assert "count-only" in str(e.value)


@pytest.mark.parametrize("n", [1000, 70_003, 400_000])
@pytest.mark.parametrize("keep_slot_order", [False, True])
def test_mask_shard_is_the_visible_list_as_bits(oracle, n, keep_slot_order):
    """gv_results_copy_mask_device: [draw_count, one bit per MIRROR entry] — through gv_pool_mirror_slots the same set as the
    index list (identity table with GV_CONFIG_KEEP_SLOT_ORDER), zero bits elsewhere, for the ordinary cull launch (ballot words
    copied) and the one-launch cull + emit of a small pool (built from the isVisible bytes), main and shadow views; and
    garden_amd.multi's CPU restatement of the encoding agrees."""
    import torch
    from garden_amd.lib import GpuVisibility
    from garden_amd.multi import expand_mask_rows, mask_words, pack_mask_shard
    sc = scene.flat_scene(n, seed=n)
    main, shadow = scene.main_camera_view(), scene.cascade_view(index=0, size=4000.0)
    words = mask_words(n) + 3  # more words than needed: the rest must come back zero
    with GpuVisibility(device=0, keep_slot_order=keep_slot_order) as vis:
        vis.bind_transforms(sc.transforms, sc.entity_to_transform)
        vis.bind_pool(0, sc.meshes)
        vis.hierarchy_rebuild()
        table = vis.mirror_slots(0, n)
        assert np.array_equal(np.sort(table), np.arange(n)) and (not keep_slot_order or np.array_equal(table, np.arange(n)))
        shard = torch.full((1 + words,), -1, dtype=torch.int32, device="cuda:0")  # poison
        torch.cuda.synchronize()  # (the fill runs on torch's stream; the library's stream is non-blocking: no implicit order between them)
        for views, vi in (([main], 0), ([main, shadow], 1), ([main, shadow], 0)):
            vis.cull(0, views)
            torch.cuda.synchronize()
            vis.copy_mask_device(vi, shard.data_ptr(), words)
            vis.wait()
            exp = oracle.prepare_meshes(sc.meshes.copy(), sc.transforms, sc.entity_to_transform, views[vi])
            assert exp["draw_count"] > 0
            slots, counts = expand_mask_rows(shard.view(1, -1), n, entry_tables=[table], index_bases=[0])
            assert counts.tolist() == [exp["draw_count"]] and np.array_equal(slots, np.sort(exp["visible_idx"]).astype(np.int64))
            inverse = np.empty(n, np.int64)
            inverse[table] = np.arange(n)
            ref = pack_mask_shard(inverse[exp["visible_idx"]], (words - 0) * 32)[:1 + words]
            assert torch.equal(shard.cpu(), ref)
        if n <= 32768:  # a shadow view alone over a small pool takes the one-launch cull + emit: no ballot words, no bytes
            vis.cull(0, [shadow])
            with pytest.raises(RuntimeError) as e:
                vis.copy_mask_device(0, shard.data_ptr(), words)
            assert "neither" in str(e.value)


@pytest.mark.parametrize("bounds", [False, True])
def test_is_visible_bytes_when_pools_of_different_sizes_share_a_context(oracle, bounds):
    """Round 3: the emit kernel leaves an empty chunk alone when its isVisible bytes are known to be zeros (a flag per quarter
    chunk). One context, pools of different sizes and contents bound in turn (what the tiles of a partitioned world do), sparse
    and dense and empty views, count-only frames in between (there the cull writes the bytes itself): every byte is right."""
    from garden_amd.lib import GpuVisibility
    away = scene.main_camera_view(camera_position=(1e7, 1e7, 1e7))  # nothing visible
    with GpuVisibility(device=0, block_bounds=bounds) as vis:
        for n, seed in [(300_000, 1), (70_000, 2), (299_000, 3), (150_001, 4), (4097, 5), (300_000, 6), (40_000, 7)]:
            sc = scene.flat_scene(n, seed=seed)
            for view in (scene.main_camera_view(seed=seed), away, scene.main_camera_view(seed=seed + 100), dict(scene.main_camera_view(seed=seed), emit_records=0),
                         scene.cascade_view(size=900.0, depth=4000.0, index=-1), away):
                view = dict(view, shadow_pass=-1)
                for got, got_vis, exp, exp_vis in run_both(vis, oracle, sc, [view]):
                    assert got["draw_count"] == exp["draw_count"]
                    assert np.array_equal(got_vis, exp_vis), (n, seed)
                    assert np.array_equal(got["is_visible"], exp_vis)


def test_changes_inside_a_batch_do_not_reach_the_culls_recorded_before_them(oracle):
    """gv_cull_batch_begin only RECORDS the culls of engine-sized pools; they are launched at the first read. A dirty mark of
    the transform pool (or a re-bind of a recorded pool) in between must not change what a recorded cull sees, nor let it run on
    an occupancy its buffers were not sized for: such a call launches the recorded culls first."""
    from garden_amd.lib import GpuVisibility, GV_DIRTY_TRANSFORM
    n = 12_000
    sc = scene.flat_scene(n, seed=5)
    twin = sc.meshes.copy()  # a second mesh system over the same entities
    view = scene.main_camera_view()
    with GpuVisibility(device=0) as vis:
        vis.bind_transforms(sc.transforms, sc.entity_to_transform)
        vis.bind_pool(0, sc.meshes)
        vis.bind_pool(1, twin)
        vis.hierarchy_rebuild()
        before = sc.transforms.copy()
        m0 = sc.meshes.copy()
        exp0 = oracle.prepare_meshes(m0, before, sc.entity_to_transform, view)
        vis.cull_batch_begin()
        vis.cull(0, [view])                                   # recorded
        sc.transforms["position"][:, 0] += np.float32(900.0)  # the world moves ...
        vis.mark_dirty(GV_DIRTY_TRANSFORM, 0, n)              # ... and says so: pool 0's cull runs now, on the old world
        vis.cull(1, [view])                                   # recorded, sees the new world
        got1 = vis.fetch(0, write_back=False, occupancy=n, pool_id=1)
        got0 = vis.fetch(0, write_back=False, occupancy=n, pool_id=0)
        m1 = twin.copy()
        exp1 = oracle.prepare_meshes(m1, sc.transforms, sc.entity_to_transform, view)
        assert exp0["draw_count"] != exp1["draw_count"]
        for got, exp, m in ((got0, exp0, m0), (got1, exp1, m1)):
            assert got["draw_count"] == exp["draw_count"]
            assert np.array_equal(got["visible_idx"], exp["visible_idx"])
            assert np.array_equal(got["baked_model"].view(np.uint32), exp["baked_model"].view(np.uint32))
            assert np.array_equal(got["is_visible"], m["isVisible"])
        # a recorded pool re-bound smaller inside the batch: its recorded cull runs on the pool it was recorded for
        small = scene.flat_scene(3_000, seed=6)
        vis.cull_batch_begin()
        vis.cull(0, [view])
        vis.bind_transforms(small.transforms, small.entity_to_transform)
        vis.bind_pool(0, small.meshes)
        vis.cull(0, [view])
        got = vis.fetch(0, write_back=False, occupancy=small.count, pool_id=0)
        ms = small.meshes.copy()
        exps = oracle.prepare_meshes(ms, small.transforms, small.entity_to_transform, view)
        assert got["draw_count"] == exps["draw_count"] and np.array_equal(got["visible_idx"], exps["visible_idx"])
        assert np.array_equal(got["is_visible"], ms["isVisible"])


def test_thousands_of_scattered_dirty_marks_on_a_slot_order_mirror(gpu_slot_order, oracle):
    """A frame that moves thousands of scattered entities marks thousands of ranges; a mirror in slot order sends them as ONE
    scattered packet (ranged copies would be five small copies per range), a few ranges still as plain copies."""
    gpu = gpu_slot_order
    n = 200_000
    sc = scene.flat_scene(n, seed=9)
    view = scene.main_camera_view()
    run_both(gpu, oracle, sc, [view])
    rng = np.random.default_rng(3)
    for count in (5, 3000):
        moved = np.sort(rng.choice(n // 3, count, replace=False)) * 3  # isolated slots: `count` ranges
        sc.transforms["position"][moved, :3] += rng.uniform(-300, 300, (count, 3)).astype(np.float32)
        sc.meshes["aabbMax"][moved, :3] *= np.float32(1.5)
        for s in moved:
            gpu.mark_dirty(0, int(s), 1)
            gpu.mark_dirty(2, int(s), 1, pool_id=0)
        gpu.cull(0, [view])
        sc.meshes["isVisible"] = 7
        got = gpu.fetch(0, write_back=True, occupancy=n)
        got_vis = sc.meshes["isVisible"].copy()
        exp = oracle.prepare_meshes(sc.meshes, sc.transforms, sc.entity_to_transform, view)
        assert_same(got, got_vis, exp, sc.meshes["isVisible"].copy())


@pytest.mark.parametrize("case", ["grow", "replace", "early_free"])
def test_record_target_lifetime_through_the_bare_c_abi(case):
    """gv_pool_set_record_target without the shim (tools/record_target_probe.py, its own process: the array is an anonymous
    mapping the script unmaps itself): `grow` — clear the target, free the array, set a larger one, every frame; `replace` — set
    the new array while the old one is still mapped, free it afterwards; `early_free` — the array is unmapped while it is still
    the target: the next gv_pool_set_record_target reports GV_E_STATE every time (the new target is in place, the frames go on)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "record_target_probe.py"), case, "8"], capture_output=True, text=True,
                       timeout=600, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    assert f"{case}: 8 frames ok" in p.stdout
    if case == "early_free":
        assert "lost registration reported 8 times" in p.stdout


def test_an_occlusion_views_results_in_every_order_of_calls(oracle):
    """Every order of calls an application can make around the cull of a single use_hiz view gives the records, isVisible bytes
    and counts of the plain path: rebuild -> fetch, fetch at once, a new depth image in between (the cull saw the old pyramid),
    dirty marks and the next cull in between, a sort, a second pool culled in between, a view array of two, and a context destroyed
    right after a cull. (Written in round 3 for an emit that was held back until the next pyramid build; that mechanism was
    withdrawn in round 4 — gv_cull enqueues everything itself again — and the last block checks exactly that: a device-side
    consumer that cached the pointers of gv_results_device sees each frame's list without calling into the library again.)"""
    import torch
    from garden_amd.lib import GpuVisibility
    n = 280_000
    sc = scene.flat_scene(n)
    other = scene.flat_scene(90_000, seed=5)
    view = scene.main_camera_view(use_hiz=1)
    depth = scene.synthetic_depth(1024, 512)
    depth2 = np.maximum(depth, np.float32(0.4))
    rng = np.random.Generator(np.random.PCG64(8))

    def expect(meshes, tr, e2t, v, hz):
        m2 = meshes.copy()
        exp = oracle.prepare_meshes(m2, tr, e2t, v, hiz=hz)
        return exp, m2

    def same(got, exp, m2, sorted_by_distance=False):
        if sorted_by_distance:
            o = np.argsort(exp["distance_sq"], kind="stable")
            assert np.array_equal(np.sort(got["visible_idx"]), np.sort(exp["visible_idx"]))
            assert np.array_equal(got["distance_sq"], exp["distance_sq"][o])
        else:
            assert np.array_equal(got["visible_idx"], np.sort(exp["visible_idx"]))
            o = np.argsort(exp["visible_idx"], kind="stable")
            assert np.array_equal(got["baked_model"].view(np.uint32), exp["baked_model"][o].view(np.uint32))
        assert np.array_equal(got["is_visible"], m2["isVisible"])

    with GpuVisibility(device=0) as vis:
        vis.hiz_build(depth)
        hz, hz2 = oracle.Hiz(depth), oracle.Hiz(depth2)
        vis.bind_transforms(sc.transforms, sc.entity_to_transform)
        vis.bind_pool(0, sc.meshes)
        vis.hierarchy_rebuild()
        exp, m2 = expect(sc.meshes, sc.transforms, sc.entity_to_transform, view, hz)
        # frames of an engine: cull, next frame's pyramid, cull, ... and a fetch at the end
        for _ in range(3):
            vis.cull(0, [view])
            vis.hiz_rebuild()
        same(vis.fetch(0, write_back=False, occupancy=n), exp, m2)
        # fetch straight after the cull
        vis.cull(0, [view])
        same(vis.fetch(0, write_back=False, occupancy=n), exp, m2)
        # a NEW depth image between the cull and the fetch: the cull saw the old pyramid, the held-back emit only compacts
        vis.cull(0, [view])
        vis.hiz_build(depth2)
        same(vis.fetch(0, write_back=False, occupancy=n), exp, m2)
        exp2, m22 = expect(sc.meshes, sc.transforms, sc.entity_to_transform, view, hz2)
        vis.cull(0, [view])
        same(vis.fetch(0, write_back=False, occupancy=n), exp2, m22)
        # dirty marks and the next cull while an emit is held
        vis.cull(0, [view])
        for s in rng.integers(0, n, 40):
            sc.transforms["position"][s, :3] += np.float32(7)
            vis.mark_dirty(0, int(s), 1)
        vis.cull(0, [view])
        vis.hiz_rebuild()
        exp3, m23 = expect(sc.meshes, sc.transforms, sc.entity_to_transform, view, hz2)
        same(vis.fetch(0, write_back=False, occupancy=n), exp3, m23)
        # a sort of the held-back records
        vis.cull(0, [view])
        vis.sort(0)
        same(vis.fetch(0, write_back=False, occupancy=n, order="raw"), exp3, m23, sorted_by_distance=True)
        # another pool culled in between, device-side copies of the first one's list afterwards
        vis.bind_pool(1, other.meshes)
        vis.cull(0, [view])
        vis.cull(1, [dict(view, use_hiz=0)])
        got1 = vis.fetch(0, write_back=False, occupancy=other.count, pool_id=1)
        e1m = other.meshes.copy()
        e1 = oracle.prepare_meshes(e1m, sc.transforms, sc.entity_to_transform, dict(view, use_hiz=0))
        assert np.array_equal(got1["visible_idx"], np.sort(e1["visible_idx"]))
        same(vis.fetch(0, write_back=False, occupancy=n, pool_id=0), exp3, m23)
        vis.cull(0, [view])
        buf = torch.full((n + 1,), -1, dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()  # (the fill runs on torch's stream; the library's stream is non-blocking: no implicit order between them)
        vis.copy_shard_device(0, buf.data_ptr(), n, index_base=0)
        vis.wait()
        host = buf.cpu().numpy()
        assert host[0] == exp3["draw_count"] and np.array_equal(np.sort(host[1:1 + host[0]]), np.sort(exp3["visible_idx"]))
        # two views: never held back
        shadow = scene.cascade_view(index=0, size=4000.0)
        vis.cull(0, [view, shadow])
        vis.hiz_rebuild()
        same(vis.fetch(0, write_back=False, occupancy=n), exp3, m23)
        # a device-side consumer with cached pointers: nothing but gv_cull between the frames
        dres = vis.results_device(0)

        class _Span:
            pass
        span = _Span()
        span.__cuda_array_interface__ = {"shape": (1,), "typestr": "<i4", "data": (int(dres.draw_count), False), "version": 2}
        count = torch.as_tensor(span, device="cuda:0")
        lib_stream = torch.cuda.ExternalStream(vis.stream(), device=torch.device("cuda", 0))
        seen = []
        for frame_view, frame_exp in ((view, exp3), (dict(view, use_hiz=0), None), (view, exp3)):
            vis.cull(0, [frame_view])
            assert vis.results_device(0).draw_count == dres.draw_count  # (same buffers from frame to frame)
            with torch.cuda.stream(lib_stream):
                seen.append(count.clone())
        lib_stream.synchronize()
        e_nohiz = oracle.prepare_meshes(sc.meshes.copy(), sc.transforms, sc.entity_to_transform, dict(view, use_hiz=0))
        assert [int(t.item()) for t in seen] == [exp3["draw_count"], e_nohiz["draw_count"], exp3["draw_count"]]
        vis.cull(0, [view])  # ... and a context destroyed right after a cull
