"""Full-size checks at BASELINE.json's configurations (10^6 / 10^7 entities, 4096^2 pyramid).

Two kinds of evidence per configuration:
  * the CPU oracle (all host cores, seconds) on the same pools -> bit-exact visible set / isVisible / records;
  * size-independent properties that need no oracle: idempotence, count == popcount(isVisible), unique ascending
    indices, Hi-Z result is a subset of the frustum-only result, union of tile shards == the whole pool (the
    cfg5 sharding pattern: a checksum of checksums), sortedness + permutation after gv_sort, MFMA sweep == VALU
    sweep, identity-parent invariance of the visible set.
Scenes are built once per module; each 10M scene is ~1.3 GB of host pools."""
import os

import numpy as np
import pytest

from garden_amd import scene
from garden_amd.lib import GV_SWEEP_MFMA, GV_SWEEP_VALU

pytestmark = pytest.mark.gpu

N_FULL = 10_000_000
HIZ = 4096
THREADS = max(1, os.cpu_count() or 1)


@pytest.fixture(scope="module")
def flat10m():
    return scene.flat_scene(N_FULL)


@pytest.fixture(scope="module")
def hier10m():
    return scene.hierarchy_scene(N_FULL)


def bind(gpu, sc):
    gpu.bind_transforms(sc.transforms, sc.entity_to_transform)
    gpu.bind_pool(0, sc.meshes)
    gpu.hierarchy_rebuild()


def check_self_consistent(got, n):
    idx = got["visible_idx"].astype(np.int64)
    assert got["draw_count"] == idx.shape[0] == got["instance_count"]
    assert np.all(np.diff(idx) > 0), "indices must be unique (and ascending after the slot-order fetch)"
    assert idx.size == 0 or (idx[0] >= 0 and idx[-1] < n)
    vis = got["is_visible"]
    assert int(vis.sum(dtype=np.int64)) == got["draw_count"] and set(np.unique(vis)) <= {0, 1}
    assert np.all(vis[idx] == 1)


def same_records(a, b):
    return (np.array_equal(a["visible_idx"], b["visible_idx"])
            and np.array_equal(a["baked_model"].view(np.uint32), b["baked_model"].view(np.uint32))
            and np.array_equal(a["distance_sq"].view(np.uint32), b["distance_sq"].view(np.uint32)))


def test_cfg2_1m_frustum_only_against_oracle(gpu, oracle):
    sc = scene.flat_scene(1_000_000)
    view = scene.main_camera_view()
    bind(gpu, sc)
    gpu.cull(0, [view])
    got = gpu.fetch(0, write_back=False, occupancy=sc.count)
    check_self_consistent(got, sc.count)
    m2 = sc.meshes.copy()
    exp = oracle.prepare_meshes(m2, sc.transforms, sc.entity_to_transform, view, threads=THREADS)
    o = np.argsort(exp["visible_idx"], kind="stable")
    exp = {k: (v[o] if isinstance(v, np.ndarray) else v) for k, v in exp.items()}
    assert same_records(got, exp) and np.array_equal(got["is_visible"], m2["isVisible"])


def test_cfg3_10m_hiz_properties_and_oracle(gpu, oracle, flat10m):
    sc = flat10m
    depth = scene.synthetic_depth(HIZ, HIZ)
    frustum_only, with_hiz = scene.main_camera_view(use_hiz=0), scene.main_camera_view(use_hiz=1)
    bind(gpu, sc)
    gpu.hiz_build(depth)

    gpu.cull(0, [frustum_only])
    fo = gpu.fetch(0, write_back=False, occupancy=sc.count)
    check_self_consistent(fo, sc.count)

    gpu.hiz_rebuild()
    gpu.cull(0, [with_hiz])
    hz1 = gpu.fetch(0, write_back=False, occupancy=sc.count)
    check_self_consistent(hz1, sc.count)
    # occlusion only ever removes entities
    assert hz1["draw_count"] < fo["draw_count"]
    assert np.all(np.isin(hz1["visible_idx"], fo["visible_idx"], assume_unique=True))
    assert np.all(hz1["is_visible"] <= fo["is_visible"])

    # idempotence: pyramid rebuild + cull again -> the same bits
    gpu.hiz_rebuild()
    gpu.cull(0, [with_hiz])
    hz2 = gpu.fetch(0, write_back=False, occupancy=sc.count)
    assert same_records(hz1, hz2) and np.array_equal(hz1["is_visible"], hz2["is_visible"])

    # the oracle at full size
    m2 = sc.meshes.copy()
    exp = oracle.prepare_meshes(m2, sc.transforms, sc.entity_to_transform, with_hiz, hiz=oracle.Hiz(depth), threads=THREADS)
    o = np.argsort(exp["visible_idx"], kind="stable")
    exp = {k: (v[o] if isinstance(v, np.ndarray) else v) for k, v in exp.items()}
    assert same_records(hz1, exp) and np.array_equal(hz1["is_visible"], m2["isVisible"])


@pytest.mark.parametrize("rg16f", [False, True])
def test_cfg3_10m_against_a_hard_depth_image(oracle, flat10m, rg16f):
    """cfg3's 10 M entities against scene.noise_depth (one occluder per 8 x 8 pixel block, at distances among the entities): the
    coarse-level early exits of the occlusion query decide almost nothing here, so every frustum survivor's answer comes from
    levels 0-2 of the pyramid — level 1 is virtual, i.e. reduced from the depth image on the fly — where cfg3's own walls are
    decided four levels up. Bit for bit against the oracle, fp32 and RG16F pyramids; both exits of the query are used."""
    from garden_amd.lib import GpuVisibility
    sc = flat10m
    depth = scene.noise_depth(HIZ, HIZ)
    view = scene.main_camera_view(use_hiz=1)
    with GpuVisibility(device=0, hiz_rg16f=rg16f, linear_scan=True) as vis:
        bind(vis, sc)
        vis.hiz_build(depth)
        vis.cull(0, [view])
        got = vis.fetch(0, write_back=False, occupancy=sc.count)
    check_self_consistent(got, sc.count)
    m2 = sc.meshes.copy()
    exp = oracle.prepare_meshes(m2, sc.transforms, sc.entity_to_transform, view, hiz=oracle.Hiz(depth, rg16f=rg16f, threads=THREADS), threads=THREADS)
    o = np.argsort(exp["visible_idx"], kind="stable")
    exp = {k: (v[o] if isinstance(v, np.ndarray) else v) for k, v in exp.items()}
    assert same_records(got, exp) and np.array_equal(got["is_visible"], m2["isVisible"])
    frustum = oracle.prepare_meshes(sc.meshes.copy(), sc.transforms, sc.entity_to_transform, dict(view, use_hiz=0), threads=THREADS)["draw_count"]
    # a real mix: the image hides a good part of the frustum's survivors and keeps a good part
    assert 0.05 * frustum < got["draw_count"] < 0.95 * frustum, (got["draw_count"], frustum)


def test_cfg3_10m_against_the_rg16f_pyramid(oracle, flat10m):
    """GV_CONFIG_HIZ_RG16F at BASELINE's size: 10 M entities against the 4096^2 pyramid kept in the reference's image
    format (min rounded toward -inf, max toward +inf; level 1 virtual): the oracle's set bit for bit, a superset of the
    fp32 pyramid's set (never culls what that one keeps), and the rounding does let a few more through."""
    from garden_amd.lib import GpuVisibility
    sc = flat10m
    depth = scene.synthetic_depth(HIZ, HIZ)
    view = scene.main_camera_view(use_hiz=1)
    with GpuVisibility(device=0, hiz_rg16f=True) as vis:
        bind(vis, sc)
        vis.hiz_build(depth)
        vis.cull(0, [view])
        got = vis.fetch(0, write_back=False, occupancy=sc.count)
    check_self_consistent(got, sc.count)
    m2 = sc.meshes.copy()
    exp = oracle.prepare_meshes(m2, sc.transforms, sc.entity_to_transform, view, hiz=oracle.Hiz(depth, rg16f=True, threads=THREADS), threads=THREADS)
    o = np.argsort(exp["visible_idx"], kind="stable")
    exp = {k: (v[o] if isinstance(v, np.ndarray) else v) for k, v in exp.items()}
    assert same_records(got, exp) and np.array_equal(got["is_visible"], m2["isVisible"])
    fp32 = oracle.prepare_meshes(sc.meshes.copy(), sc.transforms, sc.entity_to_transform, view, hiz=oracle.Hiz(depth, threads=THREADS), threads=THREADS)
    assert np.isin(fp32["visible_idx"], got["visible_idx"]).all()
    print(f"rg16f pyramid: {got['draw_count']} visible vs {fp32['draw_count']} with the fp32 pyramid")


def test_cfg5_one_world_dealt_to_eight_ranks_through_the_exchange(oracle, hier10m):
    """cfg5 pattern on one GPU: ONE 10 M hierarchical world -> partition_world(ranks=8): 16 x 16 x 16 Morton-ordered cells dealt to
    8 ranks in rotating rounds (roots by position, descendants follow, ids remapped), each rank's cells culled as ONE pool, its list
    pushed through the library-sized exchange (gv_exchange_visible on a 1-rank RCCL communicator, every travel pattern) with
    the rank's slot -> world-slot table applied on the device — the mapped union must be the whole-world ORACLE set, isVisible
    included; every rank has its share of the view (round 3's octants left half the ranks with nothing to show)."""
    import torch

    from garden_amd.lib import GpuVisibility
    from garden_amd.multi import cell_grid, partition_world
    sc = hier10m
    view = scene.main_camera_view()
    m2 = sc.meshes.copy()
    whole = oracle.prepare_meshes(m2, sc.transforms, sc.entity_to_transform, view, threads=THREADS)
    exp = np.sort(whole["visible_idx"].astype(np.int64))
    part = partition_world(sc, cell_grid(8), ranks=8)
    sizes = np.array([t.count for t in part.tiles], dtype=np.float64)
    assert sizes.sum() == sc.count and sizes.max() / sizes.mean() < 1.1  # eight even shares (trees of 1 000 go with their roots)
    union, counts, is_visible = [], [], np.full(sc.count, 255, np.uint8)
    with GpuVisibility(device=0) as vis:
        for t, tile in enumerate(part.tiles):
            vis.exchange_init(GpuVisibility.exchange_unique_id(), 0, 1)  # (a fresh communicator per "rank": frame 0 predicts nothing)
            vis.bind_transforms(tile.transforms, tile.entity_to_transform)
            vis.bind_pool(0, tile.meshes)
            vis.hierarchy_rebuild()
            vis.set_index_map(0, part.mesh_global[t])
            vis.exchange_set_mode(t % 3)  # all-gather, grouped send/recv, per-root broadcast in turn
            for frame in range(3):  # frame 0 completed by a second exchange, frames 1-2 predicted from the headers
                vis.cull(0, [view])
                sent = vis.exchange_visible(0, index_base=0)
                f = vis.exchange_acquire(sent["frame"])
                assert f["frame"] == frame and f["complete"] and bool(f["cut_ranks"]) == (frame == 0)
            got = vis.fetch(0, write_back=False, occupancy=tile.count)
            assert f["counts"] == [got["draw_count"]] and f["room"][0] >= got["draw_count"]

            class _Span:
                pass
            span = _Span()
            span.__cuda_array_interface__ = {"shape": (f["row_words"],), "typestr": "<i4", "data": (int(f["ptr"]), False), "version": 2}
            vis.wait()  # (the acquire ordered the context's stream behind the rows)
            row = torch.as_tensor(span, device="cuda:0").cpu().numpy().astype(np.int64) & 0xFFFFFFFF
            assert row[0] == got["draw_count"]
            ids = row[1:1 + row[0]]
            # the device-side table gives what the host-side map gives
            assert np.array_equal(np.sort(ids), np.sort(part.to_global(t, got["visible_idx"])))
            union.append(ids)
            counts.append(got["draw_count"])
            is_visible[part.mesh_global[t]] = got["is_visible"]
            vis.exchange_shutdown()
    union = np.sort(np.concatenate(union))
    assert np.array_equal(union, exp) and exp.shape[0] > 100_000
    assert np.array_equal(is_visible, m2["isVisible"])
    counts = np.array(counts, dtype=np.float64)
    assert counts.min() > 0 and counts.max() / counts.mean() <= 1.5, counts


def test_10m_sort_is_sorted_permutation(gpu, flat10m):
    sc = flat10m
    view = scene.main_camera_view()
    bind(gpu, sc)
    gpu.cull(0, [view])
    before = gpu.fetch(0, write_back=False, occupancy=sc.count)
    for descending in (False, True):
        gpu.cull(0, [view])
        gpu.sort(0, descending=descending)
        got = gpu.fetch(0, write_back=False, occupancy=sc.count, order="raw")
        d = got["distance_sq"]
        assert np.all(np.diff(d) <= 0) if descending else np.all(np.diff(d) >= 0)
        o = np.argsort(got["visible_idx"], kind="stable")
        after = dict(visible_idx=got["visible_idx"][o], baked_model=got["baked_model"][o], distance_sq=d[o])
        assert same_records(before, after), "sorting must only permute whole records"


def test_cfg4_10m_hierarchy_sweep_and_cull(gpu, oracle, hier10m):
    sc = hier10m
    view = scene.main_camera_view()
    bind(gpu, sc)
    # MFMA and VALU sweeps write the same bits for every one of the 10M world matrices (checked in 1M blocks)
    gpu.sweep(GV_SWEEP_VALU)
    valu = [gpu.get_world(f, 1_000_000) for f in range(0, N_FULL, 1_000_000)]
    gpu.sweep(GV_SWEEP_MFMA)
    for k, f in enumerate(range(0, N_FULL, 1_000_000)):
        assert np.array_equal(gpu.get_world(f, 1_000_000).view(np.uint32), valu[k].view(np.uint32))
    # a sampled block against the scalar chain walk of the oracle (transform.hpp:197-214)
    first, count = 9_000_000, 200_000
    exp_w = oracle.world_matrices(sc.transforms, sc.entity_to_transform, first, count)
    assert np.array_equal(valu[9][:count].view(np.uint32), exp_w.view(np.uint32))
    del valu

    gpu.cull(0, [view])
    got = gpu.fetch(0, write_back=False, occupancy=sc.count)
    check_self_consistent(got, sc.count)
    m2 = sc.meshes.copy()
    exp = oracle.prepare_meshes(m2, sc.transforms, sc.entity_to_transform, view, threads=THREADS)
    o = np.argsort(exp["visible_idx"], kind="stable")
    exp = {k: (v[o] if isinstance(v, np.ndarray) else v) for k, v in exp.items()}
    assert same_records(got, exp) and np.array_equal(got["is_visible"], m2["isVisible"])

    # the FUSED sweep + cull (bench.py --workload cfg4's default, and its VALU twin) at full size: records, isVisible
    # and world-matrix blocks against the oracle; then the same with the Hi-Z stage in the fused kernel
    from garden_amd.lib import GV_SWEEP_WITH_CULL, GV_SWEEP_WITH_CULL_VALU
    blocks = [(0, 100_000), (4_321_000, 150_000), (N_FULL - 120_000, 120_000)]
    exp_blocks = [oracle.world_matrices(sc.transforms, sc.entity_to_transform, f, c, threads=THREADS) for f, c in blocks]
    for mode in (GV_SWEEP_WITH_CULL, GV_SWEEP_WITH_CULL_VALU):
        gpu.mark_dirty(0, 0, sc.count)  # forget the cache: the fused kernel must produce every matrix itself
        gpu.sweep(mode)
        gpu.cull(0, [view])
        fused = gpu.fetch(0, write_back=False, occupancy=sc.count)
        assert same_records(fused, exp) and np.array_equal(fused["is_visible"], m2["isVisible"])
        for (f, c), e in zip(blocks, exp_blocks):
            assert np.array_equal(gpu.get_world(f, c).view(np.uint32), e.view(np.uint32))
    depth = scene.synthetic_depth(HIZ, HIZ)
    hz_view = scene.main_camera_view(use_hiz=1)
    gpu.hiz_build(depth)
    m3 = sc.meshes.copy()
    exp_hz = oracle.prepare_meshes(m3, sc.transforms, sc.entity_to_transform, hz_view, hiz=oracle.Hiz(depth, threads=THREADS), threads=THREADS)
    o = np.argsort(exp_hz["visible_idx"], kind="stable")
    exp_hz = {k: (v[o] if isinstance(v, np.ndarray) else v) for k, v in exp_hz.items()}
    for mode in (GV_SWEEP_WITH_CULL, GV_SWEEP_WITH_CULL_VALU):
        gpu.sweep(mode)
        gpu.cull(0, [hz_view])
        fused = gpu.fetch(0, write_back=False, occupancy=sc.count)
        assert 0 < fused["draw_count"] < got["draw_count"]
        assert same_records(fused, exp_hz) and np.array_equal(fused["is_visible"], m3["isVisible"])


def test_cfg2_shape_at_10_to_the_8_on_one_gpu(oracle):
    """BASELINE's largest entity count on ONE GPU (the per-GPU share of an 8 x 12.5 M node is an eighth of this): 10^8
    flat entities, frustum-only — the 7.3 GB mirror, chunk totals beyond the self-prefixing emit's limit (scan launch),
    21 M records — against the oracle on all 10^8 entities: visible set, isVisible, and a checksum of the records."""
    import psutil

    from garden_amd.lib import GpuVisibility
    if psutil.virtual_memory().available < 96 * 2 ** 30:
        pytest.skip("needs ~60 GB of free host memory")
    n = 100_000_000
    sc = scene.flat_scene(n)
    view = scene.main_camera_view()
    with GpuVisibility(device=0) as vis:
        bind(vis, sc)
        vis.cull(0, [view])
        got = vis.fetch(0, write_back=False, occupancy=n)
    exp = oracle.prepare_meshes(sc.meshes, sc.transforms, sc.entity_to_transform, view, threads=THREADS)  # isVisible written in place
    assert got["draw_count"] == exp["draw_count"] > 10_000_000
    o = np.argsort(exp["visible_idx"], kind="stable")
    assert np.array_equal(got["visible_idx"], exp["visible_idx"][o])
    assert np.array_equal(got["is_visible"], sc.meshes["isVisible"])
    # records: bit patterns summed per column (order-independent checksum) + a directly compared slice
    assert np.array_equal(got["baked_model"].view(np.uint32).sum(axis=0, dtype=np.uint64),
                          exp["baked_model"].view(np.uint32).sum(axis=0, dtype=np.uint64))
    assert np.array_equal(got["baked_model"][:500_000].view(np.uint32), exp["baked_model"][o[:500_000]].view(np.uint32))
    assert np.array_equal(got["distance_sq"].view(np.uint32).sum(dtype=np.uint64), exp["distance_sq"].view(np.uint32).sum(dtype=np.uint64))


def test_identity_parent_leaves_the_visible_set_unchanged(gpu):
    """M_parent = I: fma(1, a, fma(0, b, ...)) returns a, so hanging every entity under an identity parent must not
    move a single entity across a frustum plane (only -0 -> +0 can change in the matrices)."""
    n = 1_000_000
    sc = scene.flat_scene(n)
    view = scene.main_camera_view()
    bind(gpu, sc)
    gpu.cull(0, [view])
    flat = gpu.fetch(0, write_back=False, occupancy=n)

    tr = np.concatenate([sc.transforms, np.zeros(1, sc.transforms.dtype)])
    ident = n  # slot of the identity transform
    ident_entity = int(sc.transforms["entity"].max()) + 1
    tr["entity"][ident] = ident_entity
    tr["rotation"][ident] = (0, 0, 0, 1)
    tr["scale"][ident, :3] = (1, 1, 1)
    tr["selfActive"][ident] = tr["ancestorsActive"][ident] = tr["modelWithAncestors"][ident] = 1
    live = tr["entity"][:n] != 0
    tr["parent"][:n][live] = ident_entity
    e2t = np.full(max(sc.entity_to_transform.shape[0], ident_entity + 1), 0xFFFFFFFF, np.uint32)
    e2t[:sc.entity_to_transform.shape[0]] = sc.entity_to_transform
    e2t[ident_entity] = ident
    gpu.bind_transforms(tr, e2t)
    gpu.bind_pool(0, sc.meshes)
    gpu.hierarchy_rebuild()
    gpu.cull(0, [view])
    hung = gpu.fetch(0, write_back=False, occupancy=n)
    assert np.array_equal(flat["visible_idx"], hung["visible_idx"])
    assert np.array_equal(flat["is_visible"], hung["is_visible"])
    assert np.array_equal(flat["baked_model"] + 0.0, hung["baked_model"] + 0.0)  # equal up to the sign of zero


def test_moved_transforms_beyond_a_smaller_paired_mesh_pool_keep_block_bounds_current(oracle):
    """the transform pool has MORE slots than the exactly paired mesh pool (its first million slots are free), the
    mirror is in spatial order, and a stretch of transform slots ABOVE the mesh pool's occupancy — long enough for the device-side
    gather of dirty AoS ranges, short enough to count as "a few" — moves every frame. Those slots map to mirror entries INSIDE the
    mesh pool: the blocks holding them must be flagged so that the next cull re-derives their boxes and emit seeds
    (mark_dirty_blocks_kernel bounded the slot by the mesh pool's occupancy and dropped them). The moved entities jump from behind
    the camera to right in front of it: with stale boxes their workgroups would be skipped. Against the oracle every frame."""
    from garden_amd.lib import GV_DIRTY_TRANSFORM, GpuVisibility
    n_mesh, shift = 8_600_000, 1_000_000
    base = scene.flat_scene(n_mesh)
    tr = np.zeros(n_mesh + shift, dtype=base.transforms.dtype)  # slots [0, shift) stay free
    tr[shift:] = base.transforms
    e2t = np.asarray(base.entity_to_transform, dtype=np.uint32).copy()
    live = e2t != 0xFFFFFFFF
    e2t[live] += shift
    meshes = base.meshes
    view = scene.main_camera_view()
    first, count = n_mesh + 200_000, 3000  # transform slots above the mesh pool's occupancy
    with GpuVisibility(device=0) as vis:  # default configuration: block bounds for pools of this size
        vis.bind_transforms(tr, e2t)
        vis.bind_pool(0, meshes)
        vis.hierarchy_rebuild()
        fwd = got = None
        for frame in range(8):
            if frame >= 1:
                # onto a line of sight, a little farther each frame (frame 1: from wherever they were, mostly out of view)
                rows = tr[first:first + count]
                if fwd is None:  # a direction inside the frustum: towards an entity the first frame saw (the camera sits at the origin)
                    p0 = tr["position"][int(got["visible_idx"][got["draw_count"] // 2]) + shift, :3].astype(np.float64)
                    fwd = (p0 / np.linalg.norm(p0)).astype(np.float32)
                rows["position"][:, :3] = fwd * np.float32(300.0 + 40.0 * frame) + np.linspace(-20, 20, count, dtype=np.float32)[:, None]
                vis.mark_dirty(GV_DIRTY_TRANSFORM, first, count)
            vis.cull(0, [view])
            got = vis.fetch(0, write_back=False, occupancy=n_mesh)
            m2 = meshes.copy()
            exp = oracle.prepare_meshes(m2, tr, e2t, view, threads=THREADS)
            o = np.argsort(exp["visible_idx"], kind="stable")
            assert np.array_equal(got["visible_idx"], exp["visible_idx"][o]), frame
            assert np.array_equal(got["baked_model"].view(np.uint32), exp["baked_model"][o].view(np.uint32)), frame
            assert np.array_equal(got["is_visible"], m2["isVisible"]), frame
            if frame >= 1:
                moved = np.arange(first - shift, first - shift + count)
                assert np.isin(moved, got["visible_idx"]).mean() > 0.5, frame  # they really are in view now
        st = vis.stats()
        assert st["bounds_blocks_total"] > 0  # the culls did run with block bounds
