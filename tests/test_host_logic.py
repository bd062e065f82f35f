"""Host-side logic that needs no GPU: pool layouts, scene generator, tile partition, the all-gatherv exchange
over gloo with world_size 2."""
import os
import sys
import socket

import numpy as np
import pytest

from garden_amd import scene
from garden_amd.pools import GV_NONE, MESH_DTYPE, TRANSFORM_DTYPE, derived_mesh_dtype


def test_pool_layouts_match_reference_structs():
    assert MESH_DTYPE.itemsize == 48 and TRANSFORM_DTYPE.itemsize == 80  # mesh.hpp:45-55, transform.hpp:31-61
    assert MESH_DTYPE.fields["aabbMin"][1] == 16 and MESH_DTYPE.fields["isEnabled"][1] == 14
    assert TRANSFORM_DTYPE.fields["rotation"][1] == 48 and TRANSFORM_DTYPE.fields["selfActive"][1] == 72
    assert derived_mesh_dtype(32).itemsize == 80


def test_scene_is_deterministic_and_well_formed():
    a, b = scene.flat_scene(5000), scene.flat_scene(5000)
    assert a.meshes.tobytes() == b.meshes.tobytes() and a.transforms.tobytes() == b.transforms.tobytes()
    q = a.transforms["rotation"]
    assert np.allclose(np.sum(q * q, axis=1), 1.0, atol=1e-6)
    live = a.transforms["entity"] != 0
    assert 0.005 < 1 - live.mean() < 0.02 and 0.005 < (a.meshes["isEnabled"] == 0).mean() < 0.02
    assert np.array_equal(a.entity_to_transform[a.transforms["entity"][live]], np.nonzero(live)[0])


def test_hierarchy_scene_levels_and_active_flags():
    sc = scene.hierarchy_scene(11110, depth=4, fanout=10, defects=True)
    t = sc.transforms
    depth = np.zeros(sc.count, dtype=np.int32)
    for s in range(sc.count):
        p = t["parent"][s]
        if p:
            ps = sc.entity_to_transform[p]
            assert ps != GV_NONE and ps < s  # parents precede children (level order)
            depth[s] = depth[ps] + 1
            exp = t["selfActive"][ps] & t["ancestorsActive"][ps]
            assert t["ancestorsActive"][s] == exp  # setActive propagation (transform.cpp:75-127)
    assert depth.max() == 3 and (depth == 0).sum() == 10 or (depth == 0).sum() >= 10


def test_tile_partition_covers_everything_once():
    from garden_amd.multi import tile_of_positions
    rng = np.random.default_rng(0)
    pos = rng.uniform(-50, 50, (10000, 3)).astype(np.float32)
    for grid in ([2, 1, 1], [2, 2, 1], [2, 2, 2]):
        t = tile_of_positions(pos, 100.0, grid)
        n = grid[0] * grid[1] * grid[2]
        assert t.min() == 0 and t.max() == n - 1
        assert np.bincount(t, minlength=n).sum() == 10000 and np.bincount(t, minlength=n).min() > 10000 / n * 0.8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _exchange_worker(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from garden_amd.multi import allgatherv_indices
    from oracle import oracle_py
    # each rank culls its own tile on the CPU oracle (stand-in for the per-GPU cull), then exchanges
    n_local = 3000
    sc = scene.flat_scene(n_local, seed=scene.SEED + rank)
    view = scene.main_camera_view()
    r = oracle_py.prepare_meshes(sc.meshes, sc.transforms, sc.entity_to_transform, view)
    buf = torch.zeros(n_local, dtype=torch.int32)
    k = r["draw_count"]
    buf[:k] = torch.from_numpy((r["visible_idx"].astype(np.int64) + rank * n_local).astype(np.int32))
    gathered, counts = allgatherv_indices(buf, k, dist)
    ret[rank] = (gathered.numpy().copy(), counts.numpy().copy(), k)
    dist.barrier()
    dist.destroy_process_group()


def test_allgatherv_world_size_2_gloo(oracle):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_exchange_worker, args=(world, port, ret), nprocs=world, join=True)
    # single-process expectation: concatenation of the per-tile lists in rank order, global indices
    exp = []
    for rank in range(world):
        sc = scene.flat_scene(3000, seed=scene.SEED + rank)
        r = oracle.prepare_meshes(sc.meshes, sc.transforms, sc.entity_to_transform, scene.main_camera_view())
        exp.append(r["visible_idx"].astype(np.int64) + rank * 3000)
    exp = np.concatenate(exp)
    for rank in range(world):
        gathered, counts, k = ret[rank]
        assert np.array_equal(gathered.astype(np.int64), exp)
        assert counts.sum() == exp.shape[0] and counts[rank] == k
    assert len(np.unique(exp)) == exp.shape[0]


def _empty_shard_worker(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from garden_amd.multi import allgatherv_indices
    buf = torch.arange(100, dtype=torch.int32) + 1000 * rank
    count = 0 if rank == 1 else 40 + rank  # rank 1's tile is entirely behind the camera
    gathered, counts = allgatherv_indices(buf, count, dist)
    ret[rank] = (gathered.numpy().copy(), counts.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_allgatherv_with_an_empty_shard():
    import torch.multiprocessing as mp
    world, port = 3, _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_empty_shard_worker, args=(world, port, ret), nprocs=world, join=True)
    exp = np.concatenate([np.arange(40) + 0, np.arange(42) + 2000])
    for rank in range(world):
        gathered, counts = ret[rank]
        assert list(counts) == [40, 0, 42] and np.array_equal(gathered, exp)


def _padded_exchange_worker(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from garden_amd.multi import ShardOverflow, VisibleListExchange, allgatherv_indices, shard_capacity
    cap = shard_capacity(60, quantum=16)
    ex = VisibleListExchange(dist, "cpu", cap)
    ok = True
    for frame in range(5):  # counts change from frame to frame; slots rotate
        count = (10 + 7 * frame) * (rank + 1) if not (rank == 1 and frame == 2) else 0
        idx = torch.arange(count, dtype=torch.int32) + 1000 * rank + frame
        shard = ex.next_shard()
        shard[0] = count  # what gv_results_copy_shard_device writes
        shard[1:1 + count] = idx
        padded = ex.exchange()
        dense, counts = ex.compact(padded)
        ref_buf = torch.zeros(cap, dtype=torch.int32)
        ref_buf[:count] = idx
        ref, ref_counts = allgatherv_indices(ref_buf, count, dist)  # the exact-size form gives the same list
        ok = ok and torch.equal(dense, ref) and torch.equal(counts, ref_counts.to(torch.int64))
    ex.drain()
    # a frame that does not fit: detected from the headers, with the capacity it would have needed
    shard = ex.next_shard()
    shard[0] = cap + 5 if rank == 0 else 3
    ex.exchange()
    try:
        ex.drain()
        overflow = None
    except ShardOverflow as e:
        overflow = e.needed
    ret[rank] = (ok, overflow, cap)
    dist.barrier()
    dist.destroy_process_group()


def test_padded_exchange_world_size_2_gloo():
    """The per-frame, sync-free exchange (fixed-capacity shards, counts in the headers) against the exact all-gatherv,
    incl. an empty shard, changing counts, slot rotation and overflow detection on every rank."""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_padded_exchange_worker, args=(world, port, ret), nprocs=world, join=True)
    for rank in range(world):
        ok, overflow, cap = ret[rank]
        assert ok and overflow == cap + 5


def _uneven_exchange_worker(rank, world, port, ret, mode):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from garden_amd.multi import ShardOverflow, VisibleListExchange
    counts = [0, 700, 33][:world]          # a tile behind the camera, a full one, a sparse one
    caps = [16, 704, 48][:world]           # what travels per rank (the direct patterns); the rows stay 1 + 704 wide
    ex = VisibleListExchange(dist, "cpu", 704, mode=mode, capacities=caps)
    ok = True
    for frame in range(3):
        shard = ex.next_shard()
        shard.zero_()
        shard[0] = counts[rank]
        shard[1:1 + counts[rank]] = torch.arange(counts[rank], dtype=torch.int32) + 10_000 * rank + frame
        padded = ex.exchange()
        dense, got_counts = ex.compact(padded)
        exp = torch.cat([torch.arange(counts[r], dtype=torch.int32) + 10_000 * r + frame for r in range(world)])
        ok = ok and torch.equal(dense, exp) and got_counts.tolist() == counts
    ex.drain()
    shard = ex.next_shard()  # the sparse rank's list outgrows the part of its shard that travels (but not the row)
    shard[0] = 60 if rank == world - 1 else counts[rank]
    ex.exchange()
    try:
        ex.drain()
        overflow = None
    except ShardOverflow as e:
        overflow = e.needed
    ret[rank] = (ok, overflow)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["p2p", "broadcast"])
def test_direct_patterns_move_only_what_each_rank_needs_gloo(mode):
    """capacities=[...]: in the p2p and broadcast patterns rank r's shard travels as 1 + capacities[r] words (every rank
    knows every count, so every rank sizes every transfer the same way); results as with full rows, and a list that
    outgrows its rank's travelling part is reported on every rank."""
    import torch.multiprocessing as mp
    world, port = 3, _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_uneven_exchange_worker, args=(world, port, ret, mode), nprocs=world, join=True)
    for rank in range(world):
        ok, overflow = ret[rank]
        assert ok and overflow == 60, (rank, ok, overflow)


def _mask_exchange_worker(rank, world, port, ret, mode):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from garden_amd.multi import VisibleListExchange, expand_mask_rows, mask_words, pack_mask_shard
    slots = 5003  # not a multiple of 32: the last word is partly used
    ex = VisibleListExchange(dist, "cpu", mask_words(slots), mode=mode, payload="mask")
    rng = np.random.default_rng(7)  # the same on every rank: each one can rebuild every shard
    ok = True
    for frame in range(4):
        sets = [np.sort(rng.choice(slots, size=(0 if (r == 1 and frame == 2) else 40 + (4500 // world) * r + 11 * frame), replace=False)) for r in range(world)]
        shard = ex.next_shard()
        shard.copy_(pack_mask_shard(sets[rank], slots))  # what gv_results_copy_mask_device writes
        padded = ex.exchange()
        got, counts = expand_mask_rows(padded, slots)
        exp = np.concatenate([sets[r].astype(np.int64) + r * slots for r in range(world)])
        ok = ok and np.array_equal(got, exp) and counts.tolist() == [s.size for s in sets]
    ex.drain()  # a count above the word capacity is not an overflow for this payload
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "allgather"), (3, "p2p"), (2, "broadcast"), (8, "p2p"), (8, "broadcast")])  # 8: the node's rank count
def test_mask_payload_exchange_gloo(world, mode):
    """Shards as one bit per pool slot behind the count (the encoding for dense views: a fixed size whatever the view):
    every rank ends with every rank's visible SET, through each transport pattern, incl. an empty shard and a pool whose
    last word is partly used."""
    import torch.multiprocessing as mp
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_mask_exchange_worker, args=(world, port, ret, mode), nprocs=world, join=True)
    assert all(ret[r] for r in range(world))


def test_worker_pool_under_thread_sanitizer(tmp_path):
    """garden_amd/csrc/gv_workers.*: the persistent host worker pool behind the gathers and the isVisible write-back
    (the reference's ThreadPool::addItems split, thread-pool.cpp:180-194). Host-only: built here with
    -fsanitize=thread and stressed (coverage, back-to-back runs, concurrent callers)."""
    import subprocess
    root = os.path.join(os.path.dirname(__file__), "..")
    exe = str(tmp_path / "workers_test_tsan")
    build = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread",
                            os.path.join(root, "tests/cpp/workers_test.cpp"), os.path.join(root, "garden_amd/csrc/gv_workers.cpp"),
                            "-o", exe], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    # the fork section starts threads in a child of a multi-threaded process, which TSan refuses by default
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, TSAN_OPTIONS="die_after_fork=0"))
    assert run.returncode == 0 and '"ok": true' in run.stdout, run.stdout + run.stderr
    assert "ThreadSanitizer" not in run.stderr, run.stderr


# ---- world -> spatial tiles: partition_world + the exchange, against the whole-world oracle ----
def _mixed_world(n=6000):
    """One hierarchical world with the awkward cases: free slots, dangling parents, meshes without a transform,
    mesh and transform pools in different orders."""
    sc = scene.hierarchy_scene(n, depth=4, fanout=6)
    sc = scene.shuffled_scene(sc, fraction=0.3, drop_transforms=0.02)
    return sc


def test_partition_world_keeps_trees_together_and_maps_back(oracle):
    from garden_amd.multi import partition_world, tile_of_positions
    sc = _mixed_world()
    grid = (2, 2, 2)
    part = partition_world(sc, grid)
    tr = sc.transforms
    n = sc.count
    # every mesh and transform slot lands in exactly one tile, and the maps invert each other
    assert sorted(np.concatenate(part.mesh_global).tolist()) == list(range(n))
    assert sorted(np.concatenate(part.transform_global).tolist()) == list(range(tr.shape[0]))
    for t, ts in enumerate(part.tiles):
        assert np.array_equal(part.mesh_local[part.mesh_global[t]], np.arange(ts.count))
        assert np.all(part.mesh_tile[part.mesh_global[t]] == t)
        assert np.all(np.diff(part.mesh_global[t]) > 0)  # global slot order kept
    # roots go to the tile of their position; every live descendant sits in its root's tile
    live = tr["entity"] != 0
    side = 100.0 * n ** (1.0 / 3.0)
    root_tile = tile_of_positions(tr["position"][:, :3].astype(np.float64), side, grid)
    assert np.array_equal(part.transform_tile[live], root_tile[part.root_slot][live])
    assert len(set(part.transform_tile[live].tolist())) > 1  # really spread over tiles
    # a tile computes the same world matrices as the whole world (chains are never cut) ...
    world = oracle.world_matrices(tr, sc.entity_to_transform)
    view = scene.main_camera_view()
    whole = oracle.prepare_meshes(sc.meshes.copy(), tr, sc.entity_to_transform, view)
    union, models = [], {}
    for t, ts in enumerate(part.tiles):
        lw = oracle.world_matrices(ts.transforms, ts.entity_to_transform)
        assert np.array_equal(lw.view(np.uint32), world[part.transform_global[t]].view(np.uint32))
        r = oracle.prepare_meshes(ts.meshes.copy(), ts.transforms, ts.entity_to_transform, view)
        g = part.to_global(t, r["visible_idx"])
        union.append(g)
        for slot, bm in zip(g.tolist(), r["baked_model"]):
            models[slot] = bm
    # ... and the mapped union of the tiles' visible lists IS the whole-world visible set, records included
    union = np.sort(np.concatenate(union))
    order = np.argsort(whole["visible_idx"], kind="stable")
    assert np.array_equal(union, whole["visible_idx"][order].astype(np.int64)) and whole["draw_count"] > 0
    for slot, bm in zip(whole["visible_idx"][order].tolist(), whole["baked_model"][order]):
        assert np.array_equal(models[slot].view(np.uint32), bm.view(np.uint32))


def test_cells_dealt_round_robin_in_morton_order_share_every_view_evenly(oracle):
    """SURVEY.md §8e after VERDICT r3: one octant per rank leaves the ranks behind the camera idle (visible_by_rank
    [.., 0, 0, 0, 0] on the cfg5 shape). Cells in Morton order dealt in rotating rounds: every rank owns a cell of every
    2 x 2 x 2 block (8 ranks), so every rank gets its share of whatever the camera looks at — for the bench's view and for views in other
    directions — and the union of the ranks' visible lists is still the whole world's."""
    from garden_amd.multi import cell_grid, cell_owners, partition_world
    owners = cell_owners((8, 8, 8), 8)
    assert np.array_equal(np.bincount(owners), np.full(8, 64))
    blocks = owners.reshape(8, 8, 8)  # [z, y, x]
    for z in range(0, 8, 2):
        for y in range(0, 8, 2):
            for x in range(0, 8, 2):
                assert sorted(blocks[z:z + 2, y:y + 2, x:x + 2].ravel().tolist()) == list(range(8))
    assert cell_grid(8) == (16, 16, 16) and cell_grid(8, 64) == (8, 8, 8) and cell_grid(2, 64) == (8, 4, 4) and cell_grid(1, 1) == (1, 1, 1)
    assert np.array_equal(cell_owners((2, 2, 2), 8), np.arange(8))  # (cell id x + 2y + 4z IS the Morton code there; round 0 is not rotated)
    assert len({tuple(blocks[z:z + 2, y:y + 2, x:x + 2].ravel()) for z in range(0, 8, 2) for y in range(0, 8, 2) for x in range(0, 8, 2)}) > 4
    sc = scene.flat_scene(120_000)
    part = partition_world(sc, cell_grid(8), ranks=8)
    sizes = np.array([t.count for t in part.tiles], dtype=np.float64)
    assert sizes.max() / sizes.mean() < 1.05
    octants = partition_world(sc, (2, 2, 2))
    for seed in (0, 1, 2, 3):
        view = scene.main_camera_view() if seed == 0 else scene.main_camera_view(seed=scene.SEED + 101 * seed)
        whole = oracle.prepare_meshes(sc.meshes.copy(), sc.transforms, sc.entity_to_transform, view)
        counts, union = [], []
        for r, ts in enumerate(part.tiles):
            res = oracle.prepare_meshes(ts.meshes.copy(), ts.transforms, ts.entity_to_transform, view)
            counts.append(res["draw_count"])
            union.append(part.to_global(r, res["visible_idx"]))
        counts = np.array(counts, dtype=np.float64)
        assert counts.min() > 0 and counts.max() / counts.mean() <= 1.15, (seed, counts)
        assert np.array_equal(np.sort(np.concatenate(union)), np.sort(whole["visible_idx"].astype(np.int64)))
        if seed == 0:  # what the round-3 partition did with the same view: ranks with nothing to show
            old = [oracle.prepare_meshes(t.meshes.copy(), t.transforms, t.entity_to_transform, view)["draw_count"] for t in octants.tiles]
            assert min(old) == 0 or max(old) / (sum(old) / 8.0) > 1.5, old


def test_bench_ranks_share_the_cfg5_view_evenly(oracle):
    """bench.py's N > 1 scene (make_tile_scene): every rank's synthetic entities are spread over the cells cell_owners() deals
    to it; on the cfg5 shape with 8 ranks no rank is idle and none carries much more than its share (max / mean <= 1.5;
    round 3's octants gave [.., 0, 0, 0, 0])."""
    import bench
    counts = []
    for rank in range(8):
        sc = bench.make_tile_scene(bench.WORKLOADS["cfg5"], 60_000, rank, 8)
        counts.append(oracle.prepare_meshes(sc.meshes.copy(), sc.transforms, sc.entity_to_transform, scene.main_camera_view())["draw_count"])
    counts = np.array(counts, dtype=np.float64)
    assert counts.min() > 0 and counts.max() / counts.mean() <= 1.15, counts
    # one rank alone keeps the whole cube (no cells to deal)
    one = bench.make_tile_scene(bench.WORKLOADS["cfg5"], 5_000, 0, 1)
    assert np.array_equal(one.transforms["position"], scene.flat_scene(5_000).transforms["position"])


def test_partition_world_with_one_tile_is_the_world(oracle):
    from garden_amd.multi import partition_world
    sc = _mixed_world(2000)
    part = partition_world(sc, (1, 1, 1))
    assert len(part.tiles) == 1 and part.tiles[0].count == sc.count
    assert np.array_equal(part.mesh_global[0], np.arange(sc.count))
    view = scene.main_camera_view()
    a = oracle.prepare_meshes(sc.meshes.copy(), sc.transforms, sc.entity_to_transform, view)
    t = part.tiles[0]
    b = oracle.prepare_meshes(t.meshes.copy(), t.transforms, t.entity_to_transform, view)
    assert np.array_equal(a["visible_idx"], b["visible_idx"]) and a["draw_count"] > 0
    assert np.array_equal(a["baked_model"].view(np.uint32), b["baked_model"].view(np.uint32))


def _tile_exchange_worker(rank, world, port, mode, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from garden_amd.multi import VisibleListExchange, partition_world, shard_capacity, tile_grid
    from oracle import oracle_py
    sc = _mixed_world()
    grid = tile_grid(world)
    if grid[0] * grid[1] * grid[2] != world:  # not a power of two: slabs
        grid = (world, 1, 1)
    part = partition_world(sc, grid)  # every rank cuts the same world the same way and keeps its tile
    ts = part.tiles[rank]
    r = oracle_py.prepare_meshes(ts.meshes.copy(), ts.transforms, ts.entity_to_transform, scene.main_camera_view())
    mine = part.to_global(rank, r["visible_idx"])  # GLOBAL mesh slots travel
    ex = VisibleListExchange(dist, "cpu", shard_capacity(sc.count, quantum=64), mode=mode)
    out = None
    for frame in range(3):  # slots rotate; same lists every frame
        shard = ex.next_shard()
        shard[0] = mine.shape[0]
        shard[1:1 + mine.shape[0]] = torch.from_numpy(mine.astype(np.int32))
        out = ex.exchange()
    ex.drain()
    dense, counts = ex.compact(out)
    ret[rank] = (dense.numpy().copy(), counts.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "allgather"), (2, "p2p"), (2, "broadcast"), (3, "p2p"),
                                        (8, "allgather"), (8, "p2p"), (8, "broadcast")])  # 8 ranks = the node: cfg5's 2 x 2 x 2 octants
def test_partitioned_world_through_the_exchange_gloo(oracle, world, mode):
    """ONE world -> spatial tiles -> per-tile cull -> exchange (each transport pattern) -> every rank holds the
    whole-world visible set."""
    import torch.multiprocessing as mp
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_tile_exchange_worker, args=(world, port, mode, ret), nprocs=world, join=True)
    sc = _mixed_world()
    whole = oracle.prepare_meshes(sc.meshes.copy(), sc.transforms, sc.entity_to_transform, scene.main_camera_view())
    exp = np.sort(whole["visible_idx"].astype(np.int64))
    for rank in range(world):
        dense, counts = ret[rank]
        assert counts.shape[0] == world and counts.sum() == exp.shape[0]
        assert np.array_equal(np.sort(dense.astype(np.int64)), exp)
        assert np.array_equal(dense, ret[0][0])  # same rows in the same place on every rank


def test_bench_launch_shape_is_checked_before_the_gpu_is_touched():
    """bench.py --gpus N vs WORLD_SIZE: a mismatch is a one-line JSON error and rc 2 — decided before torch is
    imported, so it is the same on a box without a GPU."""
    import json
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(__file__), "..")
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         env=env, timeout=120)
    assert run.returncode == 2
    lines = [l for l in run.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and "error" in json.loads(lines[0])


def test_dirty_ranges_under_address_and_ub_sanitizers(tmp_path):
    """garden_amd/csrc/gv_dirty_ranges.hpp: the itemised dirty marks behind gv_mark_dirty (host-only). Random marks —
    overlapping, adjacent, empty, wrapping first + count — against a bitmap model, built with
    -fsanitize=address,undefined."""
    import subprocess
    root = os.path.join(os.path.dirname(__file__), "..")
    exe = str(tmp_path / "dirty_ranges_test")
    build = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                            os.path.join(root, "tests/cpp/dirty_ranges_test.cpp"), "-o", exe], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and '"ok": true' in run.stdout, run.stdout + run.stderr


def test_host_orchestration_and_exchange_under_thread_sanitizer(tmp_path):
    """The same host build (tests/cpp/hip_stub) and the same driver under -fsanitize=thread: the exchange with one context per rank
    THREAD (1 / 2 / 3 / 8 ranks, 24 random list sequences) and with ONE thread driving 1-4 contexts over tests/cpp/rccl_stub's worker
    threads, the host worker pool, the frame sequences — libgarden_vis keeps no unsynchronised state between contexts (the
    process-wide pieces are the RCCL binding, made once, and the worker pool behind its lock). Zero reports."""
    import subprocess
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    csrc, stub = os.path.join(root, "garden_amd", "csrc"), os.path.join(root, "tests", "cpp", "hip_stub")
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    stubs = tmp_path / "kernel_stubs.cpp"
    gen = subprocess.run([sys.executable, os.path.join(stub, "make_kernel_stubs.py"), csrc], capture_output=True, text=True)
    assert gen.returncode == 0, gen.stderr
    stubs.write_text(gen.stdout)
    flags = ["-std=c++17", "-O1", "-g", "-fsanitize=thread", "-I" + stub, "-I" + csrc]
    sources = [str(stubs), os.path.join(stub, "reorder_cpu.cpp"), os.path.join(stub, "exchange_cpu.cpp"),
               os.path.join(root, "tests", "cpp", "host_orchestration_test.cpp")] + \
              [os.path.join(csrc, f) for f in ("gv_context.cpp", "gv_results.cpp", "gv_mirror.cpp", "gv_exchange.cpp", "gv_scene.cpp", "gv_workers.cpp")]
    builds, objects = [], []
    for src in sources:
        obj = str(tmp_path / (os.path.basename(src) + ".o"))
        objects.append(obj)
        builds.append(subprocess.Popen([clang, *flags, "-c", src, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for b in builds:
        out, _ = b.communicate(timeout=900)
        assert b.returncode == 0, out[-3000:]
    exe = str(tmp_path / "host_orchestration_test_tsan")
    link = subprocess.run([clang, "-fsanitize=thread", *objects, "-o", exe, "-lpthread", "-ldl"], capture_output=True, text=True)
    assert link.returncode == 0, link.stderr[-3000:]
    transport = str(tmp_path / "librccl_stub_tsan.so")
    tb = subprocess.run([clang, *flags, "-fPIC", "-shared", os.path.join(root, "tests", "cpp", "rccl_stub", "rccl_stub.cpp"), "-o", transport,
                         "-lrt", "-lpthread"], capture_output=True, text=True)
    assert tb.returncode == 0, tb.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, GV_RCCL_LIBRARY=transport, TSAN_OPTIONS="halt_on_error=0"))
    text = run.stdout + run.stderr
    assert run.returncode == 0 and "host orchestration: ok" in run.stdout and "ThreadSanitizer" not in text, text[-4000:]
    assert run.stdout.count("exchange over the stub transport") == 4 + 24 and run.stdout.count("exchange driven by ONE thread") == 3 + 6
    assert run.stdout.count("exchange by peer stores (no communicator)") == 4 + 6


def test_rank_shares_under_address_and_ub_sanitizers(tmp_path):
    """garden_amd/csrc/host/rank_shares.hpp — what each rank of the drop-in's multi-GPU mode (one process, N contexts) holds of the
    engine's pools — on the CPU under -fsanitize=address,undefined (tests/cpp/rank_shares_test.cpp): hierarchies, free slots, meshes
    without a transform, destroyed entities, re-parenting, 1 / 2 / 3 / 8 ranks: every transform on exactly one rank (the one
    gv_cell_owner gives its root's position), parents co-located and renumbered, every mesh slot on exactly one rank."""
    import subprocess
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    exe = str(tmp_path / "rank_shares_test")
    lib_dir = os.path.join(root, "garden_amd", "lib")
    build = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-Wno-invalid-offsetof",
                            "-fno-strict-aliasing", os.path.join(root, "tests/cpp/rank_shares_test.cpp"), "-o", exe, "-L" + lib_dir, "-lgarden_vis",
                            "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and '"ok": true' in run.stdout, (run.stdout + run.stderr)[-3000:]


def test_host_orchestration_under_address_and_ub_sanitizers(tmp_path):
    """The 2.8 k lines of host orchestration behind the C-ABI (garden_amd/csrc/gv_context.cpp, gv_results.cpp, gv_mirror.cpp, gv_exchange.cpp,
    + gv_scene.cpp, gv_workers.cpp) are otherwise only ever compiled as HIP. Here they are built as plain C++ against
    tests/cpp/hip_stub (device memory = zeroed host memory, copies = memcpy, kernels = generated no-ops; the mirror re-order's
    kernels as plain loops, reorder_cpu.cpp, so that its host half sees real tables) with
    -fsanitize=address,undefined and driven through include/garden_vis.h by tests/cpp/host_orchestration_test.cpp: binds, mirror
    builds, every dirty-range path, growth / shrink / moved pools, columns, ready counts, record targets, batched ticks, sorts,
    Hi-Z builds, sweeps, scene ingest, tiles, error codes — in four context configurations. TEST-ONLY stub: the product library
    still refuses to run without a gfx950 device (test_abi.py)."""
    import subprocess
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    csrc, stub = os.path.join(root, "garden_amd", "csrc"), os.path.join(root, "tests", "cpp", "hip_stub")
    clang = "/opt/rocm/lib/llvm/bin/clang++"  # (g++ 11 has no _Float16, which gv_context.cpp's RG16F read-back uses)
    stubs = tmp_path / "kernel_stubs.cpp"
    gen = subprocess.run([sys.executable, os.path.join(stub, "make_kernel_stubs.py"), csrc], capture_output=True, text=True)
    assert gen.returncode == 0 and gen.stdout.count("hipError_t launch_") > 30, gen.stderr
    stubs.write_text(gen.stdout)
    # (-fno-sanitize=function: RCCL's entry points are called through dlsym'ed pointers whose parameter structs are declared on
    # each side of the C boundary — same layout, different C++ types)
    flags = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize=function", "-fno-sanitize-recover=undefined",
             "-I" + stub, "-I" + csrc]
    sources = [str(stubs), os.path.join(stub, "reorder_cpu.cpp"), os.path.join(stub, "exchange_cpu.cpp"),
               os.path.join(root, "tests", "cpp", "host_orchestration_test.cpp")] + \
              [os.path.join(csrc, f) for f in ("gv_context.cpp", "gv_results.cpp", "gv_mirror.cpp", "gv_exchange.cpp", "gv_scene.cpp", "gv_workers.cpp")]
    objects = []
    builds = []
    for src in sources:  # compiled side by side
        obj = str(tmp_path / (os.path.basename(src) + ".o"))
        objects.append(obj)
        builds.append(subprocess.Popen([clang, *flags, "-c", src, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for b in builds:
        out, _ = b.communicate(timeout=900)
        assert b.returncode == 0, out[-3000:]
    exe = str(tmp_path / "host_orchestration_test")
    link = subprocess.run([clang, "-fsanitize=address,undefined", *objects, "-o", exe, "-lpthread", "-ldl"], capture_output=True, text=True)
    assert link.returncode == 0, link.stderr[-3000:]
    # the exchange step with 1 / 2 / 3 / 8 ranks (threads): RCCL's entry points over shared memory (tests/cpp/rccl_stub), built
    # against the same stub runtime and sanitizers, named to the library with GV_RCCL_LIBRARY
    transport = str(tmp_path / "librccl_stub_cpu.so")
    tb = subprocess.run([clang, *flags, "-fPIC", "-shared", os.path.join(root, "tests", "cpp", "rccl_stub", "rccl_stub.cpp"), "-o", transport,
                         "-lrt", "-lpthread"], capture_output=True, text=True)
    assert tb.returncode == 0, tb.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=dict(os.environ, GV_RCCL_LIBRARY=transport))
    assert run.returncode == 0 and "host orchestration: ok" in run.stdout, (run.stdout + run.stderr)[-4000:]
    assert "exchange over the stub transport, 8 ranks, list sequence 0: ok" in run.stdout, run.stdout[-2000:]
    # ... and 24 sequences of lists that jump at random between empty and the whole pool, 2-5 ranks
    assert run.stdout.count("exchange over the stub transport") == 4 + 24 and "list sequence 24: ok" in run.stdout, run.stdout[-2000:]
    assert run.stdout.count("every row of every frame complete") == 4 + 24  # (no frame is ever handed out short)
    # ONE thread driving 1 / 3 / 4 contexts through gv_exchange_init_all / _visible_all / _acquire_all (the reference's shape: one
    # process, one Manager), scripted and random list sequences
    assert run.stdout.count("exchange driven by ONE thread") == 3 + 6 and "exchange driven by ONE thread, 4 ranks, list sequence 0: ok" in run.stdout
    # ... and ONE exchange for all the lists of a frame (gv_exchange_views_all): count table + lists per row, short predictions completed
    batched = [l for l in run.stdout.splitlines() if l.startswith("batched exchange (gv_exchange_views_all)")]
    assert len(batched) == 3 + 6 and all("6 frames acquired" in l for l in batched), batched
    assert run.stdout.count("first batched frame of a communicator, 3 ranks") == 3  # (p2p / broadcast / all-gather from frame 0: the tables arrive with the headers)
    # ... and the same calls with NO communicator (gv_exchange_init_peers: direct stores into every member's rows), 1 / 2 / 4 / 8 ranks +
    # six random list sequences: no frame ever short, a member leaving dissolves the group, a new group / a communicator afterwards
    assert run.stdout.count("exchange by peer stores (no communicator)") == 4 + 6 and "ONE thread, 8 ranks, list sequence 0: ok — 24 frames, none short" in run.stdout
    print("\n".join(l for l in run.stdout.splitlines() if l.startswith("exchange ")))
    # every allocation of a small frame sequence failing once, in turn: error codes, no leaks, contexts that recover
    assert "allocation failures:" in run.stdout and "every context recovered: ok" in run.stdout, run.stdout[-2000:]
    assert "allocation failures in a frame of mid-sized pools:" in run.stdout, run.stdout[-2000:]  # (sorts that wait for the first read, batched)
    # ... and every allocation of the exchange step (rows, staging shard, widened rows for the tails) failing in turn: GV_E_OOM, the
    # same frame tried again, acquired whole
    assert "allocation failures in the exchange:" in run.stdout and "acquired whole all the same: ok" in run.stdout, run.stdout[-2000:]
    assert "allocation failures in the exchange (peer stores):" in run.stdout, run.stdout[-2000:]  # ... and of a peer group's frames
    # ... and the bounded waits with work that NEVER finishes (streams that never drain, events that never complete): GV_E_TIMEOUT
    # inside the limit on every context of the call, GV_E_STATE afterwards, shutdown / destroy release everything, the context recovers
    assert "exchange, bounded waits:" in run.stdout and "the context recovers: ok" in run.stdout, run.stdout[-2000:]
    print(run.stdout[-600:])
    # the random schedules the GPU tier checks against the oracle (tests/schedules.py), replayed here under the sanitizers
    import schedules
    paths = []
    for seed in range(200):
        path = tmp_path / f"schedule_{seed}.txt"
        path.write_text(schedules.to_text(schedules.generate(seed)))
        paths.append(str(path))
    # (with the transport named, so that the exch / exchp operations of the schedules run instead of being skipped)
    replay = subprocess.run([exe, *paths], capture_output=True, text=True, timeout=1800, env=dict(os.environ, GV_RCCL_LIBRARY=transport))
    assert replay.returncode == 0 and "schedules: 200 replayed ok" in replay.stdout, (replay.stdout + replay.stderr)[-4000:]
