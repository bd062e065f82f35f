"""The multi-GPU path end to end, one PROCESS per rank, through the product's own pieces only: every rank builds the same world,
keeps what the dealing rule gives it (partition_world(ranks=): many Morton cells per rank — the Python twin of
gv_scene_extract_rank), binds it with its local -> world slot table (gv_pool_set_index_map), culls, and calls
gv_exchange_visible + gv_exchange_acquire; every rank then holds every rank's list in WORLD slots, and their union must be the
whole world's oracle set IN EVERY FRAME — for a camera that turns, cuts (the frame right after the cut included: predictions that
fall short are completed inside the frame) and comes back, with the frames acquired at once or a frame late. The ranks share the
box's GPU(s), so the rows travel through the tests' shared-memory transport (tests/cpp/rccl_stub, GV_RCCL_LIBRARY; its device form:
collectives are kernels on the library's exchange stream): everything but RCCL's own wire is the product path."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys, time
import numpy as np
sys.path.insert(0, {root!r})
rank, world, n, frames, tmp, how, pattern = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6], sys.argv[7]
batched = len(sys.argv) > 8 and sys.argv[8] == "batched"  # two views per frame, both lists in ONE exchange (gv_exchange_views)
import torch
from garden_amd import scene
from garden_amd.lib import GpuVisibility
from garden_amd.multi import cell_grid, partition_world
from oracle import oracle_py  # (the checker)

sc = scene.shuffled_scene(scene.hierarchy_scene(n, depth=3, fanout=6, seed=9), fraction=0.3)
grid = cell_grid(world, 64)
side = 100.0 * n ** (1.0 / 3.0)
if how == "native":
    # the C-ABI's own chain: every rank parses the scene FILE, keeps its share (gv_scene_extract_rank) and binds it (gv_scene_bind
    # installs the local -> world slot tables); the oracle culls the scene as the loader built it
    sys.path.insert(0, os.path.join({root!r}, "tests"))
    from garden_amd.lib import Scene
    from oracle import scene_json_py as sj
    from test_scene_ingest import _scene_as_aos
    whole = Scene(sj.write_scene(sc.transforms, {{"Model": sc.meshes}}, sc.entity_to_transform), {{"Model": 0}})
    sc = _scene_as_aos(whole)
    mine_native = whole.extract_rank(grid, side, rank, world)
    share = mine_native.info()["mesh_count"][0]
else:
    part = partition_world(sc, grid, side=side, ranks=world)   # every rank cuts the same world the same way ...
    mine = part.tiles[rank]                                    # ... and keeps its own share
    share = mine.count
devices = torch.cuda.device_count()
id_path = os.path.join(tmp, "unique_id.bin")
with GpuVisibility(device=rank % devices) as vis:
    if rank == 0:
        with open(id_path + ".tmp", "wb") as f:
            f.write(GpuVisibility.exchange_unique_id())
        os.rename(id_path + ".tmp", id_path)
    t0 = time.time()
    while not os.path.exists(id_path):
        if time.time() - t0 > 120:
            raise SystemExit("no unique id from rank 0")
        time.sleep(0.01)
    vis.exchange_init(open(id_path, "rb").read(), rank, world)
    if how == "native":
        mine_native.bind(vis)
    else:
        vis.bind_transforms(mine.transforms, mine.entity_to_transform)
        vis.bind_pool(0, mine.meshes)
        vis.hierarchy_rebuild()
        vis.set_index_map(0, part.mesh_global[rank])

    class _Span:
        pass

    report = []
    views, views2, sent = {{}}, {{}}, {{}}

    def check(frame):
        f = vis.exchange_acquire(frame)
        vis.wait()
        span = _Span()
        span.__cuda_array_interface__ = {{"shape": (world * f["row_words"],), "typestr": "<i4", "data": (int(f["ptr"]), False), "version": 2}}
        rows = torch.as_tensor(span, device="cuda:%d" % (rank % devices)).cpu().numpy().view(np.uint32).reshape(world, f["row_words"])
        world_result = oracle_py.prepare_meshes(sc.meshes.copy(), sc.transforms, sc.entity_to_transform, views[frame], threads=2)
        exp = np.sort(world_result["visible_idx"].astype(np.int64))
        got, got2 = [], []
        for r in range(world):
            c = int(rows[r, 0])
            assert c == f["counts"][r] and c + 1 <= f["row_words"]
            if batched:  # row = [2 + total, c_0, c_1, list 0, list 1]
                c0, c1 = int(rows[r, 1]), int(rows[r, 2])
                assert f["items"] == 2 and c == 2 + c0 + c1 and f["item_counts"][r] == [c0, c1]
                got.append(rows[r, 3:3 + c0].astype(np.int64))
                got2.append(rows[r, 3 + c0:3 + c0 + c1].astype(np.int64))
            else:
                got.append(rows[r, 1:1 + c].astype(np.int64))
        union = np.sort(np.concatenate(got))
        # the reference's gather never loses a record (mesh.cpp:177-183): the union of the rows IS the world's visible set
        ok = bool(f["complete"] and np.array_equal(union, exp))
        if batched:  # ... of BOTH views
            second = oracle_py.prepare_meshes(sc.meshes.copy(), sc.transforms, sc.entity_to_transform, views2[frame], threads=2)
            ok = ok and bool(np.array_equal(np.sort(np.concatenate(got2)), np.sort(second["visible_idx"].astype(np.int64))))
            exp = np.concatenate([exp, second["visible_idx"].astype(np.int64), np.zeros(2 * world, dtype=np.int64)])  # (a row's count includes its table's two words)
        report.append(dict(frame=frame, complete=f["complete"], cut=f["cut_ranks"], tails=f["tail_words"], ok=ok, visible=int(exp.shape[0]),
                           counts=f["counts"], room=sent[frame]["room"], mine=int(f["counts"][rank]), mode=f["mode"], pattern=pattern))

    late = None
    for frame in range(frames):
        # the camera starts near a corner of the world (it sees little), turns a little every frame, and cuts to the centre half way:
        # every rank's list jumps and the predictions fall short — that frame must arrive whole like any other
        seed = scene.SEED + (0 if frame < frames // 2 else 777) + (frame % 3)
        corner = (0.47 * side, 0.47 * side, 0.47 * side) if frame < frames // 2 else (0.0, 0.0, 0.0)
        views[frame] = scene.main_camera_view(seed=seed, camera_position=corner)
        vis.exchange_set_mode(frame % 3 if pattern == "turns" else {{"allgather": 0, "p2p": 1, "broadcast": 2}}[pattern])
        if batched:  # a second camera that looks elsewhere (a shadow cascade would be the engine's second view), same cut
            views2[frame] = scene.main_camera_view(seed=seed + 31, camera_position=corner)
            vis.cull(0, [views[frame], views2[frame]])
            sent[frame] = vis.exchange_views([(0, 0, 0), (0, 1, 0)])
        else:
            vis.cull(0, [views[frame]])
            sent[frame] = vis.exchange_visible(0, index_base=0)
        assert not sent[frame]["complete"] and sent[frame]["ptr"] is None
        if late is not None:       # the previous frame, acquired only now: this frame's send has completed it already
            check(late)
            late = None
        if frame % 3 == 1 and frame + 1 < frames:
            late = frame
        else:
            check(frame)
    report.sort(key=lambda d: d["frame"])
    vis.exchange_shutdown()
print("REPORT " + json.dumps(dict(rank=rank, frames=report, share=int(share), total=int(sc.count))))
'''


@pytest.mark.gpu
@pytest.mark.parametrize("world,how,pattern", [(2, "python", "turns"), (4, "python", "p2p"), (3, "native", "broadcast"), (8, "native", "turns"),
                                               (2, "native", "allgather"), (3, "python", "allgather"), (4, "native", "broadcast"), (8, "python", "p2p")])
def test_one_world_dealt_to_rank_processes_culled_and_exchanged_in_world_slots(tmp_path, world, how, pattern):
    _deal_cull_exchange(tmp_path, world, how, pattern, 120_000, 8, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("world,how,pattern", [(2, "python", "turns"), (3, "native", "p2p"), (8, "python", "allgather")])
def test_two_views_per_frame_travel_in_one_exchange(tmp_path, world, how, pattern):
    """gv_exchange_views with one process per rank: every frame culls two views and sends both lists in ONE exchange; on every rank
    and in every frame — the cut included — the union of the gathered first lists == the oracle's visible set of the WHOLE world for
    the first view, and the union of the second lists == the oracle's for the second."""
    _deal_cull_exchange(tmp_path, world, how, pattern, 120_000, 8, 2, extra=["batched"])


@pytest.mark.gpu
def test_a_ten_million_entity_world_dealt_to_eight_rank_processes(tmp_path):
    """The same chain at BASELINE's 10^7: one hierarchy world (trees of 43) cut into 8 ranks' shares by the cell rule, every rank a
    process with its own context, four frames with the cut to the centre in the middle, the travel pattern turning — on every rank
    and in every frame the union of the gathered rows == the oracle's visible set of the WHOLE world."""
    _deal_cull_exchange(tmp_path, 8, "python", "turns", 10_000_000, 4, 2)


def _deal_cull_exchange(tmp_path, world, how, pattern, n, frames, min_completed, extra=()):
    stub = os.path.join(ROOT, "tests", "cpp", "build", "librccl_stub.so")
    script = tmp_path / "rank.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, GV_RCCL_LIBRARY=stub)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), str(n), str(frames), str(tmp_path), how, pattern, *extra], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=1500) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err[-3000:]
    reports = sorted((json.loads(out.split("REPORT ", 1)[1]) for out, _ in outs), key=lambda d: d["rank"])
    shares = np.array([d["share"] for d in reports], dtype=np.float64)
    assert shares.sum() == reports[0]["total"] and shares.max() / shares.mean() < 1.3  # (trees of 43 go with their roots)
    completed = 0
    for k in range(frames):
        per_rank = [d["frames"][k] for d in reports]
        # EVERY frame on EVERY rank: complete, and the union of the rows == the oracle's set for the whole world
        assert all(f["ok"] and f["complete"] and f["frame"] == k for f in per_rank), per_rank
        # every rank saw the same counts and made the same decisions
        assert len({json.dumps([f["counts"], f["room"], f["cut"], f["tails"], f["mode"]]) for f in per_rank}) == 1, per_rank
        assert sum(per_rank[0]["counts"]) == per_rank[0]["visible"]
        if k >= frames // 2:
            assert min(per_rank[0]["counts"]) > 0  # every rank has a share of the view from the centre
        completed += bool(per_rank[0]["cut"])
    # frame 0 has no history, frame 1 sees three times as much, the cut to the centre changes every rank's list: predictions fell
    # short in THOSE frames (tails travelled in a second exchange), and they arrived whole like all the others
    cut_frame = reports[0]["frames"][frames // 2]
    # (the 10 M world's corner camera already sees millions: the centre's lists fit the rooms it left — there only frames 0 and 1 are short)
    centre_cut = cut_frame["cut"] and sum(cut_frame["tails"]) > 0
    assert (centre_cut or min_completed < 3) and reports[0]["frames"][0]["cut"] and completed >= min_completed, \
        [(f["visible"], f["cut"]) for f in reports[0]["frames"]]
