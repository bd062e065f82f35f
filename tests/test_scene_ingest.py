"""Scene ingest: gv_scene_parse_json puts a Garden scene file straight into column pools.

CPU tier: the columns equal, field by field and bit for bit, the AoS pools that a Python restatement of the reference's
loader (oracle/scene_json_py.py: resource.cpp:2421-2510, transform.cpp:517-583, json-serialize.cpp) builds from the
same text — on generated scenes and on the loader's edge cases. GPU tier: cull results on the bound scene equal the
oracle's prepare_meshes on those AoS pools."""
import json

import numpy as np
import pytest

from garden_amd import scene
from garden_amd.lib import GvError, Scene
from oracle import scene_json_py as sj

POOLS = {"Model": 0, "Sprite": 3}


def columns_equal_aos(sc, text, pools=POOLS, add_root_entity=False):
    tr, meshes, e2t, info = sj.read_scene(text, pools, add_root_entity=add_root_entity)
    got = sc.transform_columns()
    n = tr.shape[0]
    assert got["entity"].shape[0] == n
    assert np.array_equal(got["entity"], tr["entity"]) and np.array_equal(got["parent"], tr["parent"])
    assert np.array_equal(got["uid"], tr["uid"])
    assert np.array_equal(got["position"].view(np.uint32), np.ascontiguousarray(tr["position"][:, :3]).view(np.uint32))
    assert np.array_equal(got["scale"].view(np.uint32), np.ascontiguousarray(tr["scale"][:, :3]).view(np.uint32))
    assert np.array_equal(got["rotation"].view(np.uint32), tr["rotation"].view(np.uint32))
    assert np.array_equal(got["self_active"], tr["selfActive"])
    assert np.array_equal(got["ancestors_active"], tr["ancestorsActive"])
    assert np.array_equal(got["model_with_ancestors"], tr["modelWithAncestors"])
    assert np.array_equal(got["entity_to_transform"], e2t)
    for pid, aos in meshes.items():
        m = sc.mesh_columns(pid)
        assert np.array_equal(m["entity"], aos["entity"]) and np.array_equal(m["is_enabled"], aos["isEnabled"])
        assert np.array_equal(m["aabb_min"].view(np.uint32), np.ascontiguousarray(aos["aabbMin"][:, :3]).view(np.uint32))
        assert np.array_equal(m["aabb_max"].view(np.uint32), np.ascontiguousarray(aos["aabbMax"][:, :3]).view(np.uint32))
    i = sc.info()
    for k, v in info.items():
        assert i[k] == v, k
    assert i["transform_count"] == n and all(i["mesh_count"][pid] == meshes[pid].shape[0] for pid in meshes)
    return tr, meshes, e2t


def scene_text(n, hier):
    src = scene.hierarchy_scene(n, depth=4, fanout=5) if hier else scene.flat_scene(n)
    # some exact defaults and splats so the writer's omission / single-number forms are exercised
    src.transforms["position"][::17, :3] = 0
    src.transforms["scale"][::13, :3] = 1
    src.transforms["scale"][5::13, :3] = np.float32(1.75)
    src.transforms["rotation"][::19] = (0, 0, 0, 1)
    src.meshes["aabbMin"][::23, :3] = -0.5
    src.meshes["aabbMax"][::23, :3] = 0.5
    half = src.meshes.shape[0] // 2
    return sj.write_scene(src.transforms, {"Model": src.meshes[:half], "Sprite": src.meshes[half:]}, src.entity_to_transform)


@pytest.mark.parametrize("add_root", [False, True])
@pytest.mark.parametrize("hier", [False, True])
def test_generated_scene_columns_equal_the_reference_loader(hier, add_root):
    text = scene_text(4000, hier)
    sc = Scene(text, POOLS, add_root_entity=add_root)
    tr, meshes, _ = columns_equal_aos(sc, text, add_root_entity=add_root)
    if add_root:  # loadScene(path, true): entity 1 is the root; whatever names no parent hangs under it
        assert tr["entity"][0] == 1 and tr["parent"][0] == 0 and np.all(tr["parent"][1:] != 0)
    assert tr.shape[0] > 3800 and meshes[0].shape[0] > 1500 and meshes[3].shape[0] > 1500
    if hier:
        assert np.count_nonzero(tr["parent"]) > 3000 and np.count_nonzero(tr["ancestorsActive"] == 0) > 0
    sc.close()


U = [sj.encode_uid(0x1000 + k) for k in range(8)]


def test_loader_edge_cases():
    ents = [
        # 1: integer literals are number_integer -> ignored; splat scale; rotation must be an object
        {"components": [{".type": "Transform", "uid": U[0], "position": 5, "scale": 2.5, "rotation": 1.0},
                        {".type": "Model", "aabb": {"min": {"x": -1.0, "y": -2, "z": -3.0}, "max": 4.0}}]},
        {"components": []},                                   # no entity is created
        {"name": "no components key"},                        # nor here
        # 2: child listed BEFORE its parent gets an inactive grandparent: keeps ancestorsActive = 1 (setParent only
        #    looks at the parent's flags at link time and links run in file order)
        {"components": [{".type": "Transform", "uid": U[1], "parent": U[2], "position": {"x": 1.5, "y": 0.25}}]},
        # 3: parent of #2, itself under the inactive #4
        {"components": [{".type": "Transform", "uid": U[2], "parent": U[3]}, {".type": "Camera", "fov": 1.0}]},
        # 4: inactive root
        {"components": [{".type": "Transform", "uid": U[3], "isActive": False}]},
        # 5: linked after #3 was linked: sees the inactive chain
        {"components": [{".type": "Sprite", "isEnabled": False, "aabb": {"min": 1.0, "max": 0.5}},   # invalid box: default kept
                        {".type": "Transform", "uid": U[4], "parent": U[2]}]},
        # 6: duplicate uid (first keeps it), self parent
        {"components": [{".type": "Transform", "uid": U[0], "parent": U[0]}]},
        # 7: unknown parent, malformed uid, escapes in an unread string
        {"components": [{".type": "Transform", "uid": "short", "parent": sj.encode_uid(0xDEAD), "debugName": 'a"b\\c\u00e9\n\t/'},
                        {".type": "Model", "aabb": {"max": {"x": 1.0, "y": 1.0, "z": 1.0}}}]},                # min from default -0.5
        # 8: only components this pass does not read
        {"components": [{".type": "Light"}, {"noType": 1}]},
    ]
    text = json.dumps({"version": "0.1.0", "entities": ents, "extra": [1, {"a": None}, True, -2.5e-3]}, indent=1)
    assert '\\"' in text and "\\u00e9" in text and "\\n" in text  # the string escapes reach the parser
    rooted = Scene(text, POOLS, add_root_entity=True)
    columns_equal_aos(rooted, text, add_root_entity=True)
    rooted.close()
    sc = Scene(text, POOLS)
    tr, meshes, e2t = columns_equal_aos(sc, text)
    i = sc.info()
    assert i["entity_count"] == 8 and i["skipped_entities"] == 1 and i["duplicate_uids"] == 1
    assert i["self_parents"] == 1 and i["unresolved_parents"] == 1 and i["other_components"] == 3
    t = sc.transform_columns()
    assert list(t["position"][0]) == [0, 0, 0] and list(t["scale"][0]) == [2.5, 2.5, 2.5] and list(t["rotation"][0]) == [0, 0, 0, 1]
    assert list(t["position"][1]) == [1.5, 0.25, 0]
    # slot 1 was linked while its parent (slot 2) still looked active and keeps 1; slot 2 then went under the inactive
    # slot 3; slot 4, linked after that, sees slot 2's cleared flag
    assert list(t["ancestors_active"]) == [1, 1, 0, 1, 0, 1, 1]
    m0, m3 = sc.mesh_columns(0), sc.mesh_columns(3)
    assert list(m0["aabb_min"][0]) == [-1, -0.5, -3] and list(m0["aabb_max"][0]) == [4, 4, 4]
    assert list(m0["aabb_min"][1]) == [-0.5] * 3 and list(m0["aabb_max"][1]) == [1, 1, 1]
    assert list(m3["aabb_min"][0]) == [-0.5] * 3 and list(m3["aabb_max"][0]) == [0.5] * 3 and m3["is_enabled"][0] == 0
    sc.close()


@pytest.mark.parametrize("text,needle", [
    ('{"entities": [{"components": [{".type": "Transform"}, {".type": "Transform"}]}]}', "two Transform"),
    ('{"entities": [{"components": [{".type": "Transform", "position": {"x": 1.0,}}]}]}', "byte"),
    ('{"entities": [', "byte"),
    ('[]', "not a JSON object"),
    ('{"entities": []} trailing', "trailing"),
])
def test_malformed_scenes_are_rejected_with_a_message(text, needle):
    with pytest.raises(GvError) as e:
        Scene(text, POOLS)
    assert needle in str(e.value)


def test_empty_scene():
    sc = Scene('{"version": "1.0.0"}', POOLS)
    assert sc.info()["entity_count"] == 0 and sc.transform_columns()["entity"].shape[0] == 0
    sc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("hier,add_root", [(False, False), (True, False), (True, True)])
def test_ingested_scene_culls_like_the_reference_pools(gpu, oracle, hier, add_root):
    text = scene_text(60_000, hier)
    sc = Scene(text, POOLS, add_root_entity=add_root)
    tr, meshes, e2t = columns_equal_aos(sc, text, add_root_entity=add_root)
    sc.bind(gpu)
    view = scene.main_camera_view()
    for pid in (0, 3):
        gpu.cull(pid, [view])
        got = gpu.fetch(0, write_back=True, occupancy=meshes[pid].shape[0])
        aos = meshes[pid].copy()
        exp = oracle.prepare_meshes(aos, tr, e2t, view)
        assert exp["draw_count"] > 100
        assert np.array_equal(got["visible_idx"], exp["visible_idx"])
        assert np.array_equal(got["baked_model"].view(np.uint32), exp["baked_model"].view(np.uint32))
        assert np.array_equal(got["distance_sq"].view(np.uint32), exp["distance_sq"].view(np.uint32))
        assert np.array_equal(sc.mesh_columns(pid)["is_visible"], aos["isVisible"])  # written back into the column
    # world matrices of the ingested transforms
    gpu.sweep(1)
    assert np.array_equal(gpu.get_world(0, tr.shape[0]).view(np.uint32), oracle.world_matrices(tr, e2t).view(np.uint32))
    # rebind ordinary pools afterwards: the context does not keep pointing into the scene
    flat = scene.flat_scene(1000)
    gpu.bind_transforms(flat.transforms, flat.entity_to_transform)
    gpu.bind_pool(0, flat.meshes)
    gpu.bind_pool(3, flat.meshes[:0])
    gpu.hierarchy_rebuild()
    sc.close()


@pytest.mark.gpu
def test_scene_file_to_spatial_tiles_to_cull_to_exchange(oracle):
    """The native multi-GPU chain on one GPU: scene file -> gv_scene_parse_json -> gv_scene_extract_tile (each rank keeps
    its tile) -> gv_scene_bind (installs the tile -> world slot tables) -> cull -> gv_exchange_shards (1-rank RCCL):
    the union of the gathered ids over the tiles is the whole scene's oracle visible set, in world mesh slots."""
    import torch

    from garden_amd.lib import GpuVisibility
    from garden_amd.multi import shard_capacity
    text = scene_text(80_000, True)
    whole = Scene(text, POOLS)
    tr, meshes, e2t = columns_equal_aos(whole, text)
    view = scene.main_camera_view()
    side = 100.0 * 80_000 ** (1.0 / 3.0)
    grid = (2, 2, 2)
    expect = {pid: np.sort(oracle.prepare_meshes(meshes[pid].copy(), tr, e2t, view)["visible_idx"].astype(np.int64)) for pid in (0, 3)}
    union = {0: [], 3: []}
    with GpuVisibility(device=0) as vis:
        vis.exchange_init(GpuVisibility.exchange_unique_id(), 0, 1)
        for t in range(8):
            tile = whole.extract_tile(grid, side, t)
            tile.bind(vis)
            for pid in (0, 3):
                n = tile.info()["mesh_count"][pid]
                vis.cull(pid, [view])
                count = vis.result_count(0)
                cap = shard_capacity(count)
                gathered = torch.zeros(1 + cap, dtype=torch.int32, device="cuda:0")
                torch.cuda.synchronize()  # (the fill runs on torch's stream; the library's stream is non-blocking: no implicit order between them)
                vis.exchange_shards(0, cap, 0, gathered.data_ptr())
                vis.wait()
                row = gathered.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
                assert row[0] == count
                union[pid].append(row[1:1 + count])
                _, mesh_global = tile.tile_maps(pid)
                got = vis.fetch(0, write_back=False, occupancy=n)
                assert np.array_equal(np.sort(row[1:1 + count]), np.sort(mesh_global[got["visible_idx"]].astype(np.int64)))
            # un-bind before the tile's columns go away
            flat = scene.flat_scene(64)
            vis.bind_transforms(flat.transforms, flat.entity_to_transform)
            for pid in (0, 3):
                vis.bind_pool(pid, flat.meshes[:0])
            vis.hierarchy_rebuild()
            tile.close()
        vis.exchange_shutdown()
    for pid in (0, 3):
        assert expect[pid].shape[0] > 500
        assert np.array_equal(np.sort(np.concatenate(union[pid])), expect[pid])
    whole.close()


def test_committed_scene_fixture():
    """tests/golden/scene_golden.{json,npz} (made by tests/golden/make_scene_golden.py): the ingest and the Python
    restatement of the loader both still produce the committed pools."""
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    text = open(os.path.join(here, "scene_golden.json")).read()
    gold = np.load(os.path.join(here, "scene_golden.npz"))
    sc = Scene(text, POOLS)
    tr, meshes, e2t = columns_equal_aos(sc, text)  # ingest == loader restatement, today
    def same_fields(a, b):  # field by field: the structs have padding bytes
        return a.shape == b.shape and all(np.ascontiguousarray(a[f]).tobytes() == np.ascontiguousarray(b[f]).tobytes() for f in a.dtype.names)
    assert same_fields(tr, gold["transforms"]) and np.array_equal(e2t, gold["e2t"])
    assert same_fields(meshes[0], gold["meshes0"]) and same_fields(meshes[3], gold["meshes3"])
    i = sc.info()
    for k, v in zip(gold["info_keys"], gold["info"]):
        assert i[str(k)] == int(v), k
    sc.close()


# ---- BSON: what packed builds ship (json2bson.cpp:41-66 = nlohmann::json::to_bson of the scene's JSON, "debugName" erased)

def to_bson(value, strip_debug_names=True):
    """nlohmann::json::to_bson restated for the test: doubles 0x01, strings 0x02, objects 0x03, arrays 0x04 (keys
    "0", "1", ...), bool 0x08, null 0x0A, integers 0x10 when they fit int32 else 0x12, unsigned beyond int64 0x11."""
    import struct

    def element(key, v):
        name = key.encode() + b"\0"
        if isinstance(v, bool):
            return b"\x08" + name + (b"\1" if v else b"\0")
        if v is None:
            return b"\x0A" + name
        if isinstance(v, float):
            return b"\x01" + name + struct.pack("<d", v)
        if isinstance(v, int):
            if -2 ** 31 <= v < 2 ** 31:
                return b"\x10" + name + struct.pack("<i", v)
            if v < 2 ** 63:
                return b"\x12" + name + struct.pack("<q", v)
            return b"\x11" + name + struct.pack("<Q", v)
        if isinstance(v, str):
            raw = v.encode() + b"\0"
            return b"\x02" + name + struct.pack("<i", len(raw)) + raw
        if isinstance(v, dict):
            return b"\x03" + name + document(v)
        if isinstance(v, list):
            return b"\x04" + name + document({str(k): item for k, item in enumerate(v)})
        raise TypeError(type(v))

    def document(d):
        body = b"".join(element(k, v) for k, v in d.items() if not (strip_debug_names and k == "debugName"))
        return struct.pack("<i", len(body) + 5) + body + b"\0"

    return document(value)


def same_scene(a, b, pools=POOLS):
    ta, tb = a.transform_columns(), b.transform_columns()
    assert ta.keys() == tb.keys()
    for k in ta:
        x, y = np.asarray(ta[k]), np.asarray(tb[k])
        assert x.shape == y.shape and x.tobytes() == y.tobytes(), k
    for pid in pools.values():
        ma, mb = a.mesh_columns(pid), b.mesh_columns(pid)
        for k in ma:
            assert np.asarray(ma[k]).tobytes() == np.asarray(mb[k]).tobytes(), (pid, k)
    assert a.info() == b.info()


@pytest.mark.parametrize("add_root", [False, True])
@pytest.mark.parametrize("hier", [False, True])
def test_bson_scene_equals_the_json_scene(hier, add_root):
    """gv_scene_parse_bson(to_bson(json)) == gv_scene_parse_json(text): from_bson keeps the value categories the
    deserializer's readers test (json-serialize.cpp:383-898), so a packed build loads the same scene."""
    text = scene_text(3000, hier)
    doc = json.loads(text)
    a = Scene(text, POOLS, add_root_entity=add_root)
    b = Scene(to_bson(doc), POOLS, add_root_entity=add_root, bson=True)
    same_scene(a, b)
    columns_equal_aos(b, text, add_root_entity=add_root)  # and both equal the reference loader's AoS pools
    a.close()
    b.close()


def test_bson_value_categories_and_edge_cases():
    ents = [
        {"components": [{".type": "Transform", "uid": U[0], "position": 5, "scale": 2.5, "rotation": 1.0},   # int32: ignored
                        {".type": "Model", "aabb": {"min": {"x": -1.0, "y": -2, "z": -3.0}, "max": 4.0}}]},
        {"components": []},
        {"components": [{".type": "Transform", "uid": U[1], "parent": U[0], "position": {"x": 1.5, "y": 2 ** 40, "z": 2 ** 63 + 5},
                         "debugName": 'erased by json2bson: "quotes" \\ and \n control', "isActive": False}]},   # int64 / uint64: ignored
        {"components": [{".type": "Sprite", "isEnabled": False, "aabb": {"min": {"x": -0.25, "y": -1e-30, "z": -3e38}, "max": {"x": 1e300, "y": 0.5, "z": 0.5}}},
                        {".type": "Transform", "uid": U[2], "parent": U[1], "scale": {"x": 0.1, "y": 1.0000001, "z": 123456.789}}]},
        {"components": [{".type": "Light", "color": [1.0, 0.5, None, True, "s", {"k": [1, 2.5]}]}]},
    ]
    doc = {"version": "0.1.0", "entities": ents, "extra": [1, {"a": None}, True, -2.5e-3, 'text "with" \\ escapes \t']}
    text = json.dumps(doc)
    a, b = Scene(text, POOLS), Scene(to_bson(doc), POOLS, bson=True)
    same_scene(a, b)
    t = b.transform_columns()
    assert list(t["position"][0]) == [0, 0, 0] and list(t["scale"][0]) == [2.5] * 3       # integer literals did not count
    assert list(t["position"][1]) == [1.5, 0, 0] and t["self_active"][1] == 0                # 2^40 / 2^63+5 are integers
    assert np.isinf(b.mesh_columns(3)["aabb_max"][0][0])                                     # 1e300 -> float: +inf
    a.close()
    b.close()
    # non-finite doubles exist in BSON only (JSON text cannot carry them): number_float NaN / inf reach the columns
    doc = {"entities": [{"components": [{".type": "Transform", "uid": U[3], "position": {"x": float("nan"), "y": float("inf"), "z": float("-inf")},
                                         "scale": float("nan")}]}]}
    c = Scene(to_bson(doc), POOLS, bson=True)
    t = c.transform_columns()
    assert np.isnan(t["position"][0][0]) and t["position"][0][1] == np.inf and t["position"][0][2] == -np.inf and np.all(np.isnan(t["scale"][0]))
    c.close()


def test_malformed_bson_is_rejected():
    good = to_bson({"entities": [{"components": [{".type": "Transform", "uid": U[0], "position": {"x": 1.5}}]}]})
    Scene(good, POOLS, bson=True).close()
    import struct
    bad = [good[:-1], good[:10], b"", b"\x05\0\0\0", struct.pack("<i", len(good) + 7) + good[4:],       # truncated / wrong sizes
           good.replace(b"\x01x\0", b"\x07x\0"),                                                           # ObjectId: nlohmann parse_error.114
           good[:4] + good[4:].replace(b"\x02uid\0", b"\x02uid")]                                           # key without terminator shifts everything
    for blob in bad:
        with pytest.raises(GvError):
            Scene(blob, POOLS, bson=True)
    # every prefix and a few hundred single-byte mutations: an error or a scene, never a crash / hang
    rng = np.random.default_rng(5)
    for cut in range(0, len(good), 3):
        try:
            Scene(good[:cut], POOLS, bson=True).close()
        except GvError:
            pass
    for _ in range(400):
        blob = bytearray(good)
        blob[int(rng.integers(len(blob)))] = int(rng.integers(256))
        try:
            Scene(bytes(blob), POOLS, bson=True).close()
        except GvError:
            pass


def test_scene_parsers_under_address_and_ub_sanitizers(tmp_path):
    """gv_scene.cpp is host-only: built here with -fsanitize=address,undefined and fed every prefix and thousands of
    byte mutations of a JSON scene and of its BSON form (tests/cpp/scene_fuzz.cpp)."""
    import os
    import subprocess
    root = os.path.join(os.path.dirname(__file__), "..")
    exe = str(tmp_path / "scene_fuzz")
    build = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                            os.path.join(root, "tests/cpp/scene_fuzz.cpp"), os.path.join(root, "garden_amd/csrc/gv_scene.cpp"), "-o", exe],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    text = scene_text(60, True)
    (tmp_path / "seed.json").write_text(text)
    (tmp_path / "seed.bson").write_bytes(to_bson(json.loads(text)))
    run = subprocess.run([exe, str(tmp_path / "seed.json"), str(tmp_path / "seed.bson")], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and '"ok": true' in run.stdout, run.stdout[-500:] + run.stderr[-3000:]


def _scene_as_aos(sc_handle, pool_id=0):
    """The columns of a parsed scene as a garden_amd.scene.Scene (AoS pools), for the Python partitioner."""
    from garden_amd.pools import MESH_DTYPE, TRANSFORM_DTYPE
    c = sc_handle.transform_columns()
    n = c["entity"].shape[0]
    tr = np.zeros(n, TRANSFORM_DTYPE)
    tr["entity"], tr["parent"], tr["uid"] = c["entity"], c["parent"], c["uid"]
    tr["position"][:, :3], tr["scale"][:, :3], tr["rotation"] = c["position"], c["scale"], c["rotation"]
    tr["selfActive"], tr["ancestorsActive"], tr["modelWithAncestors"] = c["self_active"], c["ancestors_active"], c["model_with_ancestors"]
    m = sc_handle.mesh_columns(pool_id)
    ms = np.zeros(m["entity"].shape[0], MESH_DTYPE)
    ms["entity"], ms["isEnabled"] = m["entity"], m["is_enabled"]
    ms["aabbMin"][:, :3], ms["aabbMax"][:, :3] = m["aabb_min"], m["aabb_max"]
    return scene.Scene(ms, tr, c["entity_to_transform"])


@pytest.mark.parametrize("grid", [(2, 2, 2), (3, 1, 2)])
def test_native_tile_extraction_equals_partition_world(grid):
    """gv_scene_extract_tile (the multi-GPU sharding rule of SURVEY.md §8e in the C-ABI: each rank parses the scene file
    and keeps its tile) against garden_amd/multi.py::partition_world on the same scene: every column of every tile, the
    renumbered ids, the remapped parents, entity_to_transform and the tile -> world slot tables."""
    from garden_amd.multi import partition_world
    src = scene.hierarchy_scene(5000, depth=4, fanout=5)
    src = scene.shuffled_scene(src, fraction=0.4)  # mesh and transform pools in different orders
    text = sj.write_scene(src.transforms, {"Model": src.meshes}, src.entity_to_transform)
    whole = Scene(text, {"Model": 0})
    aos = _scene_as_aos(whole)
    side = 100.0 * 5000 ** (1.0 / 3.0)
    part = partition_world(aos, grid, side=side)
    seen_t, seen_m = 0, 0
    for t in range(grid[0] * grid[1] * grid[2]):
        tile = whole.extract_tile(grid, side, t)
        exp = part.tiles[t]
        c = tile.transform_columns()
        tg, mg = tile.tile_maps(0)
        assert np.array_equal(tg.astype(np.int64), part.transform_global[t]) and np.array_equal(mg.astype(np.int64), part.mesh_global[t])
        assert np.array_equal(c["entity"], exp.transforms["entity"]), t
        assert np.array_equal(c["parent"], exp.transforms["parent"]), t
        assert np.array_equal(c["uid"], exp.transforms["uid"])
        assert np.array_equal(c["position"].view(np.uint32), np.ascontiguousarray(exp.transforms["position"][:, :3]).view(np.uint32))
        assert np.array_equal(c["rotation"].view(np.uint32), exp.transforms["rotation"].view(np.uint32))
        assert np.array_equal(c["scale"].view(np.uint32), np.ascontiguousarray(exp.transforms["scale"][:, :3]).view(np.uint32))
        for a, b in (("self_active", "selfActive"), ("ancestors_active", "ancestorsActive"), ("model_with_ancestors", "modelWithAncestors")):
            assert np.array_equal(c[a], exp.transforms[b])
        assert np.array_equal(c["entity_to_transform"], exp.entity_to_transform), t
        m = tile.mesh_columns(0)
        assert np.array_equal(m["entity"], exp.meshes["entity"]) and np.array_equal(m["is_enabled"], exp.meshes["isEnabled"])
        assert np.array_equal(m["aabb_min"].view(np.uint32), np.ascontiguousarray(exp.meshes["aabbMin"][:, :3]).view(np.uint32))
        assert np.array_equal(m["aabb_max"].view(np.uint32), np.ascontiguousarray(exp.meshes["aabbMax"][:, :3]).view(np.uint32))
        i = tile.info()
        assert i["transform_count"] == exp.transforms.shape[0] and i["mesh_count"][0] == exp.count
        seen_t += exp.transforms.shape[0]
        seen_m += exp.count
        tile.close()
    assert seen_t == aos.transforms.shape[0] and seen_m == aos.count
    with pytest.raises(GvError):
        whole.extract_tile(grid, side, grid[0] * grid[1] * grid[2])  # no such tile
    with pytest.raises(GvError):
        whole.tile_maps(0)  # the whole scene is not a tile
    whole.close()


@pytest.mark.parametrize("grid,world", [((4, 4, 4), 8), ((8, 8, 8), 8), ((4, 2, 2), 3), ((2, 2, 2), 1)])
def test_native_rank_extraction_equals_partition_world_with_ranks(grid, world):
    """gv_scene_extract_rank — many cells per rank, the grid's cells in Morton order dealt round-robin — against
    partition_world(..., ranks=world) on the same scene: the slot tables, the renumbered ids and parents, every mesh column;
    every slot lands on exactly one rank (how even the shares are: test_host_logic.py)."""
    from garden_amd.multi import partition_world
    src = scene.hierarchy_scene(6000, depth=3, fanout=4)
    src = scene.shuffled_scene(src, fraction=0.4)
    text = sj.write_scene(src.transforms, {"Model": src.meshes}, src.entity_to_transform)
    whole = Scene(text, {"Model": 0})
    aos = _scene_as_aos(whole)
    side = 100.0 * 6000 ** (1.0 / 3.0)
    part = partition_world(aos, grid, side=side, ranks=world)
    assert len(part.tiles) == world
    seen_t, seen_m = [], []
    for r in range(world):
        tile = whole.extract_rank(grid, side, r, world)
        exp = part.tiles[r]
        tg, mg = tile.tile_maps(0)
        assert np.array_equal(tg.astype(np.int64), part.transform_global[r]) and np.array_equal(mg.astype(np.int64), part.mesh_global[r])
        c = tile.transform_columns()
        assert np.array_equal(c["entity"], exp.transforms["entity"]) and np.array_equal(c["parent"], exp.transforms["parent"])
        assert np.array_equal(c["entity_to_transform"], exp.entity_to_transform)
        assert np.array_equal(c["position"].view(np.uint32), np.ascontiguousarray(exp.transforms["position"][:, :3]).view(np.uint32))
        m = tile.mesh_columns(0)
        assert np.array_equal(m["entity"], exp.meshes["entity"]) and np.array_equal(m["is_enabled"], exp.meshes["isEnabled"])
        assert np.array_equal(m["aabb_max"].view(np.uint32), np.ascontiguousarray(exp.meshes["aabbMax"][:, :3]).view(np.uint32))
        seen_t.append(tg)
        seen_m.append(mg)
        tile.close()
    assert sorted(np.concatenate(seen_t).tolist()) == list(range(aos.transforms.shape[0]))
    assert sorted(np.concatenate(seen_m).tolist()) == list(range(aos.count))
    with pytest.raises(GvError):
        whole.extract_rank(grid, side, world, world)  # no such rank
    whole.close()


def test_cell_owner_is_the_dealing_rule_of_the_partitioners():
    """gv_cell_owner (what an engine asks when an entity has moved: whose cell is it in now?) == cell_owners()[tile_of_positions()]
    of garden_amd/multi.py, for positions inside, on the faces of and far outside the world cube, several grids and rank counts."""
    from garden_amd.lib import cell_owner
    from garden_amd.multi import cell_grid, cell_owners, tile_of_positions
    rng = np.random.Generator(np.random.PCG64(77))
    side = 4000.0
    pos = rng.uniform(-0.7 * side, 0.7 * side, (20000, 4)).astype(np.float32)
    pos[:50, :3] = 0.5 * side
    pos[50:100, :3] = -0.5 * side
    pos[100:110, 0] = np.float32(1e30)
    for grid, world in ((cell_grid(8), 8), ((4, 4, 2), 3), ((1, 1, 1), 1), ((16, 8, 8), 5)):
        with np.errstate(invalid="ignore"):  # (1e30: astype(int64) of a value outside the range, which the native side reproduces)
            want = cell_owners(grid, world)[tile_of_positions(pos[:, :3].astype(np.float64), side, grid)]
        assert np.array_equal(cell_owner(grid, side, world, pos).astype(np.int64), want), (grid, world)
    with pytest.raises(GvError):
        cell_owner((0, 1, 1), side, 2, pos)


def test_tile_extraction_of_positions_no_cell_can_hold():
    """Root positions that are +-inf (1e300 in the file), huge but finite, or far outside the world cube: the native extraction
    puts them where partition_world's numpy arithmetic puts them (astype(int64) of a value outside the int64 range is INT64_MIN
    -> cell 0 after the clamp), decided in double — the cast itself would be undefined behaviour."""
    from garden_amd.multi import partition_world
    src = scene.flat_scene(600, defects=False)
    marks = {900001.25: "1e300", 900002.25: "-1e300", 900003.25: "1e30", 900004.25: "-1e30", 900005.25: "7e18", 900006.25: "-7e18",
             900007.25: "123456.0"}
    for k, (m, _) in enumerate(marks.items()):
        src.transforms["position"][10 + k, k % 3] = np.float32(m)
    text = sj.write_scene(src.transforms, {"Model": src.meshes}, src.entity_to_transform)
    for m, lit in marks.items():
        assert text.count(repr(m)) == 1
        text = text.replace(repr(m), lit)
    whole = Scene(text, {"Model": 0})
    aos = _scene_as_aos(whole)
    assert np.isinf(aos.transforms["position"][10, 0]) and np.isinf(aos.transforms["position"][11, 1])
    grid, side = (2, 2, 2), 100.0 * 600 ** (1.0 / 3.0)
    with np.errstate(invalid="ignore", over="ignore"):
        part = partition_world(aos, grid, side=side)
    total = 0
    for t in range(8):
        tile = whole.extract_tile(grid, side, t)
        tg, mg = tile.tile_maps(0)
        assert np.array_equal(tg.astype(np.int64), part.transform_global[t]), t
        assert np.array_equal(mg.astype(np.int64), part.mesh_global[t]), t
        total += tg.shape[0]
        tile.close()
    assert total == 600
