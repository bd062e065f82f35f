"""Differential fuzz: many small random worlds — forests of random shape, pools of different sizes and orders, free
slots, missing transforms, inactive chains, modelWithAncestors off, degenerate and inverted boxes, entities sitting
on the frustum planes, perspective / orthographic / shadow views — through every context flavour (default spatial
mirror, slot order, block bounds), single and batched views, with and without Hi-Z and the device sort. Every output
must equal the CPU oracle's bit for bit. Seeds are fixed: failures reproduce."""
import numpy as np
import pytest

from garden_amd import scene
from garden_amd.pools import GV_NONE, MESH_DTYPE, TRANSFORM_DTYPE

pytestmark = pytest.mark.gpu


def random_world(seed):
    rng = np.random.Generator(np.random.PCG64(0xF022 + seed))
    nt = int(rng.integers(1, 900))                      # transform pool size
    nm = int(rng.integers(1, 900))                      # mesh pool size
    n_ent = nt + int(rng.integers(0, 50))               # entity ids 1..n_ent; some have no transform
    tr = np.zeros(nt, TRANSFORM_DTYPE)
    ents = rng.permutation(np.arange(1, n_ent + 1, dtype=np.uint32))[:nt]
    tr["entity"] = ents
    free = rng.random(nt) < 0.05
    tr["entity"][free] = 0
    spread = float(rng.choice([5.0, 60.0, 2000.0]))
    tr["position"][:, :3] = rng.normal(0, spread, (nt, 3)).astype(np.float32)
    tr["scale"][:, :3] = np.exp(rng.normal(0, 0.6, (nt, 3))).astype(np.float32)
    tr["scale"][rng.random(nt) < 0.03, 0] *= -1        # mirrored
    tr["scale"][rng.random(nt) < 0.02, 1] = 0          # flattened
    q = rng.normal(0, 1, (nt, 4)).astype(np.float32)
    q /= np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-6).astype(np.float32)
    tr["rotation"] = q
    tr["rotation"][rng.random(nt) < 0.1] = (0, 0, 0, 1)
    tr["selfActive"] = (rng.random(nt) > 0.08).astype(np.uint8)
    tr["ancestorsActive"] = (rng.random(nt) > 0.05).astype(np.uint8)   # stored byte: not derived (as in the engine)
    tr["modelWithAncestors"] = (rng.random(nt) > 0.1).astype(np.uint8)
    # forest: parent = an entity at a LOWER slot (no cycles), chains up to a random depth; some point at free slots
    depth_bias = float(rng.choice([0.0, 0.5, 0.9]))
    for s in range(1, nt):
        if rng.random() < depth_bias:
            p = int(rng.integers(max(0, s - int(rng.integers(1, 40))), s))
            tr["parent"][s] = tr["entity"][p]            # 0 when that slot is free: chain ends there
    if rng.random() < 0.3:
        tr["parent"][rng.random(nt) < 0.02] = n_ent + 7  # dangling parent id (entity without transform)
    e2t = np.full(n_ent + 1, GV_NONE, np.uint32)
    live = tr["entity"] != 0
    e2t[tr["entity"][live]] = np.nonzero(live)[0].astype(np.uint32)

    stride_extra = int(rng.choice([0, 0, 16, 48]))
    from garden_amd.pools import derived_mesh_dtype
    meshes = np.zeros(nm, derived_mesh_dtype(stride_extra) if stride_extra else MESH_DTYPE)
    mode = rng.random()
    if mode < 0.35 and nm <= nt:                         # exactly paired with the transform pool
        meshes["entity"] = tr["entity"][:nm]
    elif mode < 0.6:                                     # mostly paired
        k = min(nm, nt)
        meshes["entity"][:k] = tr["entity"][:k]
        sw = rng.random(k) < 0.05
        meshes["entity"][:k][sw] = rng.integers(1, n_ent + 1, int(sw.sum()))
    else:                                                # independent order, repeats and strangers allowed
        meshes["entity"] = rng.integers(0, n_ent + 1, nm)
    h = np.exp(rng.normal(-0.5, 0.8, (nm, 3))).astype(np.float32)
    c = rng.normal(0, 0.5, (nm, 3)).astype(np.float32)
    meshes["aabbMin"][:, :3] = c - h
    meshes["aabbMax"][:, :3] = c + h
    z = rng.random(nm)
    meshes["aabbMax"][z < 0.03, :3] = meshes["aabbMin"][z < 0.03, :3]                      # zero size
    inv = (z >= 0.03) & (z < 0.05)
    meshes["aabbMin"][inv, :3], meshes["aabbMax"][inv, :3] = meshes["aabbMax"][inv, :3].copy(), meshes["aabbMin"][inv, :3].copy()
    flat = (z >= 0.05) & (z < 0.08)
    meshes["aabbMax"][flat, 1] = meshes["aabbMin"][flat, 1]                                # zero on one axis only: still a box
    meshes["isEnabled"] = (rng.random(nm) > 0.06).astype(np.uint8)
    meshes["isVisible"] = 7
    return scene.Scene(meshes, tr, e2t), rng, spread


def random_view(rng, spread, k):
    pos = tuple(float(x) for x in rng.normal(0, spread, 3).astype(np.float32))
    if k % 3 == 0:
        return scene.main_camera_view(seed=int(rng.integers(1 << 30)), camera_position=pos)
    if k % 3 == 1:
        v = scene.cascade_view(seed=int(rng.integers(1 << 30)), size=float(rng.uniform(0.5, 4.0) * spread),
                               depth=float(8 * spread), index=k % 4)
        return dict(v, camera_position=np.asarray([pos[0], pos[1], pos[2], 0.0], np.float32))
    v = scene.main_camera_view(seed=int(rng.integers(1 << 30)), camera_position=pos)
    return dict(v, shadow_pass=0, distance_2d=1)


def check(vis, oracle, sc, views, hz, sort=None):
    vis.cull(0, views)
    for vi, v in enumerate(views):
        if sort is not None:
            vis.sort(vi, descending=sort)
        got = vis.fetch(vi, write_back=False, occupancy=sc.count, order="raw" if sort is not None else "slot")
        m2 = sc.meshes.copy()
        exp = oracle.prepare_meshes(m2, sc.transforms, sc.entity_to_transform, v, hiz=hz if v.get("use_hiz") else None)
        assert got["draw_count"] == exp["draw_count"], (vi, got["draw_count"], exp["draw_count"])
        o = np.argsort(exp["visible_idx"], kind="stable")
        if sort is None:
            g = got
        else:
            d = got["distance_sq"]
            assert np.all(np.diff(d) <= 0) if sort else np.all(np.diff(d) >= 0)
            k = np.argsort(got["visible_idx"], kind="stable")
            g = dict(visible_idx=got["visible_idx"][k], baked_model=got["baked_model"][k], distance_sq=d[k])
        assert np.array_equal(g["visible_idx"], exp["visible_idx"][o])
        assert np.array_equal(g["baked_model"].view(np.uint32), exp["baked_model"][o].view(np.uint32))
        assert np.array_equal(g["distance_sq"].view(np.uint32), exp["distance_sq"][o].view(np.uint32))
        if v["shadow_pass"] < 0:
            assert np.array_equal(got["is_visible"], m2["isVisible"])


@pytest.mark.parametrize("seed", range(40))
def test_random_worlds_match_the_oracle(gpu, gpu_slot_order, gpu_bounds, oracle, seed):
    sc, rng, spread = random_world(seed)
    depth = scene.synthetic_depth(int(rng.choice([64, 96, 256])), int(rng.choice([64, 80, 128])), seed=seed, rects=12)
    hz = oracle.Hiz(depth)
    views = [random_view(rng, spread, k) for k in range(3)]
    shared = [dict(v, camera_position=views[0]["camera_position"]) for v in views]  # batched: one cameraPosition
    for vis in (gpu, gpu_slot_order, gpu_bounds):
        vis.bind_transforms(sc.transforms, sc.entity_to_transform)
        vis.bind_pool(0, sc.meshes)
        vis.hierarchy_rebuild()
        vis.hiz_build(depth)
        for v in views:
            check(vis, oracle, sc, [v], hz)
            check(vis, oracle, sc, [dict(v, use_hiz=1)], hz)
        check(vis, oracle, sc, shared, hz)
        check(vis, oracle, sc, [dict(shared[0], use_hiz=1)] + shared[1:], hz)
        check(vis, oracle, sc, [views[0]], hz, sort=bool(seed & 1))
        # world matrices, both sweep forms
        exp_w = oracle.world_matrices(sc.transforms, sc.entity_to_transform)
        for mode in (0, 1):
            vis.sweep(mode)
            assert np.array_equal(vis.get_world(0, sc.transforms.shape[0]).view(np.uint32), exp_w.view(np.uint32)), mode
        # sweep riding on the cull (fused when the pool is exactly paired, two launches otherwise)
        for mode in (2, 3):
            vis.mark_dirty(0, 0, sc.transforms.shape[0])
            vis.sweep(mode)
            check(vis, oracle, sc, [views[0]], hz)
            assert np.array_equal(vis.get_world(0, sc.transforms.shape[0]).view(np.uint32), exp_w.view(np.uint32)), mode
        # edits reported as dirty ranges: TRS + flags of a transform range, boxes / enable bits of a mesh range, a few
        # re-parentings towards lower slots (ranged GV_DIRTY_HIERARCHY)
        nt, nm = sc.transforms.shape[0], sc.count
        lo = int(rng.integers(0, nt)); cnt = int(rng.integers(1, nt - lo + 1))
        sc.transforms["position"][lo:lo + cnt, :3] += rng.normal(0, 0.2 * spread, (cnt, 3)).astype(np.float32)
        sc.transforms["selfActive"][lo:lo + cnt] ^= (rng.random(cnt) < 0.2).astype(np.uint8)
        vis.mark_dirty(0, lo, cnt)
        mlo = int(rng.integers(0, nm)); mcnt = int(rng.integers(1, nm - mlo + 1))
        sc.meshes["aabbMax"][mlo:mlo + mcnt, :3] += np.float32(0.25)
        sc.meshes["isEnabled"][mlo:mlo + mcnt] ^= (rng.random(mcnt) < 0.1).astype(np.uint8)
        vis.mark_dirty(2, mlo, mcnt, pool_id=0)
        if nt > 4:
            hlo = int(rng.integers(1, nt - 1)); hcnt = int(rng.integers(1, min(16, nt - hlo) + 1))
            for s_ in range(hlo, hlo + hcnt):
                sc.transforms["parent"][s_] = sc.transforms["entity"][int(rng.integers(0, s_))]
            vis.mark_dirty(1, hlo, hcnt)
        check(vis, oracle, sc, [views[0]], hz)
        check(vis, oracle, sc, shared, hz)
        # growth: new entities appended to both pools, re-bound with the larger occupancy and no rebuild request
        add_t, add_m = int(rng.integers(1, 60)), int(rng.integers(1, 60))
        n_ent = sc.entity_to_transform.shape[0] - 1
        tr2 = np.concatenate([sc.transforms, np.zeros(add_t, sc.transforms.dtype)])
        # fresh ids, clear of the dangling parent id used above: an existing slot whose parent id starts to resolve is a
        # hierarchy change the caller would have to report, which is not what this step is about
        first_id = n_ent + 20
        new_ids = np.arange(first_id, first_id + add_t, dtype=np.uint32)
        tr2["entity"][nt:] = new_ids
        tr2["position"][nt:, :3] = rng.normal(0, spread, (add_t, 3)).astype(np.float32)
        tr2["scale"][nt:, :3] = 1
        tr2["rotation"][nt:] = (0, 0, 0, 1)
        tr2["selfActive"][nt:] = tr2["ancestorsActive"][nt:] = tr2["modelWithAncestors"][nt:] = 1
        live_old = sc.transforms["entity"][sc.transforms["entity"] != 0]
        if live_old.size:
            tr2["parent"][nt::2] = rng.choice(live_old, size=tr2["parent"][nt::2].shape[0])
        e2t2 = np.concatenate([sc.entity_to_transform, np.full(first_id - n_ent - 1, GV_NONE, np.uint32),
                               np.arange(nt, nt + add_t, dtype=np.uint32)])
        me2 = np.concatenate([sc.meshes, np.zeros(add_m, sc.meshes.dtype)])
        me2["entity"][nm:] = rng.choice(np.concatenate([new_ids, np.arange(1, n_ent + 1, dtype=np.uint32)]), size=add_m)
        me2["aabbMin"][nm:, :3] = -0.5
        me2["aabbMax"][nm:, :3] = 0.5
        me2["isEnabled"][nm:] = 1
        grown = scene.Scene(me2, tr2, e2t2)
        vis.bind_transforms(grown.transforms, grown.entity_to_transform)
        vis.bind_pool(0, grown.meshes)
        check(vis, oracle, grown, [views[0]], hz)
        check(vis, oracle, grown, shared, hz)
        vis.sweep(1)
        assert np.array_equal(vis.get_world(0, nt + add_t).view(np.uint32),
                              oracle.world_matrices(grown.transforms, grown.entity_to_transform).view(np.uint32))


# ---- random schedules over the held-back mechanisms (tests/schedules.py; the same text runs under ASan in the CPU tier) ----
RECORD_DTYPE = np.dtype({"names": ["componentOffset", "bakedModel", "distanceSq"], "formats": ["<u8", ("<f4", 12), "<f4"],
                         "offsets": [0, 8, 56], "itemsize": 64})


class ScheduleReplay:
    """Replays one schedule on a context and checks every reader against the oracle evaluated on the state AT THE gv_cull it
    reads (recorded culls see the pools and the pyramid as they were when they were recorded)."""

    def __init__(self, vis, oracle, schedule, seed, rg16f=False):
        import torch
        self.torch = torch
        self.vis, self.oracle, self.rg16f = vis, oracle, rg16f
        self.rng = np.random.Generator(np.random.PCG64(0xBEEF + seed))
        _, n_xf, *sizes = schedule[0]
        self.exchange = bool(sizes) and sizes[-1] == "x"
        if self.exchange:
            sizes = sizes[:-1]
        base = scene.flat_scene(n_xf, seed=100 + seed)
        self.tr, self.e2t = base.transforms, np.asarray(base.entity_to_transform, dtype=np.uint32)
        self.pools = []
        for p, n in enumerate(sizes):
            m = base.meshes[:n].copy()
            if p % 3 == 2:  # an independent order
                m = m[self.rng.permutation(n)]
            self.pools.append(m)
        self.hz = self.depth = None
        self.ready = {p: self.rng.choice(np.array([0, 1, 1, 2, 3], np.uint8), m.shape[0]) for p, m in enumerate(self.pools) if p % 4 == 2}
        self.targets = {}  # (pool, view) -> the caller's record array (gv_pool_set_record_target)
        self.culls = {}   # pool -> dict(meshes, tr, views, hz, sorted[view], expected[view] (lazily))
        self.last_pool = None
        self.camera = tuple(float(x) for x in self.rng.normal(0, 300.0, 3).astype(np.float32))
        vis.bind_transforms(self.tr, self.e2t)
        for p, m in enumerate(self.pools):
            vis.bind_pool(p, m)
            vis.set_record_layout(p, RECORD_DTYPE if p % 2 == 1 else None, component_stride=m.dtype.itemsize)
            if p in self.ready:
                vis.bind_ready(p, self.ready[p])
        vis.hierarchy_rebuild()
        if self.exchange:
            vis.exchange_init(type(vis).exchange_unique_id(), 0, 1)
        self.readers = 0
        self.sweep_pending = False

    def check_world(self):
        lo = int(self.rng.integers(0, self.tr.shape[0] - 64))
        exp_w = self.oracle.world_matrices(self.tr, self.e2t, lo, 64)
        assert np.array_equal(self.vis.get_world(lo, 64).view(np.uint32), exp_w.view(np.uint32)), "world matrices"

    def view_of(self, kind, k):
        cam = np.asarray([*self.camera, 0.0], np.float32)
        if kind == "s":
            v = scene.cascade_view(seed=int(self.rng.integers(1 << 30)), size=float(self.rng.uniform(2000, 30000)), depth=60000.0, index=k % 4)
            return dict(v, camera_position=cam)
        v = scene.main_camera_view(seed=int(self.rng.integers(1 << 30)), camera_position=self.camera)
        return dict(v, use_hiz=1 if kind == "h" else 0, distance_2d=1 if kind == "u" else 0, emit_records=0 if kind == "c" else 1)

    def expected(self, pool, view):
        c = self.culls[pool]
        if view not in c["expected"]:
            m2 = c["meshes"].copy()
            v = c["views"][view]
            exp = self.oracle.prepare_meshes(m2, c["tr"], self.e2t, v, hiz=c["hz"] if v.get("use_hiz") else None, ready=c["ready"])
            o = np.argsort(exp["visible_idx"], kind="stable")
            c["expected"][view] = dict(idx=exp["visible_idx"][o], model=exp["baked_model"][o], dist=exp["distance_sq"][o],
                                       count=exp["draw_count"], is_visible=m2["isVisible"].copy(), instances=exp["instance_count"])
        return c["expected"][view]

    def read(self, pool, view, write_back):
        """(visible_idx, baked_model, distance_sq) in the order the library delivers them, + is_visible (main pass)."""
        n = self.pools[pool].shape[0]
        got = self.vis.fetch(view, write_back=bool(write_back), occupancy=n, order="raw", pool_id=pool)
        if pool % 2 == 1 and got["draw_count"]:
            rec = self.vis.records(pool, view, RECORD_DTYPE)
            assert rec.shape[0] == got["draw_count"]
            if (pool, view) in self.targets:  # the fetch left them in the caller's own array
                mine = self.targets[(pool, view)][:rec.shape[0]]
                assert np.array_equal(mine.view(np.uint8), rec.view(np.uint8)), ("record target", pool, view)
            stride = self.pools[pool].dtype.itemsize
            assert np.all(rec["componentOffset"] % stride == 0)
            got["visible_idx"] = (rec["componentOffset"] // stride).astype(np.uint32)
            got["baked_model"], got["distance_sq"] = rec["bakedModel"].copy(), rec["distanceSq"].copy()
        return got

    def check_fetch(self, pool, view, write_back):
        c, exp = self.culls[pool], self.expected(pool, view)
        if not c["views"][view].get("emit_records", 1):  # count-only view: the count and the isVisible bytes
            got = self.vis.fetch(view, write_back=bool(write_back), occupancy=self.pools[pool].shape[0], order="raw", pool_id=pool)
            assert got["draw_count"] == exp["count"] and np.array_equal(got["is_visible"], exp["is_visible"]), (pool, view)
            return
        got = self.read(pool, view, write_back)
        assert got["draw_count"] == exp["count"], (pool, view, got["draw_count"], exp["count"])
        assert got["instance_count"] == exp["instances"], (pool, view, got["instance_count"], exp["instances"])
        order = c["sorted"].get(view)
        idx, model, dist = got["visible_idx"], got["baked_model"], got["distance_sq"]
        if exp["count"]:
            if order is not None:
                assert np.all(dist[:-1] >= dist[1:]) if order else np.all(dist[:-1] <= dist[1:]), (pool, view, order)
            k = np.argsort(idx, kind="stable")
            assert np.array_equal(idx[k], exp["idx"]), (pool, view)
            assert np.array_equal(model[k].view(np.uint32), exp["model"].view(np.uint32)), (pool, view)
            assert np.array_equal(dist[k].view(np.uint32), exp["dist"].view(np.uint32)), (pool, view)
            if order is None:  # unsorted results come in a deterministic order: mirror order, which a slot sort must not need twice
                assert np.unique(idx).shape[0] == idx.shape[0]
        if c["views"][view]["shadow_pass"] < 0:
            assert np.array_equal(got["is_visible"], exp["is_visible"]), (pool, view)

    def run(self, schedule):
        vis, rng, torch = self.vis, self.rng, self.torch
        for op, *a in schedule[1:]:
            if op == "begin":
                vis.cull_batch_begin()
            elif op == "end":
                vis.cull_batch_end()
            elif op == "wait":
                vis.wait()
            elif op == "sync":
                vis.sync()
            elif op == "cull":
                p, kinds = int(a[0]), a[1:]
                views = [self.view_of(k, i) for i, k in enumerate(kinds)]
                vis.cull(p, views)
                self.culls[p] = dict(meshes=self.pools[p].copy(), tr=self.tr.copy(), views=views, hz=self.hz, sorted={}, expected={},
                                     ready=self.ready[p].copy() if p in self.ready else None)
                self.last_pool = p
                if self.sweep_pending:  # GV_SWEEP_WITH_CULL[_VALU]: this cull also left the world matrices
                    self.sweep_pending = False
                    self.check_world()
            elif op == "sort":
                p, v, desc = int(a[0]), int(a[1]), int(a[2])
                vis.sort(v, descending=bool(desc), pool_id=p)
                self.culls[p]["sorted"][v] = bool(desc)
            elif op == "dirty_xf":
                first, count = int(a[0]), int(a[1])
                self.tr["position"][first:first + count, :3] += rng.normal(0, 40.0, (count, 3)).astype(np.float32)
                self.tr["selfActive"][first:first + count] ^= (rng.random(count) < 0.05).astype(np.uint8)
                vis.mark_dirty(0, first, count)
            elif op == "dirty_mesh":
                p, first, count = int(a[0]), int(a[1]), int(a[2])
                self.pools[p]["aabbMax"][first:first + count, :3] += np.float32(0.25)
                self.pools[p]["isEnabled"][first:first + count] ^= (rng.random(count) < 0.1).astype(np.uint8)
                vis.mark_dirty(2, first, count, pool_id=p)
            elif op == "move":
                p = int(a[0])
                self.pools[p] = self.pools[p].copy()
                vis.bind_pool(p, self.pools[p])
                self.culls.pop(p, None)
            elif op == "grow":
                p, extra = int(a[0]), int(a[1])
                old = self.pools[p]
                n = old.shape[0]
                new = np.concatenate([old, np.zeros(extra, old.dtype)])
                new["entity"][n:] = self.tr["entity"][n:n + extra]
                new["aabbMin"][n:, :3], new["aabbMax"][n:, :3], new["isEnabled"][n:] = -0.5, 0.5, 1
                self.pools[p] = new
                for key in [k for k in self.targets if k[0] == p]:  # the arrays are too small for the grown pool: let them go first
                    vis.set_record_target(key[0], key[1], None)
                    del self.targets[key]
                vis.bind_pool(p, new)
                if p in self.ready:
                    self.ready[p] = np.concatenate([self.ready[p], np.ones(extra, np.uint8)])
                    vis.bind_ready(p, self.ready[p])
                self.culls.pop(p, None)
            elif op == "move_xf":
                self.tr = self.tr.copy()
                vis.bind_transforms(self.tr, self.e2t)
                self.culls.clear()
            elif op == "hiz":
                self.depth = scene.synthetic_depth(int(a[0]), int(a[1]), seed=int(a[2]), rects=12)
                vis.hiz_build(self.depth)
                self.hz = self.oracle.Hiz(self.depth, rg16f=self.rg16f)
            elif op == "hiz_rebuild":
                vis.hiz_rebuild()
            elif op == "sweep":
                vis.sweep(int(a[0]))
                if int(a[0]) in (2, 3):
                    self.sweep_pending = True
                else:
                    self.sweep_pending = False
                    self.check_world()
            elif op == "rebuild":
                vis.hierarchy_rebuild()
                self.culls.clear()
            elif op == "reparent":
                first, count = int(a[0]), int(a[1])
                for s_ in range(first, first + count):  # towards a lower slot: no cycles; chains grow
                    self.tr["parent"][s_] = self.tr["entity"][int(rng.integers(0, s_))]
                vis.mark_dirty(1, first, count)
            elif op == "fetch":
                self.check_fetch(int(a[0]), int(a[1]), int(a[2]))
                self.readers += 1
            elif op == "count":
                n = pool_result_count(vis, int(a[0]), int(a[1]))
                assert n == self.expected(int(a[0]), int(a[1]))["count"]
                self.readers += 1
            elif op == "device":
                p, v = int(a[0]), int(a[1])
                d = dev_result(vis, p, v)
                word = device_words(torch, d.draw_count, 1)
                torch.cuda.synchronize()
                vis.wait()
                assert int(word.cpu()[0]) == self.expected(p, v)["count"]
                self.readers += 1
            elif op == "records":
                self.check_fetch(int(a[0]), int(a[1]), 0)
                self.readers += 1
            elif op == "bases":
                p, v = int(a[0]), int(a[1])
                bases = vis.instance_bases(p, v)
                exp = self.expected(p, v)
                if self.culls[p]["ready"] is None:
                    assert np.array_equal(bases, np.arange(exp["count"] + 1, dtype=np.uint32))
                else:  # first instance of every record, in the order the records are delivered
                    got = self.read(p, v, 0)
                    per = self.culls[p]["ready"][got["visible_idx"]].astype(np.uint64) if exp["count"] else np.zeros(0, np.uint64)
                    assert np.array_equal(bases.astype(np.uint64), np.concatenate([[0], np.cumsum(per)]))
                    assert int(bases[-1]) == exp["instances"]
                self.readers += 1
            elif op in ("shard", "mask"):
                p = self.last_pool
                if p not in self.culls:
                    continue
                exp = self.expected(p, 0)
                n = self.pools[p].shape[0]
                if op == "shard":
                    buf = torch.full((n + 1,), -1, dtype=torch.int32, device="cuda:0")
                    torch.cuda.synchronize()  # (the fill runs on torch's stream; the library's stream is non-blocking: no implicit order between them)
                    vis.copy_shard_device(0, buf.data_ptr(), n, index_base=7)
                    vis.wait()
                    host = buf.cpu().numpy().view(np.uint32)
                    assert host[0] == exp["count"] and np.array_equal(np.sort(host[1:1 + host[0]]), exp["idx"] + 7)
                else:
                    words = (n + 31) // 32
                    buf = torch.zeros(words + 1, dtype=torch.int32, device="cuda:0")
                    torch.cuda.synchronize()  # (the fill runs on torch's stream; the library's stream is non-blocking: no implicit order between them)
                    vis.copy_mask_device(0, buf.data_ptr(), words)
                    vis.wait()
                    host = buf.cpu().numpy().view(np.uint32)
                    bits = np.unpackbits(host[1:].view(np.uint8), bitorder="little")[:n]
                    slots = np.sort(vis.mirror_slots(p, n)[np.flatnonzero(bits)])
                    assert host[0] == exp["count"] and np.array_equal(slots, exp["idx"])
                self.readers += 1
            elif op == "ready":
                p, first, count = int(a[0]), int(a[1]), int(a[2])
                self.ready[p][first:first + count] = rng.choice(np.array([0, 1, 1, 2, 3], np.uint8), count)
                vis.mark_dirty(2, first, count, pool_id=p)
                self.culls.pop(p, None)
            elif op == "target":
                p, v, on = int(a[0]), int(a[1]), int(a[2])
                if on:
                    arr = np.zeros(self.pools[p].shape[0], RECORD_DTYPE)
                    vis.set_record_target(p, v, arr)
                    self.targets[(p, v)] = arr
                else:
                    vis.set_record_target(p, v, None)
                    self.targets.pop((p, v), None)
            elif op in ("exch", "exchp"):
                p = self.last_pool if op == "exch" else int(a[0])
                if p not in self.culls or not self.exchange:
                    continue
                exp = self.expected(p, 0)
                f = vis.exchange_acquire(vis.exchange_visible(0, index_base=11, pool_id=None if op == "exch" else p)["frame"])
                vis.wait()  # (the acquire ordered the context's stream behind the rows)
                row = device_words(torch, f["ptr"], f["row_words"]).cpu().numpy().view(np.uint32)
                assert f["complete"] and f["counts"] == [exp["count"]] and row[0] == exp["count"]
                # whatever the previous frames predicted, an acquired frame holds the whole list
                assert np.array_equal(np.sort(row[1:1 + row[0]]), exp["idx"] + 11)
                self.readers += 1
            else:
                raise AssertionError(f"unknown schedule operation {op}")


def pool_result_count(vis, pool, view):
    import ctypes
    n = ctypes.c_uint32()
    vis._check(vis.lib.gv_pool_result_count(vis.ctx, pool, view, ctypes.byref(n)))
    return n.value


def dev_result(vis, pool, view):
    import ctypes
    from garden_amd.lib import GvDeviceResult
    d = GvDeviceResult()
    vis._check(vis.lib.gv_pool_results_device(vis.ctx, pool, view, ctypes.byref(d)))
    return d


def device_words(torch, ptr, count):
    class _Span:
        pass
    span = _Span()
    span.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<i4", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(span, device="cuda:0")


@pytest.mark.parametrize("block", range(25))
def test_random_schedules_of_held_back_work_match_the_oracle(oracle, block):
    """200 random schedules (25 blocks of 8; tests/schedules.py) interleaving gv_cull_batch_begin / _end, recorded culls, deferred
    small-pool sorts, dirty marks, moved / grown pools, pyramid builds, sweeps and readers of every kind, in two context
    configurations; after EVERY reader the result equals the oracle's for the state at the gv_cull it reads."""
    from garden_amd.lib import GpuVisibility
    import schedules  # tests/schedules.py (the tests directory is on sys.path under pytest)
    readers = 0
    for k in range(8):
        seed = block * 8 + k
        schedule = schedules.generate(seed)
        with GpuVisibility(device=0, keep_slot_order=bool(seed & 1)) as vis:
            replay = ScheduleReplay(vis, oracle, schedule, seed)
            try:
                replay.run(schedule)
            except AssertionError as e:
                raise AssertionError(f"schedule {seed}: {e}\n{schedules.to_text(schedule)}") from e
            readers += replay.readers
    assert readers > 40


@pytest.mark.parametrize("seed", [290, 257, 830, 960])
def test_schedules_that_once_failed(oracle, seed):
    """Found by tools/schedule_soak.py (1 500 schedules, round 4; kept as text: tests/golden/schedule_<seed>.txt): a recorded cull
    that is launched later — by a reader of ANOTHER pool's results — turned the view-indexed calls (gv_results_copy_shard_device /
    _mask_device ...) towards its own pool, because the launch, not gv_cull, set "the pool of the most recent gv_cull". Schedule 960: a cull recorded BEFORE
    gv_sweep(GV_SWEEP_WITH_CULL) took the request when it was launched later, and the cull it was meant for left no world matrices."""
    import os
    from garden_amd.lib import GpuVisibility
    import schedules
    path = os.path.join(os.path.dirname(__file__), "golden", f"schedule_{seed}.txt")
    schedule = schedules.from_text(open(path).read())
    with GpuVisibility(device=0, keep_slot_order=bool(seed & 1), block_bounds=bool(seed % 5 == 3)) as vis:
        replay = ScheduleReplay(vis, oracle, schedule, seed)
        replay.run(schedule)
    assert replay.readers > 5
