"""Multi-GPU exchange step of the visibility pass: one process per GPU, entities sharded by spatial tile,
no collective on the data path except the all-gatherv of the compacted visible-index lists
(SURVEY.md §8e). RCCL has no native v-variant: counts are exchanged with one small all-gather, payloads
with an equal-size all-gather of shards padded to the largest count (nccl) or one broadcast per root (gloo in the
CPU tests).

Per-frame use goes through `VisibleListExchange`: fixed-capacity padded shards `[count, idx ...]` gathered with ONE
equal-size all-gather and no host synchronisation (the counts travel in the headers and are checked one round trip
later), so a rank's next frame is culled while its last list is still on the links. `allgatherv_indices` is the
exact-size, host-synchronising form (setup, validation, irregular use).
"""
import torch

def tile_of_positions(positions, side, grid):
    """Spatial tile id of each root position for a `grid` = (gx, gy, gz) cut of the world cube."""
    import numpy as np
    t = np.zeros(positions.shape[0], dtype=np.int64)
    mul = 1
    for a in range(3):
        c = np.clip(((positions[:, a] / side + 0.5) * grid[a]).astype(np.int64), 0, grid[a] - 1)
        t += c * mul
        mul *= grid[a]
    return t


def tile_grid(n):
    """(gx, gy, gz) with gx * gy * gz >= n, doubling x, y, z in turn: 8 GPUs -> 2 x 2 x 2."""
    g = [1, 1, 1]
    i = 0
    while g[0] * g[1] * g[2] < n:
        g[i % 3] *= 2
        i += 1
    return tuple(g)


def cell_owners(grid, ranks):
    """Which rank owns each cell of a `grid` = (gx, gy, gz) cut: cells are put in Morton (Z-curve) order of their
    (x, y, z) coordinates and dealt to the ranks in rounds — the k-th cell of that order goes to rank
    (k + h(k // ranks)) % ranks, h(b) = (b * 2654435761 mod 2^32) >> 16: every round of `ranks` consecutive cells (a compact
    block of space) gives one cell to every rank, and the rotation by h changes from round to round, so no rank always gets the
    same corner of its blocks (plain k % ranks does exactly that — with 8 ranks, the cells of one parity class — and a camera
    at a lattice point then favours half the ranks by 40 %; measured). Whatever part of the world a view looks at is shared
    out evenly. The reference's split is even by construction: equal contiguous index ranges, source/thread-pool.cpp:180-194 — a
    spatial split needs many more cells than ranks to get there. Returns int64[gx * gy * gz] indexed by the linear cell id
    x + y * gx + z * gx * gy. The same rule as gv_scene_extract_rank (garden_amd/csrc/gv_scene.cpp)."""
    import numpy as np
    gx, gy, gz = (int(g) for g in grid)
    ranks = int(ranks)
    lin = np.arange(gx * gy * gz, dtype=np.int64)
    cx, cy, cz = lin % gx, (lin // gx) % gy, lin // (gx * gy)
    code = np.zeros(lin.shape[0], dtype=np.int64)
    for b in range(12):  # up to 4096 cells per axis
        code |= ((cx >> b) & 1) << (3 * b) | ((cy >> b) & 1) << (3 * b + 1) | ((cz >> b) & 1) << (3 * b + 2)
    order = np.argsort(code, kind="stable")
    k = np.arange(lin.shape[0], dtype=np.int64)
    turn = (((k // ranks) * 2654435761) & 0xFFFFFFFF) >> 16
    owner = np.empty(lin.shape[0], dtype=np.int64)
    owner[order] = (k + turn) % ranks
    return owner


def cell_grid(ranks, cells_per_rank=512):
    """(gx, gy, gz) of at least ranks * cells_per_rank cells, doubling x, y, z in turn: 8 ranks -> 16 x 16 x 16 = 4096 cells
    (measured on the cfg5 shape, three views: max / mean of the ranks' visible counts 1.03 with 512 cells per rank, 1.11 with 64)."""
    return tile_grid(int(ranks) * int(cells_per_rank))


class WorldPartition:
    """ONE world cut into spatial tiles (the reference's nearest analogue is the contiguous range split of
    ThreadPool::addItems, source/thread-pool.cpp:173-200 — here the split is by space, so that a tile is culled as a
    unit and whole trees stay together).

    tiles[t]            Scene of tile t: its transforms and meshes in ascending global slot order, entity ids
                        renumbered 1..k per tile (free slots stay 0), parents remapped, entity_to_transform rebuilt
    transform_global[t] local transform slot -> global transform slot        (int64)
    mesh_global[t]      local mesh slot -> global mesh slot                  (int64)  <- maps a tile's visible_idx back
    mesh_tile, mesh_local   global mesh slot -> (tile, local slot)
    transform_tile, transform_local   the same for transform slots
    root_slot           global transform slot -> slot of its root ancestor (itself for roots)
    With `ranks`: a "tile" is everything ONE RANK owns — the cells cell_owners() deals to it, concatenated into one pool
    (one cull launch per rank; the mirror's spatial order and block bounds keep the locality within it).
    """

    def __init__(self, tiles, transform_global, mesh_global, mesh_tile, mesh_local, transform_tile, transform_local,
                 root_slot, grid, side, ranks=None):
        self.tiles, self.transform_global, self.mesh_global = tiles, transform_global, mesh_global
        self.mesh_tile, self.mesh_local = mesh_tile, mesh_local
        self.transform_tile, self.transform_local = transform_tile, transform_local
        self.root_slot, self.grid, self.side, self.ranks = root_slot, grid, side, ranks

    def to_global(self, tile, local_mesh_slots):
        """Global mesh slots of a tile's local visible_idx list."""
        import numpy as np
        return self.mesh_global[tile][np.asarray(local_mesh_slots, dtype=np.int64)]


def partition_world(sc, grid, side=None, ranks=None):
    """Cuts the Scene `sc` (garden_amd.scene.Scene: AoS pools + entity_to_transform) into spatial tiles:
    every ROOT transform goes to the cell of the `grid` its position falls in (tile_of_positions), every descendant follows
    its root (a parent chain is never cut, so a tile computes the same world matrices as the whole world does), a mesh follows
    the transform of its entity. Free slots and meshes whose entity has no transform go to tile 0 (with `ranks`: to rank
    slot % ranks), where the cull filters them out exactly as it does in the whole world (mesh.cpp:140-155). Within a tile,
    slots keep their global order and entity ids are renumbered from 1.
    ranks=None: one tile per cell (prod(grid) tiles). ranks=R: R tiles, tile r = all the cells cell_owners(grid, R) gives rank
    r — use a grid of many more cells than ranks (cell_grid) so that every rank holds a share of every region.
    `side`: edge of the world cube the grid cuts (default 100 * N^(1/3), the synthetic scenes' cube)."""
    import numpy as np

    from .pools import GV_NONE
    from .scene import Scene
    tr, ms, e2t = sc.transforms, sc.meshes, np.asarray(sc.entity_to_transform, dtype=np.uint32)
    nt, nm = tr.shape[0], ms.shape[0]
    if side is None:
        side = 100.0 * max(nm, 1) ** (1.0 / 3.0)
    ntiles = int(grid[0] * grid[1] * grid[2]) if ranks is None else int(ranks)

    def slot_of_entity(ent):
        """transform slot of each entity id (-1: none) — Manager::tryGet<TransformComponent>"""
        ent = np.asarray(ent, dtype=np.int64)
        ok = (ent > 0) & (ent < e2t.shape[0])
        out = np.full(ent.shape, -1, dtype=np.int64)
        s = e2t[ent[ok]].astype(np.int64)
        s[s == GV_NONE] = -1
        out[ok] = s
        return out

    # root ancestor of every transform slot by pointer jumping (chains are short; cycles cannot occur in a valid pool)
    parent_slot = slot_of_entity(tr["parent"])
    parent_slot[tr["entity"] == 0] = -1
    root = np.arange(nt, dtype=np.int64)
    up = np.where(parent_slot >= 0, parent_slot, root)
    for _ in range(64):
        nxt = up[up]
        if np.array_equal(nxt, up):
            break
        up = nxt
    else:
        raise ValueError("partition_world: parent chains deeper than 2^64 or cyclic")
    root = up
    tile_of_root = tile_of_positions(tr["position"][:, :3].astype(np.float64), side, grid)
    if ranks is not None:
        tile_of_root = cell_owners(grid, ranks)[tile_of_root]
    xf_tile = tile_of_root[root]
    mesh_xf = slot_of_entity(ms["entity"])
    if ranks is None:
        xf_tile[tr["entity"] == 0] = 0  # free transform slots: anywhere; tile 0
        mesh_tile = np.where(mesh_xf >= 0, xf_tile[np.maximum(mesh_xf, 0)], 0).astype(np.int64)
    else:  # dealt out by slot number: a per cent of a 10^8-entity world is a tenth of a rank's share
        free = tr["entity"] == 0
        xf_tile[free] = np.nonzero(free)[0] % int(ranks)
        mesh_tile = np.where(mesh_xf >= 0, xf_tile[np.maximum(mesh_xf, 0)], np.arange(nm, dtype=np.int64) % int(ranks)).astype(np.int64)

    tiles, xf_global, mesh_global = [], [], []
    xf_local = np.zeros(nt, dtype=np.int64)
    mesh_local = np.zeros(nm, dtype=np.int64)
    for t in range(ntiles):
        xs = np.nonzero(xf_tile == t)[0]
        msl = np.nonzero(mesh_tile == t)[0]
        xf_local[xs] = np.arange(xs.shape[0])
        mesh_local[msl] = np.arange(msl.shape[0])
        ltr, lms = tr[xs].copy(), ms[msl].copy()
        # new entity ids: live transforms first (1..k in slot order), then meshes whose entity has no transform here
        live = ltr["entity"] != 0
        new_id = np.zeros(e2t.shape[0] + 1, dtype=np.uint32)  # global entity id -> tile-local id
        k = int(live.sum())
        new_id[ltr["entity"][live]] = np.arange(1, k + 1, dtype=np.uint32)
        stray = (lms["entity"] != 0) & (new_id[np.minimum(lms["entity"], e2t.shape[0])] == 0)
        stray_ids = np.unique(lms["entity"][stray])
        new_id[np.minimum(stray_ids, e2t.shape[0])] = np.arange(k + 1, k + 1 + stray_ids.shape[0], dtype=np.uint32)
        old_parent = ltr["parent"].copy()
        ltr["entity"] = new_id[np.minimum(ltr["entity"], e2t.shape[0])]
        # a parent that has no transform (dangling id) stays dangling: give it an id nothing maps to
        has_parent = old_parent != 0
        mapped = new_id[np.minimum(old_parent, e2t.shape[0])]
        dangling = has_parent & (mapped == 0)
        ltr["parent"] = mapped
        lms["entity"] = new_id[np.minimum(lms["entity"], e2t.shape[0])]
        cap = k + 1 + stray_ids.shape[0] + (1 if dangling.any() else 0)
        if dangling.any():
            ltr["parent"][dangling] = cap - 1  # an entity id with no transform: the chain ends there, as in the world
        le2t = np.full(cap, GV_NONE, dtype=np.uint32)
        le2t[ltr["entity"][live]] = np.nonzero(live)[0].astype(np.uint32)
        tiles.append(Scene(lms, ltr, le2t))
        xf_global.append(xs)
        mesh_global.append(msl)
    return WorldPartition(tiles, xf_global, mesh_global, mesh_tile, mesh_local, xf_tile, xf_local, root, tuple(grid), side, ranks)


def allgatherv_indices(idx_buf, count, dist, group=None):
    """idx_buf[:count] holds this rank's global visible indices (int32 view of uint32). Returns
    (gathered 1-D tensor of all ranks' lists in rank order, counts tensor)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    backend = dist.get_backend(group)
    if backend != "nccl" and idx_buf.is_cuda:
        # test path only (gloo has no device collectives): stage through host memory
        out, counts = allgatherv_indices(idx_buf[:count].cpu(), count, dist, group)
        return out.to(idx_buf.device), counts.to(idx_buf.device)
    dev = idx_buf.device
    my_count = torch.tensor([count], dtype=torch.int64, device=dev)
    counts = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, my_count, group=group)
    counts_h = counts.cpu().tolist()
    total = int(sum(counts_h))
    out = torch.empty(total, dtype=idx_buf.dtype, device=dev)
    offs = [0]
    for c in counts_h:
        offs.append(offs[-1] + int(c))
    pieces = [out[offs[r]:offs[r + 1]] for r in range(world)]
    if backend == "nccl":
        # RCCL has no v-variant: pad every shard to the largest count and use the plain equal-size all-gather (the most
        # exercised collective there is), then cut the padding off. Shards are a few MB at most.
        cap = max(1, int(max(counts_h)))
        padded = torch.empty(world * cap, dtype=idx_buf.dtype, device=dev)
        mine = torch.zeros(cap, dtype=idx_buf.dtype, device=dev)
        if count:
            mine[:count].copy_(idx_buf[:count])
        dist.all_gather_into_tensor(padded, mine, group=group)
        rows = padded.view(world, cap)
        for r in range(world):
            if counts_h[r]:
                pieces[r].copy_(rows[r, :counts_h[r]])
        return out, counts
    if count:
        pieces[rank].copy_(idx_buf[:count])
    works = []
    for r in range(world):
        if counts_h[r]:
            works.append(dist.broadcast(pieces[r], src=dist.get_global_rank(group, r) if group else r,
                                        group=group, async_op=True))
    for w in works:
        w.wait()
    return out, counts


class _WorkSet:
    """Several async works waited on as one (the direct exchange patterns issue one per peer / per root)."""

    def __init__(self, works):
        self.works = list(works)

    def wait(self):
        for w in self.works:
            w.wait()


class ShardOverflow(RuntimeError):
    """A rank produced more visible indices than the padded shard holds; `needed` is the largest count seen."""

    def __init__(self, needed, capacity):
        super().__init__(f"visible list of {needed} indices does not fit the exchange shard capacity {capacity}")
        self.needed = needed


def shard_capacity(max_count, slack=1.25, quantum=1024):
    """Padded shard capacity for lists of up to `max_count` indices: the same on every rank (all ranks know all
    counts of the previous exchange), with head-room for frame-to-frame change."""
    return (int(max_count * slack) // quantum + 1) * quantum


def mask_words(slots):
    """uint32 words of a bit-per-slot shard for a pool of `slots` slots (gv_results_copy_mask_device)."""
    return (int(slots) + 31) // 32


def pack_mask_shard(visible, entries):
    """CPU restatement of gv_results_copy_mask_device: int32[1 + mask_words(entries)] = [count, bits...] over the ids in
    `visible` (mirror entries on the device; any id space in the tests)."""
    import numpy as np
    v = np.asarray(visible, dtype=np.int64)
    words = np.zeros(mask_words(entries), dtype=np.uint32)
    np.bitwise_or.at(words, v >> 5, (np.uint32(1) << (v & 31).astype(np.uint32)))
    return torch.from_numpy(np.concatenate([[np.uint32(v.size)], words]).astype(np.uint32).view(np.int32))


def expand_mask_rows(padded, entries_per_rank, entry_tables=None, index_bases=None):
    """Global index lists from gathered mask shards: padded = [world, 1 + words] (int32 view of uint32), row r = rank r's
    [count, bits over its mirror entries]. entry_tables[r] (gv_pool_mirror_slots of rank r, exchanged once per mirror
    rebuild; None: identity) maps entries to pool slots, index_bases[r] (default r * entries_per_rank: contiguous tiles) is
    added. Returns (indices in rank order, ascending within a rank; counts from the headers)."""
    import numpy as np
    rows = padded.cpu().numpy().view(np.uint32)
    world = rows.shape[0]
    out, counts = [], []
    for r in range(world):
        bits = np.unpackbits(rows[r, 1:].view(np.uint8), bitorder="little")[:entries_per_rank]
        ids = np.flatnonzero(bits).astype(np.int64)
        if entry_tables is not None and entry_tables[r] is not None:
            ids = np.sort(np.asarray(entry_tables[r], dtype=np.int64)[ids])
        base = r * entries_per_rank if index_bases is None else int(index_bases[r])
        out.append(ids + base)
        counts.append(int(rows[r, 0]))
    return np.concatenate(out) if out else np.zeros(0, np.int64), np.asarray(counts, dtype=np.int64)


class VisibleListExchange:
    """Sync-free all-gather of the per-tile visible lists (one instance per process).

    frame loop:   shard = ex.next_shard()                       # int32[1 + capacity], device (or CPU in tests)
                  gv_results_copy_shard_device(..., shard)      # header = count, body = global indices
                  padded = ex.exchange()                        # [world, 1 + capacity]; row r = rank r's shard
    `padded` is valid in stream order for device consumers (the collective is enqueued behind the producer stream);
    host consumers call `compact()`. Buffers rotate over `slots` frames; re-using a slot first
    orders the stream behind that slot's collective and checks its headers (already on the host by then): a count
    above the capacity raises ShardOverflow so the caller can re-exchange that frame with a larger capacity."""

    MODES = ("allgather", "p2p", "broadcast")

    def __init__(self, dist, device, capacity, stream=None, group=None, slots=2, mode="allgather", payload="indices",
                 capacities=None):
        """mode: how the shards travel. "allgather" = ONE equal-size all-gather; "p2p" = one group of send/recv pairs
        with every peer (batch_isend_irecv = ncclGroupStart ... ncclGroupEnd: each shard crosses exactly one xGMI link,
        all links at once); "broadcast" = one broadcast per root. Same rows in the same place either way — the node's
        fabric is point to point and fully connected, so ring vs direct is an A/B for real hardware."""
        assert mode in self.MODES, mode
        assert payload in ("indices", "mask"), payload
        self.mode = mode
        # "mask": shards are [count, one bit per pool slot] (gv_results_copy_mask_device; capacity = mask_words(slots)): a fixed
        # size whatever the view, so a shard cannot overflow and the header is only the count
        self.payload = payload
        # capacities[r] <= capacity: how much of rank r's shard actually travels in the "p2p" and "broadcast" patterns (every
        # rank passes the same list, e.g. shard_capacity of each rank's count in an earlier frame: all ranks know all counts
        # from the headers). The equal-size all-gather always moves `capacity` per rank; a tile behind the camera then costs
        # as much as the fullest one. Rows keep their fixed place in the result.
        self.capacities = None
        self.dist, self.group, self.device = dist, group, torch.device(device)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.backend = dist.get_backend(group)
        self.capacity, self.slots, self.frame = int(capacity), slots, 0
        if capacities is not None:
            assert len(capacities) == self.world and all(0 <= int(c) <= self.capacity for c in capacities), capacities
            self.capacities = [int(c) for c in capacities]
        self.native = self.backend == "nccl" and self.device.type == "cuda"
        self.stream = stream if self.native else None  # producer stream (torch.cuda.Stream / ExternalStream)
        self.side = torch.cuda.Stream(device=self.device) if self.native else None
        n = 1 + self.capacity
        self.shards = [torch.zeros(n, dtype=torch.int32, device=self.device) for _ in range(slots)]
        self.outs = [torch.zeros(self.world * n, dtype=torch.int32, device=self.device) for _ in range(slots)]
        self.works = [None] * slots
        self.headers = [torch.zeros(self.world, dtype=torch.int32, pin_memory=self.native) for _ in range(slots)]
        self.header_events = [None] * slots
        self.in_flight = [False] * slots
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)  # the zero fills ran on torch's stream; the producer stream is not ordered behind it

    def describe(self):
        return {"allgather": "one equal-size all-gather", "p2p": "one group of send/recv pairs with every peer",
                "broadcast": "one broadcast per root"}[self.mode]

    def _global(self, r):
        return self.dist.get_global_rank(self.group, r) if self.group is not None else r

    def _gather_rows(self, out, shard, async_op):
        """Fills out[r] with rank r's shard by the configured pattern; returns an object with .wait() (or None)."""
        n = shard.numel()
        rows = out.view(self.world, n)
        if self.mode == "allgather":
            return self.dist.all_gather_into_tensor(out, shard, group=self.group, async_op=async_op)
        rows[self.rank].copy_(shard)  # own row: a local copy in stream order
        live = (lambda r: 1 + self.capacities[r]) if self.capacities is not None else (lambda r: n)  # header + what travels
        works = []
        if self.mode == "p2p":
            ops = []
            for d in range(1, self.world):
                to, frm = (self.rank + d) % self.world, (self.rank - d) % self.world
                ops.append(self.dist.P2POp(self.dist.isend, shard[:live(self.rank)], self._global(to), self.group))
                ops.append(self.dist.P2POp(self.dist.irecv, rows[frm][:live(frm)], self._global(frm), self.group))
            works = self.dist.batch_isend_irecv(ops) if ops else []
        else:
            for r in range(self.world):
                works.append(self.dist.broadcast(rows[r][:live(r)], src=self._global(r), group=self.group, async_op=True))
        ws = _WorkSet(works)
        if not async_op:
            ws.wait()
            return None
        return ws

    def _producer(self):
        import contextlib
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def _retire(self, s):
        if not self.in_flight[s]:
            return
        if self.native:
            with self._producer():
                self.works[s].wait()  # stream-order only: the producer stream may now overwrite shard/out of slot s
            self.header_events[s].synchronize()  # enqueued a whole frame ago
        self.in_flight[s] = False
        worst = int(self.headers[s].max())
        if self.payload == "indices":
            if worst > self.capacity:
                raise ShardOverflow(worst, self.capacity)
            if self.capacities is not None and self.mode != "allgather":
                for r in range(self.world):  # a rank whose list outgrew the part of its shard that travels
                    if int(self.headers[s][r]) > self.capacities[r]:
                        raise ShardOverflow(int(self.headers[s][r]), self.capacities[r])

    def next_shard(self):
        s = self.frame % self.slots
        self._retire(s)
        return self.shards[s]

    def exchange(self):
        s = self.frame % self.slots
        self.frame += 1
        shard, out = self.shards[s], self.outs[s]
        padded = out.view(self.world, 1 + self.capacity)
        if self.native:
            with self._producer():
                self.works[s] = self._gather_rows(out, shard, async_op=True)
            with torch.cuda.stream(self.side):
                self.works[s].wait()  # side stream behind the collective; the producer stream is not held up
                self.headers[s].copy_(padded[:, 0], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.side)
                self.header_events[s] = ev
        else:  # gloo (CPU tests, or N ranks sharing one GPU): staged through host memory, synchronous
            src = shard.cpu() if shard.is_cuda else shard
            if self.mode == "allgather":
                pieces = [torch.empty_like(src) for _ in range(self.world)]
                self.dist.all_gather(pieces, src, group=self.group)
                host = torch.stack(pieces)
            else:
                host = torch.empty(self.world * src.numel(), dtype=src.dtype)
                self._gather_rows(host, src, async_op=False)
                host = host.view(self.world, src.numel())
            padded.copy_(host)
            self.headers[s].copy_(host[:, 0])
        self.in_flight[s] = True
        return padded

    def drain(self):
        """Retire every slot in flight (host-blocking): raises ShardOverflow if any frame did not fit."""
        for s in range(self.slots):
            if self.native and self.in_flight[s]:
                self.works[s].wait()
            self._retire(s)

    def compact(self, padded):
        """Dense (indices in rank order, counts) from a padded result; host-synchronising (validation, CPU consumers)."""
        if self.native:
            torch.cuda.synchronize(self.device)
        counts = padded[:, 0].to(torch.int64).cpu()
        parts = [padded[r, 1:1 + int(counts[r])] for r in range(self.world)]
        return (torch.cat(parts) if parts else padded.new_zeros(0)), counts
