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


class ShardOverflow(RuntimeError):
    """A rank produced more visible indices than the padded shard holds; `needed` is the largest count seen."""

    def __init__(self, needed, capacity):
        super().__init__(f"visible list of {needed} indices does not fit the exchange shard capacity {capacity}")
        self.needed = needed


def shard_capacity(max_count, slack=1.25, quantum=1024):
    """Padded shard capacity for lists of up to `max_count` indices: the same on every rank (all ranks know all
    counts of the previous exchange), with head-room for frame-to-frame change."""
    return (int(max_count * slack) // quantum + 1) * quantum


class VisibleListExchange:
    """Sync-free all-gather of the per-tile visible lists (one instance per process).

    frame loop:   shard = ex.next_shard()                       # int32[1 + capacity], device (or CPU in tests)
                  gv_results_copy_shard_device(..., shard)      # header = count, body = global indices
                  padded = ex.exchange()                        # [world, 1 + capacity]; row r = rank r's shard
    `padded` is valid in stream order for device consumers (the collective is enqueued behind the producer stream);
    host consumers call `compact()`. Buffers rotate over `slots` frames; re-using a slot first
    orders the stream behind that slot's collective and checks its headers (already on the host by then): a count
    above the capacity raises ShardOverflow so the caller can re-exchange that frame with a larger capacity."""

    def __init__(self, dist, device, capacity, stream=None, group=None, slots=2):
        self.dist, self.group, self.device = dist, group, torch.device(device)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.backend = dist.get_backend(group)
        self.capacity, self.slots, self.frame = int(capacity), slots, 0
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
        if worst > self.capacity:
            raise ShardOverflow(worst, self.capacity)

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
                self.works[s] = self.dist.all_gather_into_tensor(out, shard, group=self.group, async_op=True)
            with torch.cuda.stream(self.side):
                self.works[s].wait()  # side stream behind the collective; the producer stream is not held up
                self.headers[s].copy_(padded[:, 0], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.side)
                self.header_events[s] = ev
        else:  # gloo (CPU tests, or N ranks sharing one GPU): staged through host memory, synchronous
            src = shard.cpu() if shard.is_cuda else shard
            pieces = [torch.empty_like(src) for _ in range(self.world)]
            self.dist.all_gather(pieces, src, group=self.group)
            host = torch.stack(pieces)
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
