"""Multi-GPU exchange step of the visibility pass: one process per GPU, entities sharded by spatial tile,
no collective on the data path except the all-gatherv of the compacted visible-index lists
(SURVEY.md §8e). RCCL has no native v-variant: counts are exchanged with one small all-gather, payloads
either with torch's uneven all_gather (ProcessGroupNCCL lowers it to one grouped broadcast per root — each
shard then travels over its own xGMI link instead of around a ring) or, on backends without it (gloo in the
CPU tests), with one broadcast per root.
"""
import torch

_uneven_all_gather_ok = True


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
    global _uneven_all_gather_ok
    if backend == "nccl" and _uneven_all_gather_ok and min(counts_h) > 0:  # empty shards (a tile behind the camera)
        # go through the per-root path below, which simply skips them
        try:
            dist.all_gather(pieces, idx_buf[:count], group=group)  # uneven sizes -> grouped per-root broadcasts
            return out, counts
        except (RuntimeError, ValueError):
            _uneven_all_gather_ok = False  # this torch build wants equal sizes: one broadcast per root instead
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
