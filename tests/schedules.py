"""Random call schedules over the C-ABI's "held back" mechanisms: recorded culls of a batched tick
(gv_cull_batch_begin / _end), sorts of small pools deferred to the first reader, published results of sibling views, and
everything that has to flush them in the right order — dirty marks, re-binds (moved / grown pools), pyramid builds, sweeps,
readers of every kind. ONE generator, two replayers:

  * tests/test_gpu_fuzz.py replays a schedule on the GPU through ctypes and checks every reader's result against the CPU
    oracle evaluated on the state of the pools / pyramid AT THE TIME OF THE gv_cull it reads (a recorded cull sees the pools as
    they were when it was recorded, include/garden_vis.h);
  * tests/cpp/host_orchestration_test.cpp replays the same text under AddressSanitizer + UBSan against the HIP stub (kernels
    are no-ops there: what is checked is memory safety and status codes of the host orchestration).

Text form, one operation per line (first line: `world <transforms> <pool sizes...>`):
    begin | end | wait | sync | rebuild
    cull <pool> <kind>...          kind: m main camera, h main camera + Hi-Z, s shadow cascade, u UI-like 2-D key, c main camera
                                   count-only (emit_records = 0: isVisible + count, no records) (1-3 views)
    reparent <first> <count>       these transform slots were re-parented (setParent towards a lower slot): ranged GV_DIRTY_HIERARCHY
    sort <pool> <view> <0|1>       ascending | descending
    dirty_xf <first> <count>       the caller edited these transform slots (the replayer edits, then marks)
    dirty_mesh <pool> <first> <count>
    move <pool>                    the pool's storage moved (same contents, new address): re-bind
    grow <pool> <extra>            slots appended (entities created): re-bind with the larger occupancy
    move_xf                        the transform pool's storage moved: re-bind
    hiz <w> <h> <seed> | hiz_rebuild
    sweep <mode>                   GvSweepMode 0, 1, 4; 2, 3: the NEXT cull also produces the world matrices (checked behind it)
    fetch <pool> <view> <0|1>      gv_pool_results_fetch (write_back)
    count <pool> <view>            gv_pool_result_count
    device <pool> <view>           gv_pool_results_device
    records <pool> <view>          gv_pool_results_records (pools with a record layout: every odd pool id)
    bases <pool> <view>            gv_pool_results_instance_bases
    shard | mask                   gv_results_copy_shard_device / _mask_device of view 0 of the most recently culled pool
    ready <pool> <first> <count>   per-slot ready counts changed (pools with id % 4 == 2 carry a ready column, gv_pool_bind_ready)
    target <pool> <view> <0|1>     gv_pool_set_record_target: the caller's own array for the view's records (1) / removed (0);
                                   pools with a record layout only (odd ids)
    exchp p                        gv_pool_exchange_visible of view 0 of pool p — culled some time ago, other pools culled since
    exch                           gv_exchange_visible of view 0 of the most recently culled pool (1-rank communicator; contexts of
                                   schedules whose first line ends in `x`)
First line: `world <transforms> <pool sizes...> [x]`.
"""
import numpy as np

VIEW_KINDS = "mhsu"


def generate(seed, ops=60):
    """A schedule as a list of tuples (name, args...); the first is ("world", n_transforms, pool sizes...)."""
    rng = np.random.Generator(np.random.PCG64(0x5C4ED + seed))
    n_pools = int(rng.integers(2, 5))
    # sizes on both sides of every threshold the held-back paths look at: 16384 (batched sorts), 32768 (recorded culls, deferred
    # sorts, one-launch cull + emit); one pool may be larger (always launched at once)
    choices = [300, 2500, 9000, 16384, 16390, 20000, 32768, 33000, 70000]
    sizes = [int(rng.choice(choices)) for _ in range(n_pools)]
    if rng.random() < 0.15:  # above the size at which pools at rest get block bounds by default (262 144 slots)
        sizes[int(rng.integers(0, n_pools))] = 270000
    n_xf = max(sizes) + int(rng.integers(0, 500))
    exchange = rng.random() < 0.2
    out = [("world", n_xf, *sizes, "x")] if exchange else [("world", n_xf, *sizes)]
    culled = {}  # pool -> list of view kinds of its last cull (still valid)
    batching = False
    last_pool = None
    hiz = False

    def pick_culled():
        return int(rng.choice(sorted(culled))) if culled else None

    for _ in range(ops):
        r = rng.random()
        if r < 0.06:
            out.append(("end",) if batching else ("begin",))
            batching = not batching
        elif r < 0.34:
            p = int(rng.integers(0, n_pools))
            nviews = int(rng.choice([1, 1, 2, 3]))
            kinds = []
            for v in range(nviews):
                if v == 0:
                    kinds.append("h" if hiz and rng.random() < 0.4 else ("u" if rng.random() < 0.15 else ("c" if rng.random() < 0.1 and nviews == 1 else "m")))
                else:
                    kinds.append("s")
            out.append(("cull", p, *kinds))
            culled[p] = kinds
            last_pool = p
        elif r < 0.44 and culled:
            p = pick_culled()
            if culled[p][0] != "c":
                out.append(("sort", p, int(rng.integers(0, len(culled[p]))), int(rng.integers(0, 2))))
        elif r < 0.52:
            first = int(rng.integers(0, n_xf))
            most = n_xf // 2 if rng.random() < 0.12 else 4000  # now and then a range long enough for the device-side gather of AoS ranges
            out.append(("dirty_xf", first, int(rng.integers(1, min(n_xf - first, most) + 1))))
        elif r < 0.59:
            p = int(rng.integers(0, n_pools))
            first = int(rng.integers(0, sizes[p]))
            most = sizes[p] // 2 if rng.random() < 0.12 else 3000
            out.append(("dirty_mesh", p, first, int(rng.integers(1, min(sizes[p] - first, most) + 1))))
        elif r < 0.62:
            p = int(rng.integers(0, n_pools))
            out.append(("move", p))
            culled.pop(p, None)  # results of a re-bound pool are not read (the next cull replaces them)
        elif r < 0.64:
            p = int(rng.integers(0, n_pools))
            extra = int(rng.integers(1, 200))
            if sizes[p] + extra <= n_xf:
                out.append(("grow", p, extra))
                sizes[p] += extra
                culled.pop(p, None)
        elif r < 0.66:
            out.append(("move_xf",))
            culled.clear()
        elif r < 0.71:
            if hiz and rng.random() < 0.5:
                out.append(("hiz_rebuild",))
            else:
                out.append(("hiz", int(rng.choice([64, 96, 256])), int(rng.choice([64, 80, 128])), int(rng.integers(0, 1000))))
                hiz = True
        elif r < 0.74:
            out.append(("sweep", int(rng.choice([0, 1, 4, 4, 2, 3]))))
        elif r < 0.76:
            out.append((str(rng.choice(["wait", "sync", "rebuild"])),))
            if out[-1][0] == "rebuild":
                culled.clear()
        elif r < 0.775:
            first = int(rng.integers(1, n_xf - 1))
            out.append(("reparent", first, int(rng.integers(1, min(n_xf - first, 24) + 1))))
        elif r < 0.79:
            with_ready = [p for p in range(n_pools) if p % 4 == 2]
            if with_ready:
                p = int(rng.choice(with_ready))
                first = int(rng.integers(0, sizes[p]))
                out.append(("ready", p, first, int(rng.integers(1, min(sizes[p] - first, 2000) + 1))))
                culled.pop(p, None)  # instance counts / bases are summed from the column as it stands at the fetch: results of this
                                     # pool are read before its counts change or after its next cull (include/garden_vis.h)
        elif r < 0.805:
            odd = [p for p in range(n_pools) if p % 2 == 1]
            if odd:
                out.append(("target", int(rng.choice(odd)), int(rng.integers(0, 3)), int(rng.integers(0, 2))))
        elif culled:
            p = pick_culled()
            v = int(rng.integers(0, len(culled[p])))
            kind = rng.random()
            if culled[p][0] == "c":  # count-only: the count and the bytes are all there is
                kind = min(kind, 0.59)
            if kind < 0.45:
                out.append(("fetch", p, v, int(rng.integers(0, 2))))
            elif kind < 0.6:
                out.append(("count", p, v))
            elif kind < 0.7:
                out.append(("device", p, v))
            elif kind < 0.8 and p % 2 == 1:
                out.append(("records", p, v))
            elif kind < 0.88:
                out.append(("bases", p, v))
            elif exchange and kind >= 0.94:
                out.append(("exchp", p))
            elif last_pool in culled and culled[last_pool][0] != "c":
                out.append((str(rng.choice(["shard", "mask", "exch"] if exchange else ["shard", "mask"])),))
    if batching:
        out.append(("end",))
    for p in sorted(culled):  # every result still standing is read at the end
        for v in range(len(culled[p])):
            out.append(("fetch", p, v, 1))
    return out


def to_text(schedule):
    return "\n".join(" ".join(str(a) for a in op) for op in schedule) + "\n"


def from_text(text):
    """The inverse of to_text (numbers become ints): schedules kept as fixtures (tests/golden/schedule_*.txt)."""
    out = []
    for line in text.splitlines():
        parts = line.split()
        if parts and not parts[0].startswith("#"):
            out.append(tuple(int(a) if a.lstrip("-").isdigit() else a for a in parts))
    return out
