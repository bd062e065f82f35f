"""The exchange step of bench.py --gpus N: every rank's compacted visible list to every rank, once per frame — through the
library's own C-ABI step (gv_exchange_visible: rows owned and sized by the library; the default and the headline) or through
torch.distributed over this script's buffers (garden_amd/multi.py) — with its checks against an exact all-gatherv, its set-up
(unique id, trial frame, travel-pattern probe, payload choice) and the figures the line reports about it."""
import os
import sys
import time

import numpy as np

from garden_amd.lib import GpuVisibility, GvError
from garden_amd.multi import VisibleListExchange, allgatherv_indices, shard_capacity, mask_words, expand_mask_rows

EXCHANGE_MODES = {"allgather": 0, "p2p": 1, "broadcast": 2}


class FrameExchange:
    def __init__(self, run, watchdog, root):
        self.run, self.watchdog, self.root = run, watchdog, root
        r = run
        self.idx_buf = r.torch.empty(r.n, dtype=r.torch.int32, device=r.device)
        self.ex = None            # VisibleListExchange (torch path), created once the shard capacity is known
        self.entry_tables = None  # --payload mask: every rank's mirror entry -> pool slot table
        self.native = False       # the exchange runs through the library's own C-ABI step (gv_exchange_visible / gv_exchange_masks)
        self.native_rows = None   # caller-owned rows [world, 1 + words] of the native bit-shard exchange
        self.sent_frame = None
        self.frames_acquired = self.frames_completed_late = 0  # acquired through the library's exchange / of those, completed by a second exchange
        self.exact = self.exact_counts = None
        self.gathered_total = None
        self.transport_note = self.payload_note = None
        self.path_fallback = os.environ.get("GV_BENCH_FALLBACK_REASON")
        self.mode = r.args.exchange or "allgather"  # the travel pattern of the timed frames
        self.mode_probe_ms = None                   # --exchange not given: what five frames of each pattern took (ms per frame)
        self.producer = r.lib_stream if r.backend == "nccl" else None

    # ---- the per-frame step ----
    def native_frame(self):
        """gv_exchange_visible for this frame, then the PREVIOUS frame acquired, the way a consumer one frame behind does: the send
        has settled that frame (a short row completed by a second exchange inside the call), so the acquire is a stream wait —
        every frame of the timed region is handed out complete."""
        r = self.run
        try:
            f = r.vis.exchange_visible(0, index_base=r.rank * r.n)
            if self.sent_frame is not None:
                acquired = r.vis.exchange_acquire(self.sent_frame)
                self.frames_acquired += 1
                self.frames_completed_late += 1 if acquired["cut_ranks"] else 0
        except GvError as e:
            # a status code from the library's exchange in the middle of the run (GV_E_TIMEOUT, GV_E_RCCL: it has never met real RCCL
            # with several ranks): like a collective that never returns, the line is handed to a child run through torch.distributed
            if r.world > 1 and not os.environ.get("GV_BENCH_FALLBACK_REASON"):
                self.watchdog.bark(error=str(e))
            raise
        self.sent_frame = f["frame"]
        return f

    def after_compute(self):
        """The exchange half of a frame (bench.py's step() = compute() + this). The rank's list goes out as a shard [count, indices...]
        and all ranks gather the shards enqueued behind the library's stream — no host synchronisation, so the next frame is culled
        while this one's list is still on the links."""
        r = self.run
        if self.native:
            if r.args.payload == "mask":
                r.vis.exchange_masks(0, mask_words(r.n), self.native_rows.data_ptr())
                return self.native_rows
            return self.native_frame()
        if self.ex is not None:
            shard = self.ex.next_shard()
            if r.args.payload == "mask":
                r.vis.copy_mask_device(0, shard.data_ptr(), self.ex.capacity)
            else:
                r.vis.copy_shard_device(0, shard.data_ptr(), self.ex.capacity, index_base=r.rank * r.n)
            return self.ex.exchange()
        return None

    # ---- checks against the exact all-gatherv ----
    def check_exchange(self):
        """Exact-size all-gatherv of one frame (host-synchronising form): all ranks hold the same concatenated list;
        every index lies in its owner's tile range; own shard == local visible list. Sizes the padded shards."""
        r = self.run
        n, rank, world = r.n, r.rank, r.world
        r.compute()
        r.vis.copy_idx_device(0, self.idx_buf.data_ptr(), n, index_base=rank * n)
        count = r.vis.result_count(0)  # 4-byte readback on the library's stream: also fences the copy above
        gathered, counts = allgatherv_indices(self.idx_buf, count, r.dist)
        g = gathered.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        c = counts.cpu().numpy()
        problem = None
        if g.shape[0] != int(c.sum()):
            problem = "gathered length differs from the sum of the counts"
        off = 0
        for q in range(world):
            part = g[off:off + int(c[q])]
            if part.size and not (part.min() >= q * n and part.max() < (q + 1) * n):
                problem = f"rank {q} indices out of its tile"
            if q == rank:
                mine = r.vis.fetch(0, write_back=False, occupancy=n)["visible_idx"].astype(np.int64) + rank * n
                if not np.array_equal(np.sort(part), mine):
                    problem = "own shard differs from the local visible list"
            off += int(c[q])
        return g, c, problem

    def check_padded(self, padded):
        """The per-frame exchange delivered the same lists as the exact one (static scene)."""
        r, exact, exact_counts = self.run, self.exact, self.exact_counts
        world, n = r.world, r.n
        if isinstance(padded, dict):  # a frame of gv_exchange_visible: library-owned rows, handed out complete by the acquire
            padded = r.vis.exchange_acquire(padded["frame"])
            if not padded["complete"]:
                return "c-abi exchange: an acquired frame is not complete"
            if not np.array_equal(np.asarray(padded["counts"], dtype=np.int64), exact_counts):
                return "c-abi exchange: counts differ from the exact all-gatherv"
            r.torch.cuda.synchronize()
            rows = r.device_words(padded["ptr"], world * padded["row_words"]).view(world, padded["row_words"]).cpu().numpy().view(np.uint32)
            off = 0
            for q in range(world):
                c = int(exact_counts[q])
                if int(rows[q, 0]) != c or not np.array_equal(rows[q, 1:1 + c].astype(np.int64), exact[off:off + c]):
                    return f"c-abi exchange: rank {q}'s row differs from the exact all-gatherv"
                off += c
            return None
        if not self.native:
            self.ex.drain()  # raises if any frame of the run overflowed its shard
        if r.args.payload == "mask":  # bits per mirror entry: the same SETS per rank (a mask has no order)
            r.torch.cuda.synchronize()
            d, counts = expand_mask_rows(padded, n, entry_tables=self.entry_tables)
            if not np.array_equal(counts, exact_counts):
                return "mask exchange: counts differ from the exact all-gatherv"
            off = 0
            for q in range(world):
                c = int(exact_counts[q])
                if not np.array_equal(d[off:off + c], np.sort(exact[off:off + c])):
                    return f"mask exchange: rank {q}'s set differs from the exact all-gatherv"
                off += c
            return None
        dense, counts = self.ex.compact(padded)
        d = dense.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        if not np.array_equal(counts.numpy(), exact_counts):
            return "padded exchange: counts differ from the exact all-gatherv"
        if not np.array_equal(d, exact):
            return "padded exchange: lists differ from the exact all-gatherv"
        return None

    # ---- set-up ----
    def share_entry_tables(self):
        """Once per mirror build: every rank learns every rank's entry -> pool-slot table (what a consumer of the bit shards
        needs to name the entities; the static scene never rebuilds its mirror)."""
        if self.entry_tables is not None:
            return
        r = self.run
        torch, dist = r.torch, r.dist
        mine = torch.from_numpy(r.vis.mirror_slots(0, r.n).astype(np.int32))
        tables = [torch.empty_like(mine) for _ in range(r.world)]
        if r.backend == "nccl":
            dev_tables = [t.to(r.device) for t in tables]
            dist.all_gather(dev_tables, mine.to(r.device))
            tables = [t.cpu() for t in dev_tables]
        else:
            dist.all_gather(tables, mine)
        self.entry_tables = [t.numpy().view(np.uint32) for t in tables]

    def make_exchange(self, payload):
        """The frame loop's exchange for `payload`: the torch object (and, for bit shards, every rank's entry -> slot table), or
        with the C-ABI path the rows a bit-shard exchange writes (index lists: the library owns the rows). Returns the torch
        object or None."""
        r = self.run
        r.args.payload = payload
        if payload == "mask":
            self.share_entry_tables()
        if self.native:
            if payload == "mask":
                self.native_rows = r.torch.zeros(r.world, 1 + mask_words(r.n), dtype=r.torch.int32, device=r.device)
                r.torch.cuda.synchronize()  # (the fill runs on torch's stream; the library's streams are non-blocking)
            return None
        capacity = mask_words(r.n) if payload == "mask" else shard_capacity(int(self.exact_counts.max()))
        # the direct patterns move only what each rank's list needs (every rank knows every count); the all-gather cannot
        per_rank = ([shard_capacity(int(c)) for c in self.exact_counts] if payload == "indices" and self.mode != "allgather" else None)
        return VisibleListExchange(r.dist, r.device, capacity, stream=self.producer, mode=self.mode, payload=payload, capacities=per_rank)

    def shard_words_per_rank(self, x, frame=None):
        """uint32 words rank r's shard puts on each link per frame (header included) under the exchange's pattern."""
        r = self.run
        if isinstance(frame, dict):
            return list(frame["travelled_words"])
        if x is None:  # native bit shards
            return [1 + mask_words(r.n)] * r.world
        if x.capacities is not None and x.mode != "allgather":
            return [1 + c for c in x.capacities]
        return [1 + x.capacity] * r.world

    def probe_modes(self):
        """--exchange not given, several ranks, the library's exchange up: five frames of each travel pattern (one more, untimed,
        in front) between fences; the slowest rank's time decides (one all-reduce: every rank derives the same choice) and the timed
        region uses the fastest. SURVEY.md §8e argues for the direct patterns on a fully connected node — only a node can tell."""
        r = self.run
        self.mode_probe_ms = {}
        for mode in ("allgather", "p2p", "broadcast"):
            r.vis.exchange_set_mode(EXCHANGE_MODES[mode])
            r.step()
            r.fence()
            t0 = time.perf_counter()
            for _ in range(5):
                r.step()
            r.vis.exchange_acquire(self.sent_frame)
            r.fence()
            self.mode_probe_ms[mode] = r.max_over_ranks((time.perf_counter() - t0) / 5 * 1e3)
            self.watchdog.pet(f"the probe of travel pattern {mode}")
        self.mode = min(self.mode_probe_ms, key=self.mode_probe_ms.get)
        r.vis.exchange_set_mode(EXCHANGE_MODES[self.mode])

    def setup(self):
        """Everything in front of the warm-up frames: the exact lists, the library's exchange brought up and trusted with one frame
        (else, loudly, torch.distributed), the travel pattern, the payload."""
        r, args = self.run, self.run.args
        rank, world = r.rank, r.world
        self.exact, self.exact_counts, problem = self.check_exchange()
        if not r.all_agree(problem is None):
            if rank == 0:
                r.emit({"error": "exchange check failed", "detail": problem})
            r.leave(1)
        self.gathered_total = int(self.exact_counts.sum())
        if os.environ.get("GV_BENCH_FALLBACK_REASON"):  # the child a hung C-ABI exchange left behind (watchdog)
            args.exchange_path = "torch"
        if args.exchange_path == "c-abi":
            self.watchdog.pet("start of the C-ABI exchange set-up")
            # the product's own exchange step: RCCL bound by the library, unique id handed round by the process group
            if r.backend != "nccl" and "GV_RCCL_LIBRARY" not in os.environ:
                # N ranks on one GPU (GV_BENCH_BACKEND=gloo): RCCL refuses that; the rows travel through the tests' shared-memory
                # transport — a functional run of the product's exchange logic, never a measurement
                os.environ["GV_RCCL_LIBRARY"] = os.path.join(self.root, "tests", "cpp", "build", "librccl_stub.so")
            if os.environ.get("GV_RCCL_LIBRARY"):
                self.transport_note = "GV_RCCL_LIBRARY=" + os.environ["GV_RCCL_LIBRARY"]
            init_problem = None
            try:
                ids = [GpuVisibility.exchange_unique_id() if rank == 0 else None]
            except Exception as e:  # noqa: BLE001 — reported below, on every rank
                ids, init_problem = [None], f"{type(e).__name__}: {e}"
            r.dist.broadcast_object_list(ids, src=0)
            if ids[0] is not None:
                try:
                    r.vis.exchange_init(ids[0], rank, world)
                    r.vis.exchange_set_mode(EXCHANGE_MODES[self.mode])
                    # (the library's own waits are bounded — GV_E_TIMEOUT — and would end this run without a line; here the watchdog
                    # above is the one that acts, by handing over to a child run: the library's bound is set behind it)
                    r.vis.exchange_set_timeout(int(os.environ.get("GV_BENCH_EXCHANGE_TIMEOUT_MS", 2000 * self.watchdog.seconds)))
                    # one frame through it, against the exact lists, before it is trusted with the timed frames
                    self.native = True
                    requested, args.payload = args.payload, "indices"
                    try:
                        init_problem = self.check_padded(r.step())
                    finally:
                        args.payload = requested
                except Exception as e:  # noqa: BLE001
                    init_problem = f"{type(e).__name__}: {e}"
            self.watchdog.pet("the trial frame")
            if not r.all_agree(init_problem is None and ids[0] is not None):
                # the library's own exchange did not come up on some rank: the line is still measured — through torch.distributed —
                # and says so loudly (exchange_path "torch", exchange_path_fallback = what went wrong)
                self.native = False
                self.watchdog.stop()
                self.path_fallback = init_problem or "the library's exchange failed on another rank"
                print(f"bench.py: rank {rank}: C-ABI exchange unavailable ({self.path_fallback}); timing the torch.distributed path", file=sys.stderr)
            elif args.exchange is None and world > 1 and args.payload != "mask":
                self.probe_modes()
        if args.payload == "auto":
            # the smaller encoding for this view: bits beat a word per visible entry above 1/32 visible (all ranks see all counts)
            dense_view = int(self.exact_counts.sum()) * 32 > r.n * world
            self.payload_note = f"auto: {self.exact_counts.sum() / (r.n * world):.1%} of the entities visible"
            if dense_view:
                trial_problem = "trial not run"
                try:  # one frame of the bit form against the exact lists before it is trusted with the timed frames
                    self.ex = self.make_exchange("mask")
                    trial_problem = self.check_padded(r.step())
                except Exception as e:  # noqa: BLE001 — anything at all: fall back to the lists
                    trial_problem = f"{type(e).__name__}: {e}"
                if r.all_agree(trial_problem is None):
                    self.payload_note += ", bit shards (checked on a trial frame)"
                else:
                    print(f"bench.py: bit-shard trial failed on some rank ({trial_problem}); using index lists", file=sys.stderr)
                    self.payload_note += ", index lists (the bit-shard trial failed)"
                    self.ex = self.make_exchange("indices")
            else:
                self.payload_note += ", index lists"
                self.ex = self.make_exchange("indices")
        else:
            self.ex = self.make_exchange(args.payload)

    # ---- figures for the line ----
    def isolated_ms(self):
        """One ISOLATED exchange (nothing overlapped: shard copy + collective + completion, host clock), median of five, slowest rank."""
        r = self.run
        lat = []
        for _ in range(5):
            r.compute()
            r.fence()
            t0 = time.perf_counter()
            if self.native:
                if r.args.payload == "mask":
                    r.vis.exchange_masks(0, mask_words(r.n), self.native_rows.data_ptr())
                else:
                    r.vis.exchange_acquire(self.native_frame()["frame"])
            else:
                shard = self.ex.next_shard()
                if r.args.payload == "mask":
                    r.vis.copy_mask_device(0, shard.data_ptr(), self.ex.capacity)
                else:
                    r.vis.copy_shard_device(0, shard.data_ptr(), self.ex.capacity, index_base=r.rank * r.n)
                self.ex.exchange()
                self.ex.drain()
            r.vis.wait()
            r.torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
        return r.max_over_ranks(float(np.median(lat)) * 1e3)

    def describe(self, timed_frame):
        """config.exchange: what travelled, how, in words."""
        r, args = self.run, self.run.args
        if timed_frame:
            return (f"per frame, through the library's C-ABI (gv_exchange_visible): shard [count, uint32 indices...] of every rank "
                    f"into library-owned rows (row stride {timed_frame['row_words']} words; room per rank {timed_frame['room']}, sized from the "
                    f"headers of earlier frames, which reach the host through pinned memory) by {self.mode} behind the cull "
                    f"stream, no host sync; RCCL bound by the library"
                    + (f" [{self.transport_note}: N ranks share a GPU, functional only]" if self.transport_note else "") +
                    f"; {self.gathered_total} indices gathered per rank; checked against the exact all-gatherv")
        return (f"per frame: " + ("shards [count, one bit per mirror entry] " if args.payload == "mask" else "padded shards [count, uint32 indices...] ") +
                (f"(capacity {self.ex.capacity} words) travel by {self.ex.describe()} behind the cull stream, no host sync ({r.backend})"
                 if self.ex is not None else f"({mask_words(r.n)} words) through the library's C-ABI (gv_exchange_masks) by {self.mode}") +
                f"; {self.gathered_total} indices gathered per rank; checked against the exact all-gatherv")
