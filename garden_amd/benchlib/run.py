"""What the pieces of one bench run share: the process group, the library context, the scene and the timing helpers."""
import json
import os
import sys
import time

import numpy as np


class Run:
    def __init__(self, args, result_fd):
        import torch
        self.args, self.torch, self.result_fd = args, torch, result_fd
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        # GV_BENCH_BACKEND=gloo lets N ranks share one GPU (exchange staged through the host): a functional check
        # of the multi-rank path on a 1-GPU box, never a measurement.
        self.backend = os.environ.get("GV_BENCH_BACKEND", "nccl")
        if self.backend != "nccl":
            self.local_rank = self.local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(self.local_rank)
        self.device = f"cuda:{self.local_rank}"
        self.comm_device = self.device if self.backend == "nccl" else "cpu"
        # GV_BENCH_EXCHANGE=1 runs the exchange step with a 1-rank group too (functional check of the RCCL path on a
        # 1-GPU box; the default N=1 line has no exchange)
        self.exchange = self.world > 1 or os.environ.get("GV_BENCH_EXCHANGE") == "1"
        if self.exchange:
            import torch.distributed as dist
            self.dist = dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(self.backend)
        self.vis = self.sc = self.view = self.view_array = self.depth = self.wl = self.lib_stream = None
        self.n = 0
        self.compute = self.step = None  # one frame without / with the exchange step (set by bench.py)

    # ---- the one line, and leaving together ----
    def emit(self, obj):
        os.write(self.result_fd, (json.dumps(obj) + "\n").encode())

    def leave(self, code):
        """Every rank leaves through here, together."""
        if self.exchange:
            try:
                self.dist.barrier()
                self.dist.destroy_process_group()
            except Exception:  # noqa: BLE001
                pass
        sys.exit(code)

    # ---- small collectives over the process group ----
    def all_agree(self, ok):
        """False on every rank when any rank reports a failure (so that nobody is left waiting in a barrier)."""
        if self.world == 1:
            return ok
        t = self.torch.tensor([0 if ok else 1], dtype=self.torch.int32, device=self.comm_device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return int(t.item()) == 0

    def max_over_ranks(self, x):
        if self.world == 1:
            return float(x)
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.comm_device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def every_rank(self, x):
        """[x of rank 0, x of rank 1, ...] on every rank."""
        if self.world == 1:
            return [float(x)]
        torch, dist = self.torch, self.dist
        t = torch.tensor([x], dtype=torch.float64, device=self.comm_device)
        out = torch.empty(self.world, dtype=torch.float64, device=self.comm_device)
        if self.backend == "nccl":
            dist.all_gather_into_tensor(out, t)
        else:
            pieces = [torch.empty_like(t) for _ in range(self.world)]
            dist.all_gather(pieces, t)
            out = torch.cat(pieces)
        return [float(v) for v in out.cpu()]

    # ---- timing ----
    def fence(self):
        # (drain first: the library's exchange runs on its own RCCL communicator and stream; a torch.distributed collective is never
        # enqueued while kernels of the other communicator are still in flight — two communicators' kernels resident at once have no
        # agreed order between the ranks)
        self.torch.cuda.synchronize()
        if self.exchange:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def timed_steps(self, run, steps, group=1):
        """Wall clock over `steps` frames between two fences (the contract's number) + one event per `group` frames on the
        library's stream (the durations behind the median). An event record is not free: the stream drains in front of it,
        ~6 us per record on MI355X (rocprofv3 shows the gap in front of the next kernel) — one per FRAME was 4 % of the cfg3
        frame, so the timed region marks every `group`-th frame boundary only."""
        bounds = list(range(0, steps, max(1, group))) + [steps]
        marks = {k: self.torch.cuda.Event(enable_timing=True) for k in bounds}
        self.fence()
        t0 = time.perf_counter()
        last = None
        marks[0].record(self.lib_stream)
        for k in range(steps):
            last = run()
            if k + 1 in marks:
                marks[k + 1].record(self.lib_stream)
        self.fence()
        elapsed = time.perf_counter() - t0
        per = np.array([marks[a].elapsed_time(marks[b]) / (b - a) for a, b in zip(bounds[:-1], bounds[1:])], dtype=np.float64)  # ms per frame
        return elapsed, per, last

    def device_words(self, ptr, count):
        """int32 view (no copy) of `count` words of library-owned device memory at `ptr`."""
        class _Span:
            pass
        span = _Span()
        span.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<i4", "data": (int(ptr), False), "version": 2}
        return self.torch.as_tensor(span, device=self.device)
