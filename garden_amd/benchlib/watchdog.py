"""The two guards of bench.py's N > 1 runs: a collective of the library's exchange that never comes back must not cost the line."""
import os
import subprocess
import sys
import threading


class ExchangeWatchdog:
    """The library's exchange has only ever met real RCCL with one rank (no multi-GPU node was available to this build): if a
    collective of it never comes back on the first real node, the run must still print a line. Armed while the C-ABI exchange
    is the timed path and petted at every milestone; `seconds` without one and every rank (they all hang in the same
    collective) starts the script again as a CHILD with --exchange-path torch on a fresh rendezvous port, hands it the
    result descriptor and leaves with its exit code — the hung process cannot be repaired from inside, its stream is stuck
    behind the collective. The child's line says exchange_path "torch" and exchange_path_fallback = what happened.
    ONE child, ever: bark() disarms the timer first and takes a flag under a lock — an error path that barks while the timer armed
    240 s earlier is about to fire must not end with two children on one rendezvous port and one result descriptor."""

    def __init__(self, seconds, rank, result_fd, script):
        self.seconds, self.rank, self.result_fd, self.script = seconds, rank, result_fd, script
        self.where, self.timer = None, None
        self.lock, self.barked = threading.Lock(), False

    def pet(self, where):
        self.stop()
        if self.barked:
            return
        self.where = where
        self.timer = threading.Timer(self.seconds, self.bark)
        self.timer.daemon = True
        self.timer.start()

    def stop(self):
        timer, self.timer = self.timer, None
        if timer is not None:
            timer.cancel()

    def bark(self, error=None):
        with self.lock:
            if self.barked:  # (the other caller's child is the run's line; this thread must not start a second one)
                threading.Event().wait()
            self.barked = True
        self.stop()
        what = (f"the library's exchange failed after '{self.where}' (rank {self.rank}): {error}" if error else
                f"the library's exchange made no progress for {self.seconds:.0f} s after '{self.where}' (rank {self.rank})")
        reason = (what + ": timed through torch.distributed by a child run, which shared this GPU with the parent it replaced (its memory, "
                  "and a collective kernel that may still be spinning)")
        print("bench.py: " + reason, file=sys.stderr, flush=True)
        port = 1024 + (int(os.environ.get("MASTER_PORT", "29533")) + 17 - 1024) % 64000
        env = dict(os.environ, GV_BENCH_FALLBACK_REASON=reason, MASTER_PORT=str(port))
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)  # (the child's rank 0 serves its own rendezvous store on the new port)
        # a fresh CHILD (never a re-exec of this GPU-initialised process), which inherits the descriptor the one line goes to
        rc = subprocess.call([sys.executable, self.script] + sys.argv[1:], env=env, stdout=self.result_fd)
        os._exit(rc)


def guarded(seconds, on_timeout, work):
    """work() under a timer: on_timeout() (which does not return: it prints what there is and leaves) if it takes longer."""
    timer = threading.Timer(seconds, on_timeout)
    timer.daemon = True
    timer.start()
    try:
        work()
    finally:
        timer.cancel()
