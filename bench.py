#!/usr/bin/env python3
"""bench.py — entity culls/sec of the visibility hot path on N MI355X GPUs (one process per GPU).

A "step" is one frame of the hot path over one batch of resident component pools:
  cfg2: 1M static entities, flat, frustum-only cull + compaction
  cfg3: 10M entities, Hi-Z pyramid rebuild (4096^2 depth) + frustum + Hi-Z occlusion cull + compaction  [default]
  cfg5: 12.5M entities per GPU (100M over 8 GPUs), flat, frustum-only + the exchange (run with --gpus 8)
  cfg4: 10M entities, 4-deep hierarchy: MFMA world-matrix sweep fused with the frustum cull (one pass) + compaction
        (--sweep mfma|valu: separate sweep and cull launches; fused-valu: the fused pass with the v_fma chain)
For N > 1 each rank owns one spatial tile, culls it against the same view and the ranks all-gatherv the COMPACTED
global visible-index lists over RCCL (cfg5 pattern, --payload indices: what BASELINE.json's north_star names); the
default workload is then cfg5 (12.5M per GPU, frustum-only + the exchange), for N = 1 it is cfg3. The N > 1 line also
carries: the same ranks' frames WITHOUT the exchange (`n1_same_workload`, what one GPU does with one tile) and
`scaling_efficiency` = value / (N * that); `exchange_ms` (one isolated exchange) and the bytes each rank's shard puts on
the links; the bit-shard encoding timed in the same run (`mask_variant`); parity of EVERY rank's tile against the oracle.
--scaling strong --entities-total T: one world of T entities cut into N tiles (T / N per GPU; N = 1: all of it).

`python bench.py --gpus N` without a torch.distributed environment (WORLD_SIZE unset) starts the N ranks itself, as
fresh child processes (python -m torch.distributed.run ... bench.py <same arguments>) BEFORE anything here touches the
GPU, and relays their one JSON line and exit code.

Inputs are resident in HBM before the timed region; outputs stay on the device (only a 4-byte count is
read back per frame). Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
# sources whose hash identifies the dominant kernel's code: profiles/traffic.json records it at PMC-collection time and
# roofline.traffic is only reported while it still matches (a stale counter figure is worse than none)
KERNEL_SOURCES = ["garden_amd/csrc/gv_cull.hip", "garden_amd/csrc/gv_device.hpp", "garden_amd/csrc/gv_device_math.hpp",
                  "garden_amd/csrc/gv_sweep.hip", "garden_amd/csrc/gv_kernels.hpp"]


def kernel_source_sha():
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]

WORKLOADS = {
    "cfg2": dict(entities=1_000_000, hier=False, hiz=False, sweep=False,
                 name="cfg2: 1M static entities, flat hierarchy, frustum-only AABB cull, fp32"),
    "cfg3": dict(entities=10_000_000, hier=False, hiz=True, sweep=False,
                 name="cfg3: 10M entities, frustum + Hi-Z occlusion vs synthetic 4096^2 depth pyramid (rebuilt per frame)"),
    "cfg5": dict(entities=12_500_000, hier=False, hiz=False, sweep=False,
                 name="cfg5: 100M entities over 8 spatial tiles (12.5M per GPU), frustum-only cull per tile + all-gather of the visible lists"),
    "cfg4": dict(entities=10_000_000, hier=True, hiz=False, sweep=True,
                 name="cfg4: 10M entities, 4-deep transform hierarchy recomputed each frame (MFMA 4x4 chain sweep) + cull"),
}
HIZ_SIZE = 4096


def make_tile_scene(wl, n_local, rank, world):
    """What rank `rank` owns of the world cube (side 100 * N_total^(1/3); camera at the world centre): the cube is cut into
    cell_grid(world) cells (8 x 8 x 8 for 8 GPUs), the cells are dealt to the ranks round-robin in Morton order
    (garden_amd/multi.py::cell_owners — the rule of gv_scene_extract_rank) and the rank's n_local roots are spread evenly over
    ITS cells, concatenated into one pool: every rank holds a share of every region, so every rank has its share of whatever
    the camera looks at (round 3 gave each rank one octant: half the ranks had nothing in view)."""
    from garden_amd import scene
    sc = scene.hierarchy_scene(n_local, seed=scene.SEED + rank) if wl["hier"] else scene.flat_scene(n_local, seed=scene.SEED + rank)
    if world > 1:
        from garden_amd.multi import cell_grid, cell_owners
        side = 100.0 * (n_local * world) ** (1.0 / 3.0)
        local_side = 100.0 * n_local ** (1.0 / 3.0)
        g = cell_grid(world)
        mine = np.nonzero(cell_owners(g, world) == rank)[0]  # linear cell ids x + y * gx + z * gx * gy
        k = mine.shape[0]
        roots = sc.transforms["parent"] == 0
        pos = sc.transforms["position"]
        # roots were drawn uniform in [-local_side/2, local_side/2)^3: x picks the cell (k equal slabs of the local cube) and the
        # place inside it, y and z the place inside the cell
        u = (pos[roots, :3].astype(np.float64) / local_side + 0.5).clip(0.0, np.nextafter(1.0, 0.0))
        j = np.minimum((u[:, 0] * k).astype(np.int64), k - 1)
        u[:, 0] = u[:, 0] * k - j
        cell = mine[j]
        cxyz = np.stack([cell % g[0], (cell // g[0]) % g[1], cell // (g[0] * g[1])], axis=1).astype(np.float64)
        ext = side / np.array(g, dtype=np.float64)
        pos[roots, :3] = (-0.5 * side + (cxyz + u) * ext).astype(np.float32)
    return sc


def algorithmic_bytes(wl, n, frustum_survivors, visible, depth, fused=False, examined=1.0):
    """Minimal SoA stream bytes per launch for the cull kernel, and
    for the whole step (for information). `examined`: fraction of the 256-entry workgroups whose streams are read
    (1 without block bounds; with them the rest only write their outputs and read a 32-byte box)."""
    # TRS 40 + AABB 24 + flags 1 read; ballot word 1/8 written (the isVisible bytes are expanded from those words by the emit
    # kernel: 1 B per entity there, not here)
    cull = n * examined * 65.0 + n * 0.125
    if examined < 1.0:
        cull += (n / 256.0) * 32.0
    if wl["hier"]:
        cull += n * examined * 4.0  # parent index
    if wl["hiz"]:
        cull += frustum_survivors * 32.0  # 4 texels x (min,max) fp32 per frustum-surviving entity
    emit = visible * (40.0 + 4.0 + 4.0 + 48.0 + 4.0) + n * 0.125 + n * 1.0
    # level 1 is not stored: depth read + levels 2..12 written
    hiz = (HIZ_SIZE * HIZ_SIZE * 4 + sum(max(HIZ_SIZE >> k, 1) ** 2 * 8 for k in range(2, 13))) if wl["hiz"] else 0.0
    sweep = n * (40.0 + 4.0 + 48.0) if wl["sweep"] else 0.0
    if fused and wl["sweep"]:  # one pass: the TRS streams are read once, the world matrices (48 B) written beside the cull outputs
        cull += n * 48.0
        sweep = 0.0
    return dict(cull=cull, emit=emit, hiz=float(hiz), sweep=sweep)


def effective_cores():
    """CPU time this process can actually get: the hardware threads it may run on, capped by the container's cgroup quota
    (the GPU boxes show 256 hardware threads and a cpu.max of 16 CPUs: 128 busy threads then share 16 CPUs' worth of time)."""
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    quota = None
    try:  # cgroup v2
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:  # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    eff = allowed if quota is None else max(1, min(allowed, int(quota + 0.5)))
    return eff, allowed, quota


def cpu_baseline(wl, sc, view, depth, seconds=10.0):
    """The CPU path on this box's host cores, same frame as the GPU step, over the WHOLE pool: AVX2+FMA cull
    (oracle/gv_oracle_avx2.c: 8 entities per iteration over an SoA copy of the pools, bit-identical to the scalar
    restatement of mesh.cpp:111-184 + transform.hpp:197-214), the pyramid by the scalar hiz.frag restatement and (cfg4)
    the scalar world-matrix sweep, each threaded with the ThreadPool::addItems range split. The SoA arrays are first touched
    by the threads that cull them. Thread counts are tried around what the container may actually use (cgroup quota), the
    fastest one is timed. The stages are timed separately and summed: `value` = entities / (pyramid + sweep + cull) per
    frame. A reported baseline, not the optimisation target."""
    from oracle import oracle_py
    eff, allowed, quota = effective_cores()
    n = sc.count
    meshes, transforms, e2t = sc.meshes.copy(), sc.transforms, sc.entity_to_transform
    hz = oracle_py.Hiz(depth, threads=eff) if wl["hiz"] else None
    world = np.empty((n, 12), dtype=np.float32) if wl["sweep"] else None

    def timed(run, seconds, min_frames=2):
        run()  # untimed: first touch of the output arrays, worker threads started
        frames, t0 = 0, time.perf_counter()
        while True:
            run()
            frames += 1
            dt = time.perf_counter() - t0
            if dt >= seconds and frames >= min_frames:
                return dt / frames, frames

    candidates = sorted({max(1, eff // 2), eff, min(allowed, eff * 2), min(allowed, eff * 4)}, reverse=True)

    def best_threads(run_with):
        """The reference sizes its pool to the hardware threads (thread-pool.cpp:56-70); under a CPU quota that is far more
        threads than CPUs, so a few counts around the quota are tried briefly and the fastest one is what gets timed."""
        best, best_t = eff, None
        for th in candidates:
            run_with(th)
            t0 = time.perf_counter()
            run_with(th)
            run_with(th)
            dt = (time.perf_counter() - t0) / 2
            if best_t is None or dt < best_t:
                best, best_t = th, dt
        return best

    share = seconds / (1 + (1 if wl["hiz"] else 0) + (1 if wl["sweep"] else 0))
    # the SoA copy is split over, and first touched by, as many workers as will cull it: one build per candidate count
    soas = {}

    def soa_for(th):
        if th not in soas:
            soas[th] = oracle_py.Avx2Scene(meshes, transforms, e2t, threads=th)
        return soas[th]

    cull_threads = best_threads(lambda th: soa_for(th).prepare_meshes(view, hiz=hz, threads=th))
    for th in list(soas):
        if th != cull_threads:
            soas.pop(th).close()
    soa = soa_for(cull_threads)
    cull_s, cull_frames = timed(lambda: soa.prepare_meshes(view, hiz=hz, threads=cull_threads), share)
    frustum_only_s = cull_s
    if wl["hiz"]:  # the same loop without the occlusion queries: separates the scalar Hi-Z queries from the 8-wide frustum test
        frustum_only_s, _ = timed(lambda: soa.prepare_meshes(dict(view, use_hiz=0), threads=cull_threads), 1.0, 1)
    pyramid_s = pyramid_1t_s = sweep_s = 0.0
    pyramid_threads = sweep_threads = None
    if wl["hiz"]:
        pyramid_1t_s, _ = timed(lambda: hz.rebuild(1), 0.5, 1)
        pyramid_threads = best_threads(lambda th: hz.rebuild(th))
        pyramid_s, _ = timed(lambda: hz.rebuild(pyramid_threads), share)
        if pyramid_1t_s < pyramid_s:
            pyramid_s, pyramid_threads = pyramid_1t_s, 1
    if wl["sweep"]:
        sweep_threads = best_threads(lambda th: oracle_py.world_matrices(transforms, e2t, 0, n, threads=th, out=world))
        sweep_s, _ = timed(lambda: oracle_py.world_matrices(transforms, e2t, 0, n, threads=sweep_threads, out=world), share)
    # also one thread, and the scalar loop over the reference's AoS layouts (short samples)
    cull_1t_s, _ = timed(lambda: soa.prepare_meshes(view, hiz=hz, threads=1), 2.0, 1)
    scalar_s, _ = timed(lambda: oracle_py.prepare_meshes(meshes, transforms, e2t, view, hiz=hz, threads=cull_threads), 2.0, 1)
    soa.close()
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    frame_s = cull_s + pyramid_s + sweep_s
    return dict(value=n / frame_s, unit="entity culls/s", cores=eff, kind="port",
                sample=f"all {n} entities of the same scene/view; per frame: "
                       f"{str(depth.shape[1]) + 'x' + str(depth.shape[0]) + ' pyramid build (scalar hiz.frag restatement, rows split over the threads) + ' if wl['hiz'] else ''}"
                       f"{'scalar world-matrix sweep (slot ranges split over the threads) + ' if wl['sweep'] else ''}"
                       f"AVX2+FMA 8-wide SoA cull (bit-identical to the scalar oracle; arrays first touched by the culling threads), "
                       f"ranges split like ThreadPool::addItems; this process may use {eff} CPUs "
                       f"({allowed} hardware threads visible"
                       f"{', cgroup CPU quota %.1f' % quota if quota is not None else ', no cgroup quota'}): `cores` is that number, "
                       f"and the thread count per stage is the fastest of {candidates} (cull {cull_threads}"
                       f"{', pyramid ' + str(pyramid_threads) if pyramid_threads else ''}"
                       f"{', sweep ' + str(sweep_threads) if sweep_threads else ''}); stages "
                       f"timed separately ({cull_frames} cull frames) and summed",
                threads_used=dict(cull=cull_threads, pyramid=pyramid_threads, sweep=sweep_threads),
                cpu_model=model, nproc=os.cpu_count() or 1, hardware_threads_allowed=allowed, cgroup_cpu_quota=quota,
                frame_ms=frame_s * 1e3, cull_ms=cull_s * 1e3, pyramid_ms=pyramid_s * 1e3, sweep_ms=sweep_s * 1e3,
                cull_culls_per_s=n / cull_s,
                frustum_only_culls_per_s=n / frustum_only_s,
                pyramid_1_thread_ms=pyramid_1t_s * 1e3,
                avx2_soa_cull_1_thread_culls_per_s=n / cull_1t_s,
                # the cull alone on all the CPUs the process may use, against that many times one thread
                parallel_efficiency=(n / cull_s) / (eff * (n / cull_1t_s)),
                scalar_aos_cull_all_threads_culls_per_s=n / scalar_s)


def spawn_ranks(args, argv):
    """`bench.py --gpus N` outside a torch.distributed launch: start the N ranks as a fresh child job. Nothing in this
    process has touched the GPU yet (no torch import, no HIP call), and the child is a child — never an exec."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 200 timed frames by default: the timed region carries a handful of event records (five frame-group marks, four or five brackets
    # around the dominant kernel) at ~6 us of stream time each — 1.7 us per frame over 50 frames, 0.4 us over 200
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: cfg3 on one GPU, cfg5 (12.5M entities per GPU, frustum-only + exchange) on several")
    ap.add_argument("--entities", type=int, default=0, help="per-GPU entity count override")
    ap.add_argument("--sweep", default="fused", choices=["mfma", "valu", "fused", "fused-valu"],
                    help="cfg4 world-matrix sweep form; fused = MFMA sweep and cull in one pass (GV_SWEEP_WITH_CULL)")
    ap.add_argument("--profile-all", action="store_true", help="hipEvents around every kernel (slower step)")
    ap.add_argument("--block-bounds", action="store_true",
                    help="GV_CONFIG_BLOCK_BOUNDS: conservative workgroup-level frustum rejection (same results); the "
                         "roofline numerator then counts the streams of examined workgroups only")
    ap.add_argument("--hiz-rg16f", action="store_true",
                    help="GV_CONFIG_HIZ_RG16F: the pyramid in the reference's RG16F image format, rounded outward (a variant: "
                         "the headline keeps the fp32 pyramid; parity is then checked against the oracle's RG16F pyramid)")
    ap.add_argument("--exchange", default=os.environ.get("GV_BENCH_EXCHANGE_MODE", "allgather"),
                    choices=["allgather", "p2p", "broadcast"],
                    help="N > 1: how the padded shards travel — one equal-size all-gather (default), grouped point-to-point "
                         "send/recv to every peer, or one broadcast per root (A/B for the fully connected xGMI node)")
    ap.add_argument("--exchange-path", default=os.environ.get("GV_BENCH_EXCHANGE_PATH", "c-abi"), choices=["c-abi", "torch"],
                    help="N > 1: who runs the exchange — the library's own C-ABI step (default: gv_exchange_init / gv_exchange_visible, "
                         "RCCL bound by the library, rows owned and sized by it: what a C++ engine calls), or torch.distributed over "
                         "buffers this script owns (garden_amd/multi.py; timed beside the headline as config.torch_variant)")
    ap.add_argument("--payload", default=os.environ.get("GV_BENCH_EXCHANGE_PAYLOAD", "indices"), choices=["auto", "indices", "mask"],
                    help="N > 1: what a shard carries — the compacted uint32 index list (default: the all-gatherv of the visible list "
                         "BASELINE.json names), or one bit per mirror entry behind the count (1/32 word per entry whatever the view: ~7x "
                         "fewer bytes at the bench's 21 %% visibility; timed in the same run as `mask_variant` unless --no-mask-variant). "
                         "auto: the smaller of the two for the view at hand (bits above 1/32 visible), after one trial frame of the bit "
                         "form has been checked against the exact all-gatherv on every rank. The gathered sets of the timed frames are "
                         "checked against the exact all-gatherv either way")
    ap.add_argument("--no-mask-variant", action="store_true", help="N > 1: skip the bit-shard frames timed beside the index lists")
    ap.add_argument("--no-mode-variants", action="store_true", help="N > 1, c-abi path: skip the frames timed with the other travel patterns")
    ap.add_argument("--no-torch-variant", action="store_true", help="N > 1, c-abi path: skip the frames timed through torch.distributed")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): the workload's entity count PER GPU; strong: --entities-total cut into N spatial tiles")
    ap.add_argument("--entities-total", type=int, default=100_000_000, help="--scaling strong: entities of the whole world")
    ap.add_argument("--depth", default="walls", choices=["walls", "noise"],
                    help="cfg3 depth image: SURVEY.md §8d's 256 walls (default), or per-8x8-block occluders among the entities "
                         "(scene.noise_depth: the coarse-level exits of the occlusion query decide almost nothing; timed beside the "
                         "headline as config.hard_depth_variant when the headline runs on the walls)")
    ap.add_argument("--no-hard-depth-variant", action="store_true",
                    help="skip config.hard_depth_variant. Skipped by itself under rocprofv3: its launches are the dominant kernel under the "
                         "same name, on another depth image — they would mix into the profiler's per-kernel average and counters")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    # ---- launch shape, checked before torch is imported or the GPU is touched ----
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    if env_world is not None and int(env_world) != args.gpus:
        print(json.dumps({"error": f"--gpus {args.gpus} but WORLD_SIZE={env_world}: launch with "
                                   f"torch.distributed.run --nproc-per-node {args.gpus}, or run plain "
                                   f"`python bench.py --gpus {args.gpus}` (it starts the ranks itself)"}), flush=True)
        sys.exit(2)
    if os.environ.get("GV_BENCH_FALLBACK_REASON"):  # the child a hung C-ABI exchange left behind (exchange watchdog below)
        args.exchange_path = "torch"
    if args.workload is None:  # (strong scaling: the same workload at every N, N = 1 included)
        args.workload = "cfg5" if args.gpus > 1 or args.scaling == "strong" else "cfg3"

    # The contract is ONE JSON line on stdout. Libraries underneath (RCCL prints a version banner when a communicator
    # is created) write to file descriptor 1 too, so everything but the result lines is sent to stderr.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(result_fd, (json.dumps(obj) + "\n").encode())

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # GV_BENCH_BACKEND=gloo lets N ranks share one GPU (exchange staged through the host): a functional check
    # of the multi-rank path on a 1-GPU box, never a measurement.
    backend = os.environ.get("GV_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    # GV_BENCH_EXCHANGE=1 runs the exchange step with a 1-rank group too (functional check of the RCCL path on a
    # 1-GPU box; the default N=1 line has no exchange)
    exchange = world > 1 or os.environ.get("GV_BENCH_EXCHANGE") == "1"
    if exchange:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    class ExchangeWatchdog:
        """The library's exchange has only ever met real RCCL with one rank (no multi-GPU node was available to this build): if a
        collective of it never comes back on the first real node, the run must still print a line. Armed while the C-ABI exchange
        is the timed path and petted at every milestone; `seconds` without one and every rank (they all hang in the same
        collective) starts this script again as a CHILD with --exchange-path torch on a fresh rendezvous port, hands it the
        result descriptor and leaves with its exit code — the hung process cannot be repaired from inside, its stream is stuck
        behind the collective. The child's line says exchange_path "torch" and exchange_path_fallback = what happened."""

        def __init__(self, seconds):
            self.seconds, self.where, self.timer = seconds, None, None

        def pet(self, where):
            import threading
            self.stop()
            self.where = where
            self.timer = threading.Timer(self.seconds, self.bark)
            self.timer.daemon = True
            self.timer.start()

        def stop(self):
            if self.timer is not None:
                self.timer.cancel()
                self.timer = None

        def bark(self, error=None):
            import subprocess
            what = (f"the library's exchange failed after '{self.where}' (rank {rank}): {error}" if error else
                    f"the library's exchange made no progress for {self.seconds:.0f} s after '{self.where}' (rank {rank})")
            reason = (what + ": timed through torch.distributed by a child run, which shared this GPU with the parent it replaced (its memory, "
                      "and a collective kernel that may still be spinning)")
            print("bench.py: " + reason, file=sys.stderr, flush=True)
            port = 1024 + (int(os.environ.get("MASTER_PORT", "29533")) + 17 - 1024) % 64000
            env = dict(os.environ, GV_BENCH_FALLBACK_REASON=reason, MASTER_PORT=str(port))
            env.pop("TORCHELASTIC_USE_AGENT_STORE", None)  # (the child's rank 0 serves its own rendezvous store on the new port)
            rc = subprocess.call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=result_fd)
            os._exit(rc)

    exchange_watchdog = ExchangeWatchdog(float(os.environ.get("GV_BENCH_EXCHANGE_WATCHDOG_S", "240")))

    def leave(code):
        """Every rank leaves through here, together."""
        if exchange:
            try:
                dist.barrier()
                dist.destroy_process_group()
            except Exception:
                pass
        sys.exit(code)

    def all_agree(ok):
        """False on every rank when any rank reports a failure (so that nobody is left waiting in a barrier)."""
        if world == 1:
            return ok
        t = torch.tensor([0 if ok else 1], dtype=torch.int32, device=f"cuda:{local_rank}" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return int(t.item()) == 0

    from garden_amd import scene
    from garden_amd.lib import GpuVisibility, GvError, GV_SWEEP_MFMA, GV_SWEEP_VALU, GV_SWEEP_WITH_CULL, GV_SWEEP_WITH_CULL_VALU
    from garden_amd.multi import VisibleListExchange, allgatherv_indices, shard_capacity, mask_words, expand_mask_rows

    wl = WORKLOADS[args.workload]
    n = args.entities or wl["entities"]
    if args.scaling == "strong":  # one world of --entities-total cut into `world` spatial tiles (the same world at every N)
        n = args.entities_total // world
    sc = make_tile_scene(wl, n, rank, world)
    view = scene.main_camera_view(use_hiz=1 if wl["hiz"] else 0)
    depth = ((scene.noise_depth(HIZ_SIZE, HIZ_SIZE) if args.depth == "noise" else scene.synthetic_depth(HIZ_SIZE, HIZ_SIZE))
             if wl["hiz"] else None)

    # hipEvents bracket only the dominant kernel inside the timed region (each event record costs ~2 us of
    # stream time); --profile-all brackets every kernel for the per-kernel breakdown in config.kernel_ms
    vis = GpuVisibility(device=local_rank, profile_events=args.profile_all, profile_cull_only=not args.profile_all,
                        block_bounds=args.block_bounds, hiz_rg16f=args.hiz_rg16f,
                        linear_scan=not args.block_bounds)  # the headline is the flat loop SURVEY.md §8d prices (mesh.cpp:137-175)
    t_up = time.perf_counter()
    vis.bind_transforms(sc.transforms, sc.entity_to_transform)
    vis.bind_pool(0, sc.meshes)
    vis.hierarchy_rebuild()
    vis.wait()
    upload_s = time.perf_counter() - t_up
    if wl["hiz"]:
        vis.hiz_build(depth)
    lib_stream = torch.cuda.ExternalStream(vis.stream(), device=torch.device("cuda", local_rank))

    idx_buf = torch.empty(n, dtype=torch.int32, device=f"cuda:{local_rank}") if exchange else None
    ex = [None]  # VisibleListExchange, created once the shard capacity is known (first, exact exchange)
    entry_tables = [None]  # --payload mask: every rank's mirror entry -> pool slot table

    view_array = vis.views_array([view])  # the GvView structs of the frame, built once (the timed loop is the library's, not ctypes')

    def compute():
        if wl["hiz"]:
            vis.hiz_rebuild()
        if wl["sweep"]:
            vis.sweep({"mfma": GV_SWEEP_MFMA, "valu": GV_SWEEP_VALU, "fused": GV_SWEEP_WITH_CULL, "fused-valu": GV_SWEEP_WITH_CULL_VALU}[args.sweep])
        vis.cull(0, view_array)

    native = [False]       # the exchange runs through the library's own C-ABI step (gv_exchange_visible / gv_exchange_masks)
    native_rows = [None]   # caller-owned rows [world, 1 + words] of the native bit-shard exchange
    EXCHANGE_MODES = {"allgather": 0, "p2p": 1, "broadcast": 2}

    def device_words(ptr, count):
        """int32 view (no copy) of `count` words of library-owned device memory at `ptr`."""
        class _Span:
            pass
        span = _Span()
        span.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<i4", "data": (int(ptr), False), "version": 2}
        return torch.as_tensor(span, device=f"cuda:{local_rank}")

    sent_frame = [None]
    exchange_frames = [0, 0]  # frames acquired through the library's exchange / of those, frames that needed the second (tail) exchange

    def native_frame():
        """gv_exchange_visible for this frame, then the PREVIOUS frame acquired, the way a consumer one frame behind does: the send
        has settled that frame (a short row completed by a second exchange inside the call), so the acquire is a stream wait —
        every frame of the timed region is handed out complete."""
        try:
            f = vis.exchange_visible(0, index_base=rank * n)
            if sent_frame[0] is not None:
                acquired = vis.exchange_acquire(sent_frame[0])
                exchange_frames[0] += 1
                exchange_frames[1] += 1 if acquired["cut_ranks"] else 0
        except GvError as e:
            # a status code from the library's exchange in the middle of the run (GV_E_TIMEOUT, GV_E_RCCL: it has never met real RCCL
            # with several ranks): like a collective that never returns, the line is handed to a child run through torch.distributed
            if world > 1 and not os.environ.get("GV_BENCH_FALLBACK_REASON"):
                exchange_watchdog.bark(error=str(e))
            raise
        sent_frame[0] = f["frame"]
        return f

    def step():
        """One frame. With an exchange: the rank's list goes out as a shard [count, indices...] and all ranks gather the shards
        (one equal-size all-gather, or the --exchange alternative) enqueued behind the library's stream — no host
        synchronisation, so the next frame is culled while this one's list is still on the links. --exchange-path c-abi (default):
        the library's own step — it owns the rows and sizes every rank's from the headers of earlier frames; torch: this script's
        buffers and torch.distributed (garden_amd/multi.py)."""
        compute()
        if native[0]:
            if args.payload == "mask":
                vis.exchange_masks(0, mask_words(n), native_rows[0].data_ptr())
                return native_rows[0]
            return native_frame()
        if ex[0] is not None:
            shard = ex[0].next_shard()
            if args.payload == "mask":
                vis.copy_mask_device(0, shard.data_ptr(), ex[0].capacity)
            else:
                vis.copy_shard_device(0, shard.data_ptr(), ex[0].capacity, index_base=rank * n)
            return ex[0].exchange()
        return None

    def check_exchange():
        """Exact-size all-gatherv of one frame (host-synchronising form): all ranks hold the same concatenated list;
        every index lies in its owner's tile range; own shard == local visible list. Sizes the padded shards."""
        compute()
        vis.copy_idx_device(0, idx_buf.data_ptr(), n, index_base=rank * n)
        count = vis.result_count(0)  # 4-byte readback on the library's stream: also fences the copy above
        gathered, counts = allgatherv_indices(idx_buf, count, dist)
        g = gathered.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        c = counts.cpu().numpy()
        problem = None
        if g.shape[0] != int(c.sum()):
            problem = "gathered length differs from the sum of the counts"
        off = 0
        for r in range(world):
            part = g[off:off + int(c[r])]
            if part.size and not (part.min() >= r * n and part.max() < (r + 1) * n):
                problem = f"rank {r} indices out of its tile"
            if r == rank:
                mine = vis.fetch(0, write_back=False, occupancy=n)["visible_idx"].astype(np.int64) + rank * n
                if not np.array_equal(np.sort(part), mine):
                    problem = "own shard differs from the local visible list"
            off += int(c[r])
        return g, c, problem

    def check_padded(padded, exact, exact_counts):
        """The per-frame exchange delivered the same lists as the exact one (static scene)."""
        if isinstance(padded, dict):  # a frame of gv_exchange_visible: library-owned rows, handed out complete by the acquire
            padded = vis.exchange_acquire(padded["frame"])
            counts = padded["counts"]
            if not padded["complete"]:
                return "c-abi exchange: an acquired frame is not complete"
            if not np.array_equal(np.asarray(counts, dtype=np.int64), exact_counts):
                return "c-abi exchange: counts differ from the exact all-gatherv"
            torch.cuda.synchronize()
            rows = device_words(padded["ptr"], world * padded["row_words"]).view(world, padded["row_words"]).cpu().numpy().view(np.uint32)
            off = 0
            for r in range(world):
                c = int(exact_counts[r])
                if int(rows[r, 0]) != c or not np.array_equal(rows[r, 1:1 + c].astype(np.int64), exact[off:off + c]):
                    return f"c-abi exchange: rank {r}'s row differs from the exact all-gatherv"
                off += c
            return None
        if not native[0]:
            ex[0].drain()  # raises if any frame of the run overflowed its shard
        if args.payload == "mask":  # bits per mirror entry: the same SETS per rank (a mask has no order)
            torch.cuda.synchronize()
            d, counts = expand_mask_rows(padded, n, entry_tables=entry_tables[0])
            if not np.array_equal(counts, exact_counts):
                return "mask exchange: counts differ from the exact all-gatherv"
            off = 0
            for r in range(world):
                c = int(exact_counts[r])
                if not np.array_equal(d[off:off + c], np.sort(exact[off:off + c])):
                    return f"mask exchange: rank {r}'s set differs from the exact all-gatherv"
                off += c
            return None
        dense, counts = ex[0].compact(padded)
        d = dense.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        if not np.array_equal(counts.numpy(), exact_counts):
            return "padded exchange: counts differ from the exact all-gatherv"
        if not np.array_equal(d, exact):
            return "padded exchange: lists differ from the exact all-gatherv"
        return None

    def fence():
        # (drain first: the library's exchange runs on its own RCCL communicator and stream; a torch.distributed collective is never
        # enqueued while kernels of the other communicator are still in flight — two communicators' kernels resident at once have no
        # agreed order between the ranks)
        torch.cuda.synchronize()
        if exchange:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_steps(run, steps, group=1):
        """Wall clock over `steps` frames between two fences (the contract's number) + one event per `group` frames on the
        library's stream (the durations behind the median). An event record is not free: the stream drains in front of it,
        ~6 us per record on MI355X (rocprofv3 shows the gap in front of the next kernel) — one per FRAME was 4 % of the cfg3
        frame, so the timed region marks every `group`-th frame boundary only."""
        bounds = list(range(0, steps, max(1, group))) + [steps]
        marks = {k: torch.cuda.Event(enable_timing=True) for k in bounds}
        fence()
        t0 = time.perf_counter()
        last = None
        marks[0].record(lib_stream)
        for k in range(steps):
            last = run()
            if k + 1 in marks:
                marks[k + 1].record(lib_stream)
        fence()
        elapsed = time.perf_counter() - t0
        per = np.array([marks[a].elapsed_time(marks[b]) / (b - a) for a, b in zip(bounds[:-1], bounds[1:])], dtype=np.float64)  # ms per frame
        return elapsed, per, last

    gathered_total = None
    tables_ready = [False]
    if exchange:
        exact, exact_counts, problem = check_exchange()
        if not all_agree(problem is None):
            if rank == 0:
                emit({"error": "exchange check failed", "detail": problem})
            leave(1)
        gathered_total = int(exact_counts.sum())
        producer = lib_stream if backend == "nccl" else None

        def share_entry_tables():
            """Once per mirror build: every rank learns every rank's entry -> pool-slot table (what a consumer of the bit shards
            needs to name the entities; the static scene never rebuilds its mirror)."""
            if tables_ready[0]:
                return
            mine = torch.from_numpy(vis.mirror_slots(0, n).astype(np.int32))
            tables = [torch.empty_like(mine) for _ in range(world)]
            if backend == "nccl":
                dev_tables = [t.to(f"cuda:{local_rank}") for t in tables]
                dist.all_gather(dev_tables, mine.to(f"cuda:{local_rank}"))
                tables = [t.cpu() for t in dev_tables]
            else:
                dist.all_gather(tables, mine)
            entry_tables[0] = [t.numpy().view(np.uint32) for t in tables]
            tables_ready[0] = True

        def make_exchange(payload):
            """The frame loop's exchange for `payload`: the torch object (and, for bit shards, every rank's entry -> slot table), or
            with the C-ABI path the rows a bit-shard exchange writes (index lists: the library owns the rows). Returns the torch
            object or None."""
            args.payload = payload
            if payload == "mask":
                share_entry_tables()
            if native[0]:
                if payload == "mask":
                    native_rows[0] = torch.zeros(world, 1 + mask_words(n), dtype=torch.int32, device=f"cuda:{local_rank}")
                    torch.cuda.synchronize()  # (the fill runs on torch's stream; the library's streams are non-blocking)
                return None
            capacity = mask_words(n) if payload == "mask" else shard_capacity(int(exact_counts.max()))
            # the direct patterns move only what each rank's list needs (every rank knows every count); the all-gather cannot
            per_rank = ([shard_capacity(int(c)) for c in exact_counts] if payload == "indices" and args.exchange != "allgather" else None)
            return VisibleListExchange(dist, f"cuda:{local_rank}", capacity, stream=producer, mode=args.exchange, payload=payload,
                                       capacities=per_rank)

        def shard_words_per_rank(x, frame=None):
            """uint32 words rank r's shard puts on each link per frame (header included) under the exchange's pattern."""
            if isinstance(frame, dict):
                return list(frame["travelled_words"])
            if x is None:  # native bit shards
                return [1 + mask_words(n)] * world
            if x.capacities is not None and x.mode != "allgather":
                return [1 + c for c in x.capacities]
            return [1 + x.capacity] * world

        transport_note = None
        path_fallback = [os.environ.get("GV_BENCH_FALLBACK_REASON")]
        if args.exchange_path == "c-abi":
            exchange_watchdog.pet("start of the C-ABI exchange set-up")
            # the product's own exchange step: RCCL bound by the library, unique id handed round by the process group
            if backend != "nccl" and "GV_RCCL_LIBRARY" not in os.environ:
                # N ranks on one GPU (GV_BENCH_BACKEND=gloo): RCCL refuses that; the rows travel through the tests' shared-memory
                # transport — a functional run of the product's exchange logic, never a measurement
                os.environ["GV_RCCL_LIBRARY"] = os.path.join(ROOT, "tests", "cpp", "build", "librccl_stub.so")
            if os.environ.get("GV_RCCL_LIBRARY"):
                transport_note = "GV_RCCL_LIBRARY=" + os.environ["GV_RCCL_LIBRARY"]
            init_problem = None
            try:
                ids = [GpuVisibility.exchange_unique_id() if rank == 0 else None]
            except Exception as e:  # noqa: BLE001 — reported below, on every rank
                ids, init_problem = [None], f"{type(e).__name__}: {e}"
            dist.broadcast_object_list(ids, src=0)
            if ids[0] is not None:
                try:
                    vis.exchange_init(ids[0], rank, world)
                    vis.exchange_set_mode(EXCHANGE_MODES[args.exchange])
                    # (the library's own waits are bounded — GV_E_TIMEOUT — and would end this run without a line; here the watchdog
                    # above is the one that acts, by handing over to a child run: the library's bound is set behind it)
                    vis.exchange_set_timeout(int(os.environ.get("GV_BENCH_EXCHANGE_TIMEOUT_MS", 2000 * exchange_watchdog.seconds)))
                    # one frame through it, against the exact lists, before it is trusted with the timed frames
                    native[0] = True
                    requested, args.payload = args.payload, "indices"
                    try:
                        init_problem = check_padded(step(), exact, exact_counts)
                    finally:
                        args.payload = requested
                except Exception as e:  # noqa: BLE001
                    init_problem = f"{type(e).__name__}: {e}"
            exchange_watchdog.pet("the trial frame")
            if not all_agree(init_problem is None and ids[0] is not None):
                # the library's own exchange did not come up on some rank: the line is still measured — through torch.distributed —
                # and says so loudly (exchange_path "torch", exchange_path_fallback = what went wrong)
                native[0] = False
                exchange_watchdog.stop()
                path_fallback[0] = init_problem or "the library's exchange failed on another rank"
                print(f"bench.py: rank {rank}: C-ABI exchange unavailable ({path_fallback[0]}); timing the torch.distributed path", file=sys.stderr)

        payload_note = None
        if args.payload == "auto":
            # the smaller encoding for this view: bits beat a word per visible entry above 1/32 visible (all ranks see all counts)
            dense_view = int(exact_counts.sum()) * 32 > n * world
            payload_note = f"auto: {exact_counts.sum() / (n * world):.1%} of the entities visible"
            if dense_view:
                trial_problem = "trial not run"
                try:  # one frame of the bit form against the exact lists before it is trusted with the timed frames
                    ex[0] = make_exchange("mask")
                    trial_problem = check_padded(step(), exact, exact_counts)
                except Exception as e:  # noqa: BLE001 — anything at all: fall back to the lists
                    trial_problem = f"{type(e).__name__}: {e}"
                if all_agree(trial_problem is None):
                    payload_note += ", bit shards (checked on a trial frame)"
                else:
                    print(f"bench.py: bit-shard trial failed on some rank ({trial_problem}); using index lists", file=sys.stderr)
                    payload_note += ", index lists (the bit-shard trial failed)"
                    ex[0] = make_exchange("indices")
            else:
                payload_note += ", index lists"
                ex[0] = make_exchange("indices")
        else:
            ex[0] = make_exchange(args.payload)
    # Clocks: a freshly initialised GPU needs tens of milliseconds of work before it runs at its sustained clocks, and the driver's 5
    # warm-up frames are 0.7 ms (measured round 5, the driver's own command on one box: 0.1545 ms per frame without the frames below,
    # 0.145 with them; --steps 200 --warmup 20: 0.143 without). The frames below are part of bringing the device up, like the mirror
    # upload above: untimed, in front of the W warm-up frames the contract asks for, for a fixed 0.3 s of wall clock. The line says so
    # where the driver keeps it: config.untimed_frames_before_warmup (second key). GV_BENCH_PREWARM_MS=0 switches them off.
    prewarm_ms, prewarm_frames = float(os.environ.get("GV_BENCH_PREWARM_MS", "300")), 0
    t_pre = time.perf_counter()
    # (with an exchange every frame is a collective: the ranks decide TOGETHER after each 50 frames whether to go on — clocks that
    # disagree by a millisecond must not leave one rank a chunk ahead, waiting in a collective nobody else enters)
    while prewarm_ms > 0:
        for _ in range(50):
            step()
        vis.wait()
        prewarm_frames += 50
        if native[0]:
            exchange_watchdog.pet(f"{prewarm_frames} untimed frames")
        more = (time.perf_counter() - t_pre) * 1e3 < prewarm_ms
        if world > 1:
            torch.cuda.synchronize()
            more = all_agree(more)
        if not more:
            break
    for _ in range(args.warmup):
        step()
    fence()
    upload_bytes = vis.stats()["upload_bytes"]
    vis.stats_reset()
    # A bracket around a kernel is two event records = ~12 us of stream time (the stream drains in front of each): the dominant
    # kernel is bracketed on 4-5 frames of the timed region, so that the region is the frame, not its instrumentation
    # (a short timed region — the driver's 20 frames — carries fewer of them: two brackets and three marks are 7 records = 2 us per
    # frame there, where four brackets and six marks were 4 us)
    short = args.steps < 100
    sample_every = 1 if args.profile_all else max(1, args.steps // (2 if short else 4))
    vis.profile_sampling(sample_every)
    mark_group = max(1, args.steps // (2 if short else 5))
    elapsed, per_step_ms, last = timed_steps(step, args.steps, mark_group)
    if native[0]:
        exchange_watchdog.pet("the timed frames")
    st = vis.stats()
    timed = vis.profile_samples()
    vis.profile_sampling(1)
    problem = check_padded(last, exact, exact_counts) if exchange else None
    if not all_agree(problem is None):
        if rank == 0:
            emit({"error": "exchange check failed", "detail": problem})
        leave(1)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    timed_payload = args.payload if exchange else None
    timed_exchange = ex[0]

    def max_over_ranks(x):
        if world == 1:
            return float(x)
        t = torch.tensor([x], dtype=torch.float64, device=f"cuda:{local_rank}" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def every_rank(x):
        """[x of rank 0, x of rank 1, ...] on every rank."""
        if world == 1:
            return [float(x)]
        dev = f"cuda:{local_rank}" if backend == "nccl" else "cpu"
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        out = torch.empty(world, dtype=torch.float64, device=dev)
        if backend == "nccl":
            dist.all_gather_into_tensor(out, t)
        else:
            pieces = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(pieces, t)
            out = torch.cat(pieces)
        return [float(v) for v in out.cpu()]

    # With an exchange: the same frames WITHOUT it (same ranks, same run) = what one GPU does with one tile of this workload
    # (`n1_same_workload`), and so what the collective costs and what the scaling efficiency is; one ISOLATED exchange (nothing
    # overlapped: shard copy + collective + completion, host clock); and the other shard encoding timed the same way.
    no_exchange = exchange_ms = mask_variant = torch_variant = mode_variants = None
    timed_native, timed_frame = native[0], (last if isinstance(last, dict) else None)
    if exchange:
        for _ in range(3):
            compute()
        e2, per2, _ = timed_steps(compute, args.steps, mark_group)
        per_rank_ms = every_rank(e2 / args.steps * 1e3)
        e2 = max_over_ranks(e2)
        no_exchange = dict(ms_per_step=e2 / args.steps * 1e3, value=n * world * args.steps / e2,
                           value_per_gpu=n * args.steps / e2, ms_per_step_by_rank=per_rank_ms,
                           ms_per_step_median_rank0=float(np.median(per2)))
        lat = []
        for _ in range(5):
            compute()
            fence()
            t0 = time.perf_counter()
            if native[0]:
                if args.payload == "mask":
                    vis.exchange_masks(0, mask_words(n), native_rows[0].data_ptr())
                else:
                    vis.exchange_acquire(native_frame()["frame"])
            else:
                shard = ex[0].next_shard()
                if args.payload == "mask":
                    vis.copy_mask_device(0, shard.data_ptr(), ex[0].capacity)
                else:
                    vis.copy_shard_device(0, shard.data_ptr(), ex[0].capacity, index_base=rank * n)
                ex[0].exchange()
                ex[0].drain()
            vis.wait()
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
        exchange_ms = max_over_ranks(float(np.median(lat)) * 1e3)
        exchange_watchdog.stop()  # (the library's exchange is not called again before the variants, which have their own guard)

        def timed_variant(describe):
            """args.steps frames of step() as currently configured, checked against the exact all-gatherv; a failing variant is
            reported, it does not take the headline with it."""
            problem, out = None, None
            try:
                for _ in range(3):
                    step()
                e3, _, last3 = timed_steps(step, args.steps, mark_group)
                problem = check_padded(last3, exact, exact_counts)
                e3 = max_over_ranks(e3)
                out = dict(ms_per_step=e3 / args.steps * 1e3, value=n * world * args.steps / e3,
                           shard_bytes_per_rank=[4 * w for w in shard_words_per_rank(ex[0], last3)],
                           checked_against_exact_allgatherv=problem is None, **describe)
            except Exception as e:  # noqa: BLE001
                problem = f"{type(e).__name__}: {e}"
            if not all_agree(problem is None):
                out = dict(error=problem or "failed on another rank", **describe)
            return out

        def run_exchange_variants():
            """The same frames as bit shards, by the other travel patterns and through torch.distributed — run LAST, under a watchdog
            (guarded below): the travel patterns other than the headline's first meet real links inside this function, and a
            collective that never returns must not take the measured line with it."""
            nonlocal mask_variant, mode_variants, torch_variant
            if timed_payload == "indices" and not args.no_mask_variant:
                ex[0] = make_exchange("mask")
                mask_variant = timed_variant(dict(
                    delivers="every rank holds every rank's [count, one bit per mirror entry]; the entry -> pool slot tables travelled "
                             "once at set-up (a consumer that wants the index list expands the rows)",
                    exchange_path="c-abi (gv_exchange_masks)" if native[0] else "torch.distributed"))
                args.payload, ex[0] = timed_payload, timed_exchange
            if timed_native and timed_payload == "indices" and world > 1 and not args.no_mode_variants:
                # the same frames with the rows travelling by the other patterns (A/B for the fully connected xGMI node) ...
                mode_variants = {}
                for mode in ("allgather", "p2p", "broadcast"):
                    if mode == args.exchange:
                        continue
                    vis.exchange_set_mode(EXCHANGE_MODES[mode])
                    mode_variants[mode] = timed_variant(dict(exchange_path="c-abi (gv_exchange_visible)"))
                vis.exchange_set_mode(EXCHANGE_MODES[args.exchange])
            if timed_native and timed_payload == "indices" and not args.no_torch_variant:
                # ... and through torch.distributed over this script's own buffers (what round 3 timed as the headline)
                native[0] = False
                ex[0] = make_exchange("indices")
                torch_variant = timed_variant(dict(exchange_path="torch.distributed (garden_amd/multi.py::VisibleListExchange)",
                                                   capacity_words=ex[0].capacity))
                ex[0].drain()
                native[0], ex[0] = True, timed_exchange

    # also report the rate when every TRS is re-uploaded each frame (host AoS -> mirror gather + PCIe
    # + cull). Outside the timed region; never `value`.
    dirty_rate = None
    stream_peak = None
    if world == 1:
        from garden_amd.lib import GV_DIRTY_TRANSFORM
        frames, t1 = 3, time.perf_counter()
        for _ in range(frames):
            vis.mark_dirty(GV_DIRTY_TRANSFORM, 0, n)
            compute()
        vis.wait()
        dirty_rate = n * frames / (time.perf_counter() - t1)
    if rank == 0:
        # this box's read-stream peak on the cull kernel's own access pattern (five streams, 65 B per entity)
        stream_peak = vis.stream_peak(0, 20)

    # per-kernel breakdown of a frame (pyramid / sweep / cull / emit), from a few frames OUTSIDE the timed region with every
    # kernel bracketed (the timed region brackets only the dominant kernel, on every fourth frame)
    frame_kernel_ms = None
    if not args.profile_all:
        from garden_amd.lib import KERNEL_NAMES
        for _ in range(2):  # what is (re)built once a pool is at rest again (the re-upload frames above moved it) is not a frame's cost
            compute()
        vis.wait()
        vis.profile_kernels(KERNEL_NAMES)
        vis.stats_reset()
        breakdown_frames = 10
        for _ in range(breakdown_frames):
            compute()
        s2 = vis.stats()
        frame_kernel_ms = {k: s2["device_ms"][k] / breakdown_frames for k in s2["device_ms"] if s2["device_ms"][k] > 0}
        vis.profile_kernels(["cull"])

    # The frame as an ENGINE runs it: the reference consumes a frame's list in that same frame — prepareMeshes
    # waits for its tasks and sorts (mesh.cpp:548-553), the render passes draw from the list (:556-600) — before the next frame's
    # depth exists. (a) the host waits for every frame's list (gv_result_count: a 4-byte read-back behind the frame's work);
    # (b) a device-side consumer ordered on the library's stream reads every frame's count (no host wait; since round 4 this IS the
    # headline's schedule: gv_cull enqueues everything a view's results consist of).
    engine_flow = None
    if world == 1:
        flow_frames = max(5, min(args.steps, 100))

        def flow(consume):
            for _ in range(3):
                compute()
                consume()
            vis.wait()
            t_flow = time.perf_counter()
            for _ in range(flow_frames):
                compute()
                consume()
            vis.wait()
            torch.cuda.synchronize()
            return (time.perf_counter() - t_flow) / flow_frames

        host_s = flow(lambda: vis.result_count(0))
        dres = vis.results_device(0)
        count_word = device_words(dres.draw_count, 1)
        total = torch.zeros(1, dtype=torch.int64, device=f"cuda:{local_rank}")
        torch.cuda.synchronize()  # (the fill runs on torch's stream; lib_stream is non-blocking)

        def device_consumer():
            with torch.cuda.stream(lib_stream):
                total.add_(count_word)

        dev_s = flow(device_consumer)
        engine_flow = dict(frames=flow_frames,
                           host_waits_for_every_list=dict(ms_per_step=host_s * 1e3, value=n / host_s,
                                                          consumer="gv_result_count after every gv_cull (the host blocks until the frame's list is complete, "
                                                                   "as MeshRenderSystem::prepareMeshes waits for its tasks, mesh.cpp:548)"),
                           device_consumer_on_the_stream=dict(ms_per_step=dev_s * 1e3, value=n / dev_s,
                                                              consumer="a one-word kernel on gv_stream() reads every frame's draw_count through the "
                                                                       "pointers of gv_results_device (fetched once); no host wait"),
                           note="`value` is the plain frame loop: the same schedule as device_consumer_on_the_stream minus the consumer's launch")

    # cfg3 on a HARD depth image: per-8x8-block occluders among the entities instead of 256 walls centimetres from
    # the camera — what the occlusion query costs when its coarse-level exits stop deciding. Own context, same pools and view.
    hard_depth = None
    under_profiler = "ROCPROF_OUTPUT_PATH" in os.environ or "rocprofiler-sdk-tool" in os.environ.get("LD_PRELOAD", "") or \
        any(k.startswith("ROCPROF_") for k in os.environ)
    if world == 1 and wl["hiz"] and args.depth == "walls" and not args.block_bounds and (args.no_hard_depth_variant or under_profiler):
        hard_depth = dict(skipped="--no-hard-depth-variant" if args.no_hard_depth_variant else
                          "under rocprofv3: the variant's launches are the dominant kernel under the same name on another depth image and would mix "
                          "into the profiler's per-kernel average; run `bench.py --depth noise` under the profiler for them (profiles/r04_cfg3hard_*)")
    elif world == 1 and wl["hiz"] and args.depth == "walls" and not args.block_bounds:
        hard = scene.noise_depth(HIZ_SIZE, HIZ_SIZE)
        vh = GpuVisibility(device=local_rank, profile_cull_only=True, linear_scan=True, hiz_rg16f=args.hiz_rg16f)
        vh.bind_transforms(sc.transforms, sc.entity_to_transform)
        vh.bind_pool(0, sc.meshes)
        vh.hierarchy_rebuild()
        vh.hiz_build(hard)

        def hard_step():
            vh.hiz_rebuild()
            vh.cull(0, view_array)

        for _ in range(5):
            hard_step()
        vh.wait()
        vh.stats_reset()
        vh.profile_sampling(8)
        frames, t4 = 40, time.perf_counter()
        for _ in range(frames):
            hard_step()
        vh.wait()
        dt = time.perf_counter() - t4
        sh, th = vh.stats(), vh.profile_samples()
        gh = vh.fetch(0, write_back=False, occupancy=n)
        vh.close()
        hard_depth = dict(depth="scene.noise_depth: one occluder per 8 x 8 pixel block, distance log-uniform in [50 m, 20 km] (among the entities)",
                          ms_per_step=dt / frames * 1e3, value=n * frames / dt, cull_kernel_ms=sh["device_ms"]["cull"] / max(1, th["cull"]),
                          visible_fraction=gh["draw_count"] / n, traffic=None)
        if not args.no_parity:
            from oracle import oracle_py as _orc
            th_threads = max(1, os.cpu_count() or 1)
            m3 = sc.meshes.copy()
            eh = _orc.prepare_meshes(m3, sc.transforms, sc.entity_to_transform, view,
                                     hiz=_orc.Hiz(hard, threads=th_threads, rg16f=args.hiz_rg16f), threads=th_threads)
            oh = np.argsort(eh["visible_idx"], kind="stable")
            hard_depth["visible_set_bit_identical"] = bool(np.array_equal(gh["visible_idx"], eh["visible_idx"][oh]) and
                                                           np.array_equal(gh["is_visible"], m3["isVisible"]))
            survivors_h = _orc.prepare_meshes(sc.meshes.copy(), sc.transforms, sc.entity_to_transform, dict(view, use_hiz=0), threads=th_threads)["draw_count"]
            ab_h = algorithmic_bytes(wl, n, survivors_h, gh["draw_count"], hard)
            hard_depth["frac"] = ab_h["cull"] / (hard_depth["cull_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if hard_depth["cull_kernel_ms"] > 0 else None
            hard_depth["algorithmic_bytes_per_launch"] = ab_h["cull"]
            try:  # counter bytes of the same kernel on this image, collected by tools/collect_traffic.sh (only while the kernel sources match)
                tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
                if tj.get("_kernel_source_sha") == kernel_source_sha() and "cfg3_hard_depth" in tj and n == tj["cfg3_hard_depth"].get("entities"):
                    hard_depth["traffic"] = tj["cfg3_hard_depth"]["cull_kernel_hbm_bytes_per_launch"]
            except (OSError, ValueError):
                pass
            if not hard_depth["visible_set_bit_identical"]:
                emit({"error": "cfg3 on the hard depth image: results differ from the CPU oracle", "variant": hard_depth})
                leave(1)

    # correctness gate + algorithmic byte counts
    got = vis.fetch(0, write_back=False, occupancy=n)

    # Same workload as the library runs it BY DEFAULT (round 3): block bounds — conservative workgroup-level frustum and Hi-Z
    # rejection, same results — for pools above 262144 slots. Reported beside the headline (`value_with_block_bounds`), never as
    # `value`: the headline stays the linear scan SURVEY.md §8d prices (GV_CONFIG_LINEAR_SCAN).
    bounds_variant = None
    if world == 1 and not args.block_bounds and not (wl["sweep"] and args.sweep.startswith("fused")):
        vb = GpuVisibility(device=local_rank, profile_cull_only=True, block_bounds=n <= 262144)
        vb.bind_transforms(sc.transforms, sc.entity_to_transform)
        vb.bind_pool(0, sc.meshes)
        vb.hierarchy_rebuild()
        if wl["hiz"]:
            vb.hiz_build(depth)

        def bounded_step():
            if wl["hiz"]:
                vb.hiz_rebuild()
            if wl["sweep"]:
                vb.sweep({"mfma": GV_SWEEP_MFMA, "valu": GV_SWEEP_VALU}[args.sweep])
            vb.cull(0, [view])

        for _ in range(5):
            bounded_step()
        vb.wait()
        vb.stats_reset()
        vb.profile_sampling(8)  # (a bracket costs ~12 us of stream time: a few of the 30 frames)
        frames, t2 = 30, time.perf_counter()
        for _ in range(frames):
            bounded_step()
        vb.wait()
        dt = time.perf_counter() - t2
        sb, tb = vb.stats(), vb.profile_samples()
        gb = vb.fetch(0, write_back=False, occupancy=n)
        same = bool(np.array_equal(gb["visible_idx"], got["visible_idx"]) and np.array_equal(gb["is_visible"], got["is_visible"])
                    and np.array_equal(gb["baked_model"].view(np.uint32), got["baked_model"].view(np.uint32)))
        bounds_variant = dict(ms_per_step=dt / frames * 1e3, value=n * frames / dt,
                              cull_kernel_ms=sb["device_ms"]["cull"] / max(1, tb["cull"]),
                              examined_workgroup_fraction=sb["bounds_blocks_examined"] / max(1, sb["bounds_blocks_total"]),
                              outputs_identical_to_headline=same)
        vb.close()
        if not same:
            emit({"error": "block-bounds variant differs from the linear scan", "variant": bounds_variant})
            leave(1)

    # cfg4: the bench default is the MFMA chain BASELINE.json names; the bit-identical v_fma chain is timed beside it
    valu_variant = None
    if world == 1 and wl["sweep"] and args.sweep == "fused":
        def valu_step():
            vis.sweep(GV_SWEEP_WITH_CULL_VALU)
            vis.cull(0, view_array)

        for _ in range(5):
            valu_step()
        vis.wait()
        vis.stats_reset()
        frames, t3 = 30, time.perf_counter()
        for _ in range(frames):
            valu_step()
        vis.wait()
        dt = time.perf_counter() - t3
        sv, tv = vis.stats(), vis.profile_samples()
        gvv = vis.fetch(0, write_back=False, occupancy=n)
        same = bool(np.array_equal(gvv["visible_idx"], got["visible_idx"]) and np.array_equal(gvv["is_visible"], got["is_visible"])
                    and np.array_equal(gvv["baked_model"].view(np.uint32), got["baked_model"].view(np.uint32)))
        valu_variant = dict(kernel="gv::sweep_cull_valu_kernel", ms_per_step=dt / frames * 1e3, value=n * frames / dt,
                            avg_launch_ms=sv["device_ms"]["cull"] / max(1, tv["cull"]), outputs_identical=same)
        if not same:
            emit({"error": "cfg4: the VALU chain's outputs differ from the MFMA chain's", "variant": valu_variant})
            leave(1)

    visible = got["draw_count"]
    parity = None
    survivors = visible
    parity_ok = True
    # EVERY rank checks its own tile against the oracle (the host's cores shared between the ranks); the verdict is all-reduced
    from oracle import oracle_py
    cores = os.cpu_count() or 1
    threads = max(1, cores // world)
    if rank == 0 and wl["hiz"]:  # frustum survivors of rank 0's tile: the Hi-Z texel term of the roofline numerator
        m2 = sc.meshes.copy()
        survivors = oracle_py.prepare_meshes(m2, sc.transforms, sc.entity_to_transform, dict(view, use_hiz=0), threads=threads)["draw_count"]
    if not args.no_parity:
        m2 = sc.meshes.copy()
        exp = oracle_py.prepare_meshes(m2, sc.transforms, sc.entity_to_transform, view,
                                       hiz=oracle_py.Hiz(depth, threads=threads, rg16f=args.hiz_rg16f) if wl["hiz"] else None, threads=threads)
        order = np.argsort(exp["visible_idx"], kind="stable")
        same_set = bool(np.array_equal(got["visible_idx"], exp["visible_idx"][order]))
        same_vis = bool(np.array_equal(got["is_visible"], m2["isVisible"]))
        same_mat = same_set and bool(np.array_equal(got["baked_model"].view(np.uint32), exp["baked_model"][order].view(np.uint32)))
        parity_ok = same_set and same_vis and same_mat
        verdicts = every_rank((1 if same_set else 0) + (2 if same_vis else 0) + (4 if same_mat else 0))
        counts_by_rank = every_rank(visible)
        parity = dict(visible_set_bit_identical=all(int(v) & 1 for v in verdicts), is_visible_identical=all(int(v) & 2 for v in verdicts),
                      baked_model_bit_identical=all(int(v) & 4 for v in verdicts), visible=int(sum(counts_by_rank)),
                      visible_by_rank=[int(c) for c in counts_by_rank], checked_entities=int(n) * world, checked_ranks=world,
                      oracle_threads_per_rank=threads)
    if not all_agree(parity_ok):
        if rank == 0:
            emit({"error": "results differ from the CPU oracle on some rank", "parity": parity})
        vis.close()
        leave(1)

    if rank == 0:
        fused = wl["sweep"] and args.sweep.startswith("fused")
        examined = 1.0
        if args.block_bounds and st["bounds_blocks_total"]:
            examined = st["bounds_blocks_examined"] / st["bounds_blocks_total"]
        ab = algorithmic_bytes(wl, n, survivors, visible, depth, fused=fused, examined=examined)
        cull_ms = st["device_ms"]["cull"] / max(1, timed["cull"])
        achieved = ab["cull"] / (cull_ms * 1e-3) / 1e9 if cull_ms > 0 else 0.0
        if valu_variant and valu_variant["avg_launch_ms"] > 0:  # same algorithmic bytes, the other chain
            valu_variant["frac"] = ab["cull"] / (valu_variant["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        # roofline.traffic: PMC bytes of this kernel from profiles/traffic.json — only while the kernel sources still
        # hash to what they were when the counters were collected
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = {"cfg5": "cfg2_at_10M", "cfg2": "cfg2_at_10M"}.get(args.workload, args.workload)
                if args.workload == "cfg3" and args.depth == "noise":
                    key = "cfg3_hard_depth"
                entry = tj.get(key, {})
                now = kernel_source_sha()
                traffic_source = {"file": "profiles/traffic.json", "entry": key, "collected": tj.get("_collected"),
                                  "kernel_source_sha_at_collection": tj.get("_kernel_source_sha"),
                                  "kernel_source_sha_now": now,
                                  "per_entity_scaled": False}
                if tj.get("_kernel_source_sha") == now and not args.block_bounds and not args.hiz_rg16f and "cull_kernel_hbm_bytes_per_launch" in entry:
                    traffic = entry["cull_kernel_hbm_bytes_per_launch"]
                    measured_n = entry.get("entities", 10_000_000)
                    if measured_n != n:  # counters were taken at another pool size of the same streaming kernel
                        traffic = traffic * n / measured_n
                        traffic_source["per_entity_scaled"] = True
            except Exception:
                traffic, traffic_source = None, None
        median_ms = float(np.median(per_step_ms))
        out = {
            "metric": "entity culls/sec at 10M entities; visible-set bit-match vs CPU ref",
            "value": n * world * args.steps / elapsed,
            "unit": "entity culls/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            # from one hipEvent per `frames_per_mark` frame boundaries on the library's stream (rank 0): median / min / max of the
            # groups' per-frame means (an event per frame costs ~6 us of stream time per frame: not in the timed region)
            "frames_per_mark": mark_group,
            "ms_per_step_median": median_ms, "ms_per_step_min": float(per_step_ms.min()), "ms_per_step_max": float(per_step_ms.max()),
            "value_at_median_step": n * world / (median_ms * 1e-3),
            # N > 1: ONE GPU on ONE tile of this same workload (these ranks' frames without the exchange step, slowest rank) and
            # value / (N * that). Weak scaling: the efficiency of the job; strong: of the exchange only (the N = 1 run of the same
            # --entities-total is the other half: tools/scale_sweep.sh puts the runs side by side)
            "n1_same_workload": ({"value": no_exchange["value_per_gpu"], "unit": "entity culls/s", "ms_per_step": no_exchange["ms_per_step"],
                                  "entities": n, "what": "the same ranks' frames without the exchange step, same run (slowest rank)"}
                                 if no_exchange and world > 1 else None),
            "scaling_efficiency": ((n * world * args.steps / elapsed) / (world * no_exchange["value_per_gpu"])) if no_exchange and world > 1 else None,
            # the same frame through GV_CONFIG_BLOCK_BOUNDS (same outputs, checked): reported beside `value`, never as it
            "value_with_block_bounds": bounds_variant["value"] if bounds_variant else None,
            "config": {"workload": (wl["name"] if args.scaling == "weak" else
                                    f"{args.workload}-strong: ONE world of {n * world} entities cut into {world} spatial tile(s), "
                                    f"{n} per GPU; per tile as {wl['name']}"),
                       "untimed_frames_before_warmup": prewarm_frames,  # device bring-up to sustained clocks (0.3 s); then `warmup` frames, then the timed ones
                       "sweep": args.sweep if wl["sweep"] else None,
                       "block_bounds": {"examined_workgroup_fraction": examined} if args.block_bounds else None,
                       "block_bounds_variant": bounds_variant, "valu_variant": valu_variant,
                       "engine_flow_variant": engine_flow, "hard_depth_variant": hard_depth, "depth": args.depth if wl["hiz"] else None, "entities_per_gpu": n, "entities_total": n * world,
                       "scaling_mode": (f"strong: one world of {n * world} entities cut into {world} spatial tile(s)" if args.scaling == "strong"
                                        else f"weak: {n} entities per GPU"),
                       "visible_fraction": visible / n, "hiz": (f"{HIZ_SIZE}x{HIZ_SIZE}" + (" RG16F" if args.hiz_rg16f else "")) if wl["hiz"] else None,
                       "exchange": ((f"per frame, through the library's C-ABI (gv_exchange_visible): shard [count, uint32 indices...] of every rank "
                                     f"into library-owned rows (row stride {timed_frame['row_words']} words; room per rank {timed_frame['room']}, sized from the "
                                     f"headers of earlier frames, which reach the host through pinned memory) by {args.exchange} behind the cull "
                                     f"stream, no host sync; RCCL bound by the library"
                                     + (f" [{transport_note}: N ranks share a GPU, functional only]" if transport_note else "") +
                                     f"; {gathered_total} indices gathered per rank; checked against the exact all-gatherv")
                                    if timed_frame else
                                    (f"per frame: " + ("shards [count, one bit per mirror entry] " if args.payload == "mask" else "padded shards [count, uint32 indices...] ") +
                                     (f"(capacity {ex[0].capacity} words) travel by {ex[0].describe()} behind the cull stream, no host sync ({backend})"
                                      if ex[0] is not None else f"({mask_words(n)} words) through the library's C-ABI (gv_exchange_masks) by {args.exchange}") +
                                     f"; {gathered_total} indices gathered per rank; checked against the exact all-gatherv")) if exchange else None,
                       # who runs the timed exchange: the product's own C-ABI step, or torch.distributed over this script's buffers
                       "exchange_path": (("c-abi" if timed_native else "torch") if exchange else None),
                       "exchange_path_fallback": (path_fallback[0] if exchange else None),
                       "exchange_transport": ((transport_note or "RCCL (dlopen'ed by the library)") if exchange and timed_native else ("torch.distributed " + backend if exchange else None)),
                       "exchange_mode": args.exchange if exchange else None,
                       "exchange_payload": (args.payload + (f" ({payload_note})" if payload_note else "")) if exchange else None,
                       "same_frames_without_exchange": no_exchange,
                       # one isolated exchange (shard copy + collective + completion; nothing overlapped; host clock, slowest rank)
                       "exchange_ms": exchange_ms,
                       # what the exchange adds to a frame when it runs behind the next frame's cull (two slots in flight)
                       "exchange_overhead_ms_per_step": (elapsed / args.steps * 1e3 - no_exchange["ms_per_step"]) if no_exchange else None,
                       # bytes rank r's shard puts on each link per frame as it travels (padding included) / of those, list entries
                       "shard_bytes_per_rank": [4 * w for w in shard_words_per_rank(ex[0], timed_frame)] if exchange else None,
                       "list_bytes_per_rank": ([4 * (1 + int(c)) for c in exact_counts] if exchange else None),
                       "gathered_bytes_per_rank": (4 * sum(shard_words_per_rank(ex[0], timed_frame))) if exchange else None,
                       "gathered_over_list_bytes": ((sum(shard_words_per_rank(ex[0], timed_frame)) / float(sum(1 + int(c) for c in exact_counts)))
                                                    if exchange and timed_payload == "indices" else None),
                       "visible_max_over_mean_by_rank": (float(exact_counts.max() / max(1.0, exact_counts.mean())) if exchange else None),
                       # the same frames by the other travel patterns / through torch.distributed (same run, each checked)
                       "exchange_mode_variants": mode_variants, "torch_variant": torch_variant,
                       "mask_variant": mask_variant,
                       # per frame, in the TIMED region: only the bracketed kernels appear (default: the dominant one; --profile-all: all)
                       "kernel_ms": {k: (st["device_ms"][k] * (st["launches"][k] / max(1, timed[k])) / max(1, args.steps))
                                     for k in st["device_ms"] if st["device_ms"][k] > 0},
                       # per frame, every kernel, from 10 extra frames outside the timed region (hipEvents around each launch)
                       "frame_kernel_ms": frame_kernel_ms,
                       # untimed frames run before the W warm-up frames so that the device is at its sustained clocks (0.3 s of wall clock)
                       "prewarm_frames": prewarm_frames,
                       # the library's exchange: every acquired frame is complete; how many needed the second, exactly sized exchange
                       # of tails to be so (a static camera: the first frame, which has no history to predict from)
                       "exchange_frames_acquired": exchange_frames[0] if timed_native else None,
                       "exchange_frames_completed_by_a_second_exchange": exchange_frames[1] if timed_native else None,
                       "mirror_upload_s": upload_s, "mirror_upload_bytes": upload_bytes,
                       "culls_per_s_with_full_trs_upload_each_frame": dirty_rate},
            "roofline": {"bound": "hbm", "kernel": ("gv::sweep_cull_mfma_kernel" if args.sweep == "fused" else "gv::sweep_cull_valu_kernel") if fused else "gv::cull_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": ab["cull"], "avg_launch_ms": cull_ms,
                         "launches_timed": f"{timed['cull']} of {st['launches']['cull']} in the timed region (hipEvents around every "
                                           f"{sample_every}{'st' if sample_every == 1 else 'th'} launch)",
                         # SURVEY.md §8d asks for both peaks: the vendor figure above and what a read-only kernel over this
                         # kernel's five streams reaches on THIS box, measured in this run (gv_debug_stream_peak)
                         "measured_stream_peak": stream_peak,
                         "frac_of_measured_peak": (achieved / stream_peak) if stream_peak else None},
            "parity": parity,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, sc, view, depth)
    if exchange:
        import threading

        def bark():  # a variant's collective never came back: the line goes out with what had been measured, every rank leaves
            if rank == 0:
                out["config"]["variants_aborted"] = ("a variant timed beside the headline (bit shards / other travel patterns / torch.distributed) did "
                                                     "not finish within 300 s; the line carries what had been measured before it")
                emit(out)
            os._exit(0)

        watchdog = threading.Timer(300.0, bark)
        watchdog.daemon = True
        watchdog.start()
        try:
            run_exchange_variants()
        finally:
            watchdog.cancel()
        if rank == 0:
            out["config"].update(mask_variant=mask_variant, exchange_mode_variants=mode_variants, torch_variant=torch_variant)
    if rank == 0:
        emit(out)
    vis.close()
    leave(0)


if __name__ == "__main__":
    main()
