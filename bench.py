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

from garden_amd.benchlib.workloads import HBM_PEAK_GBS, HIZ_SIZE, WORKLOADS, algorithmic_bytes, counter_traffic, make_tile_scene  # noqa: E402


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


def parse_args():
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
    ap.add_argument("--exchange", default=os.environ.get("GV_BENCH_EXCHANGE_MODE"), choices=["allgather", "p2p", "broadcast"],
                    help="N > 1: how the shards travel — one equal-size all-gather, grouped point-to-point send/recv to every peer, or one "
                         "broadcast per root. Not given (default): five frames of each are timed before the warm-up and the fastest one "
                         "carries the timed frames (config.exchange_mode names it, config.exchange_mode_probe_ms holds the three figures)")
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
    return args


def cpu_baseline(wl, sc, view, depth):
    """The `cpu_baseline` leg: the oracle's AVX2 path (+ scalar pyramid / sweep) timed on this box's host cores (oracle/cpu_baseline.py)."""
    from oracle.cpu_baseline import cpu_baseline as timed_oracle
    return timed_oracle(wl, sc, view, depth)


def main():
    args = parse_args()
    # ---- launch shape, checked before torch is imported or the GPU is touched ----
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    if env_world is not None and int(env_world) != args.gpus:
        print(json.dumps({"error": f"--gpus {args.gpus} but WORLD_SIZE={env_world}: launch with "
                                   f"torch.distributed.run --nproc-per-node {args.gpus}, or run plain "
                                   f"`python bench.py --gpus {args.gpus}` (it starts the ranks itself)"}), flush=True)
        sys.exit(2)
    if args.workload is None:  # (strong scaling: the same workload at every N, N = 1 included)
        args.workload = "cfg5" if args.gpus > 1 or args.scaling == "strong" else "cfg3"

    # The contract is ONE JSON line on stdout. Libraries underneath (RCCL prints a version banner when a communicator
    # is created) write to file descriptor 1 too, so everything but the result lines is sent to stderr.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    from garden_amd.benchlib.run import Run
    from garden_amd.benchlib.watchdog import ExchangeWatchdog, guarded
    run = Run(args, result_fd)
    torch, rank, world, local_rank = run.torch, run.rank, run.world, run.local_rank
    exchange = run.exchange
    exchange_watchdog = ExchangeWatchdog(float(os.environ.get("GV_BENCH_EXCHANGE_WATCHDOG_S", "240")), rank, result_fd, os.path.abspath(__file__))

    from garden_amd import scene
    from garden_amd.benchlib import variants
    from garden_amd.lib import GpuVisibility, GV_SWEEP_MFMA, GV_SWEEP_VALU, GV_SWEEP_WITH_CULL, GV_SWEEP_WITH_CULL_VALU

    wl = run.wl = WORKLOADS[args.workload]
    n = args.entities or wl["entities"]
    if args.scaling == "strong":  # one world of --entities-total cut into `world` spatial tiles (the same world at every N)
        n = args.entities_total // world
    run.n = n
    sc = run.sc = make_tile_scene(wl, n, rank, world)
    view = run.view = scene.main_camera_view(use_hiz=1 if wl["hiz"] else 0)
    depth = run.depth = ((scene.noise_depth(HIZ_SIZE, HIZ_SIZE) if args.depth == "noise" else scene.synthetic_depth(HIZ_SIZE, HIZ_SIZE))
                         if wl["hiz"] else None)

    # hipEvents bracket only the dominant kernel inside the timed region (each event record costs ~2 us of
    # stream time); --profile-all brackets every kernel for the per-kernel breakdown in config.kernel_ms
    vis = run.vis = GpuVisibility(device=local_rank, profile_events=args.profile_all, profile_cull_only=not args.profile_all,
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
    run.lib_stream = torch.cuda.ExternalStream(vis.stream(), device=torch.device("cuda", local_rank))
    view_array = run.view_array = vis.views_array([view])  # the GvView structs of the frame, built once (the timed loop is the library's, not ctypes')
    sweep_mode = {"mfma": GV_SWEEP_MFMA, "valu": GV_SWEEP_VALU, "fused": GV_SWEEP_WITH_CULL, "fused-valu": GV_SWEEP_WITH_CULL_VALU}[args.sweep]

    def compute():
        if wl["hiz"]:
            vis.hiz_rebuild()
        if wl["sweep"]:
            vis.sweep(sweep_mode)
        vis.cull(0, view_array)

    fx = None  # the exchange step of N > 1 (garden_amd/benchlib/exchange.py)

    def step():
        """One frame; with an exchange the rank's list then goes out to every rank (FrameExchange.after_compute)."""
        compute()
        return fx.after_compute() if fx is not None else None

    run.compute, run.step = compute, step
    if exchange:
        from garden_amd.benchlib.exchange import FrameExchange
        fx = FrameExchange(run, exchange_watchdog, ROOT)
        fx.setup()
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
        if fx is not None and fx.native:
            exchange_watchdog.pet(f"{prewarm_frames} untimed frames")
        more = (time.perf_counter() - t_pre) * 1e3 < prewarm_ms
        if world > 1:
            torch.cuda.synchronize()
            more = run.all_agree(more)
        if not more:
            break
    for _ in range(args.warmup):
        step()
    run.fence()
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
    # ---- the timed region: EXACTLY args.steps frames between two fences (barrier + synchronise), the slowest rank's clock ----
    elapsed, per_step_ms, last = run.timed_steps(step, args.steps, mark_group)
    if fx is not None and fx.native:
        exchange_watchdog.pet("the timed frames")
    st = vis.stats()
    timed = vis.profile_samples()
    vis.profile_sampling(1)
    problem = fx.check_padded(last) if exchange else None
    if not run.all_agree(problem is None):
        if rank == 0:
            run.emit({"error": "exchange check failed", "detail": problem})
        run.leave(1)
    elapsed = run.max_over_ranks(elapsed)

    # With an exchange: the same frames WITHOUT it (same ranks, same run) = what one GPU does with one tile of this workload
    # (`n1_same_workload`), and so what the collective costs and what the scaling efficiency is; one ISOLATED exchange.
    no_exchange = exchange_ms = mask_variant = torch_variant = mode_variants = None
    timed_payload = args.payload if exchange else None
    timed_exchange, timed_native = (fx.ex, fx.native) if exchange else (None, False)
    timed_frame = last if isinstance(last, dict) else None
    if exchange:
        for _ in range(3):
            compute()
        e2, per2, _ = run.timed_steps(compute, args.steps, mark_group)
        per_rank_ms = run.every_rank(e2 / args.steps * 1e3)
        e2 = run.max_over_ranks(e2)
        no_exchange = dict(ms_per_step=e2 / args.steps * 1e3, value=n * world * args.steps / e2,
                           value_per_gpu=n * args.steps / e2, ms_per_step_by_rank=per_rank_ms,
                           ms_per_step_median_rank0=float(np.median(per2)))
        exchange_ms = fx.isolated_ms()
        exchange_watchdog.stop()  # (the library's exchange is not called again before the variants, which have their own guard)

    # ---- beside the headline, outside the timed region (garden_amd/benchlib/variants.py) ----
    dirty_rate = variants.dirty_rate(run) if world == 1 else None
    stream_peak = vis.stream_peak(0, 20) if rank == 0 else None  # this box's read-stream peak on the cull kernel's own access pattern
    frame_kernel_ms = variants.frame_kernel_ms(run) if not args.profile_all else None
    engine_flow = variants.engine_flow(run) if world == 1 else None

    threads_all = max(1, os.cpu_count() or 1)

    def oracle_frame(depth_image, use_hiz, threads=threads_all, with_models=False):
        """The parity leg: the CPU oracle's frame over this rank's pools -> (visible_idx ascending, isVisible, draw count[, models])."""
        from oracle import oracle_py
        m = sc.meshes.copy()
        hz = oracle_py.Hiz(depth_image, threads=threads, rg16f=args.hiz_rg16f) if use_hiz and depth_image is not None else None
        exp = oracle_py.prepare_meshes(m, sc.transforms, sc.entity_to_transform, dict(view, use_hiz=1 if hz else 0), hiz=hz, threads=threads)
        order = np.argsort(exp["visible_idx"], kind="stable")
        out = (exp["visible_idx"][order], m["isVisible"], exp["draw_count"])
        return out + (exp["baked_model"][order],) if with_models else out

    hard_depth = None
    if world == 1 and wl["hiz"] and args.depth == "walls" and not args.block_bounds:
        hard_depth = variants.hard_depth(run, ROOT, oracle_frame)
    got = vis.fetch(0, write_back=False, occupancy=n)  # correctness gate + algorithmic byte counts
    fused = wl["sweep"] and args.sweep.startswith("fused")
    bounds_variant = variants.block_bounds(run, got) if world == 1 and not args.block_bounds and not fused else None
    valu_variant = variants.valu_chain(run, got) if world == 1 and wl["sweep"] and args.sweep == "fused" else None

    # ---- parity: EVERY rank checks its own tile against the oracle (the host's cores shared between the ranks); the verdict is all-reduced
    visible = got["draw_count"]
    parity, survivors, parity_ok = None, visible, True
    threads = max(1, threads_all // world)
    if rank == 0 and wl["hiz"]:  # frustum survivors of rank 0's tile: the Hi-Z texel term of the roofline numerator
        survivors = oracle_frame(None, 0, threads)[2]
    if not args.no_parity:
        want_idx, want_vis, _, want_models = oracle_frame(depth, 1 if wl["hiz"] else 0, threads, with_models=True)
        same_set = bool(np.array_equal(got["visible_idx"], want_idx))
        same_vis = bool(np.array_equal(got["is_visible"], want_vis))
        same_mat = same_set and bool(np.array_equal(got["baked_model"].view(np.uint32), want_models.view(np.uint32)))
        parity_ok = same_set and same_vis and same_mat
        verdicts = run.every_rank((1 if same_set else 0) + (2 if same_vis else 0) + (4 if same_mat else 0))
        counts_by_rank = run.every_rank(visible)
        parity = dict(visible_set_bit_identical=all(int(v) & 1 for v in verdicts), is_visible_identical=all(int(v) & 2 for v in verdicts),
                      baked_model_bit_identical=all(int(v) & 4 for v in verdicts), visible=int(sum(counts_by_rank)),
                      visible_by_rank=[int(c) for c in counts_by_rank], checked_entities=int(n) * world, checked_ranks=world,
                      oracle_threads_per_rank=threads)
    if not run.all_agree(parity_ok):
        if rank == 0:
            run.emit({"error": "results differ from the CPU oracle on some rank", "parity": parity})
        vis.close()
        run.leave(1)

    if rank == 0:
        examined = 1.0
        if args.block_bounds and st["bounds_blocks_total"]:
            examined = st["bounds_blocks_examined"] / st["bounds_blocks_total"]
        ab = algorithmic_bytes(wl, n, survivors, visible, depth, fused=fused, examined=examined)
        cull_ms = st["device_ms"]["cull"] / max(1, timed["cull"])
        achieved = ab["cull"] / (cull_ms * 1e-3) / 1e9 if cull_ms > 0 else 0.0
        if valu_variant and valu_variant["avg_launch_ms"] > 0:  # same algorithmic bytes, the other chain
            valu_variant["frac"] = ab["cull"] / (valu_variant["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        traffic, traffic_source = counter_traffic(ROOT, args, n)
        median_ms = float(np.median(per_step_ms))
        words = fx.shard_words_per_rank(fx.ex, timed_frame) if exchange else None
        counts = fx.exact_counts if exchange else None
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
                       "exchange": fx.describe(timed_frame) if exchange else None,
                       # who runs the timed exchange: the product's own C-ABI step, or torch.distributed over this script's buffers
                       "exchange_path": (("c-abi" if timed_native else "torch") if exchange else None),
                       "exchange_path_fallback": (fx.path_fallback if exchange else None),
                       "exchange_transport": ((fx.transport_note or "RCCL (dlopen'ed by the library)") if exchange and timed_native else ("torch.distributed " + run.backend if exchange else None)),
                       # the travel pattern of the timed frames: --exchange, or the fastest of the three over five frames each (the probe's ms per frame)
                       "exchange_mode": fx.mode if exchange else None,
                       "exchange_mode_probe_ms": fx.mode_probe_ms if exchange else None,
                       "exchange_payload": (args.payload + (f" ({fx.payload_note})" if fx.payload_note else "")) if exchange else None,
                       "same_frames_without_exchange": no_exchange,
                       # one isolated exchange (shard copy + collective + completion; nothing overlapped; host clock, slowest rank)
                       "exchange_ms": exchange_ms,
                       # what the exchange adds to a frame when it runs behind the next frame's cull (two slots in flight)
                       "exchange_overhead_ms_per_step": (elapsed / args.steps * 1e3 - no_exchange["ms_per_step"]) if no_exchange else None,
                       # bytes rank r's shard puts on each link per frame as it travels (padding included) / of those, list entries
                       "shard_bytes_per_rank": [4 * w for w in words] if exchange else None,
                       "list_bytes_per_rank": ([4 * (1 + int(c)) for c in counts] if exchange else None),
                       "gathered_bytes_per_rank": (4 * sum(words)) if exchange else None,
                       "gathered_over_list_bytes": ((sum(words) / float(sum(1 + int(c) for c in counts))) if exchange and timed_payload == "indices" else None),
                       "visible_max_over_mean_by_rank": (float(counts.max() / max(1.0, counts.mean())) if exchange else None),
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
                       "exchange_frames_acquired": fx.frames_acquired if timed_native else None,
                       "exchange_frames_completed_by_a_second_exchange": fx.frames_completed_late if timed_native else None,
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
        def give_up():  # a variant's collective never came back: the line goes out with what had been measured, every rank leaves
            if rank == 0:
                out["config"]["variants_aborted"] = ("a variant timed beside the headline (bit shards / other travel patterns / torch.distributed) did "
                                                     "not finish within 300 s; the line carries what had been measured before it")
                run.emit(out)
            os._exit(0)

        def timed_variants():
            nonlocal mask_variant, mode_variants, torch_variant
            mask_variant, mode_variants, torch_variant = variants.exchange_variants(run, fx, mark_group, timed_payload, timed_exchange, timed_native)

        guarded(300.0, give_up, timed_variants)
        if rank == 0:
            out["config"].update(mask_variant=mask_variant, exchange_mode_variants=mode_variants, torch_variant=torch_variant)
    if rank == 0:
        run.emit(out)
    vis.close()
    run.leave(0)


if __name__ == "__main__":
    main()
