"""What bench.py times BESIDE the headline, outside the timed region: the same frame under other schedules, encodings, depth images
and code paths. A variant never becomes `value`; one whose outputs differ from the headline's ends the run with an error line."""
import os
import time

import numpy as np

from garden_amd import scene
from garden_amd.lib import (GpuVisibility, GV_DIRTY_TRANSFORM, GV_SWEEP_MFMA, GV_SWEEP_VALU, GV_SWEEP_WITH_CULL_VALU, KERNEL_NAMES)
from garden_amd.benchlib.exchange import EXCHANGE_MODES
from garden_amd.benchlib.workloads import HBM_PEAK_GBS, HIZ_SIZE, algorithmic_bytes, counter_traffic


def dirty_rate(run):
    """the rate when every TRS is re-uploaded each frame (host AoS -> mirror gather + PCIe + cull). Never `value`."""
    frames, t1 = 3, time.perf_counter()
    for _ in range(frames):
        run.vis.mark_dirty(GV_DIRTY_TRANSFORM, 0, run.n)
        run.compute()
    run.vis.wait()
    return run.n * frames / (time.perf_counter() - t1)


def frame_kernel_ms(run):
    """Per-kernel breakdown of a frame (pyramid / sweep / cull / emit), from a few frames OUTSIDE the timed region with every
    kernel bracketed (the timed region brackets only the dominant kernel, on every fourth frame)."""
    vis = run.vis
    for _ in range(2):  # what is (re)built once a pool is at rest again (the re-upload frames moved it) is not a frame's cost
        run.compute()
    vis.wait()
    vis.profile_kernels(KERNEL_NAMES)
    vis.stats_reset()
    frames = 10
    for _ in range(frames):
        run.compute()
    s2 = vis.stats()
    out = {k: s2["device_ms"][k] / frames for k in s2["device_ms"] if s2["device_ms"][k] > 0}
    vis.profile_kernels(["cull"])
    return out


def engine_flow(run):
    """The frame as an ENGINE runs it: the reference consumes a frame's list in that same frame — prepareMeshes
    waits for its tasks and sorts (mesh.cpp:548-553), the render passes draw from the list (:556-600) — before the next frame's
    depth exists. (a) the host waits for every frame's list (gv_result_count: a 4-byte read-back behind the frame's work);
    (b) a device-side consumer ordered on the library's stream reads every frame's count (no host wait; since round 4 this IS the
    headline's schedule: gv_cull enqueues everything a view's results consist of)."""
    vis, torch, n = run.vis, run.torch, run.n
    flow_frames = max(5, min(run.args.steps, 100))

    def flow(consume):
        for _ in range(3):
            run.compute()
            consume()
        vis.wait()
        t_flow = time.perf_counter()
        for _ in range(flow_frames):
            run.compute()
            consume()
        vis.wait()
        torch.cuda.synchronize()
        return (time.perf_counter() - t_flow) / flow_frames

    host_s = flow(lambda: vis.result_count(0))
    count_word = run.device_words(vis.results_device(0).draw_count, 1)
    total = torch.zeros(1, dtype=torch.int64, device=run.device)
    torch.cuda.synchronize()  # (the fill runs on torch's stream; lib_stream is non-blocking)

    def device_consumer():
        with torch.cuda.stream(run.lib_stream):
            total.add_(count_word)

    dev_s = flow(device_consumer)
    return dict(frames=flow_frames,
                host_waits_for_every_list=dict(ms_per_step=host_s * 1e3, value=n / host_s,
                                               consumer="gv_result_count after every gv_cull (the host blocks until the frame's list is complete, "
                                                        "as MeshRenderSystem::prepareMeshes waits for its tasks, mesh.cpp:548)"),
                device_consumer_on_the_stream=dict(ms_per_step=dev_s * 1e3, value=n / dev_s,
                                                   consumer="a one-word kernel on gv_stream() reads every frame's draw_count through the "
                                                            "pointers of gv_results_device (fetched once); no host wait"),
                note="`value` is the plain frame loop: the same schedule as device_consumer_on_the_stream minus the consumer's launch")


def hard_depth(run, root, oracle_frame):
    """cfg3 on a HARD depth image: per-8x8-block occluders among the entities instead of 256 walls centimetres from
    the camera — what the occlusion query costs when its coarse-level exits stop deciding. Own context, same pools and view.
    oracle_frame(depth_image, use_hiz) -> (expected visible_idx ascending, expected isVisible, draw count): bench.py's parity leg."""
    args, n, sc = run.args, run.n, run.sc
    under_profiler = "ROCPROF_OUTPUT_PATH" in os.environ or "rocprofiler-sdk-tool" in os.environ.get("LD_PRELOAD", "") or \
        any(k.startswith("ROCPROF_") for k in os.environ)
    if args.no_hard_depth_variant or under_profiler:
        return dict(skipped="--no-hard-depth-variant" if args.no_hard_depth_variant else
                    "under rocprofv3: the variant's launches are the dominant kernel under the same name on another depth image and would mix "
                    "into the profiler's per-kernel average; run `bench.py --depth noise` under the profiler for them (profiles/r04_cfg3hard_*)")
    hard = scene.noise_depth(HIZ_SIZE, HIZ_SIZE)
    vh = GpuVisibility(device=run.local_rank, profile_cull_only=True, linear_scan=True, hiz_rg16f=args.hiz_rg16f)
    vh.bind_transforms(sc.transforms, sc.entity_to_transform)
    vh.bind_pool(0, sc.meshes)
    vh.hierarchy_rebuild()
    vh.hiz_build(hard)

    def hard_step():
        vh.hiz_rebuild()
        vh.cull(0, run.view_array)

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
    out = dict(depth="scene.noise_depth: one occluder per 8 x 8 pixel block, distance log-uniform in [50 m, 20 km] (among the entities)",
               ms_per_step=dt / frames * 1e3, value=n * frames / dt, cull_kernel_ms=sh["device_ms"]["cull"] / max(1, th["cull"]),
               visible_fraction=gh["draw_count"] / n, traffic=None)
    if not args.no_parity:
        want_idx, want_vis, _ = oracle_frame(hard, 1)
        out["visible_set_bit_identical"] = bool(np.array_equal(gh["visible_idx"], want_idx) and np.array_equal(gh["is_visible"], want_vis))
        survivors_h = oracle_frame(hard, 0)[2]
        ab_h = algorithmic_bytes(run.wl, n, survivors_h, gh["draw_count"], hard)
        out["frac"] = ab_h["cull"] / (out["cull_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if out["cull_kernel_ms"] > 0 else None
        out["algorithmic_bytes_per_launch"] = ab_h["cull"]
        # counter bytes of the same kernel on this image, collected by tools/collect_traffic.sh (only while the kernel sources match)
        class _HardArgs:
            workload, depth, block_bounds, hiz_rg16f = "cfg3", "noise", False, args.hiz_rg16f
        traffic, source = counter_traffic(root, _HardArgs, n)
        if traffic is not None and not source["per_entity_scaled"]:
            out["traffic"] = traffic
        if not out["visible_set_bit_identical"]:
            run.emit({"error": "cfg3 on the hard depth image: results differ from the CPU oracle", "variant": out})
            run.leave(1)
    return out


def same_outputs(a, b):
    return bool(np.array_equal(a["visible_idx"], b["visible_idx"]) and np.array_equal(a["is_visible"], b["is_visible"])
                and np.array_equal(a["baked_model"].view(np.uint32), b["baked_model"].view(np.uint32)))


def block_bounds(run, got):
    """Same workload as the library runs it BY DEFAULT (round 3): block bounds — conservative workgroup-level frustum and Hi-Z
    rejection, same results — for pools above 262144 slots. Reported beside the headline (`value_with_block_bounds`), never as
    `value`: the headline stays the linear scan SURVEY.md §8d prices (GV_CONFIG_LINEAR_SCAN)."""
    args, n, sc, wl = run.args, run.n, run.sc, run.wl
    vb = GpuVisibility(device=run.local_rank, profile_cull_only=True, block_bounds=n <= 262144)
    vb.bind_transforms(sc.transforms, sc.entity_to_transform)
    vb.bind_pool(0, sc.meshes)
    vb.hierarchy_rebuild()
    if wl["hiz"]:
        vb.hiz_build(run.depth)

    def bounded_step():
        if wl["hiz"]:
            vb.hiz_rebuild()
        if wl["sweep"]:
            vb.sweep({"mfma": GV_SWEEP_MFMA, "valu": GV_SWEEP_VALU}[args.sweep])
        vb.cull(0, [run.view])

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
    same = same_outputs(vb.fetch(0, write_back=False, occupancy=n), got)
    out = dict(ms_per_step=dt / frames * 1e3, value=n * frames / dt, cull_kernel_ms=sb["device_ms"]["cull"] / max(1, tb["cull"]),
               examined_workgroup_fraction=sb["bounds_blocks_examined"] / max(1, sb["bounds_blocks_total"]), outputs_identical_to_headline=same)
    vb.close()
    if not same:
        run.emit({"error": "block-bounds variant differs from the linear scan", "variant": out})
        run.leave(1)
    return out


def valu_chain(run, got):
    """cfg4: the bench default is the MFMA chain BASELINE.json names; the bit-identical v_fma chain is timed beside it."""
    vis, n = run.vis, run.n

    def valu_step():
        vis.sweep(GV_SWEEP_WITH_CULL_VALU)
        vis.cull(0, run.view_array)

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
    same = same_outputs(vis.fetch(0, write_back=False, occupancy=n), got)
    out = dict(kernel="gv::sweep_cull_valu_kernel", ms_per_step=dt / frames * 1e3, value=n * frames / dt,
               avg_launch_ms=sv["device_ms"]["cull"] / max(1, tv["cull"]), outputs_identical=same)
    if not same:
        run.emit({"error": "cfg4: the VALU chain's outputs differ from the MFMA chain's", "variant": out})
        run.leave(1)
    return out


def exchange_variants(run, fx, mark_group, timed_payload, timed_exchange, timed_native):
    """The same frames as bit shards, by the other travel patterns and through torch.distributed — run LAST, under a watchdog
    (bench.py): the travel patterns other than the headline's first meet real links inside this function, and a collective that
    never returns must not take the measured line with it. Returns (mask_variant, mode_variants, torch_variant)."""
    args, vis, n, world = run.args, run.vis, run.n, run.world

    def timed_variant(describe):
        """args.steps frames of step() as currently configured, checked against the exact all-gatherv; a failing variant is
        reported, it does not take the headline with it."""
        problem, out = None, None
        try:
            for _ in range(3):
                run.step()
            e3, _, last3 = run.timed_steps(run.step, args.steps, mark_group)
            problem = fx.check_padded(last3)
            e3 = run.max_over_ranks(e3)
            out = dict(ms_per_step=e3 / args.steps * 1e3, value=n * world * args.steps / e3,
                       shard_bytes_per_rank=[4 * w for w in fx.shard_words_per_rank(fx.ex, last3)],
                       checked_against_exact_allgatherv=problem is None, **describe)
        except Exception as e:  # noqa: BLE001
            problem = f"{type(e).__name__}: {e}"
        if not run.all_agree(problem is None):
            out = dict(error=problem or "failed on another rank", **describe)
        return out

    mask_variant = mode_variants = torch_variant = None
    if timed_payload == "indices" and not args.no_mask_variant:
        fx.ex = fx.make_exchange("mask")
        mask_variant = timed_variant(dict(
            delivers="every rank holds every rank's [count, one bit per mirror entry]; the entry -> pool slot tables travelled "
                     "once at set-up (a consumer that wants the index list expands the rows)",
            exchange_path="c-abi (gv_exchange_masks)" if fx.native else "torch.distributed"))
        args.payload, fx.ex = timed_payload, timed_exchange
    if timed_native and timed_payload == "indices" and world > 1 and not args.no_mode_variants:
        # the same frames with the rows travelling by the other patterns (A/B for the fully connected xGMI node) ...
        mode_variants = {}
        for mode in ("allgather", "p2p", "broadcast"):
            if mode == fx.mode:
                continue
            vis.exchange_set_mode(EXCHANGE_MODES[mode])
            mode_variants[mode] = timed_variant(dict(exchange_path="c-abi (gv_exchange_visible)"))
        vis.exchange_set_mode(EXCHANGE_MODES[fx.mode])
    if timed_native and timed_payload == "indices" and not args.no_torch_variant:
        # ... and through torch.distributed over this script's own buffers (what round 3 timed as the headline)
        fx.native = False
        fx.ex = fx.make_exchange("indices")
        torch_variant = timed_variant(dict(exchange_path="torch.distributed (garden_amd/multi.py::VisibleListExchange)",
                                           capacity_words=fx.ex.capacity))
        fx.ex.drain()
        fx.native, fx.ex = True, timed_exchange
    return mask_variant, mode_variants, torch_variant
