"""bench.py's `cpu_baseline` leg: the CPU path of the same frame on the GPU box's host cores — the oracle's AVX2 cull (bit-identical
to the scalar restatement of mesh.cpp:111-184 + transform.hpp:197-214), the scalar hiz.frag pyramid and the scalar world-matrix
sweep, each threaded with the ThreadPool::addItems range split (source/thread-pool.cpp:173-200). TEST INFRASTRUCTURE like the rest
of oracle/: only bench.py's cpu_baseline leg imports it. A reported baseline, not the optimisation target."""
import os
import time

import numpy as np


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
