#!/usr/bin/env python3
"""Parity-risk census (CPU only; oracle/gv_census.cpp; test infrastructure like the oracle it drives): how many entities
of the BASELINE scenes change their visibility decision when the build-defined arithmetic is replaced by the other
operation orders a real cfnptr/math could have — float64, the other 4x4 association, the un-fused source form, GCC's
contraction of it.  python tests/parity_census.py [--small]  ->  profiles/r02_parity_census.json
"Parity unpinned"  becomes a bounded number: the flipped entities all sit within `max_margin` of a
frustum plane (world units, distance of the deciding corner), i.e. exactly on the silhouette of the frustum."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root (this file lives in tests/)
sys.path.insert(0, ROOT)
from garden_amd import scene  # noqa: E402
from oracle import oracle_py  # noqa: E402

VARIANTS = {1: "float64", 2: "other_association_root_first", 3: "unfused_source_form", 4: "gcc_contracted_source_form"}


def census_lib():
    oracle_py.load()
    lib = C.CDLL(os.path.join(ROOT, "oracle", "build", "libgv_census.so"))
    lib.gvo_census.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.gvo_census.restype = None
    return lib


def run(lib, sc, view, hiz, variant, threads):
    e2t = np.ascontiguousarray(sc.entity_to_transform, dtype=np.uint32)
    mp, tp, gv = oracle_py.mesh_pool(sc.meshes), oracle_py.transform_pool(sc.transforms, e2t), oracle_py.to_view(view)
    n = sc.count
    decision, margin = np.zeros(n, np.uint8), np.zeros(n, np.float32)
    lib.gvo_census(C.byref(mp), C.byref(tp), C.byref(gv), C.byref(hiz.c) if hiz is not None else None, variant, threads,
                   decision.ctypes.data, margin.ctypes.data)
    return decision, margin


def census(name, sc, view, depth, threads):
    lib = census_lib()
    hiz = oracle_py.Hiz(depth, threads=threads) if depth is not None else None
    base, margin = run(lib, sc, view, hiz, 0, threads)
    # variant 0 must BE the oracle
    m2 = sc.meshes.copy()
    exp = oracle_py.prepare_meshes(m2, sc.transforms, sc.entity_to_transform, view, hiz=hiz, threads=threads)
    assert np.array_equal((base == 3).astype(np.uint8), m2["isVisible"]) and int((base == 3).sum()) == exp["draw_count"]
    cam = np.asarray(view["camera_position"][:3], dtype=np.float64)
    out = {"entities": sc.count, "candidates_reaching_the_tests": int((base != 0).sum()), "visible": int((base == 3).sum()),
           "frustum_rejected": int((base == 1).sum()), "occluded": int((base == 2).sum()), "variants": {}}
    for v, label in VARIANTS.items():
        if v == 2 and not np.any(sc.transforms["parent"] != 0):
            out["variants"][label] = {"flips": 0, "note": "flat pool: no chain, the association cannot differ"}
            continue
        d, _ = run(lib, sc, view, hiz, v, threads)
        flip = d != base
        assert not np.any(flip & ((d == 0) | (base == 0)))  # the filters involve no arithmetic
        idx = np.nonzero(flip)[0]
        fr = flip & ((d == 1) | (base == 1))  # the frustum decision itself moved
        oc = flip & ~fr                        # same frustum decision, the occlusion answer moved
        rec = {"flips": int(idx.size), "frustum_flips": int(fr.sum()), "occlusion_flips": int(oc.sum()),
               "became_visible": int((flip & (d == 3)).sum()), "became_invisible": int((flip & (base == 3)).sum()),
               "flip_rate_of_candidates": float(idx.size) / max(1, out["candidates_reaching_the_tests"])}
        if fr.any():
            fi = np.nonzero(fr)[0]
            pos = sc.transforms["position"][:, :3]  # flat scenes: mesh slot i pairs with transform slot i (roots); distance is indicative
            dist = np.linalg.norm(pos[np.minimum(fi, pos.shape[0] - 1)].astype(np.float64) - cam, axis=1)
            rec["max_margin_of_a_frustum_flip_world_units"] = float(margin[fi].max())
            rec["median_margin_world_units"] = float(np.median(margin[fi]))
            rec["max_margin_relative_to_camera_distance"] = float((margin[fi] / np.maximum(dist, 1e-9)).max()) if not np.any(sc.transforms["parent"] != 0) else None
        out["variants"][label] = rec
    print(name, json.dumps(out, indent=1), flush=True)
    return out


def band_scene(n, view, seed=99):
    """Adversarial: every entity's DECIDING corner sits within +-delta of a frustum plane, delta / distance log-uniform in
    1e-9 .. 1e-2 — the only place where operation order can matter. Flat pool, camera at the origin."""
    sc = scene.flat_scene(n, seed=seed, defects=False)
    rng = np.random.default_rng(seed)
    planes = oracle_py.frustum(view["view_proj"]).astype(np.float64)
    t, m = sc.transforms, sc.meshes
    # points inside the frustum: rejection-sample directions, radius log-uniform 10 m .. 20 km
    pts = np.zeros((0, 3))
    while pts.shape[0] < n:
        d = rng.normal(size=(4 * n, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        p = d * (10.0 ** rng.uniform(1, 4.3, (4 * n, 1)))
        inside = np.all(p @ planes[:, :3].T + planes[:, 3] > 0, axis=1)
        pts = np.concatenate([pts, p[inside]])
    p = pts[:n]
    k = rng.integers(0, planes.shape[0], n)
    nrm, w = planes[k, :3], planes[k, 3]
    p = p - ((p * nrm).sum(1) + w)[:, None] * nrm  # onto the chosen face
    # support of the oriented box along the normal (float64): the farthest corner sits `support` beyond the centre
    q = t["rotation"].astype(np.float64)
    x, y, z, ww = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - ww * z), 2 * (x * z + ww * y)], 1),
                  np.stack([2 * (x * y + ww * z), 1 - 2 * (x * x + z * z), 2 * (y * z - ww * x)], 1),
                  np.stack([2 * (x * z - ww * y), 2 * (y * z + ww * x), 1 - 2 * (x * x + y * y)], 1)], 1)  # [n, row, col]
    h = m["aabbMax"][:, :3].astype(np.float64) * t["scale"][:, :3].astype(np.float64)
    support = (np.abs(np.einsum("nr,nrc->nc", nrm, R)) * h).sum(1)
    rel = 10.0 ** rng.uniform(-9, -2, n) * rng.choice([-1.0, 1.0], n)
    delta = rel * np.linalg.norm(p, axis=1)
    p = p + nrm * (-support + delta)[:, None]
    t["position"][:, :3] = p.astype(np.float32)
    return sc


def band_census(threads, n):
    view = scene.main_camera_view()
    sc = band_scene(n, view)
    lib = census_lib()
    base, margin = run(lib, sc, view, None, 0, threads)
    dist = np.linalg.norm(sc.transforms["position"][:, :3].astype(np.float64), axis=1)
    rel = margin.astype(np.float64) / np.maximum(dist, 1e-9)
    edges = 10.0 ** np.arange(-10.0, -1.5, 1.0)
    out = {"entities": n, "what": "every entity's deciding corner within +-delta of a frustum plane, delta / distance log-uniform 1e-9..1e-2",
           "bins_margin_over_distance": [f"{a:.0e}..{b:.0e}" for a, b in zip(edges[:-1], edges[1:])], "variants": {}}
    which = np.digitize(rel, edges) - 1
    counts = np.bincount(np.clip(which, 0, len(edges) - 2), minlength=len(edges) - 1)
    out["entities_per_bin"] = counts.tolist()
    for v, label in VARIANTS.items():
        if v == 2:
            continue
        d, _ = run(lib, sc, view, None, v, threads)
        flip = d != base
        per = np.bincount(np.clip(which[flip], 0, len(edges) - 2), minlength=len(edges) - 1)
        out["variants"][label] = {"flips": int(flip.sum()), "flips_per_bin": per.tolist(),
                                  "max_margin_over_distance_of_a_flip": float(rel[flip].max()) if flip.any() else None,
                                  "max_margin_world_units_of_a_flip": float(margin[flip].max()) if flip.any() else None}
    print("band", json.dumps(out, indent=1), flush=True)
    return out


def main():
    small = "--small" in sys.argv
    threads = os.cpu_count() or 1
    k = 20 if small else 1
    res = {"_what": __doc__.strip().split("\n")[0],
           "_decision": "0 filtered, 1 frustum-rejected, 2 occluded, 3 visible; a flip = the variant's decision differs from the "
                        "canonical (= oracle = GPU) one; margin = |largest corner distance| of the plane nearest to deciding, canonical arithmetic"}
    res["cfg2_1M_flat_frustum"] = census("cfg2", scene.flat_scene(1_000_000 // k), scene.main_camera_view(), None, threads)
    res["cfg3_10M_flat_frustum_hiz4096"] = census("cfg3", scene.flat_scene(10_000_000 // k), scene.main_camera_view(use_hiz=1),
                                                   scene.synthetic_depth(4096 // (4 if small else 1), 4096 // (4 if small else 1)), threads)
    res["cfg4_10M_depth4_frustum"] = census("cfg4", scene.hierarchy_scene(10_000_000 // k), scene.main_camera_view(), None, threads)
    res["band_1M_entities_on_the_frustum_planes"] = band_census(threads, 1_000_000 // k)
    if not small:
        json.dump(res, open(os.path.join(ROOT, "profiles", "r02_parity_census.json"), "w"), indent=1)
    return res


if __name__ == "__main__":
    main()
