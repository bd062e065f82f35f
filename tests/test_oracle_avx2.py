"""The AVX2 CPU path (oracle/gv_oracle_avx2.c: 8 entities per iteration, the timed cpu_baseline) must agree with
the scalar oracle bit for bit — same visible set, isVisible bytes, bakedModel and distanceSq — on flat,
hierarchical, shuffled and Hi-Z scenes, single- and multi-threaded."""
import numpy as np
import pytest

from garden_amd import scene


def both(oracle, sc, view, hiz=None, threads=1):
    m1, m2 = sc.meshes.copy(), sc.meshes.copy()
    exp = oracle.prepare_meshes(m1, sc.transforms, sc.entity_to_transform, view, hiz=hiz)
    a = oracle.Avx2Scene(m2, sc.transforms, sc.entity_to_transform)
    got = a.prepare_meshes(view, hiz=hiz, threads=threads)
    order = np.argsort(got["visible_idx"], kind="stable")
    assert got["draw_count"] == exp["draw_count"]
    assert np.array_equal(got["visible_idx"][order], exp["visible_idx"])
    assert np.array_equal(got["baked_model"][order].view(np.uint32), exp["baked_model"].view(np.uint32))
    assert np.array_equal(got["distance_sq"][order].view(np.uint32), exp["distance_sq"].view(np.uint32))
    assert np.array_equal(m2["isVisible"], m1["isVisible"])
    a.close()
    return exp["draw_count"]


@pytest.mark.parametrize("n", [1, 7, 8, 9, 1000, 50_003])
def test_flat(oracle, n):
    both(oracle, scene.flat_scene(n, seed=n), scene.main_camera_view())


def test_hierarchy_shuffled_threads_and_shadow(oracle):
    sc = scene.shuffled_scene(scene.hierarchy_scene(30_000, depth=5, fanout=6), fraction=0.5, drop_transforms=0.02)
    assert both(oracle, sc, scene.main_camera_view(), threads=1) > 0
    both(oracle, sc, scene.main_camera_view(camera_position=(10, 20, 30)), threads=5)
    both(oracle, sc, scene.cascade_view(), threads=3)
    both(oracle, sc, dict(scene.main_camera_view(), distance_2d=1), threads=2)


def test_hiz(oracle):
    sc = scene.flat_scene(30_000, seed=5)
    hz = oracle.Hiz(scene.synthetic_depth(256, 128))
    both(oracle, sc, scene.main_camera_view(use_hiz=1), hiz=hz, threads=4)


def test_soa_built_by_several_threads_is_the_same_soa(oracle):
    """bench.py's cpu_baseline lets the culling threads first-touch their own slices of the SoA arrays."""
    from garden_amd import scene
    sc = scene.hierarchy_scene(30_011, depth=4, fanout=7)
    view = scene.main_camera_view()
    a = oracle.Avx2Scene(sc.meshes.copy(), sc.transforms, sc.entity_to_transform)
    exp = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in a.prepare_meshes(view).items()}
    for threads in (2, 5, 64):
        b = oracle.Avx2Scene(sc.meshes.copy(), sc.transforms, sc.entity_to_transform, threads=threads)
        got = b.prepare_meshes(view)
        assert got["draw_count"] == exp["draw_count"] and np.array_equal(got["visible_idx"], exp["visible_idx"])
        assert np.array_equal(got["baked_model"].view(np.uint32), exp["baked_model"].view(np.uint32))
        b.close()
    a.close()


@pytest.mark.parametrize("size,rule,rg16f", [((256, 128), 0, False), ((1000, 597), 0, False), ((1000, 597), 1, False), ((333, 77), 1, True),
                                             ((1, 1), 0, False), ((4096, 8), 0, False)])
def test_eight_wide_hiz_query_equals_the_scalar_one(oracle, size, rule, rg16f):
    """hiz_occluded8 (round 3: the baseline's occlusion queries 8 at a time) against gvo_hiz_occluded through the two loops: cameras
    inside and outside the world (boxes that cross the camera plane), non-finite transforms, pyramids of every shape and rule,
    depth images with +-inf, -0 and NaN texels."""
    from garden_amd import scene
    rng = np.random.default_rng(size[0] * 7 + rule)
    depth = scene.synthetic_depth(*size)
    h, w = depth.shape
    for val in (np.inf, -np.inf, np.nan, np.float32(-0.0)):
        depth[rng.integers(0, h, 40), rng.integers(0, w, 40)] = val
    hz = oracle.Hiz(depth, rule=rule, rg16f=rg16f)
    sc = scene.hierarchy_scene(40_003, depth=3, fanout=9)
    bad = np.arange(50, sc.count, 3001)
    sc.transforms["position"][bad[0::3], 0] = np.nan
    sc.transforms["position"][bad[1::3], 2] = np.inf
    sc.transforms["scale"][bad[2::3], 1] = -np.inf
    sc.transforms["scale"][1234:1300, :3] = np.float32(400.0)  # boxes that cover the screen: the coarsest levels
    side = 100.0 * sc.count ** (1.0 / 3.0)
    for k in range(6):
        pos = tuple(float(x) for x in rng.normal(0, side * (0.05 if k % 2 else 0.7), 3).astype(np.float32))
        view = scene.main_camera_view(seed=int(rng.integers(1 << 30)), camera_position=pos, use_hiz=1)
        both(oracle, sc, view, hiz=hz, threads=1 + k % 3)
