"""Golden fixtures (tests/golden/visibility_golden.npz, made by tests/golden/make_golden.py): the oracle must
reproduce them on CPU, and the HIP library must reproduce them on the GPU — bit for bit."""
import os

import numpy as np
import pytest

from garden_amd import scene
from garden_amd.pools import MESH_DTYPE, TRANSFORM_DTYPE

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "visibility_golden.npz")
CASES = [("flat", "flat_main"), ("flat", "flat_cascade"), ("flat", "flat_rel"), ("hier", "hier_main"), ("flat", "flat_hiz")]


@pytest.fixture(scope="module")
def G():
    return np.load(PATH)


def load_scene(G, name):
    meshes = np.ascontiguousarray(G[f"{name}_meshes"]).view(MESH_DTYPE).reshape(-1).copy()
    xf = np.ascontiguousarray(G[f"{name}_transforms"]).view(TRANSFORM_DTYPE).reshape(-1).copy()
    return scene.Scene(meshes, xf, G[f"{name}_e2t"].copy())


def load_view(G, prefix):
    f = G[f"{prefix}_flags"]
    return dict(view_proj=G[f"{prefix}_view_proj"], camera_position=G[f"{prefix}_camera_position"],
                camera_offset=G[f"{prefix}_camera_offset"], shadow_pass=int(f[0]), use_hiz=int(f[1]),
                distance_2d=int(f[2]), emit_records=1)


def check(G, prefix, r, is_visible):
    assert np.array_equal(r["visible_idx"], G[f"{prefix}_visible_idx"])
    assert np.array_equal(r["baked_model"].view(np.uint32), G[f"{prefix}_baked_model"].view(np.uint32))
    assert np.array_equal(r["distance_sq"].view(np.uint32), G[f"{prefix}_distance_sq"].view(np.uint32))
    if is_visible is not None:
        assert np.array_equal(is_visible, G[f"{prefix}_is_visible"])


def test_fixture_is_nontrivial(G):
    for _, prefix in CASES:
        n = G[f"{prefix}_visible_idx"].shape[0]
        assert 0 < n < 1024, (prefix, n)
    assert G["flat_hiz_visible_idx"].shape[0] < G["flat_main_visible_idx"].shape[0]


@pytest.mark.parametrize("name,prefix", CASES)
def test_oracle_reproduces_golden(oracle, G, name, prefix):
    sc = load_scene(G, name)
    v = load_view(G, prefix)
    hz = oracle.Hiz(G["depth"], rule=0) if v["use_hiz"] else None
    r = oracle.prepare_meshes(sc.meshes, sc.transforms, sc.entity_to_transform, v, hiz=hz)
    check(G, prefix, r, sc.meshes["isVisible"])


def test_oracle_world_and_pyramid_golden(oracle, G):
    sc = load_scene(G, "hier")
    w = oracle.world_matrices(sc.transforms, sc.entity_to_transform)
    assert np.array_equal(w.view(np.uint32), G["hier_world"].view(np.uint32))
    for rule in (0, 1):
        hz = oracle.Hiz(G["depth"], rule=rule)
        for k in range(1, hz.mip_count):
            assert np.array_equal(hz.level(k), G[f"hiz_rule{rule}_mip{k}"])


@pytest.mark.gpu
@pytest.mark.parametrize("name,prefix", CASES)
def test_gpu_reproduces_golden(gpu, G, name, prefix):
    sc = load_scene(G, name)
    v = load_view(G, prefix)
    gpu.bind_transforms(sc.transforms, sc.entity_to_transform)
    gpu.bind_pool(0, sc.meshes)
    gpu.hierarchy_rebuild()
    if v["use_hiz"]:
        gpu.hiz_build(G["depth"])
    gpu.cull(0, [v])
    r = gpu.fetch(0, write_back=True, occupancy=sc.count)
    check(G, prefix, r, sc.meshes["isVisible"] if v["shadow_pass"] < 0 else None)


@pytest.mark.gpu
def test_gpu_world_and_pyramid_golden(G):
    from garden_amd.lib import GpuVisibility
    sc = load_scene(G, "hier")
    for rule in (0, 1):
        with GpuVisibility(device=0, hiz_rule=rule) as vis:
            vis.bind_transforms(sc.transforms, sc.entity_to_transform)
            for mode in (0, 1):
                vis.sweep(mode)
                assert np.array_equal(vis.get_world(0, sc.count).view(np.uint32), G["hier_world"].view(np.uint32))
            vis.hiz_build(G["depth"])
            for k in range(1, vis.hiz_mip_count()):
                e = G[f"hiz_rule{rule}_mip{k}"]
                assert np.array_equal(vis.hiz_read_level(k, e.shape[1], e.shape[0]), e)
