#!/usr/bin/env python3
"""Regenerates tests/golden/scene_golden.json + scene_golden.npz: a small scene file in the reference's format (written
by oracle/scene_json_py.py::write_scene, a restatement of TransformSystem::serialize, transform.cpp:459-515, and
JsonSerializer, json-serialize.cpp:249-310) plus hand-added loader edge cases, and the pools the Python restatement of
the loader (resource.cpp:2421-2510, transform.cpp:517-583) builds from it. NOT reference output (the reference
cannot be built here): the fixture pins the ingest and its oracle against regressions."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from garden_amd import scene  # noqa: E402
from oracle import scene_json_py as sj  # noqa: E402

POOLS = {"Model": 0, "Sprite": 3}


def build_text():
    src = scene.hierarchy_scene(400, depth=4, fanout=4)
    src.transforms["position"][::17, :3] = 0
    src.transforms["scale"][::13, :3] = 1
    src.transforms["scale"][5::13, :3] = np.float32(1.75)
    src.transforms["rotation"][::19] = (0, 0, 0, 1)
    src.meshes["aabbMin"][::23, :3] = -0.5
    src.meshes["aabbMax"][::23, :3] = 0.5
    doc = json.loads(sj.write_scene(src.transforms, {"Model": src.meshes[:200], "Sprite": src.meshes[200:]}, src.entity_to_transform))
    u = [sj.encode_uid(0xABC000 + k) for k in range(4)]
    doc["entities"] += [
        {"components": []},
        {"components": [{".type": "Transform", "uid": u[0], "parent": u[1], "position": 3, "scale": 0.5}]},
        {"components": [{".type": "Transform", "uid": u[1], "parent": u[2]}, {".type": "Camera"}]},
        {"components": [{".type": "Transform", "uid": u[2], "isActive": False}]},
        {"components": [{".type": "Sprite", "aabb": {"min": 1.0, "max": 0.5}, "isEnabled": False},
                        {".type": "Transform", "uid": u[3], "parent": u[1]}]},
        {"components": [{".type": "Transform", "uid": u[0], "parent": u[0]}]},
        {"components": [{".type": "Transform", "uid": "bad", "parent": sj.encode_uid(0xDEAD)},
                        {".type": "Model", "aabb": {"max": {"x": 1.0, "y": 1.0, "z": 1.0}}}]},
    ]
    return json.dumps(doc, indent=1)


if __name__ == "__main__":
    text = build_text()
    open(os.path.join(HERE, "scene_golden.json"), "w").write(text)
    tr, meshes, e2t, info = sj.read_scene(text, POOLS)
    np.savez_compressed(os.path.join(HERE, "scene_golden.npz"), transforms=tr, meshes0=meshes[0], meshes3=meshes[3], e2t=e2t,
                        info=np.array([info[k] for k in sorted(info)], np.int64), info_keys=np.array(sorted(info)))
    print(len(text), "bytes of JSON;", tr.shape[0], "transforms,", meshes[0].shape[0], "+", meshes[3].shape[0], "meshes")
