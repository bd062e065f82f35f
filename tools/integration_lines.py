#!/usr/bin/env python3
"""INTEGRATION.md §2a names lines of garden_amd/csrc/host/gpu_visibility_system.hpp (`:N` in the table's last column; tests/test_abi.py
checks that they still do what the row says). After an edit of the shim: python tools/integration_lines.py  — every token is set to
the line its anchor text stands on now."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "garden_amd", "csrc", "host", "gpu_visibility_system.hpp")
DOC = os.path.join(ROOT, "INTEGRATION.md")
# the shim-line tokens of §2a in order of appearance (reference-side tokens like `:814` are left alone): anchor text -> its first line
ANCHORS = [
    "transDrawIndex = uiDrawIndex = 0;", "unsortedBufferCount = sortedBufferCount = 0;", "hasAnyRefr = hasAnyOIT = hasAnyTD = false;",
    "unsortedBufferCount = sortedBufferCount = 0;", "sp.bufferIndex = sortedSeen++;", "getMeshComponentPool();        // mesh.cpp:410",
    "sp.views.push_back(makeView(uiViewProj", "auto reset = [](MeshBuffer* buffer", "meshSystem->isDrawReady(-1)",
    "meshSystem->isDrawReady(shadowPasses[s].index(s))", "continue;  // no pass draws this system", "sp.views.push_back(makeView(uiViewProj",
    "check(gv_cull(ctx, p,", "buffer->combinedMeshes.resize(occupancy)", "hasAnyRefr |=", "gv_pool_sort(ctx, p, v", "mergeRuns(transSortedMeshes, transRuns);",
    "if (shadowIndex != bufferIndex)",
]
REFERENCE_SIDE = {"`:814`", "`:899-902`", "`:393-396`", "`:39-43`", "`:917-923`", "`:404`"}

shim = open(SHIM).read().splitlines()
line_of = {a: next(i + 1 for i, text in enumerate(shim) if a in text) for a in set(ANCHORS)}
doc = open(DOC).read()
start, end = doc.index("### 2a."), doc.index("## 3. Contract details")
section, k = doc[start:end], [0]


def fix(m):
    if m.group(0) in REFERENCE_SIDE:
        return m.group(0)
    anchor = ANCHORS[k[0]]
    k[0] += 1
    return f"`:{line_of[anchor]}`"


section = re.sub(r"`:[0-9-]+`", fix, section)
assert k[0] == len(ANCHORS), (k[0], len(ANCHORS))
open(DOC, "w").write(doc[:start] + section + doc[end:])
print("INTEGRATION.md §2a:", {a[:32]: n for a, n in line_of.items()})
