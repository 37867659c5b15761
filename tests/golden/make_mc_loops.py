#!/usr/bin/env python3
"""Derives tests/golden/mc_reference_loops.json from the reference's marching-cubes data file (MCData.txt, read by
MarchingCubeGrid.py:80-94; runs only where /root/reference is mounted): per cube case the edge-table word and the
iso-polygon LOOPS of the published triangulation -- the directed boundary edges of its triangle fan(s), i.e. which
crossed edges are joined on which face and with which orientation.  The loops fix the mesh's topology (including the
choice on ambiguous faces); which diagonals triangulate a polygon is the only freedom left."""
import json
import os
from collections import Counter

REF = "/root/reference/MCData.txt"
HERE = os.path.dirname(os.path.abspath(__file__))


def parse(path=REF):
    edge, tri = [0] * 256, [[-1] * 16 for _ in range(256)]
    for li, line in enumerate(open(path)):
        if li < 32:
            v = line.split(',', 8)
            for i in range(len(v) - 1):
                edge[li * 8 + i] = int(v[i], 16)
        else:
            v = line.split(',', 16)
            for i in range(len(v) - 1):
                tri[li - 32][i] = int(v[i])
    return edge, tri


def loops(row):
    row = [int(x) for x in row if int(x) >= 0]
    c = Counter()
    for k in range(0, len(row), 3):
        a, b, d = row[k:k + 3]
        for e in ((a, b), (b, d), (d, a)):
            c[e] += 1
    return sorted([list(e) for e in c if (e[1], e[0]) not in c])


if __name__ == "__main__":
    edge, tri = parse()
    out = {"source": "derived from the reference's MCData.txt by tests/golden/make_mc_loops.py",
           "edge_table": edge, "n_triangles": [sum(1 for x in r if x >= 0) // 3 for r in tri], "loops": [loops(r) for r in tri]}
    json.dump(out, open(os.path.join(HERE, "mc_reference_loops.json"), "w"), separators=(",", ":"))
    print("wrote", len(out["loops"]), "cases")
