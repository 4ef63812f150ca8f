"""Measurement build only (tools/build_variant.sh trace -DPR_MLP_TRACE): what every workgroup of the grouped evaluation launch of a
native frame did - when it entered / left each object's tile loop, how many tiles it took.
    PR_PERF_LIB=build/variants/libplayrender_trace.so python tools/perf/mlp_trace.py [tennis|minecraft]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from playableenvironments_amd import _lib  # noqa: E402
_lib.library_path = lambda: os.path.abspath(os.environ["PR_PERF_LIB"])
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import perf_native_frame as pn  # noqa: E402

world = sys.argv[1] if len(sys.argv) > 1 else "tennis"
dev = torch.device("cuda", 0)
cfg, model, scene, size = pn.build(world, dev)
lib = _lib.load()
with torch.no_grad():
    for _ in range(3):
        model.forward_from_scene_encoding(*bench.scene_args(scene, size), 0, False, 1200, patch_stride=[4, 8])
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (1024 * 12))()
    lib.pr_debug_mlp_trace(buf)          # (zero the tile counters' baseline)
    base = [buf[i] for i in range(1024 * 12)]
    model.forward_from_scene_encoding(*bench.scene_args(scene, size), 0, False, 1200, patch_stride=[4, 8])
    torch.cuda.synchronize()
    lib.pr_debug_mlp_trace(buf)
rows = [[buf[w * 12 + i] for i in range(12)] for w in range(512)]
t0 = min(r[0] for r in rows)
us = lambda t: (t - t0) / 100.0
print("wg   start   end0   end1   end2   end3  tiles4 tilesN")
order = sorted(range(512), key=lambda w: rows[w][4])
for w in order[::16] + order[-8:]:
    r = rows[w]
    print(f"{w:4d} {us(r[0]):7.1f} {us(r[1]):7.1f} {us(r[2]):7.1f} {us(r[3]):7.1f} {us(r[4]):7.1f} {r[5] - base[w * 12 + 5]:4d} {r[6] - base[w * 12 + 6]:4d}")
ends = sorted(us(r[4]) for r in rows)
print("kernel span", ends[-1], "us; median WG end", ends[256], "; WGs ending before 50% of span:", sum(e < ends[-1] / 2 for e in ends))
tiles = [(r[5] - base[w * 12 + 5]) + (r[6] - base[w * 12 + 6]) for w, r in enumerate(rows)]
print("tiles per WG histogram:", {k: tiles.count(k) for k in sorted(set(tiles))}, "total", sum(tiles))
