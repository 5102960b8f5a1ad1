#!/usr/bin/env python3
"""Does the order in which workgroups meet the eight XCDs matter to a streaming kernel?  crx_x_hbm_stream_dev with the default order
(workgroup b on XCD b % 8: every XCD touches every eighth 4-KiB piece) against mode + 8 (every XCD one contiguous eighth of the
buffer), copy / read / write / update in place, grid-stride and one-shot grids.  JSON lines."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cpprobotics_amd as crx  # noqa: E402,F401
from cpprobotics_amd.experimental import hbm_stream  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


nbytes = 704 << 20
src = torch.ones(nbytes // 4, dtype=torch.float32, device="cuda")
dst = torch.zeros_like(src)
for mode, name, moved in ((0, "copy", 2), (1, "read", 1), (2, "write", 1), (3, "update_in_place", 2)):
    for wgs in (2048, 8192, nbytes // 4096):
        a = timeit(lambda: hbm_stream(mode, dst, src, workgroups=wgs))
        b = timeit(lambda: hbm_stream(mode + 8, dst, src, workgroups=wgs))
        print(json.dumps({"mode": name, "workgroups": wgs, "TB_per_s_default_order": moved * nbytes / a / 1e9, "TB_per_s_xcd_contiguous": moved * nbytes / b / 1e9}), flush=True)
# write-only once more: nontemporal against plain stores, both orders
for wgs in (1024, 2048, 8192, nbytes // 4096):
    row = {"mode": "write", "workgroups": wgs}
    for label, m in (("nt_default", 2), ("nt_xcd", 10), ("plain_default", 18), ("plain_xcd", 26)):
        row["TB_per_s_" + label] = nbytes / timeit(lambda: hbm_stream(m, dst, src, workgroups=wgs)) / 1e9
    print(json.dumps(row), flush=True)
