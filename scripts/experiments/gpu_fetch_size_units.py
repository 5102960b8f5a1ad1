#!/usr/bin/env python3
"""What FETCH_SIZE / WRITE_SIZE count for nontemporal and for plain accesses: one launch each of the streaming probe over a known
number of bytes (256 MiB read or written once), to be run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate
passes).  The launches are told apart by their workgroup count: 65,536 = nontemporal read, 65,537+7 (=65,544) = plain read,
65,552 = nontemporal write, 65,560 = plain write."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cpprobotics_amd as crx  # noqa: E402,F401
from cpprobotics_amd.experimental import hbm_stream  # noqa: E402

nbytes = 256 << 20
src = torch.ones(nbytes // 4, dtype=torch.float32, device="cuda")
dst = torch.zeros_like(src)
torch.cuda.synchronize()
for mode, wgs in ((1, 65536), (17, 65544), (2, 65552), (18, 65560)):
    hbm_stream(mode, dst, src, workgroups=wgs)
    torch.cuda.synchronize()
print("bytes per launch", nbytes)
# round 6: narrow accesses.  Told apart by workgroup count: 32,768 = 4 B per lane, 32,776 = 8 B per lane (each reads the 256 MiB once);
# 1,024 / 4,096 workgroups of the private-memory probe = 262,144 / 1,048,576 lanes x 4 KiB written and read back, 2 passes each
from cpprobotics_amd.experimental import fetch_units  # noqa: E402
out = torch.zeros(4096 * 256, dtype=torch.float64, device="cuda")
for mode, wgs in ((0, 32768), (1, 32776)):
    fetch_units(mode, src, out, wgs)
    torch.cuda.synchronize()
for wgs in (1024, 4096):
    fetch_units(2, None, out, wgs, passes=2)
    torch.cuda.synchronize()
    print("private probe:", wgs * 256, "lanes x 2 passes x 4096 B =", wgs * 256 * 2 * 4096, "bytes written, as many read back")
