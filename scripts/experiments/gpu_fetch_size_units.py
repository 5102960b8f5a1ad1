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
