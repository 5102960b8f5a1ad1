#!/usr/bin/env python
"""Static instruction mix of the loops of a kernel in libcrx.so's gfx950 code objects (no GPU needed): every backward branch closes a
loop; for each loop of at least `min_len` instructions the count by mnemonic.  The MPC solver's backward stage and rollout stage are
the two big inner loops.     usage: python scripts/isa_loop_mix.py KERNEL_SUBSTRING [min_len=250]"""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cpprobotics_amd._lib import disassemble_code_object, lib_path  # noqa: E402


def loops(text, name, min_len):
    cur, ins = None, []
    for line in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
        if m:
            cur = m.group(2)
            continue
        if cur and name in cur and line.startswith("\t"):
            a = re.search(r"//\s*([0-9A-Fa-f]+):", line)
            if a:
                ins.append((int(a.group(1), 16), line.split("//")[0].strip(), line))
    idx = {a: i for i, (a, _, _) in enumerate(ins)}
    out = []
    for i, (a, b, l) in enumerate(ins):
        if b.startswith(("s_cbranch", "s_branch")):
            m = re.search(r"\+0x([0-9a-f]+)>", l)
            if m:
                tgt = ins[0][0] + int(m.group(1), 16)
                if tgt <= a and tgt in idx and i - idx[tgt] + 1 >= min_len:
                    out.append((idx[tgt], i))
    return ins, sorted(set(out), key=lambda x: x[1] - x[0])


if __name__ == "__main__":
    name = sys.argv[1]
    min_len = int(sys.argv[2]) if len(sys.argv) > 2 else 250
    ins, ls = loops(disassemble_code_object(lib_path()), name, min_len)
    print(f"{name}: {len(ins)} instructions, {len(ls)} loops of >= {min_len}")
    for s, e in ls:
        c = collections.Counter(re.sub(r"_e(32|64)$|_dpp$|_sdwa$", "", b.split(" ")[0]) for _, b, _ in ins[s:e + 1])
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        salu = sum(v for k, v in c.items() if k.startswith("s_"))
        mem = sum(v for k, v in c.items() if k.startswith(("scratch_", "global_", "ds_", "buffer_", "flat_")))
        print(f"  loop [{s}, {e}]: {e - s + 1} instructions (VALU {valu}, SALU {salu}, memory {mem}): " + ", ".join(f"{k} {v}" for k, v in c.most_common(16)))
