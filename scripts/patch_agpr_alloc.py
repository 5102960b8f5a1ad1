#!/usr/bin/env python
"""Adds the LLVM function attribute "amdgpu-agpr-alloc"="B,B" to every kernel of an LLVM IR file (text), B = CRX_MPC_AGPR_BASE of
csrc/mpc_agpr.inc: the register allocator may then use the accumulator registers a0 .. a(B-1) only, and the block a(B) .. a255 that
the MPC tile kernels address by literal register names (mpc_agpr.inc) is theirs alone.  clang has no source spelling for the attribute.

usage: python scripts/patch_agpr_alloc.py in.ll out.ll      (csrc/Makefile)"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(src, dst):
    base = int(re.search(r"#define CRX_MPC_AGPR_BASE (\d+)", open(os.path.join(ROOT, "cpprobotics_amd", "csrc", "mpc_agpr.inc")).read()).group(1))
    s = open(src).read()
    groups = set(re.findall(r"^define [^\n]*amdgpu_kernel void @\S*mpc_tile\S*\([^\n]*\) (?:local_unnamed_addr )?#(\d+)", s, re.M))
    others = set(re.findall(r"^define [^\n]*amdgpu_kernel void @(?!\S*mpc_tile)[^\n]*\) (?:local_unnamed_addr )?#(\d+)", s, re.M))
    if groups & others:
        sys.exit("patch_agpr_alloc: a tile kernel shares its attribute group with another kernel")
    if not groups:
        sys.exit("patch_agpr_alloc: no amdgpu_kernel definitions in " + src)
    for g in groups:
        m = re.search(r"^attributes #%s = \{ (.*)\}$" % g, s, re.M)
        if not m:
            sys.exit("patch_agpr_alloc: attribute group #%s not found" % g)
        if "amdgpu-agpr-alloc" in m.group(1):
            sys.exit("patch_agpr_alloc: the compiler already set amdgpu-agpr-alloc: " + m.group(0)[:200])
        s = s[:m.start()] + 'attributes #%s = { "amdgpu-agpr-alloc"="%d,%d" %s}' % (g, base, base, m.group(1)) + s[m.end():]
    open(dst, "w").write(s)
    print("patch_agpr_alloc: %d kernel attribute group(s) fenced at a%d" % (len(groups), base))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
