#!/usr/bin/env python
"""Checks on the EMITTED gfx950 code of libcrx.so, run by __graft_entry__.build() on the build box (no GPU needed):

  1. DPP read-after-VALU-write: on the gfx9 family a VALU instruction with a DPP operand must not read a VGPR that a VALU instruction
     wrote less than two wait states earlier.  The compiler's hazard recogniser inserts the s_nop for code it schedules itself; the
     hand-issued blocks of csrc/dare_math.h (dq_max_perm, dq_quad_test4: non-volatile inline assembly with hand-counted `s_nop 1`)
     carry their own.  A compiler bump that reorders or splits those blocks would otherwise show up only as wrong bits on the GPU box
      — here it fails the build.
  3. the accumulator-register block of the MPC tile kernels (csrc/mpc_tile_kernels.hip.h, mpc_agpr.inc): a40 .. a255 hold the solver's
     feedback gains between inline-asm statements the compiler cannot see into.  Its register allocator may use any accumulator
     register for values of its own between two of those statements, so the block is safe only while the allocator stays below it:
     inside every tile kernel, a register of the block may appear only in complete, slot-aligned, ascending runs of twelve
     v_accvgpr_write_b32 / v_accvgpr_read_b32 (the accessors), in no other instruction and no other pattern.
  2. the exec-masked add of dq_add_lane2: its `s_and_saveexec_b64 sX, <lanes 2 mod 4>` must be followed by exactly its four v_add_f32
     and the restoring `s_mov_b64 exec, sX`, nothing in between.

usage: python scripts/check_isa.py [path/to/libcrx.so ...]    (default: both libraries of cpprobotics_amd/)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def disassemble(lib):
    sys.path.insert(0, ROOT)
    from cpprobotics_amd._lib import disassemble_code_object
    return disassemble_code_object(lib)


def vregs(op):
    """VGPR numbers an operand names: v12 -> {12}, v[4:7] -> {4..7}; anything else -> empty."""
    m = re.fullmatch(r"v(\d+)", op)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def parse(text):
    """-> list of (function, [(mnemonic, operands, raw line)])"""
    funcs, cur = [], None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = (m.group(1), [])
            funcs.append(cur)
            continue
        if cur is None or not line.startswith("\t"):
            continue
        body = line.split("//")[0].strip()
        if not body:
            continue
        mn, _, rest = body.partition(" ")
        ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", rest.split(" quad_perm")[0].split(" row_")[0])] if rest else []
        cur[1].append((mn, ops, body))
    return funcs


def agpr_block_params():
    """(base, per_slot, slots) of the generated accessor file"""
    txt = open(os.path.join(ROOT, "cpprobotics_amd", "csrc", "mpc_agpr.inc")).read()
    return tuple(int(re.search(rf"#define {k} (\d+)", txt).group(1)) for k in ("CRX_MPC_AGPR_BASE", "CRX_MPC_AGPR_PER_SLOT", "CRX_MPC_AGPR_SLOTS"))


def check_agpr_block(name, ins, base, per, slots):
    """-> (problems, number of accessor runs, highest accumulator register the compiler itself uses)"""
    problems, seq, top = [], [], -1
    for mn, ops, raw in ins:
        regs = [int(m) for m in re.findall(r"\ba(\d+)\b", raw)]
        for m in re.finditer(r"\ba\[(\d+):(\d+)\]", raw):
            regs += list(range(int(m.group(1)), int(m.group(2)) + 1))
        for r in regs:
            if r < base:
                top = max(top, r)
            elif mn in ("v_accvgpr_write_b32", "v_accvgpr_read_b32") and len(regs) == 1:
                seq.append((mn, r, raw))
            else:
                problems.append(f"{name}: `{raw}` touches a{r}, inside the gain block a{base} .. a{base + per * slots - 1}")
    k = 0
    while k < len(seq):
        mn, r, raw = seq[k]
        run = seq[k:k + per]
        if (r - base) % per or len(run) < per or any(x[0] != mn or x[1] != r + j for j, x in enumerate(run)):
            problems.append(f"{name}: the gain block is touched outside a complete slot-aligned accessor run, at `{raw}` — the compiler's "
                            "own accumulator-register use has reached the block (raise CRX_MPC_AGPR_BASE in scripts/gen_mpc_agpr.py, or "
                            "lower the kernel's register pressure)")
            break
        k += per
    return problems, len(seq) // per, top


def check(lib, text=None):
    funcs = parse(text if text is not None else disassemble(lib))
    problems, n_dpp, n_exec_blocks, n_nop_dpp = [], 0, 0, 0
    n_tile, n_runs, top_agpr = 0, 0, -1
    base, per, slots = agpr_block_params()
    for name, ins in funcs:
        if "mpc_tile_" in name:
            pr, runs, top = check_agpr_block(name, ins, base, per, slots)
            problems += pr
            n_tile += 1; n_runs += runs; top_agpr = max(top_agpr, top)
    for name, ins in funcs:
        for i, (mn, ops, raw) in enumerate(ins):
            if mn.endswith("_dpp") and len(ops) >= 2:
                n_dpp += 1
                src = vregs(ops[1])                       # src0 is the operand the DPP control permutes
                waits, j = 0, i - 1
                if j >= 0 and ins[j][0] == "s_nop":
                    n_nop_dpp += 1
                while j >= 0 and waits < 2:
                    pmn, pops, praw = ins[j]
                    if pmn == "s_nop":
                        waits += int(pops[0], 0) + 1 if pops else 1
                    else:
                        if pmn.startswith("v_") and pops and (vregs(pops[0]) & src) and not pmn.startswith("v_cmp"):
                            problems.append(f"{name}: `{raw}` reads {ops[1]} {waits} wait state(s) after `{praw}`")
                            break
                        waits += 1
                    j -= 1
            # the hand-issued block: a saveexec that goes straight into v_add_f32 (the compiler's own divergent regions continue with
            # s_cbranch_execz / s_xor and end in s_or_b64 exec)
            if mn == "s_and_saveexec_b64" and len(ops) == 2 and i + 1 < len(ins) and ins[i + 1][0].startswith("v_add_f32"):
                nxt = [x[0] for x in ins[i + 1:i + 6]]
                want = ["v_add_f32_e32"] * 4 + ["s_mov_b64"]
                tail = ins[i + 5] if i + 5 < len(ins) else ("", [], "")
                if nxt == want and tail[1][:2] == ["exec", ops[0]]:
                    n_exec_blocks += 1
                else:
                    problems.append(f"{name}: the exec-masked add after `{raw}` is no longer four v_add_f32 + `s_mov_b64 exec, {ops[0]}`: {nxt}")
    return problems, {"functions": len(funcs), "dpp_instructions": n_dpp, "of_them_behind_an_s_nop": n_nop_dpp, "exec_masked_add_blocks": n_exec_blocks,
                      "mpc_tile_kernels": n_tile, "gain_block_accessor_runs": n_runs, "highest_compiler_agpr_in_tile_kernels": top_agpr,
                      "gain_block_base": base}


def main():
    libs = sys.argv[1:] or [os.path.join(ROOT, "cpprobotics_amd", n) for n in ("libcrx.so", "libcrx_x.so")]
    bad = 0
    for lib in libs:
        try:
            problems, stats = check(lib)
        except FileNotFoundError as e:
            # a box without the ROCm LLVM tools / objcopy can still build and run the library; the check is for the build box
            print(f"check_isa: SKIPPED for {os.path.relpath(lib, ROOT)} — the code object cannot be disassembled here: {e}")
            continue
        print(f"check_isa: {os.path.relpath(lib, ROOT)}: {stats}, {len(problems)} problem(s)")
        for p in problems[:20]:
            print("  " + p)
        bad += len(problems)
        if stats["mpc_tile_kernels"] == 0 or stats["gain_block_accessor_runs"] == 0:
            print("  no MPC tile kernel (or no accessor run) found in the code object: the gain-block check checks nothing")
            bad += 1
        if stats["exec_masked_add_blocks"] == 0 or stats["of_them_behind_an_s_nop"] == 0:
            print("  the hand-issued blocks of csrc/dare_math.h were not found in the code object (renamed? inlined away?): the check checks nothing")
            bad += 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
