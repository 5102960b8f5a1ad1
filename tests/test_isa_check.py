"""scripts/check_isa.py — the checks __graft_entry__.build() runs on the emitted gfx950 code (DPP read-after-VALU-write wait states,
the hand-issued exec-masked add of csrc/dare_math.h): they pass on the built libraries and they do catch what they are there to catch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_built_libraries_pass(crx):
    import check_isa
    from cpprobotics_amd import experimental as X
    for lib in (crx.lib_path(), X.ab_lib_path()):
        problems, stats = check_isa.check(lib)
        assert not problems, problems[:5]
        assert stats["dpp_instructions"] > 500 and stats["exec_masked_add_blocks"] >= 8 and stats["of_them_behind_an_s_nop"] >= 50, stats


def test_checker_catches_a_missing_wait_state_and_a_split_block():
    import check_isa
    ok = """
0000000000001000 <k>:
\tv_max_f32_e32 v20, v1, v2                                  // 000000001000: 00000000
\ts_nop 1                                                    // 000000001004: 00000000
\tv_max_f32_dpp v20, v20, v20 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf// 000000001008: 00000000
\ts_and_saveexec_b64 s[28:29], s[26:27]                      // 00000000100c: 00000000
\tv_add_f32_e32 v6, v6, v48                                  // 000000001010: 00000000
\tv_add_f32_e32 v50, v50, v7                                 // 000000001014: 00000000
\tv_add_f32_e32 v51, v51, v46                                // 000000001018: 00000000
\tv_add_f32_e32 v8, v8, v9                                   // 00000000101c: 00000000
\ts_mov_b64 exec, s[28:29]                                   // 000000001020: 00000000
"""
    problems, stats = check_isa.check(None, text=ok)
    assert not problems and stats["dpp_instructions"] == 1 and stats["exec_masked_add_blocks"] == 1
    # the s_nop gone: the DPP reads v20 one wait state after its write
    problems, _ = check_isa.check(None, text=ok.replace("\ts_nop 1                                                    // 000000001004: 00000000\n", ""))
    assert len(problems) == 1 and "reads v20" in problems[0]
    # one independent instruction in between is still one wait state short
    problems, _ = check_isa.check(None, text=ok.replace("s_nop 1 ", "s_mov_b32 s0, 0 "))
    assert len(problems) == 1
    # the scheduler moved something into the exec-masked block
    problems, _ = check_isa.check(None, text=ok.replace("\tv_add_f32_e32 v51, v51, v46", "\tv_mov_b32_e32 v9, s19                                     // x\n\tv_add_f32_e32 v51, v51, v46"))
    assert len(problems) == 1 and "exec-masked add" in problems[0]


def test_gain_block_check_catches_a_compiler_write_into_the_block():
    """Round 6: the MPC tile kernels keep their feedback gains in accumulator registers a40 .. a255 between inline-asm statements
    (csrc/mpc_agpr.inc).  The compiler may use ANY accumulator register for a value of its own between two of them; the build
    check must accept the accessor runs and refuse everything else that touches the block."""
    import check_isa
    base, per, slots = check_isa.agpr_block_params()
    assert per == 12 and base + per * slots == 256 and base >= 40

    def run(op, first, dst=True):
        return "".join(f"\t{op} a{first + k}, v{10 + k}\n" if dst else f"\t{op} v{10 + k}, a{first + k}\n" for k in range(per))
    head = "0000000000001000 <_ZN3crx15mpc_tile_kernelILi24EEEvv>:\n"
    ok = head + "\tv_accvgpr_write_b32 a3, v1\n" + run("v_accvgpr_write_b32", base + per * 2) + f"\tv_accvgpr_read_b32 v5, a{base - 1}\n" + \
        run("v_accvgpr_read_b32", base, dst=False)
    problems, stats = check_isa.check(None, text=ok)
    assert not problems and stats["mpc_tile_kernels"] == 1 and stats["gain_block_accessor_runs"] == 2 and stats["highest_compiler_agpr_in_tile_kernels"] == base - 1
    # a stray compiler spill into the block
    problems, _ = check_isa.check(None, text=ok + f"\tv_accvgpr_write_b32 a{base + 1}, v7\n")
    assert problems and "gain block" in problems[0]
    # a run cut short (the scheduler moved one accessor instruction away)
    cut = head + "".join(run("v_accvgpr_write_b32", base).splitlines(keepends=True)[:-1])
    problems, _ = check_isa.check(None, text=cut)
    assert problems
    # a misaligned run, and a non-accessor instruction naming a block register
    problems, _ = check_isa.check(None, text=head + run("v_accvgpr_write_b32", base + 1))
    assert problems
    problems, _ = check_isa.check(None, text=head + "\tv_accvgpr_mov_b32 a100, a2\n")
    assert problems
    # the same instructions in a kernel that is not a tile kernel are none of this check's business
    problems, _ = check_isa.check(None, text="0000000000001000 <_ZN3crx10mpc_kernelILi24ELb1EEEvv>:\n\tv_accvgpr_write_b32 a" + str(base + 1) + ", v7\n")
    assert not problems


def test_generated_accessor_file_is_current():
    """csrc/mpc_agpr.inc is generated (register names must be literals in the asm text): the committed file is what the generator prints."""
    import gen_mpc_agpr
    assert open(os.path.join(ROOT, "cpprobotics_amd", "csrc", "mpc_agpr.inc")).read() == gen_mpc_agpr.text()
