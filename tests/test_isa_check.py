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
