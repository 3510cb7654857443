"""The build's ISA rule (optimal_quad_control_rl_amd/isa_lint.py): no packed-f32 instruction whose second source feeds its high dword
to the low result half -- the form MI355X computes wrongly in lanes 48-63 next to another wave's matrix instructions (root cause of the
round-4 "two waves per SIMD" corruption; reproducer tools/ubench/mfma_pk_hazard.hip, evidence profiles/r05_root_cause.txt)."""
import os

import pytest

from optimal_quad_control_rl_amd import build, isa_lint

BAD = [
    "\tv_pk_fma_f32 v[4:5], v[6:7], v[140:141], v[4:5] op_sel:[0,1,0] op_sel_hi:[1,0,1]",
    "\tv_pk_fma_f32 v[4:5], v[6:7], s[0:1], v[4:5] op_sel:[0,1,0]",
    "\tv_pk_mul_f32 v[38:39], v[20:21], v[4:5] op_sel:[0,1] neg_lo:[0,1]",
    "\tv_pk_add_f32 v[28:29], v[2:3], v[178:179] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]",
    "\tv_pk_add_f32 v[16:17], v[16:17], v[16:17] op_sel:[0,1] op_sel_hi:[1,0] ; a comment",
]
GOOD = [
    "\tv_pk_fma_f32 v[4:5], v[6:7], v[140:141], v[4:5]",
    "\tv_pk_fma_f32 v[4:5], v[6:7], v[140:141], v[4:5] op_sel:[1,0,0] op_sel_hi:[0,1,1]",
    "\tv_pk_fma_f32 v[4:5], v[6:7], v[140:141], v[4:5] op_sel:[0,0,1] op_sel_hi:[1,1,0]",
    "\tv_pk_fma_f32 v[12:13], v[8:9], s[36:37], v[10:11] op_sel_hi:[1,0,0]",
    "\tv_pk_mul_f32 v[6:7], v[138:139], s[0:1] op_sel:[1,0]",
    "\tv_pk_mov_b32 v[6:7], v[64:65], v[58:59] op_sel:[0,1]",
    "\tv_fma_f32 v34, v112, v214, 0",
    "\tv_mfma_f32_32x32x16_f16 v[0:15], v[16:19], v[198:201], 0",
]


def test_rule_recognises_exactly_the_hazardous_form():
    assert all(isa_lint.is_hazardous(l) for l in BAD)
    assert not any(isa_lint.is_hazardous(l) for l in GOOD)


def test_rewrite_exchanges_the_commutative_sources_with_their_modifiers():
    f = isa_lint.fix_asm_line
    assert f(BAD[0]) == "\tv_pk_fma_f32 v[4:5], v[140:141], v[6:7], v[4:5] op_sel:[1,0,0] op_sel_hi:[0,1,1]"
    assert f(BAD[1]) == "\tv_pk_fma_f32 v[4:5], s[0:1], v[6:7], v[4:5] op_sel:[1,0,0]"
    assert f(BAD[2]) == "\tv_pk_mul_f32 v[38:39], v[4:5], v[20:21] op_sel:[1,0] neg_lo:[1,0]"
    assert f(BAD[3]) == "\tv_pk_add_f32 v[28:29], v[178:179], v[2:3] op_sel:[1,0] neg_lo:[1,0] neg_hi:[1,0]"
    assert f(BAD[4]).startswith("\tv_pk_add_f32 v[16:17], v[16:17], v[16:17] op_sel:[1,0] op_sel_hi:[0,1]") and "a comment" in f(BAD[4])
    for l in GOOD:
        assert f(l) == l
    text, n = isa_lint.fix_asm_text("\n".join(BAD + GOOD))
    assert n == len(BAD) and not any(isa_lint.is_hazardous(l) for l in text.split("\n"))
    # (both sources high: split into scalar halves since round 6, test_both_sources_high_is_split_into_scalar_halves)
    assert f("\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,1]").split("\n") == ["\tv_mul_f32_e64 v0, v3, v5", "\tv_mul_f32_e64 v1, v3, v5"]


def test_rewrite_keeps_constants_scalar_registers_and_other_modifiers_in_place():
    f = isa_lint.fix_asm_line
    # inline constant / SGPR pair as the source that moves to position 1; clamp and a trailing comment survive
    assert f("\tv_pk_mul_f32 v[2:3], 2.0, v[4:5] op_sel:[0,1] clamp") == "\tv_pk_mul_f32 v[2:3], v[4:5], 2.0 op_sel:[1,0] clamp"
    assert f("\tv_pk_fma_f32 v[0:1], s[2:3], v[8:9], -0.5 op_sel:[0,1,0] op_sel_hi:[1,1,0]") == \
        "\tv_pk_fma_f32 v[0:1], v[8:9], s[2:3], -0.5 op_sel:[1,0,0] op_sel_hi:[1,1,0]"
    # the third source's bits never move
    assert f("\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,1] op_sel_hi:[1,0,0] neg_lo:[0,1,1] neg_hi:[1,0,1]") == \
        "\tv_pk_fma_f32 v[0:1], v[4:5], v[2:3], v[6:7] op_sel:[1,0,1] op_sel_hi:[0,1,0] neg_lo:[1,0,1] neg_hi:[0,1,1]"
    # disassembler output (address / encoding comment) is recognised by the lint as well
    assert isa_lint.is_hazardous("\tv_pk_add_f32 v[4:5], v[0:1], v[2:3] op_sel:[0,1] // 000000001234: D38F0004 18020500")
    assert not isa_lint.is_hazardous("\tv_pk_add_f32 v[4:5], v[0:1], v[2:3] op_sel:[1,0] // 000000001234: D38F0004 18020500")
    # packed f16 instructions select halves of ONE dword: measured unaffected, not touched
    assert f("\tv_pk_fma_f16 v0, v1, v2, v0 op_sel:[0,1,0]") == "\tv_pk_fma_f16 v0, v1, v2, v0 op_sel:[0,1,0]"


def test_rule_agrees_with_the_measured_matrix():
    """The lint's rule is not an opinion: every victim form of the reproducer (tools/ubench/mfma_pk_hazard.hip) that FAILED on the
    MI355X (profiles/r05_mfma_pk_hazard.txt, first section: against f16 matrix instructions on the SIMD's other wave) is flagged, and
    every form that never failed is not."""
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "tools", "ubench", "mfma_pk_hazard.hip")).read()
    forms = {name: txt for txt, name in re.findall(r'X\(\d+,\s*"([^"]+)",\s*"([^"]+)"\)', src)}
    rows, in_first = {}, False
    for ln in open(os.path.join(root, "profiles", "r05_mfma_pk_hazard.txt")):
        if ln.startswith("## victim instruction forms"):
            in_first = True
        elif ln.startswith("##"):
            in_first = False
        elif in_first and ln.count("|") == 2:
            rows[ln.split("|")[0].strip()] = int(ln.split("|")[2].split()[0])
    assert len(rows) >= 20 and set(rows) <= set(forms), set(rows) - set(forms)
    regs = {"%0": "v[0:1]", "%1": "v[2:3]", "%2": "v[4:5]", "%3": "v[6:7]"}
    for name, failures in rows.items():
        line = "\t" + forms[name]
        for k, v in regs.items():
            line = line.replace(k, v)
        assert isa_lint.is_hazardous(line) == (failures > 0), (name, failures, line)
    assert sum(1 for f in rows.values() if f > 0) >= 6


def test_store_data_rule():
    """Rule 2: a VALU write of the data registers of a > 8-byte vector-memory store needs two wait states (round 5: an inline-asm
    global_store_dwordx4 followed at once by the next tile's v_pk_mul_f32 corrupted the f32 partial gradients)."""
    S = "\tglobal_store_dwordx4 v[164:165], v[172:175], off sc1"
    hz = lambda body: isa_lint.store_data_hazards((S + "\n" + body).split("\n"))
    assert len(hz("\tv_pk_mul_f32 v[172:173], v[176:177], v[54:55]")) == 1              # the bug as it was compiled
    assert len(hz("\ts_nop 0\n\tv_mov_b32_e32 v175, 0")) == 1                          # one wait state is not enough
    assert hz("\ts_nop 1\n\tv_pk_mul_f32 v[172:173], v[176:177], v[54:55]") == []      # the fix
    assert hz("\ts_mov_b32 s0, 0\n\ts_mov_b32 s1, 0\n\tv_mov_b32_e32 v172, 0") == []  # two instructions in between
    assert hz("\tv_mov_b32_e32 v171, 0\n\tv_mov_b32_e32 v176, 0") == []               # neighbours of the data registers
    assert hz("\tv_mov_b32_e32 v164, 0") == []                                         # the ADDRESS registers are read at issue
    assert hz("\tv_cmp_lt_i32_e32 vcc, 8, v172") == []                                 # reads, does not write
    two = "\tglobal_store_dwordx2 v[164:165], v[172:173], off\n\tv_mov_b32_e32 v172, 0"
    assert isa_lint.store_data_hazards(two.split("\n")) == []                          # 8 bytes: no hazard
    buf = "\tbuffer_store_dwordx4 v[1:4], v5, s[0:3], 0 offen\n\tv_add_u32_e32 v5, 1, v5\n\tv_add_u32_e32 v4, 1, v5"
    assert [w for _, w in isa_lint.store_data_hazards(buf.split("\n"))] == ["v_add_u32_e32 v4, 1, v5"]   # buffer form: data first
    dis = "\tglobal_store_dwordx4 v[2:3], v[4:7], off   // 000000001230: DC7C0000 007F0402\n\tv_mov_b32_e32 v4, 0   // 000000001238: 7E080280"
    assert len(isa_lint.store_data_hazards(dis.split("\n"))) == 1                      # disassembler lines


def test_the_built_library_contains_no_hazardous_instruction():
    """Disassembles every code object of libquadrace.so (what the GPU will run, not what the compiler was asked for)."""
    lib = build.build_native_locked()
    assert os.path.exists(lib)
    assert isa_lint.lint_library(lib) == []


def test_both_sources_high_is_split_into_scalar_halves():
    """op_sel:[1,1,*] (hi * hi for the low half; the vectoriser can produce it) cannot be fixed by exchanging the sources.
    It is split into the two scalar instructions that compute the same IEEE results, in an order that clobbers nothing."""
    from optimal_quad_control_rl_amd import isa_lint as L

    out = L.fix_asm_line("\tv_pk_mul_f32 v[38:39], v[20:21], v[4:5] op_sel:[1,1] op_sel_hi:[0,1]")
    assert out.split("\n") == ["\tv_mul_f32_e64 v38, v21, v5", "\tv_mul_f32_e64 v39, v20, v5"]
    # destination low half is a source of the high half: high half first
    out = L.fix_asm_line("\tv_pk_fma_f32 v[2:3], v[8:9], v[4:5], v[2:3] op_sel:[1,1,0] op_sel_hi:[0,0,0] neg_hi:[0,0,1] clamp")
    assert out.split("\n") == ["\tv_fma_f32 v3, v8, v4, -v2 clamp", "\tv_fma_f32 v2, v9, v5, v2 clamp"]
    # scalar-register and literal operands are broadcast
    out = L.fix_asm_line("\tv_pk_add_f32 v[10:11], v[0:1], s[4:5] op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]")
    assert out.split("\n") == ["\tv_add_f32_e64 v10, -v1, s5", "\tv_add_f32_e64 v11, v1, s4"]
    assert not any(L.is_hazardous(ln) for ln in out.split("\n"))
    # both orders clobber a source: refused, with the kernel and the line named by fix_asm_text
    with pytest.raises(ValueError, match=r"kernel my_kernel, assembly line 3"):
        L.fix_asm_text("my_kernel:\n\ts_nop 0\n\tv_pk_fma_f32 v[2:3], v[2:3], v[4:5], v[6:7] op_sel:[1,1,0] op_sel_hi:[0,1,1]\n")


def test_lint_fails_closed_on_a_file_without_device_code(tmp_path):
    """a library in which no AMDGPU code object can be found must not be reported clean."""
    from optimal_quad_control_rl_amd import isa_lint as L

    p = tmp_path / "empty.so"
    p.write_bytes(b"\x7fELF" + bytes(200))
    with pytest.raises(RuntimeError, match="nothing to lint"):
        L.lint_library(str(p))
    from optimal_quad_control_rl_amd import build
    stats = {}
    assert L.lint_library(build.build_native_locked(), stats) == []
    assert stats["code_objects"] >= 5 and stats["symbols"] > 100 and stats["packed_f32"] > 1000
