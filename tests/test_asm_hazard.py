"""Rule from DESIGN.md section 3: NEVER feed an MFMA result to inline assembly.  The compiler's hazard recogniser does
not look inside an asm statement, so nothing guarantees the wait states between a matrix instruction and an asm read
of its result (r01: 9 % of a heat map wrong under one scheduling variant).  This test keeps the rule mechanical: every
asm statement of the policy kernels must be one of the forms known to be safe - no vector inputs at all, or vector
inputs that are ordinary VALU results - and anything new has to be added here with its reason."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# regex of the statement -> why it is safe
ALLOWED = {
    r's_mov_b32 %0, 0x7f800000" : "=s"\(pinf\)': "scalar constant: no vector operand",
    r'v_max3_f32 %0, %1, %2, 0" : "=v"\(m\) : "v"\(acc\[0\]': "k_conv1_lut: operands are table sums (VALU adds), no MFMA in that kernel",
    r'v_max3_f32 %0, %1, %2, %3" : "=v"\(m\) : "v"\(acc\[2\]': "k_conv1_lut: as above",
    r's_mov_b32 %0, 0" : "=s"\(zoff\)': "scalar zero the compiler cannot see through (keeps LDS reads inside a loop): no vector operand",
}


def _asm_statements(path):
    src = open(path).read()
    return [m.group(0) for m in re.finditer(r"\basm\s*(volatile)?\s*\((?:[^;]|\n)*?\);", src)]


def test_no_unknown_inline_asm_in_mfma_kernels():
    for f in ("ofx_policy.hip", "ofx_head.hip"):
        for st in _asm_statements(os.path.join(ROOT, "ofighters_amd", "csrc", f)):
            assert any(re.search(pat, st) for pat in ALLOWED), "unreviewed inline asm in %s: %s" % (f, st)


def test_mfma_results_go_through_compiler_visible_ops():
    """The ReLU / max behind an MFMA is __builtin_amdgcn_fmed3f (max_raw / hd_max_raw), never an asm v_max."""
    for f in ("ofx_policy.hip", "ofx_head.hip"):
        src = open(os.path.join(ROOT, "ofighters_amd", "csrc", f)).read()
        assert "__builtin_amdgcn_fmed3f" in src
        assert not re.search(r'asm[^;]*v_(max|med3)_f32[^;]*"v"\(d', src), f
