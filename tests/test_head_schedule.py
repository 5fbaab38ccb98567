"""The schedule k_head_stream (ofighters_amd/csrc/ofx_head.hip) hard-codes - producer / consumer lags, ring sizes,
the closed forms of its per-sub-step tables - checked on the CPU (tools/head_schedule.py), and the constants of the
two files compared."""
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sched():
    spec = importlib.util.spec_from_file_location("head_schedule", os.path.join(ROOT, "tools", "head_schedule.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_schedule_holds():
    m = _sched()
    ad, span3, span2 = m.check()
    assert span3 <= m.NR3 and span2 <= m.NR2
    assert ad[0] == 2 and ad[-1] == 50


def test_kernel_constants_match():
    m = _sched()
    src = open(os.path.join(ROOT, "ofighters_amd", "csrc", "ofx_head.hip")).read()

    def const(name):
        return int(re.search(r"constexpr int %s = (\d+)" % name, src).group(1))

    assert const("HS_GPR") == m.GPR and const("HS_NTILES") == m.NTILES and const("HS_NS") == m.NS
    assert int(re.search(r"HS_NR3 = (\d+)", src).group(1)) == m.NR3
    assert int(re.search(r"HS_NR2 = (\d+)", src).group(1)) == m.NR2
    assert "HS_TABP = 16 * HS_GPR" in src and m.TABP == 16 * m.GPR and "(T * 1261) >> 16" in src
    assert const("HS_NV") == 9
    # the consumers' first row and the kernels' closed forms
    assert "int R = r_in - 8;" in src and m.C_ROW0 == -8
    assert "8 * (s + 1) + ((s + 1) >> 3)" in src
    assert "s + 2 + ((s + 2) >> 2)" in src
    assert "(ga * 5042) >> 16" in src or "* 5042) >> 16" in src
