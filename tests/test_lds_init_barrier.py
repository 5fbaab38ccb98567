"""Rule from DESIGN.md section 3 (the second r02 finding): an LDS array that the WHOLE workgroup fills cooperatively
(`for (int e = tid; ...; e += THREADS) arr[...] = ...`) must not be touched again before a `__syncthreads()`.
k_head_stream's group tables were filled by all waves and first read by the producer waves in front of their first
barrier: wrong uprelu3 for one strip in ~500, seen only by the 256-ship float64 comparison.  This test keeps the rule
mechanical for the policy kernels: after every cooperative fill loop, the next `__syncthreads()` must come before the
next mention of the filled array (a textual check on the sources: conservative, cheap, runs without a GPU)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ("ofx_policy.hip", "ofx_head.hip", "ofx_raster.hip", "ofx_nn.hip", "ofx_fit.hip")

# `for (int e = tid; e < N; e += THREADS) <body>` where <body> stores into arr[...]: group(1) = body
FILL = re.compile(r"for\s*\(\s*int\s+(\w+)\s*=\s*(?:tid|threadIdx\.x|gt)\s*;[^;]*;\s*\1\s*\+=\s*[A-Za-z_0-9:.]+\s*\)\s*(\{(?:[^{}]|\{[^{}]*\})*\}|[^;]*;)")
STORE = re.compile(r"(?:reinterpret_cast<[^>]*>\(\s*&?\s*)?\b(\w+)\s*(?:\)\s*)?(?:\[[^\]]*\])+\s*=[^=]")


def _strip_comments(src):
    src = re.sub(r"//[^\n]*", "", src)
    return re.sub(r"/\*.*?\*/", "", src, flags=re.S)


def _shared_arrays(src):
    return set(re.findall(r"__shared__[^;]*?\b(\w+)\s*\[", src))


def violations(src):
    src = _strip_comments(src)
    shared = _shared_arrays(src)
    bad = []
    for m in FILL.finditer(src):
        body = m.group(2)
        names = {n for n in STORE.findall(body) if n in shared}
        # reinterpret_cast<f32x4 *>(arr)[e] = ... : the array name sits inside the cast
        names |= {n for n in re.findall(r"reinterpret_cast<[^>]*>\(\s*&?\s*(\w+)", body) if n in shared}
        rest = src[m.end():]
        sync = rest.find("__syncthreads()")
        if sync < 0:
            sync = len(rest)
        region = rest[:sync]

        def stmt_of(pos_a, pos_b):
            a = max(region.rfind(";", 0, pos_a), region.rfind("{", 0, pos_a), region.rfind("}", 0, pos_a)) + 1
            b = region.find(";", pos_b)
            return region[a:b if b >= 0 else len(region)]

        def is_plain_store(stmt, name):
            # the name sits left of the assignment and not right of it: a store to another part of the array
            eq = re.search(r"(?<![=!<>+\-*/|&^])=(?!=)", stmt)
            return bool(eq and re.search(r"\b%s\b" % re.escape(name), stmt[:eq.start()])
                        and not re.search(r"\b%s\b" % re.escape(name), stmt[eq.end():]))

        for n in names:
            watch = [n]
            for nxt in re.finditer(r"\b%s\b" % re.escape(n), region):
                stmt = stmt_of(nxt.start(), nxt.end())
                alias = re.match(r"\s*(?:const\s+)?[\w:<> ]+?\*\s*(?:const\s+)?(\w+)\s*=\s*&\s*%s\b" % re.escape(n), stmt)
                if alias:                       # T *q = &arr[...]: follow the pointer instead
                    watch.append(alias.group(1))
                    continue
                if is_plain_store(stmt, n):
                    continue
                bad.append((n, stmt.strip().replace("\n", " ")[:120]))
                break
            for al in watch[1:]:
                for nxt in re.finditer(r"\b%s\b" % re.escape(al), region):
                    stmt = stmt_of(nxt.start(), nxt.end())
                    if re.search(r"=\s*&\s*%s\b" % re.escape(n), stmt) or is_plain_store(stmt, al):
                        continue
                    bad.append((n, "through %s: %s" % (al, stmt.strip().replace("\n", " ")[:100])))
                    break
    return bad


def test_rule_catches_the_r02_bug_pattern():
    buggy = """
    __global__ void k(int *p) {
      __shared__ unsigned short tabA[100];
      const int tid = threadIdx.x;
      for (int g = tid; g < 100; g += HS_THREADS) { tabA[g] = (unsigned short)g; }
      if (wv < 4) { int x = tabA[3]; __syncthreads(); }
    }"""
    fixed = buggy.replace("if (wv < 4)", "__syncthreads();\n      if (wv < 4)")
    assert violations(buggy) and not violations(fixed)
    # the form the bug really had: the read goes through a pointer into the table, inside a lambda
    buggy2 = buggy.replace("if (wv < 4) { int x = tabA[3]; __syncthreads(); }",
                           "const unsigned short *const ta = &tabA[n16 >> 2];\n"
                           "      auto setup = [&](int T) { return ta[4 * T]; };\n      int x = setup(0); __syncthreads();")
    assert violations(buggy2)


def test_cooperative_lds_fills_are_followed_by_a_barrier():
    for f in FILES:
        src = open(os.path.join(ROOT, "ofighters_amd", "csrc", f)).read()
        v = violations(src)
        assert not v, "%s: LDS array used after a cooperative fill without a barrier in between: %s" % (f, v)
