#!/usr/bin/env python
"""Schedule of k_head_stream (ofighters_amd/csrc/ofx_head.hip): checks the producer / consumer lags and the ring sizes
the kernel hard-codes, and prints the stage-A table it embeds.  Run on the CPU; tests/test_head_schedule.py calls
check().

One workgroup streams a 100-column half of one ship's 200x200 uprelu3 plane top to bottom in NS sub-steps:
  * the B waves produce uprelu3 in 16-quad M-tiles, flat over (quad row, group of 4 quads), 13 groups per quad row:
    8 tiles per sub-step and a ninth in every eighth one (325 tiles in 40 sub-steps);
  * the C waves consume 5 uprelu3 rows per sub-step, rows 5 s - 8 .. 5 s - 4 in sub-step s (r04: the ring holds the nine
    tap planes of those rows instead of their eight channels - same rows, same lags);
  * stage A (uprelu2 row pairs from uprelu1) runs one sub-step ahead of the tiles that read them.
"""
GPR = 13                    # groups of 4 quads per quad row (52 quad slots: 50 + halo + pad)
QROWS = 100
NTILES = QROWS * GPR // 4   # 325
NS = 42                     # sub-steps
NR3 = 16                    # ring rows of uprelu3 - since r04 of its nine tap planes V_t = sum_c w4[t][c] uprelu3_c
NR2 = 16                    # uprelu2 ring rows
TABP = 208                  # entries of the group table (r04)
C_ROW0 = -8                 # first uprelu3 row of the C block of sub-step 0 (5 s + C_ROW0)


def NT(s):
    """tiles finished by the end of sub-step s"""
    if s < 0:
        return 0
    return min(NTILES, 8 * (s + 1) + ((s + 1) >> 3))


def tile_qrows(t):
    return (4 * t) // GPR, (4 * t + 3) // GPR


def sub_qrows(s):
    if NT(s) <= NT(s - 1):
        return None
    return tile_qrows(NT(s - 1))[0], tile_qrows(NT(s) - 1)[1]


def stage_a_table():
    """AD[s + 1] = uprelu2 row pairs finished by the end of sub-step s (AD[0]: by the prologue): everything the tiles
    of sub-step s + 1 read (uprelu2 rows q - 1 .. q + 1 for their quad rows q)."""
    ad = []
    for s in range(-1, NS):
        qr = None
        for t in range(s + 1, -1, -1):
            qr = sub_qrows(t)
            if qr:
                break
        ad.append(min(50, min(99, qr[1] + 1) // 2 + 1) if qr else 0)
    return ad


def check():
    ad = stage_a_table()
    assert NT(NS - 1) == NTILES and 4 * NTILES == QROWS * GPR
    span3 = span2 = 0
    for s in range(NS):
        done_q = 4 * NT(s - 1) // GPR            # quad rows complete before sub-step s
        rows = [r for r in range(5 * s + C_ROW0, 5 * s + C_ROW0 + 5) if 0 <= r < 200]
        qr = sub_qrows(s)
        if rows:
            need = min(199, max(rows) + 1)      # last uprelu3 row the C block reads
            assert need < 2 * done_q, (s, need, done_q)
        if rows and qr:
            lo = min(rows) - 1
            hi = 200 if qr[1] == QROWS - 1 else 2 * qr[1] + 1
            span3 = max(span3, hi - lo + 1)
        if qr:
            # uprelu2 rows read by the tiles of this sub-step are there; rows written now do not alias them
            assert 2 * ad[s] - 1 >= min(99, qr[1] + 1), (s, ad[s], qr)
            lo2 = qr[0] - 1
            hi2 = 100 if ad[s + 1] >= 50 else 2 * ad[s + 1] - 1
            span2 = max(span2, max(hi2, qr[1] + 1) - lo2 + 1)
        # at most one stage-A tile per B wave and sub-step: tile (pair p, half h) goes to wave (2 p + h) & 3
        assert ad[s + 1] - ad[s] <= 2
    assert 5 * (NS - 1) + C_ROW0 <= 199 <= 5 * (NS - 1) + C_ROW0 + 4
    assert span3 <= NR3 and span2 <= NR2, (span3, span2)
    # the kernel's closed form of the stage-A table
    assert ad == [min(50, s + 2 + ((s + 2) >> 2)) for s in range(NS + 1)], ad
    # group index / 13 by multiply-shift, as the kernel does it
    for g in range(4 * NTILES + 4):
        assert (g * 5042) >> 16 == g // GPR, g
    # r04: ONE group table of TABP = 16 quad rows x 13 groups entries - the (uprelu2 ring, tap-plane ring) offsets of a
    # group depend on its quad row modulo 16 only - addressed by 4 (T mod 52) + lane group, T mod 52 by multiply-shift
    assert TABP == 16 * GPR == 4 * 52
    for t in range(NTILES):
        assert t - 52 * ((t * 1261) >> 16) == t % 52, t
        for lg in range(4):
            g = 4 * t + lg
            q, gg = divmod(g, GPR)
            e = 4 * (t % 52) + lg
            qe, gge = divmod(e, GPR)
            assert gge == gg and (qe & 15) == (q & 15) and ((2 * qe) & (NR3 - 1)) == ((2 * q) & (NR3 - 1)), (t, lg)
    # LDS of a workgroup: 9 tap planes + the uprelu2 ring + the table: two workgroups per CU (160 KB)
    lds = 4 * (9 * (NR3 * 106 + 4) + 4 * (NR2 + 2) * 56 + TABP)
    assert lds <= 80 * 1024, lds
    return ad, span3, span2


if __name__ == "__main__":
    ad, s3, s2 = check()
    print("live uprelu3 rows <= %d (ring %d), uprelu2 rows <= %d (ring %d)" % (s3, NR3, s2, NR2))
    print("constexpr int kStageADone[%d] = {%s};" % (len(ad), ", ".join(map(str, ad))))
