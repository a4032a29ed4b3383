"""Generates rankpo_amd/csrc/attention_dq64w_gen.inc: the hand-placed instruction streams of fa_bwd_dq64w_kernel, the head_dim-64 dQ
kernel of the attention backward at one wave per SIMD (round 5; same construction as gen_fwd128w_body.py, whose docstring explains
the issue-cost model the placement follows).

A wave owns 64 queries (query tiles n = 0..3 of 16) of one head and walks 32-key tiles (key sub-tiles m = 0, 1 of 16).  Per key tile t:
  A(t)  S^T[m][n]  = K rows x Q^T[n]                       2 k-steps of 32 (hd 64): 16 MFMAs, accumulators in VGPRs (generations A / B)
  B(t)  dP'^T[m][n] = -delta[n] + V rows x dO^T[n]          16 MFMAs, accumulators in VGPRs (ONE generation), chain input = the row constant
  C(t)  p = exp2(c S - lse[n] log2e);  dS = p dP' (the factor `scale` is applied once, in the epilogue); packed to bf16 fragments
  D(t)  dQ^T[c][n] += K^T[c] x dS^T[n]                      4 hd tiles: 16 MFMAs, accumulators a[0:63]; K^T by transposed LDS reads
Software pipeline inside the ONE in-order stream, two statements per tile:
  X1(t): A(t + 1) with, in its gaps, the rest of C(t) (exponentials of query tiles 2, 3; all products p dP'; packing), the K^T reads
         of tile t and the V row reads of tile t + 1;
  X2(t): D(t) and B(t + 1) with, in their gaps, the first part of C(t + 1) (all exponents e = c S - lse log2e in place, exponentials of
         query tiles 0, 1), the K row reads of tile t + 2 and the wave's two LDS-DMA pieces (K and V of tile t + 4).
K tile j is read by rows in X2(j - 2) and transposed in X1(j): a ring of EIGHT 4-KiB K images (slot j % 8) lets the DMA of tile t + 4,
issued in X2(t), overwrite the slot of tile t - 4; V tile j is read once, in X1(j - 1): ring of four (slot j % 4).  The statements
come in EIGHT variants S0..S7 (t % 8: both ring slots and the generation parity are functions of it): every LDS address is a
loop-invariant per-lane register + an immediate.

Register map (literal; hipcc keeps v[0:63]):
  v[64:95]   SA[m][n] at 64 + 16 m + 4 n       v[96:127] SB[m][n]                  (score generations)
  v[128:159] DP[m][n] at 128 + 16 m + 4 n      v[160:175] KT[c] at 160 + 4 c       K^T fragments (keys 0-15 in + 0, + 1; 16-31 in + 2, + 3)
  v[176:191] DS[n] at 176 + 4 n                v[192:195] LQ[n] = lse log2e        v[196:211] DLQ[n] at 196 + 4 n: four copies of -delta
  a[0:63]    dQ^T[c][n] at 4 (4 c + n)         a[64:95] Q^T[n][ks] at 64 + 8 n + 4 ks       a[96:127] dO^T[n][ks] at 96 + 8 n + 4 ks
  a[128:143] AK[ks][m] at 128 + 4 (2 ks + m)   a[144:159] AV[ks][m] at 144 + 4 (2 ks + m)   (K / V row fragments of the tile whose chains run next)
Operands: X1: %0..%3 transposed-read addresses of K^T hd tiles 0..3 in K slot 0 (keys 16-31 at + 2048, slot s at + 4096 s), %4, %5 V row
          addresses of k-steps 0, 1 in V slot 0 (key sub-tile 1 at + 2048, slot s at + 4096 s).
          X2: %0, %1 K row addresses of k-steps 0, 1 in K slot 0, %2 = c (SGPR); X2D: + %3, %4 per-lane byte offsets of the wave's K / V
          piece inside a tile, %5, %6 the two tiles' global bases (SGPR pairs), %7, %8 the LDS addresses of the two pieces (SGPR).
          MASK: %0..%3 = D[n] = min(query, len - 1) - k0 - 4 g: key 16 m + r is masked iff 16 m + r > D[n].
"""
import os

NO_VALU = os.environ.get("GEN_NO_VALU") == "1"       # timing experiment only (results are wrong)
RATE1 = int(os.environ.get("GEN_RATE1", "8"))        # issue cycles of vector instructions per MFMA gap in X1
RATE2 = int(os.environ.get("GEN_RATE2", "8"))        # ... in X2
X2_EXP = int(os.environ.get("GEN_X2EXP", "2"))       # query tiles (of four) whose exponentials stand in X2 of the iteration before
ROW, IMG, HALF = 128, 4096, 2048
SA = lambda m, n: 64 + 16 * m + 4 * n
SB = lambda m, n: 96 + 16 * m + 4 * n
DP = lambda m, n: 128 + 16 * m + 4 * n
KT = lambda c: 160 + 4 * c
DS = lambda n: 176 + 4 * n
LQ = lambda n: 192 + n
DLQ = lambda n: 196 + 4 * n
T0 = 212
OA = lambda c, n: 4 * (4 * c + n)
QA = lambda n, ks: 64 + 8 * n + 4 * ks
DA = lambda n, ks: 96 + 8 * n + 4 * ks
AK = lambda ks, m: 128 + 4 * (2 * ks + m)
AV = lambda ks, m: 144 + 4 * (2 * ks + m)
v4 = lambda r: "v[%d:%d]" % (r, r + 3)
a4 = lambda r: "a[%d:%d]" % (r, r + 3)
mf = "v_mfma_f32_16x16x32_bf16 "
NEG = "0xf149f2ca"       # -1e30f
ALL_V = ", ".join('"v%d"' % i for i in range(64, 216))
ALL_A = ", ".join('"a%d"' % i for i in range(0, 160))
PAD = ["s_nop 15", "s_nop 7"]                        # an MFMA's result read by a vector instruction right behind it
EL = [(m, r) for m in range(2) for r in range(4)]


def chains_a(S):
    return [mf + "%s, %s, %s, %s" % (v4(S(m, n)), a4(AK(ks, m)), a4(QA(n, ks)), "0" if ks == 0 else v4(S(m, n)))
            for ks in range(2) for m in range(2) for n in range(4)]


def chains_b():
    return [mf + "%s, %s, %s, %s" % (v4(DP(m, n)), a4(AV(ks, m)), a4(DA(n, ks)), v4(DLQ(n)) if ks == 0 else v4(DP(m, n)))
            for ks in range(2) for m in range(2) for n in range(4)]


def products_d():
    return [mf + "%s, %s, %s, %s" % (a4(OA(c, n)), v4(KT(c)), v4(DS(n)), a4(OA(c, n))) for n in range(4) for c in range(4)]


def c_first(S, c_op):
    """First part of C on generation S (in X2 of the iteration before): every exponent in place, the exponentials of query tiles
    0 .. X2_EXP - 1."""
    ops = ["v_fma_f32 v%d, v%d, %s, -v%d" % (S(m, n) + r, S(m, n) + r, c_op, LQ(n)) for n in range(4) for m, r in EL]
    ops += ["v_exp_f32 v%d, v%d" % (S(m, n) + r, S(m, n) + r) for n in range(X2_EXP) for m, r in EL]
    return ops


def c_rest(S):
    """Rest of C on generation S: the other exponentials, dS = p dP', packing; query-tile major (dS^T[n] is the B operand of the
    products 4 n .. 4 n + 3 of X2)."""
    ops = []
    for n in range(4):
        if n >= X2_EXP:
            ops += ["v_exp_f32 v%d, v%d" % (S(m, n) + r, S(m, n) + r) for m, r in EL]
            ops.append("s_nop 0")                                 # transcendental result -> the VALU instruction behind it
        ops += ["v_mul_f32 v%d, v%d, v%d" % (S(m, n) + r, S(m, n) + r, DP(m, n) + r) for m, r in EL]
        ops += ["v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (DS(n) + 2 * m + h, S(m, n) + 2 * h, S(m, n) + 2 * h + 1)
                for m in range(2) for h in range(2)]
    return ops


def off(o):
    return " offset:%d" % o if o else ""


def kt_reads(first_op, slot):
    out = []
    for c in range(4):
        out += ["ds_read_b64_tr_b16 v[%d:%d], %%%d%s" % (KT(c), KT(c) + 1, first_op + c, off(IMG * slot)),
                "ds_read_b64_tr_b16 v[%d:%d], %%%d%s" % (KT(c) + 2, KT(c) + 3, first_op + c, off(IMG * slot + HALF))]
    return out


def row_reads(F, first_op, slot):
    return ["ds_read_b128 %s, %%%d%s" % (a4(F(ks, m)), first_op + ks, off(IMG * slot + HALF * m)) for ks in range(2) for m in range(2)]


def cost(ins):
    return 8 if ins.startswith("v_exp_f32") else 4


def weave(mfmas, fillers, budget, lds=(), first_gap=0, bare=()):
    """MFMAs in order; after MFMA k vector instructions whose issue costs sum to at most `budget` cycles (from gap `first_gap` on,
    none in the gaps `bare`) and one LDS read; what is left is appended."""
    out, fi, li = [], 0, 0
    fillers, lds = list(fillers), list(lds)
    for k, ins in enumerate(mfmas):
        out.append(ins)
        if li < len(lds):
            out.append(lds[li]); li += 1
        took = 0
        while k not in bare and k >= first_gap and fi < len(fillers) and took + cost(fillers[fi]) <= budget:
            took += cost(fillers[fi])
            out.append(fillers[fi]); fi += 1
    return out + lds[li:] + fillers[fi:]


def x1(S_cur, S_next, t, chains=True):
    rest = [] if NO_VALU else c_rest(S_cur)
    lds = kt_reads(0, t % 8) + (row_reads(AV, 4, (t + 1) % 4) if chains else [])
    out = weave(chains_a(S_next), rest, RATE1, lds) if chains else lds + rest
    return out + ["s_waitcnt lgkmcnt(0)"]


def dma_ops(first_op):
    vk, vv, sk, sv, mk, mv = range(first_op, first_op + 6)
    return [["s_mov_b32 m0, %%%d" % mk, "s_nop 0", "global_load_lds_dwordx4 %%%d, %%%d" % (vk, sk)],
            ["s_mov_b32 m0, %%%d" % mv, "s_nop 0", "global_load_lds_dwordx4 %%%d, %%%d" % (vv, sv)]]


DMA_AT = (20, 26)


def x2(S_next, t, nxt=True, dma=False):
    """D(t) (+ B(t + 1), the first part of C(t + 1) and the K rows of tile t + 2 when a next tile exists)."""
    if not nxt:
        return products_d() + PAD                                 # (+ pad: the epilogue reads dQ^T)
    first = [] if NO_VALU else c_first(S_next, "%2")
    out = weave(products_d() + chains_b(), first, RATE2, row_reads(AK, 0, (t + 2) % 8), first_gap=2,
                bare=[k - 1 for k in DMA_AT] if dma else ())
    if dma:
        pieces = dma_ops(3)
        for gi, at in reversed(list(enumerate(DMA_AT))):
            k = [i for i, x in enumerate(out) if x.startswith("v_mfma")][at - 1]
            out[k + 1:k + 1] = pieces[gi]
    return out + ["s_waitcnt lgkmcnt(0)"]


def mask(S):
    out = PAD + ["v_mov_b32 v%d, %s" % (T0, NEG)]
    for n in range(4):
        for m, r in EL:
            out += ["v_cmp_gt_i32 vcc, %d, %%%d" % (16 * m + r, n), "v_cndmask_b32 v%d, v%d, v%d, vcc" % (S(m, n) + r, S(m, n) + r, T0)]
    return out


def emit_block(lines_, head, tail):
    w = 118
    print(head + " " * max(1, w - len(head)) + "\\")
    print("    asm volatile(" + " " * (w - 17) + "\\")
    for t in lines_[:-1]:
        s_ = '        "%s\\n\\t"' % t
        print(s_ + " " * max(1, w - len(s_)) + "\\")
    s_ = '        "%s"' % lines_[-1]
    print(s_ + " " * max(1, w - len(s_)) + "\\")
    for t in tail[:-1]:
        print(t + " " * max(1, w - len(t)) + "\\")
    print(tail[-1])


CLOB = "        : " + ALL_V + ", " + ALL_A + ', "vcc", "memory")'
P = "RPO_DQ_"
print("// GENERATED by tools/gen/gen_dq64w_body.py -- do not edit (tests/test_host_logic.py checks that the two stay in sync).")
print("// Register map, operand lists and the pipeline: the generator's docstring.")
x1_in = ", ".join('"v"(%s)' % x for x in ("T0", "T1", "T2", "T3", "V0", "V1"))
GEN = (SA, SB)
for t in range(8):
    cur, nxt = GEN[t & 1], GEN[(t + 1) & 1]
    body = x1(cur, nxt, t, True)
    print("// %d instructions" % len(body))
    emit_block(body, "#define %sX1_S%d(T0, T1, T2, T3, V0, V1)" % (P, t), ["        :", "        : " + x1_in, CLOB])
    emit_block(x1(cur, None, t, False), "#define %sX1L_S%d(T0, T1, T2, T3, V0, V1)" % (P, t), ["        :", "        : " + x1_in, CLOB])
    body = x2(nxt, t, True, False)
    print("// %d instructions" % len(body))
    emit_block(body, "#define %sX2_S%d(K0, K1, SCL)" % (P, t), ["        :", '        : "v"(K0), "v"(K1), "s"(SCL)', CLOB])
    emit_block(x2(nxt, t, True, True), "#define %sX2D_S%d(K0, K1, SCL, VK, VV, SK, SV, MK, MV)" % (P, t),
               ["        :", '        : "v"(K0), "v"(K1), "s"(SCL), "v"(VK), "v"(VV), "s"(SK), "s"(SV), "s"(MK), "s"(MV)', CLOB])
emit_block(x2(None, 0, False), "#define %sX2L()" % P, ["        :", "        :", CLOB])
# prologue: K / V rows of tile 0, both chains of tile 0, then (behind the mask) the first part of C(0) and the K rows of tile 1
emit_block(row_reads(AK, 0, 0) + row_reads(AV, 2, 0) + ["s_waitcnt lgkmcnt(0)"], "#define %sREAD0(K0, K1, V0, V1)" % P,
           ["        :", '        : "v"(K0), "v"(K1), "v"(V0), "v"(V1)', CLOB])
emit_block(chains_a(SA) + chains_b() + PAD, "#define %sCHAIN0()" % P, ["        :", "        :", CLOB])
emit_block(row_reads(AK, 0, 1) + ([] if NO_VALU else c_first(SA, "%2")) + ["s_waitcnt lgkmcnt(0)"], "#define %sPRE0(K0, K1, SCL)" % P,
           ["        :", '        : "v"(K0), "v"(K1), "s"(SCL)', CLOB])
for name, S in (("A", SA), ("B", SB)):
    emit_block(mask(S), "#define %sMASK_%s(D0, D1, D2, D3)" % (P, name), ["        :", '        : "v"(D0), "v"(D1), "v"(D2), "v"(D3)', CLOB])
emit_block(["v_accvgpr_write_b32 a%d, 0" % i for i in range(64)], "#define %sINIT_ACC()" % P,
           ["        :", "        :", "        : " + ", ".join('"a%d"' % i for i in range(64)) + ")"])
init = ["v_mov_b32 v%d, %%%d" % (LQ(n), n) for n in range(4)] + ["v_mov_b32 v%d, %%%d" % (DLQ(n) + j, 4 + n) for n in range(4) for j in range(4)]
emit_block(init, "#define %sINIT(L0, L1, L2, L3, D0, D1, D2, D3)" % P,
           ["        :", '        : "v"(L0), "v"(L1), "v"(L2), "v"(L3), "v"(D0), "v"(D1), "v"(D2), "v"(D3)', CLOB])
for nm, F in (("Q_TO_ACC", QA), ("DO_TO_ACC", DA)):
    print("#define %s%s(N, KS, W)" % (P, nm) + " " * 60 + "\\")
    print("    do {" + " " * 100 + "\\")
    for n in range(4):
        for ks in range(2):
            r = F(n, ks)
            print("        if ((N) == %d && (KS) == %d)" % (n, ks) + " " * 70 + "\\")
            print('            asm volatile("v_accvgpr_write_b32 a%d, %%0\\n\\tv_accvgpr_write_b32 a%d, %%1\\n\\tv_accvgpr_write_b32 a%d, %%2\\n\\t"' % (r, r + 1, r + 2) + "  \\")
            print('                         "v_accvgpr_write_b32 a%d, %%3" : : "v"((W)[0]), "v"((W)[1]), "v"((W)[2]), "v"((W)[3])' % (r + 3) + "  \\")
            print('                         : "a%d", "a%d", "a%d", "a%d");' % (r, r + 1, r + 2, r + 3) + " " * 40 + "\\")
    print("    } while (0)")
print("#define %sREAD_DQ(C, N, X0, X1, X2, X3)" % P + " " * 50 + "\\")
print("    do {" + " " * 100 + "\\")
for c in range(4):
    for n in range(4):
        r = OA(c, n)
        print("        if ((C) == %d && (N) == %d)" % (c, n) + " " * 70 + "\\")
        print('            asm volatile("v_accvgpr_read_b32 %%0, a%d\\n\\tv_accvgpr_read_b32 %%1, a%d\\n\\tv_accvgpr_read_b32 %%2, a%d\\n\\t"' % (r, r + 1, r + 2) + "  \\")
        print('                         "v_accvgpr_read_b32 %%3, a%d" : "=v"(X0), "=v"(X1), "=v"(X2), "=v"(X3));' % (r + 3) + "  \\")
print("    } while (0)")
