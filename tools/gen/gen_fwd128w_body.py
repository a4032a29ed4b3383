"""Generates rankpo_amd/csrc/attention_fwd128w_gen.inc: the hand-placed instruction streams of fa_fwd128w_kernel, the head_dim-128
attention FORWARD at one wave per SIMD (round 5; the dK/dV kernels' structure, gen_dkdv128_body.py is the model).

A wave owns 64 queries (query tiles n = 0..3 of 16) of one head and walks 32-key tiles (key sub-tiles m = 0, 1 of 16).  Per key tile:
  S^T[m][n] += K rows (ks) x Q^T[n][ks]        4 k-steps of 32 (hd 128): 32 MFMAs, accumulators in VGPRs (two generations A / B)
  softmax      p = exp2(c S - mc[n]) (c = scale log2e, mc = the DEFERRED running maximum x c, below), packed to bf16 fragments
  O^T[c][n] += V^T[c] x P^T[n]                 8 hd tiles: 32 MFMAs, accumulators a[0:127]
  l[n]      += ones x P^T[n]                   4 MFMAs, accumulators a[128:143] (every register of l[n] = the row sum)
A software pipeline inside the ONE in-order stream: statement P1 of iteration t issues the S^T chains of tile t + 1 with the
exponentials of tile t in their gaps (and the V^T transposed reads of tile t); statement P2 issues the PV / l products of tile t and, in
their gaps, turns the scores of tile t + 1 into exponents IN PLACE (e = c s - mc[n]: what P1 of the next iteration hands to v_exp_f32)
and takes the LANE's largest e (and reads the K rows of tile t + 2).  Between P2(t) and P1(t + 1) the C++ side decides -- wave-uniformly,
one ballot -- whether any lane's e is over the threshold; only then RESCALE runs (reduction over a query's four lanes, mc += d,
e -= d, O and l *= 2^-d through VGPRs: the accumulators live in AGPRs, which vector instructions cannot read), so that p <= 2^THR
and the steady state has neither a rescale nor a cross-lane reduction.  Tile 0 has its own statement FIRST (mc = its row maximum).
A vector instruction holds the SIMD's issue port for 4 cycles (v_exp_f32: 8) and a 16x16x32 MFMA for 8 of its 16: the stream is
ISSUE-bound, not MFMA-bound, wherever the gaps carry more than 8 cycles of vector work per MFMA -- every instruction removed from the
steady state is 4 cycles per tile and wave.

Register map (literal; hipcc keeps v[0:63]):
  v[64:95]   SA[m][n] at 64 + 16 m + 4 n      v[96:127]  SB[m][n] at 96 + 16 m + 4 n       (score generations)
  a[208:239] AK[ks][m] at 208 + 4 (2 ks + m)  K row fragments of the tile whose chains run next (LDS reads straight into the accumulator
                                              file: the chains then read A and B from it and only their accumulators from the VGPRs the
                                              vector instructions of the gaps work on; GEN_AK_ACC=0: v[128:159])
  v[160:191] VT[c] at 160 + 4 c               V^T fragments (x = keys 0-15 in +0, +1; y = keys 16-31 in +2, +3)
  v[192:207] PF[n] at 192 + 4 n               P^T fragments: k-slots = keys {4g + j, 16 + 4g + (j - 4)}
  v[208:211] MC[n]                            the scale in force: c x (a maximum of the lane's row of query tile n, reset when outgrown)
  v[212:223] temporaries                      v[224:227] bf16 ones
  a[0:127]   O^T[c][n] at 4 (4 c + n)         a[128:143] l[n] at 128 + 4 n      a[144:207] Q^T[n][ks] at 144 + 16 n + 4 ks (B operands)
LDS: a ring of four 8-KiB K tiles at byte 0 and a ring of four 8-KiB V tiles at byte 32768 (tile j in slot j % 4 of either); the
statements come in FOUR variants S0..S3 (iteration t % 4: generation parity and both ring slots are functions of it), so that every
LDS address is a loop-invariant per-lane register + an immediate: the key-tile loop carries no address arithmetic.
Operands: P1: %0..%7 transposed-read addresses of V^T hd tiles 0..7 in V slot 0 (keys 16-31 at + 4096, slot s at + 8192 s), %8 = c (SGPR).
          P2: %0 = grow (out: the lane's largest e of the next tile), %1..%4 K row addresses of k-steps 0..3 in K slot 0 (key sub-tile 1 at + 4096,
          slot s at + 8192 s), %5 = c (SGPR).   MASK: %0..%3 = D[n] = min(query, len - 1) - k0 - 4 g of the lane's row of query tile n: key 16 m + r is masked
          iff 16 m + r > D[n].
"""
import os
import sys

HD = int(sys.argv[1]) if len(sys.argv) > 1 else 128  # head_dim: 128 (attention_fwd128w_gen.inc, macros RPO_FW_*) or 64 (attention_fwd64w_gen.inc, RPO_FW64_*)
assert HD in (64, 128)
KS = HD // 32                                        # k-steps of the S^T chains
NC = HD // 16                                        # hd tiles of O^T
ROW = 2 * HD                                         # bytes of a K / V row in LDS
IMG = 32 * ROW                                       # one 32-key image: 8 KiB (4 KiB at head_dim 64)
HALF = 16 * ROW                                      # keys 16-31 inside an image
PW = IMG // 4096                                     # LDS-DMA pieces of 1 KiB per wave and image (four waves)
PRE = "RPO_FW_" if HD == 128 else "RPO_FW64_"
A_O, A_L, A_Q, A_K = 0, 4 * 4 * NC, 4 * 4 * NC + 16, 4 * 4 * NC + 16 + 16 * KS      # accumulator file: O^T, l, Q^T, K fragments
NO_VALU = os.environ.get("GEN_NO_VALU") == "1"       # timing experiments only (results are wrong): no exponentials, no row maximum
B_VGPR = os.environ.get("GEN_B_VGPR") == "1"         # timing experiment: the chains' B operand from a VGPR instead of the Q^T AGPRs
NO_LDS = os.environ.get("GEN_NO_LDS") == "1"         # timing experiment: no LDS reads inside P1 / P2
NO_PAD = os.environ.get("GEN_NO_PAD") == "1"         # timing experiment: no hazard padding
RATE1 = int(os.environ.get("GEN_RATE1", "8"))        # issue cycles of vector instructions per MFMA gap in P1
RATE2 = int(os.environ.get("GEN_RATE2", "8"))        # ... in P2
SPLIT = os.environ.get("GEN_SPLIT", "1") == "1"      # the exponentials of query tile 3 in P2 instead of P1 (balances the two statements' issue load)
SA = lambda m, n: 64 + 16 * m + 4 * n
SB = lambda m, n: 96 + 16 * m + 4 * n
AK_ACC = os.environ.get("GEN_AK_ACC", "1") == "1"    # K row fragments in a[208:239] (LDS reads can target the accumulator file) instead of v[128:159]
AK = lambda ks, m: (A_K if AK_ACC else 128) + 4 * (2 * ks + m)
ak4 = lambda r: ("a[%d:%d]" if AK_ACC else "v[%d:%d]") % (r, r + 3)
VT = lambda c: 160 + 4 * c
PF = lambda n: 192 + 4 * n
MC = lambda n: 208 + n
T0, T1, T2, T3 = 216, 217, 218, 219
ONES = 224
OA = lambda c, n: 4 * (4 * c + n)
LA = lambda n: A_L + 4 * n
QA = lambda n, ks: A_Q + 4 * KS * n + 4 * ks
v4 = lambda r: "v[%d:%d]" % (r, r + 3)
a4 = lambda r: "a[%d:%d]" % (r, r + 3)
mf = "v_mfma_f32_16x16x32_bf16 "
NEG = "0xf149f2ca"       # -1e30f
ALL_V = ", ".join('"v%d"' % i for i in range(64, 228))
ALL_A = ", ".join('"a%d"' % i for i in range(0, A_K + 8 * KS if AK_ACC else A_K))
PAD = [] if os.environ.get("GEN_NO_PAD") == "1" else ["s_nop 15", "s_nop 7"]   # >= 18 wait states: an MFMA's result read by a vector instruction


def chain_mfmas(S):
    """S^T chains of one key tile into generation S: ks-major, the same accumulator comes round every 8 MFMAs."""
    out = []
    for ks in range(KS):
        for m in range(2):
            for n in range(4):
                c_in = "0" if ks == 0 else v4(S(m, n))
                b_op = ak4(AK(ks, m ^ 1)) if B_VGPR else a4(QA(n, ks))
                out.append(mf + "%s, %s, %s, %s" % (v4(S(m, n)), ak4(AK(ks, m)), b_op, c_in))
    return out


def pv_mfmas():
    """O^T and l products of one key tile, query-tile major (P^T of tile n is needed from MFMA 9 n on)."""
    out = []
    for n in range(4):
        for c in range(NC):
            out.append(mf + "%s, %s, %s, %s" % (a4(OA(c, n)), v4(VT(c)), v4(PF(n)), a4(OA(c, n))))
        out.append(mf + "%s, %s, %s, %s" % (a4(LA(n)), v4(ONES), v4(PF(n)), a4(LA(n))))
    return out


def exp_ops(S, ns=(0, 1, 2, 3)):
    """p = exp2(e) and packing of generation S (e = c s - mc, left in place by P2 / FIRST / RESCALE) for the query tiles ns, in
    order: P^T[n] is complete 4 instructions into the next tile's exponentials (the packing of tile n stands between them, so that no
    instruction reads the result of the transcendental right in front of it)."""
    ops, pending = [], []
    for n in ns:
        for m in range(2):
            for r in range(4):
                ops.append("v_exp_f32 v%d, v%d" % (S(m, n) + r, S(m, n) + r))
                if pending:
                    ops.append(pending.pop(0))
        # fragment words: lo = key sub-tile 0 rows 0-1, 2-3; hi = sub-tile 1
        pending = ["v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (PF(n) + 2 * m + h, S(m, n) + 2 * h, S(m, n) + 2 * h + 1) for m in range(2) for h in range(2)]
    return ops + ["s_nop 0"] + pending                            # transcendental result -> the VALU instruction behind it


TT = [216 + 2 * n for n in range(4)]                              # v[216:223]: (t, u) pairs of the four query tiles
TU = [x + 1 for x in TT]


def lane_max(S):
    """t[n] = maximum of the lane's 8 values (keys 4 g + r, 16 + 4 g + r) of query tile n, the four tiles interleaved."""
    t = TT
    steps = [lambda n: "v_max3_f32 v%d, v%d, v%d, v%d" % (t[n], S(0, n), S(0, n) + 1, S(0, n) + 2),
             lambda n: "v_max3_f32 v%d, v%d, v%d, v%d" % (t[n], t[n], S(0, n) + 3, S(1, n)),
             lambda n: "v_max3_f32 v%d, v%d, v%d, v%d" % (t[n], t[n], S(1, n) + 1, S(1, n) + 2),
             lambda n: "v_max_f32 v%d, v%d, v%d" % (t[n], t[n], S(1, n) + 3)]
    return [st(n) for st in steps for n in range(4)]


def row_max():
    """t[n] = maximum over the four lanes (g = 0..3) that hold a query's keys: two row swaps; the four chains interleaved (three
    other instructions stand between a VALU write and the swap that reads it: 2 wait states are required)."""
    t, u = TT, TU
    steps = [lambda n: "v_mov_b32 v%d, v%d" % (u[n], t[n]),
             lambda n: "v_permlane16_swap_b32 v%d, v%d" % (t[n], u[n]),
             lambda n: "v_max_f32 v%d, v%d, v%d" % (t[n], t[n], u[n]),
             lambda n: "v_mov_b32 v%d, v%d" % (u[n], t[n]),
             lambda n: "v_permlane32_swap_b32 v%d, v%d" % (t[n], u[n]),
             lambda n: "v_max_f32 v%d, v%d, v%d" % (t[n], t[n], u[n])]
    return [st(n) for st in steps for n in range(4)]


def arg_ops(S, c_op, grow_op):
    """e = c s - MC[n] in place over generation S (the exponent P1 of the next iteration feeds to v_exp_f32 as it is), and
    grow = the LANE's largest e.  No lane above the threshold means no row above it: the reduction over a query's four lanes is
    needed only where the scale is reset (RESCALE), so the steady state pays 32 + 18 instructions per tile instead of 32 + 62."""
    ops = []
    for n in range(4):
        for m in range(2):
            for r in range(4):
                ops.append("v_fma_f32 v%d, v%d, %s, -v%d" % (S(m, n) + r, S(m, n) + r, c_op, MC(n)))
    ops += lane_max(S)
    ops += ["v_max3_f32 %s, v%d, v%d, v%d" % (grow_op, TT[0], TT[1], TT[2]), "v_max_f32 %s, %s, v%d" % (grow_op, grow_op, TT[3])]
    return ops


def shift_ops(S, by):
    """e -= by[n] over generation S."""
    return ["v_sub_f32 v%d, v%d, v%d" % (S(m, n) + r, S(m, n) + r, by[n]) for n in range(4) for m in range(2) for r in range(4)]


def off(o):
    return " offset:%d" % o if o else ""


def tr_reads(first_op, slot=0):
    out = []
    for c in range(NC):
        out += ["ds_read_b64_tr_b16 v[%d:%d], %%%d%s" % (VT(c), VT(c) + 1, first_op + c, off(IMG * slot)),
                "ds_read_b64_tr_b16 v[%d:%d], %%%d%s" % (VT(c) + 2, VT(c) + 3, first_op + c, off(IMG * slot + HALF))]
    return out


def k_reads(first_op, slot=0):
    out = []
    for ks in range(KS):
        for m in range(2):
            out.append("ds_read_b128 %s, %%%d%s" % (ak4(AK(ks, m)), first_op + ks, off(IMG * slot + HALF * m)))
    return out


def cost(ins):
    """Cycles an instruction holds the SIMD's issue port (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost')."""
    return 8 if ins.startswith("v_exp_f32") else 4


def weave(mfmas, fillers, budget, lds=(), lds_per_gap=1, need=None, first_gap=0, free=0):
    """MFMAs in order; after MFMA k vector instructions whose issue costs sum to at most `budget` cycles (a 16x16x32 MFMA holds the
    port for 8 of its 16: 8 cycles of other work per gap are free) and `lds_per_gap` LDS reads.  need[k]: number of filler
    instructions that MUST be out before MFMA k issues (its operands): the rest of them is flushed in front of it.  The first
    `free` fillers may stand in the gaps before `first_gap`, the others not."""
    out, fi, li = [], 0, 0
    fillers, lds = list(fillers), list(lds)
    for k, ins in enumerate(mfmas):
        if need and need.get(k, 0) > fi:
            out += fillers[fi:need[k]]
            fi = need[k]
            out.append("s_nop 1")                                 # VALU-written VGPR -> MFMA operand
        out.append(ins)
        for _ in range(lds_per_gap):
            if li < len(lds):
                out.append(lds[li]); li += 1
        took = 0
        while fi < len(fillers) and (k >= first_gap or fi < free) and took + cost(fillers[fi]) <= budget:
            took += cost(fillers[fi])
            out.append(fillers[fi]); fi += 1
    out += lds[li:]
    out += fillers[fi:]
    return out


def p1(S_cur, S_next, chains=True, vslot=0):
    """exp / pack of S_cur -> PF, V^T reads of the same tile (V slot vslot) -> VT, and (chains) the S^T chains of the next tile -> S_next."""
    ex = [] if NO_VALU else exp_ops(S_cur, (0, 1, 2) if (chains and SPLIT) else (0, 1, 2, 3))
    tr = [] if NO_LDS else tr_reads(0, vslot)
    if chains:
        out = weave(chain_mfmas(S_next), ex, RATE1, tr, 1)
    else:
        out = tr + ex
    out.append("s_waitcnt lgkmcnt(0)")                            # V^T fragments landed
    return out


def dma_ops(first_op):
    """The wave's four LDS-DMA pieces of the tiles staged during this iteration, inside the stream (between MFMAs a piece costs
    ~60 cycles of issue; in hipcc's code behind the barrier, with the matrix pipe idle, it cost the whole of it).  Operands from
    first_op: per-lane byte offsets of the wave's two K pieces and two V pieces inside a tile (VGPR), the two tiles' global bases
    (SGPR pairs), the LDS addresses of the wave's first K / V piece (SGPR; the second piece lies 1024 bytes behind)."""
    vk0, vk1, vv0, vv1, sk, sv, mk, mv = range(first_op, first_op + 8)
    k0 = ["s_mov_b32 m0, %%%d" % mk, "s_nop 0", "global_load_lds_dwordx4 %%%d, %%%d" % (vk0, sk)]
    k1 = ["s_add_u32 m0, m0, 0x400", "s_nop 0", "global_load_lds_dwordx4 %%%d, %%%d" % (vk1, sk)]
    v0 = ["s_mov_b32 m0, %%%d" % mv, "s_nop 0", "global_load_lds_dwordx4 %%%d, %%%d" % (vv0, sv)]
    v1 = ["s_add_u32 m0, m0, 0x400", "s_nop 0", "global_load_lds_dwordx4 %%%d, %%%d" % (vv1, sv)]
    return [k0, k1, v0, v1] if PW == 2 else [k0, v0]              # (head_dim 64: one K and one V piece per wave; VK1 / VV1 unused)


def p2(S_next, pv=True, mx=True, kslot=0, dma=False, S_cur=None):
    """(pv) O^T / l products of the tile whose P^T is in PF; (mx) exponents of S_next in place, grow, and the K rows (K slot kslot) of
    the tile after it.  The vector work starts behind the SECOND product: the last chain MFMA of P1 is then >= 18 wait states back
    (a statement that reads S_next right behind P1 -- the mask, P2NOPV -- pads for itself)."""
    ops = (["v_mov_b32 %0, 0"] if NO_VALU else arg_ops(S_next, "%%%d" % (KS + 1), "%0")) if mx else []
    lds = k_reads(1, kslot) if (mx and not NO_LDS) else []
    if pv:
        ex = exp_ops(S_cur, (3,)) if (mx and SPLIT and not NO_VALU) else []          # P^T[3]: operand of the products from MFMA 3 (NC + 1) on
        out = weave(pv_mfmas(), ex + ops, RATE2, lds, 1, need={3 * (NC + 1): len(ex)} if ex else None, first_gap=2, free=len(ex))
        if dma:                                                   # one piece behind MFMAs 11, 17, 23, 29 of 36 (8, 14 of 20): the K row reads are out by then
            pieces = dma_ops(KS + 2)
            for gi, at in reversed(list(enumerate((11, 17, 23, 29) if HD == 128 else (8, 14)))):
                k = [i for i, t in enumerate(out) if t.startswith("v_mfma")][at]
                out[k + 1:k + 1] = pieces[gi]
    else:
        out = PAD + lds + ops
    out.append("s_waitcnt lgkmcnt(0)")
    return out


def mask(S):
    out = PAD + ["v_mov_b32 v%d, %s" % (T0, NEG)]
    for n in range(4):
        for m in range(2):
            for r in range(4):
                out += ["v_cmp_gt_i32 vcc, %d, %%%d" % (16 * m + r, n),
                        "v_cndmask_b32 v%d, v%d, v%d, vcc" % (S(m, n) + r, S(m, n) + r, T0)]
    return out


def first(S, kslot=1):
    """Tile 0 (generation S, scores masked already): MC[n] = c x the row maximum, e = c s - MC[n]; O and l are zero, nothing to
    scale.  Also the K rows of tile 1.  %0..%3 K row addresses, %4 = c."""
    out = PAD + ([] if NO_LDS else k_reads(0, kslot))
    out += ["v_mul_f32 v%d, %s, v%d" % (S(m, n) + r, "%%%d" % KS, S(m, n) + r) for n in range(4) for m in range(2) for r in range(4)]
    out += lane_max(S) + row_max()
    out += ["v_mov_b32 v%d, v%d" % (MC(n), TT[n]) for n in range(4)]
    out += shift_ops(S, TT)
    out.append("s_waitcnt lgkmcnt(0)")
    return out


def rescale(S):
    """A lane of generation S (e = c s - MC) went over the threshold: d[n] = max(row maximum of e, 0); MC[n] += d[n]; e -= d[n];
    O^T[.][n] and l[n] *= exp2(-d[n]) (through VGPRs: vector instructions cannot read the accumulator file)."""
    out = list(PAD)                                               # the PV products behind which this runs wrote O^T and l
    out += lane_max(S) + row_max()
    out += ["v_max_f32 v%d, 0, v%d" % (TT[n], TT[n]) for n in range(4)]
    out += ["v_add_f32 v%d, v%d, v%d" % (MC(n), MC(n), TT[n]) for n in range(4)]
    out += shift_ops(S, TT)
    out += ["v_sub_f32 v%d, 0, v%d" % (TU[n], TT[n]) for n in range(4)]
    out += ["v_exp_f32 v%d, v%d" % (TU[n], TU[n]) for n in range(4)]
    out.append("s_nop 0")
    for n in range(4):
        regs = [OA(c, n) + r for c in range(NC) for r in range(4)] + [LA(n) + r for r in range(4)]
        for i in range(0, len(regs), 4):                          # four at a time through v[212:215] (free: MN is gone)
            grp = regs[i:i + 4]
            out += ["v_accvgpr_read_b32 v%d, a%d" % (212 + j, a) for j, a in enumerate(grp)]
            out += ["v_mul_f32 v%d, v%d, v%d" % (212 + j, 212 + j, TU[n]) for j in range(len(grp))]
            out += ["v_accvgpr_write_b32 a%d, v%d" % (a, 212 + j) for j, a in enumerate(grp)]
    out.append("s_nop 3")                                         # v_accvgpr_write -> MFMA reading the register as its accumulator
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
print("// GENERATED by tools/gen/gen_fwd128w_body.py -- do not edit (tests/test_host_logic.py checks that the two stay in sync).")
print("// Register map, operand lists and the pipeline: the generator's docstring.")
TRS = ", ".join("TR%d" % i for i in range(NC))
tr_in = ", ".join('"v"(TR%d)' % i for i in range(NC))
KRS = ", ".join("R%d" % i for i in range(KS))
kr_in = ", ".join('"v"(R%d)' % i for i in range(KS))
GEN = (SA, SB)
for it in range(4):                      # iteration t = it (mod 4): current generation t % 2, V slot t % 4, K slot (t + 2) % 4
    cur, nxt = GEN[it & 1], GEN[(it + 1) & 1]
    body = p1(cur, nxt, True, it)
    print("// %d instructions" % len(body))
    emit_block(body, "#define " + PRE + "P1_S%d(%s, SCL)" % (it, TRS), ["        :", "        : " + tr_in + ', "s"(SCL)', CLOB])
    emit_block(p1(cur, None, False, it), "#define " + PRE + "P1L_S%d(%s, SCL)" % (it, TRS),
               ["        :", "        : " + tr_in + ', "s"(SCL)', CLOB])
    body = p2(nxt, True, True, (it + 2) & 3, False, cur)
    print("// %d instructions" % len(body))
    emit_block(body, "#define " + PRE + "P2_S%d(GROW, %s, SCL)" % (it, KRS),
               ['        : "=&v"(GROW)', "        : " + kr_in + ', "s"(SCL)', CLOB])
    emit_block(p2(nxt, True, True, (it + 2) & 3, True, cur), "#define " + PRE + "P2D_S%d(GROW, %s, SCL, VK0, VK1, VV0, VV1, SK, SV, MK, MV)" % (it, KRS),
               ['        : "=&v"(GROW)', "        : " + kr_in + ', "s"(SCL), "v"(VK0), "v"(VK1), "v"(VV0), "v"(VV1), '
                '"s"(SK), "s"(SV), "s"(MK), "s"(MV)', CLOB])
emit_block(first(SA), "#define " + PRE + "FIRST_A(%s, SCL)" % KRS,       # prologue: tile 0's scale and exponents, tile 1's K rows
           ["        :", "        : " + kr_in + ', "s"(SCL)', CLOB])
emit_block(p2(None, True, False) + PAD, "#define " + PRE + "P2L()", ["        :", "        :", CLOB])   # (+ pad: the epilogue reads O^T)
emit_block(k_reads(0) + ["s_waitcnt lgkmcnt(0)"], "#define " + PRE + "KREAD(%s)" % KRS,
           ["        :", "        : " + kr_in, CLOB])
emit_block(chain_mfmas(SA) + PAD, "#define " + PRE + "SCHAIN_A()", ["        :", "        :", CLOB])
for name, S in (("A", SA), ("B", SB)):
    emit_block(mask(S), "#define " + PRE + "MASK_%s(D0, D1, D2, D3)" % name,
               ["        :", '        : "v"(D0), "v"(D1), "v"(D2), "v"(D3)', CLOB])
for name, S in (("A", SA), ("B", SB)):
    emit_block(rescale(S), "#define " + PRE + "RESCALE_%s()" % name, ["        :", "        :", CLOB])
emit_block(["v_accvgpr_write_b32 a%d, 0" % i for i in range(A_Q)], "#define " + PRE + "INIT_ACC()",      # (early: under the first loads' latency;
           ["        :", "        :", "        : " + ", ".join('"a%d"' % i for i in range(A_Q)) + ")"])   # hipcc's code has no use for the accumulator file)
init = []
init += ["v_mov_b32 v%d, 0x3f803f80" % (ONES + i) for i in range(4)]
init += ["v_mov_b32 v%d, 0" % MC(n) for n in range(4)]
emit_block(init, "#define " + PRE + "INIT()", ["        :", "        :", CLOB])
# Q^T fragment (n, ks) -> a[144 + 16 n + 4 ks ...]: four 32-bit operands (an operand's sub-registers cannot be named)
print("#define " + PRE + "Q_TO_ACC(N, KS, W)" + " " * 60 + "\\")
print("    do {" + " " * 100 + "\\")
for n in range(4):
    for ks in range(KS):
        r = QA(n, ks)
        print("        if ((N) == %d && (KS) == %d)" % (n, ks) + " " * 70 + "\\")
        print('            asm volatile("v_accvgpr_write_b32 a%d, %%0\\n\\tv_accvgpr_write_b32 a%d, %%1\\n\\tv_accvgpr_write_b32 a%d, %%2\\n\\t"' % (r, r + 1, r + 2) + "  \\")
        print('                         "v_accvgpr_write_b32 a%d, %%3" : : "v"((W)[0]), "v"((W)[1]), "v"((W)[2]), "v"((W)[3])' % (r + 3) + "  \\")
        print('                         : "a%d", "a%d", "a%d", "a%d");' % (r, r + 1, r + 2, r + 3) + " " * 40 + "\\")
print("    } while (0)")
# epilogue: O^T[c][n] (4 registers) and l[n], MC[n] out of the literal registers
print("#define " + PRE + "READ_O(C, N, X0, X1, X2, X3)" + " " * 50 + "\\")
print("    do {" + " " * 100 + "\\")
for c in range(NC):
    for n in range(4):
        r = OA(c, n)
        print("        if ((C) == %d && (N) == %d)" % (c, n) + " " * 70 + "\\")
        print('            asm volatile("v_accvgpr_read_b32 %%0, a%d\\n\\tv_accvgpr_read_b32 %%1, a%d\\n\\tv_accvgpr_read_b32 %%2, a%d\\n\\t"' % (r, r + 1, r + 2) + "  \\")
        print('                         "v_accvgpr_read_b32 %%3, a%d" : "=v"(X0), "=v"(X1), "=v"(X2), "=v"(X3));' % (r + 3) + "  \\")
print("    } while (0)")
print("#define " + PRE + "READ_LM(N, L, M)" + " " * 60 + "\\")
print("    do {" + " " * 100 + "\\")
for n in range(4):
    print("        if ((N) == %d)" % n + " " * 80 + "\\")
    print('            asm volatile("v_accvgpr_read_b32 %%0, a%d\\n\\tv_mov_b32 %%1, v%d" : "=v"(L), "=v"(M));' % (LA(n), MC(n)) + "  \\")
print("    } while (0)")
