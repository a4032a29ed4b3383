"""Generates rankpo_amd/csrc/attention_dkdv128_gen.inc: the hand-placed instruction stream of one (active, unmasked) 32-query slice
of fa_bwd_dkdv128_kernel (head_dim 128; a wave owns 32 keys = key tiles n = 0, 1) and the literal-register MFMA statements of
its masked path.  The head_dim-64 kernel's generator (gen_dkdv4_body.py) is the model; what differs is the shape of a slice:

  M1  S'[m][n]  += Q rows (ks)  x K[n][ks]      4 k-steps of 32 (hd 128), 2 query tiles m, 2 key tiles n: 32 MFMAs
      dP'[m][n] += dO rows (ks) x V[n][ks]
  V   p = exp2(scale log2e S'), dS = p dP', packed to bf16 fragments: 32 VALU instructions per key tile (64 per slice;
      head_dim 64: 128), placed in the issue gaps of the MFMAs that follow the key tile's chains
  M2  dV^T[c][n] += dO^T[c] x P[n], dK^T[c][n] += Q^T[c] x dS[n]    8 hd tiles c, 2 key tiles n: 32 MFMAs

Register map (asm-owned: v[64:255] are transient inside a statement except v[96:175], which the HOT variant expects filled by
the previous statement's prefetch; AGPRs persist for the whole kernel):
  v[64:79]   S[m][n] at 64 + 8 m + 4 n            v[80:95]   dP[m][n] at 80 + 8 m + 4 n
  v[96:127]  Q row fragments aq[ks][m] at 96 + 4 (2 ks + m)      v[128:159] dO row fragments ad[ks][m] at 128 + 4 (2 ks + m)
  v[160:167] lr[m] (-lse / scale of the slice's rows)            v[168:175] dr[m] (-delta)
  v[176:207] dO^T fragments atd[c] at 176 + 4 c   v[208:239] Q^T fragments atq[c] at 208 + 4 c
  v[240:247] P fragments pf[n]                    v[248:255] dS fragments dsf[n]
  a[0:63] dV^T[c][n] at 8 c + 4 n, a[64:127] dK^T[c][n] at 64 + 8 c + 4 n, a[128:159] bk[n][ks] at 128 + 16 n + 4 ks,
  a[160:191] bv[n][ks] at 160 + 16 n + 4 ks
Operands of the body: %0..%3 row addresses of k-steps 0..3 (image base + row_off[ks]; query tile m at + 4096, dO at + 8192),
%4 row-constant address (image + 16384 + 16 g), %5..%12 transposed-read addresses of hd tiles 0..7 (dO at + 8192, queries 16..31
at + 4096), %13 scale * log2(e) (SGPR), %14..%17 / %18 = %0..%3 / %4 of the NEXT slice's image (prefetch); the DIAG bodies also take %19 = (k0 + fr) - (qb + 4 g), the
lane's key minus its first query row (causal mask of the slice that holds the wave's diagonal, see diag_init).
LDS image of a slice (fa_bwd_dkdv128_kernel): Q 32 rows x 256 B | dO 32 x 256 B | 32 x -lse / scale | 32 x -delta = 16640 B.
"""
import os
import sys

S = lambda m, n: 64 + 8 * m + 4 * n
DP = lambda m, n: 80 + 8 * m + 4 * n
AQ = lambda ks, m: 96 + 4 * (2 * ks + m)
AD = lambda ks, m: 128 + 4 * (2 * ks + m)
LR = lambda m: 160 + 4 * m
DR = lambda m: 168 + 4 * m
ATD = lambda c: 176 + 4 * c
ATQ = lambda c: 208 + 4 * c
PF = lambda n: 240 + 4 * n
DS = lambda n: 248 + 4 * n
DVA = lambda c, n: 8 * c + 4 * n
DKA = lambda c, n: 64 + 8 * c + 4 * n
BK = lambda n, ks: 128 + 16 * n + 4 * ks
BV = lambda n, ks: 160 + 16 * n + 4 * ks
v4 = lambda r: "v[%d:%d]" % (r, r + 3)
a4 = lambda r: "a[%d:%d]" % (r, r + 3)
mf = "v_mfma_f32_16x16x32_bf16 "

DO_OFF, M_OFF = 8192, 4096           # dO part of the image; query tile 1 (rows 16..31)

def row_loads(row_ops, const_op):
    """The 20 ds_read_b128 of a slice's row constants and row fragments; row_ops = operand numbers of the 4 row addresses."""
    out = []
    for m in range(2):
        out += ["ds_read_b128 %s, %%%d offset:%d" % (v4(LR(m)), const_op, 64 * m),
                "ds_read_b128 %s, %%%d offset:%d" % (v4(DR(m)), const_op, 128 + 64 * m)]
    for ks in range(4):
        for m in range(2):
            out += ["ds_read_b128 %s, %%%d offset:%d" % (v4(AQ(ks, m)), row_ops[ks], M_OFF * m),
                    "ds_read_b128 %s, %%%d offset:%d" % (v4(AD(ks, m)), row_ops[ks], DO_OFF + M_OFF * m)]
    assert len(out) == 20
    return out


loads = row_loads([0, 1, 2, 3], 4)
prefetch = row_loads([14, 15, 16, 17], 18)          # the same from the NEXT image
tr = []
for c in range(8):
    tr += ["ds_read_b64_tr_b16 v[%d:%d], %%%d offset:%d" % (ATD(c), ATD(c) + 1, 5 + c, DO_OFF),
           "ds_read_b64_tr_b16 v[%d:%d], %%%d offset:%d" % (ATD(c) + 2, ATD(c) + 3, 5 + c, DO_OFF + M_OFF),
           "ds_read_b64_tr_b16 v[%d:%d], %%%d" % (ATQ(c), ATQ(c) + 1, 5 + c),
           "ds_read_b64_tr_b16 v[%d:%d], %%%d offset:%d" % (ATQ(c) + 2, ATQ(c) + 3, 5 + c, M_OFF)]

def mfma_list(diag):
  mfma = []
  for n in range(2):       # M1, key-tile major; consecutive MFMAs of one chain (same accumulator) are 4 apart
    for ks in range(4):
        for m in range(2):
            # diagonal slices: S[m][n] was initialised element by element (row constant, or -1e30 where key > query)
            c_in = v4(LR(m)) if ks == 0 and not diag else v4(S(m, n))
            mfma.append(mf + "%s, %s, %s, %s" % (v4(S(m, n)), v4(AQ(ks, m)), a4(BK(n, ks)), c_in))
        for m in range(2):
            c_in = v4(DR(m)) if ks == 0 else v4(DP(m, n))
            mfma.append(mf + "%s, %s, %s, %s" % (v4(DP(m, n)), v4(AD(ks, m)), a4(BV(n, ks)), c_in))
  for q in range(2):       # M2, key-tile major: half q needs the packed fragments of key tile q
    for c in range(8):
        mfma.append(mf + "%s, %s, %s, %s" % (a4(DVA(c, q)), v4(ATD(c)), v4(PF(q)), a4(DVA(c, q))))
        mfma.append(mf + "%s, %s, %s, %s" % (a4(DKA(c, q)), v4(ATQ(c)), v4(DS(q)), a4(DKA(c, q))))
  assert len(mfma) == 64
  return mfma


NEG = "0xf149f2ca"       # -1e30f: scale log2(e) x it is still finite, exp2 of it is exactly 0


def diag_init(n):
    """Causal mask of a DIAGONAL slice folded into the initial accumulators of the S' chains of key tile n: element (m, r) of
    the lane belongs to query row qb + 16 m + 4 g + r and key k0 + 16 n + fr; with d = (k0 + fr) - (qb + 4 g) (operand %19, one
    VGPR per lane) the key is visible iff 16 (m - n) + r >= d.  Visible: the row constant (as in the plain body), else -1e30, so
    that p = exp2(scale log2e S') = 0 and dS = p dP' = 0 exactly -- the values hipcc's select-based path produces."""
    out = []
    for m in range(2):
        for r in range(4):
            out.append("v_cmp_ge_i32 vcc, %d, %%19" % (16 * (m - n) + r))
            # (a literal next to vcc would be a second constant-bus read: -1e30 sits in a VGPR, pf[0]'s first register, which
            # is not written before key tile 0's arithmetic starts, long after these)
            out.append("v_cndmask_b32 v%d, v%d, v%d, vcc" % (S(m, n) + r, PF(0), LR(m) + r))
    return out


chunks = []              # VALU work of key tile n, in dependency-friendly order (a result is used >= 8 instructions later)
for n in range(2):
    el = [(m, r) for m in range(2) for r in range(4)]
    ops = []
    for m, r in el:
        ops.append("v_mul_f32 v%d, %%13, v%d" % (S(m, n) + r, S(m, n) + r))
    for m, r in el:
        ops.append("v_exp_f32 v%d, v%d" % (S(m, n) + r, S(m, n) + r))
    for m, r in el:
        ops.append("v_mul_f32 v%d, v%d, v%d" % (DP(m, n) + r, S(m, n) + r, DP(m, n) + r))
    for m in range(2):   # fragment words (tile m rows 0-1), (tile m rows 2-3): k-slots {4g + j, 16 + 4g + (j - 4)}
        for h in range(2):
            ops.append("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (PF(n) + 2 * m + h, S(m, n) + 2 * h, S(m, n) + 2 * h + 1))
    for m in range(2):
        for h in range(2):
            ops.append("v_cvt_pk_bf16_f32 v%d, v%d, v%d" % (DS(n) + 2 * m + h, DP(m, n) + 2 * h, DP(m, n) + 2 * h + 1))
    chunks.append(ops)
READY = lambda n: 16 * (n + 1) + 2     # two MFMAs of the next group have issued: >= 12 wait states after chain n's last MFMA
RATE = int(os.environ.get("GEN_RATE", "3"))     # VALU instructions per MFMA gap
NO_VALU = os.environ.get("GEN_NO_VALU") == "1"      # timing experiments only (results are wrong)
NO_PREFETCH = os.environ.get("GEN_NO_PREFETCH") == "1"
NO_TR = os.environ.get("GEN_NO_TR") == "1"
ALL_V = ", ".join('"v%d"' % i for i in range(64, 256))
ALL_A = ", ".join('"a%d"' % i for i in range(0, 192))


TR_RATE = int(os.environ.get("GEN_TR_RATE", "2"))   # transposed reads issued per MFMA gap of M1 (0: all in front of M1)
PF_RATE = int(os.environ.get("GEN_PF_RATE", "1"))   # prefetch reads issued per MFMA gap of M2 (0: all in front of M2)


def build(hot, diag=False):
    """One wave per SIMD = ONE in-order stream: a burst of LDS reads stalls the MFMAs behind it while the LDS queue (shared by
    the CU's four waves: 144 KB of reads per slice against 128 B/clk) drains.  So the 32 transposed reads of THIS slice go out
    TR_RATE per MFMA gap from the start of M1 (needed by M2), the 20 row reads of the NEXT slice PF_RATE per gap of M2.
    diag: the slice that holds the wave's diagonal -- same stream, the causal mask in the S' chains' initial accumulators."""
    mfma = mfma_list(diag)
    out = []
    if not hot:
        out += loads
    out.append("s_waitcnt lgkmcnt(0)")             # row fragments / row constants of this slice are in v[96:175]
    lds = [] if NO_TR else list(tr)
    if TR_RATE == 0:
        out += lds
        lds = []
    pre = []                                        # VALU work in front of a key tile's chains (diag only)
    if diag:
        out.append("v_mov_b32 v%d, %s" % (PF(0), NEG))
        out += diag_init(0)                         # key tile 0 before M1 starts ...
        out.append("s_nop 1")
        pre = diag_init(1)                          # ... key tile 1 in the gaps of key tile 0's MFMAs (needed at k = 16)
    queue = [] if NO_VALU else [(op, READY(n), n) for n in range(2) for op in chunks[n]]
    vi = 0
    for k, ins in enumerate(mfma):
        if k == 16 and pre:
            out += pre
            pre = []
            out.append("s_nop 1")
        if k >= 32 and (k - 32) % 16 == 0:
            q = (k - 32) // 16                     # half q of M2 reads pf[q] / dsf[q]: key tile q's arithmetic must be out
            while vi < len(queue) and queue[vi][2] <= q:
                out.append(queue[vi][0]); vi += 1
            if k == 32:
                out += lds                          # (any transposed read not yet issued)
                out.append("s_waitcnt lgkmcnt(0)")  # the transposed fragments
                lds = [] if NO_PREFETCH else list(prefetch)   # M1 is done with v[96:175]: the next slice's operands
                if PF_RATE == 0:
                    out += lds
                    lds = []
            out.append("s_nop 1")                  # VALU-written VGPR -> MFMA operand
        out.append(ins)
        rate = TR_RATE if k < 32 else PF_RATE
        for _ in range(rate):
            if lds:
                out.append(lds.pop(0))
        if pre and k < 14:                          # two of key tile 1's mask instructions per gap: done by k = 8
            out += pre[:2]
            pre = pre[2:]
        took = 0
        while vi < len(queue) and queue[vi][1] <= k and took < RATE:
            out.append(queue[vi][0]); vi += 1; took += 1
    assert vi == len(queue)
    out += lds
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


def emit_body(name, out, diag=False):
    print("// %d instructions" % len(out))
    ops = ", ".join('"v"(A%d)' % i for i in range(13)) + ', "s"(SCL), ' + ", ".join('"v"(N%d)' % i for i in range(5))
    if diag:
        ops += ', "v"(DLANE)'
    emit_block(out, "#define %s(A0, A1, A2, A3, A4, A5, A6, A7, A8, A9, A10, A11, A12, SCL, N0, N1, N2, N3, N4%s)"
               % (name, ", DLANE" if diag else ""),
               ["        :", "        : " + ops, "        : " + ALL_V + ", " + ALL_A + (', "vcc"' if diag else "") + ', "memory")'])


print("// GENERATED by tools/gen/gen_dkdv128_body.py -- do not edit (tests/test_host_logic.py checks that the two stay in sync).")
print("// Register map, operand list and the LDS image layout: the generator's docstring.")
emit_body("RPO_D128_SLICE_BODY_LOAD", build(False))
emit_body("RPO_D128_SLICE_BODY_HOT", build(True))
emit_body("RPO_D128_DIAG_BODY_LOAD", build(False, True), True)
emit_body("RPO_D128_DIAG_BODY_HOT", build(True, True), True)

# ---- literal-register statements of the masked path and of the prologue / epilogue ------------------------------------------------
for ks in range(4):      # M1, one k-step of one query tile M: S'[M][n] and dP'[M][n] for both key tiles
    body = ["s_nop 1"]
    for n in range(2):
        body.append(mf + "%%%d, %%4, %s, %%%d" % (n, a4(BK(n, ks)), n))
        body.append(mf + "%%%d, %%5, %s, %%%d" % (2 + n, a4(BV(n, ks)), 2 + n))
    emit_block(body, "#define RPO_D128_M1_KS%d(M, AQ, AD)" % ks,
               ['        : "+v"(s[M][0]), "+v"(s[M][1]), "+v"(dp[M][0]), "+v"(dp[M][1])', '        : "v"(AQ), "v"(AD)',
                "        : " + ALL_A + ")"])
for c in range(8):       # M2, one hd tile: dV^T[c][n] += dO^T[c] P[n], dK^T[c][n] += Q^T[c] dS[n]
    body = ["s_nop 1"]
    for n in range(2):
        body.append(mf + "%s, %%0, %%%d, %s" % (a4(DVA(c, n)), 2 + n, a4(DVA(c, n))))
        body.append(mf + "%s, %%1, %%%d, %s" % (a4(DKA(c, n)), 4 + n, a4(DKA(c, n))))
    emit_block(body, "#define RPO_D128_M2_C%d(ATD, ATQ, PF, DS)" % c,
               ["        :", '        : "v"(ATD), "v"(ATQ), "v"(PF[0]), "v"(PF[1]), "v"(DS[0]), "v"(DS[1])', "        : " + ALL_A + ")"])
# K / V fragments -> a[128:191]
print("#define RPO_D128_KV_TO_ACC(BKF, BVF)" + " " * 60 + "\\")
print("    do {" + " " * 100 + "\\")
for arr, base in (("BKF", BK), ("BVF", BV)):
    for n in range(2):
        for ks in range(4):
            r = base(n, ks)
            print("        { const uint4_t w_ = __builtin_bit_cast(uint4_t, %s[%d][%d]);" % (arr, n, ks) + " " * 20 + "\\")
            print('          asm volatile("v_accvgpr_write_b32 a%d, %%0\\n\\tv_accvgpr_write_b32 a%d, %%1\\n\\tv_accvgpr_write_b32 a%d, %%2\\n\\t"' % (r, r + 1, r + 2) + "  \\")
            print('                       "v_accvgpr_write_b32 a%d, %%3" : : "v"(w_[0]), "v"(w_[1]), "v"(w_[2]), "v"(w_[3])' % (r + 3) + "  \\")
            print('                       : "a%d", "a%d", "a%d", "a%d"); }' % (r, r + 1, r + 2, r + 3) + " " * 40 + "\\")
print("    } while (0)")
zero = ["v_accvgpr_write_b32 a%d, 0" % i for i in range(128)]
emit_block(zero, "#define RPO_D128_ZERO_ACC()", ["        :", "        :", "        : " + ", ".join('"a%d"' % i for i in range(128)) + ")"])
