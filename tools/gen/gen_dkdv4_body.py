"""Generates the hand-placed instruction stream of one (active, unmasked) 32-query slice of fa_bwd_dkdv4_kernel:
M1 [S', dP' chains, key-tile major] -> V [exp2 / dS per key tile, packed to bf16] -> M2 [dV^T, dK^T, key-tile major], with
the VALU instructions of key tile n placed in the issue gaps of the MFMAs that follow its chains, so the matrix pipe never
waits for the vector ALU longer than one gap.  Output: the C macros RPO_D4_SLICE_BODY_LOAD / _HOT for attention.hip.

Register map (asm-owned; VGPRs are transient inside the statement except v[128:175], which the HOT variant expects filled by
the previous statement's prefetch; AGPRs persist for the whole kernel):
  v[64:95]   S[m][n]  (m = query tile 0/1, n = key tile 0..3) at 64 + 16 m + 4 n      v[96:127]  dP[m][n] likewise
  v[128:159] row fragments aq00 ad00 aq01 ad01 aq10 ad10 aq11 ad11 (aq<ks><m>)          v[160:175] lr0 lr1 dr0 dr1
  v[176:191] dO^T fragments atd[c], v[192:207] Q^T fragments atq[c] of THIS slice
  v[208:223] P fragments pf[n], v[224:239] dS fragments dsf[n]
  a[0:63] dva[c][n] at 16 c + 4 n, a[64:127] dka[c][n], a[128:159] bk[n][ks] at 128 + 8 n + 4 ks, a[160:191] bv[n][ks]
Operands: %0 row address k-step 0, %1 row address k-step 1, %2 row-constant address, %3..%6 transposed-read addresses (hd
tile c) of this slice's image, %7 scale * log2(e) (SGPR), %8..%10 = %0..%2 of the NEXT slice's image (prefetch); the DIAG bodies also take %11 = (k0 + fr) -
(qb + 4 g), the lane's first key minus its first query row (causal mask of the slices that hold the wave's diagonal).
"""
S = lambda m, n: 64 + 16 * m + 4 * n
DP = lambda m, n: 96 + 16 * m + 4 * n
AQ = {(0, 0): 128, (0, 1): 136, (1, 0): 144, (1, 1): 152}      # (ks, m) -> aq
AD = {(0, 0): 132, (0, 1): 140, (1, 0): 148, (1, 1): 156}
LR = {0: 160, 1: 164}
DR = {0: 168, 1: 172}
ATD = lambda c: 176 + 4 * c
ATQ = lambda c: 192 + 4 * c
PF = lambda n: 208 + 4 * n
DS = lambda n: 224 + 4 * n
BK = lambda n, ks: 128 + 8 * n + 4 * ks
BV = lambda n, ks: 160 + 8 * n + 4 * ks
DVA = lambda c, n: 16 * c + 4 * n
DKA = lambda c, n: 64 + 16 * c + 4 * n
v4 = lambda r: "v[%d:%d]" % (r, r + 3)
a4 = lambda r: "a[%d:%d]" % (r, r + 3)

loads = [
    "ds_read_b128 %s, %%2" % v4(LR[0]), "ds_read_b128 %s, %%2 offset:128" % v4(DR[0]),
    "ds_read_b128 %s, %%0" % v4(AQ[0, 0]), "ds_read_b128 %s, %%0 offset:4096" % v4(AD[0, 0]),
    "ds_read_b128 %s, %%2 offset:64" % v4(LR[1]), "ds_read_b128 %s, %%2 offset:192" % v4(DR[1]),
    "ds_read_b128 %s, %%0 offset:2048" % v4(AQ[0, 1]), "ds_read_b128 %s, %%0 offset:6144" % v4(AD[0, 1]),
    "ds_read_b128 %s, %%1" % v4(AQ[1, 0]), "ds_read_b128 %s, %%1 offset:4096" % v4(AD[1, 0]),
    "ds_read_b128 %s, %%1 offset:2048" % v4(AQ[1, 1]), "ds_read_b128 %s, %%1 offset:6144" % v4(AD[1, 1]),
]
prefetch = [l.replace("%0", "%8").replace("%1", "%9").replace("%2", "%10") for l in loads]
tr = []
for c in range(4):
    tr += ["ds_read_b64_tr_b16 v[%d:%d], %%%d offset:4096" % (ATD(c), ATD(c) + 1, 3 + c),
           "ds_read_b64_tr_b16 v[%d:%d], %%%d offset:6144" % (ATD(c) + 2, ATD(c) + 3, 3 + c),
           "ds_read_b64_tr_b16 v[%d:%d], %%%d" % (ATQ(c), ATQ(c) + 1, 3 + c),
           "ds_read_b64_tr_b16 v[%d:%d], %%%d offset:2048" % (ATQ(c) + 2, ATQ(c) + 3, 3 + c)]

mf = "v_mfma_f32_16x16x32_bf16 "


def mfma_list(diag):
    mfma = []
    for n in range(4):  # M1, key-tile major; the two MFMAs of a chain (same accumulator) are 4 apart
        for ks in range(2):
            for m in range(2):
                # diagonal slices: S[m][n] was initialised element by element (row constant, or -1e30 where key > query)
                c_in = v4(LR[m]) if ks == 0 and not diag else v4(S(m, n))
                mfma.append(mf + "%s, %s, %s, %s" % (v4(S(m, n)), v4(AQ[ks, m]), a4(BK(n, ks)), c_in))
            for m in range(2):
                c_in = v4(DR[m]) if ks == 0 else v4(DP(m, n))
                mfma.append(mf + "%s, %s, %s, %s" % (v4(DP(m, n)), v4(AD[ks, m]), a4(BV(n, ks)), c_in))
    for q in range(4):  # M2, key-tile major: quarter q needs the packed fragments of key tile q
        for c in range(4):
            mfma.append(mf + "%s, %s, %s, %s" % (a4(DVA(c, q)), v4(ATD(c)), v4(PF(q)), a4(DVA(c, q))))
            mfma.append(mf + "%s, %s, %s, %s" % (a4(DKA(c, q)), v4(ATQ(c)), v4(DS(q)), a4(DKA(c, q))))
    assert len(mfma) == 64
    return mfma


NEG = "0xf149f2ca"       # -1e30f: scale log2(e) x it is still finite, exp2 of it is exactly 0
NEGREG = DS(3) + 3


def diag_init(n):
    """Causal mask of a DIAGONAL slice (round 3) folded into the initial accumulators of the S' chains of key tile n: element
    (m, r) of the lane belongs to query row qb + 16 m + 4 g + r and key k0 + 16 n + fr; with d = (k0 + fr) - (qb + 4 g) (operand
    %11, one VGPR per lane) the key is visible iff 16 (m - n) + r >= d.  Visible: the row constant (as in the plain body), else
    -1e30 (held in the LAST register of dsf[3], which nothing writes before key tile 3's packing -- the masks of tiles 2 and 3 are
    placed while key tile 0's arithmetic already writes pf[0]), so that p = exp2(scale log2e S') = 0 and dS = p dP' = 0 exactly --
    the values hipcc's select-based path produces."""
    out = []
    for m in range(2):
        for r in range(4):
            out.append("v_cmp_ge_i32 vcc, %d, %%11" % (16 * (m - n) + r))
            out.append("v_cndmask_b32 v%d, v%d, v%d, vcc" % (S(m, n) + r, NEGREG, LR[m] + r))
    return out


chunks = []         # VALU work of key tile n, in dependency-friendly order (a result is used >= 8 instructions later)
for n in range(4):
    el = [(m, r) for m in range(2) for r in range(4)]
    ops = []
    for m, r in el:
        ops.append("v_mul_f32 v%d, %%7, v%d" % (S(m, n) + r, S(m, n) + r))
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
READY = lambda n: 8 * (n + 1) + 2      # two MFMAs of the next group have issued: >= 12 wait states after chain n's last MFMA
RATE = 3                               # VALU instructions per MFMA gap


import os
NO_VALU = os.environ.get("GEN_NO_VALU") == "1"      # timing experiments only (results are wrong)
NO_TR = os.environ.get("GEN_NO_TR") == "1"
NO_PREFETCH = os.environ.get("GEN_NO_PREFETCH") == "1"


TR_RATE = int(os.environ.get("GEN_TR_RATE", "1"))   # transposed reads issued per MFMA gap of M1 (0: all in front of M1, round 2)
PF_RATE = int(os.environ.get("GEN_PF_RATE", "1"))   # prefetch reads issued per MFMA gap of M2 (0: all in front of M2, round 2)


def build(hot, diag=False):
    """Round 3: (i) one wave per SIMD = ONE in-order stream, so a burst of LDS reads stalls the MFMAs behind it while the LDS
    queue drains: the 16 transposed reads of THIS slice go out TR_RATE per MFMA gap from the start of M1 (needed by M2), the 12
    row reads of the NEXT slice PF_RATE per gap of M2; (ii) diag: the slices that hold the wave's diagonal run the same stream
    with the causal mask in the S' chains' initial accumulators (key tile n's 16 mask instructions sit in front of its chains)."""
    mfma = mfma_list(diag)
    out = []
    if not hot:
        out += loads
    out.append("s_waitcnt lgkmcnt(0)")             # row fragments / row constants of this slice are in v[128:175]
    lds = [] if NO_TR else list(tr)
    if TR_RATE == 0:
        out += lds
        lds = []
    pre = {}                                        # key tile -> mask instructions still to place in front of its chains
    if diag:
        out.append("v_mov_b32 v%d, %s" % (NEGREG, NEG))
        out += diag_init(0)
        out.append("s_nop 1")
        pre = {n: diag_init(n) for n in (1, 2, 3)}
    queue = [] if NO_VALU else [(op, READY(n), n) for n in range(4) for op in chunks[n]]
    vi = 0
    for k, ins in enumerate(mfma):
        if k < 32 and k % 8 == 0 and pre.get(k // 8):
            out += pre.pop(k // 8)                  # (whatever of key tile k / 8's mask is not out yet)
            out.append("s_nop 1")
        if k >= 32 and (k - 32) % 8 == 0:
            q = (k - 32) // 8                      # quarter q of M2 reads pf[q] / dsf[q]: key tile q's arithmetic must be out
            while vi < len(queue) and queue[vi][2] <= q:
                out.append(queue[vi][0]); vi += 1
            if k == 32:
                out += lds                          # (any transposed read not yet issued)
                out.append("s_waitcnt lgkmcnt(0)")  # the transposed fragments
                lds = [] if NO_PREFETCH else list(prefetch)   # M1 is done with v[128:175]: the next slice's operands
                if PF_RATE == 0:
                    out += lds
                    lds = []
            out.append("s_nop 1")                  # VALU-written VGPR -> MFMA operand
        out.append(ins)
        for _ in range(TR_RATE if k < 32 else PF_RATE):
            if lds:
                out.append(lds.pop(0))
        nxt = k // 8 + 1                            # the mask of the NEXT key tile: 4 instructions per gap, out 4 gaps early
        if k < 24 and pre.get(nxt):
            out += pre[nxt][:4]
            pre[nxt] = pre[nxt][4:]
        took = 0
        while vi < len(queue) and queue[vi][1] <= k and took < RATE:
            out.append(queue[vi][0]); vi += 1; took += 1
    assert vi == len(queue) and not any(pre.values())
    out += lds
    return out


def emit(name, out, diag=False):
    lines = ['        "%s\\n\\t"%s' % (t, " " * max(1, 104 - len(t)) + "\\") for t in out[:-1]]
    lines += ['        "%s"%s' % (out[-1], " " * max(1, 108 - len(out[-1])) + "\\")]
    clob = ", ".join('"v%d"' % i for i in range(64, 256)) + ", " + ", ".join('"a%d"' % i for i in range(0, 192)) + (', "vcc"' if diag else "") + ', "memory"'
    print("// generated by tools/gen/gen_dkdv4_body.py (register map and operand list there); %d instructions" % len(out))
    print("#define %s(RA0, RA1, LRD, TP0, TP1, TP2, TP3, SCL, NRA0, NRA1, NLRD%s)" % (name, ", DLANE" if diag else "") + " " * 20 + "\\")
    print("    asm volatile(" + " " * 95 + "\\")
    print("\n".join(lines))
    print("        :" + " " * 107 + "\\")
    print('        : "v"(RA0), "v"(RA1), "v"(LRD), "v"(TP0), "v"(TP1), "v"(TP2), "v"(TP3), "s"(SCL), "v"(NRA0), "v"(NRA1),' + " " * 3 + "\\")
    print('          "v"(NLRD)%s' % (', "v"(DLANE)' if diag else "") + " " * 80 + "\\")
    print("        : " + clob + ")")


emit("RPO_D4_SLICE_BODY_LOAD", build(False))
emit("RPO_D4_SLICE_BODY_HOT", build(True))
emit("RPO_D4_DIAG_BODY_LOAD", build(False, True), True)
emit("RPO_D4_DIAG_BODY_HOT", build(True, True), True)
