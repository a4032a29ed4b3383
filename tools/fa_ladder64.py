"""The phase ladder of the three head_dim-64 attention kernels (fa_fwd_kernel, fa_bwd_dq_kernel, fa_bwd_dkdv4_kernel) on the cfg-2
step's own batch shape: where a BLOCK's time goes (prologue / key-tile loop / epilogue until the last store is issued / until it has
landed) and where a CU's time goes (how long no resident block is inside its loop; how long the CU sits idle at the tail of the
launch), per kernel and per XCD.  It answers one question: is there 10 % to win by making a kernel persistent (next entry's
operands fetched under the current loop, stores left in flight), or is the loop itself the kernel?

Needs the diagnostic library: tools/exp/build_variant.sh ladder -DRPO_FA_LADDER  (one s_memrealtime record per block, 100 MHz).
usage: python tools/fa_ladder64.py > profiles/r06_fa_ladder64.md
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from rankpo_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, "tools", "exp", "librankpo_hip_ladder.so")
_lib._lib = None
from rankpo_amd import ops  # noqa: E402

DEV = "cuda"
TICK_US = 0.01                      # s_memrealtime: 100 MHz


def batch_lens(seed=1234):
    g = torch.Generator().manual_seed(seed)
    ql = torch.randint(640, 1281, (8,), generator=g)
    ql[0] = 1280
    pl = torch.randint(2048, 4097, (48,), generator=g)
    pl[0] = 4096
    lens = ql.tolist() + pl.tolist()
    fill = (-sum(lens)) % 256
    return lens + ([fill] if fill else [])


def read(lib, kernel, nblocks):
    buf = (C.c_ulonglong * (nblocks * 8))()
    assert lib.rpo_debug_fa_ladder(buf, kernel, nblocks, 1) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(nblocks, 8).astype(np.int64)
    return a[a[:, 7] == 1]


def union_len(iv):
    """total length of the union of intervals [(a, b)]"""
    iv = sorted(iv)
    tot, ca, cb = 0, None, None
    for a, b in iv:
        if ca is None:
            ca, cb = a, b
        elif a <= cb:
            cb = max(cb, b)
        else:
            tot += cb - ca
            ca, cb = a, b
    return tot + (cb - ca if ca is not None else 0)


def analyse(name, rec, ms_events, what_loop):
    t0, t1, t2, t3, t4 = (rec[:, i] for i in range(5))
    xcc = (rec[:, 5] >> 32) & 0xF
    hw = rec[:, 5] & 0xFFFFFFFF
    cu_key = xcc * 65536 + ((hw >> 8) & 0xFF)            # (XCC, SE | SH | CU): one compute unit
    start, end = t0.min(), t4.max()
    span = end - start
    life = (t4 - t0).sum()
    out = [f"### `{name}`", "",
           f"{len(rec)} blocks on {len(set(cu_key.tolist()))} CUs of {len(set(xcc.tolist()))} XCDs; launch span by the blocks' own clocks "
           f"{span * TICK_US / 1e3:.3f} ms (HIP events around the instrumented launch: {ms_events:.3f} ms); {rec[:, 6].sum()} {what_loop}, "
           f"{rec[:, 6].mean():.1f} per block.", "",
           "| share of the blocks' lifetimes | % |", "|---|---:|",
           f"| prologue (entry -> first loop iteration: work-list entry, bounds, Q / dO / row constants or K / V fragments, first stages issued) | {100 * (t1 - t0).sum() / life:.1f} |",
           f"| the loop | {100 * (t2 - t1).sum() / life:.1f} |",
           f"| epilogue until the last store is issued | {100 * (t3 - t2).sum() / life:.1f} |",
           f"| ... until the stores have landed | {100 * (t4 - t3).sum() / life:.1f} |", ""]
    # per CU: time with no resident block inside its loop; idle tail; concurrency
    rows = []
    per_xcd = {}
    for key in sorted(set(cu_key.tolist())):
        m = cu_key == key
        loops = list(zip(t1[m].tolist(), t2[m].tolist()))
        lives = list(zip(t0[m].tolist(), t4[m].tolist()))
        covered = union_len(loops)
        alive = union_len(lives)
        last = t4[m].max()
        first = t0[m].min()
        rows.append((key, (first - start) / span, (alive - covered) / span, (end - last) / span, (t4[m] - t0[m]).sum() / max(alive, 1)))
        per_xcd.setdefault(key >> 16, []).append(rows[-1])
    r = np.array([x[1:] for x in rows])
    out += ["| share of the launch span, mean over CUs (min .. max) | % |", "|---|---:|",
            f"| before the CU's first block starts | {100 * r[:, 0].mean():.2f} ({100 * r[:, 0].min():.2f} .. {100 * r[:, 0].max():.2f}) |",
            f"| a block is resident but NONE is inside its loop (prologue / epilogue not hidden by a co-resident block) | {100 * r[:, 1].mean():.2f} ({100 * r[:, 1].min():.2f} .. {100 * r[:, 1].max():.2f}) |",
            f"| idle at the tail (the CU's last block has ended, the launch has not) | {100 * r[:, 2].mean():.2f} ({100 * r[:, 2].min():.2f} .. {100 * r[:, 2].max():.2f}) |",
            f"| resident blocks per CU while any is resident | {r[:, 3].mean():.2f} |", "",
            "| XCD | CUs | not in a loop % | idle tail % | last block ends at % of the span |", "|---:|---:|---:|---:|---:|"]
    for x in sorted(per_xcd):
        a = np.array([y[1:] for y in per_xcd[x]])
        out.append(f"| {x} | {len(a)} | {100 * a[:, 1].mean():.2f} | {100 * a[:, 2].mean():.2f} | {100 * (1 - a[:, 2].min()):.1f} |")
    nonloop_tail = r[:, 0].mean() + r[:, 1].mean() + r[:, 2].mean()
    out += ["", f"**non-loop + tail = {100 * nonloop_tail:.1f} % of the launch** (what a persistent form could recover at most: its own "
            "prologue still costs issue slots, only its LATENCY hides).", ""]
    return out, nonloop_tail


def main():
    lib = _lib.load()
    lib.rpo_debug_fa_ladder.restype = C.c_int
    lib.rpo_debug_fa_ladder.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    torch.manual_seed(0)
    nh, nkv, hd = 32, 8, 64
    lens = batch_lens()
    T = sum(lens)
    qkv = torch.randn(T, (nh + 2 * nkv) * hd, device=DEV).to(torch.bfloat16)
    q, k, v = (x.view(T, -1, hd) for x in qkv.split([nh * hd, nkv * hd, nkv * hd], -1))
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    pos = torch.cat([torch.arange(n) for n in lens]).to(DEV)
    inv = 1.0 / (500000.0 ** (torch.arange(0, hd, 2, device=DEV, dtype=torch.float32) / hd))
    fr = torch.outer(pos.float(), inv)
    rope = (fr.cos().contiguous(), fr.sin().contiguous())
    tiles = ops.attn_tile_table(lens, DEV, nh, nkv)
    kt = ops.attn_key_tile_table(lens, DEV, nkv)
    scale = 0.125
    go = torch.randn(T, nh, hd, device=DEV).to(torch.bfloat16)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    # warm-up (clocks, code objects), then reset the records
    for _ in range(3):
        out, lse = ops.flash_attn_varlen_fwd(q, k, v, cu, tiles, scale)
        ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tiles, kt, scale, rope=rope)
    torch.cuda.synchronize()
    for kern, n in ((0, tiles.shape[0]), (1, tiles.shape[0]), (2, kt.shape[0])):
        lib.rpo_debug_fa_ladder(None, kern, n, 1)
    e0, e1, e2 = ev(), ev(), ev()
    e0.record()
    out, lse = ops.flash_attn_varlen_fwd(q, k, v, cu, tiles, scale)
    e1.record()
    ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tiles, kt, scale, rope=rope)
    e2.record()
    torch.cuda.synchronize()
    ms_f, ms_b = e0.elapsed_time(e1), e1.elapsed_time(e2)
    pairs = sum(n * (n + 1) // 2 for n in lens)
    print("# round 6: phase ladder of the head_dim-64 attention kernels (`tools/fa_ladder64.py`, library built with `-DRPO_FA_LADDER`)")
    print()
    print(f"Batch: the cfg-2 step's packed pass -- 8 queries of 640..1280 tokens + 48 passages of 2048..4096 (+ the filler), {len(lens)} "
          f"sequences, {T} tokens, 32 q heads / 8 kv heads, head_dim 64; backward with the rotary epilogues.  One `s_memrealtime` "
          f"(100 MHz) record per block: entry, loop start, loop end, last store issued, stores landed, XCC / HW id.  Instrumented "
          f"launches: forward {ms_f:.3f} ms = {4 * hd * nh * pairs / ms_f / 1e9 / 2500:.3f} of the MFMA peak, backward (dQ + dK/dV) "
          f"{ms_b:.3f} ms = {10 * hd * nh * pairs / ms_b / 1e9 / 2500:.3f} (five stamps per block: the rates are the shipped kernels' "
          f"within ~1 %).")
    print()
    verdicts = {}
    for kern, name, n, what in ((0, "fa_fwd_kernel", tiles.shape[0], "key tiles of 64"), (1, "fa_bwd_dq_kernel", tiles.shape[0], "key tiles of 64"),
                                (2, "fa_bwd_dkdv4_kernel", kt.shape[0], "(query slice, q head) iterations of 32 queries")):
        rec = read(lib, kern, n)
        lines, nl = analyse(name, rec, ms_f if kern == 0 else ms_b, what)
        verdicts[name] = nl
        print("\n".join(lines))
    print("## Reading")
    print()
    for name, nl in verdicts.items():
        print(f"* `{name}`: non-loop + tail {100 * nl:.1f} % -> " + ("persistence has something to recover" if nl >= 0.10 else
              "below the 10 % line: the loop is the kernel"))


if __name__ == "__main__":
    main()
