"""A/B of rpo_pool_normalize_fwd: the one-wave-per-sample kernel (round 6, librankpo_hip.so) against the block-per-sample kernel of
rounds 1-5 (tools/exp/librankpo_hip_r5pool.so: HEAD's other objects + the old pool_normalize.hip), same process, same buffers,
interleaved launches, HIP events on the launch stream.  Algorithmic bytes per call: N L 8 (mask, last-token mode) + 2 N d s.
usage: python tools/pool_bench.py [old_lib.so]   -> markdown on stdout"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rankpo_amd import _lib  # noqa: E402

SHAPES = [  # (N, L, d, dtype, mode, what)
    (4096, 512, 2048, torch.bfloat16, "last", "the verdict's encode()-scale shape"),
    (4096, 512, 2048, torch.bfloat16, "cls", "the same rows without a mask (what the packed path hands over)"),
    (64, 4096, 2048, torch.bfloat16, "last", "one encode() batch on the padded path"),
    (256, 1, 2048, torch.bfloat16, "cls", "one packed encode() batch x 4"),
    (56, 1, 2048, torch.bfloat16, "cls", "the cfg-2 training step (8 + 48 rows)"),
    (4096, 512, 1024, torch.float16, "cls", "BGE-M3 width, fp16"),
    (4096, 512, 384, torch.float32, "cls", "BGE-small width, f32"),
    (16384, 128, 4096, torch.bfloat16, "last", "Llama-3-8B width, short rows"),
]


def bind(path):
    lib = C.CDLL(path)
    res, args = _lib.SIGNATURES["rpo_pool_normalize_fwd"]
    lib.rpo_pool_normalize_fwd.restype, lib.rpo_pool_normalize_fwd.argtypes = res, args
    return lib


def main():
    new = _lib.load()
    # --shape I: that shape only (under `rocprofv3 --kernel-trace --stats` the per-kernel averages are then this shape's own kernel
    # durations: the Python loop issues a launch every ~10 us, so for kernels shorter than that the event brackets below measure the
    # launch rate, not the kernel)
    only = None
    argv = sys.argv[1:]
    if "--shape" in argv:
        only = int(argv[argv.index("--shape") + 1])
        del argv[argv.index("--shape"):argv.index("--shape") + 2]
    old_path = argv[0] if argv else os.path.join(ROOT, "tools", "exp", "librankpo_hip_r5pool.so")
    old = bind(old_path) if os.path.exists(old_path) else None
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    print("| N | L | d | dtype | mode | bytes (cold) | old us | old TB/s | new us | new TB/s | of 8 TB/s | same result | what |")
    print("|---:|---:|---:|---|---|---:|---:|---:|---:|---:|---:|---|---|")
    for N, L, d, dt, mode, what in (SHAPES if only is None else [SHAPES[only]]):
        g = torch.Generator(device=dev).manual_seed(N + L)
        es = torch.empty((), dtype=dt).element_size()
        nbytes = (N * L * 8 if mode == "last" else 0) + 2 * N * d * es
        # COLD data for every launch: the kernel's lines (mask rows + pooled rows + outputs) of consecutive launches come from NB
        # different buffer sets that together exceed the 256 MB Infinity Cache, so the rate is HBM's, not the cache's
        NB = max(2, min(16, -(-600 * 2 ** 20 // nbytes)))
        hs, masks = [], []
        for b in range(NB):
            if mode == "last":
                hs.append(torch.randn((N, L, d), generator=g, device=dev).to(dt) if b < 2 or N * L * d * es < 2 ** 31 else hs[b % 2])
                lens = torch.randint(1, L + 1, (N,), generator=g, device=dev)
                masks.append((torch.arange(L, device=dev)[None, :] < lens[:, None]).to(torch.int64))
            else:                               # CLS: only row 0 of every sample is touched -> [N, 1, d] buffers, all distinct
                hs.append(torch.randn((N, 1, d), generator=g, device=dev).to(dt))
                masks.append(None)
        code = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[dt]
        outs, turn = {}, {"new": 0, "old": 0}

        def call(lib, tag):
            b = turn[tag] % NB
            turn[tag] += 1
            h, mask = hs[b], masks[b]
            out = outs.setdefault((tag, b), (torch.empty((N, d), dtype=dt, device=dev), torch.empty((N,), dtype=torch.int32, device=dev),
                                             torch.empty((N,), dtype=torch.float32, device=dev)))
            rc = lib.rpo_pool_normalize_fwd(h.data_ptr(), h.stride(0), h.stride(1), mask.data_ptr() if mode == "last" else None, N,
                                            h.shape[1], d, code, 0 if mode == "last" else 1, 1, 1e-12, out[0].data_ptr(),
                                            out[1].data_ptr(), out[2].data_ptr(), st)
            assert rc == 0, rc
        libs = [("new", new)] + ([("old", old)] if old else [])
        times = {t: [] for t, _ in libs}
        for _ in range(NB):
            for t, lib in libs:
                call(lib, t)
        torch.cuda.synchronize()
        # REPS back-to-back launches between two events: a lone launch between two event records measures the ~15 us of the
        # bracket itself (the first version of this script did: every shape read 14-18 us)
        REPS = 20
        for _ in range(12):
            for t, lib in libs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _r in range(REPS):
                    call(lib, t)
                e1.record()
                times[t].append((e0, e1))
        torch.cuda.synchronize()
        med = {t: sorted(a.elapsed_time(b) for a, b in v)[len(v) // 2] * 1e3 / REPS for t, v in times.items()}
        same = "-"
        if old:
            same = str(all(bool(torch.equal(outs[("new", b)][1], outs[("old", b)][1])
                                and (outs[("new", b)][0].float() - outs[("old", b)][0].float()).abs().max().item() <= 2.0 ** -7)
                           for b in range(NB)))
        o_us = med.get("old")
        print(f"| {N} | {L} | {d} | {str(dt).split('.')[-1]} | {mode} | {nbytes} | {o_us and round(o_us, 2)} | "
              f"{o_us and round(nbytes / o_us / 1e6, 2)} | {med['new']:.2f} | {nbytes / med['new'] / 1e6:.2f} | "
              f"{nbytes / med['new'] / 1e6 / 8.0:.3f} | {same} | {what} |")


if __name__ == "__main__":
    main()
