"""GPU tests of the hand-written causal varlen flash attention (head_dim 64, bf16, GQA) against an f32 reference
(per sequence: softmax(scale q k^T + causal mask) v in float32 on the same bf16 inputs)."""
import math
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def ref_attention(q, k, v, lens, scale):
    outs, lses, o0 = [], [], 0
    nh, nkv = q.shape[1], k.shape[1]
    for n in lens:
        qq = q[o0:o0 + n].float().transpose(0, 1)                       # [nh, n, hd]
        kk = k[o0:o0 + n].float().transpose(0, 1).repeat_interleave(nh // nkv, 0)
        vv = v[o0:o0 + n].float().transpose(0, 1).repeat_interleave(nh // nkv, 0)
        s = qq @ kk.transpose(1, 2) * scale
        s = s.masked_fill(~torch.ones(n, n, dtype=torch.bool, device=q.device).tril(), float("-inf"))
        lses.append(torch.logsumexp(s, -1))                             # [nh, n]
        outs.append((torch.softmax(s, -1) @ vv).transpose(0, 1))        # [n, nh, hd]
        o0 += n
    return torch.cat(outs), torch.cat(lses, 1)


@pytest.mark.parametrize("lens,nh,nkv,fused", [
    ([128], 4, 4, False), ([64, 1, 200, 129, 33], 8, 2, False), ([300, 17, 513, 128, 256, 5], 32, 8, True),
    ([1000, 777], 4, 1, True)])
def test_flash_attn_fwd_matches_reference(lens, nh, nkv, fused):
    from rankpo_amd import ops
    torch.manual_seed(sum(lens))
    T, hd = sum(lens), 64
    if fused:   # strided views of one q|k|v projection output, as the encoder produces them
        qkv = torch.randn(T, (nh + 2 * nkv) * hd, device=DEV).to(torch.bfloat16)
        q, k, v = qkv.split([nh * hd, nkv * hd, nkv * hd], -1)
        q, k, v = q.view(T, nh, hd), k.view(T, nkv, hd), v.view(T, nkv, hd)
    else:
        q = torch.randn(T, nh, hd, device=DEV).to(torch.bfloat16)
        k = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
        v = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    tiles = ops.attn_tile_table(lens, DEV)
    scale = 1.0 / math.sqrt(hd)
    out, lse = ops.flash_attn_varlen_fwd(q, k, v, cu, tiles, scale)
    ro, rl = ref_attention(q, k, v, lens, scale)
    assert (out.float() - ro).abs().max() < 2.5e-2          # bf16 P and bf16 output: ~2^-7 relative on O(1) values
    assert (lse - rl).abs().max() < 2e-3
    # padded [N, nh, max_len] layout (what PyTorch's backward op reads)
    _, lp = ops.flash_attn_varlen_fwd(q, k, v, cu, tiles, scale, padded_lse_len=max(lens), num_seqs=len(lens))
    o0 = 0
    for i, n in enumerate(lens):
        assert torch.equal(lp[i, :, :n], lse[:, o0:o0 + n])
        o0 += n
    # autograd path: HIP forward + flash backward op, against autograd through the f32 reference
    qa, ka, va = (t.detach().clone().requires_grad_(True) for t in (q.contiguous(), k.contiguous(), v.contiguous()))
    oa = ops.flash_attn_varlen(qa, ka, va, cu, tiles, max(lens), scale)
    go = torch.randn_like(oa)
    oa.backward(go)
    qr, kr, vr = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    ref_attention(qr, kr, vr, lens, scale)[0].backward(go.float())
    for a, b in ((qa.grad, qr.grad), (ka.grad, kr.grad), (va.grad, vr.grad)):
        assert (a.float() - b).abs().max() < 0.03 * max(1.0, b.abs().max().item())
    # hand-written backward (delta + dQ + dK/dV kernels), on the strided views as well
    qb, kb, vb = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
    ob = ops.flash_attn_varlen(qb, kb, vb, cu, tiles, max(lens), scale, k_tiles=ops.attn_key_tile_table(lens, DEV, nkv))
    ob.backward(go)
    for name, a, b in (("dq", qb.grad, qr.grad), ("dk", kb.grad, kr.grad), ("dv", vb.grad, vr.grad)):
        err = (a.float() - b).abs().max().item()
        assert err < 0.03 * max(1.0, b.abs().max().item()), (name, err, b.abs().max().item())
    # against PyTorch's own flash attention (same bf16 inputs): both are bf16-accurate, so they agree closely
    po = torch.ops.aten._flash_attention_forward(q.contiguous(), k.contiguous(), v.contiguous(), cu, cu, max(lens),
                                                 max(lens), 0.0, True, False, scale=scale)[0]
    assert (out.float() - po.float()).abs().max() < 2.5e-2


def test_flash_attn_fwd_speed_report():
    """Not a pass/fail performance gate: prints the rate next to AOTriton's varlen kernel on the cfg-2 passage shape."""
    from rankpo_amd import ops
    torch.manual_seed(0)
    nh, nkv, hd, N, L = 32, 8, 64, 48, 4096
    lens = torch.randint(L // 2, L + 1, (N,)); lens[0] = L
    lens = lens.tolist(); T = sum(lens)
    q = torch.randn(T, nh, hd, device=DEV).to(torch.bfloat16)
    k = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
    v = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    tiles = ops.attn_tile_table(lens, DEV)
    scale = 0.125

    def bench(fn, n=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
    ours = bench(lambda: ops.flash_attn_varlen_fwd(q, k, v, cu, tiles, scale))
    theirs = bench(lambda: torch.ops.aten._flash_attention_forward(q, k, v, cu, cu, max(lens), max(lens), 0.0, True, False))
    fl = sum(4 * nh * hd * n * n / 2 for n in lens)
    print(f"\nflash fwd cfg-2 passages: HIP {ours:.2f} ms = {fl / ours / 1e9:.0f} TFLOP/s ; AOTriton {theirs:.2f} ms = {fl / theirs / 1e9:.0f} TFLOP/s")
    a = ops.flash_attn_varlen_fwd(q, k, v, cu, tiles, scale)[0]
    b = torch.ops.aten._flash_attention_forward(q, k, v, cu, cu, max(lens), max(lens), 0.0, True, False)[0]
    assert (a.float() - b.float()).abs().max() < 2.5e-2
    # backward: HIP (delta + dQ + dK/dV) vs PyTorch's op
    out, lse = ops.flash_attn_varlen_fwd(q, k, v, cu, tiles, scale)
    go = torch.randn_like(out)
    kt = ops.attn_key_tile_table(lens, DEV, nkv)
    ours_b = bench(lambda: ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tiles, kt, scale), 5)
    r = torch.ops.aten._flash_attention_forward(q, k, v, cu, cu, max(lens), max(lens), 0.0, True, False)
    theirs_b = bench(lambda: torch.ops.aten._flash_attention_backward(go, q, k, v, r[0], r[1], cu, cu, max(lens), max(lens), 0.0, True, r[2], r[3]), 5)
    print(f"flash bwd cfg-2 passages: HIP {ours_b:.2f} ms = {2.5 * fl / ours_b / 1e9:.0f} TFLOP/s ; AOTriton {theirs_b:.2f} ms = {2.5 * fl / theirs_b / 1e9:.0f} TFLOP/s")
    d1 = ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tiles, kt, scale)
    d2 = torch.ops.aten._flash_attention_backward(go, q, k, v, r[0], r[1], cu, cu, max(lens), max(lens), 0.0, True, r[2], r[3])
    for x, y in zip(d1, d2):
        assert (x.float() - y.float()).abs().max() < 0.03 * max(1.0, y.float().abs().max().item())


@pytest.mark.parametrize("hd", [64, 128])
def test_flash_attn_qkv_fused_buffer_matches_split_views(hd):
    """flash_attn_varlen_qkv (reads q|k|v as column blocks of one projection output, writes ONE d(q|k|v) buffer) must give
    bit-identical results to flash_attn_varlen on separate tensors: same kernels, different strides."""
    from rankpo_amd import ops
    torch.manual_seed(3)
    nh, nkv = 8, 2
    kb = ops.ATTN_KEY_BLOCK if hd == 64 else ops.ATTN_KEY_BLOCK_HD128
    lens = [1, 63, 64, 65, 200, 129, 333]
    T = sum(lens)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    tiles = ops.attn_tile_table(lens, DEV)
    kt = ops.attn_key_tile_table(lens, DEV, nkv, block_n=kb)
    qkv = torch.randn(T, (nh + 2 * nkv) * hd, device=DEV).to(torch.bfloat16).requires_grad_(True)
    out = ops.flash_attn_varlen_qkv(qkv, nh, nkv, cu, tiles, kt, 0.125, head_dim=hd)
    go = torch.randn_like(out)
    out.backward(go)
    q, k, v = (t.detach().clone().contiguous().requires_grad_(True) for t in
               (qkv[:, :nh * hd].view(T, nh, hd), qkv[:, nh * hd:(nh + nkv) * hd].view(T, nkv, hd),
                qkv[:, (nh + nkv) * hd:].view(T, nkv, hd)))
    ref = ops.flash_attn_varlen(q, k, v, cu, tiles, max(lens), 0.125, k_tiles=kt, key_block=kb)
    ref.backward(go)
    assert torch.equal(out, ref)
    dref = torch.cat([q.grad.reshape(T, -1), k.grad.reshape(T, -1), v.grad.reshape(T, -1)], 1)
    assert torch.equal(qkv.grad, dref)


@pytest.mark.parametrize("hd", [64, 128])
def test_rope_folded_into_attention_backward(hd):
    """ops.rope_flash_attn_varlen_qkv (one autograd node: rotary pass on the k heads, q rotated in place by the attention forward
    block that loads it (rpo_flash_attn_fwd's rope_cos / rope_sin); backward = attention backward with the inverse rotation in
    the dQ / dK epilogues, rpo_flash_attn_bwd's rope_cos / rope_sin) against the two separate nodes
    (ops.rope_ + ops.flash_attn_varlen_qkv): same output bit for bit, d(q|k|v) equal to bf16 round-off and no further from the
    float32 reference (it has one rounding less); the v columns are bit-identical; the 64-key kernel refuses the tables."""
    from rankpo_amd import ops
    from rankpo_amd._lib import RankPOHipError
    torch.manual_seed(17 + hd)
    nh, nkv = 8, 2
    lens = [1, 63, 64, 65, 200, 129, 333, 31]
    T, W = sum(lens), (nh + 2 * nkv) * hd
    scale = hd ** -0.5
    kb = ops.ATTN_KEY_BLOCK if hd == 64 else ops.ATTN_KEY_BLOCK_HD128
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    tiles = ops.attn_tile_table(lens, DEV, nh, nkv)
    kt = ops.attn_key_tile_table(lens, DEV, nkv, block_n=kb)
    ang = torch.rand(T, hd // 2, device=DEV) * 6.283
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    leaf = torch.randn(T, W, device=DEV).to(torch.bfloat16).requires_grad_(True)
    go = torch.randn(T, nh, hd, device=DEV).to(torch.bfloat16)

    def run(fused):
        leaf.grad = None
        x = leaf * 1.0                                      # the fresh projection output (a non-leaf the ops may rotate in place)
        if fused:
            out = ops.rope_flash_attn_varlen_qkv(x, cos, sin, nh, nkv, cu, tiles, kt, scale, head_dim=hd)
        else:
            out = ops.flash_attn_varlen_qkv(ops.rope_(x, cos, sin, nh + nkv, hd, grad_inplace=True), nh, nkv, cu, tiles, kt,
                                            scale, head_dim=hd)
        out.backward(go)
        return out.detach(), leaf.grad.clone()
    o_sep, g_sep = run(False)
    o_fus, g_fus = run(True)
    assert torch.equal(o_sep, o_fus)
    # the forward's own piece: q rotated IN PLACE by the block that owns it = what the rotary kernel writes, bit for bit
    x_ref = ops.rope_((leaf * 1.0).detach(), cos, sin, nh + nkv, hd)
    x_own = x_ref.clone()
    x_own[:, :nh * hd] = leaf.detach()[:, :nh * hd]                                   # q columns back to un-rotated, k stays rotated
    vw = lambda t: (t[:, :nh * hd].unflatten(1, (nh, hd)), t[:, nh * hd:(nh + nkv) * hd].unflatten(1, (nkv, hd)),
                    t[:, (nh + nkv) * hd:].unflatten(1, (nkv, hd)))
    o_own, _ = ops.flash_attn_varlen_fwd(*vw(x_own), cu, tiles, scale, rope=(cos, sin))
    assert torch.equal(x_own, x_ref) and torch.equal(o_own, o_sep)
    nq = (nh + nkv) * hd
    assert torch.equal(g_sep[:, nq:], g_fus[:, nq:])                                  # dV: untouched by the rotary
    assert (g_sep.float() - g_fus.float()).abs().max() <= 2.0 ** -6 * g_sep.float().abs().max()
    # float32 reference: rotary (HF layout) + attention through autograd
    xr = leaf.detach().float().requires_grad_(True)
    q, k, v = xr[:, :nh * hd].view(T, nh, hd), xr[:, nh * hd:nq].view(T, nkv, hd), xr[:, nq:].view(T, nkv, hd)
    c2, s2 = torch.cat([cos, cos], -1)[:, None], torch.cat([sin, sin], -1)[:, None]
    rot = lambda t: torch.cat([-t[..., hd // 2:], t[..., :hd // 2]], -1)
    ro, _ = ref_attention(q * c2 + rot(q) * s2, k * c2 + rot(k) * s2, v, lens, scale)
    ro.backward(go.float())
    e_sep = ((g_sep.float() - xr.grad)[:, :nq].norm() / xr.grad[:, :nq].norm()).item()
    e_fus = ((g_fus.float() - xr.grad)[:, :nq].norm() / xr.grad[:, :nq].norm()).item()
    print(f"\nhead_dim {hd}: d(q|k) relative L2 error vs f32: separate rotary pass {e_sep:.5f}, folded {e_fus:.5f}")
    assert e_fus <= 1.02 * e_sep + 1e-5 and e_fus < 0.01
    if hd == 64:
        x = (leaf * 1.0).detach()
        q, k, v = x[:, :nh * hd].view(T, nh, hd), x[:, nh * hd:nq].view(T, nkv, hd), x[:, nq:].view(T, nkv, hd)
        out, lse = ops.flash_attn_varlen_fwd(q, k, v, cu, tiles, scale)
        with pytest.raises(RankPOHipError, match="status -2"):
            ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tiles, ops.attn_key_tile_table(lens, DEV, nkv, block_n=64),
                                      scale, key_block=64, rope=(cos, sin))
        with pytest.raises(ValueError):
            ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tiles, kt, scale, rope=(cos.double(), sin.double()))


@pytest.mark.parametrize("hd", [64, 128])
def test_flash_attn_bwd_random_shapes_deterministic(hd):
    """Random (sequence count, lengths 1..2600, heads, GQA ratio): the hand-written backward (delta + dQ + one-wave-per-SIMD
    dK/dV with its hand-placed slice body and its masked fallback path; head_dim 128: fa_bwd_dq128 / fa_bwd_dkdv128) is
    bit-reproducible run to run and agrees with PyTorch's flash-attention backward on the same bf16 inputs."""
    from rankpo_amd import ops
    rs = np.random.RandomState(123 + hd)
    scale = hd ** -0.5
    kb = ops.ATTN_KEY_BLOCK if hd == 64 else ops.ATTN_KEY_BLOCK_HD128
    for ci in range(24 if hd == 64 else 16):
        nkv = int(rs.choice([1, 2, 4, 8]))
        nh = nkv * int(rs.choice([1, 2, 4]))
        N = int(rs.randint(1, 20))
        hi = int(rs.choice([40, 300, 700, 1500, 2600]))
        lens = [int(x) for x in rs.randint(1, hi + 1, size=N)]
        T = sum(lens)
        torch.manual_seed(ci)
        q = torch.randn(T, nh, hd, device=DEV).to(torch.bfloat16)
        k = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
        v = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
        cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
        tiles = ops.attn_tile_table(lens, DEV) if ci % 2 else ops.attn_tile_table(lens, DEV, nh, nkv)
        kt = ops.attn_key_tile_table(lens, DEV, nkv, block_n=kb)
        out, lse = ops.flash_attn_varlen_fwd(q, k, v, cu, tiles, scale)
        go = torch.randn_like(out)
        a = ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tiles, kt, scale, key_block=kb)
        b = ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tiles, kt, scale, key_block=kb)
        assert all(torch.equal(x, y) for x, y in zip(a, b)), (ci, lens)
        r = torch.ops.aten._flash_attention_forward(q, k, v, cu, cu, max(lens), max(lens), 0.0, True, False, scale=scale)
        d = torch.ops.aten._flash_attention_backward(go, q, k, v, r[0], r[1], cu, cu, max(lens), max(lens), 0.0, True, r[2],
                                                     r[3], scale=scale)
        for name, x, y in zip(("dq", "dk", "dv"), a, d):
            assert torch.isfinite(x.float()).all(), (name, ci, lens)
            err = (x.float() - y.float()).abs().max().item() / max(1.0, y.float().abs().max().item())
            assert err < 0.02, (name, err, ci, nh, nkv, lens)


def test_key_block_is_an_abi_argument():
    """rpo_flash_attn_bwd's `key_block` says what the k_tiles entries mean (include/rankpo_hip.h): 256 -> one-wave-per-SIMD
    dK/dV kernel, 64 -> the 8-wave kernel; both agree with each other to bf16 round-off, and any other value is refused
    instead of consuming a table with the wrong block size (round 1 read this from an environment variable in two places)."""
    from rankpo_amd import ops
    from rankpo_amd._lib import RankPOHipError
    torch.manual_seed(11)
    nh, nkv = 8, 2
    lens = [700, 5, 256, 257, 1100]
    T = sum(lens)
    q = torch.randn(T, nh, 64, device=DEV).to(torch.bfloat16)
    k = torch.randn(T, nkv, 64, device=DEV).to(torch.bfloat16)
    v = torch.randn(T, nkv, 64, device=DEV).to(torch.bfloat16)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    tiles = ops.attn_tile_table(lens, DEV)
    out, lse = ops.flash_attn_varlen_fwd(q, k, v, cu, tiles, 0.125)
    go = torch.randn_like(out)
    res = {}
    for kb in (256, 64):
        kt = ops.attn_key_tile_table(lens, DEV, nkv, block_n=kb)
        res[kb] = ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tiles, kt, 0.125, key_block=kb)
    assert torch.equal(res[256][0], res[64][0])                       # dQ: same kernel either way
    for x, y in zip(res[256][1:], res[64][1:]):
        assert (x.float() - y.float()).abs().max() <= 2.0 ** -6 * max(1.0, y.float().abs().max().item())
    with pytest.raises(RankPOHipError, match="status -2"):           # 128-key blocks belong to head_dim 128 only
        ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tiles, ops.attn_key_tile_table(lens, DEV, nkv), 0.125,
                                  key_block=128)
    with pytest.raises(ValueError):
        ops.attn_key_tile_table(lens, DEV, nkv, block_n=32)


def test_xcd_dealt_tile_list_is_only_a_schedule():
    """The [n, 3] XCD-dealt query-tile list (what the encoder passes) and the [n, 2] list + grid.y = heads run the same
    blocks in a different order: outputs, lse and all three gradients are bit-identical; a malformed list is refused."""
    from rankpo_amd import ops
    from rankpo_amd._lib import RankPOHipError
    rs = np.random.RandomState(7)
    for nh, nkv, lens in ((8, 2, [1, 63, 64, 65, 200, 129, 333, 700]), (32, 8, [int(x) for x in rs.randint(1, 1500, size=11)]),
                          (4, 1, [513])):
        T = sum(lens)
        torch.manual_seed(T)
        q = torch.randn(T, nh, 64, device=DEV).to(torch.bfloat16)
        k = torch.randn(T, nkv, 64, device=DEV).to(torch.bfloat16)
        v = torch.randn(T, nkv, 64, device=DEV).to(torch.bfloat16)
        cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
        kt = ops.attn_key_tile_table(lens, DEV, nkv)
        t2, t3 = ops.attn_tile_table(lens, DEV), ops.attn_tile_table(lens, DEV, nh, nkv)
        o2, l2 = ops.flash_attn_varlen_fwd(q, k, v, cu, t2, 0.125)
        o3, l3 = ops.flash_attn_varlen_fwd(q, k, v, cu, t3, 0.125)
        assert torch.equal(o2, o3) and torch.equal(l2, l3)
        go = torch.randn_like(o2)
        g2 = ops.flash_attn_varlen_bwd(q, k, v, o2, go, l2, cu, t2, kt, 0.125)
        g3 = ops.flash_attn_varlen_bwd(q, k, v, o2, go, l2, cu, t3, kt, 0.125)
        assert all(torch.equal(a, b) for a, b in zip(g2, g3))
    with pytest.raises(RankPOHipError, match="status -2"):
        ops.flash_attn_varlen_fwd(q, k, v, cu, t3[:-1], 0.125)          # a 3-column list must have a multiple of 8 entries


@pytest.mark.parametrize("lens,nh,nkv,fused", [
    ([128], 2, 2, False), ([64, 1, 200, 129, 33, 31, 32], 8, 2, False), ([300, 17, 513, 128, 256, 5], 8, 2, True),
    ([1000, 777], 4, 1, True)])
def test_flash_attn_fwd_head_dim_128(lens, nh, nkv, fused):
    """head_dim 128 (Llama-3-8B architecture, BASELINE configs[4]): forward kernel with 32-key tiles and 256-byte LDS rows
    against the f32 reference (output, lse), the padded-lse layout, PyTorch's own flash attention, and the autograd path
    (HIP forward + PyTorch's flash-attention backward on the saved output / lse)."""
    from rankpo_amd import ops
    torch.manual_seed(sum(lens) + 1)
    T, hd = sum(lens), 128
    if fused:
        qkv = torch.randn(T, (nh + 2 * nkv) * hd, device=DEV).to(torch.bfloat16)
        q, k, v = qkv.split([nh * hd, nkv * hd, nkv * hd], -1)
        q, k, v = q.view(T, nh, hd), k.view(T, nkv, hd), v.view(T, nkv, hd)
    else:
        q = torch.randn(T, nh, hd, device=DEV).to(torch.bfloat16)
        k = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
        v = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    scale = 1.0 / math.sqrt(hd)
    ro, rl = ref_attention(q, k, v, lens, scale)
    for tiles in (ops.attn_tile_table(lens, DEV), ops.attn_tile_table(lens, DEV, nh, nkv)):
        out, lse = ops.flash_attn_varlen_fwd(q, k, v, cu, tiles, scale)
        assert (out.float() - ro).abs().max() < 2.5e-2
        assert (lse - rl).abs().max() < 2e-3
    _, lp = ops.flash_attn_varlen_fwd(q, k, v, cu, tiles, scale, padded_lse_len=max(lens), num_seqs=len(lens))
    o0 = 0
    for i, n in enumerate(lens):
        assert torch.equal(lp[i, :, :n], lse[:, o0:o0 + n])
        o0 += n
    po = torch.ops.aten._flash_attention_forward(q.contiguous(), k.contiguous(), v.contiguous(), cu, cu, max(lens),
                                                 max(lens), 0.0, True, False, scale=scale)[0]
    assert (out.float() - po.float()).abs().max() < 2.5e-2
    qa, ka, va = (t.detach().clone().requires_grad_(True) for t in (q.contiguous(), k.contiguous(), v.contiguous()))
    oa = ops.flash_attn_varlen(qa, ka, va, cu, tiles, max(lens), scale)
    go = torch.randn_like(oa)
    oa.backward(go)
    qr, kr, vr = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    ref_attention(qr, kr, vr, lens, scale)[0].backward(go.float())
    for a, b in ((qa.grad, qr.grad), (ka.grad, kr.grad), (va.grad, vr.grad)):
        assert (a.float() - b).abs().max() < 0.03 * max(1.0, b.abs().max().item())
    # hand-written head_dim-128 backward (fa_bwd_dq128_kernel + fa_bwd_dkdv128_kernel, 128-key blocks), on the strided views,
    # with both query-tile list formats: same tolerance as the head_dim-64 kernels, and no worse than PyTorch's op by more
    # than bf16 noise
    kt = ops.attn_key_tile_table(lens, DEV, nkv, block_n=ops.ATTN_KEY_BLOCK_HD128)
    for tl in (ops.attn_tile_table(lens, DEV), tiles):
        qb, kb, vb = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
        ob = ops.flash_attn_varlen(qb, kb, vb, cu, tl, max(lens), scale, k_tiles=kt, key_block=ops.ATTN_KEY_BLOCK_HD128)
        assert torch.equal(ob, out)
        ob.backward(go)
        for name, a, b, c in (("dq", qb.grad, qr.grad, qa.grad), ("dk", kb.grad, kr.grad, ka.grad), ("dv", vb.grad, vr.grad, va.grad)):
            err = (a.float() - b).abs().max().item()
            assert err < 0.03 * max(1.0, b.abs().max().item()), (name, err, b.abs().max().item())
            rel = ((a.float() - b).norm() / b.norm()).item()
            rel_pt = ((c.float() - b).norm() / b.norm()).item()
            assert rel <= 1.5 * rel_pt + 1e-4, (name, rel, rel_pt)
    from rankpo_amd._lib import RankPOHipError
    with pytest.raises(RankPOHipError, match="status -2"):           # a 256-key table is not what the head_dim-128 kernel reads
        ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tiles, ops.attn_key_tile_table(lens, DEV, nkv), scale, key_block=256)


def _need_onewave64(hd=64):
    """The head_dim-64 one-wave-per-SIMD kernels are an opt-in of the build (`make ONEWAVE64=1`: they lose to the default kernels,
    profiles/r05_fa_fwd128w_ladder.md); the default library answers RPO_ERR_UNSUPPORTED there, which is checked where it is skipped."""
    from rankpo_amd import _lib
    if hd == 64 and not (_lib.load().rpo_build_flags() & _lib.RPO_BUILD_ONEWAVE64):
        pytest.skip("librankpo_hip.so was built without ONEWAVE64=1")


@pytest.mark.parametrize("hd", [128, 64])
@pytest.mark.parametrize("lens,nh,nkv,fused", [
    ([128], 4, 1, False), ([1], 4, 1, False), ([64, 1, 200, 129, 33, 31, 32, 65, 63], 8, 2, False),
    ([300, 17, 513, 128, 256, 5], 8, 1, True), ([1000, 777], 4, 1, True), ([4096, 2500], 32, 8, True)])
def test_flash_attn_fwd_one_wave(lens, nh, nkv, fused, hd):
    """fa_fwd128w_kernel / fa_fwd64w_kernel (q_block = 64: entries of 64 queries x the 4 q heads of a group, one wave per SIMD,
    deferred softmax scale, Q in / O out through LDS as whole rows) against the f32 reference and against the 128-query kernel,
    on both list formats, the padded-lse layout, strided (fused-projection) views, the rotary fold, and through autograd with the
    hand-written backward (which walks the 128-row list)."""
    _need_onewave64(hd)
    from rankpo_amd import ops
    torch.manual_seed(sum(lens) + 7 + hd)
    T = sum(lens)
    if fused:
        qkv = torch.randn(T, (nh + 2 * nkv) * hd, device=DEV).to(torch.bfloat16)
        q, k, v = qkv.split([nh * hd, nkv * hd, nkv * hd], -1)
        q, k, v = q.view(T, nh, hd), k.view(T, nkv, hd), v.view(T, nkv, hd)
    else:
        q = torch.randn(T, nh, hd, device=DEV).to(torch.bfloat16)
        k = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
        v = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    scale = 1.0 / math.sqrt(hd)
    ro, rl = ref_attention(q, k, v, lens, scale)
    t128 = ops.attn_tile_table(lens, DEV, nh, nkv)
    oc, lc = ops.flash_attn_varlen_fwd(q, k, v, cu, t128, scale)                       # the 128-query kernel
    ft = ops.attn_fwd_tile_table(lens, DEV, nh, nkv, hd, force=True)
    assert ft is not None and ft.shape[1] == 3 and ft.shape[0] % 8 == 0
    outs = []
    for tiles in (ops.attn_tile_table(lens, DEV, block_m=64), ft):                     # format 2 (grid.y = head groups), format 3
        out, lse = ops.flash_attn_varlen_fwd(q, k, v, cu, tiles, scale, q_block=64)
        assert (out.float() - ro).abs().max() < 2.5e-2
        assert (lse - rl).abs().max() < 2e-3
        assert (out.float() - oc.float()).abs().max() <= 2.0 ** -6 and (lse - lc).abs().max() < 3e-3
        outs.append((out, lse))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])   # the list is only a schedule
    out, lse = outs[1]
    _, lp = ops.flash_attn_varlen_fwd(q, k, v, cu, ft, scale, padded_lse_len=max(lens), num_seqs=len(lens), q_block=64)
    o0 = 0
    for i, n in enumerate(lens):
        assert torch.equal(lp[i, :, :n], lse[:, o0:o0 + n])
        o0 += n
    # rotary fold: q rotated in place exactly as rpo_rope does it; the output follows the 128-query kernel's
    ang = torch.rand(T, hd // 2, device=DEV) * 6.283
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    qa, qb = q.contiguous().clone(), q.contiguous().clone()
    o_a, l_a = ops.flash_attn_varlen_fwd(qa, k, v, cu, t128, scale, rope=(cos, sin))
    o_b, l_b = ops.flash_attn_varlen_fwd(qb, k, v, cu, ft, scale, rope=(cos, sin), q_block=64)
    assert torch.equal(qa, qb) and not torch.equal(qa, q.contiguous())
    assert (o_a.float() - o_b.float()).abs().max() <= 2.0 ** -6 and (l_a - l_b).abs().max() < 3e-3
    # autograd: one-wave forward + the hand-written backward on its own 128-row list
    kb = ops.ATTN_KEY_BLOCK_HD128 if hd == 128 else ops.ATTN_KEY_BLOCK
    kt = ops.attn_key_tile_table(lens, DEV, nkv, block_n=kb)
    go = torch.randn(T, nh, hd, device=DEV).to(torch.bfloat16)
    qr, kr, vr = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    ref_attention(qr, kr, vr, lens, scale)[0].backward(go.float())
    qg, kg, vg = (t.detach().clone().requires_grad_(True) for t in (q, k, v))
    og = ops.flash_attn_varlen(qg, kg, vg, cu, t128, max(lens), scale, k_tiles=kt, key_block=kb, fwd_tiles=ft)
    assert torch.equal(og, out)
    og.backward(go)
    for name, a, b in (("dq", qg.grad, qr.grad), ("dk", kg.grad, kr.grad), ("dv", vg.grad, vr.grad)):
        assert (a.float() - b).abs().max() < 0.03 * max(1.0, b.abs().max().item()), name


@pytest.mark.parametrize("hd", [128, 64])
def test_flash_attn_fwd_one_wave_rescales_and_refusals(hd):
    """The deferred scale: logits that keep growing along the sequence (every few tiles some row outgrows 2^8 times its scale: the
    RESCALE statements of both score generations run, many times) and logits far below zero (the scale starts at tile 0's row
    maximum, not at 0) still match the f32 reference; what the kernel is not built for is refused, not mis-run."""
    _need_onewave64(hd)
    from rankpo_amd import ops
    from rankpo_amd._lib import RankPOHipError
    torch.manual_seed(5)
    nh, nkv = 8, 2
    lens = [1500, 700, 96]
    T = sum(lens)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    scale = 1.0 / math.sqrt(hd)
    ft = ops.attn_fwd_tile_table(lens, DEV, nh, nkv, hd, force=True)
    pos = torch.cat([torch.arange(n, dtype=torch.float32) for n in lens]).to(DEV)
    base = torch.randn(T, nkv, hd, device=DEV)
    q = (torch.randn(T, nh, hd, device=DEV) * 2.0 + 1.5).to(torch.bfloat16)            # a common component: <q, k> grows with |k|
    v = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
    for name, k in (("growing", ((base * 0.3 + 0.4) * (1.0 + pos / 60.0)[:, None, None]).to(torch.bfloat16)),
                    ("negative", (base * 0.2 - 3.0).to(torch.bfloat16))):
        ro, rl = ref_attention(q, k, v, lens, scale)
        assert (rl.max() - rl.min()).item() > 50 or rl.max().item() < -5, name          # the logits really leave the easy range
        out, lse = ops.flash_attn_varlen_fwd(q, k, v, cu, ft, scale, q_block=64)
        assert (out.float() - ro).abs().max() < 2.5e-2, name
        assert ((lse - rl).abs() / rl.abs().clamp(min=1.0)).max() < 2e-3, name
    # refusals
    assert ops.attn_fwd_tile_table(lens, DEV, 8, 4, hd, force=True) is None and ops.attn_fwd_tile_table(lens, DEV, 8, 2, 96, force=True) is None
    k = base.to(torch.bfloat16)
    with pytest.raises(RankPOHipError, match="status -2"):                               # 2 q heads per kv head
        ops.flash_attn_varlen_fwd(q, torch.randn(T, 4, hd, device=DEV).to(torch.bfloat16),
                                  torch.randn(T, 4, hd, device=DEV).to(torch.bfloat16), cu, ft, scale, q_block=64)
    with pytest.raises(RankPOHipError, match="status -2"):
        ops.flash_attn_varlen_fwd(q, k, v, cu, ft, scale, q_block=32)


@pytest.mark.parametrize("lens,nh,nkv,fused", [
    ([128], 4, 1, False), ([1], 4, 1, False), ([64, 1, 200, 129, 33, 31, 32, 65, 63], 8, 2, False),
    ([300, 17, 513, 128, 256, 5], 8, 1, True), ([1000, 777], 4, 1, True), ([4096, 2500], 32, 8, True)])
def test_flash_attn_dq_one_wave_head_dim_64(lens, nh, nkv, fused):
    """fa_bwd_dq64w_kernel (rpo_flash_attn_bwd's q_block = 64: 64 queries x 4 q heads per block, one wave per SIMD, generated
    statements) against fa_bwd_dq_kernel and the f32 reference: dq, and -- through the row constants it writes for the dK/dV
    kernel -- dk and dv; both list formats; strided views; the rotary epilogue."""
    _need_onewave64()
    from rankpo_amd import ops
    torch.manual_seed(sum(lens) + 11)
    T, hd = sum(lens), 64
    if fused:
        qkv = torch.randn(T, (nh + 2 * nkv) * hd, device=DEV).to(torch.bfloat16)
        q, k, v = qkv.split([nh * hd, nkv * hd, nkv * hd], -1)
        q, k, v = q.view(T, nh, hd), k.view(T, nkv, hd), v.view(T, nkv, hd)
    else:
        q = torch.randn(T, nh, hd, device=DEV).to(torch.bfloat16)
        k = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
        v = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    scale = 1.0 / math.sqrt(hd)
    t128 = ops.attn_tile_table(lens, DEV, nh, nkv)
    kt = ops.attn_key_tile_table(lens, DEV, nkv, block_n=ops.ATTN_KEY_BLOCK)
    out, lse = ops.flash_attn_varlen_fwd(q, k, v, cu, t128, scale)
    go = torch.randn(T, nh, hd, device=DEV).to(torch.bfloat16)
    qr, kr, vr = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    ref_attention(qr, kr, vr, lens, scale)[0].backward(go.float())
    g_old = ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, t128, kt, scale)
    res = []
    for tl in (ops.attn_tile_table(lens, DEV, block_m=64), ops.attn_tile_table(lens, DEV, nh, nkv, block_m=64, heads_per_block=4)):
        g_new = ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tl, kt, scale, q_block=64)
        for name, a, b, c in zip(("dq", "dk", "dv"), g_new, (qr.grad, kr.grad, vr.grad), g_old):
            assert (a.float() - b).abs().max() < 0.03 * max(1.0, b.abs().max().item()), name
            rel, rel_old = ((a.float() - b).norm() / b.norm()).item(), ((c.float() - b).norm() / b.norm()).item()
            assert rel <= 1.3 * rel_old + 1e-4, (name, rel, rel_old)
        assert torch.equal(g_new[1], g_old[1]) and torch.equal(g_new[2], g_old[2])      # the row constants are the old kernel's, bit for bit
        res.append(g_new[0])
    assert torch.equal(res[0], res[1])                                                   # the list is only a schedule
    ang = torch.rand(T, hd // 2, device=DEV) * 6.283
    rope = (ang.cos().contiguous(), ang.sin().contiguous())
    a = ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, t128, kt, scale, rope=rope)
    b = ops.flash_attn_varlen_bwd(q, k, v, out, go, lse, cu, tl, kt, scale, rope=rope, q_block=64)
    assert (a[0].float() - b[0].float()).abs().max() <= 2.0 ** -5 * max(1.0, a[0].float().abs().max().item())
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])


@pytest.mark.parametrize("hd", [128, 64])
def test_one_wave_kernels_on_a_long_sequence(hd):
    """16 k tokens in one sequence (512 key tiles: the K / V rings of the one-wave kernels wrap 64-128 times, the eight statement
    variants of the dQ kernel and the four of the forward come round many times) beside a short one, against the 128-query
    kernels -- forward at both head dims, and the head_dim-64 dQ kernel."""
    _need_onewave64(hd)
    from rankpo_amd import ops
    torch.manual_seed(3 + hd)
    lens, nh, nkv = [16384 + 37, 70], 4, 1
    T = sum(lens)
    q = torch.randn(T, nh, hd, device=DEV).to(torch.bfloat16)
    k = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
    v = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    scale = 1.0 / math.sqrt(hd)
    t128 = ops.attn_tile_table(lens, DEV, nh, nkv)
    t64 = ops.attn_fwd_tile_table(lens, DEV, nh, nkv, hd, force=True)
    o_a, l_a = ops.flash_attn_varlen_fwd(q, k, v, cu, t128, scale)
    o_b, l_b = ops.flash_attn_varlen_fwd(q, k, v, cu, t64, scale, q_block=64)
    assert (o_a.float() - o_b.float()).abs().max() <= 2.0 ** -6 and (l_a - l_b).abs().max() < 3e-3
    if hd == 64:
        kt = ops.attn_key_tile_table(lens, DEV, nkv, block_n=ops.ATTN_KEY_BLOCK)
        go = torch.randn(T, nh, hd, device=DEV).to(torch.bfloat16)
        g_a = ops.flash_attn_varlen_bwd(q, k, v, o_a, go, l_a, cu, t128, kt, scale)
        g_b = ops.flash_attn_varlen_bwd(q, k, v, o_a, go, l_a, cu, t64, kt, scale, q_block=64)
        assert (g_a[0].float() - g_b[0].float()).abs().max() <= 2.0 ** -6 * max(1.0, g_a[0].float().abs().max().item())
        assert torch.equal(g_a[1], g_b[1]) and torch.equal(g_a[2], g_b[2])


def test_flash_attn_bwd128_speed_report():
    """Prints the head_dim-128 backward rate next to PyTorch's flash-attention backward op on a cfg-5-like passage batch (not a gate)."""
    from rankpo_amd import ops
    torch.manual_seed(0)
    nh, nkv, hd, N, L = 32, 8, 128, 24, 4096
    lens = torch.randint(L // 2, L + 1, (N,)); lens[0] = L
    lens = lens.tolist(); T = sum(lens)
    mk = lambda h: torch.randn(T, h, hd, device=DEV).to(torch.bfloat16).requires_grad_()
    q, k, v = mk(nh), mk(nkv), mk(nkv)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    tiles = ops.attn_tile_table(lens, DEV, nh, nkv)
    kt = ops.attn_key_tile_table(lens, DEV, nkv, block_n=ops.ATTN_KEY_BLOCK_HD128)
    go = torch.randn(T, nh, hd, device=DEV).to(torch.bfloat16)
    fl = sum(10 * nh * hd * n * (n + 1) / 2 for n in lens)
    res = {}
    for name, ktab in (("HIP", kt), ("PyTorch op", None)):
        out = ops.flash_attn_varlen(q, k, v, cu, tiles, max(lens), hd ** -0.5, k_tiles=ktab, key_block=ops.ATTN_KEY_BLOCK_HD128)
        fn = lambda: torch.autograd.grad(out, (q, k, v), go, retain_graph=True)
        for _ in range(3):
            g = fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize(); res[name] = ((time.perf_counter() - t) / 10 * 1e3, g)
    for a, b in zip(res["HIP"][1], res["PyTorch op"][1]):
        assert (a.float() - b.float()).abs().max() < 0.03 * max(1.0, b.float().abs().max().item())
    print("\nflash bwd head_dim 128, %d sequences: " % N + " ; ".join(f"{n} {ms:.2f} ms = {fl / ms / 1e9:.0f} TFLOP/s" for n, (ms, _) in res.items()))


def test_flash_attn_fwd128_speed_report():
    """Prints the head_dim-128 forward rate next to AOTriton's varlen kernel on a cfg-5-like passage batch (not a gate)."""
    from rankpo_amd import ops
    torch.manual_seed(0)
    nh, nkv, hd, N, L = 32, 8, 128, 24, 4096
    lens = torch.randint(L // 2, L + 1, (N,)); lens[0] = L
    lens = lens.tolist(); T = sum(lens)
    q = torch.randn(T, nh, hd, device=DEV).to(torch.bfloat16)
    k = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
    v = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    tiles = ops.attn_tile_table(lens, DEV, nh, nkv)
    scale = 1.0 / math.sqrt(hd)

    def bench(fn, n=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
    ours = bench(lambda: ops.flash_attn_varlen_fwd(q, k, v, cu, tiles, scale))
    theirs = bench(lambda: torch.ops.aten._flash_attention_forward(q, k, v, cu, cu, max(lens), max(lens), 0.0, True, False))
    fl = sum(4 * nh * hd * n * (n + 1) / 2 for n in lens)
    print(f"\\nflash fwd head_dim 128, {N} sequences: HIP {ours:.2f} ms = {fl / ours / 1e9:.0f} TFLOP/s ; AOTriton {theirs:.2f} ms = {fl / theirs / 1e9:.0f} TFLOP/s")
    a = ops.flash_attn_varlen_fwd(q, k, v, cu, tiles, scale)[0]
    b = torch.ops.aten._flash_attention_forward(q, k, v, cu, cu, max(lens), max(lens), 0.0, True, False)[0]
    assert (a.float() - b.float()).abs().max() < 2.5e-2


def _ref_last_query(q, k, v, lens, scale):
    """f32 reference of the one-query-per-sequence attention: q [N, nh, hd] (the last token of each sequence), packed k / v."""
    outs, lses, o0 = [], [], 0
    nh, nkv = q.shape[1], k.shape[1]
    for i, n in enumerate(lens):
        kk = k[o0:o0 + n].float().transpose(0, 1).repeat_interleave(nh // nkv, 0)      # [nh, n, hd]
        vv = v[o0:o0 + n].float().transpose(0, 1).repeat_interleave(nh // nkv, 0)
        s = torch.einsum("hd,hnd->hn", q[i].float(), kk) * scale
        lses.append(torch.logsumexp(s, -1))
        outs.append(torch.einsum("hn,hnd->hd", torch.softmax(s, -1), vv))
        o0 += n
    return torch.stack(outs), torch.stack(lses)


@pytest.mark.parametrize("hd,nh,nkv,lens", [
    (64, 32, 8, [4096, 2049, 1, 7, 31, 32, 33, 300, 1280]), (64, 4, 4, [5, 64, 129]), (64, 8, 4, [1000, 17]),
    (128, 32, 8, [4096, 1, 15, 16, 17, 513, 2050]), (128, 4, 2, [3, 300])])
def test_last_query_attention_matches_reference(hd, nh, nkv, lens):
    """rpo_lastq_attn_fwd / _bwd (the last block's attention: ONE query per sequence, all keys visible; reference: the encoder
    forward behind modeling.py:219 + the pooling of :224-230) on the fused k|v buffer the encoder hands it, against an f32
    reference on the same bf16 inputs and against PyTorch's flash-attention op, which this kernel replaced."""
    from rankpo_amd import ops
    torch.manual_seed(sum(lens) + hd)
    T, N = sum(lens), len(lens)
    scale = 1.0 / math.sqrt(hd)
    kv = torch.randn(1, T, 2 * nkv * hd, device=DEV).to(torch.bfloat16).requires_grad_(True)
    q = torch.randn(N, nh, hd, device=DEV).to(torch.bfloat16).requires_grad_(True)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    assert ops.last_query_attn_ok(q, kv, nh, nkv, hd)
    out = ops.last_query_attn(q, kv, cu, nkv, hd, scale)
    go = torch.randn_like(out)
    out.backward(go)
    k, v = kv.detach()[0].split([nkv * hd, nkv * hd], -1)
    k, v = k.reshape(T, nkv, hd), v.reshape(T, nkv, hd)
    qr, kr, vr = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    ro, rl = _ref_last_query(qr, kr, vr, lens, scale)
    ro.backward(go.float())
    assert (out.float() - ro).abs().max() < 1e-2                       # f32 softmax and P.V: only the bf16 output rounding
    dkv_ref = torch.cat([kr.grad.reshape(T, -1), vr.grad.reshape(T, -1)], -1)
    for name, a, b in (("dq", q.grad, qr.grad), ("dkv", kv.grad[0], dkv_ref)):
        err = (a.float() - b).abs().max().item()
        rel = ((a.float() - b).norm() / b.norm().clamp_min(1e-30)).item()
        assert err < 0.02 * max(1.0, b.abs().max().item()) and rel < 6e-3, (name, err, rel)
    # PyTorch's op on the same inputs (what ran here before): no further from the f32 reference than that op
    cu_q = torch.arange(N + 1, device=DEV, dtype=torch.int32)
    po = torch.ops.aten._flash_attention_forward(q.detach(), k.contiguous(), v.contiguous(), cu_q, cu, 1, max(lens), 0.0, False,
                                                 False, scale=scale)[0]
    e_ours, e_theirs = (out.float() - ro).norm().item(), (po.float() - ro).norm().item()
    assert e_ours <= 1.5 * e_theirs + 1e-6, (e_ours, e_theirs)
    # deterministic: a second run is bit-identical (fixed merge order, no atomics)
    q2, kv2 = q.detach().clone().requires_grad_(True), kv.detach().clone().requires_grad_(True)
    out2 = ops.last_query_attn(q2, kv2, cu, nkv, hd, scale)
    out2.backward(go)
    assert torch.equal(out2, out) and torch.equal(q2.grad, q.grad) and torch.equal(kv2.grad, kv.grad)


def test_last_query_attention_lse_and_speed_report():
    """lse through the C call; prints the rate next to the PyTorch op on the cfg-2 shape (56 sequences, 32 / 8 heads)."""
    from rankpo_amd import ops, _lib
    torch.manual_seed(0)
    nh, nkv, hd, N, L = 32, 8, 64, 56, 4096
    lens = torch.randint(L // 2, L + 1, (N,)); lens[0] = L
    lens = lens.tolist(); T = sum(lens)
    kv = torch.randn(T, 2 * nkv * hd, device=DEV).to(torch.bfloat16)
    q = torch.randn(N, nh, hd, device=DEV).to(torch.bfloat16)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    lib = _lib.load()
    out = torch.empty(N, nh, hd, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(N, nh, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    call = lambda: lib.rpo_lastq_attn_fwd(q.data_ptr(), nh * hd, kv.data_ptr(), kv.data_ptr() + nkv * hd * 2, 2 * nkv * hd, 2 * nkv * hd,
                                          cu.data_ptr(), N, nh, nkv, hd, 0.125, out.data_ptr(), nh * hd, lse.data_ptr(), st)
    assert call() == 0
    k, v = kv.split([nkv * hd, nkv * hd], -1)
    ro, rl = _ref_last_query(q, k.reshape(T, nkv, hd), v.reshape(T, nkv, hd), lens, 0.125)
    assert (lse - rl).abs().max() < 1e-4 and (out.float() - ro).abs().max() < 1e-2
    # unsupported group size: refused, not mis-computed
    assert lib.rpo_lastq_attn_fwd(q.data_ptr(), nh * hd, kv.data_ptr(), kv.data_ptr(), 2 * nkv * hd, 2 * nkv * hd, cu.data_ptr(), N, 32, 4,
                                  hd, 0.125, out.data_ptr(), nh * hd, lse.data_ptr(), st) == -2

    def bench(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
    kc, vc = k.reshape(T, nkv, hd).contiguous(), v.reshape(T, nkv, hd).contiguous()
    cu_q = torch.arange(N + 1, device=DEV, dtype=torch.int32)
    ours = bench(call)
    theirs = bench(lambda: torch.ops.aten._flash_attention_forward(q, kc, vc, cu_q, cu, 1, max(lens), 0.0, False, False))
    qg, kvg = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
    o = ops.last_query_attn(qg, kvg, cu, nkv, hd, 0.125)
    go = torch.randn_like(o)
    ours_b = bench(lambda: torch.autograd.grad(o, (qg, kvg), go, retain_graph=True), 10)
    nbytes = 2 * T * nkv * hd * 2
    print(f"\nlast-query attention, cfg-2 shape (T = {T}): forward HIP {ours:.0f} us = {nbytes / ours / 1e3:.0f} GB/s of K|V; PyTorch op "
          f"{theirs:.0f} us; backward HIP {ours_b:.0f} us = {2 * nbytes / ours_b / 1e3:.0f} GB/s (K|V read + dK|dV written)")


def test_default_build_refuses_the_optional_one_wave_64_kernels():
    """Without ONEWAVE64=1 the q_block = 64 entries at head_dim 64 are RPO_ERR_UNSUPPORTED (status -2), forward and backward: refused
    loudly, never mis-run; with it the tests above run them."""
    from rankpo_amd import ops, _lib
    from rankpo_amd._lib import RankPOHipError
    if _lib.load().rpo_build_flags() & _lib.RPO_BUILD_ONEWAVE64:
        pytest.skip("this library holds the one-wave 64 kernels")
    lens, nh, nkv, hd = [200, 70], 8, 2, 64
    T = sum(lens)
    q = torch.randn(T, nh, hd, device=DEV).to(torch.bfloat16)
    k = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
    v = torch.randn(T, nkv, hd, device=DEV).to(torch.bfloat16)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=DEV)
    t64 = ops.attn_fwd_tile_table(lens, DEV, nh, nkv, hd, force=True)
    with pytest.raises(RankPOHipError, match="status -2"):
        ops.flash_attn_varlen_fwd(q, k, v, cu, t64, 0.125, q_block=64)
    out, lse = ops.flash_attn_varlen_fwd(q, k, v, cu, ops.attn_tile_table(lens, DEV, nh, nkv), 0.125)
    with pytest.raises(RankPOHipError, match="status -2"):
        ops.flash_attn_varlen_bwd(q, k, v, out, torch.randn_like(out), lse, cu, t64, ops.attn_key_tile_table(lens, DEV, nkv), 0.125,
                                  q_block=64)
