"""GPU: each HIP kernel (through the C ABI) against a plain PyTorch fp32 CPU
reference of the same op / the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from conftest import load_golden, rel_err

from oracle import ref_torch as R

pytestmark = pytest.mark.gpu

F32_TOL = 2e-5
BF16_TOL = 2e-2


def rnd(seed, *shape, scale=1.0):
    return torch.from_numpy((scale * np.random.default_rng(seed).standard_normal(shape)).astype(np.float32))


CONV_CASES = [
    # B, T, Cin, Cout, ks, dil
    (2, 37, 80, 512, 7, 1),     # conv_pre shape (Cin not a multiple of 32/64)
    (3, 150, 256, 256, 17, 1),  # frame prior
    (2, 70, 256, 1024, 9, 1),   # conformer FFN w_1
    (2, 70, 1024, 256, 9, 1),   # conformer FFN w_2
    (2, 200, 256, 512, 3, 8),   # DiffNet dilated
    (1, 300, 32, 32, 11, 5),    # BigVGAN last stage
    (2, 260, 64, 64, 7, 3),
    (2, 33, 256, 4, 1, 1),      # MDN head (Linear)
    (2, 45, 256, 2, 1, 1),      # pitch out layer, Cout not /4
    (1, 129, 128, 640, 3, 1),   # upsample-as-conv geometry
    (4, 5, 256, 256, 3, 1),     # very short sequences
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv1d_fwd(case, dtype, dev):
    from promptttspp_amd import ops

    B, T, Cin, Cout, ks, dil = case
    pad = (ks - 1) * dil // 2
    x = rnd(1, B, T, Cin)
    w = rnd(2, Cout, Cin, ks) / np.sqrt(Cin * ks)
    b = rnd(3, Cout, scale=0.1)
    res = rnd(4, B, T, Cout)
    lens = torch.tensor([T, max(1, T // 2), max(1, T - 3), 1][:B], dtype=torch.int32)
    if dtype == torch.bfloat16:  # compare against the same rounded operands
        x, w, res = x.bfloat16().float(), w.bfloat16().float(), res.bfloat16().float()
    mask = (torch.arange(T)[None, :] < lens[:, None]).float()[:, :, None]
    ref = F.conv1d((x * mask).transpose(1, 2), w, b, padding=pad, dilation=dil).transpose(1, 2)
    ref = res + 0.5 * F.gelu(ref) * mask
    wp = ops.pack_conv_weight(w.to(dev), dtype)
    y = ops.conv1d(x.to(dev, dtype), wp, b.to(dev), Cout, ks=ks, dil=dil, pad=pad, act="gelu", lengths=lens,
                   in_mask=True, out_mask=True, res=res.to(dev, dtype), out_scale=0.5)
    assert rel_err(y.float().cpu(), ref) < (F32_TOL if dtype == torch.float32 else BF16_TOL)
    # no masks / no residual / no bias path
    ref2 = F.conv1d(x.transpose(1, 2), w, None, padding=pad, dilation=dil).transpose(1, 2)
    y2 = ops.conv1d(x.to(dev, dtype), wp, None, Cout, ks=ks, dil=dil, pad=pad)
    assert rel_err(y2.float().cpu(), ref2) < (F32_TOL if dtype == torch.float32 else BF16_TOL)


def test_conv1d_transpose_detecting(dev):
    """asymmetric weights + identity-like input catch row/col swaps in the MFMA
    output mapping (cdna_hip_programming.md G9)."""
    from promptttspp_amd import ops

    T, C = 64, 64
    x = torch.eye(T, C).unsqueeze(0)
    w = (torch.arange(C * C, dtype=torch.float32).reshape(C, C, 1) % 97) / 97.0
    y = ops.conv1d(x.to(dev), ops.pack_conv_weight(w.to(dev), torch.float32), None, C)
    assert torch.equal(y.cpu()[0], w[:, :, 0].t())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv1d_dgrad_via_mode1_pack(dtype, dev):
    """data gradient = the same kernel on flipped/transposed packed weights."""
    from promptttspp_amd import ops

    B, T, Cin, Cout, ks, dil = 2, 90, 256, 512, 3, 4
    pad = (ks - 1) * dil // 2
    x = rnd(1, B, T, Cin).requires_grad_()
    w = rnd(2, Cout, Cin, ks) / np.sqrt(Cin * ks)
    dy = rnd(3, B, T, Cout)
    if dtype == torch.bfloat16:
        w, dy = w.bfloat16().float(), dy.bfloat16().float()
    y = F.conv1d(x.transpose(1, 2), w, None, padding=pad, dilation=dil).transpose(1, 2)
    (dx_ref,) = torch.autograd.grad(y, x, dy)
    wpt = ops.pack_conv_weight(w.to(dev), dtype, mode=1)
    dx = ops.conv1d(dy.to(dev, dtype), wpt, None, Cin, ks=ks, dil=dil, pad=(ks - 1) * dil - pad)
    assert rel_err(dx.float().cpu(), dx_ref) < (F32_TOL if dtype == torch.float32 else BF16_TOL)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(2, 150, 256, 256, 5, 1), (3, 70, 80, 256, 1, 1), (2, 90, 256, 4, 1, 1), (2, 131, 256, 256, 17, 1), (3, 45, 256, 1024, 9, 1),
                                  (2, 64, 256, 512, 3, 8)])
def test_conv1d_wgrad(case, dtype, dev):
    from promptttspp_amd import ops

    B, T, Cin, Cout, ks, dil = case
    pad = (ks - 1) * dil // 2
    x = rnd(1, B, T, Cin)
    dy = rnd(3, B, T, Cout)
    lens = torch.tensor([T, T // 2, T - 1][:B], dtype=torch.int32)
    if dtype == torch.bfloat16:
        x, dy = x.bfloat16().float(), dy.bfloat16().float()
    mask = (torch.arange(T)[None, :] < lens[:, None]).float()[:, :, None]
    w = torch.zeros(Cout, Cin, ks, requires_grad=True)
    b = torch.zeros(Cout, requires_grad=True)
    y = F.conv1d((x * mask).transpose(1, 2), w, b, padding=pad, dilation=dil).transpose(1, 2)
    dw_ref, db_ref = torch.autograd.grad(y, (w, b), dy)
    dw, db = ops.conv1d_wgrad(x.to(dev, dtype), dy.to(dev, dtype), Cin, Cout, ks, dil, pad, lengths=lens, in_mask=True)
    assert rel_err(dw.cpu(), dw_ref) < 1e-4
    assert rel_err(db.cpu(), db_ref) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [256, 768, 80])
def test_layernorm_fwd_bwd(dtype, C, dev):
    from promptttspp_amd import ops

    B, T = 3, 41
    x = (rnd(1, B, T, C, scale=2.0) + 0.3)
    res = rnd(2, B, T, C)
    g = 1.0 + 0.1 * rnd(3, C)
    bta = 0.1 * rnd(4, C)
    dy = rnd(5, B, T, C)
    lens = torch.tensor([T, 7, 1], dtype=torch.int32)
    if dtype == torch.bfloat16:
        x, res, dy = x.bfloat16().float(), res.bfloat16().float(), dy.bfloat16().float()
    tol = 1e-5 if dtype == torch.float32 else BF16_TOL
    mask = (torch.arange(T)[None, :] < lens[:, None]).float()[:, :, None]
    xs = (x + res).requires_grad_()
    gp, bp = g.clone().requires_grad_(), bta.clone().requires_grad_()
    ref = R.layer_norm_last(xs, gp, bp, 1e-5) * mask
    dx_ref, dg_ref, db_ref = torch.autograd.grad(ref, (xs, gp, bp), dy)
    y, mean, rstd, xsum = ops.layernorm_fwd(x.to(dev, dtype), g.to(dev), bta.to(dev), 1e-5, res=res.to(dev, dtype),
                                            lengths=lens, out_mask=True, save_stats=True, save_sum=True)
    assert rel_err(y.float().cpu(), ref.detach()) < tol
    dx, _, dg, db = ops.layernorm_bwd(dy.to(dev, dtype), xsum, g.to(dev), mean, rstd, lengths=lens, out_mask=True)
    assert rel_err(dx.float().cpu(), dx_ref) < tol
    assert rel_err(dg.cpu(), dg_ref) < (1e-4 if dtype == torch.float32 else BF16_TOL)
    assert rel_err(db.cpu(), db_ref) < (1e-4 if dtype == torch.float32 else BF16_TOL)
    # eps 1e-12 (ESPnet variant), no residual, no mask
    y2, *_ = ops.layernorm_fwd(x.to(dev, dtype), g.to(dev), bta.to(dev), 1e-12)
    assert rel_err(y2.float().cpu(), R.layer_norm_last(x, g, bta, 1e-12)) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_aa_snake_golden(dtype, dev):
    from promptttspp_amd import ops

    g = load_golden("aa_snake")
    up, dn = ops._taps(g["f_up"]), ops._taps(g["f_dn"])
    for T in [1, 2, 5, 6, 7, 13, 64, 131]:
        x = g[f"x{T}"]
        if dtype == torch.bfloat16:
            x = x.bfloat16().float()
        ref = R.aa_snake(x, g["alpha"], g["f_up"], g["f_dn"]) if dtype == torch.bfloat16 else g[f"y{T}"]
        y = ops.aa_snake(x.transpose(1, 2).contiguous().to(dev, dtype), g["alpha"].to(dev), up, dn)
        assert rel_err(y.float().cpu().transpose(1, 2), ref) < (1e-5 if dtype == torch.float32 else BF16_TOL), T


def test_aa_snake_long_and_wide(dev):
    """lengths spanning several per-thread runs (R = 66) and C = 512."""
    from promptttspp_amd import ops

    g = load_golden("aa_snake")
    up, dn = ops._taps(g["f_up"]), ops._taps(g["f_dn"])
    for (B, C, T) in [(2, 512, 67), (1, 32, 1000), (3, 64, 133)]:
        x = rnd(T, B, C, T, scale=1.5)
        la = 0.3 * rnd(C, C)
        ref = R.aa_snake(x, la, g["f_up"], g["f_dn"])
        y = ops.aa_snake(x.transpose(1, 2).contiguous().to(dev), la.to(dev), up, dn)
        assert rel_err(y.cpu().transpose(1, 2), ref) < 1e-5


def test_layout_bridges(dev):
    from promptttspp_amd import ops

    x = rnd(1, 3, 80, 77)
    y = ops.bct_to_btc(x.to(dev), torch.float32)
    assert torch.equal(y.cpu(), x.transpose(1, 2))
    assert torch.equal(ops.btc_to_bct(y).cpu(), x)
    yb = ops.bct_to_btc(x.to(dev), torch.bfloat16)
    assert torch.equal(yb.cpu(), x.transpose(1, 2).bfloat16())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layernorm_gelu_residual_mode(dtype, dev):
    """y = LN(res + gelu(z)) (frame prior) forward/backward vs torch autograd."""
    from promptttspp_amd import functional as PF

    B, T, C = 2, 33, 256
    z, res, dy = rnd(1, B, T, C), rnd(2, B, T, C), rnd(3, B, T, C)
    g, bta = 1.0 + 0.1 * rnd(4, C), 0.1 * rnd(5, C)
    if dtype == torch.bfloat16:
        z, res, dy = z.bfloat16().float(), res.bfloat16().float(), dy.bfloat16().float()
    zr, rr, gr, br = z.clone().requires_grad_(), res.clone().requires_grad_(), g.clone().requires_grad_(), bta.clone().requires_grad_()
    ref = R.layer_norm_last(rr + F.gelu(zr), gr, br, 1e-5)
    gz, gres, gg, gb = torch.autograd.grad(ref, (zr, rr, gr, br), dy)
    zd, rd = z.to(dev, dtype).requires_grad_(), res.to(dev, dtype).requires_grad_()
    gd, bd = g.to(dev).requires_grad_(), bta.to(dev).requires_grad_()
    y = PF.layer_norm(zd, gd, bd, 1e-5, res=rd, act_in="gelu")
    tol = 2e-5 if dtype == torch.float32 else BF16_TOL
    assert rel_err(y.float().cpu(), ref.detach()) < tol
    y.backward(dy.to(dev, dtype))
    assert rel_err(zd.grad.float().cpu(), gz) < tol and rel_err(rd.grad.float().cpu(), gres) < tol
    assert rel_err(gd.grad.cpu(), gg) < (1e-4 if dtype == torch.float32 else BF16_TOL)
    assert rel_err(bd.grad.cpu(), gb) < (1e-4 if dtype == torch.float32 else BF16_TOL)


def test_fused_dropout_is_consistent_between_forward_and_backward(dev):
    from promptttspp_amd import functional as PF

    PF.manual_seed(7)
    torch.manual_seed(7)  # (the inputs of the LayerNorm / posenc part below)
    B, T, C = 4, 200, 256
    x = torch.ones(B, T, C, device=dev, requires_grad=True)
    w = torch.eye(C, device=dev).unsqueeze(-1).requires_grad_()
    y = PF.conv1d(x, w, None, drop_p=0.5)
    kept = (y != 0).float().mean().item()
    assert abs(kept - 0.5) < 0.01                       # keep rate
    assert torch.allclose(y[y != 0], torch.full_like(y[y != 0], 2.0))  # 1/(1-p) scaling
    y.backward(torch.ones_like(y))
    assert torch.equal((x.grad != 0), (y != 0))         # same mask regenerated in backward
    y2 = PF.conv1d(x, w, None, drop_p=0.5)
    assert not torch.equal(y2 != 0, y != 0)             # fresh mask per call
    # LayerNorm output dropout + posenc dropout
    g, b = torch.ones(C, device=dev, requires_grad=True), torch.zeros(C, device=dev, requires_grad=True)
    xin = torch.randn(B, T, C, device=dev, requires_grad=True)
    z = PF.layer_norm(xin, g, b, 1e-5, drop_out=0.25)
    assert abs((z == 0).float().mean().item() - 0.25) < 0.01
    pe = PF.posenc(xin, None, 2.0, 0.1)
    assert abs((pe == 0).float().mean().item() - 0.1) < 0.01
    pe.sum().backward()
    assert torch.equal(xin.grad == 0, pe == 0)


def test_fused_adamw_matches_torch(dev):
    from promptttspp_amd.optim import FusedAdamW

    torch.manual_seed(0)
    shapes = [(256, 256, 9), (1024,), (3, 5), (80, 256, 1), (4099,)]
    ps_ref = [torch.randn(s, requires_grad=True) for s in shapes]
    ps = [p.detach().clone().to(dev).requires_grad_() for p in ps_ref]
    o_ref = torch.optim.AdamW(ps_ref, lr=1e-3, betas=(0.9, 0.98), weight_decay=0.01)
    o = FusedAdamW(ps, lr=1e-3, betas=(0.9, 0.98), weight_decay=0.01, max_grad_norm=1.0)
    for step in range(3):
        for a, b in zip(ps_ref, ps):
            gr = torch.randn(a.shape) * (3.0 if step == 0 else 0.01)
            a.grad, b.grad = gr.clone(), gr.to(dev)
        torch.nn.utils.clip_grad_norm_(ps_ref, 1.0)
        o_ref.step()
        o.step()
        for a, b in zip(ps_ref, ps):
            assert rel_err(b.detach().cpu(), a.detach()) < 1e-5, step


def test_fused_adamw_norm_reductions_agree(dev):
    """The fixed-order gradient norm (default) and the one-launch atomic form (deterministic_norm = False, the round-1 path
    kept in the ABI) clip by the same factor to f32 rounding; the fixed-order one is bit-reproducible."""
    from promptttspp_amd.optim import FusedAdamW

    torch.manual_seed(1)
    shapes = [(512, 256, 3), (4099,), (80, 256, 1), (7,)]
    grads = [torch.randn(s) * 2.0 for s in shapes]
    outs = []
    for det in (True, True, False):
        ps = [torch.ones(s, device=dev).requires_grad_() for s in shapes]
        o = FusedAdamW(ps, lr=1e-2, betas=(0.9, 0.98), weight_decay=0.0, max_grad_norm=1.0)
        o.deterministic_norm = det
        for p, g in zip(ps, grads):
            p.grad = g.to(dev)
        o.step()
        torch.cuda.synchronize()
        outs.append(([p.detach().cpu() for p in ps], float(o.grad_norm())))
    want = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads)))
    assert all(torch.equal(a, b) for a, b in zip(outs[0][0], outs[1][0])) and outs[0][1] == outs[1][1]
    assert abs(outs[0][1] - want) < 1e-5 * want and abs(outs[2][1] - want) < 1e-5 * want
    assert all(rel_err(a, b) < 1e-6 for a, b in zip(outs[0][0], outs[2][0]))


def test_conv_family_at_bench_size_properties(dev):
    """BASELINE-sized shapes (one 30 000-token batch: 19 utterances x 1580 frames, DiffNet's dilated
    256 -> 512, k = 3 conv) through size-independent properties, bf16:
      * forward: linearity  conv(a x1 + x2) = a conv(x1) + conv(x2)  (bias-free) within bf16 rounding, and
        the first two utterances against a float32 torch convolution of the same bf16 operands;
      * weight gradient (LDS-DMA kernel + workspace reduction): bit-identical across two runs
        (deterministic split-K), accumulate semantics (second call adds), one tap against an einsum;
      * ragged lengths: rows past each length contribute nothing (in_mask) and produce zeros (out_mask)."""
    from promptttspp_amd import ops

    torch.manual_seed(0)
    B, T, Cin, Cout, ks, dil = 19, 1580, 256, 512, 3, 4
    pad = dil * (ks - 1) // 2
    x1 = torch.randn(B, T, Cin, device=dev).bfloat16()
    x2 = torch.randn(B, T, Cin, device=dev).bfloat16()
    w = torch.randn(Cout, Cin, ks, device=dev) * 0.05
    wp = ops.pack_conv_weight(w, torch.bfloat16)
    conv = lambda x, **kw: ops.conv1d(x, wp, None, Cout, ks=ks, dil=dil, pad=pad, **kw).float()  # noqa: E731
    y1, y2 = conv(x1), conv(x2)
    xs = (0.5 * x1.float() + x2.float()).bfloat16()          # exactly representable scaling, rounded sum
    ys = conv(xs)
    ref = 0.5 * y1 + y2
    assert float((ys - ref).abs().max()) < 3e-2 * float(ref.abs().max())   # bf16 input/output rounding
    t_ref = F.conv1d(x1[:2].float().transpose(1, 2), w.bfloat16().float(), None, padding=pad, dilation=dil).transpose(1, 2)
    assert rel_err(y1[:2].cpu(), t_ref.cpu()) < 4e-3

    lens = torch.randint(T // 3, T + 1, (B,), dtype=torch.int32, device=dev)
    ym = conv(x1, lengths=lens, in_mask=True, out_mask=True)
    tpos = torch.arange(T, device=dev)[None, :]
    valid = (tpos < lens[:, None]).unsqueeze(-1)
    assert float(ym.masked_select(~valid.expand_as(ym)).abs().max()) == 0.0
    xz = torch.where(valid, x1, torch.zeros_like(x1))
    assert torch.equal(ym, torch.where(valid, conv(xz), torch.zeros_like(ym)))  # masked input == zeroed input

    dy = torch.randn(B, T, Cout, device=dev).bfloat16()
    dw_a, db_a = ops.conv1d_wgrad(x1, dy, Cin, Cout, ks, dil, pad)
    dw_b, db_b = ops.conv1d_wgrad(x1, dy, Cin, Cout, ks, dil, pad)
    assert torch.equal(dw_a, dw_b)                                          # deterministic split-K reduction
    ops.conv1d_wgrad(x1, dy, Cin, Cout, ks, dil, pad, dw_out=dw_b, db_out=db_b)
    assert rel_err(dw_b.cpu(), (2 * dw_a).cpu()) < 1e-6 and rel_err(db_b.cpu(), (2 * db_a).cpu()) < 1e-5
    j = 2
    sh = j * dil - pad
    xsft = torch.zeros_like(x1)
    xsft[:, : T - sh] = x1[:, sh:]
    tap = torch.einsum("btc,btd->cd", dy.float(), xsft.float())
    assert rel_err(dw_a[:, :, j].cpu(), tap.cpu()) < 1e-4
    assert rel_err(db_a.cpu(), dy.float().sum((0, 1)).cpu()) < 1e-5
    dw_m, _ = ops.conv1d_wgrad(x1, dy, Cin, Cout, ks, dil, pad, lengths=lens, in_mask=True)
    dw_z, _ = ops.conv1d_wgrad(xz, dy, Cin, Cout, ks, dil, pad)
    assert torch.equal(dw_m, dw_z)


def test_fused_adamw_invalidates_packed_weight_caches(dev):
    """FusedAdamW updates parameters through raw pointers; the packed-operand caches are keyed on
    Tensor._version, so the optimiser must bump it: the NEXT forward has to see the updated weights."""
    from promptttspp_amd import functional as PF
    from promptttspp_amd.optim import FusedAdamW

    torch.manual_seed(0)
    lin = torch.nn.Linear(64, 32).to(dev)
    x = torch.randn(2, 10, 64, device=dev)
    opt = FusedAdamW(lin.parameters(), lr=0.1)
    y0 = PF.linear(x, lin.weight, lin.bias)
    v0 = lin.weight._version
    y0.square().mean().backward()
    opt.step()
    assert lin.weight._version > v0
    y1 = PF.linear(x, lin.weight, lin.bias)
    ref = F.linear(x, lin.weight.detach(), lin.bias.detach())
    assert rel_err(y1.detach().cpu(), ref.cpu()) < 1e-5          # the new weights, not the cached pack
    assert float((y1 - y0).abs().max()) > 1e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_batched_repack_matches_per_tensor_pack(dtype, dev):
    """repack_all() (one launch over a pointer table, after FusedAdamW) must leave every cached operand --
    forward and data-gradient forms, single weights, k > 1 convs, odd channel counts with padding, fused
    q|k|v stacks -- identical to a fresh per-tensor pack of the updated weights."""
    from promptttspp_amd import functional as PF
    from promptttspp_amd import ops
    from promptttspp_amd.optim import FusedAdamW

    torch.manual_seed(1)
    PF.clear_caches()
    conv = torch.nn.Conv1d(80, 256, 5).to(dev)
    lin = torch.nn.Linear(256, 104).to(dev)
    qkv = [torch.nn.Linear(64, 64).to(dev) for _ in range(3)]
    # the tile shapes of the batched kernel: k = 9 (one pass of 32 columns in bf16, three of 12 in f32), k = 17 (two passes of
    # 16), a source whose rows are not 16-byte multiples (scalar loads), partial tiles in both directions, the gate-interleaved
    # and operand-stream orders
    ffn = torch.nn.Conv1d(96, 160, 9).to(dev)
    odd = torch.nn.Conv1d(81, 40, 3).to(dev)
    gate = torch.nn.Conv1d(64, 128, 3).to(dev)
    wide = torch.nn.Conv1d(256, 256, 17).to(dev)
    params = [conv.weight, lin.weight, ffn.weight, odd.weight, gate.weight, wide.weight] + [m.weight for m in qkv]
    opt = FusedAdamW(params, lr=0.05)
    stream = dtype == torch.bfloat16  # (modes 3 / 4 are bf16 operands)
    for step in range(3):
        got = [PF.packed(conv.weight, dtype), PF.packed(conv.weight, dtype, 1), PF.packed(lin.weight, dtype),
               PF.packed(lin.weight, dtype, 1), PF.packed_cat([m.weight for m in qkv], dtype),
               PF.packed_cat([m.weight for m in qkv], dtype, 1), PF.packed(ffn.weight, dtype), PF.packed(ffn.weight, dtype, 1),
               PF.packed(odd.weight, dtype), PF.packed(odd.weight, dtype, 1), PF.packed(gate.weight, dtype, 2)]
        cat = torch.cat([m.weight.detach() for m in qkv], dim=0)
        want = [ops.pack_conv_weight(conv.weight, dtype), ops.pack_conv_weight(conv.weight, dtype, 1),
                ops.pack_conv_weight(lin.weight, dtype), ops.pack_conv_weight(lin.weight, dtype, 1),
                ops.pack_conv_weight(cat, dtype), ops.pack_conv_weight(cat, dtype, 1), ops.pack_conv_weight(ffn.weight, dtype),
                ops.pack_conv_weight(ffn.weight, dtype, 1), ops.pack_conv_weight(odd.weight, dtype),
                ops.pack_conv_weight(odd.weight, dtype, 1), ops.pack_conv_weight(gate.weight, dtype, 2)]
        if stream:
            got += [PF.packed(wide.weight, dtype, 3), PF.packed(wide.weight, dtype, 4)]
            want += [ops.pack_conv_weight(wide.weight, dtype, 3), ops.pack_conv_weight(wide.weight, dtype, 4)]
        else:
            got += [PF.packed(wide.weight, dtype), PF.packed(wide.weight, dtype, 1)]
            want += [ops.pack_conv_weight(wide.weight, dtype), ops.pack_conv_weight(wide.weight, dtype, 1)]
        for i, (a, b) in enumerate(zip(got, want)):
            assert a.shape == b.shape and torch.equal(a, b), (step, i)
        for p in params:
            p.grad = torch.randn_like(p)
        opt.step()  # bumps versions and calls repack_all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,C", [(4097, 32), (700, 256), (50000, 64)])
def test_batchnorm_act_matches_torch_and_leaves_scratch_zero(dtype, rows, C, dev):
    """Training-mode BatchNorm + activation (ptpp_bn_stats / bn_act_fwd / bn_act_bwd: replicated
    cross-block sums) against torch.nn.BatchNorm1d on the CPU: output, running estimates, dx, dgamma,
    dbeta; the reduction scratch must be all zero again after every call."""
    from promptttspp_amd import nn_ops, ops

    x = rnd(1, rows, C, scale=1.5) + 0.5 * rnd(2, 1, C)
    dy = rnd(3, rows, C)
    if dtype == torch.bfloat16:
        x, dy = x.bfloat16().float(), dy.bfloat16().float()
    tol = 2e-5 if dtype == torch.float32 else BF16_TOL
    bn_ref = torch.nn.BatchNorm1d(C)
    with torch.no_grad():
        bn_ref.weight.copy_(1.0 + 0.2 * rnd(4, C))
        bn_ref.bias.copy_(0.1 * rnd(5, C))
        bn_ref.running_mean.copy_(0.3 * rnd(6, C))
        bn_ref.running_var.copy_(1.0 + 0.1 * rnd(7, C).abs())
    bn = torch.nn.BatchNorm1d(C)
    bn.load_state_dict(bn_ref.state_dict())
    bn = bn.to(dev)
    xr = x.clone().requires_grad_()
    zr = bn_ref(xr)
    yr = zr * torch.sigmoid(zr)  # swish, the Conformer convolution module's activation
    dx_ref, dg_ref, db_ref = torch.autograd.grad(yr, (xr, bn_ref.weight, bn_ref.bias), dy)
    xd = x.to(dev, dtype).requires_grad_()
    y = nn_ops.batch_norm_act(xd, bn, act="swish")
    dx, dg, db = torch.autograd.grad(y, (xd, bn.weight, bn.bias), dy.to(dev, dtype))
    assert rel_err(y.float().cpu(), yr.detach()) < tol
    assert rel_err(bn.running_mean.cpu(), bn_ref.running_mean) < 1e-5
    assert rel_err(bn.running_var.cpu(), bn_ref.running_var) < 1e-5
    assert int(bn.num_batches_tracked) == 1
    assert rel_err(dx.float().cpu(), dx_ref) < (1e-4 if dtype == torch.float32 else BF16_TOL)
    assert rel_err(dg.cpu(), dg_ref) < (1e-4 if dtype == torch.float32 else BF16_TOL)
    assert rel_err(db.cpu(), db_ref) < (1e-4 if dtype == torch.float32 else BF16_TOL)
    torch.cuda.synchronize()
    key = [k for k in ops._red if k[2] == ops._stream().value]
    assert key and int(ops._red[key[0]].count_nonzero()) == 0


def test_layernorm_bwd_accumulates_into_given_buffers(dev):
    """dgamma / dbeta are ADDED to the caller's buffers (the trainer hands in views of the flat gradient
    buffer), twice in a row, and the result does not depend on the buffers' previous content."""
    from promptttspp_amd import ops

    B, T, C = 5, 333, 512
    x, dy = rnd(1, B, T, C), rnd(2, B, T, C)
    g, b = 1.0 + 0.1 * rnd(3, C), 0.1 * rnd(4, C)
    xs = x.clone().requires_grad_()
    gp, bp = g.clone().requires_grad_(), b.clone().requires_grad_()
    _, dg_ref, db_ref = torch.autograd.grad(R.layer_norm_last(xs, gp, bp, 1e-5), (xs, gp, bp), dy)
    y, mean, rstd, xsum = ops.layernorm_fwd(x.to(dev), g.to(dev), b.to(dev), 1e-5, save_stats=True, save_sum=True)
    acc_g = torch.full((C,), 3.0, device=dev)
    acc_b = torch.full((C,), -2.0, device=dev)
    for _ in range(2):
        ops.layernorm_bwd(dy.to(dev), xsum, g.to(dev), mean, rstd, dgamma_out=acc_g, dbeta_out=acc_b)
    assert rel_err((acc_g.cpu() - 3.0) / 2, dg_ref) < 1e-4
    assert rel_err((acc_b.cpu() + 2.0) / 2, db_ref) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,cin,cout,act", [(32, 1024, 256, None), (19, 256, 1024, "mish"), (96, 768, 512, "relu"),
                                               (1, 512, 12, None), (128, 264, 80, "swish"), (33, 256, 5120, None)])
def test_skinny_gemm_path(rows, cin, cout, act, dtype, dev):
    """Linear layers over a handful of rows (ks = 1, <= 128 rows, K >= 256) take the split-K skinny kernel:
    same contract as the tile kernel -- bias, activation, residual, out_scale, (B, 1, C) and (1, T, C) views."""
    from promptttspp_amd import ops

    x = rnd(1, rows, cin)
    w = rnd(2, cout, cin, scale=1.0 / np.sqrt(cin))
    b = 0.1 * rnd(3, cout)
    res = rnd(4, rows, cout)
    if dtype == torch.bfloat16:
        x, w, res = x.bfloat16().float(), w.bfloat16().float(), res.bfloat16().float()
    z = x @ w.t() + b
    if act == "mish":
        z = F.mish(z)
    elif act == "relu":
        z = F.relu(z)
    elif act == "swish":
        z = F.silu(z)
    ref = res + 0.5 * z
    tol = F32_TOL * 5 if dtype == torch.float32 else BF16_TOL
    wp = ops.pack_conv_weight(w.to(dev), dtype)
    for shape in ((1, rows, cin), (rows, 1, cin)):
        xd = x.to(dev, dtype).view(shape)
        rd = res.to(dev, dtype).view(shape[0], shape[1], cout)
        y = ops.conv1d(xd, wp, b.to(dev), cout, act=act, res=rd, out_scale=0.5)
        assert y.shape == (shape[0], shape[1], cout)
        assert rel_err(y.float().cpu().view(rows, cout), ref) < tol


@pytest.mark.parametrize("late", [False, True])
def test_fast_repack_path_keeps_packed_operands_current(dev, late):
    """The trainer's configuration (FlatGradReducer gradients + FusedAdamW.stable_grads): from the second
    step on the optimiser skips its pointer scan and functional.repack_all reuses the cached launch table and
    only advances the stamps.  After several steps every cached operand must equal a fresh pack of the
    current weight, no entry may have needed a lazy re-pack, and adding a layer must fall back to the scan.
    ``late``: the last two layers' operands are marked as first read late in the forward (functional.mark_late_pack): their
    share of the re-pack is a second launch on the weight-gradient side stream, joined by whoever reads such an operand first."""
    from promptttspp_amd import functional as PF
    from promptttspp_amd import ops
    from promptttspp_amd.optim import FusedAdamW
    from promptttspp_amd.parallel import FlatGradReducer

    PF.clear_caches()
    torch.manual_seed(0)
    net = torch.nn.ModuleList([torch.nn.Linear(64, 96), torch.nn.Linear(96, 32), torch.nn.Conv1d(32, 64, 3, padding=1)]).to(dev)
    params = list(net.parameters())
    red = FlatGradReducer(params)
    opt = FusedAdamW(params, lr=0.05)
    opt.stable_grads = True
    x = torch.randn(3, 20, 64, device=dev)
    if late:
        PF.mark_late_pack([*net[1].parameters(), *net[2].parameters()])

    def fwd(extra=None):
        h = PF.linear(x, net[0].weight, net[0].bias, act="relu")
        h = PF.linear(h, net[1].weight, net[1].bias)
        h = PF.conv1d(h, net[2].weight, net[2].bias, ks=3, pad=1)
        return h if extra is None else PF.linear(h, extra.weight, extra.bias)

    try:
        calls = []
        orig = ops.pack_conv_weight
        ops.pack_conv_weight = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            for it in range(5):
                red.zero_grad()
                y = fwd()
                y.square().mean().backward()
                red.finish()
                opt.step()
                if it == 0:
                    n_first = len(calls)  # the first forward/backward packs every operand once
                if late and PF._direct["side"] is not None:
                    assert PF._repack["late"] is not None and PF._late["pending"]  # a second launch is in flight on the side stream
            assert PF._repack["gen"] is not None            # the fast path is armed ...
            assert len(calls) == n_first                    # ... and nothing was re-packed one by one since
            for m in net:
                w = m.weight if m.weight.dim() == 3 else m.weight.unsqueeze(-1)
                for mode in (0, 1):
                    cached = PF.packed(m.weight, torch.float32, mode)
                    assert torch.equal(cached, orig(w.detach(), torch.float32, mode))
            ref = F.conv1d(F.linear(F.relu(F.linear(x, net[0].weight, net[0].bias)), net[1].weight, net[1].bias).transpose(1, 2),
                           net[2].weight, net[2].bias, padding=1).transpose(1, 2)
            assert rel_err(fwd().detach().cpu(), ref.detach().cpu()) < 1e-4
            # a new layer enters the cache: the next repack must notice (generation changed) and still be right
            extra = torch.nn.Linear(64, 16).to(dev)
            fwd(extra).square().mean().backward()
            red.finish()
            opt.step()
            w = net[0].weight
            assert torch.equal(PF.packed(w, torch.float32, 0), orig(w.detach().unsqueeze(-1), torch.float32, 0))
        finally:
            ops.pack_conv_weight = orig
    finally:
        PF.enable_direct_grads(False)
        PF.clear_caches()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C", [32, 64])
def test_fused_amp_layer_matches_oracle(C, dtype, dev):
    """ptpp_amp_layer_fwd (one kernel: Snake, dilated conv, Snake, conv, residual, block mean) against the oracle's
    AMP layer (vocoders/bigvgan.py:42-47) -- every (kernel size, dilation) of the generator, lengths shorter than the
    halo, lengths that are not a multiple of the tile, several tiles per utterance (the tile seams), with and
    without the running block mean."""
    from promptttspp_amd import ops

    g = load_golden("aa_snake")
    taps = (ops._taps(g["f_up"]), ops._taps(g["f_dn"]))
    tol = {torch.float32: 5e-5, torch.bfloat16: 3e-2, torch.float16: 4e-3}[dtype]  # (half: 3 more mantissa bits than bf16)
    cases = [(ks, d, T) for ks in (3, 7, 11) for d in (1, 3, 5) for T in ((1, 7, 300) if ks == 11 else (45,))]
    cases += [(11, 5, 777), (3, 1, 1030), (7, 3, 513)]
    for n, (ks, d, T) in enumerate(cases):
        B = 2
        r = np.random.default_rng(1000 + n)
        sd = {}
        for name, dil in (("conv1", d), ("conv2", 1)):
            sd[f"l.{name}.weight"] = torch.from_numpy((r.standard_normal((C, C, ks)) / np.sqrt(C * ks)).astype(np.float32))
            sd[f"l.{name}.bias"] = torch.from_numpy((0.1 * r.standard_normal(C)).astype(np.float32))
        for a in ("act1", "act2"):
            sd[f"l.{a}.act.alpha"] = torch.from_numpy((0.3 * r.standard_normal((1, C, 1))).astype(np.float32))
            sd[f"l.{a}.up.filter"], sd[f"l.{a}.down.lowpass.filter"] = g["f_up"], g["f_dn"]
        x = torch.from_numpy(r.standard_normal((B, C, T)).astype(np.float32))
        acc = torch.from_numpy(r.standard_normal((B, C, T)).astype(np.float32))
        if dtype != torch.float32:
            x, acc = x.to(dtype).float(), acc.to(dtype).float()
            for k in list(sd):
                if k.endswith("weight"):
                    sd[k] = sd[k].to(dtype).float()
        ref = R.amp_layer(sd, "l", x, ks, d)
        xc = x.transpose(1, 2).contiguous().to(dev, dtype)
        w1 = ops.pack_conv_weight(sd["l.conv1.weight"].to(dev), dtype)
        w2 = ops.pack_conv_weight(sd["l.conv2.weight"].to(dev), dtype)
        b1, b2 = sd["l.conv1.bias"].to(dev), sd["l.conv2.bias"].to(dev)
        la1, la2 = sd["l.act1.act.alpha"].reshape(-1).to(dev), sd["l.act2.act.alpha"].reshape(-1).to(dev)
        y = ops.amp_layer(xc, w1, b1, w2, b2, la1, la2, taps, taps, ks, d)
        assert rel_err(y.float().cpu().transpose(1, 2), ref) < tol, (ks, d, T, rel_err(y.float().cpu().transpose(1, 2), ref))
        # last layer of a block: (x + conv2(..)) / 3 added to the running mean of the blocks
        y2 = ops.amp_layer(xc, w1, b1, w2, b2, la1, la2, taps, taps, ks, d, res2=acc.transpose(1, 2).contiguous().to(dev, dtype),
                           out_scale=1 / 3, res_scale=1 / 3)
        ref2 = acc + ref / 3
        assert rel_err(y2.float().cpu().transpose(1, 2), ref2) < tol, (ks, d, T, "res2")


def test_fused_amp_layer_equals_the_four_launch_pipeline_bf16(dev):
    """bf16: the fused layer rounds where the separate launches round (a1, c1, a2 and y are bf16 tensors there too), so
    on a bench-shaped tile it agrees with snake -> conv -> snake -> conv(+res) to bf16 resolution, batch entries are
    independent, and a run repeats bit for bit."""
    from promptttspp_amd import ops

    g = load_golden("aa_snake")
    taps = (ops._taps(g["f_up"]), ops._taps(g["f_dn"]))
    for C, ks, d, T in ((32, 11, 5, 4000), (64, 7, 3, 2100)):
        B = 3
        x = torch.randn(B, T, C, device=dev, generator=torch.Generator(dev).manual_seed(C)).bfloat16()
        w = [(torch.randn(C, C, ks, device=dev) / (C * ks) ** 0.5) for _ in range(2)]
        b = [0.1 * torch.randn(C, device=dev) for _ in range(2)]
        la = [0.3 * torch.randn(C, device=dev) for _ in range(2)]
        wp = [ops.pack_conv_weight(t, torch.bfloat16) for t in w]
        y = ops.amp_layer(x, wp[0], b[0], wp[1], b[1], la[0], la[1], taps, taps, ks, d)
        a = ops.aa_snake(x, la[0], *taps)
        a = ops.conv1d(a, wp[0], b[0], C, ks=ks, dil=d, pad=d * (ks - 1) // 2)
        a = ops.aa_snake(a, la[1], *taps)
        ref = ops.conv1d(a, wp[1], b[1], C, ks=ks, dil=1, pad=(ks - 1) // 2, res=x)
        assert rel_err(y.float(), ref.float()) < 2e-2
        assert float((y.float() - ref.float()).abs().mean() / ref.float().abs().mean()) < 5e-3
        assert torch.equal(y, ops.amp_layer(x, wp[0], b[0], wp[1], b[1], la[0], la[1], taps, taps, ks, d))
        assert torch.equal(y[1:2], ops.amp_layer(x[1:2].contiguous(), wp[0], b[0], wp[1], b[1], la[0], la[1], taps, taps, ks, d))


def test_fused_amp_layer_second_generation_is_bit_identical(dev, monkeypatch):
    """amp_fused_kernel (csrc/amp_fused.hip: 8 waves, padded LDS rows, clamp-free interior tiles, two channel fragments per
    wave) keeps every FMA order and rounding point of the round-2 kernel (amp_layer_kernel, still the f32 path): bit-identical
    on interior tiles, edge tiles, utterances shorter than the halo, ragged last tiles, with and without the block mean --
    for every tile-height variant."""
    from promptttspp_amd import ops

    g = load_golden("aa_snake")
    taps = (ops._taps(g["f_up"]), ops._taps(g["f_dn"]))
    gen = torch.Generator(dev).manual_seed(11)
    for C in (32, 64):
        for ks, d, T in ((3, 1, 1500), (7, 3, 2100), (11, 5, 4000), (11, 1, 777), (3, 5, 5), (7, 1, 61), (11, 3, 385)):
            B = 2
            x = torch.randn(B, T, C, device=dev, generator=gen).bfloat16()
            acc = torch.randn(B, T, C, device=dev, generator=gen).bfloat16()
            w = [ops.pack_conv_weight(torch.randn(C, C, ks, device=dev, generator=gen) / (C * ks) ** 0.5, torch.bfloat16) for _ in range(2)]
            b = [0.1 * torch.randn(C, device=dev, generator=gen) for _ in range(2)]
            la = [0.3 * torch.randn(C, device=dev, generator=gen) for _ in range(2)]

            def run(**kw):
                return ops.amp_layer(x, w[0], b[0], w[1], b[1], la[0], la[1], taps, taps, ks, d, **kw)

            monkeypatch.setenv("PTPP_AMP_OLD", "1")
            ref, ref2 = run(), run(res2=acc, out_scale=1 / 3, res_scale=1 / 3)
            monkeypatch.delenv("PTPP_AMP_OLD")
            for variant in ("0", "1", "2", "3", "4", "5", "6", "7", "8"):
                monkeypatch.setenv("PTPP_AMP_VARIANT", variant)
                assert torch.equal(run(), ref), (C, ks, d, T, variant)
                assert torch.equal(run(res2=acc, out_scale=1 / 3, res_scale=1 / 3), ref2), (C, ks, d, T, variant, "res2")
            monkeypatch.delenv("PTPP_AMP_VARIANT")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C", [128, 256])
def test_snake_conv1d_wide_stage_kernel(C, dtype, dev):
    """ptpp_snake_conv1d_fwd (csrc/amp_fused.hip: the anti-aliased Snake applied while the conv's input tile is staged, C = 128 /
    256) against the oracle's aa_snake + conv (layers/activations.py:22-44, 74-138; vocoders/bigvgan.py:42-47) on bf16-rounded
    inputs, and against the two launches it replaces (ptpp_aa_snake_fwd + ptpp_conv1d_fwd: the same rounding points, a different
    accumulation order): interior and edge tiles, utterances shorter than the halo, ragged last tiles, with residual, scales
    and the running block mean; batch entries independent; a run repeats bit for bit."""
    import torch.nn.functional as F

    from promptttspp_amd import ops

    g = load_golden("aa_snake")
    taps = (ops._taps(g["f_up"]), ops._taps(g["f_dn"]))
    gen = torch.Generator(dev).manual_seed(C)
    for ks, d, T in ((3, 1, 700), (7, 3, 1000), (11, 5, 1537), (11, 1, 333), (3, 5, 5), (7, 5, 61), (11, 3, 129)):
        B = 2
        x = torch.randn(B, T, C, device=dev, generator=gen).to(dtype)
        res = torch.randn(B, T, C, device=dev, generator=gen).to(dtype)
        acc = torch.randn(B, T, C, device=dev, generator=gen).to(dtype)
        w = (torch.randn(C, C, ks, device=dev, generator=gen) / (C * ks) ** 0.5).to(dtype).float()
        b = 0.1 * torch.randn(C, device=dev, generator=gen)
        la = 0.3 * torch.randn(C, device=dev, generator=gen)
        wp = ops.pack_conv_weight(w, dtype)
        ws = ops.amp_pack_wstream(wp, C, ks)
        pad = d * (ks - 1) // 2
        # oracle, f32 from the same bf16-rounded inputs
        a_ref = R.aa_snake(x.float().cpu().transpose(1, 2), la.cpu(), g["f_up"], g["f_dn"])
        c_ref = F.conv1d(a_ref, w.cpu(), b.cpu(), padding=pad, dilation=d).transpose(1, 2)
        y = ops.snake_conv1d(x, ws, b, la, taps, ks, d)
        tol = 2e-2 if dtype == torch.bfloat16 else 3e-3
        assert rel_err(y.float().cpu(), c_ref) < tol, (ks, d, T, rel_err(y.float().cpu(), c_ref))
        a = ops.aa_snake(x, la, *taps)
        two = ops.conv1d(a, wp, b, C, ks=ks, dil=d, pad=pad)
        assert rel_err(y.float(), two.float()) < tol / 2, (ks, d, T)
        assert float((y.float() - two.float()).abs().mean() / two.float().abs().mean()) < tol / 10, (ks, d, T)
        # residual, scales, running mean
        y2 = ops.snake_conv1d(x, ws, b, la, taps, ks, d, res=res, res2=acc, out_scale=1 / 3, res_scale=1 / 3)
        ref2 = acc.float().cpu() + (res.float().cpu() + c_ref) / 3
        assert rel_err(y2.float().cpu(), ref2) < tol, (ks, d, T, "res")
        assert torch.equal(y, ops.snake_conv1d(x, ws, b, la, taps, ks, d))
        assert torch.equal(y[1:2], ops.snake_conv1d(x[1:2].contiguous(), ws, b, la, taps, ks, d))


def test_snake_conv_post_tanh_kernel(dev):
    """ptpp_snake_conv_post_tanh (act_post + conv_post + tanh in one launch, vocoders/bigvgan.py:129-131) against the oracle's
    aa_snake + conv + tanh on bf16-rounded input and against the two launches it replaces; edge tiles, short and ragged
    lengths; batch entries independent."""
    import torch.nn.functional as F

    from promptttspp_amd import ops

    g = load_golden("aa_snake")
    taps = (ops._taps(g["f_up"]), ops._taps(g["f_dn"]))
    gen = torch.Generator(dev).manual_seed(7)
    C, ks = 32, 7
    for T in (5, 61, 512, 777, 2100):
        B = 3
        x = torch.randn(B, T, C, device=dev, generator=gen).bfloat16()
        w = 0.2 * torch.randn(ks, C, device=dev, generator=gen)
        la = 0.3 * torch.randn(C, device=dev, generator=gen)
        y = ops.snake_conv_post_tanh(x, la, taps, w, 0.05)
        a_ref = R.aa_snake(x.float().cpu().transpose(1, 2), la.cpu(), g["f_up"], g["f_dn"])
        ref = torch.tanh(F.conv1d(a_ref, w.cpu().t().unsqueeze(0), torch.tensor([0.05]), padding=ks // 2))[:, 0]
        assert float((y.cpu() - ref).abs().max()) < 2e-2, (T, float((y.cpu() - ref).abs().max()))
        two = ops.conv_post_tanh(ops.aa_snake(x, la, *taps), w, 0.05)
        # the two launches: aa_snake_kernel scales the sin argument as (u e^alpha) / 2 pi, the fused kernels as u (e^alpha / 2 pi):
        # an activated value now and then rounds to the other bf16 neighbour
        assert float((y - two).abs().max()) < 5e-3 and float((y - two).abs().mean()) < 2e-5, T
        assert torch.equal(y[1:2], ops.snake_conv_post_tanh(x[1:2].contiguous(), la, taps, w, 0.05))


def test_mel_front_end_and_lowpass_on_device(dev):
    """n1 / n2 on the GPU: the log-mel front-end on the exact-f32 GEMM (windowed DFT and filterbank as two products)
    and the zero-phase IIR kernel, against the oracle's numpy restatements."""
    from scipy import signal

    from promptttspp.transforms import MelSpectrogramTransform
    from promptttspp.utils.model import lowpass_filter

    rng = np.random.default_rng(11)
    t = MelSpectrogramTransform(sample_rate=24000, n_fft=512, win_length=480, hop_length=240, f_min=63.0, f_max=12000.0,
                                n_mels=80, norm="slaney", mel_scale="slaney").to(dev)
    for L in (24000, 7777, 600):
        wav = (0.3 * rng.standard_normal((2, L))).astype(np.float32)
        mel = t(torch.from_numpy(wav).to(dev)).cpu()
        assert mel.shape == (2, 80, 1 + L // 240)
        for i in range(2):
            ref = torch.from_numpy(R.mel_spectrogram_np(wav[i]))
            assert float((mel[i].double() - ref).abs().max()) < 1e-3, L      # log-mel, absolute
        spec = t.to_spec(torch.from_numpy(wav).to(dev)).cpu()
        cpu_spec = t.cpu().to_spec(torch.from_numpy(wav))
        t.to(dev)
        assert rel_err(spec, cpu_spec) < 1e-4

    b, a = signal.butter(5, [20 / 50], "lowpass")
    for shape in ((32, 1, 539), (1, 1, 61), (3, 1, 18)):
        x = (5.2 + 0.3 * rng.standard_normal(shape)).astype(np.float32)
        y = lowpass_filter(torch.from_numpy(x).to(dev), 100, cutoff=20).cpu()
        ref = x if shape[-1] <= 18 else R.filtfilt_zero_state(x, b, a)
        assert float((y.double() - torch.from_numpy(np.ascontiguousarray(ref))).abs().max()) < 2e-6, shape


@pytest.mark.parametrize("B,T,C,masked", [(3, 150, 128, False), (2, 333, 256, True), (5, 40, 256, False), (40, 640, 256, True),
                                          (64, 800, 256, True)])  # (the last grid takes the 128 x 128 configuration)
def test_conv1d_diffnet_post_fused_is_bit_identical_to_the_two_kernel_path(dev, B, T, C, masked):
    """ptpp_conv1d_diffnet_post (1 x 1 output projection with the DiffNet layer's residual / skip / next-input update in
    its epilogue, modules/denoiser.py:78-83) against ptpp_conv1d_fwd + ptpp_diffnet_post_fwd on the same inputs:
    xn, yin (bf16) and the f32 skip accumulator are equal bit for bit, first layer (init) and later layers, with and
    without the sequence mask, with and without a next layer (yin)."""
    from promptttspp_amd import ops

    assert ops.conv1d_diffnet_post_supported(C, C, torch.bfloat16)
    g = rnd(1, B, T, C).to(dev).bfloat16()
    x = rnd(2, B, T, C).to(dev).bfloat16()
    w = (rnd(3, 2 * C, C, 1) * 0.08).to(dev)
    bias = (rnd(4, 2 * C) * 0.1).to(dev)
    wp = ops.pack_conv_weight(w, torch.bfloat16)
    dn = rnd(5, B, C).to(dev).float().contiguous()
    lengths = torch.tensor([max(1, T - 17 * i) for i in range(B)], device=dev, dtype=torch.int32) if masked else None
    for init in (True, False):
        for dnext in (dn, None):
            skip_a = rnd(6, B, T, C).to(dev).float().contiguous()
            skip_b = skip_a.clone()
            o = ops.conv1d(g, wp, bias, 2 * C, lengths=lengths, out_mask=masked)
            xn_a, yin_a = ops.diffnet_post_fwd(o, x, skip_a, dnext, init=init)
            xn_b, yin_b = ops.conv1d_diffnet_post(g, wp, bias, x, skip_b, dnext, init=init, lengths=lengths, out_mask=masked)
            assert torch.equal(xn_a, xn_b) and torch.equal(skip_a, skip_b)
            assert (yin_a is None) == (yin_b is None) and (yin_a is None or torch.equal(yin_a, yin_b))


def test_ddpm_step_matches_the_tensor_ops(dev):
    """ptpp_ddpm_step against the tensor-op chain it replaces (predict_start_from_noise -> clamp -> q_posterior ->
    + sigma * noise, reference modules/diffusion.py:283-302) with the module's own schedule buffers: equal to the last
    bit or two in f32 (every product / sum rounded separately), for bf16 eps, per-utterance step indices, and the
    noise-free last step."""
    from promptttspp_amd import ops
    from promptttspp_amd.modules.diffusion import GaussianDiffusion

    class _Stub(torch.nn.Module):
        in_dim = 80

    gd = GaussianDiffusion(256, 80, _Stub(), K_step=100).to(dev)
    B, T, M = 5, 37, 80
    x = rnd(1, B, T, M).to(dev).float().contiguous() * 1.5
    noise = rnd(2, B, T, M).to(dev).float().contiguous()
    t = torch.tensor([99, 0, 17, 50, 3], device=dev, dtype=torch.long)
    for eps in (rnd(3, B, T, M).to(dev).float().contiguous(), rnd(3, B, T, M).to(dev).bfloat16().contiguous()):
        for nz in (noise, None):
            e = eps.float()
            x0 = gd.predict_start_from_noise(x, t, e).clamp_(-1.0, 1.0)
            mean, _, logvar = gd.q_posterior(x0, x, t)
            ref = mean if nz is None else mean + (0.5 * logvar).exp() * nz
            got = ops.ddpm_step(x, eps, nz, t, gd.sqrt_recip_alphas_cumprod, gd.sqrt_recipm1_alphas_cumprod,
                                gd.posterior_mean_coef1, gd.posterior_mean_coef2, gd.posterior_log_variance_clipped)
            d = float((ref - got).abs().max())
            print("ddpm_step max |diff|", d, "max |ref|", float(ref.abs().max()))
            assert d <= 2e-6 * max(1.0, float(ref.abs().max())), d  # (a last-bit difference in exp(0.5 * logvar) at most)


GLDS_CASES = [
    # B, T, Cin, Cout, ks, dil -- grids of >= 192 128-row tiles, i.e. the LDS-DMA kernel (csrc/conv1d_glds.h):
    (8, 1607, 256, 256, 3, 2),    # 64 x 128 tiles, ragged last tile
    (6, 1100, 128, 640, 5, 1),
    (40, 700, 256, 80, 1, 1),     # Cout = 80: one partial channel tile
    (8, 1500, 64, 192, 7, 3),     # Cout = 192: second channel tile half empty; one 64-channel K chunk
    (16, 6200, 128, 256, 3, 1),   # >= 1536 tiles: the 128 x 128 configuration
]


@pytest.mark.parametrize("case", GLDS_CASES)
def test_conv1d_lds_dma_kernel_epilogues(case, dev):
    """The bf16 LDS-DMA conv kernel at grid sizes that select it (the small cases of test_conv1d_fwd take the register-staged
    kernels): every activation of the specialised tile epilogue, masks, bias, a STRIDED residual with res_scale,
    out_scale, partial channel tiles, dropout (deterministic per seed, keep ratio, survivors scaled) -- against
    torch's own f32 convolution of the same bf16-rounded operands."""
    from promptttspp_amd import ops

    B, T, Cin, Cout, ks, dil = case
    pad = (ks - 1) * dil // 2
    x = rnd(1, B, T, Cin).bfloat16().to(dev)
    w = (rnd(2, Cout, Cin, ks) / np.sqrt(Cin * ks)).bfloat16().float().to(dev)
    b = rnd(3, Cout, scale=0.1).to(dev)
    wide = rnd(4, B, T, Cout + 64).bfloat16().to(dev)
    res = wide[:, :, 32 : 32 + Cout]  # rows of Cout channels inside a wider tensor (ld = Cout + 64)
    lens = torch.tensor([max(1, T - 97 * i) for i in range(B)], dtype=torch.int32, device=dev)
    mask = (torch.arange(T, device=dev)[None, :] < lens[:, None]).float()[:, :, None]
    wp = ops.pack_conv_weight(w, torch.bfloat16)
    acts = {"none": lambda v: v, "relu": F.relu, "gelu": F.gelu, "swish": F.silu, "tanh": torch.tanh, "mish": F.mish}
    z = F.conv1d((x.float() * mask).transpose(1, 2), w, b, padding=pad, dilation=dil).transpose(1, 2)
    for name, fn in acts.items():
        ref = 0.7 * res.float() + 0.5 * fn(z) * mask
        y = ops.conv1d(x, wp, b, Cout, ks=ks, dil=dil, pad=pad, act=name, lengths=lens, in_mask=True, out_mask=True, res=res,
                       out_scale=0.5, res_scale=0.7)
        assert rel_err(y.float().cpu(), ref.cpu()) < BF16_TOL, name
    # no masks / bias / residual
    ref2 = F.conv1d(x.float().transpose(1, 2), w, None, padding=pad, dilation=dil).transpose(1, 2)
    y2 = ops.conv1d(x, wp, None, Cout, ks=ks, dil=dil, pad=pad)
    assert rel_err(y2.float().cpu(), ref2.cpu()) < BF16_TOL
    if Cout % 4 == 0:
        d1 = ops.conv1d(x, wp, None, Cout, ks=ks, dil=dil, pad=pad, drop_p=0.25, drop_seed=11)
        d2 = ops.conv1d(x, wp, None, Cout, ks=ks, dil=dil, pad=pad, drop_p=0.25, drop_seed=11)
        d3 = ops.conv1d(x, wp, None, Cout, ks=ks, dil=dil, pad=pad, drop_p=0.25, drop_seed=12)
        assert torch.equal(d1, d2) and not torch.equal(d1, d3)
        kept = d1 != 0
        assert abs(float(kept.float().mean()) - 0.75) < 0.01
        full = (y2.float() / 0.75).bfloat16()
        assert rel_err(d1[kept].float().cpu(), full[kept].float().cpu()) < BF16_TOL


def test_conv1d_lds_dma_kernel_fused_gate(dev):
    """The fused DiffNet gate through the LDS-DMA kernel's tile epilogue (8-byte results, Cout / 2 output channels) against
    the two-launch path on the same permuted operands, at a grid size that selects that kernel."""
    from promptttspp_amd import ops

    B, T, C = 12, 1400, 256
    x = rnd(1, B, T, C).bfloat16().to(dev)
    w = (rnd(2, 2 * C, C, 3) / np.sqrt(3 * C)).to(dev)
    b = rnd(3, 2 * C, scale=0.1).to(dev)
    cond = rnd(4, B, T, 2 * C).bfloat16().to(dev)
    a = ops.conv1d(x, ops.pack_conv_weight(w, torch.bfloat16), b, 2 * C, ks=3, dil=2, pad=2, res=cond)
    g_ref = ops.gate_fwd(a)
    k = torch.arange(C // 4, device=dev)[:, None] * 4 + torch.arange(4, device=dev)[None, :]
    perm = torch.cat([k, k + C], dim=1).reshape(-1)  # [4 gate | their 4 filter partners] per 8 channels
    g = torch.empty(B, T, C, device=dev, dtype=torch.bfloat16)
    ops.conv1d(x, ops.pack_conv_weight(w[perm], torch.bfloat16), b[perm].contiguous(), 2 * C, ks=3, dil=2, pad=2, act="gate",
               res=cond[:, :, perm].contiguous(), out=g)
    assert rel_err(g.float().cpu(), g_ref.float().cpu()) < 2e-2  # (the two-launch path rounds the pre-activation to bf16)


@pytest.mark.parametrize("B,T,C", [(3, 150, 128), (2, 333, 256), (40, 640, 256), (64, 1600, 256)])
def test_conv1d_gate_bwd_fused_is_bit_identical_to_the_two_kernel_path(dev, B, T, C):
    """ptpp_conv1d_gate_bwd (data gradient of the DiffNet output projection with the gate backward in its epilogue;
    backward of modules/denoiser.py:72-78) against ptpp_conv1d_fwd + ptpp_gate_bwd: da written into a column slice of a
    wider tensor, bit for bit; the last grid takes the 128 x 128 configuration."""
    from promptttspp_amd import ops

    assert ops.conv1d_gate_bwd_supported(C, 2 * C, torch.bfloat16)
    do = rnd(1, B, T, 2 * C).to(dev).bfloat16()
    a = rnd(2, B, T, 2 * C).to(dev).bfloat16()
    w = (rnd(3, 2 * C, C, 1) * 0.06).to(dev)  # the projection C -> 2C; its data gradient maps 2C -> C
    wpt = ops.pack_conv_weight(w, torch.bfloat16, mode=1)
    wide_a = torch.zeros(B, T, 3 * 2 * C, device=dev, dtype=torch.bfloat16)
    wide_b = torch.zeros_like(wide_a)
    dg = ops.conv1d(do, wpt, None, C)
    ops.gate_bwd(a, dg, wide_a[:, :, 2 * C : 4 * C])
    ops.conv1d_gate_bwd(do, wpt, a, wide_b[:, :, 2 * C : 4 * C])
    assert torch.equal(wide_a, wide_b)
    assert float(wide_a[:, :, 2 * C : 4 * C].float().abs().max()) > 0


@pytest.mark.parametrize("B,T,G,D,masked", [(7, 53, 4, 1, True), (19, 1, 10, 256, False), (3, 40, 6, 5, True)])
def test_mdn_nll_fused_matches_the_tensor_ops(dev, B, T, G, D, masked):
    """ptpp_mdn_nll_fwd / _bwd (dimension-wise mixture NLL, reference modules/mdn.py:81-175) against the tensor-op chain
    and its autograd: every clamp regime is hit (log_sigma and log_pi below their floors, targets beyond 5 sigma),
    masked positions give +inf and zero gradients."""
    from promptttspp_amd.modules import mdn as M

    torch.manual_seed(5)
    lp_raw = rnd(1, B, T, G, D).to(dev) * 3.0
    ls = (rnd(2, B, T, G, D).to(dev) * 4.0 - 3.0)  # some below -7
    mu = rnd(3, B, T, G, D).to(dev)
    tgt = (rnd(4, B, T, D).to(dev) * 2.5).contiguous()  # some beyond 5 sigma of narrow components
    mask = None
    if masked:
        lens = torch.tensor([max(1, T - 7 * i) for i in range(B)], device=dev)
        mask = (torch.arange(T, device=dev)[None, :] < lens[:, None]).unsqueeze(-1)  # (B, T, 1) as the model passes it
    outs = []
    for fused in (False, True):
        M.FUSED_NLL = fused
        a, b, c = (t.clone().requires_grad_() for t in (lp_raw, ls, mu))
        lp = torch.log_softmax(a, dim=2) * 2.0  # (scaled: some weights fall below the -7 floor)
        nll = M.mdn_loss(lp, b, c, tgt, reduce=False, mask=mask)
        keep = mask.expand_as(nll) if mask is not None else torch.ones_like(nll, dtype=torch.bool)
        assert bool(torch.isinf(nll[~keep]).all()) if mask is not None else True
        w = rnd(9, *nll.shape).to(dev)
        (torch.where(keep, nll, torch.zeros_like(nll)) * w).sum().backward()
        outs.append((torch.where(keep, nll, torch.zeros_like(nll)).detach(), a.grad, b.grad, c.grad))
    M.FUSED_NLL = True
    ref, got = outs
    assert rel_err(got[0].cpu(), ref[0].cpu()) < 2e-6
    for g, r in zip(got[1:], ref[1:]):
        assert torch.isfinite(g).all()
        assert rel_err(g.cpu(), r.cpu()) < 2e-5


@pytest.mark.parametrize("variant,B,T,H,dk,drop", [("new", 5, 77, 2, 128, 0.0), ("new", 3, 256, 2, 128, 0.0), ("new", 19, 130, 2, 128, 0.0),
                                                   ("new", 2, 33, 4, 64, 0.0), ("plain", 19, 48, 12, 64, 0.1), ("plain", 4, 200, 2, 128, 0.0),
                                                   ("new", 1, 16, 2, 128, 0.0)])
def test_attention_forward_on_the_matrix_cores(dev, variant, B, T, H, dk, drop):
    """attn_fwd_mfma_kernel (bf16: K / V of one (utterance, head) resident in LDS, score / positional / context products on
    the MFMAs, rel-shift as a skewed read; reference esp/transformer/attention.py:63-93,237-305) against the exact-f32 row
    kernel on the same bf16-rounded operands: probabilities (saved for the backward), context, masked keys and padded query
    rows, probability dropout with the same (seed, element) mask, ragged lengths, T not a multiple of the tile sizes."""
    from promptttspp_amd import ops

    C = H * dk
    qkv = (rnd(1, B, T, 3 * C) * 0.5).to(dev).bfloat16()
    lens = torch.tensor([max(1, T - 11 * i) for i in range(B)], device=dev, dtype=torch.int32)
    pos = u = vb = None
    if variant == "new":
        pos = (rnd(2, 2 * T - 1, C) * 0.5).to(dev).bfloat16()
        u, vb = (0.1 * rnd(3, H, dk)).to(dev), (0.1 * rnd(4, H, dk)).to(dev)
    q, k, v = qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:]
    ctx, probs = ops.attention_fwd(q, k, v, pos, u, vb, lens, H, variant, save_probs=True, drop_p=drop, drop_seed=1234)
    qf = qkv.float()
    ctx_r, probs_r = ops.attention_fwd(qf[:, :, :C], qf[:, :, C:2 * C], qf[:, :, 2 * C:], pos.float() if pos is not None else None, u, vb,
                                       lens, H, variant, save_probs=True, drop_p=drop, drop_seed=1234)
    torch.cuda.synchronize()
    assert torch.isfinite(ctx.float()).all() and torch.isfinite(probs).all()
    # probabilities: rows sum to 1 over the valid keys, zero on masked keys and padded query rows
    for bi in range(B):
        n = int(lens[bi])
        assert float((probs[bi, :, :n, :n].sum(-1) - 1).abs().max()) < 1e-4
        assert float(probs[bi, :, :, n:].abs().max() if n < T else 0.0) == 0.0
        assert float(probs[bi, :, n:, :].abs().max() if n < T else 0.0) == 0.0
        assert float(ctx[bi, n:].float().abs().max() if n < T else 0.0) == 0.0
    assert float((probs - probs_r).abs().max()) < 2e-2 * float(probs_r.max())   # q + u / q + v rounded to bf16 operands
    err = float((ctx.float() - ctx_r).abs().max() / ctx_r.abs().max())
    assert err < 2e-2, err


@pytest.mark.parametrize("variant,B,T,H,dk,drop", [("new", 5, 77, 2, 128, 0.0), ("new", 19, 150, 4, 128, 0.1), ("new", 3, 176, 2, 128, 0.0), ("new", 3, 256, 2, 128, 0.0),
                                                   ("new", 4, 230, 4, 128, 0.1),
                                                   ("new", 2, 33, 4, 64, 0.0), ("plain", 6, 48, 4, 64, 0.1), ("plain", 4, 190, 2, 128, 0.0),
                                                   ("new", 1, 16, 2, 128, 0.0), ("new", 4, 65, 2, 64, 0.0)])
def test_attention_backward_on_the_matrix_cores(dev, variant, B, T, H, dk, drop):
    """attn_bwd_{q,k,pos}_mfma_kernel (bf16; reference esp/transformer/attention.py:63-93,237-305 under autograd) against the
    exact-f32 row / column kernels on the same bf16-rounded operands and the same dropout mask: dq, dk, dv, the positional
    table's gradient and the two bias gradients; ragged lengths (an utterance of one phone included), T not a multiple of any
    tile, gradients of padded rows exactly zero."""
    from promptttspp_amd import ops

    C = H * dk
    qkv = (rnd(1, B, T, 3 * C) * 0.5).to(dev).bfloat16()
    dctx = (rnd(5, B, T, C) * 0.5).to(dev).bfloat16()
    lens = torch.tensor([max(1, T - 17 * i) for i in range(B)], device=dev, dtype=torch.int32)
    pos = u = vb = None
    if variant == "new":
        pos = (rnd(2, 2 * T - 1, C) * 0.5).to(dev).bfloat16()
        u, vb = (0.1 * rnd(3, H, dk)).to(dev), (0.1 * rnd(4, H, dk)).to(dev)

    def run(dt):
        x, g = qkv.to(dt), dctx.to(dt)
        ps = pos.to(dt) if pos is not None else None
        q, k, v = x[:, :, :C], x[:, :, C:2 * C], x[:, :, 2 * C:]
        _, probs = ops.attention_fwd(q, k, v, ps, u, vb, lens, H, variant, save_probs=True, drop_p=drop, drop_seed=77)
        d = torch.full((B, T, 3 * C), float("nan"), device=dev, dtype=dt)
        dpos, du, dvb = ops.attention_bwd(q, k, v, ps, u, vb, probs, g, lens, H, variant, d[:, :, :C], d[:, :, C:2 * C], d[:, :, 2 * C:],
                                          drop_p=drop, drop_seed=77)
        torch.cuda.synchronize()
        return [d.float()] + ([dpos, du, dvb] if variant == "new" else [])

    got, ref = run(torch.bfloat16), run(torch.float32)
    for i, (a, r) in enumerate(zip(got, ref)):
        assert torch.isfinite(a).all(), i
        err = float((a - r).abs().max() / r.abs().max())
        assert err < 2e-2, (i, err)
    for bi in range(B):
        n = int(lens[bi])
        if n < T:
            assert float(got[0][bi, n:].abs().max()) == 0.0


@pytest.mark.parametrize("B,T,H,dk", [(3, 130, 2, 128), (4, 77, 4, 64)])
def test_attention_backward_on_the_matrix_cores_against_the_oracle(dev, B, T, H, dk):
    """ONE hop from the oracle: the bf16 attention forward + attn_bwd_{q,k,pos}_mfma_kernel against
    oracle/ref_torch.relpos_attention (esp/transformer/attention.py:63-93, 237-305) under f32 autograd -- the same weights,
    positional table and ragged lengths; q / k / v and the projected table enter the kernels rounded to bf16.  Compared: the
    context, and the gradients of the layer input (dq Wq + dk Wk + dv Wv), of linear_pos.weight (dpos^T pos_emb) and of the
    two biases."""
    from promptttspp_amd import ops

    C = H * dk
    L = 2 * T - 1
    sd = {}
    for i, n in enumerate(("linear_q", "linear_k", "linear_v", "linear_out")):
        sd[f"a.{n}.weight"] = rnd(10 + i, C, C) / C ** 0.5
        sd[f"a.{n}.bias"] = 0.1 * rnd(20 + i, C)
    sd["a.linear_out.weight"], sd["a.linear_out.bias"] = torch.eye(C), torch.zeros(C)   # the kernels end at the context
    sd["a.linear_pos.weight"] = rnd(30, C, C) / C ** 0.5
    sd["a.pos_bias_u"], sd["a.pos_bias_v"] = 0.1 * rnd(31, H, dk), 0.1 * rnd(32, H, dk)
    x = 0.7 * rnd(1, B, T, C)
    pos_emb = 0.7 * rnd(2, L, C)
    g = 0.5 * rnd(3, B, T, C)
    lens = torch.tensor([max(1, T - 19 * i) for i in range(B)], dtype=torch.int32)
    key_mask = torch.arange(T)[None, :] < lens[:, None]
    # oracle, f32 autograd
    leaf = {k: sd[k].clone().requires_grad_(True) for k in ("a.linear_pos.weight", "a.pos_bias_u", "a.pos_bias_v")}
    xr = x.clone().requires_grad_(True)
    ctx_ref = R.relpos_attention({**sd, **leaf}, "a", xr, pos_emb, key_mask, H, "new")
    valid = key_mask[:, :, None].float()
    (ctx_ref * g * valid).sum().backward()
    # kernels, bf16
    q = F.linear(x, sd["a.linear_q.weight"], sd["a.linear_q.bias"])
    k = F.linear(x, sd["a.linear_k.weight"], sd["a.linear_k.bias"])
    v = F.linear(x, sd["a.linear_v.weight"], sd["a.linear_v.bias"])
    qkv = torch.cat([q, k, v], -1).to(dev).bfloat16()
    pp = F.linear(pos_emb, sd["a.linear_pos.weight"]).to(dev).bfloat16()
    u, vb = sd["a.pos_bias_u"].to(dev), sd["a.pos_bias_v"].to(dev)
    ld = lens.to(dev)
    qd, kd, vd = qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:]
    ctx, probs = ops.attention_fwd(qd, kd, vd, pp, u, vb, ld, H, "new", save_probs=True)
    d = torch.zeros((B, T, 3 * C), device=dev, dtype=torch.bfloat16)
    dctx = (g * valid).to(dev).bfloat16()
    dpos, du, dvb = ops.attention_bwd(qd, kd, vd, pp, u, vb, probs, dctx, ld, H, "new", d[:, :, :C], d[:, :, C:2 * C], d[:, :, 2 * C:])
    torch.cuda.synchronize()

    def nerr(a, r):
        return float((a - r).abs().max() / r.abs().max())

    assert nerr(ctx.float().cpu() * valid, ctx_ref.detach() * valid) < 2e-2
    df = d.float().cpu()
    dx = df[:, :, :C] @ sd["a.linear_q.weight"] + df[:, :, C:2 * C] @ sd["a.linear_k.weight"] + df[:, :, 2 * C:] @ sd["a.linear_v.weight"]
    assert nerr(dx, xr.grad) < 3e-2, nerr(dx, xr.grad)
    assert nerr(dpos.cpu().t() @ pos_emb, leaf["a.linear_pos.weight"].grad) < 3e-2
    assert nerr(du.cpu().view(H, dk), leaf["a.pos_bias_u"].grad) < 3e-2
    assert nerr(dvb.cpu().view(H, dk), leaf["a.pos_bias_v"].grad) < 3e-2


@pytest.mark.parametrize("dtype,B,T,H,dk,w", [(torch.float32, 3, 77, 2, 128, 4), (torch.float32, 2, 130, 4, 64, 2),
                                              (torch.bfloat16, 5, 150, 2, 128, 4), (torch.float32, 1, 5, 2, 64, 4)])
def test_windowed_relative_attention_kernel(dev, dtype, B, T, H, dk, w):
    """ptpp_attention_win_fwd / _bwd (reference modules/transformer.py:59-137: scores += q . emb_k[j-i+w], out += P . emb_v
    inside the window) against the tensor-op form of the same module (band gathers + autograd): layer output on the valid rows,
    gradients of the input, the fused projection and both relative-position tables; ragged lengths, T shorter than the window."""
    from promptttspp_amd import config
    from promptttspp_amd.modules.transformer import RelativeMultiHeadAttention

    C = H * dk
    torch.manual_seed(3)
    m = RelativeMultiHeadAttention(C, H, 0.0, window_size=w).to(dev)
    with torch.no_grad():
        m.emb_rel_k.mul_(3.0)
        m.emb_rel_v.mul_(3.0)
    lens = torch.tensor([max(1, T - 23 * i) for i in range(B)], device=dev, dtype=torch.int32)
    valid = (torch.arange(T, device=dev)[None, :] < lens[:, None])[:, :, None].float()
    x0 = (rnd(1, B, T, C) * 0.7).to(dev)
    dy = (rnd(2, B, T, C).to(dev) * valid)
    tol = 5e-5 if dtype == torch.float32 else 3e-2

    def run(kernel):
        RelativeMultiHeadAttention.WINDOW_KERNEL = kernel
        for p_ in m.parameters():
            p_.grad = None
        with config.use_dtype(dtype):
            x = x0.to(dtype).clone().requires_grad_()
            y = m.forward_cl(x, lens)
            (y.float() * valid).backward(dy)
        torch.cuda.synchronize()
        return [(y.float() * valid).detach(), x.grad.float() * valid] + [p_.grad.float().clone() for p_ in m.parameters()]

    try:
        ref, got = run(False), run(True)
    finally:
        RelativeMultiHeadAttention.WINDOW_KERNEL = True
    names = ["y", "dx"] + [n for n, _ in m.named_parameters()]
    gscale = max(float(a.abs().max()) for a in ref[2:])
    for n, a, b in zip(names, ref, got):
        assert torch.isfinite(b).all(), n
        # (the key projection's bias has NO gradient -- a shift of every key moves all scores of a row alike -- both forms
        #  return rounding noise for it: errors are taken relative to the larger of the tensor's and 1e-3 of the largest gradient)
        err = float((a - b).abs().max() / max(float(a.abs().max()), 1e-3 * gscale))
        assert err < tol, (n, err)
    assert float(ref[0].abs().max()) > 0 and float(ref[names.index("emb_rel_k")].abs().max()) > 0


@pytest.mark.parametrize("B,T,last", [(2, 64, False), (3, 90, False), (32, 546, False), (1, 37, True)])
def test_sampler_head_kernel(dev, B, T, last):
    """ptpp_sampler_head (skip projection + ReLU + output projection, the reverse-diffusion update, the next step's input
    projection + first-layer input; reference modules/denoiser.py:147-152,131,76 and modules/diffusion.py:283-302) against the
    seven launches it replaces.  The update itself is the reference's unfused f32 sequence in both (equal to the torch ops bit
    for bit); eps may differ by one bf16 ulp in isolated elements (the conv kernel and this one feed the MFMA's K slots in a
    different order), which moves x by at most sqrt(1/abar - 1) ulp_bf16(eps) there and nothing elsewhere."""
    from promptttspp_amd import ops

    C, M, dt = 256, 80, torch.bfloat16
    r = lambda seed, *s, sc=1.0: (rnd(seed, *s) * sc).to(dev)
    s = r(1, B, T, C, sc=0.5).bfloat16()
    ws, bs, wo, bo = r(2, C, C, 1, sc=0.06), r(3, C, sc=0.1), r(4, M, C, 1, sc=0.06), r(5, M, sc=0.1)
    wi, bi = r(6, C, M, 1, sc=0.1), r(7, C, sc=0.1)
    x, noise, ds0 = r(8, B, T, M), r(9, B, T, M), r(10, B, C)
    t = torch.randint(0, 100, (B,), generator=torch.Generator().manual_seed(3)).to(dev)
    tabs = [(rnd(20 + i, 100).abs() + 0.5).to(dev) for i in range(4)] + [(-rnd(24, 100).abs()).to(dev)]
    wsp, wop, wip = (ops.pack_conv_weight(w_, dt) for w_ in (ws, wo, wi))
    h = ops.conv1d(s, wsp, bs, C, act="relu")
    eps = ops.conv1d(h, wop, bo, M)
    x1 = ops.ddpm_step(x, eps.contiguous(), None if last else noise, t, *tabs)
    # the update kernel against the reference's torch sequence: bit for bit
    tb = t[:, None, None]
    x0 = (tabs[0][tb] * x - tabs[1][tb] * eps.float()).clamp(-1, 1)
    ref = (tabs[2][tb] * x0 + tabs[3][tb] * x) + (0 if last else torch.exp(0.5 * tabs[4][tb]) * noise)
    assert torch.equal(x1, ref)
    if last:
        gx, gh0, gy = ops.sampler_head(s, wsp, bs, wop, bo, x, None, t, *tabs)
        assert gh0 is None and gy is None
    else:
        h0 = ops.conv1d(x1.to(dt), wip, bi, C, act="relu")
        _, yin0 = ops.diffnet_post_fwd(None, h0, None, ds0, init=True)
        gx, gh0, gy = ops.sampler_head(s, wsp, bs, wop, bo, x, noise, t, *tabs, win_p=wip, win_b=bi, ds0=ds0)
    torch.cuda.synchronize()
    differ = gx != x1
    assert float(differ.float().mean()) < 1e-3, float(differ.float().mean())
    bound = float(tabs[1].max()) * 2.0 ** -7 * float(eps.float().abs().max())  # |coef| x one bf16 ulp of the largest eps
    assert float((gx - x1).abs().max()) <= bound
    if not last:
        for a, b in ((gh0, h0), (gy, yin0)):
            d = (a.float() - b.float()).abs()
            assert float((d > 0).float().mean()) < 1e-3 and float(d.max()) <= 2.0 ** -6 * float(b.float().abs().max())


@pytest.mark.parametrize("shape,masked,dt", [((7, 333, 80), True, torch.bfloat16), ((7, 333, 80), True, torch.float32),
                                             ((5, 1201), False, torch.float32), ((3, 17, 80), True, torch.bfloat16)])
def test_masked_l1_mean_matches_the_tensor_expression_and_its_autograd(dev, shape, masked, dt):
    """ptpp_l1_masked_mean_fwd / _bwd (the L1 losses of reference models/prompttts_mdn_v2_final/model.py:126,138-170) against
    ((target - pred) * mask).abs().sum() / n / scale and torch autograd: the value to f32 summation-order accuracy, the gradient
    BIT FOR BIT (sign * mask * ((g / scale) / n) rounded once into pred's dtype); two calls give identical bits (fixed order)."""
    from promptttspp_amd import functional as PF

    g = torch.Generator().manual_seed(123)
    pred = torch.randn(*shape, generator=g).to(dev).to(dt).requires_grad_(True)
    target = torch.randn(*shape, generator=g).to(dev)
    target.view(-1)[::7] = pred.detach().float().view(-1)[::7]  # exact ties: sgn(0) = 0
    rows = pred.numel() // shape[-1]
    mask = (torch.rand(rows, generator=g) > 0.3).float().to(dev) if masked else None
    n = torch.tensor(float(rows) * 0.7, device=dev)
    scale = 6.0
    loss = PF.masked_l1_mean(pred, target, None if mask is None else mask.view(shape[:-1] + (1,)), n, scale)
    loss2 = PF.masked_l1_mean(pred, target, None if mask is None else mask.view(shape[:-1] + (1,)), n, scale)
    assert torch.equal(loss, loss2)
    (loss * 1.7).backward()
    got = pred.grad.clone()
    pred.grad = None
    p32 = pred.float()
    d = (target - p32)
    if mask is not None:
        d = d * mask.view(shape[:-1] + (1,))
    ref = d.abs().sum() / n / scale
    (ref * 1.7).backward()
    assert abs(float(loss) - float(ref)) <= 2e-6 * abs(float(ref))
    assert torch.equal(got, pred.grad)
