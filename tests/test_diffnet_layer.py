"""GPU: the one-launch DiffNet residual layer (csrc/diffnet_layer.hip, ptpp_diffnet_layer_fwd; reference
modules/denoiser.py:69-83) against (a) the two launches it replaces -- same accumulation order and rounding points, so every
output must be equal BIT FOR BIT -- and (b) the f32 oracle of the layer."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

C = 256


def _case(dev, B, T, seed, ldc_layers=3, layer=1):
    g = torch.Generator().manual_seed(1000 + seed)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    x = r(B, T, C).bfloat16()
    dstep = r(B, C).float()
    yin = (x.float() + dstep[:, None, :]).bfloat16()
    cond_all = r(B, T, ldc_layers * 2 * C, sc=0.7).bfloat16()  # the layer reads a slice with the row stride of all layers
    dil_w, dil_b = r(2 * C, C, 3, sc=0.04), r(2 * C, sc=0.1)
    out_w, out_b = r(2 * C, C, 1, sc=0.06), r(2 * C, sc=0.1)
    dnext = r(B, C).float()
    skip0 = r(B, T, C).float()
    return x, yin, cond_all, cond_all[:, :, layer * 2 * C:(layer + 1) * 2 * C], dil_w, dil_b, out_w, out_b, dnext, skip0


def _two_launch(PF, ops, x, yin, cond, dil_w, dil_b, out_w, out_b, dnext, skip, dil, init, lengths, save):
    perm = PF._gate_perm(2 * C, x.device)
    wp = ops.pack_conv_weight(dil_w[perm], torch.bfloat16)
    bp = dil_b[perm].contiguous()
    B, T, _ = x.shape
    g = torch.empty_like(x)
    a = None
    if save:
        a = torch.empty((B, T, 2 * C), device=x.device, dtype=x.dtype)
        ops.conv1d_gate_fwd_save(yin, wp, bp, C, 3, dil, dil, cond, g, a, lengths=lengths)
    else:
        ops.conv1d(yin, wp, bp, 2 * C, ks=3, dil=dil, pad=dil, act="gate", res=cond, out=g)
    xn, yn = ops.conv1d_diffnet_post(g, ops.pack_conv_weight(out_w, torch.bfloat16), out_b, x, skip, dnext, init=init, lengths=lengths,
                                     out_mask=lengths is not None)
    return xn, yn, a, g


def _one_launch(PF, ops, x, yin, cond, dil_w, dil_b, out_w, out_b, dnext, skip, dil, init, lengths, save, **kw):
    perm = PF._gate_perm(2 * C, x.device)
    wp = ops.pack_conv_weight(dil_w, torch.bfloat16, 2)
    assert torch.equal(wp, ops.pack_conv_weight(dil_w[perm], torch.bfloat16))
    ws = ops.diffnet_pack_wstream([wp], [ops.pack_conv_weight(out_w, torch.bfloat16)], C)
    return ops.diffnet_layer_fwd(yin, x, cond, ws[0], dil_b[perm].contiguous(), out_b, dnext, skip, dil, init, lengths=lengths, save=save, **kw)


@pytest.mark.parametrize("B,T,dil,masked,save,init", [
    (3, 300, 1, True, True, False),
    (2, 128, 2, False, True, True),      # exactly one tile per utterance
    (5, 517, 4, True, True, False),      # a ragged last tile, utterances ending inside and before tiles
    (4, 1000, 8, False, False, False),   # inference: no a / g, the gate from the unrounded pre-activation
    (1, 37, 8, False, False, True),      # shorter than the dilated window
    (7, 260, 8, True, True, True),
])
@pytest.mark.parametrize("bm", [64, 80, 96, 112, 128])
def test_one_launch_layer_is_bit_identical_to_the_two_launches(dev, monkeypatch, B, T, dil, masked, save, init, bm):
    """``bm``: rows per block (PTPP_DIFFNET_BM pins what the launcher otherwise picks from the block count: every
    instantiation must agree with the two-launch path on every shape; 80 and 112 rows exist for the inference forms of the
    default 1 x 8 kernel only -- with ``save`` the launcher ignores the request and the case repeats another height)."""
    from promptttspp_amd import functional as PF
    from promptttspp_amd import ops

    monkeypatch.setenv("PTPP_DIFFNET_BM", str(bm))

    if not ops.diffnet_layer_supported(C, torch.bfloat16):
        pytest.skip("one-launch DiffNet layer not built for this shape")
    x, yin, _, cond, dil_w, dil_b, out_w, out_b, dnext, skip0 = _case(dev, B, T, seed=dil + T)
    lengths = None
    if masked:  # includes an utterance whose last tiles lie wholly past its end
        lengths = torch.tensor([max(1, T - 150 * i) for i in range(B)], device=dev, dtype=torch.int32)
    s_ref, s_got = skip0.clone(), skip0.clone()
    ref = _two_launch(PF, ops, x, yin, cond, dil_w, dil_b, out_w, out_b, dnext, s_ref, dil, init, lengths, save)
    sc = torch.full_like(x, float("nan"))  # the skip projection's input, as the last layer of a stack writes it
    scale = 1.0 / math.sqrt(20)
    got = _one_launch(PF, ops, x, yin, cond, dil_w, dil_b, out_w, out_b, dnext, s_got, dil, init, lengths, save, skip_scaled=sc, skip_scale=scale)
    torch.cuda.synchronize()
    assert torch.equal(sc, (s_ref * scale).to(x.dtype))
    names = ["xn", "yin_next", "a", "g"]
    for n, a, b in zip(names, ref, got):
        if n in ("a", "g") and not save:
            assert b is None
            continue
        assert torch.equal(a, b), (n, float((a.float() - b.float()).abs().max()), int((a != b).sum()))
    assert torch.equal(s_ref, s_got), float((s_ref - s_got).abs().max())
    assert float(ref[0].float().abs().max()) > 0 and float(s_ref.abs().max()) > 0


def test_one_launch_layer_matches_the_f32_oracle(dev):
    """The reference arithmetic (modules/denoiser.py:69-83) in f32 on the CPU against the bf16 kernel: conv + conditioner,
    gate, output projection, residual / skip -- bf16 operand rounding is the only difference (tolerances 2e-2 of the
    output scale; a transposed fragment or a wrong channel map would be O(1))."""
    import torch.nn.functional as F
    from promptttspp_amd import functional as PF
    from promptttspp_amd import ops

    if not ops.diffnet_layer_supported(C, torch.bfloat16):
        pytest.skip("one-launch DiffNet layer not built for this shape")
    B, T, dil = 3, 333, 4
    x, yin, _, cond, dil_w, dil_b, out_w, out_b, dnext, skip0 = _case(dev, B, T, seed=11)
    skip = skip0.clone()
    xn, yn, a, g = _one_launch(PF, ops, x, yin, cond, dil_w, dil_b, out_w, out_b, dnext, skip, dil, False, None, True)
    torch.cuda.synchronize()
    # oracle: standard channel order; cond is stored gate-interleaved -> undo the permutation
    perm = PF._gate_perm(2 * C, x.device)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(2 * C, device=perm.device)
    cond_std = cond.float()[:, :, inv].cpu()
    bw = lambda w: w.bfloat16().float().cpu()
    y = F.conv1d(yin.float().cpu().transpose(1, 2), bw(dil_w), dil_b.cpu(), padding=dil, dilation=dil).transpose(1, 2) + cond_std
    a_ref = y
    g_ref = torch.sigmoid(y[:, :, :C]) * torch.tanh(y[:, :, C:])
    o = F.conv1d(g_ref.transpose(1, 2), bw(out_w), out_b.cpu()).transpose(1, 2)
    xn_ref = (x.float().cpu() + o[:, :, :C]) / math.sqrt(2.0)
    skip_ref = skip0.cpu() + o[:, :, C:]
    yn_ref = xn_ref + dnext.cpu()[:, None, :]

    def close(got, ref, tol):
        err = float((got.float().cpu() - ref).abs().max() / ref.abs().max())
        assert err < tol, err

    close(a, a_ref, 1e-2)
    close(g, g_ref, 2e-2)
    close(xn, xn_ref, 2e-2)
    close(yn, yn_ref, 2e-2)
    close(skip, skip_ref, 2e-2)


@pytest.mark.parametrize("B,T,dil,masked,bm", [(3, 333, 4, False, 128), (4, 517, 8, True, 96), (2, 100, 1, True, 64), (5, 260, 2, False, 128)])
def test_layer_with_the_conditioner_projection_inside(dev, monkeypatch, B, T, dil, masked, bm):
    """The COND instantiation (the 1 x 1 conditioner projection as 16 more stages of the first matrix pass, 80-stage operand
    stream, summed biases) against the launch that reads the precomputed slice: same arithmetic except that the projection is no
    longer rounded to bf16 before it joins the dilated conv -- pre-activation within one bf16 ulp of its scale, everything
    downstream within the bf16 tolerance; and against the f32 oracle at least as close as the slice form."""
    import torch.nn.functional as F
    from promptttspp_amd import functional as PF
    from promptttspp_amd import ops

    monkeypatch.setenv("PTPP_DIFFNET_BM", str(bm))
    x, yin, _, _, dil_w, dil_b, out_w, out_b, dnext, skip0 = _case(dev, B, T, seed=40 + dil)
    g = torch.Generator().manual_seed(77)
    condx = torch.randn(B, T, 256, generator=g).to(dev).bfloat16()
    cw, cb = (torch.randn(2 * C, 256, 1, generator=g) * 0.05).to(dev), (torch.randn(2 * C, generator=g) * 0.1).to(dev)
    lengths = torch.tensor([max(1, T - 90 * i) for i in range(B)], device=dev, dtype=torch.int32) if masked else None
    perm = PF._gate_perm(2 * C, x.device)
    dwp, owp = ops.pack_conv_weight(dil_w, torch.bfloat16, 2), ops.pack_conv_weight(out_w, torch.bfloat16)
    cwp = ops.pack_conv_weight(cw, torch.bfloat16, 2)
    # reference form: the slice (B, T, 2C) in the gate-interleaved order, rounded to bf16, bias included
    cond = ops.conv1d(condx, cwp, cb[perm].contiguous(), 2 * C)
    ws_ref = ops.diffnet_pack_wstream([dwp], [owp], C)
    s_ref, s_got = skip0.clone(), skip0.clone()
    ref = ops.diffnet_layer_fwd(yin, x, cond, ws_ref[0], dil_b[perm].contiguous(), out_b, dnext, s_ref, dil, False, lengths=lengths, save=True)
    ws = ops.diffnet_pack_wstream([dwp], [owp], C, cond_wps=[cwp])
    assert ws.shape[1] == ws_ref.shape[1] * 80 // 64
    got = ops.diffnet_layer_fwd(yin, x, None, ws[0], (dil_b + cb)[perm].contiguous(), out_b, dnext, s_got, dil, False, lengths=lengths,
                                save=True, condx=condx)
    torch.cuda.synchronize()
    # f32 oracle of the pre-activation and the gate
    bw = lambda w: w.bfloat16().float().cpu()
    a_or = (F.conv1d(yin.float().cpu().transpose(1, 2), bw(dil_w), dil_b.cpu(), padding=dil, dilation=dil)
            + F.conv1d(condx.float().cpu().transpose(1, 2), bw(cw), cb.cpu())).transpose(1, 2)
    if masked:
        keep = (torch.arange(T)[None, :] < lengths.cpu()[:, None])[:, :, None]
    names = ["xn", "yin_next", "a", "g"]
    for n, r, o in zip(names, ref, got):
        r, o = r.float(), o.float()
        if masked and n in ("a", "g"):
            # past an utterance's end the slice form keeps the bare conditioner value in a (and its gate in g), this form zeros:
            # both only ever meet the masked output projection / zero gradients
            r, o = r * keep.to(dev), o * keep.to(dev)
        scale = float(r.abs().max())
        err = float((r - o).abs().max()) / scale
        assert err < (1.2e-2 if n == "a" else 2.5e-2), (n, err)
    assert float((s_ref - s_got).abs().max() / s_ref.abs().max()) < 2.5e-2
    a_ref_err = (ref[2].float().cpu() - a_or).abs()
    a_got_err = (got[2].float().cpu() - a_or).abs()
    if masked:  # rows past an utterance's end hold the conditioner-only value in both forms; compare the valid rows
        a_ref_err, a_got_err = a_ref_err * keep, a_got_err * keep
    assert float(a_got_err.max()) <= float(a_ref_err.max()) * 1.05 + 1e-6  # one rounding instead of two
    assert float(a_got_err.mean()) < float(a_ref_err.mean())


@pytest.mark.parametrize("B,T,dil,masked,bm,cond_inside,save", [
    (3, 333, 4, False, 128, False, True), (4, 517, 8, True, 96, False, False), (2, 100, 1, True, 64, False, True),
    (3, 333, 2, True, 128, True, True), (4, 517, 8, False, 96, True, False), (2, 100, 1, True, 64, True, True)])
def test_the_three_weight_paths_are_bit_identical(dev, monkeypatch, B, T, dil, masked, bm, cond_inside, save):
    """PTPP_DIFFNET_GW = 0 (weight stages through the LDS ring, 2 x 4 wave grid), 1 (weight fragments straight from global
    memory, 2 x 4) and 2 (the default: straight from global memory on the 1 x 8 wave grid) read the SAME operand stream and
    issue the same MFMAs in the same K order: every output must be equal bit for bit, with and without the conditioner
    projection inside the launch."""
    from promptttspp_amd import functional as PF
    from promptttspp_amd import ops

    monkeypatch.setenv("PTPP_DIFFNET_BM", str(bm))
    x, yin, cond_all, cond, dil_w, dil_b, out_w, out_b, dnext, skip0 = _case(dev, B, T, seed=90 + dil)
    g = torch.Generator().manual_seed(78)
    lengths = torch.tensor([max(1, T - 70 * i) for i in range(B)], device=dev, dtype=torch.int32) if masked else None
    perm = PF._gate_perm(2 * C, x.device)
    dwp, owp = ops.pack_conv_weight(dil_w, torch.bfloat16, 2), ops.pack_conv_weight(out_w, torch.bfloat16)
    if cond_inside:
        condx = torch.randn(B, T, 256, generator=g).to(dev).bfloat16()
        cwp = ops.pack_conv_weight((torch.randn(2 * C, 256, 1, generator=g) * 0.05).to(dev), torch.bfloat16, 2)
        ws = ops.diffnet_pack_wstream([dwp], [owp], C, cond_wps=[cwp])
        kw = dict(condx=condx)
        cnd = None
    else:
        ws = ops.diffnet_pack_wstream([dwp], [owp], C)
        kw = {}
        cnd = cond
    outs = []
    for mode in ("0", "1", "2"):
        monkeypatch.setenv("PTPP_DIFFNET_GW", mode)
        sk = skip0.clone()
        o = ops.diffnet_layer_fwd(yin, x, cnd, ws[0], dil_b[perm].contiguous(), out_b, dnext, sk, dil, False, lengths=lengths, save=save, **kw)
        torch.cuda.synchronize()
        outs.append([t for t in o if t is not None] + [sk])
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)


def test_stack_with_folded_conditioner_matches_the_slice_form(dev, monkeypatch):
    """DiffNetStackFn in training, conditioner projected inside the layer launches (the default) against the (B, T, L * 2C)
    slice form: output and every gradient within the bf16 tolerance of one another."""
    from promptttspp_amd import functional as PF
    from test_stack_drivers import _run, _stack_case

    B, T, L = 4, 421, 6
    h0, cond, dsteps, lengths, params = _stack_case(dev, B, T, C, L, torch.bfloat16, True, seed=23)
    gout = (torch.randn(B, T, C, generator=torch.Generator().manual_seed(5)) * 0.5).to(dev).bfloat16()
    monkeypatch.setattr(PF, "DIFFNET_FOLD_COND", False)
    ref = _run(PF, h0, cond, dsteps, lengths, params, 4, gout)
    monkeypatch.setattr(PF, "DIFFNET_FOLD_COND", True)
    assert PF.diffnet_fold_cond_ok(h0, cond, 4)
    got = _run(PF, h0, cond, dsteps, lengths, params, 4, gout)
    again = _run(PF, h0, cond, dsteps, lengths, params, 4, gout)
    for i, (a, b, c) in enumerate(zip(ref, got, again)):
        assert torch.equal(b, c), i  # reproducible
        err = float((a.float() - b.float()).abs().max() / a.float().abs().max())
        assert err < 3e-2, (i, err)


def test_stack_driver_takes_the_one_launch_layer(dev, monkeypatch):
    """ptpp_diffnet_stack_fwd with the operand stream (training form, 20 layers, all four dilations, ragged batch) against the
    same driver without it: skip sum and every saved slab bit-identical; the stream follows a weight update."""
    from promptttspp_amd import functional as PF
    from test_stack_drivers import _stack_case

    B, T, L = 4, 421, 8
    h0, cond, dsteps, lengths, params = _stack_case(dev, B, T, C, L, torch.bfloat16, True, seed=21)
    ws = params

    def run():
        with torch.no_grad():
            gate_b, cond_b = PF.gate_biases([w[1] for w in ws], [w[3] for w in ws])
            cond_all, _ = PF.diffnet_cond_all(cond, [w[2] for w in ws], [w[3] for w in ws], bias_perm=cond_b)
            return PF.diffnet_stack_forward(h0, cond_all, dsteps, [(w[0], w[1], w[4], w[5]) for w in ws], lengths, 4, save=True, gate_b=gate_b)

    for rnd in range(2):
        monkeypatch.setattr(PF, "DIFFNET_LAYER_KERNEL", False)
        skip_ref, saved_ref = run()
        monkeypatch.setattr(PF, "DIFFNET_LAYER_KERNEL", True)
        skip_got, saved_got = run()
        torch.cuda.synchronize()
        assert torch.equal(skip_ref, skip_got)
        for a, b in zip(saved_ref, saved_got):
            assert torch.equal(a, b)
        with torch.no_grad():  # ... and the scaled form the callers take (one-launch: from the last layer's tail)
            gate_b, cond_b = PF.gate_biases([w[1] for w in ws], [w[3] for w in ws])
            cond_all, _ = PF.diffnet_cond_all(cond, [w[2] for w in ws], [w[3] for w in ws], bias_perm=cond_b)
            sc, _ = PF.diffnet_stack_forward(h0, cond_all, dsteps, [(w[0], w[1], w[4], w[5]) for w in ws], lengths, 4, save=True,
                                             gate_b=gate_b, scaled=True)
        assert torch.equal(sc, (skip_ref * (1.0 / math.sqrt(L))).to(h0.dtype))
        with torch.no_grad():  # the cached stream must follow the weights
            for w in ws:
                w[0].mul_(1.25)
                w[4].add_(0.01)
