"""GPU: the training-step glue kernels (csrc/glue.hip, round 6) against the tensor-op chains they replace and torch autograd,
the deferred finishing of parameter-gradient sums (csrc/red.hip), and the fused training forward against the general one."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _mdn_ref(y, G, D, target, lp_min=-7.0, ls_min=-7.0):
    """reference modules/mdn.py:37-66 (heads -> log-softmax) + :81-175 (dimension-wise NLL) on the raw head output."""
    from promptttspp_amd.modules import mdn as M

    B, T = y.shape[:2]
    n = G * D
    log_pi = F.log_softmax(y[..., :n].reshape(B, T, G, D), dim=2)
    old = M.FUSED_NLL
    M.FUSED_NLL = False
    try:
        return M.mdn_loss(log_pi, y[..., n:2 * n].reshape(B, T, G, D), y[..., 2 * n:].reshape(B, T, G, D), target, lp_min, ls_min,
                          reduce=False)
    finally:
        M.FUSED_NLL = old


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_tts_losses_match_the_tensor_expressions_and_autograd(dev, dt):
    """ptpp_tts_losses_fwd / _bwd against reference models/prompttts_mdn_v2_final/model.py:126-183 written with tensor ops:
    values to summation-order accuracy, gradients of the L1 parts BIT FOR BIT (sign * coefficient rounded once), the MDN parts to
    1e-5 (the log-softmax backward is fused); two calls give identical bits."""
    from promptttspp_amd import functional as PF

    g = torch.Generator().manual_seed(7)
    B, Tf, Tp, M, Gd, Gs, D = 5, 203, 31, 80, 4, 10, 256
    flen = torch.tensor([203, 150, 7, 99, 1], dtype=torch.int32)
    plen = torch.tensor([31, 20, 3, 17, 1], dtype=torch.int32)
    fm = (torch.arange(Tf)[None] < flen[:, None]).float()
    pm = torch.arange(Tp)[None] < plen[:, None]
    pred = torch.randn(B, Tf, M, generator=g).to(dt)
    noise = torch.randn(B, Tf, M, generator=g)
    noise.view(-1)[::11] = pred.float().view(-1)[::11]  # exact ties
    pv = (torch.randn(B, Tf, 2, generator=g) * fm[..., None]).to(dt)
    cf0 = torch.randn(B, Tf, generator=g) * fm
    vuv = (torch.rand(B, Tf, generator=g) > 0.4).float() * fm
    y_dur = torch.randn(B, Tp, 3 * Gd, generator=g)
    y_dur[0, 0, Gd:2 * Gd] = -9.0  # below the log-sigma clamp
    y_dur[1, 1, 0] = -30.0  # a component far below the log-pi clamp
    dur = (torch.randint(1, 20, (B, Tp), generator=g).float()) * pm
    y_sty = torch.randn(B, 1, 3 * Gs * D, generator=g) * 0.7
    sty = F.normalize(torch.randn(B, D, generator=g), dim=1)
    sty[0, :5] = 40.0  # outside 5 sigma: the clipped regime
    to = lambda t: t.to(dev)
    leaves = [to(t).requires_grad_(True) for t in (pred, pv, y_dur, y_sty)]
    args = (to(noise), to(flen), to(cf0), to(vuv), to(dur), to(plen), to(sty))
    total, comps = PF.tts_losses(*leaves, *args, Gd, Gs, 8.0)
    total2, comps2 = PF.tts_losses(*leaves, *args, Gd, Gs, 8.0)
    assert torch.equal(total, total2) and torch.equal(comps, comps2)
    (total * 1.3 + comps[1] * 0.5).backward()
    got = [l.grad.clone() for l in leaves]
    for l in leaves:
        l.grad = None
    # the tensor-op form
    p, v, yd, ys = leaves
    nf = to(fm).sum()
    dec = ((args[0] - p.float()) * to(fm)[..., None]).abs().sum() / nf / 8.0
    l_cf0 = (v.float()[..., 0] - args[2]).abs().sum() / nf
    l_vuv = (v.float()[..., 1] - args[3]).abs().sum() / nf
    d = args[4]
    log_d = torch.where(d != 0, torch.log(d.clamp_min(1e-30)), d)
    nll = _mdn_ref(yd, Gd, 1, log_d.unsqueeze(-1))
    pmb = to(pm).unsqueeze(-1)
    l_dur = torch.where(pmb, nll, torch.zeros_like(nll)).sum() / pmb.sum()
    l_sty = _mdn_ref(ys, Gs, D, args[6].unsqueeze(1)).mean()
    ref_total = dec + l_dur + l_cf0 + l_vuv + l_sty
    (ref_total * 1.3 + l_dur * 0.5).backward()
    ref = [dec, l_dur, l_cf0, l_vuv, l_sty]
    for i, r in enumerate(ref):
        assert abs(float(comps[i]) - float(r)) <= 3e-6 * max(1.0, abs(float(r))), (i, float(comps[i]), float(r))
    assert abs(float(total) - float(ref_total)) <= 3e-6 * abs(float(ref_total))
    assert float(comps[5]) == float(flen.sum()) and float(comps[6]) == float(plen.sum())
    assert torch.equal(got[0], p.grad), "decoder L1 gradient"
    assert torch.equal(got[1], v.grad), "pitch / V-UV gradient"
    for k in (2, 3):
        e = (got[k] - leaves[k].grad).abs().max() / leaves[k].grad.abs().max()
        assert float(e) < 1e-5, (k, float(e))
    assert float(got[2][2, 5:].abs().max()) == 0.0  # padded phones: zero gradient


def test_q_sample_on_the_dataset_layout_is_bit_identical_to_the_tensor_ops(dev):
    """ptpp_q_sample_bct against modules/diffusion.py::q_sample(_norm(mel_cl)) followed by the cast (reference
    diffusion.py:97-101, 304-313): same f32 operation sequence, so the bits agree -- both normalisations, ragged T."""
    from promptttspp_amd.modules.diffusion import GaussianDiffusion
    from promptttspp_amd.modules.denoiser import DiffNet
    from promptttspp_amd import ops

    g = torch.Generator().manual_seed(3)
    for norm_scale in (6.0, None):
        dif = GaussianDiffusion(256, 80, DiffNet(80, 256, 2, 256, 3, 2), norm_scale=norm_scale, a_min=-11.0, a_max=2.5).to(dev)
        for B, T in ((3, 131), (1, 64), (2, 7)):
            mel = (torch.randn(B, 80, T, generator=g) * 2 - 5).to(dev)
            noise = torch.randn(B, T, 80, generator=g).to(dev)
            t = torch.randint(0, 100, (B,), generator=g).to(dev)
            ref = dif.q_sample(dif._norm(mel.transpose(1, 2).float().contiguous()), t, noise)
            for dt in (torch.float32, torch.bfloat16):
                got = ops.q_sample_bct(mel, noise, t, dif.sqrt_alphas_cumprod, dif.sqrt_one_minus_alphas_cumprod, dif.norm_scale,
                                       dif.a_min, dif.a_max, dt)
                assert torch.equal(got, ref.to(dt)), (norm_scale, B, T, dt)


def test_step_sinusoid_and_mish(dev):
    """ptpp_step_sinusoid / ptpp_mish_* against the tensor ops of reference modules/denoiser.py:23-41 and autograd."""
    from promptttspp_amd import functional as PF
    from promptttspp_amd import ops

    t = torch.tensor([0, 1, 17, 50, 99], device=dev)
    half = 128
    f = torch.exp(torch.arange(half, device=dev) * -(math.log(10000) / (half - 1)))
    e = 1 * t[:, None] * f[None, :]
    ref = torch.cat((e.sin(), e.cos()), dim=-1)
    got = ops.step_sinusoid(t, 256, 1)
    assert float((got - ref).abs().max()) < 2e-6
    x = (torch.randn(7, 1024, generator=torch.Generator().manual_seed(1)) * 6).to(dev)
    x[0, :3] = torch.tensor([25.0, -30.0, 0.0])
    x.requires_grad_(True)
    y = PF.mish(x)
    y.backward(torch.ones_like(y) * 0.7)
    gx = x.grad.clone()
    x.grad = None
    yr = x * torch.tanh(F.softplus(x))
    yr.backward(torch.ones_like(yr) * 0.7)
    assert float((y - yr).abs().max()) < 1e-5 and float((gx - x.grad).abs().max()) < 1e-5


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("do_scale", [False, True])
def test_phoneme_embedding_kernel(dev, dt, do_scale):
    """ptpp_embed_cl_fwd / _bwd against nn.Embedding(padding_idx=0) -> scale -> mask -> cast (reference
    layers/embedding.py:21-36) and its autograd; the table gradient is reproducible bit for bit."""
    from promptttspp_amd.layers.embedding import PhonemeEmbedding

    torch.manual_seed(0)
    m = PhonemeEmbedding(90, 256, do_scale=do_scale, init_normal=True).to(dev)
    B, T = 6, 300
    lens = torch.tensor([300, 299, 1, 77, 256, 130], dtype=torch.int32, device=dev)
    ids = torch.randint(1, 90, (B, T), device=dev)
    mask = torch.arange(T, device=dev)[None] < lens[:, None]
    ids = ids * mask
    y = m.forward_cl(ids, None, dt, lengths=lens)
    w = torch.randn(B, T, 256, device=dev).to(dt)
    (y.float() * w.float()).sum().backward()
    g1 = m.emb.weight.grad.clone()
    m.emb.weight.grad = None
    m2 = PhonemeEmbedding(90, 256, do_scale=do_scale, init_normal=True)
    m2.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    y2 = m2.forward_cl(ids.cpu(), mask.unsqueeze(-1).float().cpu(), dt)  # the CPU branch: the tensor-op form
    assert torch.equal(y.detach().cpu(), y2.detach())
    (y2.float() * w.float().cpu()).sum().backward()
    e = (g1.cpu() - m2.emb.weight.grad).abs().max() / m2.emb.weight.grad.abs().max()
    assert float(e) < (2e-2 if dt == torch.bfloat16 else 1e-5)  # (bf16: the tensor-op chain rounds the gradient to bf16 before the sum)
    assert float(g1[0].abs().max()) == 0.0  # padding row
    y = m.forward_cl(ids, None, dt, lengths=lens)
    (y.float() * w.float()).sum().backward()
    assert torch.equal(g1, m.emb.weight.grad)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_scalar_embedding_add_and_its_parameter_gradients(dev, dt):
    """ptpp_scalar_embed_add / _bwd against h + ((track * w + b) * mask).to(dtype) (reference
    modules/variance_adaptor.py:139-146, pitch_emb = Conv1d(1, C, 1)) and autograd."""
    from promptttspp_amd import functional as PF

    torch.manual_seed(1)
    B, T, C = 4, 517, 256
    emb = torch.nn.Conv1d(1, C, 1).to(dev)
    lens = torch.tensor([517, 300, 1, 64], dtype=torch.int32, device=dev)
    fm = (torch.arange(T, device=dev)[None] < lens[:, None]).unsqueeze(-1).float()
    h = torch.randn(B, T, C, device=dev).to(dt).requires_grad_(True)
    track = torch.randn(B, T, device=dev) + 5.0
    y = PF.scalar_embed_add(h, track, emb, lens)
    go = torch.randn(B, T, C, device=dev).to(dt)
    y.backward(go)
    got = (h.grad.clone(), emb.weight.grad.clone(), emb.bias.grad.clone())
    h.grad = emb.weight.grad = emb.bias.grad = None
    ref = h + ((track.unsqueeze(-1) * emb.weight.reshape(1, 1, -1) + emb.bias.reshape(1, 1, -1)) * fm).to(dt)
    assert torch.equal(y, ref)
    ref.backward(go)
    assert torch.equal(got[0], h.grad)
    tol = 2e-2 if dt == torch.bfloat16 else 2e-5
    for a, b in ((got[1], emb.weight.grad), (got[2], emb.bias.grad)):
        assert float((a - b).abs().max() / b.abs().max()) < tol


def test_l2_normalize_and_broadcast_add(dev):
    """ptpp_l2norm_* against F.normalize(dim=1) (reference model.py:108,148-150) incl. the clamped regime, and
    ptpp_bcast_add_rows / ptpp_rows_sum against x + e.transpose(1, 2).to(dtype) (model.py:111) with autograd."""
    from promptttspp_amd import functional as PF

    torch.manual_seed(2)
    e = torch.randn(5, 256, 1, device=dev)
    e[1] = 0.0
    e[2] *= 1e-14
    e.requires_grad_(True)
    y = PF.l2_normalize_channels(e)
    go = torch.randn_like(y)
    y.backward(go)
    g = e.grad.clone()
    e.grad = None
    yr = F.normalize(e, dim=1)
    yr.backward(go)
    assert float((y - yr).abs().max()) < 1e-6
    assert float((g - e.grad).abs().max() / e.grad.abs().max()) < 1e-5
    for dt in (torch.bfloat16, torch.float32):
        x = torch.randn(5, 77, 256, device=dev).to(dt).requires_grad_(True)
        s = torch.randn(5, 256, 1, device=dev, requires_grad=True)
        out = PF.bcast_add_rows(x, s.reshape(5, 256))
        gg = torch.randn_like(out)
        out.backward(gg)
        gx, gs = x.grad.clone(), s.grad.clone()
        x.grad = s.grad = None
        ref = x + s.transpose(1, 2).to(dt)
        ref.backward(gg)
        assert torch.equal(out, ref) and torch.equal(gx, x.grad)
        assert float((gs - s.grad).abs().max() / s.grad.abs().max()) < (2e-2 if dt == torch.bfloat16 else 1e-5)


def test_durations_cumsum(dev):
    from promptttspp_amd import ops

    g = torch.Generator().manual_seed(5)
    for Tp in (1, 63, 64, 65, 260):
        d = torch.randint(0, 30, (7, Tp), generator=g)
        ref = torch.cumsum(d, dim=1).to(torch.int32)
        assert torch.equal(ops.durations_cumsum(d.to(dev)).cpu(), ref)
        assert torch.equal(ops.durations_cumsum(d.float().to(dev)).cpu(), ref)
    big = torch.full((1, 70), 2**30, dtype=torch.int64)
    assert int(ops.durations_cumsum(big.to(dev))[0, -1]) == 2**31 - 1


def test_deferred_parameter_sums_equal_the_immediate_ones(dev):
    """include/ptpp.h "Deferred reduction": with ptpp_red_defer on, the LayerNorm parameter gradients are complete only after
    ptpp_red_flush -- and then bit-identical to the immediate form for the same replica assignment ... up to the atomics' order;
    here: equal to 1e-6, queued count as expected, nothing queued for buffers handed back to autograd."""
    from promptttspp_amd import _lib, ops

    lib = _lib.load()
    torch.manual_seed(3)
    B, T, C = 3, 257, 256
    x = torch.randn(B, T, C, device=dev).bfloat16()
    dy = torch.randn(B, T, C, device=dev).bfloat16()
    gam = torch.randn(C, device=dev)
    bet = torch.randn(C, device=dev)
    y, mean, rstd, _ = ops.layernorm_fwd(x, gam, bet, 1e-5, save_stats=True)
    dg0, db0 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ops.layernorm_bwd(dy, x, gam, mean, rstd, dgamma_out=dg0, dbeta_out=db0)
    assert not ops.red_deferred()
    ops.red_defer_enable(dev)
    try:
        dg1, db1 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        for _ in range(3):  # three queued sums into the same destination
            ops.layernorm_bwd(dy, x, gam, mean, rstd, dgamma_out=dg1, dbeta_out=db1)
        assert lib.ptpp_red_pending() == 3
        torch.cuda.synchronize()
        assert float(dg1.abs().max()) == 0.0  # nothing delivered yet
        _, _, dg2, db2 = ops.layernorm_bwd(dy, x, gam, mean, rstd)  # fresh buffers: finished at once
        assert lib.ptpp_red_pending() == 3
        assert float((dg2 - dg0).abs().max() / dg0.abs().max()) < 1e-6
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        dg3, db3 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        with torch.cuda.stream(side):  # a producer on another stream: its own sub-arena, finished on its own stream
            ops.layernorm_bwd(dy, x, gam, mean, rstd, dgamma_out=dg3, dbeta_out=db3)
        assert lib.ptpp_red_pending() == 4
        ops.red_flush()
        assert lib.ptpp_red_pending() == 0
        torch.cuda.synchronize()
        assert float((dg1 - 3 * dg0).abs().max() / dg0.abs().max()) < 3e-6 and float((db1 - 3 * db0).abs().max() / db0.abs().max()) < 3e-6
        assert float((dg3 - dg0).abs().max() / dg0.abs().max()) < 1e-6
        # the arena was left zero: a second round gives the same sums
        dg4, db4 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        ops.layernorm_bwd(dy, x, gam, mean, rstd, dgamma_out=dg4, dbeta_out=db4)
        ops.red_flush()
        assert float((dg4 - dg0).abs().max() / dg0.abs().max()) < 1e-6
    finally:
        ops.red_defer_disable()
    assert not ops.red_deferred()


@pytest.mark.parametrize("dt,ltol,gtol", [(torch.float32, 1e-5, 2e-4), (torch.bfloat16, 2e-3, 6e-2)])
def test_fused_training_forward_equals_the_general_one(dev, dt, ltol, gtol):
    """PromptTTSMDNDurCFG._forward_fused against the general ``forward`` (tensor-op glue) on the golden model, eval mode (no
    dropout): f32 -- same losses to 1e-5, same parameter gradients to 2e-4 of each tensor's largest entry; bf16 -- the two forms
    round at the same places except the f32 loss island, losses to 2e-3."""
    import test_hip_acoustic as T
    from promptttspp_amd import config
    from promptttspp_amd import functional as PF
    from promptttspp_amd.models.prompttts_mdn_v2_final import model as MM

    config.set_compute_dtype(dt)
    PF.clear_caches()
    model, g = T._model(dev)
    model.eval()

    def run(fused):
        MM.FUSED_GLUE = fused
        model.zero_grad(set_to_none=True)
        model.decoder.injected = {"t": g["t"], "noise": g["noise"]}
        out = model(T._batch(g, dev))
        out["loss"].backward()
        return {k: float(v) for k, v in out.items()}, {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    try:
        lf, gf = run(True)
        lg, gg = run(False)
    finally:
        MM.FUSED_GLUE = True
        config.set_compute_dtype(torch.float32)
        PF.clear_caches()
    for k in lg:
        assert abs(lf[k] - lg[k]) <= ltol * max(1.0, abs(lg[k])), (k, lf[k], lg[k])
    assert set(gf) == set(gg)
    worst = max((float((gf[n] - gg[n]).abs().max() / (gg[n].abs().max() + 1e-6)), n) for n in gg)
    assert worst[0] < gtol, worst


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("cout", [1, 2, 4])
def test_small_output_linear_matches_conv1d_and_autograd(dev, dt, cout):
    """ptpp_linear_small_fwd / _bwd (the pitch / V-UV head, reference modules/variance_adaptor.py:52-62: Conv1d(256 -> 2, k = 1)
    followed by the frame mask) against F.conv1d * mask and its autograd on the same rounded operands."""
    from promptttspp_amd import functional as PF

    torch.manual_seed(4)
    B, T, C = 3, 301, 256
    lay = torch.nn.Conv1d(C, cout, 1).to(dev)
    lens = torch.tensor([301, 17, 256], dtype=torch.int32, device=dev)
    fm = (torch.arange(T, device=dev)[None] < lens[:, None]).unsqueeze(-1).float()
    x = torch.randn(B, T, C, device=dev).to(dt).requires_grad_(True)
    assert PF.linear_small_ok(x, lay.weight)
    y = PF.linear_small(x, lay.weight, lay.bias, lens)
    go = torch.randn(B, T, cout, device=dev).to(dt)
    y.backward(go)
    got = (x.grad.clone(), lay.weight.grad.clone(), lay.bias.grad.clone())
    x.grad = lay.weight.grad = lay.bias.grad = None
    ref = (F.conv1d(x.float().transpose(1, 2), lay.weight, lay.bias).transpose(1, 2) * fm)
    ref.backward(go.float())
    tol = 1e-2 if dt == torch.bfloat16 else 2e-5
    assert float((y.float() - ref).abs().max() / ref.abs().max()) < tol
    assert float(y[1, 17:].abs().max()) == 0.0 and float(got[0][1, 17:].abs().max()) == 0.0
    for a, b in zip(got, (x.grad, lay.weight.grad, lay.bias.grad)):
        assert float((a.float() - b.float()).abs().max() / b.float().abs().max()) < tol
