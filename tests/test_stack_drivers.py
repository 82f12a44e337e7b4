"""GPU: the whole-unit drivers of the C ABI (include/ptpp.h "Whole-unit drivers") against the per-launch path they
replace -- the same kernels in the same order, so every output and gradient must be equal BIT FOR BIT."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rnd(seed, *shape, scale=1.0):
    return torch.from_numpy((scale * np.random.default_rng(seed).standard_normal(shape)).astype(np.float32))


@pytest.fixture(autouse=True)
def _no_conditioner_fold(monkeypatch):
    """The bit-identity tests below compare the one-call drivers with the per-launch path, which adds a precomputed (rounded)
    conditioner slice; the drivers' default in training -- each layer projecting the conditioner input inside its launch, f32
    accumulation -- is compared against that form with a tolerance in test_diffnet_layer.py."""
    from promptttspp_amd import functional as PF

    monkeypatch.setattr(PF, "DIFFNET_FOLD_COND", False)


def _stack_case(dev, B, T, C, L, dtype, masked, seed=0):
    g = torch.Generator().manual_seed(100 + seed)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    h0 = r(B, T, C).to(dtype)
    cond = r(B, T, C).to(dtype)
    dsteps = r(B, L, C).float()
    lengths = torch.tensor([max(3, T - 13 * i) for i in range(B)], device=dev, dtype=torch.int32) if masked else None
    params = []
    for l in range(L):
        params.append([torch.nn.Parameter(t) for t in (r(2 * C, C, 3, sc=0.05), r(2 * C, sc=0.1), r(2 * C, C, 1, sc=0.06), r(2 * C, sc=0.1),
                                                      r(2 * C, C, 1, sc=0.06), r(2 * C, sc=0.1))])
    return h0, cond, dsteps, lengths, params


def _run(PF, h0, cond, dsteps, lengths, params, cycle, gout):
    h0 = h0.clone().requires_grad_(True)
    cond = cond.clone().requires_grad_(True)
    ds = dsteps.clone().requires_grad_(True)
    for lp in params:
        for p in lp:
            p.grad = None
    y = PF.diffnet_stack(h0, cond, ds, lengths, cycle, params)
    y.backward(gout)
    torch.cuda.synchronize()
    return [y.detach(), h0.grad, cond.grad, ds.grad] + [p.grad.clone() for lp in params for p in lp]


@pytest.mark.parametrize("B,T,C,L,dtype,masked", [
    (3, 150, 128, 5, torch.bfloat16, True),
    (6, 700, 256, 6, torch.bfloat16, True),      # >= 192 128-row tiles: the LDS-DMA kernels and both fused epilogues
    (2, 97, 64, 4, torch.float32, True),         # f32 parity mode: no fused tail / gate backward, o_buf and dg_buf in use
    (4, 333, 256, 3, torch.bfloat16, False),
])
def test_diffnet_stack_driver_is_bit_identical_to_the_per_launch_path(dev, monkeypatch, B, T, C, L, dtype, masked):
    """ptpp_diffnet_stack_fwd / _bwd (reference modules/denoiser.py:69-83,136-140 and its autograd) against the loop of
    single launches in functional.DiffNetStackFn: output, data gradients (h0, cond, step projections) and all 6 L parameter
    gradients equal bit for bit, through autograd's own accumulation (no direct-gradient mode here)."""
    from promptttspp_amd import functional as PF

    h0, cond, dsteps, lengths, params = _stack_case(dev, B, T, C, L, dtype, masked)
    gout = rnd(7, B, T, C).to(dev).to(dtype)
    monkeypatch.setattr(PF, "BATCHED_WGRAD", False)  # (the batched weight gradient sums in another order: next test)
    monkeypatch.setattr(PF, "STACK_DRIVERS", False)
    ref = _run(PF, h0, cond, dsteps, lengths, params, 4, gout)
    monkeypatch.setattr(PF, "STACK_DRIVERS", True)
    got = _run(PF, h0, cond, dsteps, lengths, params, 4, gout)
    assert len(ref) == len(got) == 4 + 6 * L
    for i, (a, b) in enumerate(zip(ref, got)):
        assert a.shape == b.shape and a.dtype == b.dtype
        if i >= 4 and dtype == torch.float32:
            # the exact-f32 weight-gradient kernel combines its row slices with f32 atomics (in both paths): last bits
            # depend on the arrival order
            assert torch.allclose(a, b, rtol=1e-5, atol=2e-6 * float(a.abs().max())), i
        else:
            assert torch.equal(a, b), (i, float((a.float() - b.float()).abs().max()))
    assert float(ref[0].float().abs().max()) > 0 and all(float(t.float().abs().max()) > 0 for t in ref[1:4])


@pytest.mark.parametrize("B,T,L,masked", [(6, 700, 6, True), (3, 333, 5, False), (5, 130, 9, True)])
def test_diffnet_stack_backward_takes_dout_from_the_data_gradient_epilogue(dev, monkeypatch, B, T, L, masked):
    """Where the row-tile kernel runs the dilated data gradients (frame-level row counts; forced here), the driver has no
    per-layer pass over gx / gS: the residual half of each layer's dout leaves the epilogue of the layer above
    (ptpp_conv1d_rt_fwd_aux) and one launch fills all skip halves (ptpp_diffnet_post_bwd_fill).  Everything equal bit for bit
    to the per-launch path, whose dout comes from ptpp_diffnet_post_bwd."""
    from promptttspp_amd import functional as PF
    from promptttspp_amd import ops

    monkeypatch.setattr(ops, "CONV_RT_MIN_ROWS", 1)
    monkeypatch.setenv("PTPP_CONV_RT_MIN_ROWS", "1")
    C = 256
    h0, cond, dsteps, lengths, params = _stack_case(dev, B, T, C, L, torch.bfloat16, masked, seed=31)
    gout = rnd(9, B, T, C).to(dev).bfloat16()
    monkeypatch.setattr(PF, "BATCHED_WGRAD", False)
    monkeypatch.setattr(PF, "STACK_DRIVERS", False)
    ref = _run(PF, h0, cond, dsteps, lengths, params, 4, gout)
    monkeypatch.setattr(PF, "STACK_DRIVERS", True)
    got = _run(PF, h0, cond, dsteps, lengths, params, 4, gout)
    for i, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a, b), (i, float((a.float() - b.float()).abs().max()))
    monkeypatch.setattr(PF, "BATCHED_WGRAD", True)  # the batched weight gradients read the same dout slabs
    bat = _run(PF, h0, cond, dsteps, lengths, params, 4, gout)
    for i, (a, b) in enumerate(zip(ref, bat)):
        if i < 4:
            assert torch.equal(a, b), i
        else:
            assert float((a.double() - b.double()).abs().max() / a.double().abs().max()) < 2e-5, i


def test_stack_driver_refuses_an_operand_stream_it_would_not_use(dev, monkeypatch):
    """The row-tile decision is made on both sides of the C ABI (ops.conv1d_rt_ok / stacks.cpp::rt_takes) from the SAME
    threshold (PTPP_CONV_RT_MIN_ROWS, pushed to the C side by ops.conv_rt_min_rows whenever it changes).  Should the two ever
    disagree -- forced here by setting the C side's threshold behind Python's back -- the
    driver must refuse the operand stream instead of reading it as a [Cout][ks][Cin] operand (silently wrong
    activations and gradients)."""
    from promptttspp_amd import _lib
    from promptttspp_amd import functional as PF
    from promptttspp_amd import ops

    B, T, C, L = 3, 333, 256, 5
    h0, cond, dsteps, lengths, params = _stack_case(dev, B, T, C, L, torch.bfloat16, False, seed=5)
    gout = rnd(9, B, T, C).to(dev).bfloat16()
    monkeypatch.setattr(PF, "BATCHED_WGRAD", False)
    monkeypatch.setattr(PF, "STACK_DRIVERS", True)
    monkeypatch.setenv("PTPP_CONV_RT_MIN_ROWS", "1")
    assert ops.conv_rt_min_rows() == 1                       # the Python side follows the variable
    ok = _run(PF, h0, cond, dsteps, lengths, params, 4, gout)
    real = ops.conv_rt_min_rows
    monkeypatch.setattr(ops, "conv_rt_min_rows", lambda: 1)  # Python: row-tile ...
    _lib.load().ptpp_conv_rt_set_min_rows(10 ** 9)           # ... C (threshold pushed behind Python's back): tile kernel
    try:
        with pytest.raises(_lib.PtppError, match="operand stream"):
            _run(PF, h0, cond, dsteps, lengths, params, 4, gout)
    finally:
        _lib.load().ptpp_conv_rt_set_min_rows(1)
    monkeypatch.setattr(ops, "conv_rt_min_rows", real)
    again = _run(PF, h0, cond, dsteps, lengths, params, 4, gout)
    assert all(torch.equal(a, b) for a, b in zip(ok, again))


@pytest.mark.parametrize("B,T,C,L", [(6, 700, 256, 8), (19, 1500, 256, 20)])
def test_diffnet_stack_batched_weight_gradients(dev, monkeypatch, B, T, C, L):
    """The driver's default: the 2 L weight gradients of the stack as two batched launches in which every dw element has ONE
    owner block (ptpp_conv1d_wgrad_batched; no split-K partials).  Same sums in another order than the per-layer kernels:
    equal to them within f32 summation noise, and -- unlike them -- bit-reproducible, bias gradients included."""
    from promptttspp_amd import functional as PF

    h0, cond, dsteps, lengths, params = _stack_case(dev, B, T, C, L, torch.bfloat16, True, seed=11)
    gout = rnd(8, B, T, C).to(dev).bfloat16()
    monkeypatch.setattr(PF, "STACK_DRIVERS", True)
    monkeypatch.setattr(PF, "BATCHED_WGRAD", False)
    ref = _run(PF, h0, cond, dsteps, lengths, params, 4, gout)
    monkeypatch.setattr(PF, "BATCHED_WGRAD", True)
    got = _run(PF, h0, cond, dsteps, lengths, params, 4, gout)
    again = _run(PF, h0, cond, dsteps, lengths, params, 4, gout)
    for i, (a, b, c) in enumerate(zip(ref, got, again)):
        assert torch.equal(b, c), i                      # reproducible
        if i < 4:
            assert torch.equal(a, b), i                  # nothing but the weight gradients changes
        else:
            err = float((a.double() - b.double()).abs().max() / a.double().abs().max())
            assert err < 2e-5, (i, err)


@pytest.mark.parametrize("cin,cout,ks,dils,masked", [(256, 512, 3, (1, 2, 4, 8, 1, 2), False), (256, 512, 1, (1,) * 16, False),
                                                     (128, 256, 5, (1, 1, 1, 1, 1, 1, 1, 1), True), (64, 64, 3, (1, 2), False),
                                                     (256, 512, 3, (1, 2, 4, 8) * 5, False), (256, 512, 1, (1,) * 20, True)])
@pytest.mark.parametrize("split", ["0", "1"])
def test_conv1d_wgrad_batched_against_the_single_problem_kernel(dev, monkeypatch, cin, cout, ks, dils, masked, split):
    """ptpp_conv1d_wgrad_batched (reference: autograd of nn.Conv1d, e.g. modules/denoiser.py:58-64) on problems of one shape:
    accumulates into pre-filled targets exactly what ptpp_conv1d_wgrad adds, within f32 summation noise; also against
    torch's f32 convolution backward of the same bf16-rounded operands.  (The last case is too small to batch: the entry
    point falls back to the single-problem kernel.)  ``split`` = 1: the opt-in 3-way row split of a batch that does not fill the
    chip (PTPP_WGRAD_BATCH_SPLIT; partials summed in split order by a second launch) -- same bounds, and bit-reproducible."""
    from promptttspp_amd import ops

    monkeypatch.setenv("PTPP_WGRAD_BATCH_SPLIT", split)
    B, T = 7, 900
    n = len(dils)
    lengths = torch.tensor([T - 31 * i for i in range(B)], device=dev, dtype=torch.int32) if masked else None
    xs = [rnd(20 + i, B, T, cin).to(dev).bfloat16() for i in range(n)]
    dy_all = rnd(50, B, T, n * cout).to(dev).bfloat16()  # strided dy views, as the DiffNet stack hands them over
    init_w = [rnd(70 + i, cout, cin, ks).to(dev) * 0.1 for i in range(n)]
    init_b = [rnd(90 + i, cout).to(dev) * 0.1 for i in range(n)]
    got_w, got_b = [t.clone() for t in init_w], [t.clone() for t in init_b]
    probs = []
    for i, d in enumerate(dils):
        pad = (ks - 1) * d // 2
        probs.append((xs[i], dy_all[:, :, i * cout:(i + 1) * cout], got_w[i], got_b[i], d, pad))
    ops.conv1d_wgrad_batched(probs, cin, cout, ks, lengths=lengths, in_mask=masked)
    if len(dils) >= 16:  # the batched forms have one owner per element (or a fixed-order sum of three): run twice, same bits
        rep_w, rep_b = [t.clone() for t in init_w], [t.clone() for t in init_b]
        ops.conv1d_wgrad_batched([(xs[i], dy_all[:, :, i * cout:(i + 1) * cout], rep_w[i], rep_b[i], d, (ks - 1) * d // 2)
                                  for i, d in enumerate(dils)], cin, cout, ks, lengths=lengths, in_mask=masked)
        assert all(torch.equal(a, b) for a, b in zip(rep_w + rep_b, got_w + got_b))
    for i, d in enumerate(dils):
        pad = (ks - 1) * d // 2
        dy = dy_all[:, :, i * cout:(i + 1) * cout]
        w1, b1 = init_w[i].clone(), init_b[i].clone()
        ops.conv1d_wgrad(xs[i], dy, cin, cout, ks, d, pad, lengths, masked, True, dw_out=w1, db_out=b1)
        for a, b in ((w1, got_w[i]), (b1, got_b[i])):
            err = float((a.double() - b.double()).abs().max() / a.double().abs().max())
            assert err < 2e-5, (i, err)
        if i == 0:  # torch: gradient of conv1d w.r.t. weight / bias for upstream gradient dy
            x32 = xs[i].float()
            if masked:
                x32 = x32 * (torch.arange(T, device=dev)[None, :, None] < lengths[:, None, None])
            w = torch.zeros(cout, cin, ks, device=dev, requires_grad=True)
            bb = torch.zeros(cout, device=dev, requires_grad=True)
            y = torch.nn.functional.conv1d(x32.transpose(1, 2), w, bb, padding=pad, dilation=d)
            y.backward(dy.float().transpose(1, 2))
            for ref, g, ini in ((w.grad, got_w[i], init_w[i]), (bb.grad, got_b[i], init_b[i])):
                err = float(((g - ini).double() - ref.double()).abs().max() / ref.double().abs().max())
                assert err < 1e-4, err


@pytest.mark.parametrize("B,T,dtype", [(3, 200, torch.bfloat16), (20, 640, torch.bfloat16), (2, 50, torch.float32)])
def test_diffnet_stack_driver_inference_forward(dev, monkeypatch, B, T, dtype):
    """The no-save form the sampler calls once per reverse-diffusion step (two ping-pong slabs; bf16: the gate fused into
    the dilated conv's epilogue, conditioner projections in gate order) against the per-launch loop."""
    from promptttspp_amd import functional as PF

    C, L = 256, 8
    h0, cond, dsteps, _, params = _stack_case(dev, B, T, C, L, dtype, False, seed=3)
    with torch.no_grad():
        ws, bs = [p[2] for p in params], [p[3] for p in params]
        cond_all, _ = PF.diffnet_cond_all(cond, ws, bs, gate_perm=PF.diffnet_fused_gate(dtype))
        weights = [(p[0], p[1], p[4], p[5]) for p in params]
        monkeypatch.setattr(PF, "STACK_DRIVERS", False)
        ref, _ = PF.diffnet_stack_forward(h0, cond_all, dsteps, weights, None, 4, save=False)
        monkeypatch.setattr(PF, "STACK_DRIVERS", True)
        got, saved = PF.diffnet_stack_forward(h0, cond_all, dsteps, weights, None, 4, save=False)
    assert saved is None and torch.equal(ref, got) and float(ref.abs().max()) > 0


def test_diffnet_stack_driver_direct_gradients_on_the_side_stream(dev, monkeypatch):
    """Direct-accumulation mode (the trainer's): the driver forks the weight-gradient launches onto the side stream and adds
    into ``p.grad`` views of a flat buffer; equal to the per-launch path in the same mode, and the buffer is complete after
    the join."""
    from promptttspp_amd import functional as PF
    from promptttspp_amd.parallel import FlatGradReducer

    B, T, C, L = 5, 500, 256, 4
    h0, cond, dsteps, lengths, params = _stack_case(dev, B, T, C, L, torch.bfloat16, True, seed=5)
    flat_params = [p for lp in params for p in lp]
    gout = rnd(9, B, T, C).to(dev).bfloat16()
    outs = []
    monkeypatch.setattr(PF, "BATCHED_WGRAD", False)  # same summation order as the per-launch path
    try:
        red = FlatGradReducer(flat_params)
        assert PF.direct_grads_enabled()
        for drivers in (False, True):
            monkeypatch.setattr(PF, "STACK_DRIVERS", drivers)
            red.zero_grad()
            y = PF.diffnet_stack(h0.clone().requires_grad_(True), cond, dsteps, lengths, 4, params)
            y.backward(gout)
            red.finish()
            torch.cuda.synchronize()
            outs.append((y.detach().clone(), red.flat.clone()))
    finally:
        PF.enable_direct_grads(False)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float(outs[0][1].abs().max()) > 0


@pytest.mark.parametrize("dtype,train", [(torch.float32, False), (torch.bfloat16, True), (torch.bfloat16, False)])
def test_frozen_encoder_layers_driver_is_bit_identical(dev, monkeypatch, dtype, train):
    """ptpp_encoder_layers_fwd (the 11 frozen BERT layers of the prompt encoder, reference modules/prompt_encoder.py:25-38
    / transformers BertLayer) against the per-launch ``_frozen_layer`` loop: the CLS state is equal bit for bit, in eval mode
    and in train mode with all three dropout sites on (same seeds), f32 and bf16; f32 eval also against the fixture generated
    from transformers' BertModel."""
    from conftest import key_shapes, load_golden, rel_err
    from test_oracle_golden_am import synth_sd

    from promptttspp_amd import config
    from promptttspp_amd import functional as PF
    from promptttspp_amd.modules.prompt_encoder import BertWrapper

    g = load_golden("bert")
    bw = BertWrapper("bert-base-uncased")
    bw.model.load_state_dict(synth_sd(key_shapes(g["keys"]), 80), strict=False)
    bw = bw.to(dev)
    bw.train(train)
    ids, am = g["ids"].to(dev), g["am"].to(dev)
    old = config.compute_dtype()
    config.set_compute_dtype(dtype)
    try:
        outs = []
        for drivers in (False, True):
            monkeypatch.setattr(PF, "STACK_DRIVERS", drivers)
            PF.manual_seed(77)
            with torch.no_grad():
                outs.append(bw((ids, am), dev).clone())
    finally:
        config.set_compute_dtype(old)
    assert torch.equal(outs[0], outs[1]) and float(outs[0].abs().max()) > 0
    if dtype == torch.float32 and not train:
        assert rel_err(outs[1].cpu(), g["cls"]) < 1e-4


def _module_grads(mod, run, x, gout):
    xin = x.clone().requires_grad_(True)
    for p in mod.parameters():
        p.grad = None
    y = run(mod, xin)
    y.backward(gout)
    torch.cuda.synchronize()
    return [y.detach().clone(), xin.grad.clone()] + [p.grad.clone() for p in mod.parameters()]


@pytest.mark.parametrize("kind,dtype,B,T", [("pitch", torch.bfloat16, 5, 700), ("pitch", torch.float32, 3, 90),
                                            ("frame_prior", torch.bfloat16, 5, 700), ("frame_prior", torch.float32, 2, 120),
                                            ("frame_prior", torch.bfloat16, 3, 300)])
def test_conv_ln_stack_driver_is_bit_identical(dev, monkeypatch, kind, dtype, B, T):
    """ptpp_conv_ln_stack_fwd / _bwd against the chains of Conv1dFn / LayerNormFn nodes they replace, in train mode with
    dropout on (same seeds): the pitch predictor's layers (reference modules/variance_adaptor.py:23-62: conv k5 -> ReLU -> LN ->
    dropout 0.5 -> mask) and the frame prior network (modules/frame_prior.py:76-89: x = LN(x + dropout(gelu(conv k17(x * mask))))):
    output, input gradient and every parameter gradient equal bit for bit (exact-f32 weight gradients: atomics, 1e-5)."""
    from promptttspp_amd import config
    from promptttspp_amd import functional as PF
    from promptttspp_amd.modules.frame_prior import FramePriorNetwork
    from promptttspp_amd.modules.variance_adaptor import Predictor

    torch.manual_seed(3)
    C = 256
    if kind == "pitch":
        mod = Predictor(C, 2, 5, 0.5, 5).to(dev).train()
        run = lambda m, x: m.cl(x, lengths)  # noqa: E731
        gout = rnd(5, B, T, 2).to(dev)
    else:
        mod = FramePriorNetwork(C, C, 6, 17, 0.1).to(dev).train()
        run = lambda m, x: m.forward_cl(x, lengths)  # noqa: E731
        gout = rnd(5, B, T, C).to(dev).to(dtype)
    for p in mod.parameters():
        p.data.add_(0.05 * torch.randn_like(p))
    lengths = torch.tensor([max(8, T - 37 * i) for i in range(B)], device=dev, dtype=torch.int32)
    x = rnd(6, B, T, C).to(dev).to(dtype)
    x = x * (torch.arange(T, device=dev)[None, :, None] < lengths[:, None, None])
    old = config.compute_dtype()
    config.set_compute_dtype(dtype)
    monkeypatch.setattr(PF, "BATCHED_WGRAD", False)
    try:
        outs = []
        for drivers in (False, True):
            monkeypatch.setattr(PF, "STACK_DRIVERS", drivers)
            PF.manual_seed(99)
            outs.append(_module_grads(mod, run, x, gout))
    finally:
        config.set_compute_dtype(old)
    names = ["y", "dx"] + [n for n, _ in mod.named_parameters()]
    assert len(outs[0]) == len(outs[1]) == len(names)
    for n, a, b in zip(names, *outs):
        assert a.shape == b.shape and a.dtype == b.dtype and float(a.float().abs().max()) > 0, n
        if dtype == torch.float32 and n not in ("y", "dx"):
            assert torch.allclose(a, b, rtol=2e-5, atol=2e-6 * float(a.abs().max())), n   # f32 weight-gradient kernel: atomics
        elif "gamma" in n or "beta" in n or n.startswith("out_layer"):
            # LayerNorm parameter gradients -- and, since round 6, those of the pitch head's 256 -> 2 projection
            # (ptpp_linear_small_bwd): per-block totals meet in the 32 replicas of the reduction scratch through f32
            # atomics (include/ptpp.h "Reduction scratch"), in both paths: last bits depend on the arrival order
            assert torch.allclose(a, b, rtol=2e-5, atol=2e-6 * float(a.abs().max())), n
        else:
            assert torch.equal(a, b), (n, float((a.float() - b.float()).abs().max()))


def test_conv_ln_stack_batched_weight_gradients(dev, monkeypatch):
    """Frame prior network at the bench shape with the batched weight gradient (6 layers x 24 tiles: one launch, no split-K
    partials): everything but the conv weight / bias gradients is unchanged, those agree within f32 summation noise and
    are bit-reproducible."""
    from promptttspp_amd import config
    from promptttspp_amd import functional as PF
    from promptttspp_amd.modules.frame_prior import FramePriorNetwork

    torch.manual_seed(4)
    B, T, C = 19, 1500, 256
    mod = FramePriorNetwork(C, C, 6, 17, 0.1).to(dev).train()
    lengths = torch.tensor([T - 29 * i for i in range(B)], device=dev, dtype=torch.int32)
    x = rnd(6, B, T, C).to(dev).bfloat16() * (torch.arange(T, device=dev)[None, :, None] < lengths[:, None, None])
    gout = rnd(5, B, T, C).to(dev).bfloat16()
    run = lambda m, xx: m.forward_cl(xx, lengths)  # noqa: E731
    old = config.compute_dtype()
    config.set_compute_dtype(torch.bfloat16)
    monkeypatch.setattr(PF, "STACK_DRIVERS", True)
    try:
        outs = []
        for batched in (False, True, True):
            monkeypatch.setattr(PF, "BATCHED_WGRAD", batched)
            PF.manual_seed(99)
            outs.append(_module_grads(mod, run, x, gout))
    finally:
        config.set_compute_dtype(old)
    names = ["y", "dx"] + [n for n, _ in mod.named_parameters()]
    for n, a, b, c in zip(names, *outs):
        if n.startswith("convs"):
            assert torch.equal(b, c), n                  # reproducible
            err = float((a.double() - b.double()).abs().max() / a.double().abs().max())
            assert err < 2e-5, (n, err)
        elif "gamma" in n or "beta" in n:                # (atomics in the reduction scratch, see above)
            assert torch.allclose(a, b, rtol=2e-5, atol=2e-6 * float(a.abs().max())), n
        else:
            assert torch.equal(a, b) and torch.equal(b, c), n


@pytest.mark.parametrize("mode", ["1", "2"])
def test_branch_streams_give_the_same_losses_and_gradients(dev, monkeypatch, mode):
    """PTPP_BRANCH_STREAMS: the prompt branch (mode 1) and the reference encoder (mode 2, the default) of the TRAINING forward
    on their own streams -- forward here, backward by autograd on the same streams -- against the single-stream step on the same
    weights, batch and dropout seeds (f32, reference model.py:72-183): the six losses are equal bit for bit, every parameter
    gradient within the noise of the order-dependent reductions (a missed cross-stream dependency would show as a wrong or
    missing gradient, orders of magnitude above that), three times in a row."""
    import test_hip_acoustic as T

    from promptttspp_amd import config
    from promptttspp_amd import functional as PF
    from promptttspp_amd import ops
    from promptttspp_amd.models.prompttts_mdn_v2_final import model as M

    old = config.compute_dtype()
    config.set_compute_dtype(torch.float32)
    try:
        m, g = T._model(dev)
        m.train()
        batch = T._batch(g, dev)

        def run():
            for p in m.parameters():
                p.grad = None
            m.decoder.injected = {"t": g["t"], "noise": g["noise"]}
            PF.manual_seed(321)
            torch.manual_seed(5)
            out = m(batch)
            out["loss"].backward()
            torch.cuda.synchronize()
            return {k: float(v) for k, v in out.items()}, {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

        monkeypatch.setattr(M, "BRANCH_STREAMS", "")
        ref_l, ref_g = run()
        monkeypatch.setattr(M, "BRANCH_STREAMS", mode)
        monkeypatch.setattr(ops, "_NO_PIN", True)
        for rep in range(3):
            got_l, got_g = run()
            assert got_l == ref_l, (rep, got_l, ref_l)
            assert got_g.keys() == ref_g.keys()
            for n in ref_g:
                a, b = ref_g[n], got_g[n]
                assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * float(a.abs().max()) + 1e-12), (rep, n, float((a - b).abs().max()))
    finally:
        config.set_compute_dtype(old)


@pytest.mark.parametrize("variant,dtype,train", [("new", torch.bfloat16, True), ("new", torch.float32, True), ("legacy", torch.bfloat16, True),
                                                 ("new", torch.bfloat16, False)])
def test_conformer_block_driver_is_bit_identical(dev, monkeypatch, variant, dtype, train):
    """ptpp_conformer_block_fwd / _bwd (reference modules/esp/conformer/encoder_layer.py:74-162) against the chain of ~20
    autograd nodes per block they replace, on the 4-block encoder of the model's config, train mode with every dropout site on
    (same seeds) and train-mode BatchNorm: output and input gradient equal bit for bit.  The conv / linear weight and bias
    gradients are sums over rows in two different fixed orders (per-launch path: split-K partials + ordered reduction per layer;
    driver: ONE grouped launch per block whose owner blocks walk all rows, ptpp_conv1d_wgrad_grouped) and LayerNorm / BatchNorm /
    pos_bias / depthwise gradients go through f32 atomics in both paths: tolerance 2e-5; the BatchNorm running statistics
    advance identically."""
    import copy

    from promptttspp_amd import config
    from promptttspp_amd import functional as PF
    from promptttspp_amd.modules.esp import ConformerEncoder

    torch.manual_seed(11)
    old = config.compute_dtype()
    config.set_compute_dtype(dtype)

    enc = ConformerEncoder(idim=256, attention_dim=256, attention_heads=2, linear_units=1024, num_blocks=4, dropout_rate=0.2,
                           positionwise_layer_type="conv1d", positionwise_conv_kernel_size=9, macaron_style=True,
                           pos_enc_layer_type="rel_pos", selfattention_layer_type="rel_selfattn", activation_type="swish",
                           use_cnn_module=True, cnn_module_kernel=7, rel_pos_type=variant).to(dev)
    with torch.no_grad():
        for n_, p_ in enc.named_parameters():
            if p_.dim() == 1 and ("norm" in n_ or "bias" in n_):
                p_.add_(0.1 * torch.randn_like(p_))
    enc.train(train)
    B, Tn, C = 5, 77, 256
    lengths = torch.tensor([77, 60, 33, 12, 1], device=dev, dtype=torch.int32)
    mask = (torch.arange(Tn, device=dev)[None, :] < lengths[:, None]).unsqueeze(-1).float()
    x = (rnd(3, B, Tn, C).to(dev) * mask).to(dtype)
    gout = rnd(4, B, Tn, C).to(dev).to(dtype)
    state0 = copy.deepcopy(enc.state_dict())
    monkeypatch.setattr(PF, "BATCHED_WGRAD", False)
    try:
        outs, stats = [], []
        for drivers in (False, True):
            monkeypatch.setattr(PF, "STACK_DRIVERS", drivers)
            enc.load_state_dict(state0)
            PF.manual_seed(99)
            xin = x.clone().requires_grad_(True)
            for p_ in enc.parameters():
                p_.grad = None
            y = enc.forward_cl(xin, lengths, mask)
            if train:
                y.backward(gout)
            torch.cuda.synchronize()
            outs.append([y.detach().clone()] + ([xin.grad.clone()] + [p_.grad.clone() for p_ in enc.parameters()] if train else []))
            stats.append({k: v.clone() for k, v in enc.state_dict().items() if "running" in k or "num_batches" in k})
    finally:
        config.set_compute_dtype(old)
    names = ["y", "dx"] + [n_ for n_, _ in enc.named_parameters()]
    assert len(outs[0]) == len(outs[1]) == (len(names) if train else 1)
    for n_, a, b in zip(names, *outs):
        assert a.shape == b.shape and a.dtype == b.dtype, n_
        exact = n_ in ("y", "dx")
        if exact:
            assert torch.equal(a, b), (n_, float((a.float() - b.float()).abs().max()))
        elif "linear_pos" in n_ and dtype == torch.bfloat16:
            # its upstream gradient dpos is summed over batch groups with f32 atomics and then rounded to bf16 (in both paths):
            # a last-bit difference before the rounding moves single elements by one bf16 step
            err = float((a - b).norm() / a.norm())
            assert err < 2e-3, (n_, err)
        else:  # (atol floor: linear_k.bias has a structurally zero gradient -- softmax is shift invariant -- i.e. pure rounding noise)
            assert torch.allclose(a, b, rtol=2e-5, atol=max(3e-6 * float(a.abs().max()), 1e-7)), (n_, float((a - b).abs().max()))
    for k in stats[0]:
        assert torch.equal(stats[0][k], stats[1][k]) or torch.allclose(stats[0][k].float(), stats[1][k].float(), rtol=1e-6, atol=1e-7), k


@pytest.mark.parametrize("B,L,I,H", [(5, 9, 256, 128), (19, 24, 1024, 256), (1, 1, 64, 128), (3, 7, 64, 256)])
def test_gru_whole_sequence_kernels_match_torch_gru(B, L, I, H):
    """nn_ops.gru_last_state (one launch for the whole recurrence, H = 128 / 256) against torch.nn.GRU on packed sequences
    (the reference's op, modules/reference_encoder.py:108-123) in f32 on the CPU: last valid hidden state and the gradients of
    the input, both weight matrices and both biases; and against this package's own per-step path."""
    from promptttspp_amd import nn_ops as NO
    from promptttspp_amd import config

    dev = torch.device("cuda:0")
    gru = torch.nn.GRU(I, H, 1, batch_first=True)
    x = rnd(1, B, L, I, scale=0.7)
    lens = torch.tensor([max(1, L - (3 * i) % L) for i in range(B)], dtype=torch.long)
    gout = rnd(2, B, H)
    # reference
    xr = x.clone().requires_grad_(True)
    packed = torch.nn.utils.rnn.pack_padded_sequence(xr, lens, batch_first=True, enforce_sorted=False)
    _, hn = gru(packed)
    (hn[-1] * gout).sum().backward()
    want = [hn[-1].detach(), xr.grad] + [p.grad.clone() for p in gru.parameters()]
    outs = {}
    with config.use_dtype(torch.float32):
        for seq in (True, False):
            NO.GRU_SEQ = seq
            try:
                g2 = torch.nn.GRU(I, H, 1, batch_first=True).to(dev)
                g2.load_state_dict(gru.state_dict())
                xd = x.to(dev).requires_grad_(True)
                h = NO.gru_last_state(xd, g2.weight_ih_l0, g2.weight_hh_l0, g2.bias_ih_l0, g2.bias_hh_l0, lens.to(dev))
                (h * gout.to(dev)).sum().backward()
                torch.cuda.synchronize()
                outs[seq] = [h.detach().cpu(), xd.grad.cpu()] + [p.grad.cpu() for p in g2.parameters()]
            finally:
                NO.GRU_SEQ = True
    for i, (a, b, c) in enumerate(zip(outs[True], want, outs[False])):
        scale = float(b.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-5 * scale + 1e-7, (i, float((a - b).abs().max()), scale)
        assert float((a - c).abs().max()) <= 2e-5 * scale + 1e-7, (i, float((a - c).abs().max()), scale)


@pytest.mark.parametrize("dtype,B,T", [(torch.float32, 3, 203), (torch.bfloat16, 5, 640), (torch.bfloat16, 19, 1500)])
def test_reference_encoder_conv_stack_driver_matches_the_per_launch_path(dtype, B, T):
    """modules.reference_encoder in training mode: the two C calls of nn_ops.RefEncConvsFn (+ the one-launch GRU) against the
    per-launch path (im2col + GEMM + BatchNorm kernels issued from Python): same kernels on the same operands, so the
    embedding, the running estimates and every gradient agree -- up to the sums the kernels accumulate with f32 atomics
    (BatchNorm statistics / gamma / beta, f32 weight gradients): 2e-5 of the largest element in f32; in bf16 a statistic that
    moves by an ulp moves bf16 roundings and ReLU gates downstream, so there the bound is 3 % relative L2 per tensor
    (tools/diag_refenc.py on MI355X, 19 x 1500 frames: repeated runs of EITHER path fall into two classes that differ by 2.7e-3
    in the embedding and 5e-5 in the running estimates; within a class the two paths agree bit for bit)."""
    from promptttspp_amd import config
    from promptttspp_amd import functional as PF
    from promptttspp_amd.modules.reference_encoder import ReferenceEncoder

    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    ref = ReferenceEncoder().to(dev)
    mel = (rnd(5, B, 80, T) * 1.5).to(dev)
    lens = torch.tensor([max(40, T - 37 * i) for i in range(B)], device=dev)
    gout = rnd(6, B, 128, 1).to(dev)
    state = {k: v.clone() for k, v in ref.state_dict().items()}
    outs = {}
    with config.use_dtype(dtype):
        for drv in (True, False):
            ref.load_state_dict(state)
            ref.train()
            for p in ref.parameters():
                p.grad = None
            PF.STACK_DRIVERS = drv
            try:
                y = ref(mel, lens)
                (y * gout).sum().backward()
                torch.cuda.synchronize()
            finally:
                PF.STACK_DRIVERS = True
            outs[drv] = ([y.detach()] + [p.grad.clone() for p in ref.parameters()], {k: v.clone() for k, v in ref.state_dict().items()})
    names = ["y"] + [n for n, _ in ref.named_parameters()]
    errs = []
    for n, a, b in zip(names, outs[True][0], outs[False][0]):
        assert a.shape == b.shape and torch.isfinite(a).all(), n
        if dtype == torch.float32:
            errs.append((n, float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12)))
        else:
            errs.append((n, float((a - b).norm() / (b.norm() + 1e-12))))
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    assert all(e <= tol for _, e in errs), [x for x in errs if x[1] > tol]
    for k in state:
        a, b = outs[True][1][k].float(), outs[False][1][k].float()
        assert float((a - b).abs().max()) <= (1e-5 if dtype == torch.float32 else 5e-4) * (float(b.abs().max()) + 1e-12), k
        if "num_batches" in k:
            assert int(outs[True][1][k]) == int(state[k]) + 1


def test_grouped_weight_gradients_match_the_per_layer_launches():
    """ptpp_conv1d_wgrad_grouped: the weight / bias gradients of layers of different shapes over phone-level rows in ONE launch
    (every dw tile owned by one block) against one ptpp_conv1d_wgrad per layer: equal to f32 summation-order noise, bit-reproducible
    run to run, accumulating into non-zero targets, with input masks, taps, row strides and a row-flattened k = 1 problem; f32
    and shapes the grouped kernel does not take fall back to the per-layer launches."""
    from promptttspp_amd import ops

    dev = torch.device("cuda:0")
    B, T = 7, 150
    lengths = torch.tensor([150, 149, 97, 64, 33, 2, 1], device=dev, dtype=torch.int32)
    for dtype in (torch.bfloat16, torch.float32):
        big = (rnd(1, B, T, 3 * 256) * 0.5).to(dev).to(dtype)
        specs = [  # cin, cout, ks, pad, masked, x view, dy view
            (256, 1024, 9, 4, True), (1024, 256, 9, 4, False), (256, 256, 1, 0, False), (256, 512, 1, 0, False), (256, 256, 3, 1, True)]
        probs, singles = [], []
        for i, (cin, cout, ks, pad, masked) in enumerate(specs):
            x = (rnd(10 + i, B, T, cin) * 0.5).to(dev).to(dtype)
            dy = big[:, :, 256:512] if (cout == 256 and ks == 1) else (rnd(20 + i, B, T, cout) * 0.5).to(dev).to(dtype)
            base = rnd(30 + i, cout, cin, ks).to(dev)
            bbase = rnd(40 + i, cout).to(dev)
            ln = lengths if masked else None
            probs.append((x, dy, base.clone(), bbase.clone(), ks, 1, pad, ln))
            dw, db = base.clone(), bbase.clone()
            ops.conv1d_wgrad(x, dy, cin, cout, ks, 1, pad, lengths=ln, in_mask=masked, dw_out=dw, db_out=db)
            singles.append((dw, db))
        ops.conv1d_wgrad_grouped(probs)
        again = [(p[0], p[1], rnd(30 + i, *p[2].shape).to(dev), rnd(40 + i, *p[3].shape).to(dev)) + p[4:] for i, p in enumerate(probs)]
        ops.conv1d_wgrad_grouped(again)
        torch.cuda.synchronize()
        for (x, dy, dw, db, *_), (dw1, db1), (_, _, dw2, db2, *_) in zip(probs, singles, again):
            if dtype == torch.bfloat16:
                assert torch.equal(dw, dw2) and torch.equal(db, db2)                  # one owner per element: reproducible
            s = float((dw1 - 0).abs().max())
            assert float((dw - dw1).abs().max()) <= 2e-5 * s + 1e-6, (dw.shape, float((dw - dw1).abs().max()), s)
            assert float((db - db1).abs().max()) <= 2e-5 * float(db1.abs().max()) + 1e-6


@pytest.mark.parametrize("B,T,masked", [(3, 200, True), (19, 1100, False)])
def test_fused_gate_with_kept_pre_activation_is_bit_identical(B, T, masked, monkeypatch):
    """Training forward in bf16: the DiffNet gate in the dilated conv's epilogue with the pre-activation kept
    (ptpp_conv1d_gate_fwd_save; weights from the pack cache in the gate-interleaved order, mode 2; biases and conditioner
    slice in the same order) against conv + gate_fwd: stack output and every gradient equal bit for bit, on both the driver
    and the per-launch path; the mode-2 operand of the pack cache equals the permuted pack, also after a batched repack."""
    from promptttspp_amd import functional as PF
    from promptttspp_amd import ops

    dev = torch.device("cuda:0")
    C, L = 256, 4
    h0, cond, dsteps, lengths, params = _stack_case(dev, B, T, C, L, torch.bfloat16, masked, seed=5)
    gout = (rnd(9, B, T, C) * 0.1).to(dev).bfloat16()
    outs = {}
    for drivers in (True, False):
        for fused in (True, False):
            monkeypatch.setattr(PF, "STACK_DRIVERS", drivers)
            monkeypatch.setattr(PF, "FUSE_GATE_SAVE", fused)
            outs[(drivers, fused)] = _run(PF, h0, cond, dsteps, lengths, params, 4, gout)
    ref = outs[(True, False)]
    for key, got in outs.items():
        for i, (a, b) in enumerate(zip(got, ref)):
            assert torch.equal(a, b), (key, i, float((a.float() - b.float()).abs().max()))
    # the pack cache's mode-2 operand
    w = params[0][0]
    perm = PF._gate_perm(2 * C, dev)
    want = ops.pack_conv_weight(w.detach()[perm], torch.bfloat16)
    assert torch.equal(PF.packed(w, torch.bfloat16, mode=2), want)
    with torch.no_grad():
        w.mul_(1.5)
    PF.repack_all()
    assert torch.equal(PF.packed(w, torch.bfloat16, mode=2), ops.pack_conv_weight(w.detach()[perm], torch.bfloat16))
