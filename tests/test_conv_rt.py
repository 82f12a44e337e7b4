"""GPU: the row-tile conv kernel (csrc/conv1d_rt.hip, ptpp_conv1d_rt_fwd) against the tile kernel it replaces on the
frame-level 256-channel layers -- frame prior network k = 17 (reference modules/frame_prior.py:85-89), pitch predictor k = 5
(modules/variance_adaptor.py:31-36), the DiffNet dilated conv's data gradient (modules/denoiser.py:58-64): same accumulation
order and epilogue arithmetic, so outputs must be equal BIT FOR BIT; plus the f32 oracle (torch conv1d on the CPU)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(dev, B, T, cin, ks, seed):
    g = torch.Generator().manual_seed(500 + seed)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    return r(B, T, cin).bfloat16(), r(256, cin, ks, sc=(cin * ks) ** -0.5), r(256, sc=0.1), r(B, T, 256).bfloat16()


@pytest.mark.parametrize("B,T,cin,ks,dil,act,masked,use_res,bm", [
    (14, 650, 256, 17, 1, None, True, False, 128),    # frame prior forward: conv mask on the input, no activation (T > 512:
                                                      # below that ops.conv1d hands the tile kernel a split-K scratch)
    (20, 450, 256, 5, 1, "relu", False, False, 96),   # pitch predictor forward
    (9, 1000, 512, 3, 8, None, True, True, 128),      # dilated data gradient: residual with scale, masked input
    (70, 130, 256, 3, 1, "relu", True, True, 64),     # short utterances: ragged last tiles
    (16, 617, 320, 7, 2, None, False, True, 0),       # Cin not a power of two, launcher's own tile choice
    (9, 650, 256, 17, 1, None, True, False, 160),     # 160-row blocks of the global-weights form (ten row tiles per wave; T > 512, see above)
    (7, 700, 512, 3, 8, "relu", True, True, 160),
    (65, 459, 256, 5, 1, "relu", False, False, 0),    # 260 blocks of 128 rows: the launcher takes 160 rows on its own
    (12, 650, 1024, 1, 1, None, False, False, 128),   # 1 x 1 with a long K (round 6: the DiffNet conditioner's data gradient, K = 10 240)
    (9, 700, 2560, 1, 1, None, True, True, 0),
])
def test_row_tile_conv_is_bit_identical_to_the_tile_kernel(dev, monkeypatch, B, T, cin, ks, dil, act, masked, use_res, bm):
    from promptttspp_amd import ops

    monkeypatch.setattr(ops, "CONV_RT_MIN_ROWS", 1)  # (the product takes the kernel from ~200 row tiles on: small cases here)
    if bm:
        monkeypatch.setenv("PTPP_CONV_RT_BM", str(bm))
    x, w, b, res = _mk(dev, B, T, cin, ks, ks + dil)
    pad = dil * (ks - 1) // 2
    lengths = torch.tensor([max(1, T - 37 * i) for i in range(B)], device=dev, dtype=torch.int32) if masked else None
    kw = dict(ks=ks, dil=dil, pad=pad, act=act, lengths=lengths, in_mask=masked, out_mask=masked and act is not None,
              res=res if use_res else None, res_scale=0.7071067811865476 if use_res else 1.0)
    ref = ops.conv1d(x, ops.pack_conv_weight(w, torch.bfloat16), b, 256, **kw)
    ws = ops.pack_conv_weight(w, torch.bfloat16, 3)
    assert ops.conv1d_rt_ok(x, 256, ks, dil, act)
    got = ops.conv1d(x, None, b, 256, wstream=ws, **kw)
    torch.cuda.synchronize()
    assert torch.equal(ref, got), (float((ref.float() - got.float()).abs().max()), int((ref != got).sum()))
    assert float(ref.float().abs().max()) > 0


def test_row_tile_data_gradient_operand_and_oracle(dev, monkeypatch):
    """Pack mode 4 (the stream form of the transposed, tap-flipped operand) against mode 1 through the tile kernel, and both
    against the f32 oracle: the data gradient of a dilated conv is conv_transpose1d."""
    import torch.nn.functional as F
    from promptttspp_amd import ops

    monkeypatch.setattr(ops, "CONV_RT_MIN_ROWS", 1)
    B, T, cout, cin, ks, dil = 12, 800, 512, 256, 3, 4  # gradient flows 512 -> 256 channels
    g = torch.Generator().manual_seed(77)
    dy = (torch.randn(B, T, cout, generator=g)).to(dev).bfloat16()
    w = (torch.randn(cout, cin, ks, generator=g) * (cout * ks) ** -0.5).to(dev)
    pad = dil * (ks - 1) // 2
    ref = ops.conv1d(dy, ops.pack_conv_weight(w, torch.bfloat16, 1), None, cin, ks=ks, dil=dil, pad=pad)
    got = ops.conv1d(dy, None, None, cin, ks=ks, dil=dil, pad=pad, wstream=ops.pack_conv_weight(w, torch.bfloat16, 4))
    torch.cuda.synchronize()
    assert torch.equal(ref, got)
    orc = F.conv_transpose1d(dy.float().cpu().transpose(1, 2), w.bfloat16().float().cpu(), padding=pad, dilation=dil).transpose(1, 2)
    err = float((got.float().cpu() - orc).abs().max() / orc.abs().max())
    assert err < 1e-2, err


@pytest.mark.parametrize("B,T,masked,bm", [(9, 700, True, 128), (12, 450, False, 96), (40, 130, True, 64), (5, 1000, False, 0),
                                          (6, 459, True, 160)])
def test_gate_backward_on_the_row_tile_engine_is_bit_identical(dev, monkeypatch, B, T, masked, bm):
    """ptpp_conv1d_rt_gate_bwd (the DiffNet output projection's data gradient, 512 -> 256, 1 x 1, with the gate backward as its
    epilogue; reference modules/denoiser.py:76-83 differentiated) against ptpp_conv1d_gate_bwd on the tile kernel: the same MFMA
    order and epilogue arithmetic, so da must be equal bit for bit -- and against the f32 oracle of the two steps."""
    from promptttspp_amd import ops

    monkeypatch.setattr(ops, "CONV_RT_MIN_ROWS", 1)
    if bm:
        monkeypatch.setenv("PTPP_CONV_RT_BM", str(bm))
    C = 256
    g = torch.Generator().manual_seed(900 + B)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    do = r(B, T, 2 * C).bfloat16()
    a = r(B, T, 2 * C, sc=1.5).bfloat16()
    w = r(2 * C, C, 1, sc=(2 * C) ** -0.5)
    lengths = torch.tensor([max(1, T - 53 * i) for i in range(B)], device=dev, dtype=torch.int32) if masked else None
    if masked:  # as the backward has it: do is zero past an utterance's end
        do = do * (torch.arange(T, device=dev)[None, :, None] < lengths[:, None, None])
    ldc = 3 * 2 * C  # da is a slice of the all-layer tensor
    da_ref = torch.zeros(B, T, ldc, device=dev, dtype=torch.bfloat16)
    da_got = torch.zeros_like(da_ref)
    assert ops.conv1d_gate_bwd_supported(C, 2 * C, torch.bfloat16) and ops.conv1d_rt_gate_bwd_ok(do, C)
    ops.conv1d_gate_bwd(do, ops.pack_conv_weight(w, torch.bfloat16, 1), a, da_ref[:, :, 2 * C:4 * C], lengths=lengths)
    ops.conv1d_rt_gate_bwd(do, ops.pack_conv_weight(w, torch.bfloat16, 4), a, da_got[:, :, 2 * C:4 * C], lengths=lengths)
    torch.cuda.synchronize()
    assert torch.equal(da_ref, da_got), int((da_ref != da_got).sum())
    # oracle: dg = do @ W (bf16 operands, f32 accumulation), then the gate's derivative
    dg = torch.einsum("btk,kc->btc", do.float().cpu(), w[:, :, 0].bfloat16().float().cpu()).bfloat16().float()
    s_, f_ = a[:, :, :C].float().cpu(), a[:, :, C:].float().cpu()
    sg, th = torch.sigmoid(s_), torch.tanh(f_)
    ora = torch.cat([dg * th * sg * (1 - sg), dg * sg * (1 - th * th)], dim=2)
    got = da_got[:, :, 2 * C:4 * C].float().cpu()
    assert float((got - ora).abs().max()) < 2e-2 * float(ora.abs().max())
    assert float(got.abs().max()) > 0


@pytest.mark.parametrize("B,T,cin,cout,act,drop,res,bm,nsplit", [
    (19, 169, 256, 1024, "relu", 0.2, False, 0, 0),    # feed-forward w_1 at the bench's phone-level shape: four column groups
    (19, 169, 1024, 256, None, 0.2, True, 0, 0),       # w_2: the input channels split over blocks, residual + 0.5 scale + dropout
    (7, 200, 256, 1024, None, 0.0, False, 96, 0),      # w_2's data gradient (256 -> 1024), 96-row blocks
    (5, 300, 1024, 256, None, 0.0, False, 128, 2),     # w_1's data gradient (1024 -> 256), two splits of 128-row blocks
    (40, 60, 1024, 256, "relu", 0.1, True, 64, 1),     # no split (one block walks all of Cin)
    (3, 33, 256, 512, "relu", 0.0, False, 64, 0),      # a single ragged tile per utterance
    (6, 150, 256, 1024, None, 0.3, False, 0, 0),       # dropout pattern: the same elements as the tile kernel drops
    (6, 150, 1024, 256, None, 0.3, False, 0, 0),
])
def test_feed_forward_convs_on_the_row_tile_engine(dev, monkeypatch, B, T, cin, cout, act, drop, res, bm, nsplit):
    """ptpp_conv1d_rt_fwd_ex (the Conformer blocks' k = 9 feed-forward convs, reference
    modules/esp/transformer/multi_layer_conv.py:52-67, and their data gradients) against the tile kernel on the same operands and
    dropout seed: the SAME elements are dropped, values agree to bf16 rounding of sums taken in another order, and both match the
    f32 oracle (torch conv1d on the bf16-rounded operands)."""
    import torch.nn.functional as F
    from promptttspp_amd import ops

    if bm:
        monkeypatch.setenv("PTPP_CONV_RT_BM", str(bm))
    if nsplit:
        monkeypatch.setenv("PTPP_CONV_RT_NSPLIT", str(nsplit))
    g = torch.Generator().manual_seed(900 + cin + B)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    ks, pad = 9, 4
    x = r(B, T, cin).bfloat16()
    w = r(cout, cin, ks, sc=(cin * ks) ** -0.5)
    b = r(cout, sc=0.1)
    rs = r(B, T, cout).bfloat16() if res else None
    lengths = torch.tensor([max(1, T - 11 * i) for i in range(B)], device=dev, dtype=torch.int32)
    kw = dict(ks=ks, pad=pad, act=act, lengths=lengths, in_mask=True, out_mask=True, res=rs, out_scale=0.5 if res else 1.0,
              drop_p=drop, drop_seed=1234567)
    assert ops.conv1d_rt_ex_ok(x, cout, ks, 1, act)
    ref = ops.conv1d(x, ops.pack_conv_weight(w, torch.bfloat16), b, cout, **kw)
    got = ops.conv1d(x, None, b, cout, wstream=ops.pack_conv_weight(w, torch.bfloat16, 3), **kw)
    got2 = ops.conv1d(x, None, b, cout, wstream=ops.pack_conv_weight(w, torch.bfloat16, 3), **kw)
    torch.cuda.synchronize()
    assert torch.equal(got, got2)  # deterministic
    base = rs.float() if res else torch.zeros_like(ref, dtype=torch.float32)
    if drop > 0.0 and not res and act is None:  # the same elements are dropped (no residual / ReLU zeros to confuse the pattern)
        assert torch.equal(ref == 0, got == 0)
    scale = float(ref.float().abs().max())
    assert float((ref.float() - got.float()).abs().max()) <= 2e-2 * scale
    m = (torch.arange(T, device=dev)[None, :] < lengths[:, None]).unsqueeze(-1).float()
    orc = F.conv1d((x.float() * m).transpose(1, 2), w.bfloat16().float(), b, padding=pad).transpose(1, 2)
    if act == "relu":
        orc = torch.relu(orc)
    orc = orc * m * (0.5 if res else 1.0)
    if drop == 0.0:
        orc = orc + base
        assert float((got.float() - orc).abs().max()) <= 2e-2 * float(orc.abs().max())
    else:  # kept elements: the oracle scaled by 1 / keep
        keep = ((got.float() - base) != 0) & ((ref.float() - base) != 0)
        thr = round(drop * 65536) / 65536
        err = ((got.float() - base) - orc / (1 - thr))[keep].abs().max()
        assert float(err) <= 2e-2 * float(orc.abs().max()) / (1 - thr)
    # the data-gradient operand (pack mode 4) against mode 1 through the tile kernel
    dy = r(B, T, cout).bfloat16()
    kw2 = dict(ks=ks, pad=ks - 1 - pad, lengths=lengths, in_mask=False, out_mask=True)
    if ops.conv1d_rt_ex_ok(dy, cin, ks, 1, None):
        ref2 = ops.conv1d(dy, ops.pack_conv_weight(w, torch.bfloat16, 1), None, cin, **kw2)
        got3 = ops.conv1d(dy, None, None, cin, wstream=ops.pack_conv_weight(w, torch.bfloat16, 4), **kw2)
        assert float((ref2.float() - got3.float()).abs().max()) <= 2e-2 * float(ref2.float().abs().max())


@pytest.mark.parametrize("B,T,cin,ks,dil,masked,bm", [(9, 1000, 512, 3, 8, True, 128), (19, 900, 512, 3, 1, True, 0), (7, 333, 512, 3, 4, False, 96),
                                                     (40, 130, 512, 3, 2, True, 64), (5, 1531, 256, 5, 1, False, 160)])
def test_row_tile_conv_column_sums_from_the_epilogue(dev, monkeypatch, B, T, cin, ks, dil, masked, bm):
    """ptpp_conv1d_rt_fwd_cs (round 6): the per-utterance column sums of the ROUNDED output -- what the DiffNet backward sums over
    every layer's input gradient (reference modules/denoiser.py:72 differentiated: the step projection is broadcast over time) --
    as per-tile partials from the conv's epilogue.  y unchanged bit for bit; slots a tile covers but does not sum into are
    written as zeros (the buffer arrives poisoned); the slot sums equal a f64 sum of the stored y within f32 rounding, and
    are bit-reproducible."""
    from promptttspp_amd import ops

    monkeypatch.setattr(ops, "CONV_RT_MIN_ROWS", 1)
    if bm:
        monkeypatch.setenv("PTPP_CONV_RT_BM", str(bm))
    x, w, b, res = _mk(dev, B, T, cin, ks, 3 * ks + dil)
    pad = dil * (ks - 1) // 2
    lengths = torch.tensor([max(1, T - 41 * i) for i in range(B)], device=dev, dtype=torch.int32) if masked else None
    assert ops.conv1d_rt_colpart_ok(cin, ks, [dil], B, T)
    wst = ops.pack_conv_weight(w, torch.bfloat16, 3)
    kw = dict(ks=ks, dil=dil, pad=pad, lengths=lengths, in_mask=masked, res=res, res_scale=0.7071067811865476, wstream=wst)
    y0 = ops.conv1d(x, None, b, 256, **kw)
    nslot = (T + 31) // 32
    outs = []
    for _ in range(2):
        part = torch.full((B, nslot, 256), float("nan"), device=dev)
        y1 = ops.conv1d(x, None, b, 256, colpart=part, **kw)
        assert torch.equal(y0, y1)
        assert bool(torch.isfinite(part).all())
        outs.append(ops.colsum_batch(part))
    assert torch.equal(outs[0], outs[1])
    ref = y0.double().sum(1)
    err = float((outs[0].double() - ref).abs().max() / ref.abs().max())
    assert err < 1e-5, err


@pytest.mark.parametrize("B,T,nsplit,drop", [(19, 199, 0, 0.1), (41, 98, 0, 0.0), (3, 150, 2, 0.1), (7, 333, 0, 0.25)])
def test_feed_forward_data_gradient_with_the_relu_backward_in_its_epilogue(dev, monkeypatch, B, T, nsplit, drop):
    """ptpp_conv1d_rt_fwd_ex_relu_bwd (round 6): the Conformer feed-forward backward's  conv (256 -> 1024, k = 9) -> ReLU / dropout
    backward  as one launch (reference modules/esp/transformer/multi_layer_conv.py:52-67 differentiated) against the two launches it
    replaces -- ptpp_conv1d_rt_fwd_ex and ptpp_epilogue_bwd on the stored bf16 result -- BIT for bit, in the fused form (unsplit
    launches) and in the fallback a split launch takes."""
    from promptttspp_amd import ops

    if nsplit:
        monkeypatch.setenv("PTPP_CONV_RT_NSPLIT", str(nsplit))
    g = torch.Generator().manual_seed(77 + B)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    cin, cout, ks, pad = 256, 1024, 9, 4
    x = r(B, T, cin).bfloat16()
    w = r(cout, cin, ks, sc=(cin * ks) ** -0.5)
    saved = torch.relu(r(B, T, cout)).bfloat16()           # the forward's hidden activation: zero where ReLU or dropout cut it
    lengths = torch.tensor([max(1, T - 7 * i) for i in range(B)], device=dev, dtype=torch.int32)
    wst = ops.pack_conv_weight(w, torch.bfloat16, 3)
    assert ops.conv1d_rt_ex_ok(x, cout, ks, 1, None)
    y = ops.conv1d(x, None, None, cout, ks=ks, pad=pad, lengths=lengths, wstream=wst)
    ref = ops.epilogue_bwd(y, saved, lengths, 1.0, True, True, drop, 1)
    got = ops.conv1d_rt_ex_relu_bwd(x, wst, saved, cout, ks, pad, lengths, drop)
    torch.cuda.synchronize()
    assert torch.equal(ref, got), float((ref.float() - got.float()).abs().max())
    assert float(ref.float().abs().max()) > 0
    m = torch.arange(T, device=dev)[None, :] >= lengths[:, None]
    assert float(got[m].float().abs().max()) == 0.0 if bool(m.any()) else True
