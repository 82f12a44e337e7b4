"""Variance adaptor: MDN duration predictor, length regulator, frame-prior network,
pitch/V-UV predictor (reference: promptttspp/modules/variance_adaptor.py:23-206).

Same classes / constructor kwargs / state-dict keys as the reference.  Every
PredictorLayer is two launches (MFMA conv with fused ReLU; fused
LayerNorm+dropout+mask); the length regulator is an integer gather
(``ptpp_length_regulate_*``) instead of a dense (B,Tp,Tf) path matmul, which
makes frame->phone alignment exact by construction.
"""
import torch
import torch.nn as nn

from .. import functional as PF
from .. import ops
from ..config import compute_dtype
from ..layers.norm import LayerNorm
from .mdn import MDNLayer, mdn_get_most_probable_sigma_and_mu


def _lengths_of(mask):
    """(B,1,T) prefix mask -> (B,) int32 lengths"""
    return mask.sum(dim=(1, 2)).to(torch.int32)


class PredictorLayer(nn.Module):
    def __init__(self, channels, kernel_size, dropout):
        super().__init__()
        self.kernel_size, self.p = kernel_size, dropout
        self.conv = nn.Conv1d(channels, channels, kernel_size, padding=kernel_size // 2)
        self.norm = LayerNorm(channels)

    def cl(self, x, lengths):
        """dropout(LN(relu(conv(x)))) * mask; the conv input is NOT masked (the
        reference adds the style embedding to padded phones too, model.py:111)."""
        k = self.kernel_size
        h = PF.conv1d(x, self.conv.weight, self.conv.bias, ks=k, pad=k // 2, act="relu")
        return self.norm.forward_cl(h, lengths=lengths, out_mask=True, drop_out=self.p if self.training else 0.0)


def _predictor_layers(layers, x, lengths, training):
    """All PredictorLayers of a predictor: one autograd node issued by two C calls (functional.ConvLnStackFn) where the
    layers share kernel size and dropout (every reference config), else layer by layer."""
    if len(layers) == 0:
        return x
    l0 = layers[0]
    convs = [l.conv for l in layers]
    if all(l.kernel_size == l0.kernel_size and l.p == l0.p and l.norm.eps == l0.norm.eps for l in layers) \
            and PF.conv_ln_stack_ok(x, convs):
        return PF.conv_ln_stack(x, convs, [l.norm for l in layers], l0.kernel_size, l0.norm.eps, lengths, conv_act="relu",
                                drop_out=l0.p if training else 0.0, out_mask=1)
    for layer in layers:
        x = layer.cl(x, lengths)
    return x


class Predictor(nn.Module):
    def __init__(self, channels, out_channels, kernel_size, dropout, num_layers, detach=False):
        super().__init__()
        self.layers = nn.ModuleList([PredictorLayer(channels, kernel_size, dropout) for _ in range(num_layers)])
        self.out_layer = nn.Conv1d(channels, out_channels, 1)
        self.detach = detach

    def cl(self, x, lengths, as_float=True):
        """(B,T,C) -> (B,T,out_channels) f32 (``as_float=False``: in the compute dtype, for the fused loss), masked."""
        if self.detach:
            x = x.detach()
        x = _predictor_layers(self.layers, x, lengths, self.training)
        ol = self.out_layer
        if ol.bias is not None and PF.linear_small_ok(x, ol.weight):  # 1-4 output channels: one HBM-bound launch each way
            y = PF.linear_small(x, ol.weight, ol.bias, ops.i32(lengths, x.device))
        else:
            y = PF.conv1d(x, ol.weight, ol.bias, lengths=lengths, out_mask=True)
        return y.float() if as_float else y

    def forward(self, x, mask):
        """Reference signature: x (B,C,T), mask (B,1,T) -> (B,out,T)."""
        y = self.cl(ops.bct_to_btc(x, compute_dtype()), _lengths_of(mask))
        return y.transpose(1, 2)

    def infer(self, x, mask):
        return self(x, mask)


class MDNPredictor(nn.Module):
    def __init__(self, channels, out_channels, kernel_size, dropout, num_layers, num_gaussians=4, dim_wise=True,
                 detach=False, disable_amp=False):
        super().__init__()
        self.layers = nn.ModuleList([PredictorLayer(channels, kernel_size, dropout) for _ in range(num_layers)])
        self.out_layer = MDNLayer(channels, out_channels, num_gaussians, dim_wise)
        self.detach, self.disable_amp = detach, disable_amp

    def cl(self, x, lengths, raw=False):
        if self.detach:
            x = x.detach()
        x = _predictor_layers(self.layers, x, lengths, self.training)
        # MDN island: float32 regardless of the compute dtype (``raw``: the heads' outputs before the log-softmax / reshape)
        return self.out_layer.raw(x) if raw else self.out_layer(x)

    def infer_cl(self, x, lengths):
        """log-normal mean of the most probable component -> (B, T) log-duration."""
        sigma, mu = mdn_get_most_probable_sigma_and_mu(*self.cl(x, lengths))
        return (mu + sigma.pow(2).clamp_min(1e-14) / 2).squeeze(-1)

    def forward(self, x, mask):
        return self.cl(ops.bct_to_btc(x, compute_dtype()), _lengths_of(mask))

    def infer(self, x, mask):
        return self.infer_cl(ops.bct_to_btc(x, compute_dtype()), _lengths_of(mask)).unsqueeze(1)


class VarianceAdaptor(nn.Module):
    def __init__(self, duration_predictor, pitch_predictor, pitch_emb, energy_predictor=None, energy_emb=None,
                 frame_prior_network=None):
        super().__init__()
        self.duration_predictor = duration_predictor
        self.pitch_predictor = pitch_predictor
        self.pitch_emb = pitch_emb
        self.energy_predictor = energy_predictor
        self.energy_emb = energy_emb
        self.frame_prior_network = frame_prior_network

    # -- channels-last core --------------------------------------------------------
    def _embed_scalar(self, emb, v, fmask_bt1, dtype):
        """Conv1d(1 -> C, k=1) on a (B, T) scalar track: an outer product (K = 1,
        no GEMM), left to a broadcast multiply-add."""
        w = emb.weight.reshape(1, 1, -1)
        return ((v.unsqueeze(-1) * w + emb.bias.reshape(1, 1, -1)) * fmask_bt1).to(dtype)

    def _add_embedding(self, h, emb, track, flen, fmask_bt1):
        """h + emb(track) * mask: one launch on the GPU when the track needs no gradient (teacher forcing / inference)."""
        if (h.is_cuda and not (track.requires_grad and torch.is_grad_enabled()) and h.shape[-1] % 256 == 0 and h.shape[-1] <= 1024
                and emb.bias is not None):
            return PF.scalar_embed_add(h, track, emb, flen)
        if fmask_bt1 is None:
            fmask_bt1 = (torch.arange(h.shape[1], device=h.device)[None, :] < flen[:, None]).unsqueeze(-1).float()
        return h + self._embed_scalar(emb, track, fmask_bt1, h.dtype)

    def _frames(self, x, durations, flen, Tf, fmask_bt1, log_cf0_in=None, energy_in=None, pitch_stream=None, raw_pitch=False):
        h = PF.length_regulate(x, durations, Tf)
        if self.frame_prior_network is not None:
            h = self.frame_prior_network.forward_cl(h, flen)
        if pitch_stream is not None:
            # training with teacher-forced pitch: the predictor's output feeds only its two losses -- an independent branch,
            # issued (and by autograd differentiated) on its own stream beside the decoder (model.py, PTPP_BRANCH_STREAMS)
            pitch_stream.wait_stream(torch.cuda.current_stream())
            with ops.unpinned(), torch.cuda.stream(pitch_stream):
                pv = self.pitch_predictor.cl(h, flen, as_float=not raw_pitch)
        else:
            pv = self.pitch_predictor.cl(h, flen, as_float=not raw_pitch)  # (B,Tf,2) f32 (raw_pitch: compute dtype)
        if raw_pitch:  # (training with the fused loss: the caller reads the (B,Tf,2) tensor as it is; the embedding is teacher forced)
            assert log_cf0_in is not None and self.energy_predictor is None
            return self._add_embedding(h, self.pitch_emb, log_cf0_in, flen, fmask_bt1), pv, None, None
        log_cf0, vuv = pv[..., 0], pv[..., 1]
        # both predictors read the frame-prior output; the embeddings are added together afterwards
        # (variance_adaptor.py:139-146: energy_predictor(x) runs BEFORE x = x + pitch_emb + energy_emb)
        energy = None
        if self.energy_predictor is not None:
            energy = self.energy_predictor.cl(h, flen)[..., 0]
        h = self._add_embedding(h, self.pitch_emb, log_cf0 if log_cf0_in is None else log_cf0_in, flen, fmask_bt1)
        if self.energy_predictor is not None:
            h = self._add_embedding(h, self.energy_emb, energy if energy_in is None else energy_in, flen, fmask_bt1)
        return h, log_cf0, vuv, energy

    def forward_cl(self, x, plen, flen, fmask_bt1, duration, log_cf0, energy=None, branch_streams=None, raw=False, Tf=None):
        """Training forward.  x (B,Tp,C); duration (B,Tp) frames (integer valued);
        log_cf0 (B,Tf).  Returns (h (B,Tf,C), mdn_out, log_cf0_pred, vuv_pred, energy_pred).
        ``raw`` (the fused-loss path of the model): mdn_out is the duration head's raw output (B,Tp,3G), the second pitch
        slot holds the (B,Tf,2) pitch / V-UV prediction in the compute dtype, and ``fmask_bt1`` may be None (``Tf`` given)."""
        sd, sp = branch_streams if branch_streams is not None else (None, None)
        if sd is not None and self.duration_predictor.detach:
            # the duration predictor reads a DETACHED copy of x and feeds only loss_dur: another independent branch
            sd.wait_stream(torch.cuda.current_stream())
            with ops.unpinned(), torch.cuda.stream(sd):
                dur_out = self.duration_predictor.cl(x, plen, raw=raw)
        else:
            dur_out = self.duration_predictor.cl(x, plen, raw=raw)
        teacher_forced = log_cf0 is not None and self.energy_predictor is None
        h, cf0_p, vuv_p, en_p = self._frames(x, duration, flen, fmask_bt1.shape[1] if Tf is None else Tf, fmask_bt1, log_cf0, energy,
                                             pitch_stream=sp if teacher_forced else None, raw_pitch=raw)
        return h, dur_out, cf0_p, vuv_p, en_p

    def durations_cl(self, x, plen, pmask_bt):
        """Integer durations (B, Tp) int64 of the most probable mixture component (variance_adaptor.py:97-102,178-181)."""
        log_d = self.duration_predictor.infer_cl(x, plen)
        dur = log_d.exp().round().clamp_min(1).long()
        if pmask_bt is not None:
            dur = dur * pmask_bt.to(dur.dtype)
        return dur

    def infer_cl(self, x, plen, pmask_bt, dur=None):
        """Inference.  Returns (h, flen (B,) int64, fmask (B,Tf,1), log_cf0 (B,Tf), vuv, durations (B,Tp) int64).
        ``dur``: durations computed by the caller (the f32 island of the model's bf16 mode)."""
        if dur is None:
            dur = self.durations_cl(x, plen, pmask_bt)
        flen = dur.sum(dim=-1)
        Tf = int(flen.max())  # host sync: the output length is data dependent (as in the reference)
        fmask = (torch.arange(Tf, device=x.device)[None, :] < flen[:, None]).unsqueeze(-1).float()
        h, cf0, vuv, _ = self._frames(x, dur, flen.to(torch.int32), Tf, fmask)
        return h, flen, fmask, cf0, vuv, dur

    # -- reference signatures ((B, C, T) tensors) ------------------------------------
    def forward(self, x, phone_mask, frame_mask, duration, log_cf0, vuv, energy):
        xc = ops.bct_to_btc(x, compute_dtype())
        h, dur_out, cf0_p, vuv_p, en_p = self.forward_cl(
            xc, _lengths_of(phone_mask), _lengths_of(frame_mask), frame_mask.transpose(1, 2).float(),
            duration.squeeze(1), log_cf0.squeeze(1), None if energy is None or self.energy_emb is None else energy.squeeze(1))
        return (ops.btc_to_bct(h), dur_out, cf0_p.unsqueeze(1), vuv_p.unsqueeze(1),
                None if en_p is None else en_p.unsqueeze(1))

    def infer(self, x, phone_mask, return_f0=False):
        return self.infer_batch(x, phone_mask, return_f0, _zero_padded=False)

    def infer_batch(self, x, phone_mask, return_f0=False, _zero_padded=True):
        xc = ops.bct_to_btc(x, compute_dtype())
        plen = _lengths_of(phone_mask)
        h, flen, fmask, cf0, vuv, _ = self.infer_cl(xc, plen, phone_mask.squeeze(1) if _zero_padded else None)
        out = (ops.btc_to_bct(h), fmask.transpose(1, 2).to(x.dtype))
        if return_f0:
            out = out + (cf0.unsqueeze(1), vuv.unsqueeze(1))
        return out
