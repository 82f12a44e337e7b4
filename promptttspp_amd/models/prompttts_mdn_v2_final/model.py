"""PromptTTS++ acoustic model on the MI355X HIP path (reference:
promptttspp/models/prompttts_mdn_v2_final/model.py:28-344).

Same class name, constructor, methods (``forward`` / ``infer`` / ``infer_batch`` /
``sample_style_emb`` / ``generate_style_emb``), loss-dict keys and state-dict
layout as the reference, so ``hydra.utils.instantiate(cfg.model)`` and reference
checkpoints work unchanged.  Internally every per-phone / per-frame tensor is
channels-last (B, T, C) in the compute dtype and masks are int32 lengths; the
(B, C, T) layout only exists at the public method boundary.

Deliberate deviations from the reference's side effects (results identical):
* ``forward`` does not mutate the caller's ``duration`` tensor (the reference's
  ``to_log_scale`` does);
* the prompt may be ``List[str]`` or pre-tokenised ``(input_ids, attention_mask)``.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import functional as PF
from ... import ops
from ...config import compute_dtype
from ...modules.diffusion import GaussianDiffusion
from ...modules.esp import ConformerEncoder
from ...modules.transformer import Transformer
from ...modules.mdn import MDNLayer, mdn_get_most_probable_sigma_and_mu, mdn_loss, mdn_sample_sigma_and_mu
from ...utils.model import sequence_mask


# "" / "0" off | "1" prompt branch | "2" (default) + reference encoder, on two extra streams (see forward) | "3" EXPERIMENTAL: +
# duration and pitch predictors -- 0.13 ms faster, but with direct gradient accumulation 4 of 6 repeated runs of
# test_direct_gradient_accumulation_equals_autograd produced a wrong duration-predictor weight gradient (an unordered
# dependency that is not found yet; modes 0 / 1 / 2: 6 of 6 clean), so it is never the default
BRANCH_STREAMS = os.environ.get("PTPP_BRANCH_STREAMS", "2")
REJOIN = not os.environ.get("PTPP_NO_REJOIN")  # the reference-encoder branch is differentiated from its join (functional.RejoinBranchFn)
EARLY_QSAMPLE = not os.environ.get("PTPP_QSAMPLE_LATE")  # the decoder's draws + q_sample before the reference-encoder join
PROMPT_FIRST = not os.environ.get("PTPP_PROMPT_LATE")  # the prompt branch is issued at the start of the forward
FUSED_GLUE = not os.environ.get("PTPP_NO_FUSED_GLUE")  # (tests compare the fused training forward with the general one)
JOIN_PROBE = None  # tools/diag_joins.py sets a list: (name, event on the waiting stream before the wait, event at the branch's end)


def _probe(name, waiting, branch):
    if JOIN_PROBE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(waiting)
        e1.record(branch)
        JOIN_PROBE.append((name, e0, e1))
if BRANCH_STREAMS in ("0", "off", "no"):
    BRANCH_STREAMS = ""
_branch = {}


def _branch_stream(dev, idx=0):
    """Extra streams per device for the independent branches of the training forward (0: prompt branch, 1: reference
    encoder), each with a slab of its own in the caching allocator (a stream's free blocks serve only that stream: without
    it every new batch shape grows the pool by hipMalloc inside the step -- what made round 1's multi-stream forward slower)."""
    st = _branch.get((dev, idx))
    if st is None:
        st = _branch[(dev, idx)] = ops.aux_stream(dev, idx)
        gib = float(os.environ.get("PTPP_BRANCH_RESERVE_GIB", "2" if idx == 0 else "6"))
        try:  # (a smaller or shared GPU: go without the reservation, the pool then grows on demand)
            free = torch.cuda.mem_get_info(dev)[0]
            gib = min(gib, 0.1 * free / (1 << 30))
            with torch.cuda.stream(st):
                slab = torch.empty(int(gib * (1 << 30)), device=dev, dtype=torch.uint8)
                del slab
        except RuntimeError:
            pass
        PF.register_gradient_stream(st)  # bucket collectives of the data-parallel reducer wait for its gradient kernels too
        if idx == 0:  # the prompt branch's backward ends ~7 ms before the phone encoder's starts: its stream takes part of those
            PF.offer_second_wgrad_stream(st)  # weight gradients (functional.second_wgrad_stream)
    return st


class PromptTTSMDNDurCFG(nn.Module):
    def __init__(self, phoneme_embedding, encoder, variance_adaptor, reference_encoder, prompt_encoder, decoder,
                 out_conv=None, style_mdn=None, norm_style_emb=False, mdn_disable_amp=False, loss_dec_scale=8.0):
        super().__init__()
        self.phoneme_emb = phoneme_embedding
        self.encoder = encoder
        self.variance_adaptor = variance_adaptor
        self.reference_encoder = reference_encoder
        self.prompt_encoder = prompt_encoder
        self.style_mdn = style_mdn
        self.decoder = decoder
        self.out_conv = out_conv
        self.norm_style_emb = norm_style_emb
        self.mdn_disable_amp = mdn_disable_amp
        self.loss_dec_scale = loss_dec_scale
        if not isinstance(encoder, (ConformerEncoder, Transformer)) or \
                not isinstance(decoder, (GaussianDiffusion, ConformerEncoder)):
            raise NotImplementedError(
                "promptttspp_amd implements encoder=ConformerEncoder (prompttts_mdn_v2_wo_erg_final.yaml) or "
                "modules.transformer.Transformer (model.py:95) with decoder=GaussianDiffusion(DiffNet) or "
                "decoder=ConformerEncoder + out_conv (model.py:123-126)")
        self.conformer_decoder = isinstance(decoder, ConformerEncoder)
        if self.conformer_decoder:
            assert out_conv is not None  # reference model.py:66-67
        assert self.variance_adaptor.frame_prior_network is not None

    # ---------------------------------------------------------------------------------
    def _encode(self, phoneme, phone_lengths):
        """ids (B,Tp) -> encoder output (B,Tp,C) channels-last + masks."""
        dt = compute_dtype()
        Tp = phoneme.shape[-1]
        plen = phone_lengths.to(device=phoneme.device, dtype=torch.int32)
        pmask = (torch.arange(Tp, device=phoneme.device)[None, :] < plen[:, None])  # (B,Tp) bool
        pm1 = pmask.unsqueeze(-1).float()
        x = self.phoneme_emb.forward_cl(phoneme, pm1, dt)
        if isinstance(self.encoder, ConformerEncoder):
            x = self.encoder.forward_cl(x.contiguous(), plen, pm1)
        else:  # the FFT-block Transformer plug-in: encoder(x, phone_mask) (model.py:95)
            x = self.encoder.forward_cl(x.contiguous(), plen)
        return x, plen, pmask

    def _decode_conformer(self, h, flen, fm1):
        """(B,Tf,C) -> out_conv(decoder(h)) * frame_mask, (B,Tf,80) channels-last (model.py:124-125)."""
        oc = self.out_conv
        y = self.decoder.forward_cl(h.contiguous(), flen, fm1)
        return PF.conv1d(y, oc.weight, oc.bias, ks=oc.kernel_size[0], dil=oc.dilation[0], pad=oc.padding[0],
                         lengths=flen, out_mask=True)

    def _norm_style(self, e):
        if not self.norm_style_emb:
            return e
        return PF.l2_normalize_channels(e) if e.is_cuda else F.normalize(e, dim=1)

    def _fused_glue_ok(self, dev):
        """The reference's training configuration (prompttts_mdn_v2_wo_erg_final*.yaml) on the GPU: diffusion decoder, both MDN
        heads dimension-wise, no energy branch -- its masks, losses and small tensor chains run as single launches
        (``_forward_fused``).  Anything else takes the general path below."""
        va = self.variance_adaptor
        dp = va.duration_predictor
        return (FUSED_GLUE and dev.type == "cuda" and not self.conformer_decoder and self.style_mdn is not None
                and self.style_mdn.dim_wise and va.energy_predictor is None and va.energy_emb is None
                and isinstance(getattr(dp, "out_layer", None), MDNLayer) and dp.out_layer.dim_wise and dp.out_layer.out_dim == 1
                and getattr(va.pitch_predictor.out_layer, "out_channels", 0) == 2
                and compute_dtype() in (torch.float32, torch.bfloat16) and isinstance(self.encoder, ConformerEncoder))

    def _forward_fused(self, batch):
        """``forward`` for the configuration of ``_fused_glue_ok``: the same graph with the glue as single launches -- phoneme
        embedding + mask (1), style broadcast add (1), pitch embedding (1), q_sample on the (B, M, T) mel (1), step embedding
        (sinusoid 1 + Mish 1), and ALL losses with both MDN log-softmaxes as one launch forward and one backward
        (functional.TtsLossesFn; reference model.py:126-183).  No mask tensors are built."""
        (phoneme, duration, phone_lengths, mel, log_cf0, vuv, energy, frame_lengths, prompt) = batch
        dev = phoneme.device
        dt = compute_dtype()
        branches = BRANCH_STREAMS and self.training
        if branches:
            import ctypes

            PF._direct["main"] = torch.cuda.current_stream()
            PF._direct["main_h"] = ctypes.c_void_p(PF._direct["main"].cuda_stream)
        if self.training and not getattr(self, "_late_marked", False):
            # operands first read after the phone encoder: their share of the per-step weight re-pack runs beside it (functional._late)
            PF.mark_late_pack([*self.variance_adaptor.parameters(), *self.decoder.parameters()])
            self._late_marked = True
        sa = _branch_stream(dev, 1) if (branches and BRANCH_STREAMS in ("2", "3")) else None
        if sa is not None:
            sa.wait_stream(torch.cuda.current_stream())
            with ops.unpinned(), torch.cuda.stream(sa):
                style_emb = self._norm_style(self.reference_encoder(mel, frame_lengths))  # (B,C,1) f32
        # the prompt branch (BERT -> adaptor -> style MDN head) depends on the prompt only and feeds only the style loss: issued
        # now (round 5 issued it after the phoneme encoder: the main stream then waited ~0.7 ms for it at the losses)
        bs = _branch_stream(dev, 0) if branches else None
        if PROMPT_FIRST:  # (same program order -- dropout seeds -- with and without the branch stream)
            if bs is not None:
                bs.wait_stream(torch.cuda.current_stream())
                with ops.unpinned(), torch.cuda.stream(bs):
                    prompt_emb = self._norm_style(self.prompt_encoder(prompt, dev))
                    y_sty = self.style_mdn.raw(prompt_emb.transpose(-1, -2))
            else:
                prompt_emb = self._norm_style(self.prompt_encoder(prompt, dev))
                y_sty = self.style_mdn.raw(prompt_emb.transpose(-1, -2))
        plen = phone_lengths.to(device=dev, dtype=torch.int32)
        flen = frame_lengths.to(device=dev, dtype=torch.int32)
        x = self.phoneme_emb.forward_cl(phoneme, None, dt, lengths=plen)
        x = self.encoder.forward_cl(x, plen, None)
        Tf = mel.shape[-1]
        # (the decoder's draws and q_sample need neither branch: issued here, they fill the main stream's wait for the join below)
        prep = self.decoder.prepare_bct(mel, dt) if EARLY_QSAMPLE else None
        if sa is not None:
            _probe("reference encoder -> x + style_emb", torch.cuda.current_stream(), sa)
            torch.cuda.current_stream().wait_stream(sa)
            style_emb.record_stream(torch.cuda.current_stream())
            if REJOIN:  # (the branch's backward is scheduled from HERE, not after the whole encoder backward: RejoinBranchFn)
                style_emb = PF.rejoin_branch(style_emb, sa)
        else:
            style_emb = self._norm_style(self.reference_encoder(mel, frame_lengths))
        x = PF.bcast_add_rows(x, style_emb.float().reshape(style_emb.shape[0], -1))  # every phone, padded ones too (model.py:111)
        if not PROMPT_FIRST:
            if bs is not None:
                bs.wait_stream(torch.cuda.current_stream())
                with ops.unpinned(), torch.cuda.stream(bs):
                    prompt_emb = self._norm_style(self.prompt_encoder(prompt, dev))
                    y_sty = self.style_mdn.raw(prompt_emb.transpose(-1, -2))
            else:
                prompt_emb = self._norm_style(self.prompt_encoder(prompt, dev))
                y_sty = self.style_mdn.raw(prompt_emb.transpose(-1, -2))
        vb = (bs, sa) if (branches and BRANCH_STREAMS == "3") else None
        h, y_dur, pv, _, _ = self.variance_adaptor.forward_cl(x, plen, flen, None, duration.squeeze(1), log_cf0.squeeze(1), None,
                                                              branch_streams=vb, raw=True, Tf=Tf)
        noise, pred = self.decoder.forward_bct(h, mel, flen, prep=prep)
        if bs is not None:  # join: the losses read the branches' outputs
            main = torch.cuda.current_stream()
            _probe("prompt branch -> losses", main, bs)
            main.wait_stream(bs)
            y_sty.record_stream(main)
            if REJOIN:
                # The prompt branch was issued FIRST, so autograd would differentiate it LAST: its ~30 launches and their weight
                # gradients then sit at the very end of the side stream's queue although their inputs exist 8 ms earlier
                # (profiles/r06_step_tail_ws2.txt).  Re-joined here, its backward is the first thing the pass enqueues.
                y_sty = PF.rejoin_branch(y_sty, bs)
            if vb is not None:
                main.wait_stream(sa)
                y_dur.record_stream(main)
                pv.record_stream(main)
        sm, dl = self.style_mdn, self.variance_adaptor.duration_predictor.out_layer
        total, comps = PF.tts_losses(pred, pv, y_dur, y_sty, noise, flen, log_cf0.squeeze(1).float().contiguous(),
                                     vuv.squeeze(1).float().contiguous(), duration.squeeze(1).float().contiguous(), plen,
                                     style_emb.detach().float().reshape(style_emb.shape[0], -1), dl.num_gaussians, sm.num_gaussians,
                                     self.loss_dec_scale)
        return dict(loss=total, dec=comps[0], dur=comps[1], cf0=comps[2], vuv=comps[3], style=comps[4])

    def forward(self, batch):
        (phoneme, duration, phone_lengths, mel, log_cf0, vuv, energy, frame_lengths, prompt) = batch
        dev = phoneme.device
        if self._fused_glue_ok(dev):
            return self._forward_fused(batch)
        dt = compute_dtype()
        branches = BRANCH_STREAMS and self.training and dev.type == "cuda"
        if branches:
            import ctypes

            PF._direct["main"] = torch.cuda.current_stream()
            PF._direct["main_h"] = ctypes.c_void_p(PF._direct["main"].cuda_stream)
        sa = _branch_stream(dev, 1) if (branches and BRANCH_STREAMS in ("2", "3")) else None
        if sa is not None:
            # the reference encoder (mel -> style embedding) and the phoneme encoder are independent until x + style_emb:
            # two chains of short launches side by side (PTPP_BRANCH_STREAMS=2)
            sa.wait_stream(torch.cuda.current_stream())
            with ops.unpinned(), torch.cuda.stream(sa):
                style_emb = self._norm_style(self.reference_encoder(mel, frame_lengths))  # (B,C,1) f32
        x, plen, pmask = self._encode(phoneme, phone_lengths)

        Tf = mel.shape[-1]
        flen = frame_lengths.to(device=dev, dtype=torch.int32)
        fmask = (torch.arange(Tf, device=dev)[None, :] < flen[:, None])  # (B,Tf)
        fm1 = fmask.unsqueeze(-1).float()
        n_frames = fm1.sum()

        if sa is not None:
            _probe("reference encoder -> x + style_emb", torch.cuda.current_stream(), sa)
            torch.cuda.current_stream().wait_stream(sa)
            style_emb.record_stream(torch.cuda.current_stream())
        else:
            style_emb = self._norm_style(self.reference_encoder(mel, frame_lengths))  # (B,C,1) f32
        # The prompt branch (BERT -> adaptor -> style MDN head) feeds nothing but loss_style in training (model.py:147-163):
        # an independent chain of ~100 short launches, issued -- forward here, backward by autograd on the same stream --
        # beside the main chain (PTPP_BRANCH_STREAMS)
        bs = _branch_stream(dev, 0) if branches else None
        if bs is not None:
            bs.wait_stream(torch.cuda.current_stream())
            with ops.unpinned(), torch.cuda.stream(bs):
                prompt_emb = self._norm_style(self.prompt_encoder(prompt, dev))
                style_mdn_out = self.style_mdn(prompt_emb.transpose(-1, -2)) if self.style_mdn is not None else None
        else:
            prompt_emb = self._norm_style(self.prompt_encoder(prompt, dev))
            if self.style_mdn is not None:
                style_mdn_out = self.style_mdn(prompt_emb.transpose(-1, -2))
        x = x + style_emb.transpose(1, 2).to(dt)  # broadcast over every phone, padded ones too (model.py:111)

        # "3": also the duration predictor (detached input -> loss_dur only) and the pitch predictor (teacher-forced pitch in
        # training -> its two losses only) as branches, on the two streams above
        vb = (bs, sa) if (branches and BRANCH_STREAMS == "3") else None
        h, dur_out, cf0_pred, vuv_pred, energy_pred = self.variance_adaptor.forward_cl(
            x, plen, flen, fm1, duration.squeeze(1), log_cf0.squeeze(1),
            None if self.variance_adaptor.energy_emb is None else energy.squeeze(1), branch_streams=vb)

        mel_cl = mel.transpose(1, 2).float().contiguous()
        if self.conformer_decoder:
            # non-diffusion branch (model.py:123-126): Conformer over the frame sequence, out_conv, L1 to the mel
            pred = self._decode_conformer(h, flen, fm1)
            loss_dec = (pred.float() - mel_cl).abs().sum() / n_frames / self.loss_dec_scale
        else:
            if dev.type == "cuda":
                # (the prediction stays in the compute dtype: the fused loss reads it as it is, one launch each way)
                noise, pred = self.decoder.forward_cl(h, mel_cl, flen, pred_f32=False)
                loss_dec = PF.masked_l1_mean(pred, noise, fm1, n_frames, self.loss_dec_scale)
            else:
                noise, pred = self.decoder.forward_cl(h, mel_cl, flen)
                loss_dec = ((noise - pred) * fm1).abs().sum() / n_frames / self.loss_dec_scale

        if bs is not None:  # join: the losses read the branches' outputs
            main = torch.cuda.current_stream()
            _probe("prompt branch -> losses", main, bs)
            main.wait_stream(bs)
            for t in ((prompt_emb,) if style_mdn_out is None else tuple(style_mdn_out)):
                t.record_stream(main)
            if vb is not None:
                main.wait_stream(sa)
                for t in tuple(dur_out) + (cf0_pred, vuv_pred):
                    t.record_stream(main)
        dur = duration.squeeze(1).float()
        log_dur = torch.where(dur != 0, torch.log(dur.clamp_min(1e-30)), dur)  # to_log_scale, out of place
        pmb = pmask.unsqueeze(-1)
        # reference: .masked_select(mask).mean() -- a dynamic-size gather whose size the host must
        # wait for; the masked mean below is the same number without the device sync
        nll = mdn_loss(*dur_out, log_dur.unsqueeze(-1), reduce=False, mask=pmb)
        loss_dur = torch.where(pmb, nll, torch.zeros_like(nll)).sum() / pmb.sum()

        if dev.type == "cuda":
            loss_cf0 = PF.masked_l1_mean(cf0_pred, log_cf0.squeeze(1), None, n_frames)
            loss_vuv = PF.masked_l1_mean(vuv_pred, vuv.squeeze(1), None, n_frames)
        else:
            loss_cf0 = (cf0_pred - log_cf0.squeeze(1)).abs().sum() / n_frames
            loss_vuv = (vuv_pred - vuv.squeeze(1)).abs().sum() / n_frames
        if self.style_mdn is not None:
            loss_style = mdn_loss(*style_mdn_out, style_emb.detach().transpose(-1, -2)).mean()
        else:
            loss_style = (style_emb.detach() - prompt_emb).pow(2).mean()

        loss = loss_dec + loss_dur + loss_cf0 + loss_vuv + loss_style
        out = dict(loss=loss, dec=loss_dec, dur=loss_dur, cf0=loss_cf0, vuv=loss_vuv, style=loss_style)
        if energy_pred is not None:
            loss_energy = PF.masked_l1_mean(energy_pred, energy.squeeze(1), None, n_frames) if dev.type == "cuda" else \
                (energy_pred - energy.squeeze(1)).abs().sum() / n_frames
            out["loss"] = loss + loss_energy
            out["energy"] = loss_energy
        return out

    # ---------------------------------------------------------------------------------
    def sample_style_emb(self, log_pi, log_sigma, mu, noise_scale, use_max):
        if use_max:
            sigma, mu = mdn_get_most_probable_sigma_and_mu(log_pi, log_sigma, mu)
        else:
            sigma, mu = mdn_sample_sigma_and_mu(log_pi, log_sigma, mu)
        style_emb = mu + sigma * torch.randn_like(sigma) * noise_scale
        if self.norm_style_emb:
            style_emb = F.normalize(style_emb, dim=-1)
        return style_emb.transpose(-1, -2)

    def _style(self, style_prompt, reference_mel, ref_lengths, device, use_max, noise_scale):
        assert (style_prompt is not None) ^ (reference_mel is not None), "One of style inputs must not be None."
        if self.integer_island and not self.training and compute_dtype() != torch.float32:
            from ...config import use_dtype

            with use_dtype(torch.float32):  # (the durations depend on the style embedding: see integer_island)
                return self._style_in_dtype(style_prompt, reference_mel, ref_lengths, device, use_max, noise_scale)
        return self._style_in_dtype(style_prompt, reference_mel, ref_lengths, device, use_max, noise_scale)

    def _style_in_dtype(self, style_prompt, reference_mel, ref_lengths, device, use_max, noise_scale):
        if style_prompt is not None:
            emb = self._norm_style(self.prompt_encoder(style_prompt, device))
            if self.style_mdn is not None:
                emb = self.sample_style_emb(*self.style_mdn(emb.transpose(-1, -2)), noise_scale=noise_scale,
                                            use_max=use_max)
            return emb
        return self._norm_style(self.reference_encoder(reference_mel, ref_lengths))

    # Integer island of the bf16 mode: everything the INTEGER outputs depend on -- style embedding (prompt or reference
    # encoder), phoneme embedding, phoneme encoder, duration predictor + MDN head, exp / round -- runs in float32 at
    # inference, whatever the compute dtype: the durations and frame lengths are then bit for bit those of the f32 mode
    # (which are the reference's, tests/test_hip_acoustic.py).  Phone-level work: a few thousand rows, ~1 % of a synthesis
    # call; the frame-level path and the sampler keep the compute dtype.
    integer_island = True
    # ... and the rest of the conditioning path with it (pitch predictor, frame prior network: ONE pass over the frames,
    # 2-3 % of a synthesis call whose time is the sampler's 100 denoiser evaluations): the bf16 mode's mel error came from
    # here -- five / six bf16 ReLU -> LayerNorm layers moved log-F0 by 4 % and the decoder's conditioning with it (mel MSE
    # 4.9e-3 against the f32 reference), while 100 bf16 denoiser evaluations on an exact conditioning cost 6e-5
    # (tools/experiments/bf16_sampler_emulation.py).  The decoder and the vocoder keep the compute dtype.
    f32_conditioning = True

    @torch.no_grad()
    def _synthesize(self, phoneme, phone_lengths, style_emb, zero_padded_durations, noise_fn=None):
        from ...config import use_dtype

        dt = compute_dtype()
        if dt == torch.float16 and not (self.integer_island and self.f32_conditioning and not self.conformer_decoder):
            raise NotImplementedError("compute dtype float16 serves the diffusion decoder's sampler only (csrc/diffnet_layer.hip, "
                                      "sampler_head.hip): keep integer_island and f32_conditioning on (the text -> conditioning path in f32)")
        dur = None
        pm = lambda pmask: pmask if zero_padded_durations else None  # noqa: E731
        if self.integer_island and dt != torch.float32:
            with use_dtype(torch.float32):
                x, plen, pmask = self._encode(phoneme, phone_lengths)
                x = x + style_emb.transpose(1, 2).to(x.dtype)
                if self.f32_conditioning:
                    h, flen, fm1, cf0, vuv, dur = self.variance_adaptor.infer_cl(x, plen, pm(pmask))
                else:
                    dur = self.variance_adaptor.durations_cl(x, plen, pm(pmask))
            if self.f32_conditioning:
                h = h.to(dt)
            else:
                h, flen, fm1, cf0, vuv, dur = self.variance_adaptor.infer_cl(x.to(dt), plen, pm(pmask), dur=dur)
        else:
            x, plen, pmask = self._encode(phoneme, phone_lengths)
            x = x + style_emb.transpose(1, 2).to(x.dtype)
            h, flen, fm1, cf0, vuv, dur = self.variance_adaptor.infer_cl(x, plen, pm(pmask))
        if self.conformer_decoder:
            mel = self._decode_conformer(h, flen, fm1).float()
        else:
            mel = self.decoder.inference_cl(h, noise_fn) * fm1  # (B,Tf,80) f32
        self.last_durations = dur
        return mel.transpose(1, 2).contiguous(), cf0.unsqueeze(1), vuv.unsqueeze(1), flen

    @torch.no_grad()
    def infer(self, x, style_prompt=None, reference_mel=None, use_max=True, noise_scale=1.0, return_f0=False,
              noise_fn=None):
        """x: (1, L) phoneme ids -> mel (1, 80, Tf) [, log_cf0, vuv]."""
        lengths = torch.full((x.shape[0],), x.shape[-1], device=x.device, dtype=torch.long)
        ref_len = None if reference_mel is None else torch.LongTensor([reference_mel.shape[-1]])
        style = self._style(style_prompt, reference_mel, ref_len, x.device, use_max, noise_scale)
        mel, cf0, vuv, _ = self._synthesize(x, lengths, style, False, noise_fn)
        return (mel, cf0, vuv) if return_f0 else mel

    @torch.no_grad()
    def infer_batch(self, phoneme, phone_lengths, style_prompt=None, reference_mel=None, ref_lengths=None,
                    use_max=True, noise_scale=1.0, return_f0=False, noise_fn=None):
        if reference_mel is not None:
            assert ref_lengths is not None
        style = self._style(style_prompt, reference_mel, ref_lengths, phoneme.device, use_max, noise_scale)
        mel, cf0, vuv, flen = self._synthesize(phoneme, phone_lengths, style, True, noise_fn)
        return (mel, cf0, vuv, flen) if return_f0 else (mel, flen)

    @torch.no_grad()
    def generate_style_emb(self, style_prompt, reference_mel, use_max=True, noise_scale=1.0):
        prompt_emb = self._norm_style(self.prompt_encoder(style_prompt, reference_mel.device))
        if self.style_mdn is not None:
            prompt_emb = self.sample_style_emb(*self.style_mdn(prompt_emb.transpose(-1, -2)), noise_scale=noise_scale,
                                               use_max=use_max)
        prompt_emb = self._norm_style(prompt_emb)
        ref_emb = self._norm_style(self.reference_encoder(reference_mel, torch.LongTensor([reference_mel.shape[-1]])))
        return prompt_emb, ref_emb
