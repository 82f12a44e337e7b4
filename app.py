"""Synthesis entry point (reference: app.py): load the acoustic model + F0-aware BigVGAN from
``conf/demo.yaml`` and synthesise from a phoneme sequence and a style prompt or a reference mel.

The reference wraps this in a Gradio UI fed by g2p_en / nltk; that text front-end is outside this
build, so ``synthesize`` starts from phoneme ids.  With gradio and the
reference's text package importable, ``build_ui`` offers the same two-tab demo."""
import os

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory (promptttspp_amd/__init__.py); before the HIP runtime starts
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from promptttspp.utils.model import lowpass_filter  # noqa: E402

try:
    from hydra.utils import instantiate
except ImportError:
    from promptttspp_amd.hydra_lite import instantiate

CONF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "egs", "proposed", "bin", "conf")


def load_model(model_cfg, model_ckpt_path, vocoder_cfg, vocoder_ckpt_path, device=None):
    """reference app.py:28-40.  A checkpoint path of None / "" means random initialisation (tests, benchmarks);
    a path that is given but does not exist raises, as ``torch.load`` does in the reference."""
    from promptttspp_amd.modules.prompt_encoder import allow_random_bert

    device = device or torch.device("cuda")
    with allow_random_bert():  # the checkpoint holds prompt_encoder.bert.model.* (or the caller asked for random weights)
        model = instantiate(model_cfg)
    if model_ckpt_path:
        if not os.path.exists(str(model_ckpt_path)):
            raise FileNotFoundError(f"model_ckpt_path={model_ckpt_path!r} does not exist")
        model.load_state_dict(torch.load(model_ckpt_path, map_location="cpu")["model"])
    vocoder = instantiate(vocoder_cfg)
    if vocoder_ckpt_path:
        if not os.path.exists(str(vocoder_ckpt_path)):
            raise FileNotFoundError(f"vocoder_ckpt_path={vocoder_ckpt_path!r} does not exist")
        vocoder.load_state_dict(torch.load(vocoder_ckpt_path, map_location="cpu")["generator"])
    return model.to(device).eval(), vocoder.to(device).eval()


@torch.no_grad()
def synthesize(model, vocoder, phoneme_ids, style_prompt=None, reference_mel=None, mel_stats=None, noise_scale=0.5):
    """reference app.py:50-82 (``onclick_synthesis``) from phoneme ids (1, Tp):
    infer -> 20 Hz low-pass on log-F0 -> F0 -> de-normalise mel -> vocoder.  Returns (wav, mel)."""
    assert style_prompt is not None or reference_mel is not None
    device = next(model.parameters()).device
    mean = mel_stats["mean"] if mel_stats is not None else 0.0
    std = mel_stats["std"] if mel_stats is not None else 1.0
    kw = dict(use_max=True, noise_scale=noise_scale, return_f0=True)
    if style_prompt is not None:
        dec, log_cf0, vuv = model.infer(phoneme_ids.to(device), style_prompt=style_prompt, **kw)
    else:
        dec, log_cf0, vuv = model.infer(phoneme_ids.to(device), reference_mel=((reference_mel - mean) / std).to(device), **kw)
    modfs = int(1.0 / (10 * 0.001))
    f0 = lowpass_filter(log_cf0, modfs, cutoff=20).exp()
    f0[vuv < 0.5] = 0
    dec = dec * std + mean
    return vocoder(dec, f0).squeeze(1).cpu(), dec.cpu()


@torch.no_grad()
def synthesize_batch(model, vocoder, phoneme_ids, style_prompts=None, reference_mels=None, mel_stats=None,
                     noise_scale=0.5, batch_vocoder=False):
    """Many utterances at once (the batch the reference's synthesize.py walks one row at a time):
    ``phoneme_ids``: list of 1-D id tensors; ``style_prompts``: list of str, or ``reference_mels``: list of
    (80, T_i) un-normalised mels.  One ``infer_batch`` for the acoustic model; the vocoder runs per utterance
    on its exact-length mel (identical to the one-row pipeline) or, ``batch_vocoder=True``, once on the padded
    batch (the last few ms of each utterance then see the neighbour padding instead of zeros).
    Returns (list of 1-D wavs, list of (80, Tf_i) de-normalised mels), on the CPU."""
    assert (style_prompts is None) != (reference_mels is None)
    device = next(model.parameters()).device
    mean = mel_stats["mean"] if mel_stats is not None else 0.0
    std = mel_stats["std"] if mel_stats is not None else 1.0
    n = len(phoneme_ids)
    plen = torch.tensor([len(p) for p in phoneme_ids], dtype=torch.long)
    ph = torch.zeros(n, int(plen.max()), dtype=torch.long)
    for i, p in enumerate(phoneme_ids):
        ph[i, : len(p)] = torch.as_tensor(p, dtype=torch.long)
    kw = dict(use_max=True, noise_scale=noise_scale, return_f0=True)
    if style_prompts is not None:
        # a (input_ids, attention_mask) pair of tensors is passed through pre-tokenised (offline boxes
        # have no BERT vocabulary); anything else is a sequence of strings
        pre = isinstance(style_prompts, tuple) and len(style_prompts) == 2 and torch.is_tensor(style_prompts[0])
        mel, cf0, vuv, flen = model.infer_batch(ph.to(device), plen.to(device),
                                                style_prompt=style_prompts if pre else list(style_prompts), **kw)
    else:
        rlen = torch.tensor([m.shape[-1] for m in reference_mels], dtype=torch.long)
        ref = torch.zeros(n, reference_mels[0].shape[0], int(rlen.max()))
        for i, m in enumerate(reference_mels):
            ref[i, :, : m.shape[-1]] = (m - mean) / std
        mel, cf0, vuv, flen = model.infer_batch(ph.to(device), plen.to(device), reference_mel=ref.to(device),
                                                ref_lengths=rlen, **kw)
    flen = [int(x) for x in flen.cpu()]
    modfs = int(1.0 / (10 * 0.001))
    mel = mel * std + mean
    wavs, mels = [], []
    if batch_vocoder:
        f0 = lowpass_filter(cf0, modfs, cutoff=20).exp()  # NB: filters the padded tracks; see the docstring
        f0[vuv < 0.5] = 0
        wav = vocoder(mel, f0).squeeze(1)
        hop = wav.shape[-1] // mel.shape[-1]
        for i in range(n):
            wavs.append(wav[i, : flen[i] * hop].float().cpu())
            mels.append(mel[i, :, : flen[i]].float().cpu())
        return wavs, mels
    for i in range(n):
        T = flen[i]
        f0 = lowpass_filter(cf0[i : i + 1, :, :T], modfs, cutoff=20).exp()
        f0[vuv[i : i + 1, :, :T] < 0.5] = 0
        wavs.append(vocoder(mel[i : i + 1, :, :T].contiguous(), f0.contiguous()).reshape(-1).float().cpu())
        mels.append(mel[i, :, :T].float().cpu())
    return wavs, mels


def build_ui(model, vocoder, to_mel, mel_stats):  # pragma: no cover - needs gradio + the text front-end
    import gradio as gr
    from g2p_en import G2p
    from promptttspp.text.eng import symbols, text_to_sequence

    g2p = G2p()

    def ids_of(text):
        ph = [p if p not in [",", "."] else "sil" for p in g2p(text)]
        return torch.LongTensor(text_to_sequence(" ".join(p for p in ph if p in symbols)))[None, :]

    def with_prompt(content, style):
        wav, _ = synthesize(model, vocoder, ids_of(content), style_prompt=[style], mel_stats=mel_stats)
        return to_mel.sample_rate, wav.squeeze().numpy()

    with gr.Blocks() as demo:
        gr.Markdown("# PromptTTS++ (MI355X build)")
        content = gr.Textbox("This is text to speech demo.", lines=3, label="Content prompt")
        style = gr.Textbox("A man speaks slowly in a low tone.", lines=3, label="Style prompt")
        button = gr.Button("Synthesize")
        wav = gr.Audio(label="Output wav")
        button.click(with_prompt, inputs=[content, style], outputs=[wav])
    demo.launch()


def main(argv=None):
    from promptttspp_amd.hydra_lite import compose

    cfg = compose(CONF, "demo", list(argv if argv is not None else sys.argv[1:]))
    model, vocoder = load_model(cfg.model, cfg.model_ckpt_path, cfg.vocoder, cfg.vocoder_ckpt_path)
    to_mel = instantiate(cfg.transforms)
    stats = None
    if os.path.exists(str(cfg.mel_stats_file)):
        import yaml

        stats = yaml.safe_load(open(cfg.mel_stats_file))
    build_ui(model, vocoder, to_mel, stats)


if __name__ == "__main__":
    main()
