"""Batch evaluation driver (reference: egs/proposed/bin/synthesize.py:94-217): for every row of the
evaluation label file synthesise the utterance twice -- conditioned on the reference recording's mel and
on the style prompt (+ the speaker prompt) -- and write

    <output_dir>/<spk>/ref/wav/<utt>.wav        <output_dir>/<spk>/prompt/wav/<utt>.wav      <output_dir>/finish

Same config contract (conf/synthesize.yaml), input files and output layout as the reference.  MI355X-first:
the reference walks the rows one at a time; here ``batch_size`` rows (sorted by length so a batch pads
little) go through ONE ``infer_batch`` per leg (``app.synthesize_batch``).  wav I/O is scipy (torchaudio is
not in this image): 32-bit float PCM, what ``torchaudio.save`` writes for a float tensor.

    python egs/proposed/bin/synthesize.py path.root=/data/promptttspp ckpt_path=out/ckpt/last.ckpt \\
        vocoder_ckpt_path=.../last.ckpt output_dir=out/generate batch_size=32
"""
import csv
import os

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory (promptttspp_amd/__init__.py); before the HIP runtime starts
import sys
from pathlib import Path

import numpy as np
import torch
import yaml

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, ROOT)

import app  # noqa: E402
from promptttspp_amd import config as ptpp_config  # noqa: E402
from promptttspp_amd.hydra_lite import compose, instantiate  # noqa: E402
from promptttspp.utils.seed import seed_everything  # noqa: E402

CONF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conf")


def read_prompt_candidate(filepath):
    """``style_key|prompt;prompt;...`` -> {style_key: [lower-cased, stripped prompts]}"""
    out = {}
    with open(filepath, newline="") as fh:
        for key, prompts in csv.reader(fh, delimiter="|"):
            out[key] = [p.lower().strip() for p in prompts.split(";")]
    return out


def read_spk_prompt_candidate(filepath):
    """``spk|word,word,...`` -> {spk: [words]} (speaker ids compared as strings AND ints, like pandas would)"""
    out = {}
    with open(filepath, newline="") as fh:
        for spk, words in csv.reader(fh, delimiter="|"):
            out[spk] = words.split(",")
    return out


def add_spk_prompt(style_prompt, words):
    return f"{style_prompt}. The speaker identity can be described as {words}."


def read_labels(label_file):
    """rows of the evaluation csv: (spk_id, item_name, style_prompt_key, seq)"""
    with open(label_file, newline="") as fh:
        return [(r["spk_id"], r["item_name"], r["style_prompt_key"], r["seq"]) for r in csv.DictReader(fh)]


def load_wav(path):
    from scipy.io import wavfile

    sr, x = wavfile.read(path)
    if x.dtype.kind == "i":
        x = x.astype(np.float32) / float(np.iinfo(x.dtype).max + 1)
    x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    return sr, (x if x.dim() == 1 else x.mean(dim=1))


def save_wav(path, wav, sr):
    from scipy.io import wavfile

    wavfile.write(path, int(sr), wav.detach().cpu().float().numpy())


def prompt_of(style_prompt_key, spk, prompt_candidate, spk_prompt_candidate, use_spk_prompt):
    style = prompt_candidate[style_prompt_key][0]
    if use_spk_prompt and str(spk) in spk_prompt_candidate:
        return add_spk_prompt(style, ", ".join(spk_prompt_candidate[str(spk)]))
    return style


def run(cfg):
    data_root, output_dir = Path(cfg.path.data_root), Path(cfg.output_dir)
    seed_everything(cfg.train.seed)
    prompt_candidate = read_prompt_candidate(cfg.path.prompt_candidate_file)
    spk_prompt_candidate = read_spk_prompt_candidate(cfg.path.spk_prompt_candidate_file)
    mel_stats = yaml.safe_load(open(f"{cfg.path.mel_dir}/stats.yaml"))
    dt = torch.bfloat16 if str(getattr(cfg, "compute_dtype", "bf16")) in ("bf16", "bfloat16") else torch.float32
    ptpp_config.set_compute_dtype(dt)
    device = torch.device("cuda")
    model, vocoder = app.load_model(cfg.model, cfg.ckpt_path, cfg.vocoder, cfg.vocoder_ckpt_path, device)
    if hasattr(vocoder, "set_compute_dtype"):
        vocoder.set_compute_dtype(dt)
    to_mel = instantiate(cfg.transforms).to(device).eval()

    rows = read_labels(cfg.label_file)
    order = sorted(range(len(rows)), key=lambda i: len(rows[i][3].split()))
    bs = int(cfg.batch_size)
    for s in range(0, len(order), bs):
        chunk = [rows[i] for i in order[s : s + bs]]
        ids = [torch.tensor([int(t) for t in seq.split()], dtype=torch.long) for _, _, _, seq in chunk]
        ref_mels = []
        for spk, utt, _, _ in chunk:
            sr, wav = load_wav(data_root / f"{spk}/wav24k/{utt}.wav")
            assert sr == to_mel.sample_rate, f"{utt}: {sr} Hz, the mel front-end expects {to_mel.sample_rate}"
            ref_mels.append(to_mel(wav[None, :].to(device))[0].float().cpu())
        prompts = [prompt_of(key, spk, prompt_candidate, spk_prompt_candidate, cfg.use_spk_prompt)
                   for spk, _, key, _ in chunk]
        legs = (("ref", dict(reference_mels=ref_mels)), ("prompt", dict(style_prompts=prompts)))
        for leg, kw in legs:
            wavs, _ = app.synthesize_batch(model, vocoder, ids, mel_stats=mel_stats, noise_scale=1.0,
                                           batch_vocoder=bool(cfg.batch_vocoder), **kw)
            for (spk, utt, _, _), wav in zip(chunk, wavs):
                for sub in ("mel", "plot", "wav"):
                    (output_dir / str(spk) / leg / sub).mkdir(parents=True, exist_ok=True)
                save_wav(output_dir / str(spk) / leg / "wav" / f"{utt}.wav", wav, to_mel.sample_rate)
    (output_dir / "finish").write_text("finish")
    return len(rows)


def main(argv=None):
    run(compose(CONF, "synthesize", list(argv if argv is not None else sys.argv[1:])))


if __name__ == "__main__":
    main()
