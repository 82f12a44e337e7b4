"""Disk-backed training corpus in the reference's prepared layout (reference:
promptttspp/datasets/all_with_spk_prompt_norm.py:25-257) -- the data format immediately before the hot path
(SURVEY section 8f n1):

* ``file_path``: csv with the columns spk_id, item_name, gender, pitch, speaking_speed, energy, style_prompt_key,
  seq (space separated phoneme ids), durations (space separated frames per phone);
* ``mel_dir/<spk>/<utt>.npy`` (80, Tf) log-mel, ``mel_dir/stats.yaml`` {mean, std} (global normalisation,
  :180); ``feats_dir/<spk>/cf0/<utt>.npy`` and ``.../vuv/<utt>.npy`` (Tf,) continuous log-F0 and V/UV;
* ``prompt_candidate_file`` ``style_key|prompt;prompt;...`` and ``spk_prompt_candidate_file`` ``spk|word,word,...``.

Items are the 9-tuples ``PromptTTSCollator`` consumes: (spk, utt, phonemes i64 (Tp), durations f32 (Tp),
normalised mel (80,Tf), log_cf0 (Tf), vuv (Tf), energy (Tf), style prompt str).  Host-side code: plain
csv / numpy / yaml (no pandas / omegaconf dependency).  The random prompt composition draws from ``random`` in the
reference's order (candidate choice; augmentation; speaker words shuffle, count, template; combination), so a
seeded run composes the same prompts."""
import csv
import random
from pathlib import Path

import numpy as np
import torch
import yaml

_COLS = ("spk_id", "item_name", "gender", "pitch", "speaking_speed", "energy", "style_prompt_key", "seq", "durations")
_ADVERBS = ["very", "extremely", "highly", "really", "particularly"]
_TEMPLATES = ["The speaker identity can be described as {words}.",
              "The voice characteristics can be described as {words}.",
              "The speaker's voice can be described as {words}."]
# (trait column, words an adverb may be put in front of) -- all_with_spk_prompt_norm.py:98-142
_AUGMENT = (("pitch", (" high pitch ", " high-pitched ", " high-pitched,", " low pitch ", " low-pitched ", " low-pitched,")),
            ("speaking_speed", (" fast ", " quick ", " quickly ", " quickly,", " slow ", " slowly ", " slowly,",
                                " rapidly ", " rapidly,")),
            ("energy", (" loud ", " loudly ", " loudly,", " quiet ", " quietly ", " quietly,")))


def read_prompt_candidate(filepath):
    """``style_key|prompt;prompt;...`` -> {style_key: [lower-cased, stripped prompts]} (:74-86)"""
    out = {}
    with open(filepath, newline="") as fh:
        for key, prompts in csv.reader(fh, delimiter="|"):
            out[key] = [p.lower().strip() for p in prompts.split(";")]
    return out


def read_spk_prompt_candidate(filepath):
    """``spk|word,word,...`` -> {int spk: [words]} (:88-93; pandas parses the speaker column as integers)"""
    out = {}
    with open(filepath, newline="") as fh:
        for spk, words in csv.reader(fh, delimiter="|"):
            out[int(spk)] = words.split(",")
    return out


class AllWithSpkPromptNormDataset(torch.utils.data.Dataset):
    def __init__(self, file_path, data_root, feats_dir, mel_dir, to_mel=None, prompt_candidate_file=None,
                 spk_prompt_candidate_file=None, use_spk_prompt=True, p_augment=0.0):
        for what, p in (("file_path", file_path), ("feats_dir", feats_dir), ("mel_dir", mel_dir),
                        ("prompt_candidate_file", prompt_candidate_file),
                        ("spk_prompt_candidate_file", spk_prompt_candidate_file)):
            if p is None or not Path(str(p)).exists():
                raise FileNotFoundError(
                    f"AllWithSpkPromptNormDataset: {what}={p!r} does not exist -- set path.root to the prepared "
                    "corpus (egs/proposed/bin/conf/path/default.yaml) or train on generated data with dataset=synthetic")
        self.data_root, self.feats_dir, self.mel_dir = Path(str(data_root)), Path(str(feats_dir)), Path(str(mel_dir))
        self.to_mel = to_mel
        self.data, self.lengths = self.read_data(file_path)
        self.prompt_candidate = read_prompt_candidate(prompt_candidate_file)
        self.spk_prompt_candidate = read_spk_prompt_candidate(spk_prompt_candidate_file)
        self.use_spk_prompt, self.p_augment = use_spk_prompt, p_augment
        with open(self.mel_dir / "stats.yaml") as fh:
            self.stats = yaml.safe_load(fh)

    @staticmethod
    def read_data(file_path):
        """rows in _COLS order + the frame count of every utterance (sum of its durations, :48-72)"""
        rows, lengths = [], []
        with open(file_path, newline="") as fh:
            for rec in csv.DictReader(fh):
                rows.append([rec[c] for c in _COLS])
                lengths.append(sum(int(d) for d in rec["durations"].split()))
        return rows, lengths

    # -- prompt composition ------------------------------------------------------------------
    def augment_style_prompt(self, style_prompt, pitch, speaking_speed, energy):
        if random.random() > self.p_augment:
            return style_prompt
        traits = {"pitch": pitch, "speaking_speed": speaking_speed, "energy": energy}
        for col, phrases in _AUGMENT:
            if "very" in traits[col]:
                adverb = random.choice(_ADVERBS)
                for ph in phrases:
                    style_prompt = style_prompt.replace(ph, f" {adverb}{ph}")
        return style_prompt

    @staticmethod
    def words2prompt(words, min_words=5):
        random.shuffle(words)  # in place, like the reference
        n = random.randint(min_words, len(words))
        return random.choice(_TEMPLATES).format(words=", ".join(words[:n]))

    def add_spk_prompt(self, style_prompt, spk_id):
        words = self.spk_prompt_candidate.get(int(spk_id))
        if words is None:
            return style_prompt
        spk_prompt = self.words2prompt(words)
        return random.choice([f"{style_prompt} {spk_prompt}", f"{spk_prompt} {style_prompt}", f"{spk_prompt}",
                              f"{style_prompt}"])

    # -- features ------------------------------------------------------------------------------
    def get_data(self, spk, utt_id, seq, durations):
        phonemes = torch.tensor([int(s) for s in seq.split()], dtype=torch.long)
        dur = torch.tensor([int(d) for d in durations.split()], dtype=torch.float32)
        mel = torch.from_numpy(np.load(self.mel_dir / f"{spk}/{utt_id}.npy")).float()
        mel_norm = (mel - self.stats["mean"]) / self.stats["std"]
        log_cf0 = torch.from_numpy(np.load(self.feats_dir / f"{spk}/cf0/{utt_id}.npy")).float()
        vuv = torch.from_numpy(np.load(self.feats_dir / f"{spk}/vuv/{utt_id}.npy")).float()
        energy = mel.exp().pow(2).sum(dim=0).sqrt().view(-1)  # from the UN-normalised log-mel (:183)
        assert mel.shape[-1] == log_cf0.shape[-1] == vuv.shape[-1] == energy.shape[-1], (spk, utt_id)
        if mel.shape[-1] < dur.sum():  # the aligner may overshoot by one frame (:185-186)
            dur[-1] = dur[-1] - 1
        assert mel.shape[-1] == dur.sum(), (spk, utt_id, mel.shape[-1], float(dur.sum()))
        return spk, utt_id, phonemes, dur, mel_norm, log_cf0, vuv, energy

    def __getitem__(self, idx):
        spk_id, utt_id, _gender, pitch, speed, energy, key, seq, durations = self.data[idx]
        style_prompt = random.choice(self.prompt_candidate[key])
        style_prompt = self.augment_style_prompt(style_prompt, pitch, speed, energy) + "."
        if self.use_spk_prompt:
            style_prompt = self.add_spk_prompt(style_prompt, spk_id)
        return (*self.get_data(spk_id, utt_id, seq, durations), style_prompt)

    def __len__(self):
        return len(self.data)

    def num_tokens(self, index):
        return self.lengths[index]

    def ordered_indices(self):
        return np.argsort(np.asarray(self.lengths), kind="mergesort")
