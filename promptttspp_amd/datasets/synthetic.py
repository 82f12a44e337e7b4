"""Synthetic LibriTTS-R-shaped utterances: phones per utterance
~ clip(round(lognormal(ln 62, 0.55)), 8, 260), frames per phone 1 + Poisson(7) (edge
tokens 1-3), mel ~ N(0,1) (the reference normalises by global mean/std), log-F0 ~
N(5.2, 0.25^2), voiced runs, BERT prompts as token ids [CLS] U{1000..29999}^(L-2) [SEP].
Deterministic per (seed, index); stands in for the absent corpus."""
import numpy as np
import torch
from torch.utils.data import Dataset


class SyntheticLibriTTSR(Dataset):
    def __init__(self, num_utts=20000, seed=1234, to_mel=None):
        self.seed = seed
        r = np.random.default_rng(seed)
        self.tp = np.clip(np.round(r.lognormal(np.log(62.0), 0.55, num_utts)), 8, 260).astype(np.int64)
        self.dur_seed = r.integers(0, 2**31 - 1, size=num_utts)
        self.lengths = np.array([int(self._durations(i).sum()) for i in range(num_utts)], dtype=np.int64)

    def _durations(self, i):
        r = np.random.default_rng(int(self.dur_seed[i]))
        d = 1 + r.poisson(7, int(self.tp[i]))
        d[0], d[-1] = r.integers(1, 4), r.integers(1, 4)
        return d.astype(np.int64)

    def __len__(self):
        return len(self.tp)

    def num_tokens(self, index):
        return int(self.lengths[index])

    def ordered_indices(self):
        return np.argsort(self.lengths, kind="mergesort")

    def __getitem__(self, i):
        r = np.random.default_rng([self.seed, int(i)])
        d = self._durations(i)
        tp, tf = len(d), int(d.sum())
        ph = r.integers(3, 90, size=tp)
        ph[0], ph[-1] = 1, 2
        mel = r.standard_normal((80, tf)).astype(np.float32)
        cf0 = (5.2 + 0.25 * r.standard_normal((1, tf))).astype(np.float32)
        vuv = np.zeros((1, tf), dtype=np.float32)
        t = 0
        while t < tf:
            run = int(5 + r.integers(0, 40))
            vuv[0, t : t + run] = float(r.random() < 0.65)
            t += run
        L = int(r.integers(12, 49))
        prompt = np.concatenate([[101], r.integers(1000, 30000, size=L - 2), [102]]).astype(np.int64)
        return ("spk", f"utt{i}", torch.from_numpy(ph), torch.from_numpy(d).float().unsqueeze(0), torch.from_numpy(mel),
                torch.from_numpy(cf0), torch.from_numpy(vuv), torch.zeros(1, tf), torch.from_numpy(prompt))
