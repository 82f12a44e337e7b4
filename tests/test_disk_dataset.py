"""CPU: the disk-backed dataset in the reference's prepared layout (datasets/all_with_spk_prompt_norm.py:25-257)
on a tiny corpus written by the test, through the collator -- and ``dataset=mel`` refusing to train on nothing."""
import csv
import os
import random

import numpy as np
import pytest
import torch
from conftest import ROOT


def _corpus(root, n=5):
    r = np.random.default_rng(0)
    (root / "metadata").mkdir(parents=True)
    (root / "metadata" / "style.csv").write_text(
        "m-low-slow|A man speaks slowly in a low pitch ; A low-pitched male voice, slowly\nf-high-fast|A woman speaks fast\n")
    (root / "metadata" / "spk.csv").write_text("101|deep,calm,warm,steady,clear,soft\n")
    mel_dir, feats = root / "mel63", root / "feats"
    rows = []
    for i in range(n):
        spk = 101 if i % 2 == 0 else 202
        tp = int(r.integers(4, 9))
        d = r.integers(1, 6, size=tp)
        tf = int(d.sum())
        overshoot = i == 3          # the aligner's off-by-one: durations sum to Tf + 1
        for sub in (mel_dir / str(spk), feats / str(spk) / "cf0", feats / str(spk) / "vuv"):
            sub.mkdir(parents=True, exist_ok=True)
        np.save(mel_dir / str(spk) / f"u{i}.npy", (-5 + 2 * r.standard_normal((80, tf))).astype(np.float32))
        np.save(feats / str(spk) / "cf0" / f"u{i}.npy", (5.2 + 0.2 * r.standard_normal(tf)).astype(np.float32))
        np.save(feats / str(spk) / "vuv" / f"u{i}.npy", (r.random(tf) > 0.4).astype(np.float32))
        if overshoot:
            d[-1] += 1
        rows.append([spk, f"u{i}", "M", "very low" if i == 0 else "normal", "very slow" if i == 0 else "normal", "normal",
                     "m-low-slow" if i % 2 == 0 else "f-high-fast", " ".join(map(str, r.integers(3, 89, size=tp))),
                     " ".join(map(str, d)), "unused-extra-column"])
    (mel_dir / "stats.yaml").write_text("mean: -5.0\nstd: 2.0\n")
    with open(root / "trn.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["spk_id", "item_name", "gender", "pitch", "speaking_speed", "energy", "style_prompt_key", "seq",
                    "durations", "extra"])
        w.writerows(rows)
    return rows


def test_disk_dataset_items_and_collation(tmp_path):
    from promptttspp.datasets.all_with_spk_prompt_norm import AllWithSpkPromptNormDataset
    from promptttspp.datasets.prompttts import PromptTTSCollator

    rows = _corpus(tmp_path)
    ds = AllWithSpkPromptNormDataset(tmp_path / "trn.csv", tmp_path, tmp_path / "feats", tmp_path / "mel63", None,
                                     tmp_path / "metadata" / "style.csv", tmp_path / "metadata" / "spk.csv", p_augment=1.0)
    assert len(ds) == 5
    lens = [sum(int(x) for x in r[8].split()) for r in rows]
    assert [ds.num_tokens(i) for i in range(5)] == lens
    assert list(ds.ordered_indices()) == list(np.argsort(np.array(lens), kind="mergesort"))
    random.seed(0)
    items = [ds[i] for i in range(5)]
    for i, (spk, utt, ph, dur, mel, cf0, vuv, en, prompt) in enumerate(items):
        raw = np.load(tmp_path / "mel63" / str(rows[i][0]) / f"u{i}.npy")
        assert utt == f"u{i}" and ph.dtype == torch.long and ph.tolist() == [int(s) for s in rows[i][7].split()]
        assert torch.allclose(mel, torch.from_numpy((raw + 5.0) / 2.0))                 # (mel - mean) / std
        assert torch.allclose(en, torch.from_numpy(np.sqrt((np.exp(raw) ** 2).sum(0))), rtol=1e-5)  # un-normalised
        assert float(dur.sum()) == mel.shape[-1] == cf0.shape[-1] == vuv.shape[-1]
        assert isinstance(prompt, str) and len(prompt) > 3
    assert items[3][3][-1] == int(rows[3][8].split()[-1]) - 1                          # the off-by-one repair
    # augmentation: "very" traits put an adverb in front of the matching words (p_augment = 1)
    assert any(a + " low" in items[0][8] or a + " slowly" in items[0][8] or "speaker" in items[0][8].lower()
               for a in ("very", "extremely", "highly", "really", "particularly"))
    # speaker 202 has no speaker prompt: its style prompt passes through with the final full stop
    assert items[1][8] == "a woman speaks fast."
    # seeded: same prompts (a fresh dataset: the speaker word lists are shuffled IN PLACE, like the reference's)
    ds2 = AllWithSpkPromptNormDataset(tmp_path / "trn.csv", tmp_path, tmp_path / "feats", tmp_path / "mel63", None,
                                      tmp_path / "metadata" / "style.csv", tmp_path / "metadata" / "spk.csv", p_augment=1.0)
    random.seed(0)
    assert [ds2[i][8] for i in range(5)] == [it[8] for it in items]
    batch = PromptTTSCollator()(items)
    assert batch[2].shape[0] == 5 and batch[5].shape[1] == 80 and batch[5].shape[2] == max(it[4].shape[-1] for it in items)
    assert batch[9].tolist() == [it[4].shape[-1] for it in items] and list(batch[10]) == [it[8] for it in items]


def test_dataset_mel_refuses_to_train_without_a_corpus(tmp_path):
    from promptttspp_amd.hydra_lite import compose, instantiate

    cfg = compose(os.path.join(ROOT, "egs", "proposed", "bin", "conf"), "train", [f"path.root={tmp_path}/nowhere"])
    assert cfg.dataset.train._target_.endswith("AllWithSpkPromptNormDataset")
    with pytest.raises(FileNotFoundError, match="dataset=synthetic"):
        instantiate(cfg.dataset.train, to_mel=None)
