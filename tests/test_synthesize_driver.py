"""egs/proposed/bin/synthesize.py (the batch evaluation driver) end to end on the GPU
with a small on-disk dataset in the reference's layout, plus the CPU-side file parsers."""
import csv
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.join(os.path.dirname(__file__), "..")
sys.path.insert(0, ROOT)


def _driver():
    spec = importlib.util.spec_from_file_location("ptpp_synthesize", os.path.join(ROOT, "egs", "proposed", "bin", "synthesize.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _write_metadata(root):
    (root / "metadata").mkdir(parents=True)
    (root / "metadata" / "style_prompt_candidates_v230922.csv").write_text(
        "m-low-slow|A man speaks SLOWLY in a low tone ; a low male voice\nf-high-fast|A woman speaks fast; quick\n")
    (root / "metadata" / "speaker_prompt_candidates_v230922.csv").write_text("101|deep,calm\n")


def test_prompt_files_and_prompt_composition(tmp_path):
    syn = _driver()
    _write_metadata(tmp_path)
    pc = syn.read_prompt_candidate(tmp_path / "metadata" / "style_prompt_candidates_v230922.csv")
    assert pc["m-low-slow"] == ["a man speaks slowly in a low tone", "a low male voice"]  # lower-cased, stripped
    sc = syn.read_spk_prompt_candidate(tmp_path / "metadata" / "speaker_prompt_candidates_v230922.csv")
    assert sc == {"101": ["deep", "calm"]}
    # reference synthesize.py:87-91,139-147: "<style>. The speaker identity can be described as <w1, w2>."
    assert syn.prompt_of("m-low-slow", 101, pc, sc, True) == \
        "a man speaks slowly in a low tone. The speaker identity can be described as deep, calm."
    assert syn.prompt_of("m-low-slow", 101, pc, sc, False) == "a man speaks slowly in a low tone"
    assert syn.prompt_of("f-high-fast", 202, pc, sc, True) == "a woman speaks fast"  # speaker without a prompt


class _FakeTokenizer:
    """Stands in for the BERT vocabulary (not in this image): words -> stable ids, right-padded."""

    def __call__(self, prompts, padding=True, return_tensors="pt"):
        rows = [[101] + [1000 + sum(map(ord, w)) % 20000 for w in p.split()] + [102] for p in prompts]
        n = max(map(len, rows))
        ids = torch.tensor([r + [0] * (n - len(r)) for r in rows])
        am = torch.tensor([[1] * len(r) + [0] * (n - len(r)) for r in rows])

        class Enc(dict):
            def to(self, device):
                return Enc({k: v.to(device) for k, v in self.items()})

        return Enc(input_ids=ids, attention_mask=am)


@pytest.mark.gpu
def test_synthesize_driver_end_to_end(tmp_path):
    from scipy.io import wavfile

    import test_hip_acoustic as T
    from oracle.fill import fill_state_dict
    from promptttspp_amd import config
    from promptttspp_amd.hydra_lite import compose, instantiate
    from promptttspp_amd.modules.prompt_encoder import BertWrapper

    syn = _driver()
    dev = torch.device("cuda:0")
    root = tmp_path / "corpus"
    _write_metadata(root)
    rng = np.random.default_rng(0)
    rows = []
    for k, (spk, key, n_ph, secs) in enumerate([(101, "m-low-slow", 9, 0.8), (101, "f-high-fast", 17, 1.1),
                                                (202, "f-high-fast", 5, 0.6)]):
        d = root / "data_prep" / "out" / "libritts_r_per_spk_cleaned" / str(spk) / "wav24k"
        d.mkdir(parents=True, exist_ok=True)
        wavfile.write(d / f"utt{k}.wav", 24000, (0.1 * rng.standard_normal(int(24000 * secs))).astype(np.float32))
        seq = " ".join(str(int(x)) for x in rng.integers(3, 87, n_ph))
        rows.append([spk, f"utt{k}", "M", "low", "slow", "low", "a prompt", key, seq])
    mel_dir = root / "dump" / "libritts_r_per_spk_cleaned" / "mel63"
    mel_dir.mkdir(parents=True)
    (mel_dir / "stats.yaml").write_text("mean: -5.0\nstd: 2.0\n")
    df = root / "dump" / "libritts_r_per_spk_cleaned" / "df_filtered"
    df.mkdir(parents=True)
    with open(df / "eval_filtered.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["spk_id", "item_name", "gender", "pitch", "speaking_speed", "energy", "style_prompt", "style_prompt_key", "seq"])
        w.writerows(rows)

    # checkpoints in the reference's format: the tamed synthetic acoustic model of the parity tests + a vocoder
    model, _ = T._model(dev)
    torch.save({"model": model.state_dict()}, tmp_path / "am.ckpt")
    conf = os.path.join(ROOT, "egs", "proposed", "bin", "conf")
    voc = instantiate(compose(conf, "synthesize", []).vocoder)
    fill_state_dict(voc, seed=5, overrides={"weight_g": 0.4})
    torch.save({"generator": voc.state_dict()}, tmp_path / "voc.ckpt")
    del model, voc

    old_init = BertWrapper.__init__

    def init_with_fake_vocab(self, *a, **kw):
        old_init(self, *a, **kw)
        self.tokenizer = _FakeTokenizer()

    BertWrapper.__init__ = init_with_fake_vocab
    try:
        outs = {}
        for tag, extra in (("a", []), ("b", []), ("bv", ["batch_vocoder=true"])):  # bv: padded-batch vocoder leg (coverage only:
            # its RNG draws differ from the per-utterance leg, so neither samples nor durations are comparable)
            out = tmp_path / f"gen_{tag}"
            n = syn.run(compose(conf, "synthesize", [f"path.root={root}", f"ckpt_path={tmp_path / 'am.ckpt'}",
                                                     f"vocoder_ckpt_path={tmp_path / 'voc.ckpt'}", f"output_dir={out}",
                                                     "batch_size=2", "compute_dtype=f32"] + extra))
            assert n == 3 and (out / "finish").read_text() == "finish"
            for spk, utt, *_ in rows:
                for leg in ("ref", "prompt"):
                    assert (out / str(spk) / leg / "mel").is_dir() and (out / str(spk) / leg / "plot").is_dir()
                    sr, x = wavfile.read(out / str(spk) / leg / "wav" / f"{utt}.wav")
                    assert sr == 24000 and x.dtype == np.float32 and len(x) > 0 and len(x) % 240 == 0
                    assert np.isfinite(x).all()
                    outs[tag, utt, leg] = x
        for (tag, utt, leg), x in outs.items():
            if tag == "b":  # seeded: a second run reproduces the first bit for bit
                assert np.array_equal(x, outs["a", utt, leg])
    finally:
        BertWrapper.__init__ = old_init
        config.set_compute_dtype(torch.float32)
