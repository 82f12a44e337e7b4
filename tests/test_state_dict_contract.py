"""CPU: the product modules expose exactly the reference's state-dict names and
shapes (captured in the golden fixtures), built through the YAML config tree."""
import os
import warnings

import pytest
from conftest import ROOT, key_shapes, load_golden

warnings.simplefilter("ignore")
CONF = os.path.join(ROOT, "egs", "proposed", "bin", "conf")


def model_cfg(name="prompttts_mdn_v2_wo_erg_final"):
    from promptttspp_amd import hydra_lite as H

    return H.load_node(os.path.join(CONF, "model", name + ".yaml"))


def keyset(m, drop=()):
    return sorted((k, tuple(v.shape)) for k, v in m.state_dict().items() if not k.endswith(drop))


def test_alias_package_is_the_same_objects():
    import promptttspp.vocoders
    import promptttspp_amd.vocoders

    assert promptttspp.vocoders.BigVGAN is promptttspp_amd.vocoders.BigVGAN
    from promptttspp.modules.esp import ConformerEncoder  # noqa: F401
    from promptttspp.utils.lr_scheduler import NoamLR  # noqa: F401


def test_full_model_contract():
    from promptttspp_amd import hydra_lite as H

    m = H.instantiate(model_cfg())
    ref = sorted(key_shapes(load_golden("model_forward")["keys"]))
    # transformers' BertModel registers position_ids/token_type_ids as non-persistent buffers
    assert keyset(m) == ref
    n = sum(p.numel() for k, p in m.named_parameters() if not k.startswith("prompt_encoder.bert"))
    assert n == 72214110
    trainable_bert = sum(p.numel() for k, p in m.named_parameters() if k.startswith("prompt_encoder.bert") and p.requires_grad)
    assert trainable_bert == 2363904


@pytest.mark.parametrize("node,fixture,key", [("encoder", "conformer", "keys_new"), ("variance_adaptor", "variance_adaptor", "keys"),
                                              ("reference_encoder", "style_encoder", "keys"), ("decoder", "diffusion", "keys"),
                                              ("style_mdn", "mdn", "keys_s")])
def test_submodule_contracts(node, fixture, key):
    from promptttspp_amd import hydra_lite as H

    m = H.instantiate(model_cfg()[node])
    assert keyset(m) == sorted(key_shapes(load_golden(fixture)[key]))


def test_config_overrides_and_interpolation():
    from promptttspp_amd import hydra_lite as H

    cfg = H.compose(CONF, "train", ["model=prompttts_mdn_v2_wo_erg_final", "dataset.max_tokens=30000", "train.fp16=false",
                                    "output_dir=./out/proposed", "train=noam", "path=default", "dataset=mel"])
    assert cfg.dataset.max_tokens == 30000 and cfg.train.lr_scheduler.warmup_steps == 4000
    assert cfg.model.variance_adaptor.duration_predictor.channels == 256
    assert cfg.model.variance_adaptor.pitch_predictor.dropout == 0.5
    assert cfg.model.variance_adaptor.duration_predictor.disable_amp is True
    assert cfg.model.encoder.rel_pos_type == "new"
    demo = H.compose(CONF, "demo")
    assert demo.model.encoder.rel_pos_type == "legacy" and demo.vocoder._target_.endswith("F0AwareBigVGAN")


def test_host_logic_matches_reference():
    import numpy as np

    from promptttspp_amd.datasets.utils import batch_by_size
    from promptttspp_amd.utils.lr_scheduler import noam_scale

    d = load_golden("host_logic")
    fr = d["frames"].numpy()
    order = np.argsort(fr, kind="stable")
    for mt in (10000, 30000):
        for W in (1, 2, 8):
            bs = batch_by_size(order, lambda i: int(fr[i]), max_tokens=mt * W, required_batch_size_multiple=W)
            assert np.array_equal(np.concatenate([np.asarray(b) for b in bs]), d[f"flat_{mt}_{W}"].numpy())
            assert np.array_equal(np.array([len(b) for b in bs]), d[f"sizes_{mt}_{W}"].numpy())
    for s, lr in zip(d["noam_steps"].tolist(), d["noam_lr"].tolist()):
        assert abs(1e-3 * noam_scale(int(s), 4000) - lr) < 1e-12
