"""Training entry point (reference: egs/proposed/bin/train.py): composes ``conf/train.yaml`` with
Hydra-style ``key=value`` overrides and runs ``promptttspp.trainers.tts.TTSTrainer``.

    python egs/proposed/bin/train.py dataset=synthetic optimizer=fused_adamw output_dir=./out
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 egs/proposed/bin/train.py dataset=synthetic

With hydra-core installed this is the reference's ``@hydra.main`` program; without it (this
image) the in-tree composer ``promptttspp_amd.hydra_lite`` reads the same YAML tree."""
import os

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # kernel arguments in device memory (promptttspp_amd/__init__.py); before the HIP runtime starts
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "..")))

from promptttspp.trainers.tts import TTSTrainer  # noqa: E402  (alias package of promptttspp_amd)

CONF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conf")

try:
    import hydra

    @hydra.main(version_base=None, config_path="conf/", config_name="train")
    def main(cfg):
        TTSTrainer(cfg).run()
except ImportError:
    from promptttspp_amd.hydra_lite import compose

    def main():
        TTSTrainer(compose(CONF, "train", sys.argv[1:])).run()


if __name__ == "__main__":
    main()
