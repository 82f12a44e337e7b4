"""Build-container only: dump the reference's phone table and a few conversions to
tests/golden/text_eng.json (data, no source).  Usage: python oracle/gen_golden_text.py"""
import json
import os
import random
import sys

sys.path.insert(0, "/root/reference")
from promptttspp.text import eng  # noqa: E402

rng = random.Random(0)
cases = []
for n in (0, 1, 7, 40):
    ph = [rng.choice(eng.phonemes) for _ in range(n)]
    text = " ".join(ph)
    cases.append({"text": text, "with": eng.text_to_sequence(text), "without": eng.text_to_sequence(text, False)})
seq = eng.text_to_sequence("HH AH0 L OW1 sil W ER1 L D")
out = {"symbols": eng.symbols, "num_vocab": eng.num_vocab(), "cases": cases, "seq": seq,
       "back": eng.sequence_to_text(seq), "back_stripped": eng.sequence_to_text(seq, True)}
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "text_eng.json")
json.dump(out, open(dst, "w"))
print("wrote", dst, len(out["symbols"]), "symbols")
