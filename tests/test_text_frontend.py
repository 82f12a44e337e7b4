"""Phone table / id conversion against the fixture generated from the reference (oracle/gen_golden_text.py)."""
import json
import os

import pytest

from promptttspp.text import eng as alias_eng
from promptttspp_amd.text import eng

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "text_eng.json")))


def test_symbol_table_is_the_references():
    assert eng.symbols == G["symbols"] and eng.num_vocab() == G["num_vocab"] == 90
    assert alias_eng is eng  # promptttspp.text.eng resolves to the same module
    assert [eng.symbol_to_id(s) for s in eng.symbols] == list(range(90))
    assert [eng.id_to_symbol(i) for i in range(90)] == G["symbols"]


def test_conversions():
    for c in G["cases"]:
        assert eng.text_to_sequence(c["text"]) == c["with"]
        assert eng.text_to_sequence(c["text"], add_special_token=False) == c["without"]
    assert eng.sequence_to_text(G["seq"]) == G["back"]
    assert eng.sequence_to_text(G["seq"], remove_special_token=True) == G["back_stripped"]
    with pytest.raises(KeyError):
        eng.text_to_sequence("AA9")
