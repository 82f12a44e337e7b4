"""MFA English phone inventory and phone <-> id conversion (reference: promptttspp/text/eng.py:6-156).

The id of a phone is part of the checkpoint contract (``phoneme_emb.emb.weight`` has one row per
symbol, 90 rows), so the table is reproduced exactly: ``_ ^ $`` then the ARPAbet inventory in ASCII
order -- every vowel bare and with the stress digits 0/1/2, the consonants bare -- then ``spn sil sp``.
``tests/test_text_frontend.py`` pins it against a fixture generated from the reference."""

PAD, BOS, EOS = "_", "^", "$"

_VOWELS = ("AA", "AE", "AH", "AO", "AW", "AY", "EH", "ER", "EY", "IH", "IY", "OW", "OY", "UH", "UW")
_CONSONANTS = ("B", "CH", "D", "DH", "F", "G", "HH", "JH", "K", "L", "M", "N", "NG", "P", "R", "S", "SH", "T", "TH", "V",
               "W", "Y", "Z", "ZH")
phonemes = sorted([v + s for v in _VOWELS for s in ("", "0", "1", "2")] + list(_CONSONANTS)) + ["spn", "sil", "sp"]
symbols = [PAD, BOS, EOS] + phonemes
symbol2id = {s: i for i, s in enumerate(symbols)}


def symbol_to_id(symbol):
    return symbol2id[symbol]  # KeyError on an unknown phone, like the reference


def id_to_symbol(idnum):
    return symbols[idnum]


def num_vocab():
    return len(symbols)


def text_to_sequence(text, add_special_token=True):
    """Whitespace-separated phones -> ids, wrapped in BOS / EOS unless ``add_special_token`` is False."""
    ids = [symbol2id[p] for p in text.split()]
    return [symbol2id[BOS]] + ids + [symbol2id[EOS]] if add_special_token else ids


def sequence_to_text(seq, remove_special_token=False):
    """ids -> list of phones; ``remove_special_token`` drops the first and last id."""
    seq = list(seq)[1:-1] if remove_special_token else seq
    return [symbols[int(i)] for i in seq]
