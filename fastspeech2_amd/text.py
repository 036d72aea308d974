"""Symbol inventory and phoneme-string -> id conversion: the data format on the INPUT side of the hot path.

The embedding table `encoder.src_word_emb` (V = 361 rows) is indexed by these ids, so the inventory and its ORDER
are part of the checkpoint format (reference text/symbols.py:8-29: pad, special, punctuation, letters, "@"-prefixed
ARPAbet, "@"-prefixed pinyin, silences).  `tests/golden/symbols.json` holds the reference's own list (dumped by
tests/golden/make_golden.py) and `tests/test_data_cpu.py` pins this construction against it.

Only what `dataset.py` / `synthesize.py --mode batch` need is here: metadata lines carry phoneme strings in curly
braces (`{DH AH0 sp ...}`), which map 1:1 to ids (reference text/__init__.py:16-41,64-75).  Grapheme-to-phoneme
(g2p_en / pypinyin / lexicon lookup, synthesize.py:20-84) is host string work outside the hot path;
`text_to_sequence` still accepts free text outside braces and maps it character-wise after `basic_cleaners`
(lower-case + whitespace collapse), which is what the reference does for characters once cleaned.
"""
import re

_PAD = "_"
_SPECIAL = "-"
_PUNCTUATION = "!'(),.:;? "
_LETTERS = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz"
_SILENCES = ["@sp", "@spn", "@sil"]


def _arpabet():
    """The 84 CMUdict phones: 15 vowels x {no stress, 0, 1, 2} + 24 consonants, in lexicographic order."""
    vowels = "AA AE AH AO AW AY EH ER EY IH IY OW OY UH UW".split()
    consonants = "B CH D DH F G HH JH K L M N NG P R S SH T TH V W Y Z ZH".split()
    return sorted([v + s for v in vowels for s in ("", "0", "1", "2")] + consonants)


def _pinyin():
    """23 initials, 37 finals x 5 tones, and the erhua marker."""
    initials = "b c ch d f g h j k l m n p q r s sh t w x y z zh".split()
    finals = ("a ai an ang ao e ei en eng er i ia ian iang iao ie ii iii in ing iong iou o ong ou u ua uai uan uang "
              "uei uen uo v van ve vn").split()
    return initials + [f + str(t) for f in finals for t in range(1, 6)] + ["rr"]


symbols = ([_PAD] + list(_SPECIAL) + list(_PUNCTUATION) + list(_LETTERS) + ["@" + s for s in _arpabet()]
           + ["@" + s for s in _pinyin()] + _SILENCES)
_symbol_to_id = {s: i for i, s in enumerate(symbols)}
_id_to_symbol = dict(enumerate(symbols))

_BRACES = re.compile(r"(.*?)\{(.+?)\}(.*)")
_WS = re.compile(r"\s+")


def basic_cleaners(text):
    return _WS.sub(" ", text.lower())


_CLEANERS = {"basic_cleaners": basic_cleaners, "transliteration_cleaners": basic_cleaners}


def _clean(text, cleaner_names):
    for name in cleaner_names:
        if name == "english_cleaners":
            # number / abbreviation expansion needs `inflect` + `unidecode` (absent here); inside the training and
            # batch-synthesis path all text is already phonemes in braces, so only the plain normalisation applies.
            # Free text OUTSIDE braces that the reference's english_cleaners would rewrite (digits, "$", "Mr." ...) would map to
            # different ids here: say so instead of degrading silently.
            if re.search(r"[0-9$£]|\b(mrs?|dr|st|co|jr|maj|gen|drs|rev|lt|hon|sgt|capt|esq|ltd|col|ft)\.", text, re.I) or not text.isascii():
                import warnings
                warnings.warn("english_cleaners: number / abbreviation / unicode expansion is not built (needs inflect + unidecode); "
                              f"text outside {{}} is only lower-cased: {text[:40]!r}", stacklevel=3)
            text = basic_cleaners(text)
        elif name in _CLEANERS:
            text = _CLEANERS[name](text)
        else:
            raise Exception("Unknown cleaner: %s" % name)
    return text


def _keep(s):
    return s in _symbol_to_id and s != "_" and s != "~"


def _to_ids(syms):
    return [_symbol_to_id[s] for s in syms if _keep(s)]


def text_to_sequence(text, cleaner_names=()):
    """`"{HH AW1 S sp T AH0 N}"` -> [ids].  Unknown symbols are dropped, as the reference does."""
    seq = []
    while text:
        m = _BRACES.match(text)
        if not m:
            seq += _to_ids(_clean(text, cleaner_names))
            break
        seq += _to_ids(_clean(m.group(1), cleaner_names))
        seq += _to_ids(["@" + s for s in m.group(2).split()])
        text = m.group(3)
    return seq


def sequence_to_text(sequence):
    out = ""
    for i in sequence:
        s = _id_to_symbol.get(int(i))
        if s is None:
            continue
        out += "{%s}" % s[1:] if (len(s) > 1 and s[0] == "@") else s
    return out.replace("}{", " ")
