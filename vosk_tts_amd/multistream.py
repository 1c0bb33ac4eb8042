"""Front-end of the `multistream_v*` model flavours: text -> five id streams per symbol
(mirror of vosk_tts.Synth.g2p_multistream / add_pos, vosk_tts/synth.py:258-347).

Per symbol the graph receives (text_encoder.py:113-127 of the StableTTS tree):
  0 the phoneme (with word-position suffix _B/_I/_E/_S for multistream_v2), or ' ', '^', '$'
  1 the punctuation attached to the word boundary that follows, '_' when none
  2 1 inside double quotes, else 0 (used directly as an id)
  3 the most recent punctuation seen when reading the utterance BACKWARDS from the end
  4 the most recent sentence-level punctuation ('...', '.', '!', '?', '-') seen backwards
The optional per-word BERT vectors are fanned out to the symbols of each word (index 0 = before the first word).
"""
import re

from .g2p import convert

_TOKENS = re.compile(r"(\.\.\.|- |[ ,.?!;:\"()])")
_SENTENCE_MARKS = ("...", ".", "!", "?", "-")  # priority order of the reference's elif chain (synth.py:331-340)


def word_positions(phones):
    """B/I/E/S suffixes (synth.py:258-270)"""
    if len(phones) == 1:
        return [phones[0] + "_S"]
    return [p + ("_B" if i == 0 else "_E" if i == len(phones) - 1 else "_I") for i, p in enumerate(phones)]


def _symbols(text, dic, word_pos):
    """[(symbol, punctuation list, in_quote, word index)] in reading order, '^' first, ' ' + '$' last"""
    out = [("^", [], 0, 0)]
    quote, pending, widx = 0, [], 1
    for tok in _TOKENS.split(text.replace(" -", "- ").lower()):
        if tok == "":
            continue
        if tok == '"':
            quote ^= 1
        elif tok in ("- ", "-"):
            pending.append("-")
        elif tok == " ":
            out.append((" ", pending, quote, widx))
            pending = []
        elif _TOKENS.fullmatch(tok):
            pending.append(tok)
        else:
            phones = (dic[tok] if tok in dic else convert(tok)).split()
            if word_pos:
                phones = word_positions(phones)
            out.extend((p, [], quote, widx) for p in phones)
            pending = []
            widx += 1
    out.append((" ", pending, quote, widx))
    out.append(("$", [], 0, widx))
    return out


def g2p_multistream(text, dic, phoneme_id_map, bert_embeddings=None, word_pos=False):
    """-> (ids: list of 5-tuples, one per symbol; bert: list of per-symbol vectors or [])"""
    syms = _symbols(text, dic, word_pos)
    last, last_sentence = " ", " "
    ids, bert = [None] * len(syms), []
    for k in range(len(syms) - 1, -1, -1):  # the two "last ..." streams accumulate from the END of the utterance
        sym, puncs, quote, widx = syms[k]
        for mark in _SENTENCE_MARKS:
            if mark in puncs:
                last_sentence = mark
                break
        here = puncs[0] if puncs else "_"
        if puncs:
            last = puncs[0]
        ids[k] = (phoneme_id_map[sym], phoneme_id_map[here], quote, phoneme_id_map[last], phoneme_id_map[last_sentence])
    if bert_embeddings is not None:
        bert = [bert_embeddings[widx] for (_, _, _, widx) in syms]
    return ids, bert
