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
_TOKENS_V3 = re.compile(r"(\.\.\.|- |[ ,.?!;:\"()_])")  # multistream_v3: '_' marks a forced pause (synth.py:363)
PAUSE_FRAMES = 20.0  # phone_duration_extra of a '_' mark (synth.py:432-435)
_SENTENCE_MARKS = ("...", ".", "!", "?", "-")  # priority order of the reference's elif chain (synth.py:331-340)


def word_positions(phones):
    """B/I/E/S suffixes (synth.py:258-270)"""
    if len(phones) == 1:
        return [phones[0] + "_S"]
    return [p + ("_B" if i == 0 else "_E" if i == len(phones) - 1 else "_I") for i, p in enumerate(phones)]


def _symbols(text, dic, word_pos, tokens=_TOKENS):
    """[(symbol, punctuation list, in_quote, word index)] in reading order, '^' first, ' ' + '$' last"""
    out = [("^", [], 0, 0)]
    quote, pending, widx = 0, [], 1
    for tok in tokens.split(text.replace(" -", "- ").lower()):
        if tok == "":
            continue
        if tok == '"':
            quote ^= 1
        elif tok in ("- ", "-"):
            pending.append("-")
        elif tok == " ":
            out.append((" ", pending, quote, widx))
            pending = []
        elif tokens.fullmatch(tok):
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


def g2p_multistream(text, dic, phoneme_id_map, bert_embeddings=None, word_pos=False, pause_marks=False):
    """-> (ids: list of 5-tuples, one per symbol; bert: list of per-symbol vectors or [])
    pause_marks=True is g2p_multistream_scales (multistream_v3, synth.py:360-456): '_' is a punctuation token, word
    positions are always on, and a third list carries 20.0 for symbols whose boundary holds a '_' (else 0.0)."""
    syms = _symbols(text, dic, word_pos or pause_marks, _TOKENS_V3 if pause_marks else _TOKENS)
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
    if pause_marks:
        return ids, bert, [PAUSE_FRAMES if "_" in puncs else 0.0 for (_, puncs, _, _) in syms]
    return ids, bert


_BERT_PUNCT = re.compile(r"[-,.?!;:\"]")


def word_bert_rows(tokens, nopunc=False):
    """Rows of the BERT output that stand for words (synth.py:36-42): the first word piece of every token ('#'-prefixed
    continuation pieces are dropped), optionally without punctuation tokens.  [CLS] and [SEP] are kept, as in the
    reference, so row 0 belongs to '^' and the last row to the closing ' ' / '$'."""
    return [i for i, t in enumerate(tokens) if t[0] != "#" and not (nopunc and _BERT_PUNCT.match(t))]
