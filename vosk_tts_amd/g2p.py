"""Rule-based Russian grapheme-to-phoneme conversion for stress-marked words.

Behavioural mirror of the reference front-end `vosk_tts/g2p.py:84-107` (`convert`), written as
a single table-driven pass with one character of look-ahead.  Known answers
(reference header, g2p.py:5-11):

    абстракцион+истов -> a0 b s t r a0 k c i0 o0 nj i1 s t o0 v
    абстр+акцию       -> a0 b s t r a1 k c i0 j u0
    абстр+акция       -> a0 b s t r a1 k c i0 j a0

Rules (g2p.py:59-82):
  * '+' marks the following letter as stressed and is itself dropped;
  * paired consonants are palatalised ('j' suffix) when the next letter is one of я ё ю и ь е;
  * я ю е ё are iotated (a 'j' phoneme is inserted) at a syllable start, i.e. after the word
    boundary, ъ, ь, '-' or another vowel;
  * vowels carry the stress digit; ъ ь - and the boundary are dropped; anything else passes through.
"""

_PAIRED = {"б": "b", "в": "v", "г": "g", "Г": "g", "д": "d", "з": "z", "к": "k", "л": "l", "м": "m", "н": "n",
           "п": "p", "р": "r", "с": "s", "т": "t", "ф": "f", "х": "h"}
_UNPAIRED = {"ж": "zh", "ц": "c", "ч": "ch", "ш": "sh", "щ": "sch", "й": "j"}
_VOWEL = {"а": "a", "я": "a", "у": "u", "ю": "u", "о": "o", "ё": "o", "э": "e", "е": "e", "и": "i", "ы": "y"}
_SOFTENING = frozenset("яёюиье")
_IOTATED = frozenset("яюеё")
_SYLLABLE_START = frozenset("#ъьаяоёуюэеиы-")
_SILENT = frozenset("#+-ьъ")


def convert(stressword):
    """'прив+ет' -> 'p rj i0 vj e1 t'"""
    letters = []  # (char, stressed)
    stressed = 0
    for ch in "#" + stressword + "#":
        if ch == "+":
            stressed = 1
            continue
        letters.append((ch, stressed))
        stressed = 0
    out = []
    prev = ""  # what the previous position looked like to the iotation rule
    last = len(letters) - 1
    for i, (ch, st) in enumerate(letters):
        if prev in _SYLLABLE_START and ch in _IOTATED:
            out.append("j")
        if i < last and ch in _PAIRED:
            ph = _PAIRED[ch] + ("j" if letters[i + 1][0] in _SOFTENING else "")
            out.append(ph)
            prev = ph
        elif i < last and ch in _UNPAIRED:
            ph = _UNPAIRED[ch]
            out.append(ph)
            prev = ph
        elif ch in _VOWEL:
            out.append(_VOWEL[ch] + str(st))
            prev = ch
        else:
            if ch not in _SILENT:
                out.append(ch)
            prev = ch
    return " ".join(out)
