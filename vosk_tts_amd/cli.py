#!/usr/bin/env python3
"""Console entry point with the flag surface of the reference `vosk-tts` script (vosk_tts/cli.py:12-43):
-m/--model, -n/--model-name, -l/--lang, -i/--input, -s/--speaker, -r/--speech-rate, -o/--output,
--list-models, --list-languages, --log-level.  Behaviour mirrored from cli.py:45-65: listing flags
short-circuit, a missing --input logs a hint and exits with status 1, otherwise one utterance is
synthesised to a 22.05 kHz mono 16-bit WAV through the MI355X engine.

    python -m vosk_tts_amd.cli -m /path/to/model-dir -i "прив+ет м+ир" -s 2 -o out.wav
"""
import argparse
import logging
import sys

# (flags, argparse keyword arguments) — one row per option of the reference CLI
_OPTIONS = (
    (("-m", "--model"), dict(type=str, metavar="DIR", help="directory holding model.vitsw, dictionary and config.json")),
    (("-n", "--model-name"), dict(type=str, metavar="NAME", help="pick a locally installed model by its directory name")),
    (("-l", "--lang"), dict(type=str, default="en-us", metavar="LANG", help="pick a locally installed model by language code")),
    (("-i", "--input"), dict(type=str, metavar="TEXT", help="text to speak ('+' marks the stressed vowel)")),
    (("-s", "--speaker"), dict(type=int, metavar="ID", help="speaker index of a multi-speaker voice")),
    (("-r", "--speech-rate"), dict(type=float, default=1.0, metavar="X", help="tempo multiplier (>1 is faster)")),
    (("-o", "--output"), dict(type=str, default="out.wav", metavar="WAV", help="where to write the audio")),
    (("--list-models",), dict(action="store_true", help="print the models found in the local search path and exit")),
    (("--list-languages",), dict(action="store_true", help="print the language codes of the local models and exit")),
    (("--log-level",), dict(default="INFO", metavar="LEVEL", help="python logging level (INFO shows the RTF line)")),
)


def build_parser():
    ap = argparse.ArgumentParser(prog="vosk-tts-amd", description="Text to speech on the MI355X-native VITS2 engine")
    for flags, kw in _OPTIONS:
        ap.add_argument(*flags, **kw)
    return ap


def run(opts):
    from . import model as model_mod

    logging.getLogger().setLevel(opts.log_level.upper())
    for wanted, action in ((opts.list_models, model_mod.list_models), (opts.list_languages, model_mod.list_languages)):
        if wanted:
            action()
            return 0
    if not opts.input:
        logging.info("Please specify input text or file")
        return 1
    from .synth import Synth

    voice = model_mod.Model(opts.model, opts.model_name, opts.lang)
    Synth(voice).synth(opts.input, opts.output, opts.speaker, speech_rate=opts.speech_rate)
    return 0


def main(argv=None):
    status = run(build_parser().parse_args(argv))
    if status:
        sys.exit(status)


if __name__ == "__main__":
    main()
