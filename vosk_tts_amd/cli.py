#!/usr/bin/env python3
"""`python -m vosk_tts_amd.cli` — same flags as the reference console script
(vosk_tts/cli.py:12-65) on top of the MI355X engine."""
import argparse
import logging
import sys

from .model import Model, list_languages, list_models
from .synth import Synth

parser = argparse.ArgumentParser(description="Synthesize input")
parser.add_argument("--model", "-m", type=str, help="model path")
parser.add_argument("--list-models", default=False, action="store_true", help="list available models")
parser.add_argument("--list-languages", default=False, action="store_true", help="list available languages")
parser.add_argument("--model-name", "-n", type=str, help="select model by name")
parser.add_argument("--lang", "-l", default="en-us", type=str, help="select model by language")
parser.add_argument("--input", "-i", type=str, help="input string")
parser.add_argument("--speaker", "-s", type=int, help="speaker id for multispeaker model")
parser.add_argument("--speech-rate", "-r", type=float, default=1.0, help="speech rate of the synthesis")
parser.add_argument("--output", "-o", default="out.wav", type=str, help="optional output filename path")
parser.add_argument("--log-level", default="INFO", help="logging level")


def main(argv=None):
    args = parser.parse_args(argv)
    logging.getLogger().setLevel(args.log_level.upper())
    if args.list_models:
        list_models()
        return
    if args.list_languages:
        list_languages()
        return
    if not args.input:
        logging.info("Please specify input text or file")
        sys.exit(1)
    model = Model(args.model, args.model_name, args.lang)
    Synth(model).synth(args.input, args.output, args.speaker, speech_rate=args.speech_rate)


if __name__ == "__main__":
    main()
