"""`Synth` — mirror of vosk_tts.Synth (vosk_tts/synth.py:11-150) for the VITS flavour
(`g2p_noembed`, synth.py:223-255): same signature, defaults, feed construction, int16 conversion,
RTF log line and WAV output.  The only change is what sits behind `self.model.onnx.run`.
"""
import logging
import re
import time
import wave

import numpy as np

from .g2p import convert

_SPLIT = "([,.?!;:\"() ])"


class Synth:
    def __init__(self, model):
        self.model = model

    def audio_float_to_int16(self, audio, max_wav_value=32767.0):
        """Normalize audio and convert to int16 range (synth.py:16-23)"""
        audio_norm = np.clip(audio * max_wav_value, -max_wav_value, max_wav_value)
        return audio_norm.astype("int16")

    @staticmethod
    def normalize(text):
        """the two text fix-ups synth_audio applies before any front-end (synth.py:58-59)"""
        return re.sub("—", "-", text.strip())

    def _feed(self, text, speaker_id, noise_level, speech_rate, duration_noise_level, scale):
        """Runtime defaults and the six-key feed of synth.py:50-56,100-120."""
        inf = self.model.config.get("inference", {})
        if noise_level is None:
            noise_level = inf.get("noise_level", 0.8)
        if speech_rate is None:
            speech_rate = inf.get("speech_rate", 1.0)
        if duration_noise_level is None:
            duration_noise_level = inf.get("duration_noise_level", 0.8)
        if scale is None:
            scale = inf.get("scale", 1.0)

        text = self.normalize(text)
        model_type = self.model.config.get("model_type") or ""
        bert_embs = None
        phone_duration_extra = None
        have_bert = self.model.tokenizer is not None
        if model_type.startswith("multistream"):
            # synth.py:64-87: v3 = lower-cased text for BERT + '_' pause marks, v2 = word positions, v1 = plain phonemes;
            # v2 also runs without a tokenizer (zero BERT vectors, synth.py:77-81)
            from .multistream import g2p_multistream

            idmap = self.model.config["phoneme_id_map"]
            if model_type == "multistream_v3" and have_bert:
                bert = self.get_word_bert(text.lower(), nopunc=True)
                stream_ids, per_symbol, extra = g2p_multistream(text, self.model.dic, idmap, bert, pause_marks=True)
                phone_duration_extra = np.expand_dims(np.array(extra, dtype=np.float32), 0)
            elif model_type in ("multistream_v1", "multistream_v2") and have_bert:
                bert = self.get_word_bert(text, nopunc=True)
                stream_ids, per_symbol = g2p_multistream(text, self.model.dic, idmap, bert, word_pos=model_type == "multistream_v2")
            elif model_type == "multistream_v2":
                stream_ids, _ = g2p_multistream(text, self.model.dic, idmap, None, word_pos=True)
                per_symbol = None
            else:
                # without a tokenizer the reference falls through to g2p_noembed for v1/v3 (synth.py:100-103), which a
                # five-stream graph cannot take
                raise NotImplementedError(f"{model_type} needs bert/ (vocab.txt + model.bertw) next to the model")
            ids = np.expand_dims(np.transpose(np.array(stream_ids, dtype=np.int64)), 0)  # [1, 5, T]
            if per_symbol is None:
                bert_embs = np.zeros((1, 768, ids.shape[2]), dtype=np.float32)
            else:
                bert_embs = np.expand_dims(np.transpose(np.array(per_symbol, dtype=np.float32)), 0)  # [1, 768, T]
            lengths = np.array([ids.shape[2]], dtype=np.int64)
        elif have_bert:
            # BERT-conditioned VITS flavours (synth.py:88-99): per-word vectors fanned out to the phonemes of each word,
            # with (g2p) or without (g2p_noblank, config "no_blank") the interspersed blank
            bert = self.get_word_bert(text)
            fe = self.g2p_noblank if self.model.config.get("no_blank", 0) != 0 else self.g2p
            phoneme_ids, emb = fe(text, bert)
            bert_embs = np.expand_dims(np.transpose(np.array(emb, dtype=np.float32)), 0)
            ids = np.expand_dims(np.array(phoneme_ids, dtype=np.int64), 0)
            lengths = np.array([ids.shape[1]], dtype=np.int64)
        else:
            phoneme_ids = self.g2p_noembed(text)
            ids = np.expand_dims(np.array(phoneme_ids, dtype=np.int64), 0)
            lengths = np.array([ids.shape[1]], dtype=np.int64)
        scales = np.array([noise_level, 1.0 / speech_rate, duration_noise_level], dtype=np.float32)
        if speaker_id is None:
            speaker_id = 0
        sid = np.array([speaker_id], dtype=np.int64)
        args = {"input": ids, "input_lengths": lengths, "scales": scales, "sid": sid, "bert": bert_embs,
                "phone_duration_extra": phone_duration_extra}
        return args, scale

    def get_word_bert(self, text, nopunc=False):
        """Word-level BERT vectors (synth.py:25-44): encode the text without stress marks, run the encoder, keep the
        rows of first word pieces (optionally dropping punctuation tokens).  -> float32 [n_words + 2, 768]"""
        from .multistream import word_bert_rows

        tokens = self.model.tokenizer.encode(text.replace("+", "").replace("_", ""))
        bert = self.model.bert_onnx.run(None, {"input_ids": [tokens.ids], "attention_mask": [tokens.attention_mask],
                                               "token_type_ids": [tokens.type_ids]})[0]
        return bert[word_bert_rows(tokens.tokens, nopunc)]

    def synth_audio(self, text, speaker_id=0, noise_level=None, speech_rate=None, duration_noise_level=None, scale=None):
        args, scale = self._feed(text, speaker_id, noise_level, speech_rate, duration_noise_level, scale)

        start_time = time.perf_counter()
        run_pcm16 = getattr(self.model.onnx, "run_pcm16", None)
        if run_pcm16 is not None:
            # same three steps as below (squeeze, * scale, audio_float_to_int16) fused behind the boundary: the device
            # converts and only int16 crosses PCIe (vits_synthesize_pcm16)
            audio = run_pcm16(args, scale).squeeze()
        else:
            audio = self.model.onnx.run(None, args)[0]
            audio = audio.squeeze()
            audio = audio * scale
            audio = self.audio_float_to_int16(audio)
        end_time = time.perf_counter()

        audio_duration_sec = audio.shape[-1] / 22050
        infer_sec = end_time - start_time
        real_time_factor = infer_sec / audio_duration_sec if audio_duration_sec > 0 else 0.0
        logging.info("Real-time factor: %0.2f (infer=%0.2f sec, audio=%0.2f sec)" % (real_time_factor, infer_sec, audio_duration_sec))
        return audio

    def synth_stream(self, text, speaker_id=0, noise_level=None, speech_rate=None, duration_noise_level=None, scale=None,
                     chunk_frames=64):
        """Generator of int16 PCM chunks (chunk_frames*256 samples each, ~0.74 s at the default): what a streaming
        `SynthesizeStream` handler would put into successive AudioChunk messages (tts_service.proto:46-54) instead
        of the single whole-utterance chunk of tts_server.py:54.  Same conversion as synth_audio per chunk."""
        args, scale = self._feed(text, speaker_id, noise_level, speech_rate, duration_noise_level, scale)
        if not hasattr(self.model.onnx, "run_stream"):
            raise NotImplementedError("this session type has no run_stream (VitsSession: vits_stream_open, SttsSession: stts_stream_open)")
        for chunk in self.model.onnx.run_stream(None, args, chunk_frames=chunk_frames):
            yield self.audio_float_to_int16(chunk * scale)

    def synth(self, text, oname, speaker_id=0, noise_level=None, speech_rate=None, duration_noise_level=None, scale=None):
        audio = self.synth_audio(text, speaker_id, noise_level, speech_rate, duration_noise_level, scale)
        with wave.open(oname, "w") as f:
            f.setnchannels(1)
            f.setsampwidth(2)
            f.setframerate(22050)
            f.writeframes(audio.tobytes())

    def _phonemes_and_words(self, text):
        """One pass over the split text: the phoneme string ('^' ... '$', punctuation kept, dictionary or rule G2P per word) and,
        per phoneme, the index of the token it belongs to in get_word_bert's row order: 0 = '^' ([CLS]), tokens counted from 1,
        spaces emit a phoneme but do not advance the count, -1 = '$' ([SEP])  (synth.py:152-171 / 190-209 / 224-236)."""
        phonemes, words = ["^"], [0]
        w = 1
        for token in re.split(_SPLIT, text.lower()):
            if token == "":
                continue
            if re.match(_SPLIT, token) or token == "-":
                ps = [token]
            elif token in self.model.dic:
                ps = self.model.dic[token].split()
            else:
                ps = convert(token).split()
            phonemes.extend(ps)
            words.extend([w] * len(ps))
            if token != " ":
                w += 1
        phonemes.append("$")
        words.append(-1)
        return phonemes, words

    def phonemize(self, text):
        """Words -> dictionary lookup or rule G2P, punctuation kept, '^' ... '$' (synth.py:224-236)."""
        return self._phonemes_and_words(text)[0]

    def g2p_noblank(self, text, embeddings):
        """BERT-conditioned front-end without blanks (synth.py:190-220): ids of the phoneme string and, per phoneme, the BERT row
        of its word (embeddings[0] for '^', embeddings[-1] for '$')."""
        phonemes, words = self._phonemes_and_words(text)
        id_map = self.model.config["phoneme_id_map"]
        logging.info(f"Text: {text}")
        logging.info(f"Phonemes: {phonemes}")
        return [id_map[p] for p in phonemes], [embeddings[w] for w in words]

    def g2p(self, text, embeddings):
        """BERT-conditioned front-end with the interspersed blank 0 (synth.py:152-188): every blank carries the BERT row of the
        phoneme that FOLLOWS it."""
        ids, emb = self.g2p_noblank(text, embeddings)
        out_ids, out_emb = [ids[0]], [emb[0]]
        for i, e in zip(ids[1:], emb[1:]):
            out_ids += [0, i]
            out_emb += [e, e]
        return out_ids, out_emb

    def g2p_noembed(self, text):
        phonemes = self.phonemize(text)
        # ids interspersed with blank 0; id-map values may be lists (synth.py:238-251)
        id_map = self.model.config["phoneme_id_map"]
        as_list = (lambda v: list(v)) if isinstance(id_map[phonemes[0]], list) else (lambda v: [v])
        phoneme_ids = as_list(id_map[phonemes[0]])
        for p in phonemes[1:]:
            phoneme_ids.append(0)
            phoneme_ids.extend(as_list(id_map[p]))
        logging.info(f"Text: {text}")
        logging.info(f"Phonemes: {phonemes}")
        return phoneme_ids
