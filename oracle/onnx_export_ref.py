"""Run the reference's own ONNX export procedures in THIS container (TEST INFRASTRUCTURE, container-only).

`torch.onnx.export` (TorchScript exporter) needs the `onnx` package only in its last step, to splice onnxscript
functions into the serialized model; the reference graphs have none, so that step is bypassed.  Everything else is
the real exporter: initializer naming, de-duplication, constant folding - exactly what vosk_tts_amd/onnx_import.py must
undo.  Used by tests/test_onnx_import.py (skipped where /root/reference is absent).
"""
import io
import warnings


def _exporter():
    import torch
    from torch.onnx._internal.torchscript_exporter import onnx_proto_utils as opu

    opu._add_onnxscript_fn = lambda model_bytes, custom_opsets: model_bytes
    return torch


def export_vits(net, n_symbols=50, opset=15):
    """training/vits2/onnx_export.py:60-110: infer_forward as the module's forward, dummy inputs of 50 symbols,
    input / input_lengths / scales / sid -> output, dynamic batch and phoneme axes."""
    torch = _exporter()

    def infer_forward(text, text_lengths, scales, sid=None):
        return net.infer(text, text_lengths, noise_scale=scales[0], length_scale=scales[1], noise_scale_w=scales[2], sid=sid)[0].unsqueeze(1)

    net.forward = infer_forward
    text = torch.randint(low=0, high=net.n_vocab, size=(1, n_symbols), dtype=torch.long)
    args = (text, torch.LongTensor([n_symbols]), torch.FloatTensor([0.667, 1.0, 0.8]), torch.LongTensor([0]) if net.n_speakers > 0 else None)
    f = io.BytesIO()
    with warnings.catch_warnings(), torch.no_grad():
        warnings.simplefilter("ignore")
        torch.onnx.export(model=net, args=args, f=f, verbose=False, opset_version=opset, input_names=["input", "input_lengths", "scales", "sid"],
                          output_names=["output"], dynamo=False,
                          dynamic_axes={"input": {0: "batch_size", 1: "phonemes"}, "input_lengths": {0: "batch_size"}, "output": {0: "batch_size", 1: "time"}})
    return f.getvalue()


def export_stts(matcha, vocoder, n_timesteps=5, opset=17, n_symbols=50):
    """training/stabletts/matcha/onnx/export.py:21-51,64-98,150-186: MatchaWithVocoder.forward = synthesise -> vocoder.decode(mel)
    .clamp(-1, 1); inputs input [1,5,T] / input_lengths / scales / sid (multi-speaker) / bert [1,768,T] / phone_duration_extra."""
    torch = _exporter()

    def onnx_forward_func(x, x_lengths, scales, spks=None, bert=None, phone_duration_extra=None):
        out = matcha.synthesise(x, x_lengths, n_timesteps, scales[0], scales[2], spks, bert, scales[1], phone_duration_extra)
        return out["mel"], out["mel_lengths"]

    matcha.forward = onnx_forward_func

    class WithVocoder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.matcha = matcha
            self.vocoder = vocoder

        def forward(self, x, x_lengths, scales, spks=None, bert=None, phone_duration_extra=None):
            mel, mel_lengths = self.matcha(x, x_lengths, scales, spks, bert, phone_duration_extra)
            return self.vocoder.decode(mel).clamp(-1, 1).squeeze(1), mel_lengths * 256

    multi = matcha.n_spks > 1
    inputs = [torch.randint(low=0, high=20, size=(1, 5, n_symbols), dtype=torch.long), torch.LongTensor([n_symbols]), torch.Tensor([0.8, 0.8, 1.0])]
    names = ["input", "input_lengths", "scales"]
    if multi:
        inputs.append(torch.LongTensor([1])); names.append("sid")
    inputs += [torch.rand(1, 768, n_symbols), torch.rand(1, n_symbols)]
    names += ["bert", "phone_duration_extra"]
    axes = {"input": {0: "batch_size", 2: "time"}, "input_lengths": {0: "batch_size"}, "bert": {0: "batch_size", 2: "time"},
            "phone_duration_extra": {0: "batch_size", 1: "time"}, "wav": {0: "batch_size", 1: "time"}, "wav_lengths": {0: "batch_size"}}
    if multi:
        axes["sid"] = {0: "batch_size"}
    f = io.BytesIO()
    with warnings.catch_warnings(), torch.no_grad():
        warnings.simplefilter("ignore")
        torch.onnx.export(WithVocoder().eval(), tuple(inputs), f, input_names=names, output_names=["wav", "wav_lengths"], dynamic_axes=axes,
                          opset_version=opset, export_params=True, do_constant_folding=True, dynamo=False)
    return f.getvalue()


def export_bert(model, opset=17):
    """training/stabletts/matcha/onnx/bert-export.py:5-33: BertModel subclass returning hidden_states[-3] squeezed, inputs
    input_ids / attention_mask / token_type_ids with dynamic batch and sequence axes."""
    torch = _exporter()

    class OurBert(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.bert = model

        def forward(self, input_ids, attention_mask, token_type_ids):
            out = self.bert(input_ids=input_ids, attention_mask=attention_mask, token_type_ids=token_type_ids, output_hidden_states=True)
            return torch.cat(out["hidden_states"][-3:-2], -1).squeeze(0)

    ids = torch.tensor([[2, 17, 9, 31, 5, 3]])
    f = io.BytesIO()
    dyn = {0: "batch_size", 1: "sequence"}
    with warnings.catch_warnings(), torch.no_grad():
        warnings.simplefilter("ignore")
        torch.onnx.export(OurBert().eval(), (ids, torch.ones_like(ids), torch.zeros_like(ids)), f, input_names=["input_ids", "attention_mask", "token_type_ids"],
                          output_names=["logits"], dynamic_axes={"input_ids": dyn, "attention_mask": dyn, "token_type_ids": dyn, "logits": dyn},
                          do_constant_folding=True, opset_version=opset, dynamo=False)
    return f.getvalue()


def transformers_bert():
    """(BertConfig, BertModel).  transformers probes optional packages with importlib.util.find_spec when it is first imported;
    the sys.modules stand-ins that refimport.py / refimport_stts.py install for librosa / torchaudio / ... (no __spec__, not
    real packages) must not be visible to that probe."""
    import sys

    hidden = {n: sys.modules.pop(n) for n in list(sys.modules)
              if n.split(".")[0] in ("librosa", "torchaudio", "onnxruntime", "lightning", "torchdiffeq") and getattr(sys.modules[n], "__file__", None) is None}
    try:
        from transformers import BertConfig, BertModel
    finally:
        sys.modules.update(hidden)
    return BertConfig, BertModel
