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
