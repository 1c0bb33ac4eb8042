"""Compile-time invariants of the hot kernels, read from the gfx950 ISA hipcc emits (no GPU needed): register allocation / occupancy,
no scratch (spills) where the design assumes none, and no waterfall loops around the buffer accesses of the 128x128 kernel.
A change that silently costs a workgroup per CU or spills in a tap loop shows up here, not three rounds later in a profile."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if not os.path.exists(HIPCC) or not shutil.which("c++filt"):
        pytest.skip("hipcc / c++filt not available")
    out = tmp_path_factory.mktemp("isa") / "engine.s"
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-w", "-S", "-o", str(out), "engine.hip"],
                          cwd=os.path.join(ROOT, "vosk_tts_amd", "csrc"))
    text = out.read_text()
    mangled = re.findall(r"^(_Z[0-9A-Za-z_]+):\s+; @", text, flags=re.M)
    names = subprocess.run(["c++filt"] + mangled, capture_output=True, text=True, check=True).stdout.splitlines()
    kernels = {}
    for m, d in zip(mangled, names):
        i = text.index("\n" + m + ":")
        j = text.index(".Lfunc_end", i)
        meta = text[j:j + 6000]
        kernels[d.split("(")[0].replace("void ", "")] = dict(
            body=text[i:j], vgprs=int(re.search(r"; NumVgprs: (\d+)", meta).group(1)), scratch=int(re.search(r"; ScratchSize: (\d+)", meta).group(1)),
            occupancy=int(re.search(r"; Occupancy: (\d+)", meta).group(1)))
    return kernels


def test_big_tile_kernels_keep_three_workgroups_per_cu_without_spills(isa):
    for name in ("conv_mfma_kernel<2, 2, 2, 2, 0>", "conv_bf3_kernel<2, 0, 2>", "conv_bf3_kernel<1, 0, 2>"):
        k = isa[name]
        assert k["scratch"] == 0 and k["occupancy"] >= 3, (name, k["vgprs"], k["scratch"], k["occupancy"])
    # 256-thread workgroups, 3 per CU = 3 waves per SIMD: at most 512 / 3 registers
    assert isa["conv_mfma_kernel<2, 2, 2, 2, 0>"]["vgprs"] <= 168


def test_big_tile_kernel_streams_on_buffer_addressing_without_waterfall_loops(isa):
    body = isa["conv_mfma_kernel<2, 2, 2, 2, 0>"]["body"]
    assert body.count("buffer_load_dwordx4") >= 8 and body.count("buffer_store_dword") >= 64
    # a descriptor or scalar offset the compiler believes divergent turns every access into a readfirstlane / compare / branch loop
    assert "s_and_saveexec_b64 vcc, vcc" not in body
    # the tap loops: fragment loads are buffer loads (no per-load 64-bit VALU address), 32 MFMAs per tap
    taps = [b for b in body.split("s_cbranch") if b.count("v_mfma_f32_32x32x2_f32") >= 32]
    assert taps and all("global_load_dwordx4" not in b for b in taps)


def test_single_utterance_kernels_register_budgets(isa):
    wp = isa["conv_wp_kernel<8>"]
    assert wp["vgprs"] <= 128 and wp["occupancy"] >= 4 and wp["scratch"] <= 32  # two 8-wave workgroups per CU
    for name in ("conv16_kernel<0, 4, 8, 0>", "conv16_kernel<1, 4, 20, 0>", "conv16_kernel<0, 8, 8, 1>", "relpos_attention16_kernel<96, 4>"):
        assert isa[name]["scratch"] == 0, (name, isa[name])


def test_persistent_kernel_keeps_its_operands_out_of_loop_carried_registers(isa):
    """persist_kernel (csrc/persist.hip.h): one 512-thread workgroup per CU, so 256 registers are available -- but the step loop must
    not carry prefetched operands in registers across its back edge (round 3: that cost an s_waitcnt vmcnt(0) + 40 moves per step
    and 85 VGPRs).  Budget with the operands requested at the top of the step: no scratch, well under the file."""
    k = isa["persist_kernel"]
    assert k["scratch"] == 0 and k["vgprs"] <= 216, (k["vgprs"], k["scratch"])


def test_software_pipelined_conv_kernel_loop_shape(isa):
    """conv_sp_kernel (csrc/conv_sp.hip.h): what makes it worth having at one to three workgroups per CU is the SHAPE of its tap loop --
    three workgroups per CU without spills, B fragments as two ds_read_b128 per tap (the [column][channel] LDS layout; ds_read_b32
    would mean the layout regressed to eight reads + eight address adds per tap), weight fragments on buffer addressing, and no
    s_waitcnt vmcnt(0) inside the steady tap loop (a drained queue there means the 4-slot weight ring is being rotated through moves)."""
    for name in ("conv_sp_kernel<0, 1>", "conv_sp_kernel<0, 2>", "conv_sp_kernel<2, 2>", "conv_sp_kernel<3, 2>"):
        k = isa[name]
        assert k["scratch"] == 0 and k["occupancy"] >= 3 and k["vgprs"] <= 168, (name, k["vgprs"], k["scratch"], k["occupancy"])
    body = isa["conv_sp_kernel<0, 2>"]["body"]
    blocks = re.split(r"\n\.LBB\d+_\d+:", body)
    loops = [b for b in blocks if b.count("v_mfma_f32_32x32x2_f32") >= 24]  # the steady group-of-four-taps loop (and the peeled group)
    assert loops, "no block with >= 24 MFMAs: the tap groups are no longer unrolled by four"
    for b in loops:
        n_mfma = b.count("v_mfma_f32_32x32x2_f32")
        assert b.count("ds_read_b128") * 4 >= n_mfma // 2 and "ds_read_b32" not in b, "B fragments must be two ds_read_b128 per tap"
        assert "global_load_dwordx4" not in b and b.count("buffer_load_dwordx4") >= n_mfma // 8 * 2 - 2
        assert not re.search(r"s_waitcnt vmcnt\(0\)", b), "the tap loop drains the memory queue"
