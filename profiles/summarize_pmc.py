#!/usr/bin/env python3
"""Per-kernel, per-launch averages of the rocprofv3 --pmc passes written by tools/pmc_passes.sh.
usage: summarize_pmc.py <dir with *_counter_collection.csv> <workload>  > profiles/<round>_pmc_<workload>.json

Derived figures (MI355X_MICROARCH.md, "HBM" and "rocprofv3 PMC slots"):
  hbm_bytes_per_launch = FETCH_SIZE*1024*f_read + WRITE_SIZE*1024*f_write   (FETCH_SIZE/WRITE_SIZE are in KB).
                         The guide calibrates only the 16 B/lane streaming read on gfx950 (reported = 1/2 of the
                         bytes -> f = 2) and says to calibrate other widths on a known byte count: tools/pmc_calib
                         streams 1 GiB with 4 B/lane and 16 B/lane loads and stores under the same counters, and
                         f = true bytes / reported bytes of the calib_*_dw kernels (the conv kernels move
                         activations 4 B/lane) is applied; without calibration data the guide's f_read = 2,
                         f_write = 1 are used and "calibrated" is false.
  mfma_flops_per_launch = SQ_INSTS_VALU_MFMA_MOPS_F32 * 512      executed MFMA work incl. tile padding
  avg_us                = End-Start timestamps of the dispatch in the counter run (counter runs serialize
                          dispatches; the authoritative duration is the --kernel-trace --stats run)
Kernel names are normalised to the instantiation labels bench.py prints."""
import csv, glob, json, os, re, sys

EPI = {"0": "STORE", "1": "GATE", "2": "RESSKIP", "3": "COUPLE"}


def label(name):
    m = re.match(r"void (conv_mfma_kernel)<(\d+), (\d+), (\d+), (\d+), (\d+)>", name)
    if m:
        return f"{m[1]}<{m[2]},{m[3]},{m[4]},{m[5]},{EPI[m[6]]}>"
    m = re.match(r"void (conv_mfma_ks_kernel)<(\d+), (\d+), (\d+), (\d+)>", name)
    if m:
        return f"{m[1]}<{m[2]},{m[3]},{EPI[m[4]]},{m[5]}>"
    m = re.match(r"void (conv16_kernel)<(\d+), (\d+), (\d+), (\d+)>", name)
    if m:  # <EPI, NW, MAXU, PRO> -> the engine's label: PRO 1 = DDSConv prologue, 2 = LayerNorm prologue
        if m[5] == "1":
            return f"{m[1]}<{EPI[m[2]]},dds>"
        return f"{m[1]}<{EPI[m[2]]},{'ln,' if m[5] == '2' else ''}{m[3]}>"
    m = re.match(r"void (conv_wp_kernel)<(\d+)>", name)
    if m:
        return f"{m[1]}<{m[2]}>"
    m = re.match(r"void (conv_bf3_kernel)<(\d+), (\d+)>", name)
    if m:  # <MI, EPI>: 64*MI output rows per tile
        return f"{m[1]}<{m[2]},{EPI[m[3]]}>"
    m = re.match(r"(?:void )?(\w+)", name)
    return m[1] if m else name


def main():
    d, workload = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "?")
    acc = {}
    for path in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
        seen = {}
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                k = label(r["Kernel_Name"])
                if k.startswith("__amd") or k.startswith("at"):
                    continue
                a = acc.setdefault(k, {})
                c = a.setdefault(r["Counter_Name"], [0.0, 0])
                c[0] += float(r["Counter_Value"]); c[1] += 1
                key = (k, r["Dispatch_Id"])
                if key not in seen:
                    seen[key] = 1
                    t = a.setdefault("_dur_ns", [0.0, 0])
                    t[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); t[1] += 1
    calib = {}
    GIB = float(1 << 30)
    for k, a in acc.items():
        if k.startswith("calib_"):
            n = "FETCH_SIZE" if "read" in k else "WRITE_SIZE"
            if n in a and a[n][0] > 0:
                calib[k] = GIB / (a[n][0] / a[n][1] * 1024.0)
    f_read, f_write = calib.get("calib_read_dw", 2.0), calib.get("calib_write_dw", 1.0)
    calibrated = "calib_read_dw" in calib and "calib_write_dw" in calib
    out = {}
    for k, a in acc.items():
        if k.startswith("calib_"):
            continue
        o = {n: v[0] / v[1] for n, v in a.items() if not n.startswith("_")}
        o["launches_sampled"] = max(v[1] for n, v in a.items() if not n.startswith("_"))
        o["avg_us_in_counter_runs"] = a["_dur_ns"][0] / a["_dur_ns"][1] / 1e3
        if "FETCH_SIZE" in o and "WRITE_SIZE" in o:
            o["hbm_bytes_per_launch"] = (f_read * o["FETCH_SIZE"] + f_write * o["WRITE_SIZE"]) * 1024.0
        if "SQ_INSTS_VALU_MFMA_MOPS_F32" in o:
            o["mfma_flops_per_launch"] = o["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512.0
        if o.get("SQ_INSTS_VALU_MFMA_MOPS_BF16"):  # split-bf16 kernels: 3 bf16 MFMAs per fp32-equivalent product
            o["mfma_bf16_flops_per_launch"] = o["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512.0
        if "SQ_WAVE_CYCLES" in o and o["SQ_WAVE_CYCLES"] > 0:
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if n in o:
                    o[n.lower() + "_frac_of_wave_cycles"] = o[n] / o["SQ_WAVE_CYCLES"]
        if o.get("SQ_LDS_IDX_ACTIVE"):
            o["lds_bank_conflict_frac"] = o.get("SQ_LDS_BANK_CONFLICT", 0.0) / o["SQ_LDS_IDX_ACTIVE"]
        out[k] = o
    json.dump({"workload": workload, "units": {"FETCH_SIZE": "KB", "WRITE_SIZE": "KB"}, "calibrated": calibrated,
               "f_read": f_read, "f_write": f_write, "calibration_true_over_reported": calib, "kernels": out}, sys.stdout, indent=1, sort_keys=True)
    print()


if __name__ == "__main__":
    main()
