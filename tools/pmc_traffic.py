#!/usr/bin/env python3
"""Builds OUT/pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE (and the mask kernel's SQ_INSTS_VALU / SQ_WAVES) passes that
tools/gpu_visit.sh `traffic` ran: HBM bytes per launch = FETCH_SIZE x 2 (the gfx950 correction for wide coalesced reads,
MI355X_MICROARCH.md "HBM") + WRITE_SIZE, both in KiB.  Every entry is stamped with the sha256 of the kernel's source file: bench.py
prints roofline.traffic only when that stamp matches the source it runs."""
import hashlib
import json
import re
import sys

out = sys.argv[1]
res = {}


def sha(name):
    return hashlib.sha256(open("transferia_amd/csrc/" + name, "rb").read()).hexdigest()


try:
    mv = {m.group(1): float(m.group(2)) for m in (re.search(r"(SQ_INSTS_VALU|SQ_WAVES)\s+per_dispatch=([0-9.e+]+)", l) for l in open(out + "/pmc_mask.log") if "mask_hmac" in l) if m}
    if "SQ_INSTS_VALU" in mv and mv.get("SQ_WAVES"):
        res["mask_hmac_sha256"] = {"workload": "csv", "rows_per_launch": 1 << 20, "valu_wave_instructions_per_launch": mv["SQ_INSTS_VALU"], "waves": mv["SQ_WAVES"],
                                   "valu_instructions_per_value": round(mv["SQ_INSTS_VALU"] / mv["SQ_WAVES"], 1), "source_file": "tf_transform.hip", "source_sha256": sha("tf_transform.hip"),
                                   "source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES of this build (a wave-instruction is one instruction for each of the wave's 64 values)"}
except OSError:
    pass
for kern, wl, src, rows in (("csv_parse_regular", "csv", "tf_csv.hip", 1 << 20), ("json_parse_quick", "json", "tf_jsonquick.inc", 1 << 18), ("ser_chunk_write", "configs3", "tf_serialize.hip", 1 << 20)):
    vals = {}
    for leg in ("fetch", "write"):
        try:
            for line in open(f"{out}/pmc_{wl}_{leg}.log"):
                m = re.search(r"(FETCH_SIZE|WRITE_SIZE)\s+per_dispatch=([0-9.e+]+)", line)
                if m and kern[:12] in line:
                    vals[m.group(1)] = float(m.group(2))
        except OSError:
            pass
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        b = int(vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024)
        res[kern] = {"rows_per_launch": rows, "workload": wl, "bytes_per_launch": b, "fetch_kib": vals["FETCH_SIZE"], "write_kib": vals["WRITE_SIZE"], "source_file": src, "source_sha256": sha(src),
                     "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB, separate passes) of this build, tools/gpu_visit.sh evidence; FETCH_SIZE x2 per MI355X_MICROARCH.md"}
try:  # the dominant kernel's instruction counters (one pass of their own): what its time is made of
    cv = {m.group(1): float(m.group(2)) for m in (re.search(r"(SQ_INSTS_VALU|SQ_WAVES|SQ_WAIT_INST_LDS|SQ_INSTS_LDS|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES)\s+per_dispatch=([0-9.e+]+)", l)
                                                  for l in open(out + "/pmc_csv_valu.log") if "csv_parse_regul" in l) if m}
    if "csv_parse_regular" in res and cv.get("SQ_WAVES"):
        res["csv_parse_regular"].update({"valu_wave_instructions_per_launch": cv.get("SQ_INSTS_VALU"), "lds_wave_instructions_per_launch": cv.get("SQ_INSTS_LDS"), "waves": cv["SQ_WAVES"],
                                         "wait_inst_lds_cycles": cv.get("SQ_WAIT_INST_LDS"), "wave_cycles": cv.get("SQ_WAVE_CYCLES"),
                                         "valu_issue_ms_at_4_cycles": round((cv.get("SQ_INSTS_VALU") or 0) * 4 / 1024 / 2.4e9 * 1e3, 4)})
except OSError:
    pass
try:  # the busy side: what the counters can and cannot say about "VALU-bound"
    bv = {m.group(1): float(m.group(2)) for m in (re.search(r"(SQ_ACTIVE_INST_VALU|SQ_ACTIVE_INST_ANY|SQ_WAIT_ANY|SQ_WAIT_INST_ANY|SQ_THREAD_CYCLES_VALU|SQ_INSTS_SALU|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES)\s+per_dispatch=([0-9.e+]+)", l)
                                                  for l in open(out + "/pmc_csv_busy.log") if "csv_parse_regul" in l) if m}
    e = res.get("csv_parse_regular")
    if e and bv.get("SQ_BUSY_CYCLES") and e.get("valu_wave_instructions_per_launch"):
        insts = e["valu_wave_instructions_per_launch"]
        busy = bv["SQ_BUSY_CYCLES"] / 32.0            # summed over the 32 shader engines of the chip (8 XCDs x 4): cycles the kernel was resident
        e["busy"] = {"sq_busy_cycles_per_se": round(busy), "kernel_ms_at_2p4_ghz": round(busy / 2.4e9 * 1e3, 4),
                     "sq_active_inst_valu": bv.get("SQ_ACTIVE_INST_VALU"), "note_active_inst_valu": "on gfx950 this counter counts INSTRUCTIONS (it equals SQ_INSTS_VALU), not busy cycles: calibrated in round 3 (profiles/r05_valu_lds_microbench.txt)",
                     "valu_issue_cycles_per_simd_at_4": round(insts * 4 / 1024), "valu_busy_frac_at_4_cycles_per_inst": round(insts * 4 / 1024 / busy, 3),
                     "valu_busy_frac_at_2p4_cycles_per_inst": round(insts * 2.4 / 1024 / busy, 3),
                     "active_lanes_per_valu_inst": round(bv["SQ_THREAD_CYCLES_VALU"] / insts, 1) if bv.get("SQ_THREAD_CYCLES_VALU") else None,
                     "wave_wait_any_frac": round(bv["SQ_WAIT_ANY"] / bv["SQ_WAVE_CYCLES"], 3) if bv.get("SQ_WAIT_ANY") and bv.get("SQ_WAVE_CYCLES") else None,
                     "wave_wait_inst_any_frac": round(bv["SQ_WAIT_INST_ANY"] / bv["SQ_WAVE_CYCLES"], 3) if bv.get("SQ_WAIT_INST_ANY") and bv.get("SQ_WAVE_CYCLES") else None,
                     "salu_instructions": bv.get("SQ_INSTS_SALU"),
                     "reading": "VALU issue fills 0.6 (every instruction at the 2.4-cycle rate of a two-operand add) to 1.0 (every instruction at the 4.4 cycles most of this kernel's "
                                "instructions cost: v_perm / alignbyte / dot4 / bfe / cndmask / mad, tools/microbench/valu_rate.hip) of the kernel's resident cycles; a wave waits half of ITS "
                                "cycles (wave_wait_any_frac) — on a barrier, an LDS result, its turn at the VALU — while the SIMD it sits on issues for one of its five neighbours"}
except OSError:
    pass
json.dump(res, open(out + "/pmc_traffic.json", "w"), indent=1)
print("== traffic", {k: v.get("bytes_per_launch", v.get("valu_instructions_per_value")) for k, v in res.items()})
