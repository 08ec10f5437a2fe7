#!/usr/bin/env python3
"""Measures the instruction-issue ceilings of the device (viennaray_amd/csrc/vr_bench.hip through
vr_debug_issue_rate) for the mixes the tracer and the generator are made of, at 1..8 resident
waves per SIMD, and the clock the chip sustains meanwhile.

    python3 tools/issue_ceiling.py [out.json]
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv \
        -d gpurun_out/issue_pmc -- python3 tools/issue_ceiling.py      (cross-checks the counts)

Per row: counted instructions / s chip-wide, the same per SIMD-cycle (VALU kinds; per CU-cycle for
the SALU kind) at the measured clock — i.e. cycles per wave-instruction = 1 / that figure.
MI355X_MICROARCH.md says a wave64 VALU instruction issues over 2 cycles (4 with one wave alone)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import viennaray_amd as vr  # noqa: E402

KINDS = {0: "valu_f32_independent", 1: "valu_f32_dependent_chain", 2: "mt19937_64_seed_step",
         3: "salu", 4: "packet_mix_24valu_16salu", 5: "independent_mix_24valu_16salu",
         6: "mt_seed_step_three_chained_mad64_rejected"}


def main():
    t = vr.TraceDisk(3)
    rows = []
    for kind, name in KINDS.items():
        for w in (1, 2, 4, 6, 7, 8):
            r = t.debugIssueRate(kind, w, iters=40000 if kind not in (2, 6) else 20000)
            simds, cus = 1024, 256
            clk = r["clock_hz"]
            row = dict(kind=name, waves_per_simd=w, rate=r["rate"], clock_ghz=clk / 1e9, seconds=r["seconds"],
                       count=r["count"])
            if kind == 3:
                row["per_cu_cycle"] = r["rate"] / (cus * clk)
            else:
                row["per_simd_cycle"] = r["rate"] / (simds * clk)
                row["cycles_per_wave_instr"] = simds * clk / r["rate"]
            if kind in (4, 5):
                row["salu_per_cu_cycle"] = r["rate"] * (16.0 / 24.0) / (cus * clk)
            rows.append(row)
            print(json.dumps(row))
    if len(sys.argv) > 1:
        json.dump(dict(device="MI355X gfx950", rows=rows), open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
