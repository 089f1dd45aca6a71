#!/usr/bin/env python3
"""Summarise the SQ counter passes of tools/collect_r04_valu.sh into profiles/<tag>_pmc_valu.csv.

Usage: tools/pmc_valu_summary.py <pass A csv> <pass B csv> <tag> <macroblocks per launch of the per-picture kernels>

Pass A: SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS      (wave instructions issued, summed over the chip)
Pass B: SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
Per kernel (averages per launch): VALU wave-instructions, the same per wave and per macroblock, the share of the chip's
VALU issue slots the launch used = SQ_INSTS_VALU / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs / 4 cycles per issue).
SQ_ACTIVE_INST_VALU counts quad-cycles (MI355X_MICROARCH.md, cycle constants table): x4 = SIMD-cycles with a VALU
instruction in flight."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("cedar::"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: (sum(v) / len(v), len(v)) for c, v in d.items()} for k, d in agg.items()}


def main():
    a_csv, b_csv, tag, mbs = sys.argv[1:5]
    mbs = float(mbs)
    A, B = per_kernel(a_csv), per_kernel(b_csv)
    rows = []
    out = {}
    for k in sorted(set(A) | set(B)):
        a, b = A.get(k, {}), B.get(k, {})
        g = lambda d, c: d.get(c, (0.0, 0))[0]
        n = a.get("SQ_INSTS_VALU", (0, 0))[1]
        valu, waves, salu, lds = g(a, "SQ_INSTS_VALU"), g(a, "SQ_WAVES"), g(a, "SQ_INSTS_SALU"), g(a, "SQ_INSTS_LDS")
        act, busy, wcyc, gui = g(b, "SQ_ACTIVE_INST_VALU"), g(b, "SQ_BUSY_CYCLES"), g(b, "SQ_WAVE_CYCLES"), g(b, "GRBM_GUI_ACTIVE")
        slots = gui / 8.0 * 256 * 4 / 4.0 if gui else 0.0   # VALU issue slots of the chip while the launch ran
        rows.append((k, n, valu, waves, valu / waves if waves else 0, valu / mbs, salu, lds, act * 4, busy, wcyc, gui / 8.0,
                     valu / slots if slots else 0))
        # an issue slot = 4 SIMD cycles with a VALU instruction in flight = one SQ_ACTIVE_INST_VALU count (a quarter-rate
        # v_qsad_pk_u16_u8 holds its SIMD for 4 of them: SQ_INSTS_VALU counts it once)
        out[k.replace("cedar::", "")] = {"valu_per_launch": valu, "valu_per_macroblock": valu / mbs, "waves": waves,
                                         "issue_slot_share": valu / slots if slots else None,
                                         "active_slots_per_macroblock": act / mbs,
                                         "active_slot_share": act / slots if slots else None}
    path = os.path.join(ROOT, "profiles", "%s_pmc_valu.csv" % tag)
    with open(path, "w") as fh:
        fh.write("kernel,launches,SQ_INSTS_VALU,SQ_WAVES,valu_per_wave,valu_per_macroblock(%d per launch),SQ_INSTS_SALU,SQ_INSTS_LDS,"
                 "SQ_ACTIVE_INST_VALU_x4_simd_cycles,SQ_BUSY_CYCLES,SQ_WAVE_CYCLES,GRBM_GUI_ACTIVE_per_xcd,share_of_chip_valu_issue_slots\n" % mbs)
        for r in rows:
            fh.write("%s,%d,%.0f,%.0f,%.1f,%.1f,%.0f,%.0f,%.0f,%.0f,%.0f,%.0f,%.4f\n" % r)
    jp = os.path.join(ROOT, "profiles", "valu.json")
    old = json.load(open(jp)) if os.path.exists(jp) else {}
    old[tag] = out
    json.dump(old, open(jp, "w"), indent=1, sort_keys=True)
    print(open(path).read())


if __name__ == "__main__":
    main()
