#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs into profiles/.

Usage: tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <tag> <WxHxS>
FETCH_SIZE / WRITE_SIZE are in KiB. On gfx950 FETCH_SIZE reports exactly half of
the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section), so reads
are doubled; k_input (a pure streaming copy of known size) serves as the
in-run calibration: its corrected read must equal width*height*1.5 bytes."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def avg_per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    fetch_csv, write_csv, tag, geom = sys.argv[1:5]
    f = avg_per_kernel(fetch_csv, "FETCH_SIZE")
    w = avg_per_kernel(write_csv, "WRITE_SIZE")
    rows = []
    traffic = {}
    for k in sorted(set(f) | set(w)):
        if not k.startswith("cedar::"):
            continue
        fr, n = f.get(k, (0.0, 0))
        wr, _ = w.get(k, (0.0, 0))
        rd_bytes = 2.0 * fr * 1024.0  # gfx950 correction
        wr_bytes = wr * 1024.0
        rows.append((k, n, fr, wr, rd_bytes, wr_bytes, rd_bytes + wr_bytes))
        stage = k.replace("cedar::k_", "").split("<")[0]
        stage = {"deblock2": "deblock", "deblock2_planes": "deblock", "intra_recon2": "intra_recon", "cavlc2": "cavlc"}.get(stage, stage)
        if k.startswith("cedar::k_deblock<") and "," in k:   # older k_deblock<IDR, planes>: 1 luma, 2 chroma, 3 both
            planes = k.split(",")[1].strip(" >")
            stage = {"1": "deblock_luma", "2": "deblock_chroma"}.get(planes, "deblock")
        key = "%s@%s" % (stage, geom)
        if n >= traffic.get("_n_" + key, 0):  # IDR / P variants: keep the most frequent one
            traffic[key] = rd_bytes + wr_bytes
            traffic["_n_" + key] = n
    traffic = {k: round(v) for k, v in traffic.items() if not k.startswith("_n_")}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    out_csv = os.path.join(ROOT, "profiles", "%s_pmc_hbm_traffic.csv" % tag)
    with open(out_csv, "w") as fh:
        fh.write("kernel,launches,FETCH_SIZE_KiB_avg,WRITE_SIZE_KiB_avg,read_bytes_corrected_x2,write_bytes,hbm_bytes_per_launch\n")
        for r in rows:
            fh.write("%s,%d,%.1f,%.1f,%.0f,%.0f,%.0f\n" % r)
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    old = json.load(open(tpath)) if os.path.exists(tpath) else {}
    old.update(traffic)
    old["_note"] = ("HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (KiB->bytes), rocprofv3 --pmc, separate "
                    "passes; entropy = sum of its kernels is not listed, per-kernel rows are in *_pmc_hbm_traffic.csv")
    json.dump(old, open(tpath, "w"), indent=1, sort_keys=True)
    print(open(out_csv).read())


if __name__ == "__main__":
    main()
