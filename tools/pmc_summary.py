#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes per kernel.

    python tools/pmc_summary.py OUT.json  PASS_DIR [PASS_DIR ...]

Each PASS_DIR is the -d directory of one `rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py ...` run (counters are
collected in separate passes, never together with a trace).  Per kernel family the script reports launches and the per-launch
mean of every counter found, plus, when FETCH_SIZE and WRITE_SIZE are both present, the HBM-side bytes per launch
    (2 x FETCH_SIZE + WRITE_SIZE) x 1024
(MI355X_MICROARCH.md, HBM section: both counters are in KiB, and on gfx950 FETCH_SIZE reports half of a wide coalesced read stream).
bench.py copies `hbm_bytes_per_launch` of the dominant kernel into `roofline.traffic`.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

FAMILIES = [("gather_conv_h3_kernel", "gather_conv"), ("halo_conv_h3_kernel", "gather_conv"), ("vgg_conv1_kernel", "vgg_conv1"), ("gather_conv_kernel", "gather_conv_f32"), ("col2im_rgb_tanh_kernel", "convt_rgb"),
            ("l2_prepare_kernel", "l2_prepare"), ("l2_knn_i8", "l2_knn"), ("feat_knn_h1", "feat_knn"), ("feat_knn_kernel", "feat_knn_split"),
            ("lpips_tap", "lpips_tap"), ("maxpool2", "maxpool2"), ("vgg_input", "vgg_input"), ("row_sqnorm", "row_sqnorm"), ("pixelnorm_split", "pixelnorm")]


def family(name):
    for needle, fam in FAMILIES:
        if needle in name:
            return fam
    return None


def main():
    out_path, dirs = sys.argv[1], sys.argv[2:]
    sums = defaultdict(lambda: defaultdict(float))          # fam -> counter -> sum over dispatches
    launches = defaultdict(lambda: defaultdict(set))        # fam -> counter -> dispatch ids
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    fam = family(row["Kernel_Name"])
                    if fam is None:
                        continue
                    c = row["Counter_Name"]
                    sums[fam][c] += float(row["Counter_Value"])
                    launches[fam][c].add((f, row["Dispatch_Id"]))
    res = {"_how": __doc__.strip().splitlines()[0] + " See tools/pmc_summary.py; passes: " + ", ".join(os.path.basename(os.path.normpath(d)) for d in dirs)}
    for fam in sums:
        e = {}
        for c, v in sums[fam].items():
            n = len(launches[fam][c])
            e["launches"] = n
            e[c + "_per_launch"] = v / n
        if "FETCH_SIZE_per_launch" in e and "WRITE_SIZE_per_launch" in e:
            e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE_per_launch"] + e["WRITE_SIZE_per_launch"]) * 1024.0
        res[fam] = e
    with open(out_path, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
