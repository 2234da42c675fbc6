"""Aggregate rocprofv3 --pmc csv output (counter_collection.csv) per kernel and counter."""
import csv
import glob
import sys
from collections import defaultdict


def main(root):
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"].split("(")[0][:60]
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[k].add((path, row["Dispatch_Id"]))
    for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", acc[k].get("FETCH_SIZE", 0))):
        if not any(s in k for s in ("decoder_kernel", "decoder_pp_kernel", "cost_volume", "window_attention", "encoder_block", "conv_kernel",
                                    "conv_stem", "qkv_", "instance_norm", "wa_bwd", "gemm_", "eb_")):
            continue
        n = max(len(calls[k]), 1)
        print(f"## {k}  ({n} dispatch rows)")
        for c, v in sorted(acc[k].items()):
            print(f"  {c:32s} total {v:18.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
