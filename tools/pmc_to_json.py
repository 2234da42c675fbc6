"""rocprofv3 --pmc csv output -> profiles/decoder_counters.json (what bench.py's roofline object quotes).

    python tools/pmc_to_json.py <dir with pmc_* runs of tools/prof_render.py> <frames per run> [out.json]

Per launch of the dominant kernel (the fused decoder):
  hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB -> bytes / launches
      FETCH_SIZE counts 64 B per 128-B request on gfx950 (MI355X_MICROARCH.md, "HBM"): doubled; WRITE_SIZE is
      taken as reported (uncalibrated, ibid.)
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)
The file carries the hash of the kernel sources it was measured on; bench.py ignores it when they changed.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main(root, frames, out):
    from matchnerf_amd.csrc.build import source_hash
    acc = defaultdict(lambda: defaultdict(float))
    rows = defaultdict(lambda: defaultdict(set))
    for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                rows[k][row["Counter_Name"]].add((path, row["Dispatch_Id"]))
    dec = [k for k in acc if k.startswith("decoder_kernel") or k.startswith("decoder_pp_kernel")]
    if not dec:
        raise SystemExit(f"no decoder kernel rows under {root}")
    k = max(dec, key=lambda k: acc[k].get("SQ_WAVE_CYCLES", 0) + acc[k].get("FETCH_SIZE", 0))
    c = acc[k]
    launches = {name: len(v) for name, v in rows[k].items()}
    n_fetch, n_write = launches.get("FETCH_SIZE", 0), launches.get("WRITE_SIZE", 0)
    res = dict(kernel=k, build_hash=source_hash(), frames_profiled=frames, launch_rays=65536,
               source=f"tools/profile_round.sh -> {os.path.basename(out)} (rocprofv3 --pmc passes over tools/prof_render.py)")
    if n_fetch and n_write:
        res["hbm_bytes_per_launch"] = int((2 * c["FETCH_SIZE"] / n_fetch + c["WRITE_SIZE"] / n_write) * 1024)
        res["fetch_kib_per_launch_raw"] = c["FETCH_SIZE"] / n_fetch
        res["write_kib_per_launch_raw"] = c["WRITE_SIZE"] / n_write
    if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        res["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
        res["sq_insts_mfma_per_launch"] = c.get("SQ_INSTS_MFMA", 0) / max(launches.get("SQ_INSTS_MFMA", 1), 1)
        res["sq_insts_valu_per_launch"] = c.get("SQ_INSTS_VALU", 0) / max(launches.get("SQ_INSTS_VALU", 1), 1)
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else os.path.join(REPO, "profiles", "decoder_counters.json"))
