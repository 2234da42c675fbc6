"""rocprofv3 --pmc csv output -> profiles/current/decoder_counters.json (what bench.py's roofline object quotes).

    python tools/pmc_to_json.py <dir with pmc_* runs of tools/prof_render.py> <frames per run> [out.json]

Per launch of the dominant kernel (the fused decoder):
  hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB -> bytes / launches
      FETCH_SIZE counts 64 B per 128-B request on gfx950 (MI355X_MICROARCH.md, "HBM"): doubled; WRITE_SIZE is
      taken as reported (uncalibrated, ibid.)
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)
The file carries the hash of the kernel sources it was measured on; bench.py ignores it when they changed.

Also written, next to it: cost_volume_counters.json for the stand-alone cost-volume kernel (bench.py's roofline.cost_volume),
per launch: vector instructions, the quad-cycles waves spent executing them (SQ_ACTIVE_INST_VALU counts quad-cycles:
MI355X_MICROARCH.md), vector-memory read instructions, the texture-address units' busy cycles summed over the 256 CUs,
vector-L1 accesses (64 B each), the launch's cycles (GRBM_GUI_ACTIVE / 8 XCDs) and the HBM-side bytes as above:
  valu_busy_frac = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x cycles)       ta_busy_frac = TA_TA_BUSY_sum / (256 CUs x cycles)
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
    cost_volume(acc, rows, frames, os.path.join(os.path.dirname(out), "cost_volume_counters.json"))


def cost_volume(acc, rows, frames, out):
    from matchnerf_amd.csrc.build import cost_volume_source_hash
    ks = [k for k in acc if k.startswith("cost_volume")]
    if not ks:
        return
    k = max(ks, key=lambda k: acc[k].get("SQ_WAVE_CYCLES", 0) + acc[k].get("FETCH_SIZE", 0))
    c = acc[k]
    n = {name: max(len(v), 1) for name, v in rows[k].items()}

    def per_launch(name):
        return c[name] / n[name] if name in c else None

    n_launch = max(n.values())
    res = dict(kernel=k, build_hash=cost_volume_source_hash(), frames_profiled=frames, launch_rays=65536, launches_profiled=n_launch,
               source=f"tools/profile_round.sh -> {os.path.basename(out)} (rocprofv3 --pmc passes over tools/prof_render.py)")
    cycles = per_launch("GRBM_GUI_ACTIVE")
    if cycles:
        cycles /= 8.0
        res["cycles_per_launch"] = cycles
    for name, key in (("SQ_INSTS_VALU", "valu_insts_per_launch"), ("SQ_ACTIVE_INST_VALU", "valu_active_quadcycles_per_launch"),
                      ("SQ_INSTS_VMEM_RD", "vmem_rd_insts_per_launch"), ("TA_TA_BUSY_sum", "ta_busy_cycles_sum_per_launch"),
                      ("TCP_TOTAL_CACHE_ACCESSES_sum", "l1_accesses_per_launch"), ("SQ_INSTS_LDS", "lds_insts_per_launch"),
                      ("SQ_LDS_BANK_CONFLICT", "lds_bank_conflict_cycles_per_launch"), ("SQ_WAIT_ANY", "wait_any_quadcycles_per_launch"),
                      ("SQ_WAIT_INST_ANY", "wait_inst_quadcycles_per_launch"), ("SQ_INSTS_SALU", "salu_insts_per_launch"),
                      ("SQ_INSTS_MFMA", "mfma_insts_per_launch"), ("SQ_VALU_MFMA_BUSY_CYCLES", "mfma_busy_cycles_per_launch"),
                      ("SQ_WAVE_CYCLES", "wave_quadcycles_per_launch")):
        v = per_launch(name)
        if v is not None:
            res[key] = v
    if cycles and "valu_active_quadcycles_per_launch" in res:
        res["valu_busy_frac"] = round(4 * res["valu_active_quadcycles_per_launch"] / (1024 * cycles), 4)
    if cycles and res.get("mfma_busy_cycles_per_launch"):  # the matrix form (cost_volume_mm_kernel): matrix pipe busy over 1024 SIMDs
        res["mfma_busy_frac"] = round(res["mfma_busy_cycles_per_launch"] / (1024 * cycles), 4)
    if cycles and "ta_busy_cycles_sum_per_launch" in res:
        res["ta_busy_frac"] = round(res["ta_busy_cycles_sum_per_launch"] / (256 * cycles), 4)
    if "l1_accesses_per_launch" in res:
        res["l1_bytes_per_launch"] = int(res["l1_accesses_per_launch"] * 64)
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        res["hbm_bytes_per_launch"] = int((2 * per_launch("FETCH_SIZE") + per_launch("WRITE_SIZE")) * 1024)
        res["fetch_kib_per_launch_raw"], res["write_kib_per_launch_raw"] = per_launch("FETCH_SIZE"), per_launch("WRITE_SIZE")
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else os.path.join(REPO, "profiles", "current", "decoder_counters.json"))
