"""Aggregate rocprofv3 --pmc passes (one directory per pass, each with *_counter_collection.csv) into one per-kernel
table + the JSON bench.py reads for roofline.traffic.
usage: pmc_summarize.py <dir with pass subdirs> <out.csv> <out.json>
HBM-side bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are KB per dispatch; on gfx950
FETCH_SIZE tallies 128-byte read requests at 64 B, so wide coalesced reads are DOUBLED (checked here on the
LayerNorm kernel, whose algorithmic read is exactly its fp32 input); WRITE_SIZE is used as reported."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text_to_sound_synthesis_amd.build import source_fingerprint

src, out_csv, out_json = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(src + "/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, c in acc.items():
    if "ds_" not in k:
        continue
    n = max(len(v) for v in c.values())
    avg = lambda name: sum(c[name]) / len(c[name]) if c.get(name) else None
    fetch, write = avg("FETCH_SIZE"), avg("WRITE_SIZE")
    hit, miss = sum(c.get("TCC_HIT_sum", [])), sum(c.get("TCC_MISS_sum", []))
    mfma, gui = avg("SQ_VALU_MFMA_BUSY_CYCLES"), avg("GRBM_GUI_ACTIVE")
    rows.append({
        "kernel": k.split("(")[0].replace("void ", ""), "dispatches": n,
        "hbm_read_MB_per_launch": None if fetch is None else round(2 * fetch * 1024 / 1e6, 2),
        "hbm_write_MB_per_launch": None if write is None else round(write * 1024 / 1e6, 2),
        "l2_hit_rate": None if hit + miss == 0 else round(hit / (hit + miss), 4),
        "mfma_busy_cycles_per_launch": None if mfma is None else round(mfma),
        "grbm_gui_active_per_launch": None if gui is None else round(gui),
    })
rows.sort(key=lambda r: -(r["hbm_read_MB_per_launch"] or 0) * r["dispatches"])
with open(out_csv, "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)
table = {r["kernel"]: r for r in rows}
# bench.py reports roofline.traffic from this file only while the kernel sources are the ones that were measured
table["_meta"] = {"source_sha16": source_fingerprint(),
                  "workload": "tools/pmc_step.py: two B=64 denoiser sampling steps (ds_denoiser_step_rng, padded-row mode), "
                              "default precision",
                  "gemm_instantiations": sorted(k for k in table if "ds_gemm" in k)}
json.dump(table, open(out_json, "w"), indent=1)
for r in rows[:8]:
    print(r)
