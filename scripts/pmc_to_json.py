#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; scripts/pmc_traffic.py workload) into
profiles/pmc_traffic.json: HBM-side bytes per launch of the benchmark kernels, corrected as
MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 on gfx950, verified on the k_halo_pack calibration launch whose
true byte count is known; WRITE_SIZE exact).  bench.py reports these as roofline.traffic."""
import csv, json, sys, collections
fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
def load(d):
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{d}/r02_counter_collection.csv")):
        rows[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]) * 1024.0)
    return {k: sum(v) / len(v) for k, v in rows.items()}
F, W = load(fetch_dir), load(write_dir)
cal = [(k, v) for k, v in F.items() if "k_halo_pack" in k[0]][0]
n = cal[0][1]                                  # one thread per packed double
true_read = n * 8 + n * 4
factor = true_read / cal[1]
res = {"fetch_correction": round(factor, 4), "calibration": {"kernel": "k_halo_pack identity gather", "true_read_bytes": true_read,
       "FETCH_SIZE_bytes": cal[1], "true_write_bytes": n * 8, "WRITE_SIZE_bytes": W[cal[0]]}, "kernels": {}}
for (k, g), v in F.items():
    if "k_elem_apply" in k or "k_gather_sum" in k or "k_apply_wave" in k or "k_wave_perim" in k:
        name = ("k_apply_wave<3,UMAT>" if "k_apply_wave" in k else "k_wave_perim" if "k_wave_perim" in k else
                "k_elem_apply<3,UMAT>" if "k_elem_apply" in k else "k_gather_sum<2>")
        res["kernels"].setdefault(name, []).append({"grid_threads": g, "read_bytes": v * 2.0 if abs(factor - 2) < 0.1 else v * factor,
                                                     "write_bytes": W.get((k, g)), "total_bytes": (v * factor) + W.get((k, g), 0.0)})
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
