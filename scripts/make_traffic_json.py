#!/usr/bin/env python3
"""profiles/traffic.json from the PMC CSVs of scripts/profile_round.sh.
usage: make_traffic_json.py OUTDIR SOURCE_TAG > profiles/traffic.json
Counters are in KiB; FETCH_SIZE is doubled (gfx950 counts a coalesced stream at half its bytes -- calibrated on the
all-miss launch, which reads exactly 28 B/ray), WRITE_SIZE is exact (MI355X_MICROARCH.md, HBM section)."""
import csv, json, sys
out, tag = sys.argv[1], sys.argv[2]
R, N = 67108864, 4096
def first(path, kernel, counter):
    vals = [float(r["value"]) for r in csv.DictReader(open(path)) if kernel in r["kernel"] and r["counter"] == counter]
    return vals
res = {"_how": "scripts/profile_round.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in SEPARATE passes over "
               "`python scripts/prof_kernels.py --iters 1 fwd adj miss`; KiB; FETCH_SIZE x2 (gfx950), WRITE_SIZE exact.",
       "source": tag}
f = first(f"{out}/pmc_FETCH_SIZE.csv", "hf_trace_kernel<2", "FETCH_SIZE"); w = first(f"{out}/pmc_WRITE_SIZE.csv", "hf_trace_kernel<2", "WRITE_SIZE")
# dispatch order of prof_kernels: warm-up fwd, fwd (x iters+1), ..., the last trace<2> launches are the all-miss ones
fwd_f, fwd_w, miss_f = f[0], w[0], f[-1]
res["calibration"] = {"all_miss_FETCH_SIZE_KiB": miss_f, "all_miss_read_bytes_exact": 28.0 * R, "ratio": 28.0 * R / (miss_f * 1024)}
res["hf_trace_kernel<2>"] = {"FETCH_SIZE_KiB": fwd_f, "WRITE_SIZE_KiB": fwd_w, "read_bytes": 2 * fwd_f * 1024, "write_bytes": fwd_w * 1024,
                             "hbm_bytes_per_launch": 2 * fwd_f * 1024 + fwd_w * 1024, "algorithmic_bytes": R * 104.0 + N * N * 20.0 / 3.0}
f = first(f"{out}/pmc_FETCH_SIZE.csv", "hf_adjoint_kernel", "FETCH_SIZE"); w = first(f"{out}/pmc_WRITE_SIZE.csv", "hf_adjoint_kernel", "WRITE_SIZE")
res["hf_adjoint_kernel"] = {"FETCH_SIZE_KiB": f[-1], "WRITE_SIZE_KiB": w[-1], "read_bytes": 2 * f[-1] * 1024, "write_bytes": w[-1] * 1024,
                            "hbm_bytes_per_launch": 2 * f[-1] * 1024 + w[-1] * 1024}
print(json.dumps(res, indent=1))
