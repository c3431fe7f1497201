"""Summarise rocprofv3 --pmc passes (csv output) per kernel: HBM bytes per launch and MFMA utilisation.

usage: pmc_summary.py <dir_fetch> <dir_write> <dir_sq> <out_summary.txt> <out_traffic.json>
Conventions (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 a wide
coalesced read is counted as 64 B per 128-B request, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  Values are those
of the LARGEST launch of each kernel."""
import csv, glob, json, os, re, sys
from collections import defaultdict


def load(d):
    out = defaultdict(lambda: defaultdict(list))          # kernel -> counter -> values per dispatch
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0].replace("ffvd::", "")
            out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


fetch, write, sq = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3])
lines = ["rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "
         "(gram route); separate passes for FETCH_SIZE, WRITE_SIZE and the SQ/GRBM set",
         "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (rocprofv3 reports KB; gfx950 FETCH_SIZE counts 64 B per 128-B request "
         "for wide coalesced reads)",
         "MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); values are those of the LARGEST "
         "launch of each kernel", ""]
traffic = {}
for k in sorted(set(fetch) | set(write)):
    fmax = max(fetch.get(k, {}).get("FETCH_SIZE", [0.0]))
    wmax = max(write.get(k, {}).get("WRITE_SIZE", [0.0]))
    b = (2 * fmax + wmax) * 1024
    s = "%-44s FETCH_SIZE_max=%11.0f WRITE_SIZE_max=%11.0f hbm_bytes_per_launch=%.4e" % (k[:44], fmax, wmax, b)
    c = sq.get(k, {})
    if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE"):
        i = max(range(len(c["GRBM_GUI_ACTIVE"])), key=lambda j: c["GRBM_GUI_ACTIVE"][j])
        busy, act = c["SQ_VALU_MFMA_BUSY_CYCLES"][i], c["GRBM_GUI_ACTIVE"][i]
        if busy > 0 and act > 0:
            s += " mfma_util=%.3f" % (busy / (act / 8 * 1024))
            if c.get("SQ_WAIT_INST_ANY") and c.get("SQ_WAVE_CYCLES") and sum(c["SQ_WAVE_CYCLES"]) > 0:
                s += " wait_any_frac=%.3f" % (sum(c["SQ_WAIT_INST_ANY"]) / sum(c["SQ_WAVE_CYCLES"]))   # all launches
    lines.append(s)
    traffic[k] = b
open(sys.argv[4], "w").write("\n".join(lines) + "\n")
# <1>: epilogue with the trace partials; <3>: raw tiles kept for the deferred trace pass (the full-batch launch since r01)
gram = max((v for k, v in traffic.items() if k.startswith("gram_kernel<1>") or k.startswith("gram_kernel<3>")), default=None)
kfu = max((v for k, v in traffic.items() if k.startswith("kfu_build_kernel")), default=None)
json.dump({"gram_H": gram, "project_F": kfu,
           "_note": "HBM bytes per launch (largest launch) = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc "
                    "passes, see profiles/r01_pmc_summary.txt; gram route"}, open(sys.argv[5], "w"), indent=1)
print("\n".join(lines))
