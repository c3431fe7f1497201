"""Per kernel (its LARGEST launch): the raw counters of tools/pmc_tcc.sh and the ratios that say what a kernel waits for."""
import csv, glob, os, re, sys
from collections import defaultdict
out = sys.argv[1]
data = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for p in ("p1", "p2", "p3", "p4"):
    for f in glob.glob(os.path.join(out, p, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0].replace("ffvd::", "")
            data[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(out, p, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0].replace("ffvd::", "")
            dur[(p, k)].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
print("rocprofv3 --kernel-trace --pmc <set> -- python3 bench.py ... --steps 2 --warmup 1; one pass per counter set; per kernel the LARGEST value over its launches")
print("L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS); fabric read bytes ~ 64 B x (TCC_EA0_RDREQ - RDREQ_128B) + 128 B x RDREQ_128B (guide: FETCH_SIZE tallies 128-B requests at 64 B);")
print("DRAM_CREDIT_STALL = cycles (summed over the 16 x 8 L2 channels) a read request waited for a memory-side credit; mfma = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)")
print()
for k in sorted(data, key=lambda k: -max(data[k].get("TCC_REQ_sum", [0]))):
    c = {n: max(v) for n, v in data[k].items()}
    if c.get("TCC_REQ_sum", 0) < 1e6:
        continue
    d3 = max(dur.get(("p3", k), [0.0]))
    line = ["%-34s" % k[:34]]
    if "TCC_HIT_sum" in c:
        line.append("L2 req %.3e hit %.3f tag_stall/req %.2f" % (c["TCC_REQ_sum"], c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1), c.get("TCC_TAG_STALL_sum", 0) / c["TCC_REQ_sum"]))
    if "TCC_EA0_RDREQ_sum" in c:
        rb = 64.0 * (c["TCC_EA0_RDREQ_sum"] - c.get("TCC_EA0_RDREQ_128B_sum", 0)) + 128.0 * c.get("TCC_EA0_RDREQ_128B_sum", 0)
        d2 = max(dur.get(("p2", k), [0.0]))
        line.append("| EA rdreq %.3e (128B %.3e, DRAM %.3e) = %.1f GB -> %.2f TB/s over %.0f us; dram_credit_stall/rdreq %.2f" % (
            c["TCC_EA0_RDREQ_sum"], c.get("TCC_EA0_RDREQ_128B_sum", 0), c.get("TCC_EA0_RDREQ_DRAM_sum", 0), rb / 1e9, rb / max(d2, 1e-9) / 1e6, d2,
            c.get("TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum", 0) / max(c["TCC_EA0_RDREQ_sum"], 1)))
    if "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"] > 0:
        wc = max(c.get("SQ_WAVE_CYCLES", 1), 1)
        line.append("| mfma %.3f; of wave-cycles: wait_any %.2f wait_inst %.2f active_inst %.2f; %.0f us" % (
            c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (c["GRBM_GUI_ACTIVE"] / 8 * 1024), c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc,
            c.get("SQ_ACTIVE_INST_ANY", 0) / wc, d3))
    if "SQ_INSTS_LDS" in c:
        line.append("| lds insts %.3e bank_conflict/idx_active %.3f wait_inst_lds %.3e" % (c["SQ_INSTS_LDS"], c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1), c.get("SQ_WAIT_INST_LDS", 0)))
    print(" ".join(line))
