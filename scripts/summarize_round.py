"""Per-kernel summary of scripts/profile_round.sh's rocprofv3 passes: time (kernel trace), MFMA-busy share, HBM bytes."""
import csv, glob, json, os, re, sys
from collections import defaultdict

out_dir, out_json = sys.argv[1], sys.argv[2]
def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("hct::", "").replace("(anonymous namespace)::", "")
    return n.strip()
def find(sub, pat):
    c = glob.glob(os.path.join(out_dir, sub, "**", pat), recursive=True)
    return c[0] if c else None
res = {"kernels": {}}
kt = find("kt", "*kernel_trace.csv")
dur = defaultdict(list)
if kt:
    for r in csv.DictReader(open(kt)):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
steps = 7  # bench: 2 warm-up + 5 timed
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    res["kernels"][k] = {"launches_per_step": round(len(v) / steps, 2), "avg_us": round(sum(v) / len(v), 2), "ms_per_step": round(sum(v) / steps / 1e3, 3)}
def counters(sub):
    f = find(sub, "*counter_collection.csv")
    acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
    if not f:
        return acc, n
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"]); acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    return acc, n
acc, n = counters("mfma")
for k, c in acc.items():
    if k not in res["kernels"] or "GRBM_GUI_ACTIVE" not in c:
        continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0  # the counter is summed over the 8 XCDs
    e = res["kernels"][k]
    e["mfma_busy_frac"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024), 4) if cyc else None  # 256 CUs x 4 SIMDs
    e["cu_busy_frac"] = round(c.get("SQ_BUSY_CU_CYCLES", 0.0) / (cyc * 256), 4) if cyc and "SQ_BUSY_CU_CYCLES" in c else None
    e["clock_GHz"] = round(cyc / (sum(dur[k]) / len(dur[k]) * n[k]["GRBM_GUI_ACTIVE"]) / 1e3, 3) if k in dur else None
for sub, name, mul in (("fetch", "FETCH_SIZE", 2.0), ("write", "WRITE_SIZE", 1.0)):
    a, m = counters(sub)
    for k, c in a.items():
        if k in res["kernels"] and name in c:
            res["kernels"][k][name.lower() + "_bytes_per_launch" + ("_x2_corrected" if mul == 2.0 else "")] = round(c[name] / m[k][name] * 1024 * mul)
# VALU / MFMA / LDS instruction counters of the third pass (wave-instructions per launch; VALU counts include the MFMAs)
a, m = counters("valu")
for k, c in a.items():
    if k not in res["kernels"]:
        continue
    e = res["kernels"][k]
    for name, key in (("SQ_INSTS_VALU", "valu_wave_insts_per_launch"), ("SQ_INSTS_MFMA", "mfma_wave_insts_per_launch"),
                      ("SQ_ACTIVE_INST_VALU", "valu_active_cycles_per_launch"), ("SQ_ACTIVE_INST_LDS", "lds_active_cycles_per_launch")):
        if name in c and m[k][name]:
            e[key] = round(c[name] / m[k][name])
    if "valu_wave_insts_per_launch" in e and e.get("clock_GHz"):
        # share of the launch the SIMDs would spend issuing those instructions at 4 clocks each (an issue-bound estimate, 1024 SIMDs)
        e["valu_issue_frac_at_4clk"] = round(e["valu_wave_insts_per_launch"] * 4 / 1024 / (e["avg_us"] * 1e3 * e["clock_GHz"]), 3)
res["note"] = ("mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); FETCH_SIZE doubled per the gfx950 correction "
               "(MI355X_MICROARCH.md, HBM); valu_issue_frac_at_4clk = SQ_INSTS_VALU * 4 clk / 1024 SIMDs / launch cycles; separate rocprofv3 passes of `bench.py --steps 5 --warmup 2`")
json.dump(res, open(out_json, "w"), indent=1)
tot = sum(v["ms_per_step"] for v in res["kernels"].values())
print(f"sum of kernel time: {tot:.2f} ms/step")
for k, v in list(res["kernels"].items())[:22]:
    print(f"{v['ms_per_step']:8.3f} ms  {v['launches_per_step']:7.1f} x {v['avg_us']:8.1f} us  mfma {v.get('mfma_busy_frac')}  {k[:70]}")
