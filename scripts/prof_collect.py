"""Condense the rocprofv3 output of scripts/prof_r0N.sh into small files fit for profiles/:
   <what>_kernel_stats.csv   (name, calls, total / average / min / max duration in us, percentage)
   <what>_pmc_summary.txt    (mean counter value per dispatch, per kernel) + derived figures
   <what>_traffic.json       (HBM-side bytes per launch: 2 * FETCH_SIZE + WRITE_SIZE, KiB counters; gfx950 note in
                              MI355X_MICROARCH.md: FETCH_SIZE tallies 128-B requests at 64 B)

    python scripts/prof_collect.py gpurun_out/r02
"""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
summ = os.path.join(root, "summary")
os.makedirs(summ, exist_ok=True)

# Every *_traffic.json names the binary its counters were collected on: sha256 of the libmik.so of THIS checkout (the one the
# profiled commands loaded) and the mangled names of the kernels it lists, so that bench.py can tell whether the constants still describe the
# library it has loaded (`traffic_binary_matches`) and withholds them otherwise.
import hashlib
import re
import subprocess
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "iterativesolvers.jl_amd", "libmik.so")


def binary_record(kernel_names=()):
    rec = {"libmik_sha256": None, "libmik_bytes": None, "mangled": {}}
    try:
        blob = open(LIB, "rb").read()
    except OSError:
        return rec
    rec["libmik_sha256"], rec["libmik_bytes"] = hashlib.sha256(blob).hexdigest(), len(blob)
    # kernel symbols of the embedded gfx950 code object: strings of the form _Z...<name>...; demangled with c++filt and matched on the text
    cands = sorted(set(m.decode() for m in re.findall(rb"_Z[A-Za-z0-9_]{6,400}", blob) if b"k_" in m))
    import shutil
    filt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    try:
        dem = subprocess.run([filt], input="\n".join(cands), capture_output=True, text=True, timeout=60).stdout.splitlines() if cands and filt else []
    except Exception:
        dem = []
    table = {}
    for m, d in zip(cands, dem):
        d = d.replace("void ", "").replace("(anonymous namespace)::", "").strip().split("(")[0]
        table.setdefault(d, m)
    for k in kernel_names:
        if k in table:
            rec["mangled"][k] = table[k]
    return rec


def tkey(k):
    """key of a kernel in <what>_traffic.json: the name up to its template list; k_map / k_map_pro kernels by their Op"""
    import re
    m = re.match(r"(k_map(?:_pro)?)<[^,]+, [^,]+, (Op\w+)", k)
    return f"{m.group(1)}<{m.group(2)}>" if m else k.split("<")[0]


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "").strip()
    return n.split("(")[0][:110]


for d in sorted(glob.glob(os.path.join(root, "*"))):
    what = os.path.basename(d)
    if not os.path.isdir(d) or what == "summary":
        continue
    # ---- kernel stats ---------------------------------------------------------------------------
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        with open(os.path.join(summ, f"{what}_kernel_stats.csv"), "w") as o:
            o.write("kernel,calls,total_us,avg_us,min_us,max_us,percent\n")
            for r in rows:
                o.write(f"\"{short(r['Name'])}\",{r['Calls']},{float(r['TotalDurationNs']) / 1e3:.1f},{float(r['AverageNs']) / 1e3:.2f},"
                        f"{float(r['MinNs']) / 1e3:.2f},{float(r['MaxNs']) / 1e3:.2f},{r['Percentage']}\n")
    for f in glob.glob(os.path.join(d, "*_under_rocprof.*")):
        open(os.path.join(summ, f"{what}_{os.path.basename(f)}"), "w").write(open(f).read())
    # ---- counters ---------------------------------------------------------------------------------
    means = collections.defaultdict(dict)
    counts = {}
    for f in sorted(glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                means[k][c] = sum(v) / len(v)
                counts[k] = len(v)
    if not means:
        continue
    if what.startswith("gmres_large_"):
        # whole-call traffic of the HBM-bound GMRES leg: every dispatch of the solver's kernels (SpMV, vector sweeps, batched dots, gemv,
        # finalisers) of ONE profiled call, 2 x FETCH_SIZE + WRITE_SIZE summed, divided by the call's inner iterations
        tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
        per_kernel = collections.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "dispatches": 0})
        for f in sorted(glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
            for row in csv.DictReader(open(f)):
                c = row["Counter_Name"]
                k = short(row["Kernel_Name"])
                if c in tot and any(t in k for t in ("spmv", "k_map", "multidot", "gemv", "finalize", "k_mgs", "k_cgs")):
                    tot[c] += float(row["Counter_Value"])
                    per_kernel[tkey(k)][c] += float(row["Counter_Value"])
                    per_kernel[tkey(k)]["dispatches"] += 1 if c == "FETCH_SIZE" else 0
        inner = None
        for lf in sorted(glob.glob(os.path.join(d, "pmc_fetch.log"))):
            for line in open(lf):
                if line.startswith("{"):
                    j = json.loads(line)
                    name = what[len("gmres_large_"):]
                    if name in j:
                        inner = j[name]["inner_iterations"] * j[name].get("calls_timed", 1)
        if inner:
            rd, wr = 2.0 * tot["FETCH_SIZE"] * 1024.0, tot["WRITE_SIZE"] * 1024.0
            json.dump({"_binary": binary_record(), "inner_iterations_profiled": inner, "fetch_bytes_x2": rd, "write_bytes": wr,
                       "traffic_bytes_per_inner_iteration": (rd + wr) / inner,
                       "per_kernel_bytes_per_inner_iteration": {k: {"traffic": (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0 / inner,
                                                                     "dispatches_per_inner_iteration": v["dispatches"] / inner} for k, v in per_kernel.items()}},
                      open(os.path.join(summ, f"{what}_traffic.json"), "w"), indent=1)
    traffic = {}
    with open(os.path.join(summ, f"{what}_pmc_summary.txt"), "w") as o:
        o.write("mean counter value per dispatch (separate rocprofv3 --pmc passes; scripts/prof_r0N.sh)\n")
        for k in sorted(means, key=lambda k: -means[k].get("SQ_WAVE_CYCLES", means[k].get("FETCH_SIZE", 0))):
            m = means[k]
            if not any(s in k for s in ("spmv", "k_map", "k_cg", "multidot", "gemv", "k_mgs", "k_cgs", "finalize", "k_rowdot")):
                continue
            o.write(f"\n{k}   (dispatches sampled: {counts[k]})\n")
            for c in sorted(m):
                o.write(f"    {c:38s} {m[c]:18.1f}\n")
            der = []
            if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
                rd, wr = 2.0 * m["FETCH_SIZE"] * 1024.0, m["WRITE_SIZE"] * 1024.0
                der.append(f"HBM-side traffic per launch: read {rd / 1e9:.4f} GB (2 x FETCH_SIZE) + write {wr / 1e9:.4f} GB = {(rd + wr) / 1e9:.4f} GB")
                kk = tkey(k)
                rec = {"kernel": k, "fetch_bytes_x2": rd, "write_bytes": wr, "traffic_bytes_per_launch": rd + wr}
                # several instantiations share a key (fused dot or not, long rows merged or not): the key holds the one that moves the
                # most (the whole-matrix launch), every instantiation is listed under "variants"
                var = traffic.get(kk, {}).get("variants", {})
                var[k] = rd + wr
                if kk not in traffic or rd + wr > traffic[kk]["traffic_bytes_per_launch"]:
                    traffic[kk] = rec
                traffic[kk]["variants"] = var
            if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
                der.append(f"L2 hit rate {m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']):.3f}")
                if tkey(k) in traffic and traffic[tkey(k)]["kernel"] == k:
                    traffic[tkey(k)]["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
            if "SQ_WAVE_CYCLES" in m and "SQ_WAIT_ANY" in m:
                der.append(f"SQ_WAIT_ANY / SQ_WAVE_CYCLES = {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.3f}; "
                           f"SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES = {m.get('SQ_ACTIVE_INST_ANY', 0) / m['SQ_WAVE_CYCLES']:.3f}")
            if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE"):
                der.append(f"LDS bank-conflict cycles / LDS active cycles = {m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE']:.3f}")
            if "TA_BUSY_avr" in m and "GRBM_GUI_ACTIVE" in m:
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs (value / duration = 8 x the shader clock), TA_BUSY_avr is the mean over the TAs
                cyc = m["GRBM_GUI_ACTIVE"] / 8.0
                der.append(f"TA_BUSY_avr / (GRBM_GUI_ACTIVE / 8 XCDs) = {m['TA_BUSY_avr'] / cyc:.3f}  (a TA counts as busy while requests are outstanding: "
                           f"~0.7 for EVERY HBM-bound kernel here, streaming or gathering)")
                if "TA_ADDR_STALLED_BY_TC_CYCLES_sum" in m:
                    der.append(f"per TA: address path stalled by the cache {m['TA_ADDR_STALLED_BY_TC_CYCLES_sum'] / 256 / cyc:.3f}, "
                               f"data return stalled {m.get('TA_DATA_STALLED_BY_TC_CYCLES_sum', 0) / 256 / cyc:.3f} of the kernel's cycles")
            if "TCP_GATE_EN1_sum" in m and "TCP_PENDING_STALL_CYCLES_sum" in m and m["TCP_GATE_EN1_sum"]:
                der.append(f"TCP pending-stall cycles / TCP active cycles = {m['TCP_PENDING_STALL_CYCLES_sum'] / m['TCP_GATE_EN1_sum']:.3f}")
            if "TCP_TCC_READ_REQ_sum" in m and m.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
                der.append(f"L1 (TCP) -> L2 read requests / TCP cache accesses = {m['TCP_TCC_READ_REQ_sum'] / m['TCP_TOTAL_CACHE_ACCESSES_sum']:.3f}")
            if "SQ_ACTIVE_INST_VALU" in m and "SQ_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
                # SQ_ACTIVE_INST_* count cycles (x4: the counters tick once per 4 clocks on this part? no -- they are summed over
                # the SIMDs of all CUs); normalised here per SIMD: / (1024 SIMDs x kernel cycles)
                cyc = m["GRBM_GUI_ACTIVE"] / 8.0
                per = lambda c: m.get(c, 0.0) / (1024.0 * cyc)
                der.append("issue activity per SIMD (counter / (1024 SIMDs x kernel cycles)): "
                           f"VALU {per('SQ_ACTIVE_INST_VALU'):.3f}, scalar {per('SQ_ACTIVE_INST_SCA'):.3f}, VMEM {per('SQ_ACTIVE_INST_VMEM'):.3f}, "
                           f"LDS {per('SQ_ACTIVE_INST_LDS'):.3f}, misc {per('SQ_ACTIVE_INST_MISC'):.3f}; SALU inst cycles {per('SQ_INST_CYCLES_SALU'):.3f}")
            if "SQ_INST_LEVEL_VMEM" in m and "SQ_LEVEL_WAVES" in m and "GRBM_GUI_ACTIVE" in m:
                cyc = m["GRBM_GUI_ACTIVE"] / 8.0
                der.append(f"mean resident waves per CU {m['SQ_LEVEL_WAVES'] / (256.0 * cyc):.2f}; mean vector-memory instructions in flight per CU "
                           f"{m['SQ_INST_LEVEL_VMEM'] / (256.0 * cyc):.2f} (scalar-memory {m.get('SQ_INST_LEVEL_SMEM', 0) / (256.0 * cyc):.2f}); "
                           f"TA address FIFO full {m.get('SQ_VMEM_TA_ADDR_FIFO_FULL', 0) / (256.0 * cyc):.3f}, TA command FIFO full "
                           f"{m.get('SQ_VMEM_TA_CMD_FIFO_FULL', 0) / (256.0 * cyc):.3f} of the cycles")
            for line in der:
                o.write(f"    => {line}\n")
    if traffic:
        traffic["_binary"] = binary_record([v["kernel"] for v in traffic.values() if isinstance(v, dict) and "kernel" in v])
        json.dump(traffic, open(os.path.join(summ, f"{what}_traffic.json" if not what.startswith("gmres_large_") else f"{what}_traffic_per_launch.json"), "w"), indent=1)
print("summaries:", sorted(os.listdir(summ)))
