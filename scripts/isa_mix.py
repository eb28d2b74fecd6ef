"""Instruction mix of the SpMV kernels as hipcc emits them for gfx950 (no GPU needed): compiles csrc/mik_core.hip to
device assembly and counts, per kernel body, the vector-memory / LDS / VALU / SALU / wait instructions -- the evidence behind
"what the kernel issues" in DESIGN.md (the dynamic counts are in profiles/r03_bench_pmc_summary.txt, r03_s27_pmc_summary.txt: SQ_INSTS_*).

    python scripts/isa_mix.py > profiles/r03_isa_mix.txt
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "iterativesolvers.jl_amd", "csrc", "mik_core.hip")
WANT = {"k_spmv_sdiab2<double, true, true, 7, 3>  (two rows per lane; both paths)": "_Z13k_spmv_sdiab2IdLb1ELb1ELi7ELi3EE",
        "k_spmv_sdiab<double, true, true, 2, 7, 3>  (both paths: compiled-in class and slot by slot)": "_Z12k_spmv_sdiabIdLb1ELb1ELi2ELi7ELi3EE",
        "k_spmv_sdiac<double, true, true, 2>": "_Z12k_spmv_sdiacIdLb1ELb1ELi2EE", "k_spmv_sdia<double, true, true>": "_Z11k_spmv_sdiaIdLb1ELb1EE", "k_spmv_rowgather<double, true, true>": "_Z16k_spmv_rowgatherIdLb1ELb1EE",
        "k_spmv_rowblock<double, true, true, true, false>": "_Z15k_spmv_rowblockIdLb1ELb1ELb1ELb0EE",
        "k_spmv_sdiaw2<double, true, true>  (wide slice-constant layout, two rows per lane; all paths)": "_Z13k_spmv_sdiaw2IdLb1ELb1EE",
        "k_spmv_jds<float, false, true, false>  (jagged slices)": "_Z10k_spmv_jdsIfLb0ELb1ELb0EE"}
with tempfile.TemporaryDirectory() as tmp:
    asm = os.path.join(tmp, "core.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S",
                           SRC, "-o", asm])
    text = open(asm).read()
print("static instruction mix per kernel (hipcc -O3 --offload-arch=gfx950 -ffp-contract=off; one row per opcode class)\n")
for name, mangled in WANT.items():
    m = re.search(r"^(%s\w*):[^\n]*\n(.*?)^\.Lfunc_end" % re.escape(mangled), text, re.S | re.M)      # the whole body (early exits included)
    if not m:
        print(name, ": not found")
        continue
    body = m.group(2)
    ops = collections.Counter()
    for line in body.split("\n"):
        line = line.strip()
        if not line or line.startswith((";", ".")) or line.endswith(":"):
            continue
        ops[line.split()[0]] += 1
    cls = collections.Counter()
    for op, c in ops.items():
        key = ("global_load_lds (LDS-DMA)" if op.startswith("global_load_lds") else
               op if op.startswith(("global_load", "global_store", "buffer_load", "buffer_store", "ds_read", "ds_write", "s_waitcnt", "s_barrier")) else
               "v_* (VALU)" if op.startswith("v_") else "s_load (SMEM)" if op.startswith("s_load") else "s_* (SALU / branch)" if op.startswith("s_") else op)
        cls[key] += c
    res = re.search(r"; NumVgprs: (\d+).*?; ScratchSize: (\d+).*?; Occupancy: (\d+).*?; LDSByteSize: (\d+)", text[m.end():m.end() + 8000], re.S)
    print(f"{name}   ({sum(ops.values())} instructions" + (f"; VGPRs {res.group(1)}, scratch {res.group(2)}, occupancy {res.group(3)} waves/SIMD, LDS {res.group(4)} B" if res else "") + ")")
    for k, c in sorted(cls.items(), key=lambda kv: (not kv[0].startswith(("global", "buffer", "ds_")), kv[0])):
        print(f"    {k:34s} {c:5d}")
    nt = len(re.findall(r"(?:global|buffer)_load\w* .* nt\b", body))
    print(f"    (loads carrying the nt hint: {nt}; stores carrying nt: {len(re.findall(r'(?:global|buffer)_store.* nt', body))})\n")
