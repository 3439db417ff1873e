#!/usr/bin/env python3
"""rocprofv3 PMC passes of the decode step -> HBM bytes per launch per kernel family.

    python3 tools/traffic_from_pmc.py gpurun_out/prof_<tag> <tag>

Reads <dir>/pmc_FETCH_SIZE/**/*_counter_collection.csv and <dir>/pmc_WRITE_SIZE/... (separate passes, as the MI355X guide
prescribes), keeps the dispatches of the LAST decode steps (context 2039..2048: the eager --no-graph bench run ends with
them), averages per kernel family and applies the gfx950 correction of MI355X_MICROARCH.md "HBM": FETCH_SIZE (KB) counts
a wide coalesced streaming read at half its bytes -> bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
Writes <dir>/traffic.json (+ a text table): {"csrc_sha256_16": ..., "q4": {family: bytes per launch}, "source": ...};
copy it to profiles/traffic.json (bench.py reports `roofline.traffic` from it only while the hash matches the kernel
sources it runs) and the table to profiles/<tag>_pmc_q4.txt."""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernel name fragments -> family (the names of bench.py's roofline / gten_hip_prof_family_name)
def family(name):
    if name.startswith("void k_dec_gemv8<"):
        args = name[len("void k_dec_gemv8<"):].split(">")[0].split(",")
        wt, pro, nch, r, epi = (int(a) for a in args[:5])
        if epi == 1:
            return "decode_gemv_gateup"
        if pro == 4:
            return "decode_gemv_down"
        if pro in (2, 3):
            return "decode_gemv_o"
        if pro == 0:
            return "decode_gemv_qkv"
        return "decode_gemv_head" if r >= 4 else "decode_gemv_qkv"
    if name.startswith("void k_dec_attn_one64<"):
        return "decode_attn_score"
    if name.startswith("k_dec_argmax"):
        return "decode_argmax"
    return None


def csrc_fingerprint():
    """the same hash bench.py checks (kernel sources + compiler flags): one definition, bench.py's"""
    sys.path.insert(0, ROOT)
    import bench
    return bench.csrc_fingerprint()


def read_counter(dirname, mode, counter, last_steps=8):
    sub = os.path.join(dirname, f"pmc_{mode}_{counter}")
    if not os.path.isdir(sub):
        sub = os.path.join(dirname, "pmc_" + counter)            # (round-2 layout: q4 only)
    files = glob.glob(os.path.join(sub, "**", "*_counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {sub}")
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    # the decode steps are the dispatches that end with k_dec_argmax; keep the last `last_steps` of them
    ends = [i for i, r in enumerate(rows) if r[1].startswith("k_dec_argmax")]
    if len(ends) > last_steps:
        rows = rows[ends[-last_steps - 1] + 1: ends[-1] + 1]
    agg = defaultdict(list)
    for _, name, v in rows:
        fam = family(name)
        if fam:
            agg[fam].append(v)
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    d, tag = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r05")
    modes = sys.argv[3:] or ["q4"]
    out = {"csrc_sha256_16": csrc_fingerprint(),
           "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/collect_all.sh, profiles/{tag}_pmc_<mode>.txt): "
                     "eager decode steps at n = 2041..2048, bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: a 16 B/lane "
                     "streaming read is counted at half its bytes)"}
    for mode in modes:
        fetch, write = read_counter(d, mode, "FETCH_SIZE"), read_counter(d, mode, "WRITE_SIZE")
        out[mode] = {}
        lines = [f"# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), python3 bench.py --brief --mode {mode} --fill prefill --no-graph --steps 8 --warmup 2,",
                 "# decode launches of the last 8 steps (n = 2041..2048); per-launch means; gfx950 correction: FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM)",
                 f"# kernel sources sha256/16 = {out['csrc_sha256_16']}"]
        for fam in sorted(fetch):
            f_kb, n = fetch[fam]
            w_kb = write.get(fam, (0.0, 0))[0]
            b = int((2 * f_kb + w_kb) * 1024)
            out[mode][fam] = b
            lines.append(f"{fam:22s} launches {n:5d}  FETCH_SIZE {f_kb:10.1f} KB  WRITE_SIZE {w_kb:8.1f} KB  HBM bytes (2*F+W)*1024 = {b}")
        open(os.path.join(d, f"{tag}_pmc_{mode}.txt"), "w").write("\n".join(lines) + "\n")
        print("\n".join(lines))
    json.dump(out, open(os.path.join(d, "traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
