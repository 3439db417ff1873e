"""Turn the rocprofv3 PMC passes of tools/collect_counters.sh into counters.json:
    python3 tools/counters_from_pmc.py gpurun_out/counters_<tag> <tag>

  lanes256.hbm_bytes_per_step   sum over the wide decoder's dispatches (k_dec_mmvh, k_dec_attn_mm_g, the staging launches) of (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: a 16 B/lane
                                streaming read is counted at half its bytes -- MI355X_MICROARCH.md, HBM), divided by the steps
                                in the trace (= k_dec_attn_mm_g dispatches / (22 blocks x 2 lanes))
  prefill2048.mfma_busy_frac    SQ_VALU_MFMA_BUSY_CYCLES of the prompt GEMM's dispatches / (their GRBM_GUI_ACTIVE / 8 XCDs x 1024
                                SIMDs): the share of the chip's matrix pipes' cycles that issued MFMA work during those kernels
Keyed by bench.csrc_fingerprint() like traffic.json."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rows_of(d, counter):
    out = []
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                out.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    out.sort()
    return out


def main():
    d, tag = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r04")
    sys.path.insert(0, ROOT)
    import bench
    out = {"csrc_sha256_16": bench.csrc_fingerprint(),
           "source": f"tools/collect_counters.sh {tag} (rocprofv3 --pmc, one counter per pass, eager launches), profiles/{tag}_counters.txt"}
    lines = [f"# tools/collect_counters.sh {tag}; kernel sources sha256/16 = {out['csrc_sha256_16']}"]
    # ---- the 256-sequence step
    try:
        fetch, write = rows_of(os.path.join(d, "lanes_FETCH_SIZE"), "FETCH_SIZE"), rows_of(os.path.join(d, "lanes_WRITE_SIZE"), "WRITE_SIZE")
        # the WIDE decoder's kernels only (the bench's own batch-1 timed region is in the trace too): the matrix-core W.x, the
        # grouped attention, and k_dec_gemv8's staging launches (EPI_STAGE_FRAG = 3, its fifth template argument)
        def wide(name):
            if "k_dec_mmvh<" in name or "k_dec_attn_mm_g<" in name:
                return True
            if "k_dec_gemv8<" in name:
                args = name.split("k_dec_gemv8<")[1].split(">")[0].split(",")
                return len(args) >= 5 and args[4].strip() == "3"
            return False
        dec = lambda rows: [r for r in rows if wide(r[1])]
        # 22 blocks x 2 lanes of attention launches per 256-sequence step
        steps = sum(1 for r in dec(fetch) if "k_dec_attn_mm_g<" in r[1]) / 44.0
        fkb, wkb = sum(r[2] for r in dec(fetch)), sum(r[2] for r in dec(write))
        per_kernel = defaultdict(lambda: [0.0, 0])
        for _, name, v in dec(fetch):
            k = name.split("(")[0][:70]
            per_kernel[k][0] += v; per_kernel[k][1] += 1
        if steps > 0:
            b = (2 * fkb + wkb) * 1024 / steps
            out["lanes256"] = {"hbm_bytes_per_step": int(b), "steps_in_trace": steps, "fetch_kb": fkb, "write_kb": wkb}
            lines.append(f"lanes256: {steps:.0f} steps in the trace, FETCH_SIZE {fkb:.0f} KB, WRITE_SIZE {wkb:.0f} KB over the wide decoder's dispatches -> "
                         f"(2 F + W) * 1024 / steps = {int(b)} B per step")
            for k, (v, n) in sorted(per_kernel.items(), key=lambda kv: -kv[1][0])[:12]:
                lines.append(f"    {k:70s} dispatches {n:6d}  FETCH_SIZE {v / n:10.1f} KB per dispatch")
    except Exception as e:                          # noqa: BLE001
        lines.append(f"lanes256: not collected ({e!r})")
    # ---- the prompt GEMM
    try:
        busy, act = rows_of(os.path.join(d, "prefill_SQ_VALU_MFMA_BUSY_CYCLES"), "SQ_VALU_MFMA_BUSY_CYCLES"), rows_of(os.path.join(d, "prefill_GRBM_GUI_ACTIVE"), "GRBM_GUI_ACTIVE")
        res = {}
        for key in ("k_matmul_mfma", "k_attn_tiled"):
            bsum = sum(r[2] for r in busy if key in r[1]); asum = sum(r[2] for r in act if key in r[1])
            nb, na = sum(1 for r in busy if key in r[1]), sum(1 for r in act if key in r[1])
            if nb and na and asum > 0:
                # the two passes hold the same dispatches (same command): per-dispatch means
                frac = (bsum / nb) / ((asum / na) / 8.0 * 1024.0)
                res[key] = {"mfma_busy_cycles_per_dispatch": bsum / nb, "grbm_gui_active_per_dispatch": asum / na, "mfma_busy_frac": round(frac, 4), "dispatches": nb}
                lines.append(f"prefill2048 {key}: SQ_VALU_MFMA_BUSY_CYCLES {bsum / nb:.0f} / (GRBM_GUI_ACTIVE {asum / na:.0f} / 8 x 1024) = {frac:.4f} over {nb} dispatches")
        if res:
            out["prefill2048"] = res
    except Exception as e:                          # noqa: BLE001
        lines.append(f"prefill2048: not collected ({e!r})")
    json.dump(out, open(os.path.join(d, "counters.json"), "w"), indent=1)
    open(os.path.join(d, f"{tag}_counters.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
