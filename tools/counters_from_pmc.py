"""Turn the rocprofv3 PMC passes of tools/collect_all.sh into counters.json:
    python3 tools/counters_from_pmc.py gpurun_out/prof_<tag> <tag>

  lanes256.hbm_bytes_per_step   sum over the wide decoder's dispatches (k_dec_mmvh, k_dec_attn_mm_g, the staging launches) of (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: a 16 B/lane
                                streaming read is counted at half its bytes -- MI355X_MICROARCH.md, HBM), divided by the steps
                                in the trace (= k_dec_attn_mm_g dispatches / (22 blocks x 2 lanes))
  prefill2048.mfma_busy_frac    SQ_VALU_MFMA_BUSY_CYCLES of the prompt GEMM's dispatches / (their GRBM_GUI_ACTIVE / 8 XCDs x 1024
                                SIMDs): the share of the chip's matrix pipes' cycles that issued MFMA work during those kernels
  lanes256_by_request_size      the same dispatches' memory-side read requests by size (TCC_EA0_RDREQ_32B / _64B / _128B, one pass each):
                                32 n32 + 64 n64 + 128 n128 bytes -- the guide's doubling is exact for 16 B/lane streaming reads (all
                                128-byte requests) and "uncalibrated" for narrower ones; this is the calibration, per kernel
Keyed by bench.csrc_fingerprint() like traffic.json.  A group that was not re-collected keeps its section from the counters.json already
in the directory when that file carries the same fingerprint (tools/collect_all.sh <tag> <modes> "2 4")."""
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
    d, tag = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r05")
    sys.path.insert(0, ROOT)
    import bench
    out = {"csrc_sha256_16": bench.csrc_fingerprint(),
           "source": f"tools/collect_all.sh {tag} (rocprofv3 --pmc, one counter per pass, eager launches), profiles/{tag}_counters.txt"}
    old = {}
    try:
        old = json.load(open(os.path.join(d, "counters.json")))
        if old.get("csrc_sha256_16") != out["csrc_sha256_16"]:
            old = {}
    except (OSError, ValueError):
        pass
    text = dict(old.get("_text", {}))                  # section -> its lines of <tag>_counters.txt
    lines = []
    def close(key):
        # a section that was collected now replaces the old one; one that was not keeps what the directory's counters.json held
        nonlocal lines
        if key in out:
            text[key] = lines
        elif key in old:
            out[key] = old[key]
        elif lines:
            text[key] = lines
        lines = []
    # ---- the multi-sequence legs: 8 sequences (GEMV kernels, grouped one-launch attention), 64 (one lane), 256 (two lanes of 128)
    ATTN = ("k_dec_attn_hm<", "k_dec_attn_mm_g<", "k_dec_attn_one_g<", "k_dec_attn_score_g<")       # one per (block, lane) and step
    def wide(name):
        # the multi-sequence decoder's kernels only (the bench's own batch-1 timed region is in the trace too): its W.x kernels, its
        # attention, and k_dec_gemv8's staging launches (EPI_STAGE = 2 / EPI_STAGE_FRAG = 3, the fifth template argument)
        if any(k in name for k in ("k_dec_mmvh<", "k_dec_mmv<", "k_dec_ffn_q4<", "k_dec_gemvm<", "k_dec_silumul_rows", "k_dec_attn_pv_g<") + ATTN):
            return True
        if "k_dec_gemv8<" in name:
            args = name.split("k_dec_gemv8<")[1].split(">")[0].split(",")
            return len(args) >= 5 and args[4].strip() in ("2", "3")
        return False
    for S, lanes in ((8, 1), (64, 1), (256, 2)):
        key = f"lanes{S}"
        try:
            fetch, write = rows_of(os.path.join(d, f"{key}_FETCH_SIZE"), "FETCH_SIZE"), rows_of(os.path.join(d, f"{key}_WRITE_SIZE"), "WRITE_SIZE")
            dec = lambda rows: [r for r in rows if wide(r[1])]
            steps = sum(1 for r in dec(fetch) if any(k in r[1] for k in ATTN)) / (22.0 * lanes)
            fkb, wkb = sum(r[2] for r in dec(fetch)), sum(r[2] for r in dec(write))
            per_kernel = defaultdict(lambda: [0.0, 0.0, 0])
            for _, name, v in dec(fetch):
                k = name.split("(")[0][:70]
                per_kernel[k][0] += v; per_kernel[k][2] += 1
            for _, name, v in dec(write):
                per_kernel[name.split("(")[0][:70]][1] += v
            if steps > 0:
                b = (2 * fkb + wkb) * 1024 / steps
                out[key] = {"hbm_bytes_per_step": int(b), "steps_in_trace": steps, "fetch_kb": fkb, "write_kb": wkb}
                lines.append(f"{key}: {steps:.0f} steps in the trace, FETCH_SIZE {fkb:.0f} KB, WRITE_SIZE {wkb:.0f} KB over the decoder's dispatches -> "
                             f"(2 F + W) * 1024 / steps = {int(b)} B per step")
                for k, (v, w, n) in sorted(per_kernel.items(), key=lambda kv: -kv[1][0])[:12]:
                    lines.append(f"    {k:70s} dispatches {n:6d}  FETCH_SIZE {v / n:10.1f} KB  WRITE_SIZE {w / n:9.1f} KB per dispatch  (2 F + W) = {(2 * v + w) / n / 1024:8.2f} MB")
        except Exception as e:                          # noqa: BLE001
            lines.append(f"{key}: not collected ({e!r})")
        close(key)
    # ---- read requests by size, 256 sequences
    try:
        key = "lanes256_by_request_size"
        n = {sz: rows_of(os.path.join(d, f"lanes256_TCC_EA0_RDREQ_{sz}B_sum"), f"TCC_EA0_RDREQ_{sz}B_sum") for sz in (32, 64, 128)}
        dec = lambda rows: [r for r in rows if wide(r[1])]
        steps = sum(1 for r in dec(n[128]) if any(k in r[1] for k in ATTN)) / 44.0
        per_kernel = defaultdict(lambda: {32: 0.0, 64: 0.0, 128: 0.0, "n": 0})
        for sz in (32, 64, 128):
            for _, name, v in dec(n[sz]):
                k = name.split("(")[0][:70]
                per_kernel[k][sz] += v
                if sz == 128:
                    per_kernel[k]["n"] += 1
        if steps > 0:
            tot = {sz: sum(r[2] for r in dec(n[sz])) for sz in (32, 64, 128)}
            rb = 32 * tot[32] + 64 * tot[64] + 128 * tot[128]
            wb = out.get("lanes256", old.get("lanes256", {})).get("write_kb", 0.0) * 1024 / max(out.get("lanes256", old.get("lanes256", {})).get("steps_in_trace", 1), 1)
            out[key] = {"read_bytes_per_step": int(rb / steps), "requests_32B": tot[32] / steps, "requests_64B": tot[64] / steps, "requests_128B": tot[128] / steps,
                        "steps_in_trace": steps, "hbm_bytes_per_step": int(rb / steps + wb)}
            lines.append(f"{key}: {steps:.0f} steps in the trace; per step {tot[32] / steps:.0f} x 32 B + {tot[64] / steps:.0f} x 64 B + {tot[128] / steps:.0f} x 128 B "
                         f"= {int(rb / steps)} B read; + WRITE_SIZE of the lanes256 passes = {int(rb / steps + wb)} B per step")
            for k, c in sorted(per_kernel.items(), key=lambda kv: -(32 * kv[1][32] + 64 * kv[1][64] + 128 * kv[1][128]))[:12]:
                m = max(c["n"], 1)
                lines.append(f"    {k:70s} dispatches {c['n']:6d}  per dispatch: 32 B x {c[32] / m:9.0f}  64 B x {c[64] / m:9.0f}  128 B x {c[128] / m:9.0f}  = {(32 * c[32] + 64 * c[64] + 128 * c[128]) / m / 1e6:8.2f} MB read")
    except Exception as e:                          # noqa: BLE001
        lines.append(f"lanes256_by_request_size: not collected ({e!r})")
    close("lanes256_by_request_size")
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
    close("prefill2048")
    out["_text"] = text
    json.dump(out, open(os.path.join(d, "counters.json"), "w"), indent=1)
    body = [f"# tools/collect_all.sh {tag}; kernel sources sha256/16 = {out['csrc_sha256_16']}"]
    for key in ("lanes8", "lanes64", "lanes256", "lanes256_by_request_size", "prefill2048"):
        body += text.get(key, [])
    open(os.path.join(d, f"{tag}_counters.txt"), "w").write("\n".join(body) + "\n")
    print("\n".join(body))


if __name__ == "__main__":
    main()
