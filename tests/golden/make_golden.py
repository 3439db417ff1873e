#!/usr/bin/env python3
"""Generate the committed golden fixtures from the REAL reference.

Runs only in the build container (needs /root/reference and oracle/_ref built
from it); the GPU box and CI only ever read the .npz files written here.
Fixtures are data (inputs and expected outputs); no reference source text is
stored.

  converter_pins.npz   bytes emitted by the reference converter's q8_quantize /
                       q4_quantize / fp16 cast (tinyllama_to_gten.py:24-148) for
                       small f32 tensors incl. an all-zero block and exact .5 ties
  ops_golden.npz       every operator of gten/ops.h at small shapes, 3 dtype modes,
                       outputs of the reference's AVX build and of its scalar build
  tiny_model_golden.npz full logits, prefill + decode, of a 2-layer model assembled
                       from the reference's own modules (weights: this repo's
                       seeded generator, so only the seed is stored)
  full_model_golden.npz TinyLlama-1.1B (the reference's own TinyLlama class) on seeded
                       synthetic weights: per step top-8 (id, logit), mean/std, probe
                       logits, greedy ids; f16 / q8 / q4; plus one long-context probe

usage: python tests/golden/make_golden.py [--skip-full] [--skip-long]
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", "8")

from __graft_entry__ import load_package  # noqa: E402
from helpers import F16, F32, MODES, Q4, Q8, act_rows, rng, row_bytes, tiny_config, weight_rows  # noqa: E402
from oracle import orc  # noqa: E402

REFERENCE = "/root/reference"
PROBE_IDS = np.array([0, 1, 2, 13, 1000, 15000, 29871, 31999, 32000, 32002], np.int32)


def converter_functions():
    """q8_quantize / q4_quantize of the reference converter.  The script runs argparse
    and a full conversion at import, so only its function definitions are executed."""
    src = open(os.path.join(REFERENCE, "tinyllama_to_gten.py")).read()
    head = src[: src.index("def convert_model_to_gten")]
    ns = {}
    exec(compile(head, "tinyllama_to_gten.py[defs]", "exec"), ns)
    return ns["q8_quantize"], ns["q4_quantize"]


def make_converter_pins():
    import torch
    q8q, q4q = converter_functions()
    r = rng(20240)
    w = (0.02 * r.standard_normal((8, 256))).astype(np.float32)
    w[1, 32:64] = 0.0                                  # all-zero block: delta 0
    w[2, :32] = np.arange(32, dtype=np.float32) - 15.5  # absmax 16.5
    w[3, :32] = 0.0
    w[3, 0] = 127.0; w[3, 1] = 0.5; w[3, 2] = 1.5; w[3, 3] = 2.5; w[3, 4] = -0.5; w[3, 5] = -2.5   # Q8 half-even ties
    w[4, :32] = 0.0
    w[4, 0] = 7.0; w[4, 1] = 0.5; w[4, 2] = 1.5; w[4, 3] = 2.5; w[4, 16] = -3.5; w[4, 17] = -7.0     # Q4 half-even ties
    t = torch.from_numpy(w.copy())
    d8, q8 = q8q(t.clone())
    d4, q4 = q4q(t.clone())
    blocks8 = np.concatenate([d8.numpy().view(np.uint8).reshape(-1, 2), q8.numpy().view(np.uint8).reshape(-1, 32)], axis=1)
    blocks4 = np.concatenate([d4.numpy().view(np.uint8).reshape(-1, 2), q4.numpy().reshape(-1, 16)], axis=1)
    f16 = t.to(torch.float16).numpy().view(np.uint8)
    out = dict(w=w, q8=blocks8.reshape(8, -1), q4=blocks4.reshape(8, -1), f16=f16.reshape(8, -1))
    # checkpoints stored in bf16 (TinyLlama's own) or f16: the reference divides absmax in the CHECKPOINT's dtype before
    # it goes to f32 (tinyllama_to_gten.py:38-39,68-69), so the deltas -- and with them the quants -- differ from the
    # f32 path.  The inputs below are `w` rounded to that dtype (stored here as the f32 values they hold).
    for tag, dt in (("bf16", torch.bfloat16), ("f16src", torch.float16)):
        ts = t.to(dt)
        d8, q8 = q8q(ts.clone())
        d4, q4 = q4q(ts.clone())
        out[f"w_{tag}"] = ts.to(torch.float32).numpy()
        out[f"q8_{tag}"] = np.concatenate([d8.numpy().view(np.uint8).reshape(-1, 2), q8.numpy().view(np.uint8).reshape(-1, 32)], axis=1).reshape(8, -1)
        out[f"q4_{tag}"] = np.concatenate([d4.numpy().view(np.uint8).reshape(-1, 2), q4.numpy().reshape(-1, 16)], axis=1).reshape(8, -1)
        out[f"f16_{tag}"] = ts.to(torch.float16).numpy().view(np.uint8).reshape(8, -1)
    np.savez_compressed(os.path.join(HERE, "converter_pins.npz"), **out)
    print("converter_pins.npz", blocks8.shape, blocks4.shape)


def make_ops_golden(oracle, refs):
    out = {}
    for name, wd, ad in MODES():
        r = rng(hash(name) % 1000 + 7)
        # matmul n=3, d_in=256, d_out=96, start_pos 0 and 2; outputs in adtype and f32
        x, _ = act_rows(oracle, r, 3, 256, ad)
        w, _ = weight_rows(oracle, r, 96, 256, wd)
        out[f"{name}.matmul.x"] = x
        out[f"{name}.matmul.w"] = w
        for kind, ref in refs.items():
            for sp in (0, 2):
                for od, odn in ((ad, "a"), (F32, "f32")):
                    o = np.zeros((3, row_bytes(od, 96)), np.uint8)
                    ref.matmul_2d(x, ad, w, wd, o, od, 3, 256, 96, sp)
                    out[f"{name}.matmul.out.{kind}.sp{sp}.{odn}"] = o
        # token_embed
        tab, _ = weight_rows(oracle, r, 50, 256, wd)
        toks = np.array([3, 49, 0, 7, 7], np.int32)
        out[f"{name}.embed.table"] = tab
        out[f"{name}.embed.tokens"] = toks
        o = np.zeros((5, row_bytes(ad, 256)), np.uint8)
        refs["avx"].token_embed(tab, wd, toks, o, ad, 256, 1)
        out[f"{name}.embed.out"] = o
    for ad, an in ((F16, "f16"), (Q8, "q8")):
        r = rng(31 + ad)
        n, d = 4, 256
        x, _ = act_rows(oracle, r, n, d, ad)
        y, _ = act_rows(oracle, r, n, d, ad)
        wn = (1 + 0.05 * r.standard_normal(d)).astype(np.float16)
        out[f"{an}.row.x"] = x; out[f"{an}.row.y"] = y; out[f"{an}.row.w"] = wn
        ref = refs["avx"]                       # these ops have no SIMD-order dependence
        for sp in (0, 3):
            o = np.zeros_like(x); ref.rms_norm(x, ad, wn, o, n, d, sp); out[f"{an}.rms_norm.sp{sp}"] = o
            a = x.copy(); ref.rotary_emb(a, ad, n, d, 64, sp); out[f"{an}.rope.sp{sp}"] = a
            o = np.zeros_like(x); ref.silu(x, o, ad, n, d, sp); out[f"{an}.silu.sp{sp}"] = o
            o = np.zeros_like(x); ref.mul(x, y, o, ad, n, d, sp); out[f"{an}.mul.sp{sp}"] = o
            o = np.zeros_like(x); ref.add(x, y, o, ad, n, d, sp); out[f"{an}.add.sp{sp}"] = o
        # rope at the last positions (angles up to 2047 rad)
        xr, _ = act_rows(oracle, r, 2048, 128, ad)
        a = xr.copy(); ref.rotary_emb(a, ad, 2048, 128, 64, 2044)
        out[f"{an}.rope_far.x"] = xr[2044:]; out[f"{an}.rope_far.out"] = a[2044:]
        # attention: prefill and decode, 8 heads / 2 groups, d_head 64
        for n_att, sp in ((5, 0), (33, 0), (40, 0), (40, 39)):
            q, _ = act_rows(oracle, r, n_att, 512, ad)
            k, _ = act_rows(oracle, r, n_att, 128, ad)
            v, _ = act_rows(oracle, r, n_att, 128, ad)
            key = f"{an}.attn.n{n_att}.sp{sp}"
            out[key + ".q"] = q; out[key + ".k"] = k; out[key + ".v"] = v
            for kind, rf in refs.items():
                o = np.zeros((n_att, row_bytes(ad, 512)), np.uint8)
                rf.qkv_attn(q, k, v, o, ad, n_att, 8, 2, 64, sp)
                out[key + f".out.{kind}"] = o
    np.savez_compressed(os.path.join(HERE, "ops_golden.npz"), **out)
    print("ops_golden.npz", len(out), "arrays")


def host_cfg(pkg, c):
    return pkg.HostConfig(**{k: getattr(c, k) for k, _ in c._fields_})


def make_tiny_model_golden(pkg, host, refs):
    out = {}
    for name, wd, ad in MODES():
        ocfg = tiny_config(wd, ad, n_heads=4, n_kv_heads=2)
        cfg = host_cfg(pkg, ocfg)
        toks = list(host.synthetic_tokens(9, seed=7, n_vocab=cfg.n_vocab))
        models = {k: r.model(ocfg) for k, r in refs.items()}
        for i in range(models["avx"].n_weights()):
            w = host.synth_weight(cfg, 4321, i)
            for m in models.values():
                m.set_weight(i, w)
        logits = {k: [] for k in refs}
        for step in range(8):
            sp = 0 if step == 0 else len(toks) - 1
            for k, m in models.items():
                logits[k].append(m.logits(toks, sp))
            toks.append(int(np.argmax(logits["avx"][-1])))
        out[f"{name}.tokens"] = np.array(toks, np.int32)
        for k in refs:
            out[f"{name}.logits.{k}"] = np.stack(logits[k])
        for m in models.values():
            m.close()
    out["seed"] = np.array([4321]); out["token_seed"] = np.array([7])
    np.savez_compressed(os.path.join(HERE, "tiny_model_golden.npz"), **out)
    print("tiny_model_golden.npz")


def summarize(lg):
    """top-8 (id, logit), mean/std, and every 32nd logit (1001 values: enough samples for a stable rms)"""
    top = np.argsort(-lg, kind="stable")[:8].astype(np.int32)
    probes = np.concatenate([lg[PROBE_IDS], lg[::32]]).astype(np.float32)
    return top, lg[top].astype(np.float32), np.array([lg.mean(), lg.std()], np.float32), probes


def add_long_probe(host, refs, out):
    """long-context probe (q4): teacher-forced ids, single-token steps from n=1 (single-row decode never
    triggers the reference's probability-scratch stride quirk).  Both reference builds, so that the
    fixture carries the reference's OWN spread at long context next to the expected values."""
    toks = host.synthetic_tokens(2048, seed=12345)
    ns = (257, 1024, 2047, 2048)
    for kind in ("avx", "scalar"):
        m = refs[kind].tinyllama(2048, Q4, Q8)
        m.load("/tmp/gten_golden_q4.gten")
        t0 = time.time()
        for n in range(1, 2049):
            lg = m.logits(toks[:n], n - 1)
            if n in ns:
                t, v, s, p = summarize(lg)
                suffix = "" if kind == "avx" else ".scalar"
                out[f"long.q4.n{n}.top_ids{suffix}"] = t; out[f"long.q4.n{n}.top_logits{suffix}"] = v
                out[f"long.q4.n{n}.stats{suffix}"] = s; out[f"long.q4.n{n}.probes{suffix}"] = p
            if n % 256 == 0:
                print(f"  long probe {kind} n={n} {time.time() - t0:.0f}s", flush=True)
        m.close()
    out["long.q4.ns"] = np.array(ns, np.int32)


def make_full_extra_golden(host, refs):
    """full_extra_golden.npz (round 2): what full_model_golden.npz does not hold --
      * PROMPT PROCESSING at full size: a 96-id prompt (>= 16 rows: the matrix-core W.x of gten_mfma.hip and the
        tiled attention at 2048 / 5632 widths) and three teacher-forced decode steps after it, f16 / q8 / q4, the
        reference's AVX build and (the yardstick: its own spread) its scalar build; max_ctx 256 >= 2 n (stride quirk);
      * long-context probes for f16 and q8 like long.q4 (single-token steps from n = 1 to 2048, both builds)."""
    probe_ids = np.concatenate([PROBE_IDS, np.arange(0, 32003, 32, dtype=np.int32)])
    out = {"probe_ids": probe_ids, "seed": np.array([1234]), "token_seed": np.array([12345]), "prompt_seed": np.array([4242])}
    P = 96
    prompt = list(host.synthetic_tokens(P, seed=4242))
    for name, wd, ad in MODES():
        cfg = host.default_config(wd, ad)
        path = f"/tmp/gten_golden_{name}.gten"
        if not os.path.exists(path):
            host.write_gten(cfg, 1234, path)
        for kind in ("avx", "scalar"):
            t0 = time.time()
            m = refs[kind].tinyllama(256, wd, ad)
            m.load(path)
            toks = list(prompt)
            forced = out.get(f"prefill.{name}.avx.tokens") if kind == "scalar" else None
            tops, vals, stats, probes = [], [], [], []
            for step in range(4):
                lg = m.logits(toks, 0 if step == 0 else len(toks) - 1)
                t, v, s, p = summarize(lg)
                tops.append(t); vals.append(v); stats.append(s); probes.append(p)
                toks.append(int(forced[len(toks)]) if forced is not None else int(t[0]))
            m.close()
            out[f"prefill.{name}.{kind}.tokens"] = np.array(toks, np.int32)
            out[f"prefill.{name}.{kind}.top_ids"] = np.stack(tops)
            out[f"prefill.{name}.{kind}.top_logits"] = np.stack(vals)
            out[f"prefill.{name}.{kind}.stats"] = np.stack(stats)
            out[f"prefill.{name}.{kind}.probes"] = np.stack(probes)
            print(f"prefill {name}/{kind}: {time.time() - t0:.0f}s", flush=True)
        np.savez_compressed(os.path.join(HERE, "full_extra_golden.npz"), **out)
    toks = host.synthetic_tokens(2048, seed=12345)
    ns = (257, 1024, 2047, 2048)
    for name, wd, ad in MODES()[:2]:                # f16, q8 (q4: full_model_golden.npz long.q4)
        for kind in ("avx", "scalar"):
            m = refs[kind].tinyllama(2048, wd, ad)
            m.load(f"/tmp/gten_golden_{name}.gten")
            t0 = time.time()
            for n in range(1, 2049):
                lg = m.logits(toks[:n], n - 1)
                if n in ns:
                    t, v, s, p = summarize(lg)
                    suffix = "" if kind == "avx" else ".scalar"
                    out[f"long.{name}.n{n}.top_ids{suffix}"] = t; out[f"long.{name}.n{n}.top_logits{suffix}"] = v
                    out[f"long.{name}.n{n}.stats{suffix}"] = s; out[f"long.{name}.n{n}.probes{suffix}"] = p
                if n % 256 == 0:
                    print(f"  long probe {name}/{kind} n={n} {time.time() - t0:.0f}s", flush=True)
            m.close()
        out[f"long.{name}.ns"] = np.array(ns, np.int32)
        np.savez_compressed(os.path.join(HERE, "full_extra_golden.npz"), **out)
    print("full_extra_golden.npz")


def make_full_model_golden(pkg, host, refs, skip_long):
    probe_ids = np.concatenate([PROBE_IDS, np.arange(0, 32003, 32, dtype=np.int32)])
    out = {"probe_ids": probe_ids, "seed": np.array([1234]), "token_seed": np.array([12345])}
    n_steps = 24
    for name, wd, ad in MODES():
        cfg = host.default_config(wd, ad)
        path = f"/tmp/gten_golden_{name}.gten"
        t0 = time.time()
        if not os.path.exists(path):
            host.write_gten(cfg, 1234, path)
        prompt = list(host.synthetic_tokens(15, seed=12345))
        for kind, ref in refs.items():
            if kind == "scalar" and name != "q4":
                continue                       # scalar build: q4 only (time)
            m = ref.tinyllama(256, wd, ad)      # the reference's own TinyLlama class; max_ctx >= 2*n_prompt (stride quirk)
            m.load(path)
            toks = list(prompt)
            # the scalar build is teacher-forced with the AVX build's greedy ids, so that the
            # two runs see identical inputs and their difference is the reference's own spread
            forced = out.get(f"{name}.avx.tokens") if kind == "scalar" else None
            tops, vals, stats, probes = [], [], [], []
            for step in range(n_steps):
                sp = 0 if step == 0 else len(toks) - 1
                lg = m.logits(toks, sp)
                t, v, s, p = summarize(lg)
                tops.append(t); vals.append(v); stats.append(s); probes.append(p)
                toks.append(int(forced[len(toks)]) if forced is not None else int(t[0]))
            m.close()
            out[f"{name}.{kind}.tokens"] = np.array(toks, np.int32)
            out[f"{name}.{kind}.top_ids"] = np.stack(tops)
            out[f"{name}.{kind}.top_logits"] = np.stack(vals)
            out[f"{name}.{kind}.stats"] = np.stack(stats)
            out[f"{name}.{kind}.probes"] = np.stack(probes)
            print(f"full {name}/{kind}: {time.time() - t0:.0f}s, greedy {toks[15:23]}")
    if not skip_long:
        add_long_probe(host, refs, out)
    np.savez_compressed(os.path.join(HERE, "full_model_golden.npz"), **out)
    print("full_model_golden.npz")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-full", action="store_true")
    ap.add_argument("--skip-long", action="store_true")
    ap.add_argument("--only-full", action="store_true")
    ap.add_argument("--only-long", action="store_true", help="re-run just the long-context probe into the existing fixture")
    ap.add_argument("--only-pins", action="store_true", help="just converter_pins.npz")
    ap.add_argument("--only-extra", action="store_true", help="just full_extra_golden.npz (full-size prompt processing; f16 / q8 long-context probes)")
    args = ap.parse_args()
    assert os.path.isdir(REFERENCE), "run this in the build container (needs /root/reference)"
    if args.only_pins:
        make_converter_pins()
        return
    orc.build(ref=True)
    oracle = orc.load_oracle()
    refs = {"avx": orc.load_ref("avx"), "scalar": orc.load_ref("scalar")}
    pkg = load_package()
    pkg.build.build_all()
    host = pkg.load_host()
    if args.only_extra:
        make_full_extra_golden(host, refs)
        return
    if args.only_long:
        path = os.path.join(HERE, "full_model_golden.npz")
        out = dict(np.load(path))
        if not os.path.exists("/tmp/gten_golden_q4.gten"):
            host.write_gten(host.default_config(Q4, Q8), 1234, "/tmp/gten_golden_q4.gten")
        add_long_probe(host, refs, out)
        np.savez_compressed(path, **out)
        return
    if not args.only_full:
        make_converter_pins()
        make_ops_golden(oracle, refs)
        make_tiny_model_golden(pkg, host, refs)
    if not args.skip_full:
        make_full_model_golden(pkg, host, refs, args.skip_long)


if __name__ == "__main__":
    main()
