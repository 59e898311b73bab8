#!/usr/bin/env python3
"""Headline benchmark of the MI355X decode path: audio-codec frames/s/GPU for Dia-1.6B (+ RTF).

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run)

A "step" is one decode step of the whole batch (one frame per utterance = 9 codebook tokens).
Workload (BASELINE.json):
  N = 1   configs[1] — Dia-1.6B shapes, bf16 weights + bf16 K/V, batch 1, 1024 decode steps, README prompt.
          The same run also measures, at 1024 steps each whatever --steps says, every other single-GPU
          configuration of BASELINE.json (`configs` object of the JSON line): batch 1 with fp32 K/V (the
          configuration that meets the 1e-3 logit parity bound; also as two bf16 planes per value through the MFMA attention
          kernel, "bf16x2"), batch 8 mixed text lengths 32..512 (configs[2]),
          the 50 %-structured-pruned, compacted checkpoint at batch 8 and batch 1 (configs[3]), and all 64
          utterances of configs[4] on this one GPU (512 steps): the N = 1 point of the multi-GPU curve.
  N > 1   configs[4] — 64 utterances in all, 64/N per GPU (mixed lengths), weights broadcast once from rank 0.
Synthetic seeded weights (no checkpoint exists offline).  Everything is resident in HBM before the timed
region; the loop replays one hipGraph per step and never syncs with the host inside the region.
Prints ONE JSON line (rank 0).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "dia-tts-prune_amd"))
sys.path.insert(0, ROOT)

import numpy as np
import torch

PROMPT = "[S1] Dia is an open weights text to dialogue model. [S2] You get full control over scripts and voices."
MIXED_L = [32, 64, 96, 128, 192, 256, 384, 512]      # batch-8 mixed lengths, sum 1664
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
MFMA_PEAK_TFLOPS = 2500.0      # dense bf16, same guide
FRAME_RATE = 44100.0 / 512.0   # 86.13 frames per second of audio
LAUNCH_NAMES = ["qkv", "attn_self", "o", "cq", "attn_cross", "co", "wi", "wo"]
TOTAL_UTTERANCES_MULTI_GPU = 64   # BASELINE.json configs[4]


def default_batch(world: int) -> int:
    """utterances per GPU when --batch is not given: BASELINE configs[1] on one GPU, configs[4] (64 in all) on N"""
    return 1 if world == 1 else max(1, TOTAL_UTTERANCES_MULTI_GPU // world)


def cpu_threads() -> int:
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    return max(1, min(16, n))     # a 1-GPU job owns a 16-core share of the host


def texts_for(batch: int, cfg):
    """batch 1: the README prompt (98 byte tokens); else SURVEY.md §8d's mixed lengths 32..512 (sum 1664 per 8)"""
    if batch == 1:
        return [PROMPT]
    from dia_hip.tokens import synthetic_text
    return [synthetic_text(MIXED_L[i % len(MIXED_L)], cfg) for i in range(batch)]


def launch_bytes(sess, w, n_keys: int):
    """algorithmic HBM bytes of every launch of one decode step, in launch order: the matrix a GEMV
    streams, the K/V an attention launch reads (self: both CFG rows of every utterance at length n_keys; cross: the cond
    rows' text keys), the logits head; the sampler reads the logits it is handed (L2-resident: 0)"""
    d = sess.cfg.model.decoder
    kvb = 2 if sess.kv_code == 1 else 4        # bf16: 2 bytes; fp32 and the two-plane bf16 caches: 4
    kv_self = 2 * sess.R * d.kv_heads * 128 * kvb * n_keys
    kv_cross = 2 * d.cross_query_heads * 128 * kvb * sum(sess.lens)
    out = []
    if getattr(sess, "seg", False):        # persistent MLP segments: co, wi, wo and the NEXT layer's qkv are one launch
        for i, L in enumerate(w.dec_layers):
            if i == 0:
                out.append(L["qkv"].nbytes)
            out += [kv_self, L["o"].nbytes, L["cq"].nbytes, kv_cross, w.seg_layers[i].numel() * 2]
    else:
        for L in w.dec_layers:
            out += [L["qkv"].nbytes, kv_self, L["o"].nbytes, L["cq"].nbytes, kv_cross, L["co"].nbytes, L["wi"].nbytes, L["wo"].nbytes]
    out += [w.logits.nbytes, 0]
    return out


def launch_ops(sess, nl: int):
    """op label of every launch of a step, in launch order"""
    if getattr(sess, "seg", False):
        ops = []
        for i in range(nl):
            ops += (["qkv"] if i == 0 else []) + ["attn_self", "o", "cq", "attn_cross", "seg_co_wi_wo_qkv" if i + 1 < nl else "seg_co_wi_wo"]
        return ops + ["logits", "sample_fsm_embed"]
    return [LAUNCH_NAMES[i % 8] for i in range(nl * 8)] + ["logits", "sample_fsm_embed"]


def kernel_table(sess, w, reps: int, graph_step_us: float):
    """Per kernel instantiation, live: `reps` eager steps chained behind a device-side delay, every kernel bracketed by its own
    dispatch-level start / stop events (timestamps of the dispatch packet).
      kernel_us      mean begin -> end of the kernel itself
      us_per_launch  kernel_us + boundary, boundary = (graph-replayed step time - sum of kernel_us) / launches: what a launch
                     costs inside the replayed chain.  rocprofv3 --kernel-trace reports this quantity as the kernel's
                     duration under graph replay (its begin[n+1] == end[n]: the durations partition the step), so the
                     committed profiles/*kernel_stats* averages are the cross-check; the roofline fractions use it.
    The eager chain itself is NOT the measure of step time (per-kernel events add ~5 us between launches)."""
    names, ms = None, []
    for _ in range(reps):
        ms.append(sess.time_step())
        names = sess.last_kernel_names
    ms = np.stack(ms)                                   # [reps, launches]
    n_keys = int(sess.cur.max().item())
    byts = launch_bytes(sess, w, n_keys)
    nl = sess.cfg.model.decoder.n_layer
    ops = launch_ops(sess, nl)
    kus = ms.mean(axis=0) * 1e3                          # per launch
    boundary = max(0.0, (graph_step_us - float(kus.sum())) / len(kus))
    tab = {}
    for i, nm in enumerate(names):
        e = tab.setdefault(nm, {"launches_per_step": 0, "kus": 0.0, "bytes": 0.0, "ops": set()})
        e["launches_per_step"] += 1
        e["kus"] += float(kus[i])
        e["bytes"] += byts[i] if i < len(byts) else 0
        e["ops"].add(ops[i] if i < len(ops) else "?")
    rows = []
    for nm, e in tab.items():
        n = e["launches_per_step"]
        k_us, b = e["kus"] / n, e["bytes"] / n
        us = k_us + boundary
        rows.append({"kernel": nm, "ops": sorted(e["ops"]), "launches_per_step": n, "us_per_launch": round(us, 3),
                     "kernel_us": round(k_us, 3), "bytes_per_launch": int(b),
                     "achieved": round(b / (us * 1e-6) / 1e9, 1), "frac": round(b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                     "share_of_step_time": round(us * n / graph_step_us, 4)})
    rows.sort(key=lambda r: -r["share_of_step_time"])
    per_op = {}
    for i, o in enumerate(ops[: len(kus)]):
        per_op.setdefault(o, []).append(float(kus[i]) + boundary)
    per_op = {o: round(float(np.mean(v)), 3) for o, v in per_op.items()}
    return rows, per_op, round(boundary, 3)


def measure(w, cfg, *, batch, kv, steps, warmup, use_graph=True, seeds=None, dist=None, dev=None, profile_reps=0):
    """One configuration: session, prefill (warm pass timed on the GPU), `warmup` untimed + `steps` timed decode steps
    between barriers; returns the numbers of the JSON line for it."""
    from dia_hip.engine import DecodeSession
    from dia_hip.tokens import effective_text, encode_text

    texts = texts_for(batch, cfg)
    ids = [encode_text(effective_text(t), cfg) for t in texts]
    max_tokens = 1 + warmup + steps + profile_reps + 2
    if max_tokens > cfg.data.audio_length:
        raise SystemExit(f"warmup+steps must stay below audio_length={cfg.data.audio_length}")
    sess = DecodeSession(w, ids, kv_dtype=kv, max_tokens=max_tokens, seeds=seeds or [42 + i for i in range(batch)], ignore_eos=True)
    tp = time.time()
    sess.prefill()
    sess.sync()
    prefill_s = time.time() - tp
    # second, warm pass bracketed by events on the session's stream: GPU-side prefill time (the first pass pays
    # module loading and allocator growth), for the MFMA utilisation of the prefill GEMMs (north star)
    # The ~80 launches are queued BEHIND a device-side spin of a few milliseconds, so that the interval between the events is the GPU's own
    # time whatever the host needs to enqueue them (0.5-0.8 ms from Python at batch 1 — as long as the GPU takes; reported beside it)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sess.stream):
        torch.cuda._sleep(int(8e6))
    ev0.record(sess.stream)
    th = time.time()
    sess.prefill()
    prefill_enqueue_s = time.time() - th
    ev1.record(sess.stream)
    sess.sync()
    prefill_gpu_s = ev0.elapsed_time(ev1) * 1e-3
    e_ = cfg.model.encoder
    rows_ = sum(sess.lens)
    prefill_params = w.prefill_weight_bytes() / 2.0          # encoder + cross-K/V matrices as loaded (compacted if pruned)
    attn_flops = e_.n_layer * e_.n_head * 4 * 128 * sum(l * l for l in sess.lens)
    prefill_flops = 2.0 * rows_ * prefill_params + attn_flops
    alg_tflops = prefill_flops / prefill_gpu_s / 1e12
    prefill = {"text_bytes": rows_, "gpu_s": round(prefill_gpu_s, 5), "host_enqueue_s": round(prefill_enqueue_s, 5), "host_s_first_call": round(prefill_s, 4),
               "algorithmic_tflop": round(prefill_flops / 1e12, 4), "achieved_tflops_algorithmic": round(alg_tflops, 1),
               "peak_tflops_bf16_dense": MFMA_PEAK_TFLOPS, "mfma_frac": round(alg_tflops / MFMA_PEAK_TFLOPS, 4),
               "mfma_frac_issued": round(3 * alg_tflops / MFMA_PEAK_TFLOPS, 4),
               "note": "gpu_s = events around the whole warm prefill pass, its launches queued behind a device-side spin (the GPU's own time); host_enqueue_s = "
                       "what the Python host needed to enqueue them.  mfma_frac = ALGORITHMIC flops (2*rows*params + attention) / gpu_s / 2.5 PFLOP/s; every product is fp32-exact = 3 bf16 MFMAs (hi/mid/lo activation "
                       "plane x bf16 weight), so the matrix pipe issues 3x that (mfma_frac_issued)"}

    sess.ensure_noise(warmup + steps + profile_reps + 2)      # host RNG + upload stay outside the timed region
    sess.decode(warmup, use_graph)
    sess.sync()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t1 = time.perf_counter()
    ev0.record(sess.stream)
    sess.decode(steps, use_graph)
    ev1.record(sess.stream)
    sess.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t1
    dev_ms = ev0.elapsed_time(ev1)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    cur_after = int(sess.cur.min().item())
    assert cur_after == 1 + warmup + steps, (cur_after, warmup, steps)        # every timed step really executed
    n_mid = warmup + (steps + 1) / 2.0                      # whole-step algorithmic traffic at the region's mean KV length
    step_bytes = sess.step_bytes(int(round(n_mid)))
    step_gbs = step_bytes / (dev_ms / steps * 1e-3) / 1e9
    out = {"batch": batch, "kv": kv, "steps": steps, "warmup": warmup, "text_bytes": list(sess.lens),
           "elapsed_s": elapsed, "ms_per_step": round(elapsed / steps * 1e3, 4), "device_ms_per_step": round(dev_ms / steps, 4),
           "frames_per_s": round(batch * steps / elapsed, 2), "rtf_per_utterance": round(steps / elapsed / FRAME_RATE, 3),
           "decode_weight_bytes": int(w.decode_weight_bytes()),
           "activations": "fp32 tiles, planes split in registers (more than 4 rows)" if sess.act_f32 else "three bf16 planes (hi + mid + lo == fp32)",
           "launches_per_step": sess.launches_per_step(), "persistent_segments": bool(getattr(sess, "seg", False)),
           "step_roofline": {"bytes_per_step": int(step_bytes), "achieved": round(step_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(step_gbs / HBM_PEAK_GBS, 4)},
           "prefill": prefill}
    if profile_reps > 0:
        rows, per_op, boundary = kernel_table(sess, w, profile_reps, dev_ms / steps * 1e3)
        out["kernels"] = rows
        out["us_per_launch_by_op"] = per_op
        out["boundary_us_per_launch"] = boundary
    sess.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--batch", type=int, default=0, help="utterances per GPU (default: 1 on one GPU, 64/N on N GPUs)")
    ap.add_argument("--kv", default="bf16", choices=["bf16", "f32", "bf16x2"])
    ap.add_argument("--cpu-steps", type=int, default=40, help="decode steps of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=3, help="eager steps timed kernel by kernel for the roofline objects (0 = skip)")
    ap.add_argument("--pruned", type=float, default=0.0, help="structured dim-0 pruning amount applied to the synthetic checkpoint (BASELINE config 4: 0.5)")
    ap.add_argument("--no-compact", action="store_true", help="with --pruned: stream the zeros instead of compacting")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` object (the other single-GPU BASELINE configurations)")
    ap.add_argument("--config-steps", type=int, default=1024, help="decode steps of each `configs` entry")
    ap.add_argument("--preheat", type=int, default=128, help="untimed decode steps on a throw-away session before anything is measured "
                    "(a freshly leased, idle GPU needs ~0.1 s of load to reach its clocks; independent of --warmup, which is part of the measured session)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N>1 launch with python -m torch.distributed.run --nproc-per-node N bench.py ...")
    # rehearsal knob (1-GPU boxes): DIA_BENCH_SHARE_DEVICE=1 puts every rank on cuda:0 and uses gloo,
    # which exercises the whole N>1 code path except RCCL itself; never set by the driver
    share = os.environ.get("DIA_BENCH_SHARE_DEVICE") == "1"
    dev_index = 0 if share else local
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from dia_hip import config as C
    from dia_hip.engine import DeviceWeights
    from dia_hip.weights import synthetic_state_dict
    from dia_hip.dist import broadcast_weights

    cfg = C.dia_1_6b_config()
    K, Wm = args.steps, args.warmup
    batch = args.batch if args.batch > 0 else default_batch(world)
    use_graph = not args.no_graph

    # ---- weights: rank 0 builds + repacks, the other ranks receive the flat arena in ONE broadcast over RCCL
    t0 = time.time()
    need_sd = rank == 0 or args.pruned > 0
    sd_gpu = synthetic_state_dict(cfg, seed=1234, std=0.02, device=dev) if need_sd else None
    sd_run = sd_gpu
    if args.pruned > 0:
        from dia_hip.pruning import structured_prune_state_dict
        sd_run, _ = structured_prune_state_dict(cfg, sd_gpu, amount=args.pruned, dim=0, n=2)     # offline_prune.py defaults
    if need_sd:
        w = DeviceWeights(cfg, sd_run, dev, compact="off" if args.no_compact else "auto")
    else:
        w = DeviceWeights.empty_like_config(cfg, dev)
    bcast_s = 0.0
    if world > 1 and not args.pruned:
        torch.cuda.synchronize()
        tb = time.time()
        broadcast_weights(w, src=0)
        torch.cuda.synchronize()
        bcast_s = time.time() - tb
    load_s = time.time() - t0

    if args.preheat > 0:                                      # clock ramp on an idle GPU: not part of any measurement
        from dia_hip.engine import DecodeSession
        from dia_hip.tokens import effective_text, encode_text
        pre = DecodeSession(w, [encode_text(effective_text(PROMPT), cfg)], kv_dtype=args.kv, max_tokens=args.preheat + 4, seeds=[1], ignore_eos=True)
        pre.prefill(); pre.decode(args.preheat, use_graph); pre.sync(); pre.close()
        del pre
    seeds = [42 + 1000 * rank + i for i in range(batch)]
    prof = args.profile_steps if (rank == 0) else 0
    m = measure(w, cfg, batch=batch, kv=args.kv, steps=K, warmup=Wm, use_graph=use_graph, seeds=seeds, dist=dist, dev=dev, profile_reps=prof)

    frames = world * batch * K
    value = frames / m["elapsed_s"]
    pruned_txt = f" {round(args.pruned * 100)}%-structured-pruned (dim 0, compacted={not args.no_compact})" if args.pruned else ""
    if world > 1:
        workload = (f"Dia-1.6B{pruned_txt} bf16 weights, {args.kv} K/V, {world * batch} utterances sharded data-parallel over {world} GPUs "
                    f"({batch} per GPU, mixed text lengths 32..512), {K} decode steps, weights broadcast once from rank 0 "
                    f"(one flat {w.flat.numel() / 1e9:.2f} GB buffer), no per-step collective; the N = 1 point of this curve is the "
                    f"`configs.batch64_mixed_bf16kv` entry of the default single-GPU line (same 64 utterances on one GPU)")
    else:
        workload = (f"Dia-1.6B{pruned_txt} bf16 weights, {args.kv} K/V, batch {batch}, {K} decode steps, text bytes {m['text_bytes']}")
    workload += f", cfg 3.0 / T 1.3 / top-p 0.95 / top-k 35, hipGraph={use_graph}"
    out = {
        "metric": "audio-codec frames/sec (Dia-1.6B decode, whole job)", "value": round(value, 2), "unit": "frames/s",
        "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": m["ms_per_step"], "higher_is_better": True,
        # N > 1 without --batch: BASELINE configs[4], 64 utterances in all over the N GPUs = fixed total work ("strong");
        # an explicit --batch fixes the per-GPU work ("weak"), and so does the single-GPU line
        "scaling": "strong" if (world > 1 and args.batch <= 0) else "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": workload, "batch_per_gpu": batch, "parallelism": f"dp{world}" if world > 1 else "single"},
        "frames_per_s_per_gpu": round(value / world, 2), "rtf_per_gpu": round(value / world / batch / FRAME_RATE, 3),
        "rtf_aggregate": round(value / FRAME_RATE, 2),
        "prefill_s": m["prefill"]["host_s_first_call"], "prefill": m["prefill"], "weights_load_s": round(load_s, 2),
        "weights_bcast_s": round(bcast_s, 3),
        "device_ms_per_step": m["device_ms_per_step"], "decode_weight_bytes": m["decode_weight_bytes"], "activations": m["activations"], "preheat_steps": args.preheat,
        "step_roofline": m["step_roofline"],
    }
    if "kernels" in m:
        rows = m["kernels"]
        top = dict(rows[0])
        # HBM bytes per launch from the PMC counters (separate rocprofv3 --pmc passes, FETCH_SIZE doubled as the gfx950 guide
        # prescribes) cannot be collected from inside this process; they are measured by scratch/profile_round.sh on the
        # same command and committed under profiles/ keyed by kernel name
        traffic, src = None, None
        tr = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.isfile(tr) and not args.pruned:
            try:
                ent = json.load(open(tr)).get(f"batch{batch}", {}).get(top["kernel"])
                if ent:
                    traffic, src = ent["hbm_bytes_per_launch"], ent.get("source")
            except Exception:
                pass
        out["roofline"] = {"bound": "hbm", "kernel": top["kernel"], "ops": top["ops"], "launches_per_step": top["launches_per_step"],
                           "achieved": top["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": top["frac"],
                           "bytes_per_launch": top["bytes_per_launch"], "us_per_launch": top["us_per_launch"],
                           "kernel_us": top["kernel_us"], "share_of_step_time": top["share_of_step_time"], "traffic": traffic,
                           "traffic_source": src,
                           "how": "dominant kernel BY TIME of the decode step.  kernel_us = mean begin -> end of its dispatches (HIP start/stop "
                                  f"events per kernel, {args.profile_steps} eager steps on the engine's stream); us_per_launch = kernel_us + "
                                  "boundary_us_per_launch = (graph-replayed step time - sum of kernel_us) / launches: its cost inside the replayed "
                                  "chain, the quantity rocprofv3 --kernel-trace reports as its duration there (begin[n+1] == end[n]); achieved = "
                                  "mean algorithmic bytes of its launches (SURVEY.md §8d) / us_per_launch"}
        out["roofline_by_kernel"] = rows
        out["us_per_launch_by_op"] = m["us_per_launch_by_op"]
        out["boundary_us_per_launch"] = m["boundary_us_per_launch"]

    # ---- the other single-GPU configurations of BASELINE.json, 1024 steps each whatever --steps says
    if rank == 0 and world == 1 and not args.no_configs and not args.pruned and args.batch in (0, 1) and args.kv == "bf16":
        cs = args.config_steps
        cfgs = {}

        def brief(r, name):
            return {"workload": name, "frames_per_s": r["frames_per_s"], "ms_per_step": r["ms_per_step"], "steps": r["steps"],
                    "rtf_per_utterance": r["rtf_per_utterance"], "decode_weight_bytes": r["decode_weight_bytes"], "activations": r["activations"],
                    "step_bytes": r["step_roofline"]["bytes_per_step"], "step_frac_of_hbm_peak": r["step_roofline"]["frac"],
                    "prefill_gpu_ms": round(r["prefill"]["gpu_s"] * 1e3, 2), "prefill_mfma_frac": r["prefill"]["mfma_frac"]}

        if K == cs and Wm == 16:
            cfgs["batch1_bf16kv"] = brief(m, "BASELINE configs[1]: batch 1, bf16 K/V (= the headline line)")
        else:
            cfgs["batch1_bf16kv"] = brief(measure(w, cfg, batch=1, kv="bf16", steps=cs, warmup=16), "BASELINE configs[1]: batch 1, bf16 K/V")
        cfgs["batch1_f32kv"] = brief(measure(w, cfg, batch=1, kv="f32", steps=cs, warmup=16),
                                     "batch 1, fp32 K/V: the configuration whose logits meet 1e-3 against the fp32 reference path")
        cfgs["batch1_bf16x2kv"] = brief(measure(w, cfg, batch=1, kv="bf16x2", steps=cs, warmup=16),
                                        "batch 1, K/V as two bf16 planes (hi + lo, the bytes of fp32 caches) through the MFMA attention kernel: logits within 1e-3 too")
        cfgs["batch8_mixed_bf16kv"] = brief(measure(w, cfg, batch=8, kv="bf16", steps=cs, warmup=16),
                                            "BASELINE configs[2]: batch 8, text bytes 32..512 (sum 1664), bf16 K/V")
        cfgs["batch8_mixed_f32kv"] = brief(measure(w, cfg, batch=8, kv="f32", steps=cs, warmup=16), "batch 8 mixed, fp32 K/V (parity configuration)")
        cfgs["batch8_mixed_bf16x2kv"] = brief(measure(w, cfg, batch=8, kv="bf16x2", steps=cs, warmup=16), "batch 8 mixed, two-plane bf16 K/V (MFMA attention, parity bound met)")
        # the N = 1 point of BASELINE configs[4]'s scaling curve: all 64 utterances on this one GPU (what `--gpus N` shards 64/N per GPU)
        cfgs["batch64_mixed_bf16kv"] = brief(measure(w, cfg, batch=TOTAL_UTTERANCES_MULTI_GPU, kv="bf16", steps=min(cs, 512), warmup=16),
                                             "BASELINE configs[4] at N = 1: 64 utterances on one GPU (text bytes 32..512 x 8), bf16 K/V, 512 steps; "
                                             "the `--gpus N` lines shard these 64 utterances 64/N per GPU")
        from dia_hip.pruning import structured_prune_state_dict
        psd, _ = structured_prune_state_dict(cfg, sd_gpu, amount=0.5, dim=0, n=2)
        wp = DeviceWeights(cfg, psd, dev)
        del psd
        cfgs["pruned50_batch8_bf16kv"] = brief(measure(wp, cfg, batch=8, kv="bf16", steps=cs, warmup=16),
                                               "BASELINE configs[3]: 50 %-structured-pruned (offline_prune.py defaults), compacted, batch 8 mixed")
        cfgs["pruned50_batch1_bf16kv"] = brief(measure(wp, cfg, batch=1, kv="bf16", steps=cs, warmup=16), "same checkpoint, batch 1")
        del wp
        torch.cuda.empty_cache()
        out["configs"] = cfgs

    # ---- CPU baseline: the oracle in mirror mode (= the reference's op sequence incl. its dead
    #      cross-K/V work), fp32, same weights/prompt/seed, bounded sample, rank 0 at N=1 only
    if rank == 0 and world == 1 and args.cpu_steps > 0:
        from oracle import dia_oracle as O
        nthr = cpu_threads()
        torch.set_num_threads(nthr)
        sd_cpu = {k: v.cpu() for k, v in sd_gpu.items()}
        del sd_gpu
        tc = time.time()
        r = O.generate(sd_cpu, cfg, PROMPT, max_tokens=cfg.data.audio_length, seed=42, mirror=True,
                       keep_logits=False, max_steps=args.cpu_steps)
        total = time.time() - tc
        med = float(np.median(r.step_ms))
        out["cpu_baseline"] = {
            "value": round(1e3 / med, 3), "unit": "frames/s", "cores": nthr, "kind": "port",
            "sample": f"oracle mirror mode (reference op sequence incl. dead cross-K/V re-projection), fp32, batch 1, "
                      f"prefill {r.prep_s:.1f}s + {len(r.step_ms)} decode steps, median {med:.0f} ms/step, {total:.1f}s total",
        }
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
