#!/usr/bin/env python3
"""Headline benchmark of the MI355X decode path: audio-codec frames/s/GPU for Dia-1.6B (+ RTF).

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run)

A "step" is one decode step of the whole batch (one frame per utterance = 9 codebook tokens).
Workload at N=1: BASELINE.json configs[1] — Dia-1.6B shapes, bf16 weights + bf16 K/V, batch 1,
1024 decode steps, synthetic seeded weights (no checkpoint exists offline), README prompt.
Everything is resident in HBM before the timed region; the loop replays one hipGraph per step and
never syncs with the host inside the region.  Prints ONE JSON line (rank 0).
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
MIXED_L = [32, 64, 96, 128, 192, 256, 384, 512]
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
FRAME_RATE = 44100.0 / 512.0   # 86.13 frames per second of audio
LAUNCH_NAMES = ["qkv", "attn_self", "o", "cq", "attn_cross", "co", "wi", "wo"]


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--batch", type=int, default=1, help="utterances per GPU")
    ap.add_argument("--kv", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--cpu-steps", type=int, default=40, help="decode steps of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--prefetch", type=int, default=0, help="weight prefetch lookahead in launches (experiment)")
    ap.add_argument("--pruned", type=float, default=0.0, help="structured dim-0 pruning amount applied to the synthetic checkpoint (BASELINE config 4: 0.5)")
    ap.add_argument("--no-compact", action="store_true", help="with --pruned: stream the zeros instead of compacting")
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
    from dia_hip.engine import DecodeSession, DeviceWeights
    from dia_hip.tokens import effective_text, encode_text
    from dia_hip.weights import synthetic_state_dict
    from dia_hip.dist import broadcast_weights

    cfg = C.dia_1_6b_config()
    K, Wm = args.steps, args.warmup
    max_tokens = 1 + Wm + K + args.profile_steps + 1
    if max_tokens > cfg.data.audio_length:
        raise SystemExit(f"warmup+steps must stay below audio_length={cfg.data.audio_length}")

    # ---- weights: rank 0 builds + repacks, the other ranks receive the repacked tensors over RCCL
    t0 = time.time()
    sd_gpu = synthetic_state_dict(cfg, seed=1234, std=0.02, device=dev) if (rank == 0 or args.pruned > 0) else None
    if args.pruned > 0:
        from dia_hip.pruning import structured_prune_state_dict
        sd_gpu, _ = structured_prune_state_dict(cfg, sd_gpu, amount=args.pruned, dim=0, n=2)     # offline_prune.py defaults
    if rank == 0 or args.pruned > 0:
        w = DeviceWeights(cfg, sd_gpu, dev, compact="off" if args.no_compact else "auto")
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

    texts = texts_for(args.batch, cfg)
    ids = [encode_text(effective_text(t), cfg) for t in texts]
    seeds = [42 + 1000 * rank + i for i in range(args.batch)]
    sess = DecodeSession(w, ids, kv_dtype=args.kv, max_tokens=max_tokens, seeds=seeds, ignore_eos=True)
    if args.prefetch > 0:
        sess.set_prefetch(args.prefetch)
    tp = time.time()
    sess.prefill()
    sess.sync()
    prefill_s = time.time() - tp
    # second, warm pass bracketed by events on the session's stream: GPU-side prefill time (the first pass pays
    # module loading and allocator growth), for the MFMA utilisation of the prefill GEMMs (north star)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(sess.stream)
    sess.prefill()
    ev1.record(sess.stream)
    sess.sync()
    prefill_gpu_s = ev0.elapsed_time(ev1) * 1e-3
    e_, d_ = cfg.model.encoder, cfg.model.decoder
    rows_ = sum(sess.lens)
    prefill_params = w.prefill_weight_bytes() / 2.0          # encoder + cross-K/V matrices as loaded (compacted if pruned)
    attn_flops = e_.n_layer * e_.n_head * 4 * 128 * sum(l * l for l in sess.lens)
    prefill_flops = 2.0 * rows_ * prefill_params + attn_flops
    MFMA_PEAK_TFLOPS = 2500.0                      # dense bf16, MI355X_MICROARCH.md
    prefill = {"text_bytes": rows_, "gpu_s": round(prefill_gpu_s, 5), "host_s_first_call": round(prefill_s, 4),
               "algorithmic_tflop": round(prefill_flops / 1e12, 4),
               "achieved_tflops_algorithmic": round(prefill_flops / prefill_gpu_s / 1e12, 1),
               "mfma_tflops_issued": round(3 * prefill_flops / prefill_gpu_s / 1e12, 1),
               "peak_tflops_bf16_dense": MFMA_PEAK_TFLOPS,
               "mfma_frac": round(3 * prefill_flops / prefill_gpu_s / 1e12 / MFMA_PEAK_TFLOPS, 4),
               "note": "every product is fp32-exact = 3 bf16 MFMAs (hi/mid/lo activation plane x bf16 weight); mfma_frac counts the "
                       "issued MFMA work over the whole prefill interval incl. launch gaps of the host-driven chain"}

    use_graph = not args.no_graph
    sess.ensure_noise(Wm + K + args.profile_steps + 1)       # host RNG + upload stay outside the timed region
    sess.decode(Wm, use_graph)
    sess.sync()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t1 = time.perf_counter()
    ev0.record(sess.stream)
    sess.decode(K, use_graph)
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
    assert cur_after == 1 + Wm + K, (cur_after, Wm, K)        # every timed step really executed

    # ---- per-launch HIP-event timing of eager steps (KV length ~ Wm+K), on the engine's stream
    prof = np.stack([sess.profile_step() for _ in range(args.profile_steps)]) if args.profile_steps > 0 else None
    nl = cfg.model.decoder.n_layer
    roof = None
    breakdown = None
    if prof is not None:
        per = prof[:, : nl * 8].reshape(-1, nl, 8)                 # [rep, layer, kind]
        kind_ms = per.mean(axis=(0, 1))
        breakdown = {LAUNCH_NAMES[i]: round(float(kind_ms[i]) * 1e3, 2) for i in range(8)}
        breakdown["logits"] = round(float(prof[:, nl * 8].mean()) * 1e3, 2)
        breakdown["sample_fsm_embed"] = round(float(prof[:, nl * 8 + 1].mean()) * 1e3, 2)
        breakdown["unit"] = "us between per-launch HIP events of one eager step (each interval carries ~3 us of event/boundary overhead)"
        wi_ms = sess.time_wi_launches(reps=5) * 1e3          # dispatch-level start/stop events per launch
        wi_bytes = sum(L["wi"].nbytes for L in w.dec_layers) / len(w.dec_layers)   # algorithmic bytes of the dominant kernel
        rows = 2 * args.batch
        wi_k, wi_n = w.dec_layers[0]["wi"].kt * 32, w.dec_layers[0]["wi"].ns * 16
        # the dispatcher's choice for this shape (dia_gemm in gemm.hip); rocprofv3's kernel name in
        # profiles/ is the authority
        if rows <= 4:
            kname = "k_gemv_small<NW=16,KPW=%d,RS=%d,MULTI>" % (wi_k // 32 // 16, 2 if rows <= 2 else 4)
        elif rows <= 16:
            kname = "k_gemm16<NW=8,KPW=%d,MULTI>" % (wi_k // 32 // 8)
        else:
            kname = "k_gemm<MT=%d,NW=4,KPW=%d>" % (min(4, (rows + 15) // 16), wi_k // 32 // 4)
        roof = {"bound": "hbm", "kernel": kname + " on wi_fused [%d x %d] bf16 (SwiGLU epilogue), 18 launches/step" % (wi_k, wi_n),
                "achieved": round(wi_bytes / (wi_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(wi_bytes / (wi_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "bytes_per_launch": wi_bytes, "us_per_launch": round(wi_ms * 1e3, 2), "traffic": None}
        # HBM bytes per launch from the PMC counters (separate rocprofv3 --pmc passes, FETCH_SIZE doubled
        # as the gfx950 guide prescribes); measured offline, committed under profiles/
        tr = os.path.join(ROOT, "profiles", "traffic_wi.json")
        if os.path.isfile(tr) and args.batch == 1 and not args.pruned:      # measured for exactly this kernel and shape
            try:
                roof["traffic"] = json.load(open(tr)).get("hbm_bytes_per_launch")
            except Exception:
                pass

    frames = world * args.batch * K
    value = frames / elapsed
    ms_per_step = elapsed / K * 1e3
    # whole-step algorithmic traffic, averaged over the timed region's KV lengths
    n_mid = Wm + (K + 1) / 2.0
    step_bytes = sess.step_bytes(int(round(n_mid)))
    step_gbs = step_bytes / (dev_ms / K * 1e-3) / 1e9

    out = {
        "metric": "audio-codec frames/sec (Dia-1.6B decode, whole job)", "value": round(value, 2), "unit": "frames/s",
        "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"Dia-1.6B{' %d%%-structured-pruned (dim 0, compacted=%s)' % (round(args.pruned * 100), not args.no_compact) if args.pruned else ''} bf16 weights, {args.kv} K/V, batch {args.batch} per GPU, {K} decode steps, "
                               f"text bytes {sess.lens}, cfg 3.0 / T 1.3 / top-p 0.95 / top-k 35, hipGraph={use_graph}",
                   "batch_per_gpu": args.batch, "parallelism": f"dp{world}" if world > 1 else "single"},
        "frames_per_s_per_gpu": round(value / world, 2), "rtf_per_gpu": round(value / world / args.batch / FRAME_RATE, 3),
        "rtf_aggregate": round(value / FRAME_RATE, 2),
        "prefill_s": round(prefill_s, 4), "prefill": prefill, "weights_load_s": round(load_s, 2), "weights_bcast_s": round(bcast_s, 3),
        "device_ms_per_step": round(dev_ms / K, 4), "decode_weight_bytes": int(w.decode_weight_bytes()),
        "step_roofline": {"bytes_per_step": int(step_bytes), "achieved": round(step_gbs, 1), "peak": HBM_PEAK_GBS,
                          "unit": "GB/s", "frac": round(step_gbs / HBM_PEAK_GBS, 4)},
        "launch_breakdown": breakdown,
    }
    if roof is not None:
        out["roofline"] = roof

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
    sess.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
