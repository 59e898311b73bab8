"""bench.py's N > 1 path on a one-GPU box: two ranks launched the way the driver launches them
(`python -m torch.distributed.run`), both on cuda:0 with gloo as the process group (rehearsal knob
DIA_BENCH_SHARE_DEVICE=1 — everything but RCCL itself: rank-0 weight build, broadcast of the repacked tensors,
per-rank sessions, barrier-bracketed timing, max over ranks, one JSON line from rank 0)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_two_ranks_share_device():
    env = dict(os.environ, DIA_BENCH_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "32", "--warmup", "4",
           "--cpu-steps", "0", "--profile-steps", "0", "--batch", "3"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:                                      # the children's own tracebacks come before the launcher's summary
        i = r.stderr.find("Traceback")
        raise AssertionError(f"bench.py --gpus 2 failed (rc {r.returncode}):\n" + (r.stderr[i: i + 3000] if i >= 0 else r.stderr[-3000:]))
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                  # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 32 and d["warmup"] == 4 and d["scaling"] == "weak"
    assert d["config"]["parallelism"] == "dp2" and d["config"]["batch_per_gpu"] == 3
    assert "6 utterances sharded data-parallel over 2 GPUs" in d["config"]["workload"]
    assert d["value"] > 0 and abs(d["value"] - 2 * 3 * 1000.0 / d["ms_per_step"]) <= 1e-2 * d["value"]     # whole-job frames/s
    assert "configs" not in d                                 # the single-GPU configuration sweep belongs to N = 1
    assert d["weights_bcast_s"] > 0
    assert "cpu_baseline" not in d or d["cpu_baseline"] is None


def test_rccl_world1_broadcast_and_gather_on_device_tensors():
    """The collectives of dia_hip/dist.py under the production backend ("nccl" = RCCL) with tensors on the GPU, as far as one GPU
    reaches: a world of ONE rank initialises RCCL, takes the arena-signature all-reduces and the broadcast of a flat arena, and
    gathers int32 token buffers that live on the device (the gloo tests run them on CPU tensors, which RCCL cannot carry)."""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(%r, "dia-tts-prune_amd"))
from dia_hip import dist as D
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", %r)
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
class W: pass
w = W(); w.flat = torch.arange(1 << 20, dtype=torch.uint8, device=dev); w.compacted = False; w.weight_planes = 1
n = D.broadcast_weights(w, src=0)
assert n == w.flat.numel()
bufs = [torch.full((5 + u, 9), u, dtype=torch.int32, device=dev) for u in range(3)]
got = D.gather_utterances(bufs, 3, 1, 0)
assert all(g.is_cuda and tuple(g.shape) == (5 + u, 9) and bool((g == u).all()) for u, g in enumerate(got))
toks = D.gather_token_buffers(torch.ones(2, 4, 9, dtype=torch.int32, device=dev), 1)
assert len(toks) == 1 and toks[0].is_cuda
dist.barrier(); dist.destroy_process_group()
print("RCCL_WORLD1_OK")
''' % (ROOT, str(free_port()))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
