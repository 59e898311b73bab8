"""The N>1 path on CPU: world_size-2 gloo — weight broadcast, utterance sharding, token gather."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "dia-tts-prune_amd"))
    import torch.distributed as dist
    from dia_hip import dist as D

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    ref = [torch.randn(33, 7), torch.randn(5, 64, 8).bfloat16(), torch.arange(12, dtype=torch.int32).reshape(3, 4)]
    mine = [t.clone() if rank == 0 else torch.zeros_like(t) for t in ref]
    nbytes = D.broadcast_tensors(mine, src=0)
    ok = all(torch.equal(a, b) for a, b in zip(mine, ref)) and nbytes == sum(t.numel() * t.element_size() for t in ref)
    shard = D.shard_utterances(5, world, rank)
    toks = torch.full((2, 4, 9), rank, dtype=torch.int32)
    gathered = D.gather_token_buffers(toks, world)
    ok = ok and all(int(g[0, 0, 0]) == r for r, g in enumerate(gathered))
    q.put((rank, ok, shard))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_broadcast_and_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=60) for _ in range(2))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1]
    assert res[0][2] == [0, 2, 4] and res[1][2] == [1, 3]
    # every utterance is owned by exactly one rank
    assert sorted(res[0][2] + res[1][2]) == list(range(5))


def test_weight_tensor_inventory_is_complete():
    """broadcast_weights must cover every tensor a DecodeSession reads from DeviceWeights."""
    sys.path.insert(0, os.path.join(ROOT, "dia-tts-prune_amd"))
    from dia_hip import config as C
    from dia_hip import dist as D
    from dia_hip.engine import DeviceWeights
    from dia_hip.weights import synthetic_state_dict

    cfg = C.mid_config()
    w = DeviceWeights(cfg, synthetic_state_dict(cfg, 1, 0.02), torch.device("cpu"))
    ids = {id(t) for t in D.weight_tensors(w)}
    seen = set()

    def walk(o):
        if isinstance(o, torch.Tensor):
            seen.add(id(o))
        elif isinstance(o, dict):
            for v in o.values():
                walk(v)
        elif isinstance(o, (list, tuple)):
            for v in o:
                walk(v)
        elif hasattr(o, "t") and isinstance(getattr(o, "t"), torch.Tensor):
            seen.add(id(o.t))

    for v in vars(w).values():
        walk(v)
    assert seen == ids
    z = DeviceWeights.empty_like_config(cfg, torch.device("cpu"))
    assert [tuple(t.shape) for t in D.weight_tensors(z)] == [tuple(t.shape) for t in D.weight_tensors(w)]
