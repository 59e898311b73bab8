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

    for name, v in vars(w).items():
        if name != "flat":
            walk(v)
    assert seen == ids
    # every tensor lives inside the one flat arena (a single allocation = a single broadcast)
    lo, hi = w.flat.data_ptr(), w.flat.data_ptr() + w.flat.numel()
    for t in D.weight_tensors(w):
        assert lo <= t.data_ptr() and t.data_ptr() + t.numel() * t.element_size() <= hi
        assert t.data_ptr() % 256 == lo % 256
    z = DeviceWeights.empty_like_config(cfg, torch.device("cpu"))
    assert [tuple(t.shape) for t in D.weight_tensors(z)] == [tuple(t.shape) for t in D.weight_tensors(w)]
    assert z.flat.numel() == w.flat.numel()


def _worker_model(rank, world, port, q):
    """rank 0 holds the model, the other ranks receive it in ONE broadcast of the flat arena; then 5 utterances are
    sharded u -> u mod world, every rank decodes its own (the CPU oracle on the tiny config stands in for the HIP session)
    and the token buffers are gathered back in utterance order."""
    sys.path.insert(0, os.path.join(ROOT, "dia-tts-prune_amd"))
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch.distributed as dist
    from dia_hip import config as C
    from dia_hip import dist as D
    from dia_hip.engine import DeviceWeights
    from dia_hip.weights import synthetic_state_dict
    from oracle import dia_oracle as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg = C.mid_config()                 # real head_dim (the device layouts are built for 128)
    sd = synthetic_state_dict(cfg, seed=7, std=0.02)
    cpu = torch.device("cpu")
    w = DeviceWeights(cfg, sd, cpu) if rank == 0 else DeviceWeights.empty_like_config(cfg, cpu)
    calls = []
    orig = dist.broadcast

    def counting(*a, **k):
        calls.append(1)
        return orig(*a, **k)

    dist.broadcast = counting
    nbytes = D.broadcast_weights(w, src=0)
    dist.broadcast = orig
    ref = DeviceWeights(cfg, sd, cpu)
    same = all(torch.equal(a, b) for a, b in zip(D.weight_tensors(w), D.weight_tensors(ref)))
    texts = ["[S1] one.", "[S2] two two.", "[S1] three three three.", "[S1] four. [S2] four.", "[S2] five!"]
    mt = 10
    mine = D.shard_utterances(len(texts), world, rank)
    local = [torch.from_numpy(O.generate(sd, cfg, texts[u], max_tokens=mt, seed=100 + u, keep_logits=False).tokens) for u in mine]
    allb = D.gather_utterances(local, len(texts), world, rank)
    q.put((rank, same, len(calls), nbytes == w.flat.numel(), mine, [b.numpy() for b in allb]))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_one_broadcast_and_sharded_decode():
    sys.path.insert(0, os.path.join(ROOT, "dia-tts-prune_amd"))
    sys.path.insert(0, ROOT)
    import numpy as np
    from dia_hip import config as C
    from dia_hip.weights import synthetic_state_dict
    from oracle import dia_oracle as O

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker_model, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(2)), key=lambda t: t[0])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, ncalls, nb_ok, mine, bufs in res:
        assert same and nb_ok
        assert ncalls == 1                                   # the whole model travels in ONE collective
        assert mine == [u for u in range(5) if u % 2 == rank]
    # single-process run of all five utterances: the gathered buffers are those, in utterance order, on both ranks
    cfg = C.mid_config()
    sd = synthetic_state_dict(cfg, seed=7, std=0.02)
    texts = ["[S1] one.", "[S2] two two.", "[S1] three three three.", "[S1] four. [S2] four.", "[S2] five!"]
    torch.set_num_threads(4)
    for u, t in enumerate(texts):
        want = O.generate(sd, cfg, t, max_tokens=10, seed=100 + u, keep_logits=False).tokens
        for _, _, _, _, _, bufs in res:
            assert np.array_equal(bufs[u], want), u


def _worker_edges(rank, world, port, q):
    """gather with an empty shard and ragged [T, C]; a weight-arena mismatch must raise on EVERY rank (no rank is left
    waiting in the broadcast)."""
    sys.path.insert(0, os.path.join(ROOT, "dia-tts-prune_amd"))
    import torch.distributed as dist
    from dia_hip import config as C
    from dia_hip import dist as D
    from dia_hip.engine import DeviceWeights
    from dia_hip.weights import synthetic_state_dict

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cpu = torch.device("cpu")
    # one utterance over two ranks: rank 1 owns nothing and names the device itself
    local = [torch.arange(6 * 9, dtype=torch.int32).reshape(6, 9)] if rank == 0 else []
    got = D.gather_utterances(local, 1, world, rank, device=cpu)
    ok_empty = len(got) == 1 and torch.equal(got[0], torch.arange(6 * 9, dtype=torch.int32).reshape(6, 9))
    # three utterances with different row counts: buffers come back cut to what their owner sent
    rows = {0: 4, 1: 7, 2: 5}
    mine = D.shard_utterances(3, world, rank)
    local = [torch.full((rows[u], 9), u, dtype=torch.int32) for u in mine]
    got = D.gather_utterances(local, 3, world, rank)
    ok_ragged = all(tuple(got[u].shape) == (rows[u], 9) and bool((got[u] == u).all()) for u in range(3))
    # arena mismatch: rank 0 holds three weight planes, rank 1 one -> both raise before any broadcast
    cfg = C.mid_config()                 # real head_dim (the device layouts are built for 128)
    if rank == 0:
        w = DeviceWeights(cfg, synthetic_state_dict(cfg, 3, 0.02), cpu, weight_planes=3)
    else:
        w = DeviceWeights.empty_like_config(cfg, cpu)
    raised = False
    try:
        D.broadcast_weights(w, src=0)
    except ValueError as e:
        raised = "differ across ranks" in str(e)
    # and the matching receiver goes through
    w1 = DeviceWeights(cfg, synthetic_state_dict(cfg, 3, 0.02), cpu, weight_planes=3) if rank == 0 else \
        DeviceWeights.empty_like_config(cfg, cpu, weight_planes=3)
    n = D.broadcast_weights(w1, src=0)
    q.put((rank, ok_empty, ok_ragged, raised, n == w1.flat.numel() and float(w1.flat.float().abs().sum()) > 0))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_gather_edges_and_arena_mismatch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker_edges, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_empty, ok_ragged, raised, ok_bcast in res:
        assert ok_empty and ok_ragged and raised and ok_bcast, (rank, ok_empty, ok_ragged, raised, ok_bcast)
