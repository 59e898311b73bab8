"""GPU time and host enqueue time of the prefill (encoder + cross-K/V) at batch 1 (98 text bytes) and batch 8 (1664)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dia-tts-prune_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from dia_hip import config as C
from dia_hip.engine import DeviceWeights, DecodeSession
from dia_hip.weights import synthetic_state_dict
from dia_hip.tokens import effective_text, encode_text
cfg = C.dia_1_6b_config(); dev = torch.device("cuda:0")
w = DeviceWeights(cfg, synthetic_state_dict(cfg, seed=1234, std=0.02, device=dev), dev)
for batch in ([int(a) for a in sys.argv[1:]] or [1, 8]):
    texts = bench.texts_for(batch, cfg)
    ids = [encode_text(effective_text(t), cfg) for t in texts]
    s = DecodeSession(w, ids, kv_dtype="bf16", max_tokens=16, seeds=list(range(batch)), ignore_eos=True)
    for rep in range(4):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # the launches are queued BEHIND a device-side spin of a few milliseconds: the interval between the events is then the GPU's own time for
        # the ~80 launches, whatever the host needs to enqueue them (0.5-0.8 ms from Python, close to the GPU time at batch 1)
        with torch.cuda.stream(s.stream):
            torch.cuda._sleep(int(6e6))
        t0 = time.perf_counter(); ev0.record(s.stream); s.prefill(); ev1.record(s.stream); t1 = time.perf_counter(); s.sync()
        print(f"batch {batch} prefill pass {rep}: GPU {ev0.elapsed_time(ev1):.3f} ms, host enqueue {1e3 * (t1 - t0):.3f} ms")
    s.close()
