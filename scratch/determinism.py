"""bit-reproducibility of long runs: graph replay twice + eager, batch 8, 1200 steps, full size"""
import sys
sys.path.insert(0, "dia-tts-prune_amd")
import numpy as np, torch
from dia_hip import config as C
from dia_hip.engine import DecodeSession, DeviceWeights
from dia_hip.tokens import effective_text, encode_text
from dia_hip.weights import synthetic_state_dict
cfg = C.dia_1_6b_config(); dev = torch.device("cuda:0")
w = DeviceWeights(cfg, synthetic_state_dict(cfg, seed=1234, std=0.02, device=dev), dev)
base = "[S1] Dia is an open weights text to dialogue model. [S2] You get full control over scripts and voices. "
texts = [(base * k)[: n] for k, n in ((1, 32), (1, 64), (1, 96), (2, 128), (2, 192), (3, 256), (4, 384), (5, 512))]
ids = [encode_text(effective_text(t), cfg) for t in texts]
outs = []
for mode in ("graph", "graph", "eager"):
    s = DecodeSession(w, ids, kv_dtype="bf16", max_tokens=1201, seeds=list(range(8)), ignore_eos=True)
    s.prefill(); s.run(use_graph=(mode != "eager"))
    outs.append(np.stack([r.tokens for r in s.results()])); s.close()
    print(mode, "done", outs[-1].shape, flush=True)
print("graph == graph:", np.array_equal(outs[0], outs[1]), " graph == eager:", np.array_equal(outs[0], outs[2]))
