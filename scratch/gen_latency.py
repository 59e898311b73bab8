"""wall-time anatomy of one generate call (batch 1, full size)"""
import sys, time
sys.path.insert(0, "dia-tts-prune_amd")
import torch
from dia_hip import config as C
from dia_hip.engine import DecodeSession, DeviceWeights
from dia_hip.tokens import effective_text, encode_text
from dia_hip.weights import synthetic_state_dict
cfg = C.dia_1_6b_config(); dev = torch.device("cuda:0")
w = DeviceWeights(cfg, synthetic_state_dict(cfg, seed=1234, std=0.02, device=dev), dev)
ids = [encode_text(effective_text("[S1] Dia is an open weights text to dialogue model. [S2] You get full control over scripts and voices."), cfg)]
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    s = DecodeSession(w, ids, kv_dtype="bf16", max_tokens=3072, seeds=[rep], ignore_eos=True)
    torch.cuda.synchronize(); t1 = time.time()
    s.prefill(); s.sync(); t2 = time.time()
    s.decode(1, True); s.sync(); t3 = time.time()          # first frame (includes graph capture)
    s.decode(199, True); s.sync(); t4 = time.time()
    r = s.results(); t5 = time.time(); s.close()
    print(f"rep {rep}: session {1e3*(t1-t0):.1f} ms, prefill {1e3*(t2-t1):.1f} ms, first step {1e3*(t3-t2):.1f} ms, 199 steps {1e3*(t4-t3):.1f} ms, results {1e3*(t5-t4):.1f} ms")
