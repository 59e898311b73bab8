"""audio-prompt prefill at full size: batched MFMA prefill vs replay through the decode step"""
import sys, os, time
sys.path.insert(0, "dia-tts-prune_amd")
import numpy as np, torch
from dia_hip import config as C
from dia_hip.engine import DecodeSession, DeviceWeights
from dia_hip.tokens import effective_text, encode_text
from dia_hip.weights import synthetic_state_dict
cfg = C.dia_1_6b_config(); dev = torch.device("cuda:0")
w = DeviceWeights(cfg, synthetic_state_dict(cfg, seed=1234, std=0.02, device=dev), dev)
text = "[S1] Dia is an open weights text to dialogue model. [S2] You get full control over scripts and voices."
for B, tp in ((1, 430), (8, 430)):
    ids = [encode_text(effective_text(text, "[S1] prompt transcript."), cfg)] * B
    prompts = [np.random.RandomState(b).randint(0, 1024, size=(tp, 9)).astype(np.int32) for b in range(B)]
    for replay in (0, 1):
        s = DecodeSession(w, ids, kv_dtype="bf16", max_tokens=tp + 66, seeds=list(range(B)), audio_prompts=prompts, ignore_eos=True,
                          prompt_prefill="replay" if replay else "auto")
        s.prefill(); s.sync()                      # warm (allocator, modules)
        s.cur.fill_(1); t0 = time.time(); s.prefill(); s.sync(); t1 = time.time()
        n = (tp + 1) - int(s.cur.min().item())     # replay steps still to do before the first sampled step
        s.decode(n, True); s.sync(); t2 = time.time()
        print(f"B={B} prompt {tp} frames, replay={replay}: prefill {1e3 * (t1 - t0):7.1f} ms + {n} replay steps {1e3 * (t2 - t1):7.1f} ms = {1e3 * (t2 - t0):7.1f} ms to the first sampled step", flush=True)
        s.close()
