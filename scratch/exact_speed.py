"""decode speed of the exact-fp32-weights configuration (three bf16 planes per weight, generic kernel, fp32 K/V), Dia-1.6B shapes"""
import sys, time
sys.path.insert(0, "dia-tts-prune_amd")
import torch
from dia_hip import config as C
from dia_hip.engine import DecodeSession, DeviceWeights
from dia_hip.tokens import effective_text, encode_text
from dia_hip.weights import synthetic_state_dict
cfg = C.dia_1_6b_config(); dev = torch.device("cuda:0")
sd = synthetic_state_dict(cfg, seed=1234, std=0.02, device=dev)
g = torch.Generator(device=dev).manual_seed(1)
sd = {k: (v + v.abs().mean() * 2.0 ** -10 * torch.randn(v.shape, generator=g, device=dev)) if v.ndim >= 2 and "embedding" not in k else v for k, v in sd.items()}
ids = [encode_text(effective_text("[S1] Dia is an open weights text to dialogue model. [S2] You get full control over scripts and voices."), cfg)]
for planes in (1, 3):
    w = DeviceWeights(cfg, sd, dev, weight_planes=planes)
    s = DecodeSession(w, ids, kv_dtype="f32", max_tokens=300, seeds=[1], ignore_eos=True)
    t0 = time.time(); s.prefill(); s.sync(); tp = time.time() - t0
    s.decode(16, True); s.sync()
    t0 = time.time(); s.decode(256, True); s.sync(); dt = time.time() - t0
    print(f"weight planes {planes}: prefill {tp * 1e3:.1f} ms, decode {256 / dt:.1f} frames/s ({dt / 256 * 1e3:.3f} ms/step), weights {w.decode_weight_bytes() / 1e9:.2f} GB per step")
    s.close(); del w
