import ctypes as C, sys
sys.path.insert(0, "dia-tts-prune_amd")
import numpy as np, torch
from dia_hip import binding as hb, config as Cf
from dia_hip.engine import DecodeSession, DeviceWeights
from dia_hip.tokens import effective_text, encode_text
from dia_hip.weights import synthetic_state_dict
L = hb.lib(); L.dia_dbg_sstamps.argtypes = [C.c_void_p]
cfg = Cf.dia_1_6b_config() if len(sys.argv) > 1 else Cf.mid_config()
w = DeviceWeights(cfg, synthetic_state_dict(cfg, 1234, 0.02, device=torch.device("cuda:0")), torch.device("cuda:0"))
s = DecodeSession(w, [encode_text(effective_text("[S1] hello there. [S2] hi"), cfg)], kv_dtype="bf16", max_tokens=64, seeds=[1], ignore_eos=True)
s.prefill(); s.decode(30, use_graph=False); s.sync()
buf = np.zeros(16, dtype=np.int64)
assert L.dia_dbg_sstamps(buf.ctypes.data_as(C.c_void_p)) == 0
t = (buf[:8] - buf[0]) / 100.0
names = ["start", "loads+cfg+temp", "top-k", "top-p", "final softmax+argmax", "(barrier)", "fsm", "embed"]
for n, a, b in zip(names[1:], t[:-1], t[1:]): print(f"{n:24s} {b - a:6.2f} us   (at {b:6.2f})")
