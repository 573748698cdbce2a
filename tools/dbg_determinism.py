import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
if len(sys.argv) > 1:
    from musicfpaugment_amd import _lib
    _lib.set_library_path(sys.argv[1])
from musicfpaugment_amd import ops_demucs as D, synth
from musicfpaugment_amd.training.model import Demucs
from musicfpaugment_amd.training.demucs_weights import formula_state_dict
B = 256
wav = np.stack([synth.clip(5000 + i, tonal=(i % 4 != 0)) for i in range(B)])
x = torch.from_numpy(wav).cuda()
net = Demucs(); net.load_state_dict(formula_state_dict(0)); net = net.cuda().eval()
def dig(t): return hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()[:10]
def run(label, n=8):
    ds = [dig(net(x)) for _ in range(n)]
    print(label, len(set(ds)), "distinct of", n, ds[:4], flush=True)
run("all on")
D.FUSE_FIRST_LEVEL = False; run("first level unfused"); D.FUSE_FIRST_LEVEL = True
D.FUSE_LAST_LEVEL = False; run("last level unfused"); D.FUSE_LAST_LEVEL = True
D.PERSISTENT_LSTM = False; run("per-step lstm"); D.PERSISTENT_LSTM = True
D.PRESPLIT_WEIGHTS = False; net._packed = None; run("weights split on the fly"); D.PRESPLIT_WEIGHTS = True; net._packed = None
print("err", D.lstm_seq_error())
