import sys; sys.path.insert(0,'.')
import numpy as np, torch
from musicfpaugment_amd import ops, ops_train, synth
from musicfpaugment_amd.ops_train import UNetTrainEngine
from musicfpaugment_amd.training.unet import UNet
from musicfpaugment_amd.training.weights import formula_state_dict
clean = synth.batch(2, seed=500, n=8000)
aug = (0.7 * clean + 0.3 * synth.batch(2, seed=600, n=8000, tonal=False)).astype(np.float32)
cm, cmax = ops.stft_mag(torch.from_numpy(clean).cuda(), torch.float64)
am, amax = ops.stft_mag(torch.from_numpy(aug).cuda(), torch.float64)
clean_spec = ops.normalize_(cm, cmax, per_clip=False)
aug_den = amax.max().expand(2).contiguous()
out = {}
for flag in (False, True):
    ops_train.USE_BF16_DZ = flag
    net = UNet(1, 1, rate=0.0); net.load_state_dict(formula_state_dict(2))
    eng = UNetTrainEngine(net.cuda().train(), lr=1e-3, precision=2, wgrad_precision=2)
    pred = eng.forward(spec64=am, denom=aug_den)
    loss, dpred = eng.l1_loss(pred, clean_spec)
    eng.backward(dpred)
    out[flag] = {k: v.clone() for k, v in eng.named_grads().items()}
for k in out[False]:
    a, b = out[False][k].double(), out[True][k].double()
    r = float((a - b).abs().sum() / a.abs().sum().clamp_min(1e-30))
    print(f"{k:55s} rel {r:.3e}")
