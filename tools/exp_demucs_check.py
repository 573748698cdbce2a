import sys, time, json, torch, numpy as np
sys.path.insert(0, "/root/repo")
from musicfpaugment_amd import ops_demucs as D, synth
from musicfpaugment_amd.training.demucs_weights import formula_state_dict
from musicfpaugment_amd.training.model import Demucs
net = Demucs(); net.load_state_dict(formula_state_dict(0)); net = net.cuda().eval()
base = synth.batch(32, seed=1)
wav = torch.from_numpy(np.concatenate([base]*8)).cuda()
def run(n=6):
    for _ in range(2): net(wav)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): net(wav)
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
print("with check   ", run())
orig = D.lstm_results_ok
D.lstm_results_ok = lambda dev: True
print("without check", run())
D.lstm_results_ok = orig
print("with check   ", run())
# host enqueue time of one forward
torch.cuda.synchronize(); t=time.perf_counter(); D.lstm_results_ok = lambda dev: True; net(wav); e=time.perf_counter()-t; torch.cuda.synchronize(); print("host enqueue ms", e*1e3)
from musicfpaugment_amd.afp.audfprint.peak_extractor import Audfprint_peaks
from musicfpaugment_amd import ops_unet
D.lstm_results_ok = orig
ext = Audfprint_peaks(None, device=torch.device("cuda"))
def step():
    den = net(wav)[:, 0].contiguous()
    return ext.find_peaks_batch(den)
def run2(n=6, timer=False):
    for _ in range(2): step()
    torch.cuda.synchronize()
    if timer:
        tm = ops_unet.KernelTimer(); ops_unet.set_timer(tm)
    t=time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/n*1e3
    ops_unet.set_timer(None)
    return dt
print("step (forward + peaks)       ", run2())
print("step with the kernel timer on", run2(timer=True))
