"""Demucs train step at the bench shape: cost of reading the persistent-LSTM error words before Adam (host catches up with the GPU once per step)."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicfpaugment_amd import ops_demucs as D, synth
from musicfpaugment_amd.constants import DEMUCS_LEARNING_RATE, FACTOR_MAG, FACTOR_SC
from musicfpaugment_amd.ops_demucs_train import DemucsTrainEngine
from musicfpaugment_amd.training.demucs_weights import formula_state_dict
from musicfpaugment_amd.training.loss import MultiResolutionSTFTLoss
dev = torch.device("cuda")
eng = DemucsTrainEngine(formula_state_dict(0), dev, lr=DEMUCS_LEARNING_RATE, precision=1,
                        mrstft=MultiResolutionSTFTLoss(factor_sc=FACTOR_SC, factor_mag=FACTOR_MAG, precision=1).to(dev))
base = synth.batch(16, seed=1); noise = synth.batch(16, seed=2, tonal=False)
clean = torch.from_numpy(np.concatenate([base] * 4)).to(dev)
aug = torch.from_numpy(np.concatenate([(0.7 * base + 0.3 * noise).astype(np.float32)] * 4)).to(dev)
def run(n=6):
    for _ in range(2): eng.train_step(clean, aug)
    torch.cuda.synchronize(); t = time.perf_counter(); host = 0.0
    for _ in range(n):
        h = time.perf_counter(); eng.train_step(clean, aug); host += time.perf_counter() - h
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3, host / n * 1e3
orig = D.lstm_results_ok
print("with check    step %.2f ms, host in train_step %.2f ms" % run())
D.lstm_results_ok = lambda dev: True
print("without check step %.2f ms, host in train_step %.2f ms" % run())
D.lstm_results_ok = orig
print("with check    step %.2f ms, host in train_step %.2f ms" % run())
