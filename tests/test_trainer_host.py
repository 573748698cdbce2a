"""Host-side logic of training/train.py that needs no GPU: EarlyStopping and the plateau scheduler against the reference's /
torch's own semantics (training/train.py:582-612, :662-666)."""
import math

import torch

from musicfpaugment_amd.training.train import EarlyStopping, ReduceLROnPlateau


def _reference_early_stopping(patience, min_delta, losses):
    """training/train.py:582-612 restated literally (the checker)."""
    counter, best, stop = 0, float("-inf"), False
    for v in losses:
        if best == float("-inf") or best - v > min_delta:
            best, counter = v, 0
        else:
            counter += 1
            if counter >= patience:
                stop = True
    return counter, best, stop


def test_early_stopping_counts_plateaus_and_nan_like_the_reference():
    cases = [[1.0, 1.0, 1.0, 1.0],                       # exact plateau with min_delta 0: every repeat counts
             [1.0, 0.9, 0.9, 0.95, 0.8, 0.8],
             [1.0, float("nan"), float("nan"), float("nan")],   # a diverged run stops after `patience` epochs
             [1.0, 0.99, 0.98, 0.5]]
    for losses in cases:
        for patience, min_delta in ((2, 0.0), (3, 0.0), (2, 0.01)):
            es = EarlyStopping(patience, min_delta)
            for v in losses:
                es(v)
            want = _reference_early_stopping(patience, min_delta, losses)
            got = (es.counter, es.best_loss, es.early_stop)
            assert got[0] == want[0] and got[2] == want[2], (losses, patience, min_delta, got, want)
            assert got[1] == want[1] or (math.isnan(got[1]) and math.isnan(want[1]))
    assert EarlyStopping().patience == 5 and EarlyStopping().best_loss == float("-inf")      # the reference's defaults


class _Eng:
    lr = 1e-3


def test_plateau_scheduler_matches_torch_and_round_trips_its_state():
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=1e-3)
    ref = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, "min", factor=0.1, patience=2)
    eng = _Eng()
    mine = ReduceLROnPlateau(eng, factor=0.1, patience=2)
    for v in [1.0, 0.9, 0.95, 0.95, 0.95, 0.95, 0.5, 0.6, 0.6, 0.6]:
        ref.step(v)
        mine.step(v)
        assert abs(eng.lr - opt.param_groups[0]["lr"]) < 1e-18
    sd = mine.state_dict()
    assert {"best", "num_bad_epochs", "factor", "patience", "last_epoch"} <= set(sd) and set(sd) <= set(ref.state_dict()) | {"_last_lr"}
    other = ReduceLROnPlateau(_Eng(), factor=0.5, patience=9)
    other.load_state_dict(ref.state_dict())               # torch's own state loads
    assert (other.best, other.num_bad) == (ref.best, ref.num_bad_epochs) and (other.factor, other.patience) == (0.1, 2)
