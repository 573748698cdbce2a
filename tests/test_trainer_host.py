"""Host-side logic of training/train.py that needs no GPU: EarlyStopping and the plateau scheduler against the reference's /
torch's own semantics (training/train.py:582-612, :662-666)."""
import math

import os

import pytest

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


def test_early_stopping_is_pickled_under_the_reference_name_and_read_with_weights_only():
    """last_epoch.pt's "early_stopping" entry (training/train.py:215 pickles the OBJECT): written under the reference's class name,
    so the reference unpickles it into its own class, and read back with torch.load(weights_only=True) plus that one allow-listed
    name -- no arbitrary unpickling, no sys.modules alias left behind."""
    import io
    import sys
    import torch
    import types
    from musicfpaugment_amd.training.train import _MainEarlyStopping, _RefEarlyStopping, _atomic_save, _pickling_as_reference_class
    es = EarlyStopping(7, 0.01)
    es(1.0)
    es(1.5)
    with _pickling_as_reference_class(es) as obj:
        buf = io.BytesIO()
        torch.save({"early_stopping": obj, "t": torch.ones(3)}, buf)
    assert "training.train" not in sys.modules and "training" not in sys.modules and b"training.train" in buf.getvalue()
    # a real `training.train` that merely lacks the class (e.g. a user package of that name) is neither replaced nor popped
    real_pkg, real_mod = types.ModuleType("training"), types.ModuleType("training.train")
    real_mod.marker = 1
    sys.modules["training"], sys.modules["training.train"] = real_pkg, real_mod
    try:
        with _pickling_as_reference_class(es) as obj2:
            torch.save({"early_stopping": obj2}, io.BytesIO())
        assert sys.modules["training.train"] is real_mod and sys.modules["training"] is real_pkg
        assert not hasattr(real_mod, "EarlyStopping") and real_mod.marker == 1
    finally:
        sys.modules.pop("training.train", None); sys.modules.pop("training", None)
    buf.seek(0)
    with pytest.raises(Exception):
        torch.load(buf, weights_only=True)                               # not allow-listed: refused
    buf.seek(0)
    with torch.serialization.safe_globals([_RefEarlyStopping, _MainEarlyStopping]):
        got = torch.load(buf, weights_only=True)["early_stopping"]
    assert vars(got) == dict(patience=7, min_delta=0.01, counter=1, best_loss=1.0, early_stop=False)
    assert (type(got).__module__, type(got).__name__) == ("training.train", "EarlyStopping")


def test_checkpoints_are_replaced_atomically(tmp_path):
    """_atomic_save: written to <name>.tmp and os.replace'd -- a failed write leaves the previous file readable."""
    import torch
    from musicfpaugment_amd.training.train import _atomic_save
    path = str(tmp_path / "last_epoch.pt")
    _atomic_save({"epoch": 1}, path)

    class Boom:
        def __reduce__(self):
            raise RuntimeError("disk full")
    with pytest.raises(RuntimeError):
        _atomic_save({"epoch": 2, "x": Boom()}, path)
    assert torch.load(path, weights_only=True)["epoch"] == 1
    assert not os.path.exists(path + ".tmp")                             # ... and no stray temporary


def test_pickling_context_releases_its_lock_when_enter_fails():
    """__exit__ is not called when __enter__ raises: a foreign `training.train.EarlyStopping` whose __new__ fails must not leave the
    module-level lock held (the next checkpoint save would deadlock) nor stub modules behind."""
    import sys
    import types
    from musicfpaugment_amd.training import train as T

    class Hostile:
        def __new__(cls, *a, **k):
            raise TypeError("not constructible")
    pkg, mod = types.ModuleType("training"), types.ModuleType("training.train")
    mod.EarlyStopping = Hostile
    sys.modules["training"], sys.modules["training.train"] = pkg, mod
    try:
        with pytest.raises(TypeError):
            with T._pickling_as_reference_class(EarlyStopping(3, 0.0)):
                pass
        assert T._PICKLE_LOCK.acquire(timeout=1.0)                       # released
        T._PICKLE_LOCK.release()
        assert sys.modules["training.train"] is mod and mod.EarlyStopping is Hostile
    finally:
        sys.modules.pop("training.train", None); sys.modules.pop("training", None)
    with T._pickling_as_reference_class(EarlyStopping(3, 0.0)) as obj:   # and the context still works afterwards
        assert type(obj).__name__ == "EarlyStopping"
    assert "training.train" not in sys.modules
