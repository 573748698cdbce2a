"""`Trainer` / `EarlyStopping` on MI355X -- mirror of the reference's training/train.py: the spec branch (UNet) and the audio
branch (Demucs: L1 + MultiResolutionSTFTLoss, train.py:275-312, through ops_demucs_train.DemucsTrainEngine).

Reference: Trainer.train_epoch (:245-359), validation_epoch (:361-468), EarlyStopping (:582-612),
__main__ wiring (:645-667: UNet(1,1,rate), L1Loss, Adam(lr 1e-3, betas (0.9, 0.999)),
ReduceLROnPlateau(min, factor 0.1, patience 10)).

What is kept: constructor argument names, `train_epoch(epoch) -> {"loss": float}`,
`validation_epoch() -> ({"loss": float}, {"psnr": float})`, the loop quirk (range(1, steps) iterations,
loss divided by `steps`, train.py:257,341), the epoch loop `range(epoch_start, nb_epochs)` with its resume quirk (a resumed run
starts again AT the saved epoch number, train.py:134-136,174), and checkpoint dictionaries with the reference's keys and
value formats (train.py:197-221): `best_epoch.pt` = {model_state_dict, best_val_loss}; `last_epoch.pt` = {epoch,
model_state_dict, optimizer_state_dict (a torch.optim.Adam state_dict over model.parameters()), scheduler_state_dict
(ReduceLROnPlateau's keys), early_stopping (the object), train_loss, val_losses, best_val_loss} -- either side loads the other's.
What is different: no tf.data / tensorboard / Progbar; the loaders are plain Python iterators yielding
`(clean (B,T,1) or (B,T), augmented (B,T,1) or (B,T))` float32 tensors; the step itself runs through
UNetTrainEngine (HIP kernels, no autograd); one process per GPU with RCCL gradient all-reduce when
torch.distributed is initialised (the reference is single-GPU).
"""
from __future__ import annotations

import os
from typing import Any, Dict, Iterator, Optional, Tuple

import torch

from .. import ops
from ..constants import FACTOR_MAG, FACTOR_SC, LEARNING_RATE, TRAIN_STEPS, VAL_STEPS
from ..ops_train import UNetTrainEngine
from .unet import UNet


class EarlyStopping:
    """training/train.py:582-612, branch for branch: an epoch that does not improve the best loss by more than `min_delta`
    -- an exact plateau and a NaN loss included -- counts; `patience` such epochs in a row stop the run."""

    def __init__(self, patience: int = 5, min_delta: float = 0.0):
        self.patience, self.min_delta = patience, min_delta
        self.counter = 0
        self.best_loss = float("-inf")
        self.early_stop = False

    def __call__(self, val_loss: float) -> None:
        if self.best_loss == float("-inf") or self.best_loss - val_loss > self.min_delta:
            self.best_loss = val_loss
            self.counter = 0
        else:
            self.counter += 1
            if self.counter >= self.patience:
                self.early_stop = True


class _RefEarlyStopping(EarlyStopping):
    """EarlyStopping under the reference's pickle name (training.train.EarlyStopping, train.py:582-612, same attributes): what
    `last_epoch.pt` holds under "early_stopping", so that the reference unpickles this package's checkpoints into ITS class and
    this package reads the reference's with `torch.load(weights_only=True)` plus this one allow-listed name."""


_RefEarlyStopping.__module__ = "training.train"
_RefEarlyStopping.__qualname__ = _RefEarlyStopping.__name__ = "EarlyStopping"


class _MainEarlyStopping(EarlyStopping):
    """The same class as pickled by a reference run started with `python -m training.train` (train.py:697): there the defining
    module is `__main__`.  Allow-listed for loading only."""


_MainEarlyStopping.__module__ = "__main__"
_MainEarlyStopping.__qualname__ = _MainEarlyStopping.__name__ = "EarlyStopping"


_PICKLE_LOCK = __import__("threading").Lock()


class _pickling_as_reference_class:
    """Context: `training.train.EarlyStopping` resolves while pickle WRITES (it looks the class up by name to verify it).  Inside a
    process that has the reference imported that is the reference's own class and nothing is touched.  Otherwise the missing
    names -- and only those -- are provided for the duration of the write and removed again: an existing `training` /
    `training.train` module is never replaced or popped (a real module that merely lacks the attribute gets it set, then deleted).
    Serialised by a lock (DataLoader workers importing `training.*` see either the old or the restored state)."""

    def __init__(self, es: EarlyStopping):
        self.es = es

    def __enter__(self):
        import sys
        import types
        _PICKLE_LOCK.acquire()
        self.added_modules, self.added_attr = [], None
        try:
            return self._enter(sys, types)
        except BaseException:
            self.__exit__(None, None, None)          # undo the module additions and release the lock: __exit__ is not called when __enter__ raises
            raise

    def _enter(self, sys, types):
        mod = sys.modules.get("training.train")
        if mod is not None and hasattr(mod, "EarlyStopping"):
            cls = mod.EarlyStopping
        else:
            cls = _RefEarlyStopping
            if "training" not in sys.modules:
                sys.modules["training"] = types.ModuleType("training")
                self.added_modules.append("training")
            if mod is None:
                mod = types.ModuleType("training.train")
                sys.modules["training.train"] = mod
                self.added_modules.append("training.train")
            mod.EarlyStopping = cls
            self.added_attr = mod
        es = self.es
        obj = cls.__new__(cls)
        obj.__dict__.update(patience=es.patience, min_delta=es.min_delta, counter=es.counter, best_loss=es.best_loss, early_stop=es.early_stop)
        return obj

    def __exit__(self, *exc):
        import sys
        try:
            if self.added_attr is not None and "training.train" not in self.added_modules:
                try:
                    delattr(self.added_attr, "EarlyStopping")
                except AttributeError:
                    pass
            for k in self.added_modules:
                sys.modules.pop(k, None)
        finally:
            _PICKLE_LOCK.release()
        return False


def _atomic_save(obj, path: str) -> None:
    """torch.save to `<path>.tmp`, then os.replace: a crash mid-write never leaves an unreadable resume file."""
    tmp = path + ".tmp"
    try:
        torch.save(obj, tmp)
        os.replace(tmp, path)
    except BaseException:
        try:
            os.remove(tmp)                           # a failed write leaves no stray .tmp behind
        except OSError:
            pass
        raise


class ReduceLROnPlateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau(mode="min", factor, patience) on the engine's lr
    (default relative threshold 1e-4, cooldown 0), as wired at train.py:662-666."""

    def __init__(self, engine: UNetTrainEngine, factor: float = 0.1, patience: int = 10, threshold: float = 1e-4):
        self.engine, self.factor, self.patience, self.threshold = engine, factor, patience, threshold
        self.best = float("inf")
        self.num_bad = 0
        self.last_epoch = 0

    def step(self, metric: float) -> None:
        if metric < self.best * (1.0 - self.threshold):
            self.best, self.num_bad = metric, 0
        else:
            self.num_bad += 1
        if self.num_bad > self.patience:
            self.engine.lr *= self.factor
            self.num_bad = 0
        self.last_epoch += 1

    def state_dict(self) -> Dict[str, Any]:
        """The keys of torch's ReduceLROnPlateau.state_dict() that carry state (train.py:213), so either side loads the other's."""
        return {"factor": self.factor, "patience": self.patience, "threshold": self.threshold, "threshold_mode": "rel", "mode": "min",
                "cooldown": 0, "cooldown_counter": 0, "min_lrs": [0], "eps": 1e-8, "best": self.best, "num_bad_epochs": self.num_bad,
                "last_epoch": self.last_epoch, "_last_lr": [self.engine.lr]}

    def load_state_dict(self, sd: Dict[str, Any]) -> None:
        self.best, self.num_bad = float(sd["best"]), int(sd["num_bad_epochs"])
        self.last_epoch = int(sd.get("last_epoch", 0))
        self.factor, self.patience = float(sd.get("factor", self.factor)), int(sd.get("patience", self.patience))
        self.threshold = float(sd.get("threshold", self.threshold))


def _dist_group(group=None):
    """(torch.distributed, world size, rank) of `group`, or (None, 1, 0) outside a process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_world_size(group), dist.get_rank(group)
    return None, 1, 0


def _global_max(clip_max: torch.Tensor, group=None) -> torch.Tensor:
    """spectrogram() divides by ONE max over the whole batch (visualisation.py:29); with the batch sharded over
    ranks that is a scalar MAX all-reduce over the engine's process group, so single-device results are reproduced."""
    m = clip_max.max()
    dist, world, _ = _dist_group(group)
    if world > 1:
        dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
    return m


class Trainer:
    def __init__(self, model, train_loader: Iterator, val_loader: Optional[Iterator] = None,
                 learning_rate: float = LEARNING_RATE, train_steps: int = TRAIN_STEPS, val_steps: int = VAL_STEPS,
                 device="cuda", ckpt_path: Optional[str] = None, input_type: str = "spec",
                 scheduler_factor: float = 0.1, scheduler_patience: int = 10, early_stop_patience: int = 20,
                 precision: int = 0, sync_bn: bool = False, factor_sc: float = FACTOR_SC, factor_mag: float = FACTOR_MAG,
                 betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, early_stop_min_delta: float = 0.0,
                 nb_epochs: Optional[int] = None, process_group=None):
        """`precision` (not in the reference): 0 = exact fp32 products on the fp32 matrix cores, 1 = bf16x3 (2.4x the
        step rate; gradients deviate ~1e-2 relative from fp32 autograd, see tests/test_gpu_train.py).  `sync_bn` (multi-GPU):
        BatchNorm statistics over the global batch, so that N GPUs x B/N clips reproduce the reference's single-GPU step on
        B clips (tests/test_gpu_dist.py); default per-GPU statistics."""
        if input_type not in ("spec", "audio"):
            raise ValueError("input_type must be 'spec' (UNet) or 'audio' (Demucs)")
        self.input_type = input_type
        self.group = process_group                              # the data-parallel ranks (None = the default group)
        self.device = torch.device(device)
        self.model = model.to(self.device)
        if input_type == "audio":
            from ..ops_demucs_train import DemucsTrainEngine
            from .loss import MultiResolutionSTFTLoss
            self.mrsl = MultiResolutionSTFTLoss(factor_sc=factor_sc, factor_mag=factor_mag, precision=precision).to(self.device)   # train.py:652-655
            self.engine = DemucsTrainEngine(self.model.state_dict(), self.device, lr=learning_rate, betas=tuple(betas), eps=eps,
                                            precision=precision, mrstft=self.mrsl, module=self.model, dist_group=process_group)
        else:
            self.engine = UNetTrainEngine(self.model, lr=learning_rate, betas=tuple(betas), eps=eps, precision=precision,
                                          sync_bn=sync_bn, process_group=process_group)
        self.scheduler = ReduceLROnPlateau(self.engine, scheduler_factor, scheduler_patience)
        self.early_stopping = EarlyStopping(early_stop_patience, early_stop_min_delta)
        self.nb_epochs = nb_epochs
        self.save = True                                        # from_reference: the reference's `save` flag
        self.train_loader_iter, self.val_loader_iter = train_loader, val_loader
        self.train_steps, self.val_steps = train_steps, val_steps
        self.ckpt_path = ckpt_path
        self.epoch = 0                                          # the last epoch that ran
        self.epoch_start = 1                                    # train.py:127; a resume sets it to the SAVED epoch number (:134-136)
        self.best_val_loss = float("inf")                       # the reference's min_valid_loss
        self.losses: Dict[str, list] = {"train": [], "val": []}

    @classmethod
    def from_reference(cls, model, train_loader, train_steps: int, val_loader, val_steps: int, criterion: Dict[str, Any],
                       optimizer, scheduler, early_stopping, nb_epochs: int, device, metadatas: Optional[Dict[str, str]] = None,
                       checkpoint: Optional[str] = None, monitoring: bool = False, save: bool = False,
                       input_type: str = "audio", *, precision: int = 0, sync_bn: bool = False) -> "Trainer":
        """The reference's constructor call (training/train.py:52-70, as `__main__` makes it, :676-693) mapped onto this class:
        the torch objects are read for their hyper-parameters -- Adam's lr / betas / eps, ReduceLROnPlateau's factor / patience,
        EarlyStopping's patience / min_delta, the MultiResolutionSTFTLoss factors -- because the optimiser step, the schedule
        and the losses run on the engine's flat buffers, not on `model.parameters()`.  `monitoring` (TensorBoard) is not built;
        checkpoints are written when `save` is set, and an existing `last_epoch.pt` under `checkpoint` is resumed (:130-161)."""
        if monitoring:
            raise NotImplementedError("TensorBoard monitoring is outside the hot path")
        if type(optimizer).__name__ != "Adam":
            raise ValueError("the reference trains with torch.optim.Adam (train.py:661); no other optimiser is built")
        g = optimizer.param_groups[0]
        if g.get("weight_decay", 0) or g.get("amsgrad", False):
            raise ValueError("Adam with weight decay / amsgrad is not built (the reference uses neither)")
        mrsl = criterion.get("mrsl") if criterion else None
        ck = checkpoint if checkpoint and (save or os.path.exists(os.path.join(checkpoint, "last_epoch.pt"))) else None
        self = cls(model, iter(train_loader), iter(val_loader) if val_loader is not None else None, learning_rate=float(g["lr"]),
                   train_steps=train_steps, val_steps=val_steps, device=device, ckpt_path=ck, input_type=input_type,
                   scheduler_factor=float(getattr(scheduler, "factor", 0.1)), scheduler_patience=int(getattr(scheduler, "patience", 10)),
                   early_stop_patience=int(getattr(early_stopping, "patience", 20)),
                   early_stop_min_delta=float(getattr(early_stopping, "min_delta", 0.0)), precision=precision, sync_bn=sync_bn,
                   factor_sc=float(getattr(mrsl, "factor_sc", FACTOR_SC)), factor_mag=float(getattr(mrsl, "factor_mag", FACTOR_MAG)),
                   betas=tuple(float(b) for b in g["betas"]), eps=float(g["eps"]), nb_epochs=int(nb_epochs))
        self.metadatas = dict(metadatas or {})
        self.save = bool(save)
        return self

    # ------------------------------------------------------------------ one batch
    def _specs(self, clean_audios: torch.Tensor, augmented_audios: torch.Tensor):
        clean = clean_audios.to(self.device, torch.float32)
        aug = augmented_audios.to(self.device, torch.float32)
        clean = clean.squeeze(-1) if clean.dim() == 3 else clean          # (B,T,1) -> (B,T)   train.py:260
        aug = aug.squeeze(-1) if aug.dim() == 3 else aug
        cm, cmax = ops.stft_mag(clean.contiguous(), torch.float64)
        am, amax = ops.stft_mag(aug.contiguous(), torch.float64)
        B = cm.shape[0]
        ops.normalize_(cm, _global_max(cmax, self.group).expand(B).contiguous(), per_clip=True)     # clean_specs, float64 target
        return am, _global_max(amax, self.group).expand(B).contiguous(), cm

    def _waves(self, clean_audios: torch.Tensor, augmented_audios: torch.Tensor):
        clean = clean_audios.to(self.device, torch.float32)
        aug = augmented_audios.to(self.device, torch.float32)
        clean = clean.squeeze(-1) if clean.dim() == 3 else clean          # (B,T,1) -> (B,T)   train.py:260-262
        aug = aug.squeeze(-1) if aug.dim() == 3 else aug
        return clean.contiguous(), aug.contiguous()

    def train_step(self, clean_audios, augmented_audios) -> torch.Tensor:
        if self.input_type == "audio":
            clean, aug = self._waves(clean_audios, augmented_audios)
            return self.engine.train_step(clean, aug)
        am, aden, clean_spec = self._specs(clean_audios, augmented_audios)
        return self.engine.train_step(am, aden, clean_spec)

    # ------------------------------------------------------------------ epochs
    def train_epoch(self, epoch: int) -> Dict[str, Any]:
        self.model.train()
        total = torch.zeros(1, dtype=torch.float64, device=self.device)
        parts = torch.zeros(3, dtype=torch.float64, device=self.device)
        for _ in range(1, self.train_steps):                    # reference quirk: steps-1 iterations (train.py:257)
            clean, aug = next(self.train_loader_iter)
            total += self.train_step(clean, aug)                # loss stays on the device: no per-step .item() sync
            if self.input_type == "audio":
                parts += torch.stack([v.double() for v in self.engine.last_losses])
        out = {"loss": float(total.item()) / self.train_steps}  # ... divided by steps (train.py:341)
        if self.input_type == "audio":                          # train.py:347-356
            l1, sc, mag = (parts / self.train_steps).tolist()
            out.update({"l1_loss": l1, "sc_loss": sc, "mag_loss": mag})
        return out

    @torch.no_grad()
    def validation_epoch(self) -> Tuple[Dict[str, Any], Dict[str, Any]]:
        if self.val_loader_iter is None:
            raise ValueError("no validation loader")
        self.engine.sync_to_module()
        self.model.eval()
        total = torch.zeros(1, dtype=torch.float64, device=self.device)
        psnr_total = 0.0
        if self.input_type == "audio":
            return self._validation_epoch_audio()
        for _ in range(1, self.val_steps):
            clean, aug = next(self.val_loader_iter)
            am, aden, clean_spec = self._specs(clean, aug)
            pred = self.model.denoise_spectrogram(am, aden, per_clip=True)
            loss, _ = self.engine.l1_loss(pred, clean_spec, want_grad=False)
            total += loss
            mse = torch.mean((pred.double() - clean_spec) ** 2)
            rng = clean_spec.max() - clean_spec.min()
            psnr_total += float(10.0 * torch.log10(rng * rng / mse))
        val_loss, psnr = self._mean_over_ranks([float(total.item()) / self.val_steps, psnr_total / self.val_steps])
        self.scheduler.step(val_loss)                           # train.py:462
        self.model.train()
        return {"loss": val_loss}, {"psnr": psnr}

    def _validation_epoch_audio(self) -> Tuple[Dict[str, Any], Dict[str, Any]]:
        """train.py:418-445: predicted = model(augmented); L1 + sc + mag; PSNR on the waveforms."""
        parts = torch.zeros(3, dtype=torch.float64, device=self.device)
        psnr_total = 0.0
        for _ in range(1, self.val_steps):
            clean, aug = self._waves(*next(self.val_loader_iter))
            pred = self.model(aug)[:, 0].contiguous()
            l1, sc, mag, _ = self.engine.loss_and_grad(pred, clean)
            parts += torch.stack([l1.double(), sc.double(), mag.double()])
            mse = torch.mean((pred.double() - clean.double()) ** 2)
            rng = clean.max() - clean.min()
            psnr_total += float(10.0 * torch.log10(rng.double() ** 2 / mse))
        l1, sc, mag, psnr = self._mean_over_ranks((parts / self.val_steps).tolist() + [psnr_total / self.val_steps])
        val_loss = l1 + sc + mag
        self.scheduler.step(val_loss)
        self.model.train()
        return ({"loss": val_loss, "l1_loss": l1, "sc_loss": sc, "mag_loss": mag}, {"psnr": psnr})

    def _mean_over_ranks(self, values):
        """Every rank validates its own shard; the numbers that drive the schedule, early stopping and the checkpoints are the
        MEAN over the ranks of the engine's process group, so all replicas take the same decisions in the same epoch (otherwise
        Adam would run with per-rank learning rates and one rank could leave the loop while the others wait in an all-reduce)."""
        dist, world, _ = _dist_group(self.group)
        if world == 1:
            return list(values)
        t = torch.tensor(list(values), dtype=torch.float64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return (t / world).tolist()

    @torch.no_grad()
    def start_epoch(self) -> Tuple[Dict[str, Any], Dict[str, Any]]:
        """train.py:470-578: the losses and PSNR of the UN-denoised validation inputs (augmented vs clean) before training --
        all `val_steps` batches, unlike the epochs' steps-1.  Monitoring only (plain device arithmetic, no model); the
        reference prints the two dictionaries, this also returns them."""
        if self.val_loader_iter is None:
            raise ValueError("no validation loader")
        total, parts, psnr_total = 0.0, [0.0, 0.0, 0.0], 0.0
        for _ in range(self.val_steps):
            clean, aug = next(self.val_loader_iter)
            if self.input_type == "audio":
                clean, aug = self._waves(clean, aug)
                l1, sc, mag, _ = self.engine.loss_and_grad(aug, clean)
                vals = [float(l1), float(sc), float(mag)]
                parts = [p + v for p, v in zip(parts, vals)]
                total += sum(vals)
                a, c = aug.double(), clean.double()
            else:
                am, aden, c = self._specs(clean, aug)
                a = am / aden[:, None, None]
                total += float(torch.mean(torch.abs(a - c)))
            rng = c.max() - c.min()
            psnr_total += float(10.0 * torch.log10(rng * rng / torch.mean((a - c) ** 2)))
        losses: Dict[str, Any] = {"loss": total / self.val_steps}
        if self.input_type == "audio":
            losses.update({k: v / self.val_steps for k, v in zip(["l1_loss", "sc_loss", "mag_loss"], parts)})
        metrics = {"psnr": psnr_total / self.val_steps}
        print(f"\nStart Loss: {losses}")
        print(f"\nStart Metrics: {metrics}")
        return losses, metrics

    # ------------------------------------------------------------------ checkpoints (train.py:197-221, :130-161)
    def _optimizer_state_dict(self) -> Dict[str, Any]:
        """The engine's flat Adam state as `torch.optim.Adam(model.parameters()).state_dict()` would hold it (train.py:212)."""
        names = [n for n, _ in self.model.named_parameters()]
        m, v = self.engine.named_moments()
        state = {}
        if self.engine.step_count > 0:
            for i, n in enumerate(names):
                state[i] = {"step": torch.tensor(float(self.engine.step_count)), "exp_avg": m[n].detach().clone(),
                            "exp_avg_sq": v[n].detach().clone()}
        group = {"lr": self.engine.lr, "betas": tuple(self.engine.betas), "eps": self.engine.eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group]}

    def _load_optimizer_state_dict(self, opt: Dict[str, Any]) -> None:
        if "state" not in opt or "param_groups" not in opt:
            raise ValueError("optimizer_state_dict is not a torch.optim.Adam state_dict ('state' / 'param_groups' missing): "
                             "checkpoints written by round 1 of this package used a private format and cannot be resumed")
        names = [n for n, _ in self.model.named_parameters()]
        g = opt["param_groups"][0]
        if len(g["params"]) != len(names):
            raise ValueError(f"optimizer state for {len(g['params'])} parameters, the model has {len(names)}")
        self.engine.lr = float(g["lr"])
        st = opt["state"]
        if len(st) == 0:                                         # saved before the first step
            self.engine.flat_m.zero_(); self.engine.flat_v.zero_(); self.engine.step_count = 0
            return
        zeros = {n: torch.zeros_like(p) for n, p in self.model.named_parameters()}
        m = {n: st[k]["exp_avg"] if k in st else zeros[n] for n, k in zip(names, g["params"])}
        v = {n: st[k]["exp_avg_sq"] if k in st else zeros[n] for n, k in zip(names, g["params"])}
        self.engine.load_named_moments(m, v)
        self.engine.step_count = int(float(next(iter(st.values()))["step"]))       # an int in torch 1.11, a tensor since 1.12

    def save_checkpoint(self, val_loss: float, train_loss: Optional[Dict[str, Any]] = None,
                        val_losses: Optional[Dict[str, Any]] = None) -> None:
        """train.py:188-221: first the best model (updating the running minimum), then `last_epoch.pt` with that minimum.  Rank 0
        writes; the other ranks only keep the same minimum (they hold the same loss, _mean_over_ranks) and wait at a barrier."""
        best = val_loss < self.best_val_loss
        if best:
            self.best_val_loss = val_loss
        if self.ckpt_path is None or not self.save:
            return
        dist, world, rank = _dist_group(self.group)
        self.engine.sync_to_module()          # every rank: whoever reads trainer.model afterwards sees the trained weights
        err = None
        if rank == 0:
            try:
                os.makedirs(self.ckpt_path, exist_ok=True)
                sd = self.model.state_dict()
                if best:
                    _atomic_save({"model_state_dict": sd, "best_val_loss": self.best_val_loss}, os.path.join(self.ckpt_path, "best_epoch.pt"))
                with _pickling_as_reference_class(self.early_stopping) as es_obj:
                    _atomic_save({"epoch": self.epoch, "model_state_dict": sd, "optimizer_state_dict": self._optimizer_state_dict(),
                                  "scheduler_state_dict": self.scheduler.state_dict(), "early_stopping": es_obj,
                                  "train_loss": train_loss if train_loss is not None else {"loss": self.losses["train"][-1] if self.losses["train"] else None},
                                  "val_losses": val_losses if val_losses is not None else {"loss": val_loss},
                                  "best_val_loss": self.best_val_loss, "losses": self.losses},
                                 os.path.join(self.ckpt_path, "last_epoch.pt"))
            except Exception as e:            # the other ranks are waiting: tell them before raising
                err = e
        if world > 1:
            flag = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=self.device)
            dist.broadcast(flag, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            if err is None and int(flag.item()):
                raise RuntimeError("rank 0 failed to write the checkpoint")
        if err is not None:
            raise err

    def load_checkpoint(self) -> bool:
        """train.py:130-161.  Read with `torch.load(weights_only=True)`: tensors and plain containers, plus ONE allow-listed class
        name -- training.train.EarlyStopping (`__main__.EarlyStopping` when the reference ran as `python -m training.train`), the object
        the reference pickles under "early_stopping" (train.py:215) and this
        package writes under the same name (_RefEarlyStopping) -- so nothing in a `last_epoch.pt` can run code here, and the file
        goes both ways: the reference's loader unpickles this package's checkpoint into its own EarlyStopping class."""
        path = None if self.ckpt_path is None else os.path.join(self.ckpt_path, "last_epoch.pt")
        if path is None or not os.path.exists(path):
            return False
        with torch.serialization.safe_globals([_RefEarlyStopping, _MainEarlyStopping]):
            ck = torch.load(path, map_location=self.device, weights_only=True)
        self.model.load_state_dict(ck["model_state_dict"])
        self.engine.load_from_module()
        self._load_optimizer_state_dict(ck["optimizer_state_dict"])
        self.scheduler.load_state_dict(ck["scheduler_state_dict"])
        es = ck["early_stopping"]
        get = (lambda k, d: es.get(k, d)) if isinstance(es, dict) else (lambda k, d: getattr(es, k, d))
        self.early_stopping.counter, self.early_stopping.best_loss = int(get("counter", 0)), float(get("best_loss", float("-inf")))
        self.early_stopping.early_stop = bool(get("early_stop", False))
        self.early_stopping.patience, self.early_stopping.min_delta = int(get("patience", self.early_stopping.patience)), float(get("min_delta", self.early_stopping.min_delta))
        self.epoch_start = int(ck["epoch"])                      # the reference runs the saved epoch number again (train.py:134-136,174)
        self.epoch = self.epoch_start - 1
        self.best_val_loss = float(ck["best_val_loss"])
        self.losses = ck.get("losses", self.losses)
        return True

    def training_loop(self, nb_epochs: Optional[int] = None) -> None:
        """train.py:171-243 without the logging side outputs: `for epoch in range(epoch_start, nb_epochs)` -- epochs 1 .. nb_epochs-1,
        like the reference -- early stopping tested at the top of an epoch, best then last checkpoint after it."""
        nb_epochs = self.nb_epochs if nb_epochs is None else nb_epochs
        if nb_epochs is None:
            raise ValueError("nb_epochs was given neither here nor to the constructor")
        self.load_checkpoint()
        for epoch in range(self.epoch_start, nb_epochs):
            if self.early_stopping.early_stop:
                break
            self.epoch = epoch
            train_loss = self.train_epoch(epoch)
            self.losses["train"].append(train_loss["loss"])
            if self.val_loader_iter is None:
                continue
            val, _ = self.validation_epoch()
            self.losses["val"].append(val["loss"])
            self.early_stopping(val["loss"])
            self.save_checkpoint(val["loss"], train_loss, val)
