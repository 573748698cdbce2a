"""`Trainer` / `EarlyStopping` on MI355X -- mirror of the reference's training/train.py: the spec branch (UNet) and the audio
branch (Demucs: L1 + MultiResolutionSTFTLoss, train.py:275-312, through ops_demucs_train.DemucsTrainEngine).

Reference: Trainer.train_epoch (:245-359), validation_epoch (:361-468), EarlyStopping (:582-612),
__main__ wiring (:645-667: UNet(1,1,rate), L1Loss, Adam(lr 1e-3, betas (0.9, 0.999)),
ReduceLROnPlateau(min, factor 0.1, patience 10)).

What is kept: constructor argument names, `train_epoch(epoch) -> {"loss": float}`,
`validation_epoch() -> ({"loss": float}, {"psnr": float})`, the loop quirk (range(1, steps) iterations,
loss divided by `steps`, train.py:257,341), checkpoint dictionaries with the reference's keys
(`{"model_state_dict": ...}` for best, + optimizer/epoch for last, train.py:197-221).
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
    """training/train.py:582-612: stop after `patience` epochs without a `min_delta` improvement."""

    def __init__(self, patience: int = 20, min_delta: float = 0.0):
        self.patience, self.min_delta = patience, min_delta
        self.counter = 0
        self.best_loss: Optional[float] = None
        self.early_stop = False

    def __call__(self, val_loss: float) -> None:
        if self.best_loss is None:
            self.best_loss = val_loss
        elif self.best_loss - val_loss > self.min_delta:
            self.best_loss = val_loss
            self.counter = 0
        elif self.best_loss - val_loss < self.min_delta:
            self.counter += 1
            if self.counter >= self.patience:
                self.early_stop = True


class ReduceLROnPlateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau(mode="min", factor, patience) on the engine's lr
    (default relative threshold 1e-4, cooldown 0), as wired at train.py:662-666."""

    def __init__(self, engine: UNetTrainEngine, factor: float = 0.1, patience: int = 10, threshold: float = 1e-4):
        self.engine, self.factor, self.patience, self.threshold = engine, factor, patience, threshold
        self.best = float("inf")
        self.num_bad = 0

    def step(self, metric: float) -> None:
        if metric < self.best * (1.0 - self.threshold):
            self.best, self.num_bad = metric, 0
        else:
            self.num_bad += 1
        if self.num_bad > self.patience:
            self.engine.lr *= self.factor
            self.num_bad = 0


def _global_max(clip_max: torch.Tensor) -> torch.Tensor:
    """spectrogram() divides by ONE max over the whole batch (visualisation.py:29); with the batch sharded over
    ranks that is a scalar MAX all-reduce, so single-device results are reproduced."""
    m = clip_max.max()
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return m


class Trainer:
    def __init__(self, model, train_loader: Iterator, val_loader: Optional[Iterator] = None,
                 learning_rate: float = LEARNING_RATE, train_steps: int = TRAIN_STEPS, val_steps: int = VAL_STEPS,
                 device="cuda", ckpt_path: Optional[str] = None, input_type: str = "spec",
                 scheduler_factor: float = 0.1, scheduler_patience: int = 10, early_stop_patience: int = 20,
                 precision: int = 0, sync_bn: bool = False, factor_sc: float = FACTOR_SC, factor_mag: float = FACTOR_MAG,
                 betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, early_stop_min_delta: float = 0.0,
                 nb_epochs: Optional[int] = None):
        """`precision` (not in the reference): 0 = exact fp32 products on the fp32 matrix cores, 1 = bf16x3 (2.4x the
        step rate; gradients deviate ~1e-2 relative from fp32 autograd, see tests/test_gpu_train.py).  `sync_bn` (multi-GPU):
        BatchNorm statistics over the global batch, so that N GPUs x B/N clips reproduce the reference's single-GPU step on
        B clips (tests/test_gpu_dist.py); default per-GPU statistics."""
        if input_type not in ("spec", "audio"):
            raise ValueError("input_type must be 'spec' (UNet) or 'audio' (Demucs)")
        self.input_type = input_type
        self.device = torch.device(device)
        self.model = model.to(self.device)
        if input_type == "audio":
            from ..ops_demucs_train import DemucsTrainEngine
            from .loss import MultiResolutionSTFTLoss
            self.mrsl = MultiResolutionSTFTLoss(factor_sc=factor_sc, factor_mag=factor_mag, precision=precision).to(self.device)   # train.py:652-655
            self.engine = DemucsTrainEngine(self.model.state_dict(), self.device, lr=learning_rate, betas=tuple(betas), eps=eps,
                                            precision=precision, mrstft=self.mrsl, module=self.model)
        else:
            self.engine = UNetTrainEngine(self.model, lr=learning_rate, betas=tuple(betas), eps=eps, precision=precision,
                                          sync_bn=sync_bn)
        self.scheduler = ReduceLROnPlateau(self.engine, scheduler_factor, scheduler_patience)
        self.early_stopping = EarlyStopping(early_stop_patience, early_stop_min_delta)
        self.nb_epochs = nb_epochs
        self.save = True                                        # from_reference: the reference's `save` flag
        self.train_loader_iter, self.val_loader_iter = train_loader, val_loader
        self.train_steps, self.val_steps = train_steps, val_steps
        self.ckpt_path = ckpt_path
        self.epoch = 0
        self.best_val_loss = float("inf")
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
        ops.normalize_(cm, _global_max(cmax).expand(B).contiguous(), per_clip=True)     # clean_specs, float64 target
        return am, _global_max(amax).expand(B).contiguous(), cm

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
        val_loss = float(total.item()) / self.val_steps
        self.scheduler.step(val_loss)                           # train.py:462
        self.model.train()
        return {"loss": val_loss}, {"psnr": psnr_total / self.val_steps}

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
        l1, sc, mag = (parts / self.val_steps).tolist()
        val_loss = l1 + sc + mag
        self.scheduler.step(val_loss)
        self.model.train()
        return ({"loss": val_loss, "l1_loss": l1, "sc_loss": sc, "mag_loss": mag}, {"psnr": psnr_total / self.val_steps})

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
    def save_checkpoint(self, val_loss: float) -> None:
        if self.ckpt_path is None or not self.save:
            return
        os.makedirs(self.ckpt_path, exist_ok=True)
        self.engine.sync_to_module()
        sd = self.model.state_dict()
        torch.save({"epoch": self.epoch, "model_state_dict": sd,
                    "optimizer_state_dict": {"exp_avg": self.engine.flat_m, "exp_avg_sq": self.engine.flat_v,
                                             "step": self.engine.step_count, "lr": self.engine.lr},
                    "losses": self.losses, "best_val_loss": self.best_val_loss},
                   os.path.join(self.ckpt_path, "last_epoch.pt"))
        if val_loss < self.best_val_loss:
            self.best_val_loss = val_loss
            torch.save({"model_state_dict": sd, "best_val_loss": val_loss}, os.path.join(self.ckpt_path, "best_epoch.pt"))

    def load_checkpoint(self) -> bool:
        path = None if self.ckpt_path is None else os.path.join(self.ckpt_path, "last_epoch.pt")
        if path is None or not os.path.exists(path):
            return False
        ck = torch.load(path, map_location=self.device)
        self.model.load_state_dict(ck["model_state_dict"])
        self.engine.load_from_module()
        opt = ck["optimizer_state_dict"]
        self.engine.flat_m.copy_(opt["exp_avg"]); self.engine.flat_v.copy_(opt["exp_avg_sq"])
        self.engine.step_count, self.engine.lr = opt["step"], opt["lr"]
        self.epoch, self.losses, self.best_val_loss = ck["epoch"], ck["losses"], ck["best_val_loss"]
        return True

    def training_loop(self, nb_epochs: Optional[int] = None) -> None:
        """train.py:171-243 without the logging side outputs (`nb_epochs` defaults to the constructor's, as in the reference)."""
        nb_epochs = self.nb_epochs if nb_epochs is None else nb_epochs
        if nb_epochs is None:
            raise ValueError("nb_epochs was given neither here nor to the constructor")
        self.load_checkpoint()
        while self.epoch < nb_epochs:
            self.epoch += 1
            self.losses["train"].append(self.train_epoch(self.epoch)["loss"])
            if self.val_loader_iter is not None:
                val, _ = self.validation_epoch()
                self.losses["val"].append(val["loss"])
                self.early_stopping(val["loss"])
                self.save_checkpoint(val["loss"])
                if self.early_stopping.early_stop:
                    break
