"""Demucs training step on MI355X: forward (activations kept) + L1 + MultiResolutionSTFTLoss + backward + Adam over the C ABI.

Reference: Trainer.train_epoch, audio branch, training/train.py:275-312 -- predicted = Demucs(augmented);
loss = L1(predicted, clean) + sc_loss + mag_loss (MultiResolutionSTFTLoss, training/loss.py:128-186); zero_grad / backward /
Adam.step -- over Demucs.forward, training/model.py:290-326.

torch autograd is not used.  The engine owns ONE flat float32 buffer with every parameter in the layout the kernels read
("master" layouts, below), flat gradient / Adam-moment buffers of the same shape, and one fused Adam launch over it.

Master layouts (rows padded with zeros to a multiple of 64 where a GEMM reads them as its W operand; padding rows receive
zero gradients, so Adam leaves them at zero):
  enc0.w (8, 48) tap-major, encI.w (C_I, 8*C_{I-1}) rows [co][j][c], GLU 1x1 weights packed in [32 values | 32 gates] row tiles,
  decD.w (C, 8*cout) rows [c][j][co] -- the layout in which the transposed convolution's INPUT gradient is a plain strided
  Conv1d GEMM; its forward operand (rows [j][co], K = [previous row: tap j+4 | current row: tap j]) is re-derived from it every
  step, exactly as encI's input-gradient operand is derived from encI.w.  LSTM W_ih / W_hh as nn.LSTM stores them.
Weight re-layouts (transposes for the input-gradient GEMMs) are device copies made once per step (19 M parameters).

Input gradients run through mfpa_gemm_mfma (ops_demucs.gemm); weight gradients through mfpa_gemm_tn; see csrc/demucs_train.hip.
"""
from __future__ import annotations

import ctypes
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch

from . import ops_demucs as D
from . import ops_unet as _K
from ._lib import GemmTnDesc, check, lib, ptr, stream

DEPTH, KERNEL, STRIDE, FLOOR = D.DEPTH, D.KERNEL, D.STRIDE, D.FLOOR
HID = 48
CH = [HID * 2 ** i for i in range(DEPTH)]          # 48 .. 768
H = CH[-1]


def _pad64(n: int) -> int:
    return (n + 63) // 64 * 64


def _npad_glu(h: int) -> int:
    return (h + 31) // 32 * 64


def _p(t: torch.Tensor, off: int = 0) -> int:
    return ptr(t) + 4 * off


WGRAD_PRECISION = 0   # module default of gemm_tn: 0 fp32 MFMA, 1 bf16x3, 2 plain bf16 (the engine passes its own)


def gemm_tn(A: int, lda, strideA, Bm: int, ldb, strideB, C: torch.Tensor, ldc, batch, R, M, N, precision: Optional[int] = None,
            colsum: Optional[torch.Tensor] = None):
    """C[m][n] += sum_{b, r} A[b][r][m] * Bm[b][r][n] (mfpa_gemm_tn); A, Bm device addresses, strides in floats.
    colsum (M): += the column sums of A over all rows (the bias gradient), computed from the tiles the kernel stages anyway."""
    d = GemmTnDesc(A=A, lda=lda, strideA=strideA, Bm=Bm, ldb=ldb, strideB=strideB, C=ptr(C), ldc=ldc, batch=batch, R=R, M=M, N=N,
                   colsum=ptr(colsum), precision=WGRAD_PRECISION if precision is None else precision)
    t0 = _K._TIMER.start() if _K._TIMER is not None else None
    check(lib().mfpa_gemm_tn(ctypes.byref(d), stream()), "mfpa_gemm_tn")
    if t0 is not None:
        _K._TIMER.stop(t0)


def colsum(x: int, rows, C, ld, out: torch.Tensor):
    check(lib().mfpa_colsum_any(x, rows, C, ld, ptr(out), stream()), "mfpa_colsum_any")


def _window_t(Wm: torch.Tensor, rows: int, small: int) -> torch.Tensor:
    """Master (rows, 8*small) [r][j][s] -> the overlap-add operand (4*small, 2*rows): row j*small + s,
    K = [previous time step: tap j+4 | current time step: tap j] (the ConvTranspose1d form of csrc/demucs.hip)."""
    W3 = Wm[:rows].view(rows, 8, small)
    prev = W3[:, 4:8, :].permute(1, 2, 0)
    cur = W3[:, 0:4, :].permute(1, 2, 0)
    return torch.cat([prev, cur], dim=2).reshape(4 * small, 2 * rows).contiguous()


def _t_pad(w: torch.Tensor, rows: int) -> torch.Tensor:
    """(n, k) -> (pad64(k'), n) transposed copy of the first `rows` columns... i.e. w[:, :rows].T zero-padded to 64 rows."""
    out = torch.zeros((_pad64(rows), w.shape[0]), dtype=torch.float32, device=w.device)
    out[:rows] = w[:, :rows].t()
    return out


class _Phase:
    """HIP-event stopwatch over a phase of the step (bench.py's breakdown); a no-op unless engine.phases is a dict."""

    def __init__(self, eng, name):
        self.eng, self.name = eng, name

    def __enter__(self):
        if self.eng.phases is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if self.eng.phases is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.eng.phases.setdefault(self.name, []).append((self.e0, e1))


def _padded(new, zeros, B: int, L: int, C: int) -> torch.Tensor:
    """(B, L + 2, C) buffer whose rows 0 and L + 1 are zero; the L rows between are written by the GEMM that follows."""
    if not D._PAD_ROWS_ONLY:
        return zeros(B, L + 2, C)
    t = new(B, L + 2, C)
    t[:, 0].zero_()
    t[:, L + 1].zero_()
    return t


class DemucsTrainEngine:
    def __init__(self, state_dict: Dict[str, torch.Tensor], device, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 precision: int = 1, mrstft=None, dist_group=None, module=None, wgrad_precision: Optional[int] = None):
        """precision: forward / input-gradient GEMMs (0 exact fp32 products, 1 bf16x3).  wgrad_precision: weight-gradient GEMMs
        (0 fp32, 1 bf16x3, 2 plain bf16); default 2 with precision 1 (like the UNet engine), else 0."""
        self.device = torch.device(device)
        self.wgrad_precision = (2 if precision == 1 else 0) if wgrad_precision is None else wgrad_precision
        self.module = module                  # optional training.model.Demucs whose parameters mirror the flat buffer
        self.lr, self.betas, self.eps, self.precision = lr, betas, eps, precision
        self.step_count = 0
        self.phases: Optional[Dict[str, list]] = None
        self.debug: Optional[Dict[str, torch.Tensor]] = None     # set to {} to keep copies of the backward intermediates
        self.mrstft = mrstft
        self.dist_group = dist_group
        # ---- flat layout
        shapes: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
        shapes["enc0.w"], shapes["enc0.b"] = (8, HID), (HID,)
        for i in range(DEPTH):
            if i:
                shapes[f"enc{i}.w"], shapes[f"enc{i}.b"] = (_pad64(CH[i]), 8 * CH[i - 1]), (_pad64(CH[i]),)
            shapes[f"enc{i}.gw"], shapes[f"enc{i}.gb"] = (_npad_glu(CH[i]), CH[i]), (_npad_glu(CH[i]),)
        for layer in range(2):
            shapes[f"lstm{layer}.wih"], shapes[f"lstm{layer}.whh"] = (4 * H, H), (4 * H, H)
            shapes[f"lstm{layer}.bih"], shapes[f"lstm{layer}.bhh"] = (4 * H,), (4 * H,)
        for d in range(DEPTH):
            C = CH[DEPTH - 1 - d]
            shapes[f"dec{d}.gw"], shapes[f"dec{d}.gb"] = (_npad_glu(C), C), (_npad_glu(C),)
            if d < DEPTH - 1:
                cout = CH[DEPTH - 2 - d]
                shapes[f"dec{d}.w"], shapes[f"dec{d}.b"] = (_pad64(C), 8 * cout), (cout,)
        shapes["decL.w"], shapes["decL.b"] = (8, HID), (4,)          # the scalar bias in slot 0 (16-byte block)
        self.shapes = shapes
        n = sum(int(torch.Size(s).numel()) for s in shapes.values())
        self.n_params = n
        new = lambda: torch.zeros(n, dtype=torch.float32, device=self.device)
        self.flat_p, self.flat_g, self.flat_m, self.flat_v = new(), new(), new(), new()
        self.P: Dict[str, torch.Tensor] = {}
        self.G: Dict[str, torch.Tensor] = {}
        off = 0
        for k, s in shapes.items():
            cnt = int(torch.Size(s).numel())
            self.P[k] = self.flat_p[off:off + cnt].view(*s)
            self.G[k] = self.flat_g[off:off + cnt].view(*s)
            off += cnt
        self.sinc = D.sinc_kernel(self.device)
        self.load_state_dict(state_dict)
        self.loss_buf = torch.zeros(1, dtype=torch.float64, device=self.device)
        self.l1_ws = torch.empty(lib().mfpa_red_blocks(), dtype=torch.float64, device=self.device)
        self.last_losses: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None

    # ------------------------------------------------------------------ parameters <-> the reference's state_dict
    def _views(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        out, off = {}, 0
        for k, s in self.shapes.items():
            cnt = int(torch.Size(s).numel())
            out[k] = flat[off:off + cnt].view(*s)
            off += cnt
        return out

    def named_moments(self):
        """Adam's exp_avg / exp_avg_sq under the reference's parameter names (torch-Adam-compatible checkpoints)."""
        return self._export(self._views(self.flat_m)), self._export(self._views(self.flat_v))

    def load_named_moments(self, exp_avg: Dict[str, torch.Tensor], exp_avg_sq: Dict[str, torch.Tensor]) -> None:
        self.load_state_dict(exp_avg, self.flat_m)
        self.load_state_dict(exp_avg_sq, self.flat_v)

    @torch.no_grad()
    def load_state_dict(self, sd: Dict[str, torch.Tensor], flat: Optional[torch.Tensor] = None) -> None:
        """Tensors under the reference's state_dict keys -> the kernel layouts of the parameter buffer (default) or of `flat`."""
        f = lambda k: sd[k].detach().to(self.device, torch.float32)
        flat = self.flat_p if flat is None else flat
        P = self.P if flat is self.flat_p else self._views(flat)
        flat.zero_()
        P["enc0.w"].copy_(f("encoder.0.0.weight")[:, 0, :].t())
        P["enc0.b"].copy_(f("encoder.0.0.bias"))
        for i in range(DEPTH):
            if i:
                w = f(f"encoder.{i}.0.weight")                         # (C_i, cin, 8)
                P[f"enc{i}.w"][:CH[i]] = w.permute(0, 2, 1).reshape(CH[i], -1)
                P[f"enc{i}.b"][:CH[i]] = f(f"encoder.{i}.0.bias")
            gw, gb = D._pack_glu(f(f"encoder.{i}.2.weight")[:, :, 0], f(f"encoder.{i}.2.bias"))
            P[f"enc{i}.gw"].copy_(gw); P[f"enc{i}.gb"].copy_(gb)
        for layer in range(2):
            P[f"lstm{layer}.wih"].copy_(f(f"lstm.lstm.weight_ih_l{layer}"))
            P[f"lstm{layer}.whh"].copy_(f(f"lstm.lstm.weight_hh_l{layer}"))
            P[f"lstm{layer}.bih"].copy_(f(f"lstm.lstm.bias_ih_l{layer}"))
            P[f"lstm{layer}.bhh"].copy_(f(f"lstm.lstm.bias_hh_l{layer}"))
        for d in range(DEPTH):
            C = CH[DEPTH - 1 - d]
            gw, gb = D._pack_glu(f(f"decoder.{d}.0.weight")[:, :, 0], f(f"decoder.{d}.0.bias"))
            P[f"dec{d}.gw"].copy_(gw); P[f"dec{d}.gb"].copy_(gb)
            w = f(f"decoder.{d}.2.weight")                             # (C, cout, 8)
            if d < DEPTH - 1:
                P[f"dec{d}.w"][:C] = w.permute(0, 2, 1).reshape(C, -1)
                P[f"dec{d}.b"].copy_(f(f"decoder.{d}.2.bias"))
            else:
                P["decL.w"].copy_(w[:, 0, :].t())
                P["decL.b"][0] = f(f"decoder.{d}.2.bias")[0]

    @staticmethod
    def _unpack_glu(wp: torch.Tensor, bp: torch.Tensor, h: int):
        w = torch.empty((2 * h, wp.shape[1]), dtype=wp.dtype, device=wp.device)
        b = torch.empty((2 * h,), dtype=wp.dtype, device=wp.device)
        for t in range((h + 31) // 32):
            n = min(32, h - 32 * t)
            w[32 * t:32 * t + n] = wp[64 * t:64 * t + n]
            w[h + 32 * t:h + 32 * t + n] = wp[64 * t + 32:64 * t + 32 + n]
            b[32 * t:32 * t + n] = bp[64 * t:64 * t + n]
            b[h + 32 * t:h + 32 * t + n] = bp[64 * t + 32:64 * t + 32 + n]
        return w, b

    @torch.no_grad()
    def _export(self, T: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Blocks in master layout (parameters or gradients) -> tensors under the reference's state_dict keys."""
        out: Dict[str, torch.Tensor] = {}
        out["encoder.0.0.weight"] = T["enc0.w"].t().reshape(HID, 1, 8).clone()
        out["encoder.0.0.bias"] = T["enc0.b"].clone()
        for i in range(DEPTH):
            if i:
                out[f"encoder.{i}.0.weight"] = T[f"enc{i}.w"][:CH[i]].view(CH[i], 8, CH[i - 1]).permute(0, 2, 1).clone()
                out[f"encoder.{i}.0.bias"] = T[f"enc{i}.b"][:CH[i]].clone()
            w, b = self._unpack_glu(T[f"enc{i}.gw"], T[f"enc{i}.gb"], CH[i])
            out[f"encoder.{i}.2.weight"], out[f"encoder.{i}.2.bias"] = w.unsqueeze(-1), b
        for layer in range(2):
            out[f"lstm.lstm.weight_ih_l{layer}"] = T[f"lstm{layer}.wih"].clone()
            out[f"lstm.lstm.weight_hh_l{layer}"] = T[f"lstm{layer}.whh"].clone()
            out[f"lstm.lstm.bias_ih_l{layer}"] = T[f"lstm{layer}.bih"].clone()
            out[f"lstm.lstm.bias_hh_l{layer}"] = T[f"lstm{layer}.bhh"].clone()
        for d in range(DEPTH):
            C = CH[DEPTH - 1 - d]
            w, b = self._unpack_glu(T[f"dec{d}.gw"], T[f"dec{d}.gb"], C)
            out[f"decoder.{d}.0.weight"], out[f"decoder.{d}.0.bias"] = w.unsqueeze(-1), b
            if d < DEPTH - 1:
                cout = CH[DEPTH - 2 - d]
                out[f"decoder.{d}.2.weight"] = T[f"dec{d}.w"][:C].view(C, 8, cout).permute(0, 2, 1).clone()
                out[f"decoder.{d}.2.bias"] = T[f"dec{d}.b"].clone()
            else:
                out[f"decoder.{d}.2.weight"] = T["decL.w"].t().reshape(HID, 1, 8).clone()
                out[f"decoder.{d}.2.bias"] = T["decL.b"][:1].clone()
        return out

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return self._export(self.P)

    def sync_to_module(self) -> None:
        if self.module is not None:
            self.module.load_state_dict(self.state_dict())

    def load_from_module(self) -> None:
        if self.module is not None:
            self.load_state_dict(self.module.state_dict())

    def grad_dict(self) -> Dict[str, torch.Tensor]:
        return self._export(self.G)

    # ------------------------------------------------------------------ per-step derived operands
    @torch.no_grad()
    def _derive(self) -> Dict[str, torch.Tensor]:
        P, W = self.P, {}
        for i in range(1, DEPTH):
            W[f"enc{i}.wT"] = _window_t(P[f"enc{i}.w"], CH[i], CH[i - 1])                      # input-gradient operand
        for i in range(DEPTH):
            W[f"enc{i}.gwT"] = _t_pad(P[f"enc{i}.gw"], CH[i])
        for d in range(DEPTH):
            C = CH[DEPTH - 1 - d]
            W[f"dec{d}.gwT"] = _t_pad(P[f"dec{d}.gw"], C)
            if d < DEPTH - 1:
                cout = CH[DEPTH - 2 - d]
                W[f"dec{d}.wf"] = _window_t(P[f"dec{d}.w"], C, cout)                            # forward operand
                W[f"dec{d}.bf"] = P[f"dec{d}.b"].repeat(4).contiguous()
        for layer in range(2):
            whh = P[f"lstm{layer}.whh"]
            W[f"lstm{layer}.whh_grouped"] = whh.view(4, H // 16, 16, H).permute(1, 0, 2, 3).reshape(4 * H, H).contiguous()
            W[f"lstm{layer}.whhT"] = whh.t().contiguous()
            W[f"lstm{layer}.wihT"] = P[f"lstm{layer}.wih"].t().contiguous()
            W[f"lstm{layer}.b"] = P[f"lstm{layer}.bih"] + P[f"lstm{layer}.bhh"]
        return W

    # ------------------------------------------------------------------ forward (activations kept in self.S)
    @torch.no_grad()
    def forward(self, wav: torch.Tensor) -> torch.Tensor:
        """(B, T) float32 -> (B, T); keeps every activation the backward pass reads."""
        P, L = self.P, lib()
        W = self._derive()
        prec = self.precision
        B, T = wav.shape
        dev = wav.device
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        zeros = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
        S: Dict[str, object] = {"W": W, "B": B, "T": T}
        VL = D.valid_length(T)
        x, std = new(B, VL), new(B)
        check(L.mfpa_demucs_prep(ptr(wav), B, T, VL, FLOOR, ptr(x), ptr(std), stream()), "mfpa_demucs_prep")
        for _ in range(2):
            y = new(B, 2 * x.shape[1])
            check(L.mfpa_upsample2(ptr(x), B, x.shape[1], ptr(self.sinc), ptr(y), stream()), "mfpa_upsample2")
            x = y
        S["xup"], S["std"] = x, std
        # ---- encoder
        Ls, a_s, u_s, h_s = [], [], [], []
        h, Lin = None, x.shape[1]
        for i in range(DEPTH):
            C = CH[i]
            Lout = (Lin - KERNEL) // STRIDE + 1
            a = new(B, Lout, C)
            if i == 0:
                check(L.mfpa_conv1d_c1(ptr(x), B, Lin, Lout, C, ptr(P["enc0.w"]), ptr(P["enc0.b"]), 1, ptr(a), stream()),
                      "mfpa_conv1d_c1")
            else:
                Cin = CH[i - 1]
                D.gemm(_p(h), STRIDE * Cin, Lin * Cin, B, Lout, P[f"enc{i}.w"], P[f"enc{i}.b"], C, _p(a), C, Lout * C, relu=1,
                       precision=prec)
            npad = _npad_glu(C)
            u = new(B, Lout, npad)
            h = new(B, Lout, C)
            D.gemm(_p(a), C, Lout * C, B, Lout, P[f"enc{i}.gw"], P[f"enc{i}.gb"], C, _p(h), C, Lout * C, mode=1, precision=prec,
                   C2=_p(u), ldc2=npad, strideC2=Lout * npad)
            Ls.append(Lout); a_s.append(a); u_s.append(u); h_s.append(h)
            Lin = Lout
        S["L"], S["a"], S["u"], S["h"] = Ls, a_s, u_s, h_s
        # ---- LSTM
        Tn = Lin
        xsum, lst = D.lstm_two_layers(h, h_s[-1], [P["lstm0.wih"], P["lstm1.wih"]], [W["lstm0.b"], W["lstm1.b"]],
                                      [W["lstm0.whh_grouped"], W["lstm1.whh_grouped"]], prec, train=True)
        S["lstm"], S["Tn"] = lst, Tn
        # ---- decoder
        x = xsum
        Lcur = Tn
        xin_s, ud_s, P_s, r_s, Ld = [], [], [], [], []
        for d in range(DEPTH):
            C = CH[DEPTH - 1 - d]
            npad = _npad_glu(C)
            Pd = _padded(new, zeros, B, Lcur, C)
            ud = new(B, Lcur, npad)
            D.gemm(_p(x), C, Lcur * C, B, Lcur, P[f"dec{d}.gw"], P[f"dec{d}.gb"], C, _p(Pd, C), C, (Lcur + 2) * C, mode=1,
                   precision=prec, C2=_p(ud), ldc2=npad, strideC2=Lcur * npad)
            xin_s.append(x); ud_s.append(ud); P_s.append(Pd); Ld.append(Lcur)
            Lnext = 4 * (Lcur + 1)
            if d < DEPTH - 1:
                cout = CH[DEPTH - 2 - d]
                skip = h_s[DEPTH - 2 - d]
                y, r = new(B, Lnext, cout), new(B, Lnext, cout)
                D.gemm(_p(Pd), C, (Lcur + 2) * C, B, Lcur + 1, W[f"dec{d}.wf"], W[f"dec{d}.bf"], 4 * cout, _p(y), 4 * cout,
                       Lnext * cout, mode=2, relu=2, addend=_p(skip), ldadd=4 * cout, strideAdd=Lnext * cout, precision=prec,
                       C2=_p(r), ldc2=4 * cout, strideC2=Lnext * cout)
                r_s.append(r)
            else:
                y = new(B, Lnext)
                check(L.mfpa_convT1d_c1_dev(ptr(Pd), B, Lcur, C, ptr(P["decL.w"]), ptr(P["decL.b"]), ptr(y), stream()),
                      "mfpa_convT1d_c1_dev")
            x, Lcur = y, Lnext
        S["xin"], S["ud"], S["P"], S["r"], S["Ld"], S["Lfull"] = xin_s, ud_s, P_s, r_s, Ld, Lcur
        half = new(B, (Lcur + 1) // 2)
        check(L.mfpa_downsample2(ptr(x), B, Lcur, ptr(self.sinc), ptr(half), half.shape[1], 0, 0, stream()), "mfpa_downsample2")
        out = new(B, T)
        check(L.mfpa_downsample2(ptr(half), B, half.shape[1], ptr(self.sinc), ptr(out), T, ptr(std), T, stream()),
              "mfpa_downsample2")
        S["Lh"] = half.shape[1]
        self.S = S
        return out

    # ------------------------------------------------------------------ backward
    @torch.no_grad()
    def backward(self, dout: torch.Tensor) -> None:
        """dout (B, T) = dL/d(output of forward): accumulates every parameter gradient into self.flat_g (zeroed here)."""
        S, P, G, L = self.S, self.P, self.G, lib()
        W = S["W"]
        prec = self.precision
        import functools
        gemm_tn = functools.partial(globals()["gemm_tn"], precision=self.wgrad_precision)
        B, T = S["B"], S["T"]
        dev = dout.device
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        zeros = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
        self._last_mark = None
        if self.phases is not None:
            self._last_mark = torch.cuda.Event(enable_timing=True)
            self._last_mark.record()
        self.flat_g.zero_()
        Lfull, Lh = S["Lfull"], S["Lh"]
        dhalf = new(B, Lh)
        check(L.mfpa_downsample2_adjoint(ptr(dout), B, T, T, ptr(self.sinc), ptr(S["std"]), Lh, ptr(dhalf), stream()),
              "mfpa_downsample2_adjoint")
        dy = new(B, Lfull)
        check(L.mfpa_downsample2_adjoint(ptr(dhalf), B, Lh, Lh, ptr(self.sinc), 0, Lfull, ptr(dy), stream()),
              "mfpa_downsample2_adjoint")
        # ---- last ConvTranspose1d (C -> 1)
        Ld, Ps, uds, xins, rs = S["Ld"], S["P"], S["ud"], S["xin"], S["r"]
        d = DEPTH - 1
        C, Lcur = CH[0], Ld[d]
        colsum(ptr(dy), B * Lfull // 4, 4, 4, G["decL.b"])
        G["decL.b"][0] = G["decL.b"].sum()
        G["decL.b"][1:].zero_()
        check(L.mfpa_c1_wgrad(ptr(dy), Lfull, _p(Ps[d], C), C, (Lcur + 2) * C, B, Lcur, C, ptr(G["decL.w"]), stream()),
              "mfpa_c1_wgrad")
        dg = new(B, Lcur, C)
        check(L.mfpa_conv1d_c1(ptr(dy), B, Lfull, Lcur, C, ptr(P["decL.w"]), 0, 0, ptr(dg), stream()), "mfpa_conv1d_c1")
        self._dbg("dy", dy); self._dbg("dg4", dg)
        dskip = [None] * DEPTH
        dxsum = None
        for d in range(DEPTH - 1, -1, -1):
            C, Lcur = CH[DEPTH - 1 - d], Ld[d]
            npad = _npad_glu(C)
            ud = uds[d]
            check(L.mfpa_glu_bwd(ptr(ud), B * Lcur, npad, C, ptr(dg), C, stream()), "mfpa_glu_bwd")
            self._dbg(f"du{d}", ud)
            gemm_tn(ptr(ud), npad, 0, ptr(xins[d]), C, 0, G[f"dec{d}.gw"], C, 1, B * Lcur, npad, C, colsum=G[f"dec{d}.gb"])
            dxin = new(B, Lcur, C)                                       # gradient of x + skip: both addends receive it
            if d == 0:
                D.gemm(_p(ud), npad, Lcur * npad, B, Lcur, W[f"dec{d}.gwT"], None, C, _p(dxin), C, Lcur * C, precision=prec)
                dxsum = dxin
                dskip[DEPTH - 1] = dxin
                break
            # masked by the ReLU of the transposed convolution below (C) and unmasked for the skip connection (C2)
            dyc = new(B, Lcur, C)
            D.gemm(_p(ud), npad, Lcur * npad, B, Lcur, W[f"dec{d}.gwT"], None, C, _p(dyc), C, Lcur * C, mode=3,
                   addend=_p(rs[d - 1]), ldadd=C, strideAdd=Lcur * C, precision=prec, C2=_p(dxin), ldc2=C, strideC2=Lcur * C)
            dskip[DEPTH - 1 - d] = dxin
            self._dbg(f"dyc{d}", dyc); self._dbg(f"dxin{d}", dxin)
            # ---- ConvTranspose1d of decoder d-1: (B, Lp, Cp) -> (B, Lcur, C)
            Cp, Lp = CH[DEPTH - d], Ld[d - 1]
            colsum(ptr(dyc), B * Lcur, C, C, G[f"dec{d - 1}.b"])
            gemm_tn(_p(Ps[d - 1], Cp), Cp, (Lp + 2) * Cp, ptr(dyc), 4 * C, Lcur * C, G[f"dec{d - 1}.w"], 8 * C, B, Lp, Cp, 8 * C)
            dg = new(B, Lp, Cp)
            D.gemm(_p(dyc), 4 * C, Lcur * C, B, Lp, P[f"dec{d - 1}.w"], None, Cp, _p(dg), Cp, Lp * Cp, precision=prec)
            self._dbg(f"dg{d - 1}", dg)
        # ---- LSTM
        self._mark("bwd_decoder")
        Tn = S["Tn"]
        (seq0, g0, hseq0, cseq0), (seq1, g1, hseq1, cseq1) = S["lstm"]
        dc0, dc1, dx1 = new(B, H), new(B, H), new(B, Tn, H)
        t0 = _K._TIMER.start() if _K._TIMER is not None else None
        timer, _K._TIMER = _K._TIMER, None

        pipelined = D.PIPELINE_LSTM and D.PIPELINE_LSTM_BWD and Tn > D.LSTM_CHUNK and B <= D.PIPELINE_MAX_CLIPS
        # two persistent launches run side by side in the chunked pipeline: each may keep half the CUs' worth of workgroups resident
        wg_budget = torch.cuda.get_device_properties(dev).multi_processor_count // 2 if pipelined else 0
        bwork = {id(g1): D._lstm_work(dev, 1, B, H, backward=True), id(g0): D._lstm_work(dev, 0, B, H, backward=True)} if D.PERSISTENT_LSTM_BWD else None

        bwgs = D.lstm_seq_workgroups(B, H, wg_budget, backward=True) if bwork is not None else 0

        def bwd(whhT, gates, cseq, dhout, dc, a, b):
            if bwork is not None:            # one persistent launch for the range (csrc/demucs_train.hip: lstm_bwd_seq_kernel)
                done = D._GUARD.admit(dev, bwgs)
                check(L.mfpa_lstm_layer_bwd_seq(ptr(whhT), ptr(gates), ptr(cseq), ptr(dhout), ptr(dc), B, Tn, H, a, b, wg_budget,
                                                ptr(bwork[id(gates)]), stream()), "mfpa_lstm_layer_bwd_seq")
                done()
                return
            check(L.mfpa_lstm_layer_bwd_range(ptr(whhT), ptr(gates), ptr(cseq), ptr(dhout), ptr(dc), B, Tn, H, a, b, stream()),
                  "mfpa_lstm_layer_bwd_range")

        def dx_chunk(a, b):                                              # dL/d(h0)[:, a:b] = dgates1[:, a:b] W_ih1
            D.gemm(_p(g1, a * 4 * H), 4 * H, Tn * 4 * H, B, b - a, W["lstm1.wihT"], None, H, _p(dx1, a * H), H, Tn * H, precision=prec)

        try:
            if not pipelined:
                bwd(W["lstm1.whhT"], g1, cseq1, dxsum, dc1, 0, Tn)
                dx_chunk(0, Tn)
                bwd(W["lstm0.whhT"], g0, cseq0, dx1, dc0, 0, Tn)
            else:       # layer 1 runs backwards through the chunks on this stream, layer 0 follows one chunk behind on the side stream
                main, side = torch.cuda.current_stream(dev), D._side_stream(dev)
                side.wait_stream(main)
                starts = list(range(0, Tn, D.LSTM_CHUNK))
                for a in reversed(starts):
                    b = min(Tn, a + D.LSTM_CHUNK)
                    bwd(W["lstm1.whhT"], g1, cseq1, dxsum, dc1, a, b)
                    ev = torch.cuda.Event()
                    ev.record(main)
                    with torch.cuda.stream(side):
                        side.wait_event(ev)
                        dx_chunk(a, b)
                        bwd(W["lstm0.whhT"], g0, cseq0, dx1, dc0, a, b)
                main.wait_stream(side)
        finally:
            _K._TIMER = timer
        if t0 is not None:
            _K._TIMER.stop(t0)
        if bwork is not None:
            D.lstm_mark(dev)                 # the error words of this step's persistent launches, copied right behind the recurrence
        for layer, (seq, gates, hseq) in enumerate(((seq0, g0, hseq0), (seq1, g1, hseq1))):
            gemm_tn(ptr(gates), 4 * H, 0, ptr(seq), H, 0, G[f"lstm{layer}.wih"], H, 1, B * Tn, 4 * H, H, colsum=G[f"lstm{layer}.bih"])
            G[f"lstm{layer}.bhh"].copy_(G[f"lstm{layer}.bih"])
            if Tn > 1:
                gemm_tn(_p(gates, 4 * H), 4 * H, Tn * 4 * H, ptr(hseq), H, Tn * H, G[f"lstm{layer}.whh"], H, B, Tn - 1, 4 * H, H)
        dh_enc = new(B, Tn, H)                                           # dL/d(h_4) = layer 0's input gradient + the first decoder skip's
        D.gemm(_p(g0), 4 * H, 0, 1, B * Tn, W["lstm0.wihT"], None, H, _p(dh_enc), H, 0, mode=2, addend=_p(dxsum), ldadd=H, strideAdd=0,
               precision=prec)
        # ---- encoder
        self._mark("bwd_lstm")
        Ls, a_s, u_s, h_s = S["L"], S["a"], S["u"], S["h"]
        dh = dh_enc
        for i in range(DEPTH - 1, -1, -1):
            C, Li = CH[i], Ls[i]
            npad = _npad_glu(C)
            u = u_s[i]
            check(L.mfpa_glu_bwd(ptr(u), B * Li, npad, C, ptr(dh), C, stream()), "mfpa_glu_bwd")
            gemm_tn(ptr(u), npad, 0, ptr(a_s[i]), C, 0, G[f"enc{i}.gw"], C, 1, B * Li, npad, C, colsum=G[f"enc{i}.gb"])
            if i == 0:
                da = new(B, Li, C)
                D.gemm(_p(u), npad, Li * npad, B, Li, W[f"enc{i}.gwT"], None, C, _p(da), C, Li * C, mode=3, addend=_p(a_s[i]),
                       ldadd=C, strideAdd=Li * C, precision=prec)
                colsum(ptr(da), B * Li, C, C, G["enc0.b"])
                check(L.mfpa_c1_wgrad(ptr(S["xup"]), S["xup"].shape[1], ptr(da), C, Li * C, B, Li, C, ptr(G["enc0.w"]), stream()),
                      "mfpa_c1_wgrad")
                break
            dA = _padded(new, zeros, B, Li, C)                           # rows 0 and Li + 1 stay zero
            D.gemm(_p(u), npad, Li * npad, B, Li, W[f"enc{i}.gwT"], None, C, _p(dA, C), C, (Li + 2) * C, mode=3, addend=_p(a_s[i]),
                   ldadd=C, strideAdd=Li * C, precision=prec)
            Cin, Lprev = CH[i - 1], Ls[i - 1]
            gemm_tn(_p(dA, C), C, (Li + 2) * C, ptr(h_s[i - 1]), 4 * Cin, Lprev * Cin, G[f"enc{i}.w"], 8 * Cin, B, Li, C, 8 * Cin,
                    colsum=G[f"enc{i}.b"][:C])
            dprev = new(B, Lprev, Cin)                                   # ConvTranspose1d form + the decoder skip's gradient
            D.gemm(_p(dA), C, (Li + 2) * C, B, Li + 1, W[f"enc{i}.wT"], None, 4 * Cin, _p(dprev), 4 * Cin, Lprev * Cin, mode=2,
                   addend=_p(dskip[i - 1]), ldadd=4 * Cin, strideAdd=Lprev * Cin, precision=prec)
            dh = dprev
        self._mark("bwd_encoder")
        self.S = None

    def _dbg(self, name: str, t: torch.Tensor) -> None:
        if self.debug is not None:
            self.debug[name] = t.detach().clone()

    def _mark(self, name: str) -> None:
        """Phase boundary inside backward(): time since the previous mark (or since backward() began)."""
        if self.phases is None:
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        if self._last_mark is not None:
            self.phases.setdefault(name, []).append((self._last_mark, e))
        self._last_mark = e

    # ------------------------------------------------------------------ loss + step
    @torch.no_grad()
    def _mrstft(self):
        if self.mrstft is None:
            from .training.loss import MultiResolutionSTFTLoss
            self.mrstft = MultiResolutionSTFTLoss().to(self.device)
        return self.mrstft

    def loss_and_grad(self, pred: torch.Tensor, clean: torch.Tensor, Cys=None):
        """loss = L1(pred, clean) + sc + mag (train.py:292-297) and d loss / d pred.  Returns (l1, sc, mag, dpred).
        Cys: the target's DFT rows per resolution if they were computed ahead (train_step does, on a side stream)."""
        n = pred.numel()
        dpred = torch.empty_like(pred)
        check(lib().mfpa_l1_loss(ptr(pred), ptr(clean.double().contiguous()), n, ptr(dpred), ptr(self.loss_buf), ptr(self.l1_ws),
                                 stream()), "mfpa_l1_loss")
        l1 = self.loss_buf.clone()[0]
        sc, mag, _ = self._mrstft().value_and_grad(pred, clean, dx=dpred, accumulate=True, Cys=Cys)
        return l1, sc, mag, dpred

    @torch.no_grad()
    def adam_step(self) -> None:
        import torch.distributed as dist
        world = 1
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.dist_group) > 1:
            world = dist.get_world_size(self.dist_group)
            dist.all_reduce(self.flat_g, group=self.dist_group)          # SUM over ranks (RCCL); Adam divides by world
        self.step_count += 1
        check(lib().mfpa_adam_step(ptr(self.flat_p), ptr(self.flat_g), ptr(self.flat_m), ptr(self.flat_v), self.n_params,
                                   self.lr, self.betas[0], self.betas[1], self.eps, self.step_count, 1.0 / world, stream()),
              "mfpa_adam_step")

    @torch.no_grad()
    def train_step(self, clean: torch.Tensor, augmented: torch.Tensor) -> torch.Tensor:
        """One step of train.py:257-317 (audio branch) on (B, T) float32 waveforms; returns the loss (float64, on the device)."""
        clean = clean.contiguous()
        # the clean signal's three STFTs do not depend on the model: a side stream computes them while the forward pass runs
        main, side = torch.cuda.current_stream(clean.device), D._side_stream(clean.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            Cys = self._mrstft().target_transforms(clean)
        for c in Cys:
            c.record_stream(main)
        with _Phase(self, "forward"):
            pred = self.forward(augmented.contiguous())
        main.wait_stream(side)
        with _Phase(self, "loss"):
            l1, sc, mag, dpred = self.loss_and_grad(pred, clean, Cys)
        with _Phase(self, "backward"):
            self.backward(dpred)
        # the persistent LSTM launches' error words, read BEFORE the parameters are touched (one event wait: the host catches up with
        # the GPU at the end of the backward recurrence, the encoder's backward launches stay queued behind it).  A launch that gave
        # up (its grid was not co-resident) leaves garbage gradients: the step is repeated on the per-step kernels.
        if not D.lstm_results_ok(clean.device):
            pred = self.forward(augmented.contiguous())
            l1, sc, mag, dpred = self.loss_and_grad(pred, clean, Cys)
            self.backward(dpred)
            D.lstm_results_ok(clean.device)
        with _Phase(self, "adam"):
            self.adam_step()
        self.last_losses = (l1, sc, mag)
        return l1 + sc.double() + mag.double()
