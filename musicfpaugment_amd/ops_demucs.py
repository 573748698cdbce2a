"""Demucs forward over the C ABI (csrc/demucs.hip): weight packing and the layer schedule.

Reference: Demucs.forward, training/model.py:290-326.  Activations are (B, L, C) float32 tensors."""
from __future__ import annotations

import ctypes
import os
import math
from typing import Dict

import torch

from . import ops_unet as _K
from ._lib import GemmDesc, check, lib, ptr, stream

DEPTH, KERNEL, STRIDE, RESAMPLE, FLOOR, ZEROS = 5, 8, 4, 4, 1e-3, 56


def sinc_kernel(device) -> torch.Tensor:
    """kernel_upsample2 / kernel_downsample2 (model.py:28-38,56-66), computed with torch exactly as the reference."""
    win = torch.hann_window(4 * ZEROS + 1, periodic=False)
    winodd = win[1::2]
    t = torch.linspace(-ZEROS + 0.5, ZEROS - 0.5, 2 * ZEROS) * math.pi
    return (torch.sin(t) / t * winodd).contiguous().to(device)


def valid_length(length: int) -> int:
    """Demucs.valid_length, model.py:269-285."""
    length = math.ceil(length * RESAMPLE)
    for _ in range(DEPTH):
        length = max(math.ceil((length - KERNEL) / STRIDE) + 1, 1)
    for _ in range(DEPTH):
        length = (length - 1) * STRIDE + KERNEL
    return int(math.ceil(length / RESAMPLE))


PAD_N_TO_WIDE_TILE = True   # K >= 128 layers with 128 < N, N % 128 != 0 (the two N = 192 layers): weight rows zero-padded to a multiple of
                            # 128 so that the pipelined 256 x 128 tile serves them (a third of the second column tile is idle) instead of
                            # the round-1 128 x 64 tile


def _wide_mult(n: int, k: int) -> int:
    return 128 if (PAD_N_TO_WIDE_TILE and n > 128 and n % 128 and k >= 128 and k % 64 == 0) else 64


def _pad_rows(w: torch.Tensor, mult: int = 64) -> torch.Tensor:
    n = w.shape[0]
    npad = (n + mult - 1) // mult * mult
    if npad == n:
        return w.contiguous()
    out = torch.zeros((npad,) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)
    out[:n] = w
    return out


def _pack_glu(w: torch.Tensor, b: torch.Tensor):
    """(2h, h) value|gate rows -> tiles of 64 rows: [32 value rows | their 32 gate rows] (zero padded), same for the bias."""
    h = w.shape[0] // 2
    nt = (h + 31) // 32
    wp = torch.zeros((nt * 64, w.shape[1]), dtype=torch.float32, device=w.device)
    bp = torch.zeros((nt * 64,), dtype=torch.float32, device=w.device)
    for t in range(nt):
        n = min(32, h - 32 * t)
        wp[64 * t:64 * t + n] = w[32 * t:32 * t + n]
        wp[64 * t + 32:64 * t + 32 + n] = w[h + 32 * t:h + 32 * t + n]
        bp[64 * t:64 * t + n] = b[32 * t:32 * t + n]
        bp[64 * t + 32:64 * t + 32 + n] = b[h + 32 * t:h + 32 * t + n]
    return wp, bp


def pack_demucs_weights(sd: Dict[str, torch.Tensor], device) -> Dict[str, torch.Tensor]:
    f = lambda k: sd[k].detach().to(device, torch.float32)
    pw: Dict[str, torch.Tensor] = {"sinc": sinc_kernel(device)}
    for i in range(DEPTH):
        w = f(f"encoder.{i}.0.weight")                         # (h, cin, 8)
        if i == 0:
            pw["enc0.w"] = w[:, 0, :].t().contiguous()         # (8, h) tap-major
            pw["enc0.b"] = f("encoder.0.0.bias").contiguous()
        else:
            wk = w.permute(0, 2, 1).reshape(w.shape[0], -1)                          # [co][j][c]: K = j*cin + c
            mult = _wide_mult(wk.shape[0], wk.shape[1])
            pw[f"enc{i}.w"] = _pad_rows(wk, mult)
            pw[f"enc{i}.b"] = _pad_rows(f(f"encoder.{i}.0.bias"), mult)
        pw[f"enc{i}.gw"], pw[f"enc{i}.gb"] = _pack_glu(f(f"encoder.{i}.2.weight")[:, :, 0], f(f"encoder.{i}.2.bias"))
    for d in range(DEPTH):
        pw[f"dec{d}.gw"], pw[f"dec{d}.gb"] = _pack_glu(f(f"decoder.{d}.0.weight")[:, :, 0], f(f"decoder.{d}.0.bias"))
        w = f(f"decoder.{d}.2.weight")                         # (h, cout, 8)
        h, cout = w.shape[0], w.shape[1]
        if d == DEPTH - 1:
            pw["decL.w"] = w[:, 0, :].t().contiguous()         # (8, h): rows j (current row taps 0..3) and j+4 (previous row)
            pw["decL.b"] = float(f(f"decoder.{d}.2.bias")[0])
            # the same layer as a strided-window GEMM (row t = [g[t-1] | g[t]], K = 2h, N = 4 outputs, padded to one 64-row tile)
            wg = torch.zeros((64, 2 * h), dtype=torch.float32, device=w.device)
            wg[:4, :h] = w[:, 0, 4:8].t()
            wg[:4, h:] = w[:, 0, 0:4].t()
            pw["decL.wg"] = wg
            bg = torch.zeros(64, dtype=torch.float32, device=w.device)
            bg[:4] = f(f"decoder.{d}.2.bias")[0]
            pw["decL.bg"] = bg
        else:
            # row n = j*cout + co, K = [previous row g[t-1] -> tap j+4 | current row g[t] -> tap j]
            wt = torch.cat([w[:, :, 4:8].permute(2, 1, 0), w[:, :, 0:4].permute(2, 1, 0)], dim=2)   # (4, cout, 2h)
            mult = _wide_mult(4 * cout, 2 * h)
            pw[f"dec{d}.w"] = _pad_rows(wt.reshape(4 * cout, 2 * h), mult)
            pw[f"dec{d}.b"] = _pad_rows(f(f"decoder.{d}.2.bias").repeat(4), mult)
    for layer in range(2):
        pw[f"lstm{layer}.wih"] = f(f"lstm.lstm.weight_ih_l{layer}").contiguous()
        whh = f(f"lstm.lstm.weight_hh_l{layer}")                # (4H, H), gate blocks i | f | g | o
        pw[f"lstm{layer}.whh"] = whh.contiguous()
        H = whh.shape[1]                                         # rows regrouped [H/16][i16|f16|g16|o16] for mfpa_lstm_step
        pw[f"lstm{layer}.whh_grouped"] = whh.reshape(4, H // 16, 16, H).permute(1, 0, 2, 3).reshape(4 * H, H).contiguous()
        pw[f"lstm{layer}.b"] = (f(f"lstm.lstm.bias_ih_l{layer}") + f(f"lstm.lstm.bias_hh_l{layer}")).contiguous()
    if PRESPLIT_WEIGHTS:
        for k, w in pw.items():
            if k.endswith((".w", ".gw", ".wih")):
                attach_split(w)
    return pw


def _p(t: torch.Tensor, off_floats: int = 0) -> int:
    """Device address of element `off_floats` of a contiguous float32 tensor."""
    return ptr(t) + 4 * off_floats


FUSE_FIRST_LAYER = True   # False: run mfpa_conv1d_c1_relu as its own launch
FUSE_FIRST_LEVEL = True   # False: the first convolution inside the loader of the 128 x 64-tile GLU GEMM (evaluated twice per row)
_PAD_ROWS_ONLY = True   # zero only the two padding rows of the GLU output (0: memset the whole buffer)
FUSE_LAST_LEVEL = True    # False: the last decoder level as two launches (1x1 + GLU, then the transposed convolution)
LAST_LAYER_GEMM = True    # False: the stand-alone VALU kernel mfpa_convT1d_c1 for the last ConvTranspose1d
PRESPLIT_WEIGHTS = True   # False: the GEMMs split the fp32 weights on the fly (the only form the training engine uses)
SPLIT_MIN_ROWS = 1        # rows of A from which the pre-split operand (the 128 x 128 kernel) is used
PRECISION = 1     # 0: exact fp32 products (v_mfma_f32_32x32x2_f32); 1: bf16x3 (3 bf16 MFMAs per product, fp32 accumulate)


def split_rows(W: torch.Tensor) -> torch.Tensor:
    """[Npad][K] float32 (K a multiple of 32) -> the same-shaped float32 container whose every 32-element chunk of a row holds
    [32 bf16 hi | 32 bf16 lo], w = hi + lo (+ O(2^-17 w)): the pre-split weight operand of mfpa_gemm_mfma precision 2."""
    n, k = W.shape
    w4 = W.reshape(n, k // 32, 32)
    hi = w4.to(torch.bfloat16)
    lo = (w4 - hi.float()).to(torch.bfloat16)
    return torch.cat([hi, lo], dim=-1).contiguous().view(torch.float32).reshape(n, k)


def attach_split(W: torch.Tensor) -> torch.Tensor:
    """A packed inference weight does not change between calls: hang a pre-split copy on it for the wide bf16x3 GEMM (the split
    otherwise runs in every workgroup for every K chunk).  An in-place update of W is noticed through its version counter."""
    if W.dim() == 2 and W.shape[1] % 32 == 0 and W.shape[1] >= 128 and W.shape[0] % 128 == 0 and W.is_cuda:
        W._mfpa_split = (W._version, split_rows(W))
    return W


def gemm(A: int, lda, strideA, batch, M, W, bias, N, C: int, ldc, strideC, *, mode=0, relu=0, addend: int = 0, ldadd=0,
         strideAdd=0, precision=None, c1=None, C2: int = 0, ldc2=0, strideC2=0):
    """C[b][m][:N] = epi(A-window[b][m] @ W^T + bias); A, C, addend, C2 are device addresses, strides in floats."""
    precision = PRECISION if precision is None else precision
    wptr = ptr(W)
    if precision == 1 and c1 is None:
        ent = getattr(W, "_mfpa_split", None)
        if ent is not None and ent[0] == W._version and M >= SPLIT_MIN_ROWS:
            wptr, precision = ptr(ent[1]), 2
    d = GemmDesc(A=A, lda=lda, strideA=strideA, W=wptr, bias=ptr(bias), addend=addend, ldadd=ldadd,
                 strideAdd=strideAdd, C=C, ldc=ldc, strideC=strideC, batch=batch, M=M, N=N, K=W.shape[1], npad=W.shape[0],
                 mode=mode, relu=int(relu), precision=precision,
                 C2=C2, ldc2=ldc2, strideC2=strideC2)
    if c1 is not None:                 # (x (B, Lin), w (8, K), b (K)): A is the first encoder layer, computed in the loader
        d.c1_x, d.c1_lin, d.c1_w, d.c1_b = ptr(c1[0]), c1[0].shape[1], ptr(c1[1]), ptr(c1[2])
    t0 = _K._TIMER.start() if _K._TIMER is not None else None
    check(lib().mfpa_gemm_mfma(ctypes.byref(d), stream()), "mfpa_gemm_mfma")
    if t0 is not None:
        _K._TIMER.stop(t0)



PIPELINE_LSTM = True      # False: the two LSTM layers one after the other on the current stream
PIPELINE_MAX_CLIPS = 96   # above this a step fills the chip on its own: forward 7.1 -> 5.5 ms at 16 clips, 13.6 -> 12.6 at 64, 21.3 -> 22.0 at 128
LSTM_CHUNK = 31   # time steps per pipeline stage (248 = 8 x 31)
_SIDE_STREAMS: Dict[int, "torch.cuda.Stream"] = {}


def _side_stream(dev) -> "torch.cuda.Stream":
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _SIDE_STREAMS:
        _SIDE_STREAMS[idx] = torch.cuda.Stream(device=dev)
    return _SIDE_STREAMS[idx]


PERSISTENT_LSTM = True    # False: one launch per time step (mfpa_lstm_layer_range)
PERSISTENT_LSTM_BWD = True   # training: the backward recurrence as one launch per range too (mfpa_lstm_layer_bwd_seq)
PIPELINE_LSTM_BWD = False    # training: the two layers' backward recurrences as a chunk pipeline on two streams, like the forward (measured at 64 clips,
                             # ms per step, three runs each: persistent + one layer after the other 41.1 / 41.7 / 42.8; per-step launches + pipeline
                             # 42.3 / 42.6 / 42.7; persistent + pipeline 43.2 / 43.5 / 45.6; per-step, no pipeline 44.1 / 44.3 / 45.4)
_LSTM_WORK: Dict[tuple, list] = {}
_LSTM_TOUCHED: Dict[int, list] = {}      # device index -> work-buffer entries used since the last lstm_results_ok()


def _dev_index(dev) -> int:
    return dev.index if dev.index is not None else torch.cuda.current_device()


def _lstm_work(dev, layer: int, B: int, H: int, backward: bool = False) -> torch.Tensor:
    """Scratch of the persistent LSTM kernel for one layer (mfpa_lstm_layer_seq): the exchange buffers of h and the slab counters.
    Kept per (device, stream, layer, shape).  Every entry handed out is remembered until lstm_results_ok() has looked at its error
    word -- the operators call that before their results leave them (demucs_forward, DemucsTrainEngine.train_step)."""
    L = lib()
    size_fn = L.mfpa_lstm_bwd_seq_work_bytes if backward else L.mfpa_lstm_seq_work_bytes      # the backward form exchanges 4H columns
    if torch.cuda.is_current_stream_capturing():
        # inside a HIP graph capture: scratch from the graph's own pool, zeroed by a captured fill on every replay; no host-side
        # look at the error word (nothing may synchronise or query here) -- a replayed graph reports through its results only
        nbytes = ctypes.c_longlong(0)
        check(size_fn(B, H, ctypes.addressof(nbytes)), "mfpa_lstm_seq_work_bytes")
        return torch.zeros(nbytes.value // 4, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream(dev)
    key = (_dev_index(dev), st.cuda_stream, layer, B, H, backward)
    ent = _LSTM_WORK.get(key)
    if ent is None:
        nbytes = ctypes.c_longlong(0)
        check(size_fn(B, H, ctypes.addressof(nbytes)), "mfpa_lstm_seq_work_bytes")
        buf = torch.zeros(nbytes.value // 4, dtype=torch.int32, device=dev)
        ent = _LSTM_WORK[key] = [buf, torch.zeros(1, dtype=torch.int32).pin_memory()]
    touched = _LSTM_TOUCHED.setdefault(key[0], [])
    if not any(e is ent for e in touched):
        touched.append(ent)
    return ent[0]


persistent_lstm_fallbacks = 0      # how often a persistent launch gave up and the per-step kernels took over (bench.py puts it in its line)
_LSTM_MARK: Dict[int, "torch.cuda.Event"] = {}     # device index -> event behind the last persistent launches and their error-word copies


def lstm_mark(dev) -> None:
    """Call right behind a group of persistent LSTM launches (on the stream that is ordered after all of them): queues the copy of
    their scratch buffers' error words to pinned host memory and records the event lstm_results_ok() will wait for.  The wait then
    ends as soon as the GPU has passed the RECURRENCE -- whatever the operator queues afterwards (the decoder, the encoder's
    backward) is still in the queue while the host looks at the words, so the GPU never idles on a slow host."""
    idx = _dev_index(dev)
    touched = _LSTM_TOUCHED.get(idx)
    if not touched or torch.cuda.is_current_stream_capturing():
        return
    off = lib().mfpa_lstm_seq_error_offset() // 4
    for buf, host in touched:
        host.copy_(buf[off:off + 1], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    _LSTM_MARK[idx] = ev


def lstm_results_ok(dev) -> bool:
    """Did every persistent LSTM launch issued on `dev` since the last call finish its waits?  Called where results leave an operator.

    The kernels spin on inter-workgroup counters and need their whole grid co-resident; a wait that gives up (another process on the
    GPU, a grid that could not become resident) raises an error word in the launch's scratch and the kernel finishes with garbage.
    lstm_mark() queued the copy of those words behind the launches; here the host waits for that event -- ONE wait, which ends when
    the GPU has passed the recurrence, with the rest of the operator's launches still queued -- and reads them.
    On error: the words are cleared, the persistent path is switched OFF for the process (PERSISTENT_LSTM / _BWD = False: the
    per-step kernels need no co-residency) and False is returned -- the caller re-runs its launches.  Under HIP-graph capture nothing
    can be checked (no host access): returns True, and the capture's scratch lives in the graph's pool."""
    global PERSISTENT_LSTM, PERSISTENT_LSTM_BWD, persistent_lstm_fallbacks
    idx = _dev_index(dev)
    touched = _LSTM_TOUCHED.get(idx)
    if not touched or torch.cuda.is_current_stream_capturing():
        return True
    off = lib().mfpa_lstm_seq_error_offset() // 4
    if _LSTM_MARK.get(idx) is None:          # launches issued without a mark (direct users of the C entry points): mark now
        cur = torch.cuda.current_stream(dev)
        side = _SIDE_STREAMS.get(idx)
        if side is not None and side is not cur:
            cur.wait_stream(side)
        lstm_mark(dev)
    _LSTM_MARK.pop(idx).synchronize()
    bad = [e for e in touched if int(e[1][0]) != 0]
    _LSTM_TOUCHED[idx] = []
    if not bad:
        return True
    torch.cuda.synchronize(dev)                      # the scratch buffers are keyed per stream: nothing may still be using them
    for key, ent in _LSTM_WORK.items():              # the word is sticky (it ends every later wait): clear it in every scratch of the device
        if key[0] == idx:
            ent[0][off:off + 1].zero_()
    torch.cuda.synchronize(dev)
    persistent_lstm_fallbacks += 1
    PERSISTENT_LSTM = False
    PERSISTENT_LSTM_BWD = False
    import warnings
    warnings.warn("a persistent LSTM launch gave up waiting for its slab (its grid was not co-resident: another process or kernel held "
                  "the CUs); the per-step LSTM kernels are used from now on and the affected call is re-run", RuntimeWarning)
    return False


class _ResidentGuard:
    """Accounting of persistent LSTM grids in flight on one device, so that launches issued from different streams (or host threads)
    of this process never need more resident workgroups than the device has CUs: a grid whose workgroups cannot all become resident
    next to another resident persistent grid would spin forever (until its bounded waits give up).  Before a launch of `wgs`
    workgroups on stream s, launches still in flight on OTHER streams are summed; while they and the new one exceed the CU count,
    s is made to wait (device-side, stream.wait_event -- the host never blocks) for the oldest of them."""

    def __init__(self):
        import threading
        self.lock = threading.Lock()
        self.inflight: Dict[int, list] = {}          # device index -> [(event, stream id, workgroups)]

    def admit(self, dev, wgs: int):
        """Call right before the launch; returns a callable to call right after it (records the launch's completion event)."""
        if wgs <= 0 or torch.cuda.is_current_stream_capturing():
            return lambda: None
        idx = _dev_index(dev)
        cus = torch.cuda.get_device_properties(idx).multi_processor_count
        st = torch.cuda.current_stream(dev)
        with self.lock:
            live = [e for e in self.inflight.get(idx, []) if not e[0].query()]
            others = [e for e in live if e[1] != st.cuda_stream]
            while others and sum(e[2] for e in others) + wgs > cus:
                oldest = others.pop(0)
                st.wait_event(oldest[0])
                live.remove(oldest)
            self.inflight[idx] = live

        def done():
            ev = torch.cuda.Event()
            ev.record(st)
            with self.lock:
                self.inflight.setdefault(idx, []).append((ev, st.cuda_stream, wgs))
        return done


_GUARD = _ResidentGuard()


def lstm_seq_workgroups(B: int, H: int, wg_budget: int = 0, backward: bool = False) -> int:
    n = ctypes.c_int(0)
    fn = lib().mfpa_lstm_bwd_seq_workgroups if backward else lib().mfpa_lstm_seq_workgroups
    check(fn(B, H, wg_budget, ctypes.addressof(n)), "mfpa_lstm_seq_workgroups")
    return n.value


def lstm_seq_error() -> bool:
    """Synchronise and report whether any persistent LSTM launch so far gave up a wait (tests)."""
    torch.cuda.synchronize()
    off = lib().mfpa_lstm_seq_error_offset() // 4
    return any(int(ent[0][off]) != 0 for ent in _LSTM_WORK.values())


def lstm_two_layers(x: torch.Tensor, skip: torch.Tensor, wih, bias, whh_grouped, precision: int, train: bool):
    """Both LSTM layers (model.py:91-110) on x (B, Tn, H): returns (xsum = h1 + skip, saved) with saved = per layer
    (input, gates-or-projections, hseq, cseq-or-None).

    A time step of one layer occupies 48-96 of the 256 CUs (it is bound by what those CUs can stream), so the two
    layers run as a PIPELINE on two streams: while layer 0 works on chunk k+1 of the sequence on the current stream, a side
    stream projects chunk k of its output (h0 W_ih1^T) and runs layer 1 on it.  Joined before returning."""
    B, Tn, H = x.shape
    dev = x.device
    L = lib()
    new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    xp = [new(B, Tn, 4 * H), new(B, Tn, 4 * H)]
    hs = [new(B, Tn, H), new(B, Tn, H)]
    cs = [new(B, Tn, H), new(B, Tn, H)] if train else [None, None]
    cstate = [None, None] if train else [new(B, H), new(B, H)]
    xsum = new(B, Tn, H)
    t0ev = _K._TIMER.start() if _K._TIMER is not None else None        # projection + recurrence of both layers as one timed group
    timer, _K._TIMER = _K._TIMER, None
    try:
        gemm(_p(x), H, 0, 1, B * Tn, wih[0], bias[0], 4 * H, _p(xp[0]), 4 * H, 0, precision=precision)

        work = [_lstm_work(dev, k, B, H) for k in range(2)] if PERSISTENT_LSTM else None
        pipelined = PIPELINE_LSTM and Tn > LSTM_CHUNK and B <= PIPELINE_MAX_CLIPS
        # two persistent launches run side by side in the chunked pipeline: each may keep half the CUs' worth of workgroups resident
        wg_budget = torch.cuda.get_device_properties(dev).multi_processor_count // 2 if pipelined else 0
        wgs = lstm_seq_workgroups(B, H, wg_budget) if work is not None else 0

        def layer(k, a, b):
            if work is not None:
                done = _GUARD.admit(dev, wgs)
                check(L.mfpa_lstm_layer_seq(ptr(whh_grouped[k]), ptr(xp[k]), ptr(hs[k]), ptr(cs[k]) if train else 0,
                                            0 if train else ptr(cstate[k]), B, Tn, H, ptr(xsum) if k == 1 else 0,
                                            ptr(skip) if k == 1 else 0, int(train), a, b, wg_budget, ptr(work[k]), stream()),
                      "mfpa_lstm_layer_seq")
                done()
                return
            check(L.mfpa_lstm_layer_range(ptr(whh_grouped[k]), ptr(xp[k]), ptr(hs[k]), ptr(cs[k]) if train else 0,
                                          0 if train else ptr(cstate[k]), B, Tn, H, ptr(xsum) if k == 1 else 0,
                                          ptr(skip) if k == 1 else 0, int(train), a, b, stream()), "mfpa_lstm_layer_range")

        def project(a, b):                                               # xp1[:, a:b] = h0[:, a:b] W_ih1^T + bias
            gemm(_p(hs[0], a * H), H, Tn * H, B, b - a, wih[1], bias[1], 4 * H, _p(xp[1], a * 4 * H), 4 * H, Tn * 4 * H,
                 precision=precision)

        if not pipelined:
            layer(0, 0, Tn)
            gemm(_p(hs[0]), H, 0, 1, B * Tn, wih[1], bias[1], 4 * H, _p(xp[1]), 4 * H, 0, precision=precision)
            layer(1, 0, Tn)
        else:
            main, side = torch.cuda.current_stream(dev), _side_stream(dev)
            side.wait_stream(main)
            for a in range(0, Tn, LSTM_CHUNK):
                b = min(Tn, a + LSTM_CHUNK)
                layer(0, a, b)
                ev = torch.cuda.Event()
                ev.record(main)
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    project(a, b)
                    layer(1, a, b)
            main.wait_stream(side)
    finally:
        _K._TIMER = timer
    if t0ev is not None:
        _K._TIMER.stop(t0ev)
    if work is not None:
        lstm_mark(dev)                       # both layers are ordered before this point of the current stream (main.wait_stream(side))
    return xsum, [(x, xp[0], hs[0], cs[0]), (hs[0], xp[1], hs[1], cs[1])]


def demucs_forward(pw: Dict[str, torch.Tensor], wav: torch.Tensor, precision: int = PRECISION) -> torch.Tensor:
    """(B, T) float32 on the GPU -> (B, T) denoised waveform.  model.py:290-326.  `precision` selects the GEMM arithmetic
    (the fused LSTM step is always bf16x3).  Before the result is returned the persistent LSTM launches' error words are read
    (lstm_results_ok: one event wait, the decoder stays queued behind it); a launch that gave up is re-run on the per-step kernels."""
    out = _demucs_forward(pw, wav, precision)
    if not lstm_results_ok(wav.device):
        out = _demucs_forward(pw, wav, precision)               # PERSISTENT_LSTM is off now: no co-residency needed
        lstm_results_ok(wav.device)
    return out


def _demucs_forward(pw: Dict[str, torch.Tensor], wav: torch.Tensor, precision: int = PRECISION) -> torch.Tensor:
    import functools
    gemm_p = functools.partial(gemm, precision=precision)
    B, T = wav.shape
    dev = wav.device
    L = lib()
    new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    VL = valid_length(T)
    x, std = new(B, VL), new(B)
    check(L.mfpa_demucs_prep(ptr(wav), B, T, VL, FLOOR, ptr(x), ptr(std), stream()), "mfpa_demucs_prep")
    for _ in range(2):                                           # resample 4 = two sinc x2 stages (model.py:303-307)
        y = new(B, 2 * x.shape[1])
        check(L.mfpa_upsample2(ptr(x), B, x.shape[1], ptr(pw["sinc"]), ptr(y), stream()), "mfpa_upsample2")
        x = y
    # ---- encoder: [Conv1d(k8,s4) + ReLU, Conv1d(1x1) + GLU] x 5
    chans = [pw[f"enc{i}.gw"].shape[1] for i in range(DEPTH)]    # 48 ... 768
    skips, h, Lin = [], None, x.shape[1]
    for i in range(DEPTH):
        C = chans[i]
        Lout = (Lin - KERNEL) // STRIDE + 1
        if i == 0 and FUSE_FIRST_LAYER:
            # Conv1d(1 -> 48, k8, s4) + ReLU is evaluated inside the loader of the 1x1 + GLU GEMM: its (B, L, 48) output
            # (12 MB per clip) is never written
            h = new(B, Lout, C)
            if FUSE_FIRST_LEVEL and precision == 1 and C == 48 and Lin % 4 == 0:   # one workgroup per 128 rows owns all packed GLU columns
                t0 = _K._TIMER.start() if _K._TIMER is not None else None      # counted with the GEMM family (bench.py's roofline block)
                check(L.mfpa_conv1d_c1_glu(ptr(x), B, Lin, Lout, C, ptr(pw["enc0.w"]), ptr(pw["enc0.b"]), ptr(pw["enc0.gw"]),
                                           ptr(pw["enc0.gb"]), ptr(h), stream()), "mfpa_conv1d_c1_glu")
                if t0 is not None:
                    _K._TIMER.stop(t0)
            else:
                gemm_p(0, C, Lout * C, B, Lout, pw["enc0.gw"], pw["enc0.gb"], C, _p(h), C, Lout * C, mode=1,
                       c1=(x, pw["enc0.w"], pw["enc0.b"]))
        else:
            a = new(B, Lout, C)
            if i == 0:
                check(L.mfpa_conv1d_c1_relu(ptr(x), B, Lin, Lout, C, ptr(pw["enc0.w"]), ptr(pw["enc0.b"]), ptr(a), stream()),
                      "mfpa_conv1d_c1_relu")
            else:
                Cin = chans[i - 1]                               # row t = h[4t : 4t+8] flattened: stride 4*Cin, K = 8*Cin
                gemm_p(_p(h), STRIDE * Cin, Lin * Cin, B, Lout, pw[f"enc{i}.w"], pw[f"enc{i}.b"], C, _p(a), C, Lout * C, relu=1)
            h = new(B, Lout, C)
            gemm_p(_p(a), C, Lout * C, B, Lout, pw[f"enc{i}.gw"], pw[f"enc{i}.gb"], C, _p(h), C, Lout * C, mode=1)
        skips.append(h)
        Lin = Lout
    # ---- LSTM: 2 layers, unidirectional, zero initial state (model.py:91-110); the last layer also emits h + skip
    Tn, H = Lin, chans[-1]
    xsum, _ = lstm_two_layers(h, skips[-1], [pw["lstm0.wih"], pw["lstm1.wih"]], [pw["lstm0.b"], pw["lstm1.b"]],
                              [pw["lstm0.whh_grouped"], pw["lstm1.whh_grouped"]], precision, train=False)
    # ---- decoder: (x + skip) -> Conv1d(1x1) + GLU -> ConvTranspose1d(k8,s4) [+ ReLU], next skip added in the epilogue
    x = xsum
    skips.pop()
    Lcur = Tn
    for d in range(DEPTH):
        C = chans[DEPTH - 1 - d]
        if d == DEPTH - 1 and FUSE_LAST_LEVEL and precision == 1 and C == 48:
            y = new(B, 4 * (Lcur + 1))                           # 1x1 + GLU + ConvTranspose1d(48 -> 1) without the GLU output in memory
            t0 = _K._TIMER.start() if _K._TIMER is not None else None
            check(L.mfpa_glu_convT1d_c1(ptr(x), B, Lcur, C, ptr(pw[f"dec{d}.gw"]), ptr(pw[f"dec{d}.gb"]), ptr(pw["decL.w"]), pw["decL.b"],
                                        ptr(y), stream()), "mfpa_glu_convT1d_c1")
            if t0 is not None:
                _K._TIMER.stop(t0)
            x, Lcur = y, 4 * (Lcur + 1)
            break
        if _PAD_ROWS_ONLY:
            P = new(B, Lcur + 2, C)                                                # rows 0 and L+1 are the zero padding
            P[:, 0].zero_()
            P[:, Lcur + 1].zero_()
        else:
            P = torch.zeros((B, Lcur + 2, C), dtype=torch.float32, device=dev)      # rows 0 and L+1 stay zero
        gemm_p(_p(x), C, Lcur * C, B, Lcur, pw[f"dec{d}.gw"], pw[f"dec{d}.gb"], C, _p(P, C), C, (Lcur + 2) * C, mode=1)
        Lnext = 4 * (Lcur + 1)                                   # (L - 1) * 4 + 8
        if d < DEPTH - 1:
            cout = chans[DEPTH - 2 - d]
            skip = skips.pop()                                   # (B, Lnext, cout)
            y = new(B, Lnext, cout)                              # row t = [g[t-1] | g[t]] -> positions 4t .. 4t+3
            gemm_p(_p(P), C, (Lcur + 2) * C, B, Lcur + 1, pw[f"dec{d}.w"], pw[f"dec{d}.b"], 4 * cout, _p(y), 4 * cout,
                 Lnext * cout, mode=2, relu=2, addend=_p(skip), ldadd=4 * cout, strideAdd=Lnext * cout)
        elif LAST_LAYER_GEMM and precision == 1:
            # ConvTranspose1d(48 -> 1) as the same strided-window GEMM (K = 96, four output columns): the VALU kernel reads every
            # row of P eight times (2.2 ms per 256 clips), the short-K bf16x3 GEMM streams it once (1.2 ms)
            y = new(B, Lnext)
            gemm_p(_p(P), C, (Lcur + 2) * C, B, Lcur + 1, pw["decL.wg"], pw["decL.bg"], 4, _p(y), 4, Lnext)
        else:
            y = new(B, Lnext)
            check(L.mfpa_convT1d_c1(ptr(P), B, Lcur, C, ptr(pw["decL.w"]), pw["decL.b"], ptr(y), stream()), "mfpa_convT1d_c1")
        x, Lcur = y, Lnext
    # ---- sinc x4 down, trim to T, times std (model.py:319-326)
    half = new(B, (Lcur + 1) // 2)
    check(L.mfpa_downsample2(ptr(x), B, Lcur, ptr(pw["sinc"]), ptr(half), half.shape[1], 0, 0, stream()), "mfpa_downsample2")
    out = new(B, T)
    check(L.mfpa_downsample2(ptr(half), B, half.shape[1], ptr(pw["sinc"]), ptr(out), T, ptr(std), T, stream()),
          "mfpa_downsample2")
    return out
