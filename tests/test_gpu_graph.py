"""The C ABI never synchronises and allocates nothing (include/mfpa.h): the inference paths can be captured in a hipGraph
(torch.cuda.CUDAGraph) and replayed on new input with identical results."""
import pytest
import torch

from musicfpaugment_amd import synth

pytestmark = pytest.mark.gpu


def _capture(fn, static_in):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fn(static_in)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn(static_in)
    return g, out


def test_unet_hot_path_and_demucs_forward_replay_as_graphs():
    from musicfpaugment_amd.pipeline import HotPath
    from musicfpaugment_amd.training.demucs_weights import formula_state_dict as demucs_formula
    from musicfpaugment_amd.training.model import Demucs
    from musicfpaugment_amd.training.unet import UNet
    from musicfpaugment_amd.training.weights import formula_state_dict
    net = UNet(1, 1)
    net.load_state_dict(formula_state_dict(0))
    net = net.cuda().eval()
    net.precision = 1
    hp = HotPath(net)
    x = torch.from_numpy(synth.batch(4, seed=59, n=16000)).cuda()
    g, (mask, npk) = _capture(hp, x)
    x.copy_(torch.from_numpy(synth.batch(4, seed=77, n=16000)).cuda())       # new input, same buffers
    g.replay()
    torch.cuda.synchronize()
    want_mask, want_n = hp(x)
    assert torch.equal(mask, want_mask) and torch.equal(npk, want_n) and int(npk.sum()) > 0
    dm = Demucs()
    dm.load_state_dict(demucs_formula(0))
    dm = dm.cuda().eval()
    g2, out = _capture(dm, x)                                                # 2 x 62 fused LSTM step launches inside
    x.copy_(torch.from_numpy(synth.batch(4, seed=78, n=16000)).cuda())
    g2.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, dm(x))
