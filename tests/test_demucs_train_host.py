"""Host logic of the Demucs training engine that needs no GPU: the flat master-layout buffer holds every reference parameter
exactly once (plus zero padding), and the reference's state_dict round-trips through it."""
import torch

from musicfpaugment_amd.training.demucs_weights import formula_state_dict, state_dict_shapes


def test_flat_layout_round_trip_on_cpu():
    from musicfpaugment_amd.ops_demucs_train import CH, DemucsTrainEngine
    sd = formula_state_dict(3)
    eng = DemucsTrainEngine(sd, "cpu")
    back = eng.state_dict()
    assert list(back.keys()) != [] and set(back.keys()) == set(state_dict_shapes().keys())
    assert all(torch.equal(back[k], sd[k]) for k in sd)
    n_ref = sum(v.numel() for v in sd.values())
    assert n_ref == 18_867_937
    # every non-zero entry of the flat buffer is a reference parameter; the rest is GEMM row padding
    assert int((eng.flat_p != 0).sum()) <= n_ref <= eng.n_params
    assert eng.n_params - n_ref < 0.02 * n_ref
    # padding rows of the 96-channel level (96 -> 128 rows) are zero and stay addressable as a W operand
    assert eng.P["enc1.w"].shape == (128, 8 * CH[0]) and float(eng.P["enc1.w"][96:].abs().max()) == 0.0
    # gradients exported through the same mapping
    eng.flat_g.copy_(eng.flat_p)
    gd = eng.grad_dict()
    assert all(torch.equal(gd[k], sd[k]) for k in sd)


def test_derived_operands_match_the_inference_packing():
    """The per-step re-layouts (forward ConvTranspose1d operand, grouped W_hh) equal what ops_demucs.pack_demucs_weights builds
    from the reference layout, so the training forward and the inference forward read identical weights."""
    from musicfpaugment_amd import ops_demucs as D
    from musicfpaugment_amd.ops_demucs_train import DemucsTrainEngine
    sd = formula_state_dict(1)
    eng = DemucsTrainEngine(sd, "cpu")
    W = eng._derive()
    pw = D.pack_demucs_weights(sd, "cpu")
    def same_rows(train, packed):
        # the inference packing may append zero rows (ops_demucs.PAD_N_TO_WIDE_TILE: N padded to the wide GEMM tile); the rows that
        # carry weights are the training operand's, bit for bit
        n = train.shape[0]
        return packed.shape[0] >= n and torch.equal(train, packed[:n]) and not packed[n:].any()

    for d in range(4):
        assert same_rows(W[f"dec{d}.wf"], pw[f"dec{d}.w"])
        assert same_rows(W[f"dec{d}.bf"], pw[f"dec{d}.b"])
        assert same_rows(eng.P[f"dec{d}.gw"], pw[f"dec{d}.gw"])
    for i in range(1, 5):
        assert same_rows(eng.P[f"enc{i}.w"], pw[f"enc{i}.w"])
    for layer in range(2):
        assert torch.equal(W[f"lstm{layer}.whh_grouped"], pw[f"lstm{layer}.whh_grouped"])
        assert torch.equal(W[f"lstm{layer}.b"], pw[f"lstm{layer}.b"])


def test_unet_flop_count_matches_the_survey_figure():
    """pipeline.unet_mfma_gflop: the per-clip FLOPs the roofline figure is built from (93.398 GFLOP forward,
    0.074 + 0.008 of it in the two one-channel VALU layers)."""
    from musicfpaugment_amd.pipeline import UNET_MFMA_GFLOP_PER_CLIP, unet_mfma_gflop
    assert abs(unet_mfma_gflop(257, 251) - UNET_MFMA_GFLOP_PER_CLIP) < 1e-3
    assert abs(unet_mfma_gflop(257, 94) - (34.177 - 0.027 - 0.003)) < 5e-3       # the reference's 3 s training length
    assert unet_mfma_gflop(257, 249) < unet_mfma_gflop(257, 251)


def test_presplit_weight_image_layout_and_accuracy():
    """ops_demucs.split_rows (the weight operand of mfpa_gemm_mfma precision 2): every 32-element chunk of a row becomes
    [32 bf16 hi | 32 bf16 lo] in the same float32 container, hi = bf16(w), lo = bf16(w - hi), and hi + lo reproduces w to 2^-16."""
    from musicfpaugment_amd import ops_demucs as D
    g = torch.Generator().manual_seed(0)
    W = torch.randn(128, 96, generator=g)
    S = D.split_rows(W)
    assert S.shape == W.shape and S.dtype == torch.float32
    halves = S.view(torch.bfloat16).reshape(128, 3, 2, 32)                    # (row, chunk, hi | lo, 32)
    hi, lo = halves[:, :, 0].float(), halves[:, :, 1].float()
    w3 = W.reshape(128, 3, 32)
    assert torch.equal(hi, w3.to(torch.bfloat16).float())
    assert torch.equal(lo, (w3 - hi).to(torch.bfloat16).float())
    assert float(((hi + lo) - w3).abs().max()) <= 2.0 ** -16 * float(w3.abs().max())
    # CPU tensors and shapes the pre-split kernel does not take get no copy attached
    assert not hasattr(D.attach_split(W), "_mfpa_split")
