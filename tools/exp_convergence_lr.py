#!/usr/bin/env python3
"""lr 1e-3 (the reference's): fp32 twice, plain bf16 with float32 / bfloat16 activations twice each -- the chaotic regime's spread, for the gate of
tests/test_gpu_train.py::test_plain_bf16_train_step_converges_where_fp32_does."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from musicfpaugment_amd.training.selfcheck import run_convergence
lr = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3
res = run_convergence({"fp32 a": (0, 0), "fp32 b": (0, 0), "fp32 c": (0, 0), "z32 a": (2, 2, False), "z32 b": (2, 2, False), "z32 c": (2, 2, False),
                       "z16 a": (2, 2, True), "z16 b": (2, 2, True), "z16 c": (2, 2, True)}, lr=lr)
for k, v in res.items():
    print(f"{k}: train {v[0]:.5f} held-out {v[1]:.5f}")
