"""Shared by the CPU pin (tests/test_oracle_golden.py) and the GPU tests of the 16-bit storage paths: the reference's own
bf16 / fp16 outputs (tests/golden/lowp.npz, written by tools/make_golden_lowp.py) and the STATED tolerance against them.

The RankPO kernel keeps its [B, 2] scores in float32 whatever the storage dtype (f32 accumulation of the 16-bit inputs, no rounding
afterwards), where the reference rounds them to the storage dtype and then runs the whole loss chain in it
(rankpo_trainer.py:436-443, 545-566).  The bound below is what ONE unit roundoff u of the storage dtype per reference operation
allows; with u = 2^-8 (bf16: 8 significant bits) or 2^-11 (fp16):

  scores      |s_ref - s| <= u |s|                                              (the matmul output is rounded once)
  logit       z = beta ((c - r - (c_ref - r_ref)) / T - gamma):
              |z_ref - z| <= (beta / T) u (|c| + |r| + |c_ref| + |r_ref|)  +  4 u (|z| + beta gamma)
                             (the four score roundings)                        (sub, div, sub, mul each rounded)
  loss        |l'(z)| <= 1 for the sigmoid and the hinge loss alike, so a sample's loss moves by at most the logit error;
              sft term: CE(s / T, 0) moves by at most 2 u max|s| / T
  the sum     6 u (|L| + 1e-3) for the roundings of logsigmoid, label smoothing, the weights, the mean and the sum.
"""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
UNIT_ROUNDOFF = {"bf16": 2.0 ** -8, "fp16": 2.0 ** -11}


def load():
    g = np.load(os.path.join(GOLDEN, "lowp.npz"), allow_pickle=False)
    return g, json.loads(str(g["meta"]))


def round_to(x, tag):
    """float64 values of x after the cast to the storage dtype (round to nearest even)."""
    from oracle import scoring_ref as R
    if tag == "bf16":
        return R.round_bf16(x).astype(np.float64)
    return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float64)


def rankpo_loss_bound(case, scores, ref_c, ref_r, loss, tag):
    """The stated tolerance on |loss - reference's loss in 16-bit storage| (module docstring); scores: exact [B, 2]."""
    u = UNIT_ROUNDOFF[tag]
    beta, T, gamma = case["beta"], case["temperature"], case["gamma_beta_ratio"]
    with_ref = not case["reference_free"]
    sc = np.asarray(scores, dtype=np.float64)
    rc = np.asarray(ref_c, dtype=np.float64) if with_ref else 0.0
    rr = np.asarray(ref_r, dtype=np.float64) if with_ref else 0.0
    z = beta * ((sc[:, 0] - sc[:, 1] - (rc - rr)) / T - gamma)
    bz = (beta / T) * u * (np.abs(sc).sum(1) + np.abs(rc) + np.abs(rr)) + 4 * u * (np.abs(z) + beta * gamma)
    b_sft = (2 * u / T) * np.abs(sc).max(1).mean()
    return case["rankpo_weight"] * bz.mean() + case["sft_weight"] * b_sft + 6 * u * (abs(loss) + 1e-3)
