"""Shared helpers for the test-suite (synthetic weights, golden fixtures, error metrics)."""
import functools
import json
import os

import numpy as np
import torch

import aggregator_oracle as orc
from omnivggt_official_amd import weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
MANIFEST = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))
CASES = {"s2_images_only": (2, [], []), "s3_partial_aux": (3, [1], [0, 2]), "s2_full_aux": (2, [0, 1], [0, 1]),
         "s2_392x518_aux": (2, [1], [0, 1], (392, 518))}            # (S, depth_gt_index, camera_gt_index[, (H, W)])
TOK_LAYERS = (0, 4, 11, 17, 23)
TOK_ROWS = (0, 1, 4, 5, 700, -1)     # -1: last token of the view
GOLD_SEED = 2          # seed used by oracle/gen_golden.py


@functools.lru_cache(maxsize=2)
def full_state_dict(seed=GOLD_SEED):
    return weights.synthetic_state_dict(MANIFEST, seed=seed)


def reduced_state_dict(depth, dino_depth, seed=GOLD_SEED):
    return weights.synthetic_state_dict(weights.reduce_manifest(MANIFEST, depth, dino_depth), seed=seed)


def load_golden(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz")))


def max_rel(a, b):
    """max|a-b| / max|b|  (the tolerance metric of SURVEY.md section 8c)."""
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def sample_tokens(toks, layer):
    return toks[layer][0][:, list(TOK_ROWS)][..., ::8]


def case(name):
    """(S, depth_gt_index, camera_gt_index, hw) of a golden case; hw = 518 (square) unless the case names (H, W)."""
    c = CASES[name]
    return c[0], list(c[1]), list(c[2]), (c[3] if len(c) > 3 else 518)


def inputs_for(S, device="cpu", hw=518):
    inp = orc.synthetic_inputs(S, hw=hw)
    return {k: v.to(device) for k, v in inp.items()}
