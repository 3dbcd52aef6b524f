"""Generate tests/golden/*.npz from the REAL reference (build container only).

For each case: build the reference OmniVGGT (oracle/ref_shim.py), load the seeded synthetic
state dict (strict=True -- proves the key contract), run the reference forward on the seeded
synthetic inputs, run the oracle restatement on the same weights/inputs, assert they agree,
and store sub-sampled reference outputs as small fixtures for the GPU box (where
/root/reference does not exist).

    python oracle/gen_golden.py            # all cases (~3 min on 8 cores)
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import aggregator_oracle as orc  # noqa: E402
import ref_shim  # noqa: E402
from omnivggt_official_amd import weights  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CASES = {
    # name: (S, depth_gt_index, camera_gt_index[, (H, W)])   -- 518 x 518 unless given
    "s2_images_only": (2, [], []),
    "s3_partial_aux": (3, [1], [0, 2]),
    "s2_full_aux": (2, [0, 1], [0, 1]),
    # non-square input = the reference's own example geometry (392 x 518 after load_and_preprocess_images):
    # exercises the bicubic-antialias pos-embed interpolation (vision_transformer.py:180-212) and gh != gw RoPE
    "s2_392x518_aux": (2, [1], [0, 1], (392, 518)),
}
TOK_LAYERS = (0, 4, 11, 17, 23)
TOK_ROWS = (0, 1, 4, 5, 700, -1)        # -1 = last token of the view (1373 at 518 x 518)


def sample_tokens(toks):
    """(B,S,P,2C) list -> {layer: [S, len(TOK_ROWS), 256]} (every 8th channel)."""
    return {"tok_L%d" % l: toks[l][0][:, list(TOK_ROWS)][..., ::8].contiguous().numpy() for l in TOK_LAYERS}


def rel_err(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def main():
    torch.set_num_threads(os.cpu_count())
    manifest = json.load(open(os.path.join(GOLD, "state_dict_manifest.json")))
    t0 = time.time()
    sd = weights.synthetic_state_dict(manifest, seed=2)
    print("synthetic weights: %.1fs" % (time.time() - t0))
    model = ref_shim.build_reference_model()
    missing = model.load_state_dict(sd, strict=True)
    print("reference load_state_dict(strict=True):", missing)
    only = sys.argv[1:]                       # optional: regenerate just these cases, keep the rest of the report
    rep_path = os.path.join(GOLD, "oracle_vs_reference_report.json")
    report = json.load(open(rep_path)) if (only and os.path.exists(rep_path)) else {}
    for name, case in CASES.items():
        if only and name not in only:
            continue
        S, dgi, cgi = case[:3]
        inp = orc.synthetic_inputs(S, hw=case[3] if len(case) > 3 else 518)
        t0 = time.time()
        with torch.no_grad():
            captured = {}

            def hook(mod, args, out, captured=captured):
                captured["toks"] = out[0]
            h = model.aggregator.register_forward_hook(hook)
            ref = model(inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], list(dgi), list(cgi))
            h.remove()
        t_ref = time.time() - t0
        t0 = time.time()
        with torch.no_grad():
            mine = orc.model_forward(sd, inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], list(dgi), list(cgi))
        t_orc = time.time() - t0
        errs = {"tokens_L%d" % l: rel_err(mine["_tokens"][l], captured["toks"][l]) for l in range(24)}
        for k in ("pose_enc", "depth", "depth_conf", "world_points", "world_points_conf"):
            errs[k] = rel_err(mine[k], ref[k])
        worst = max(errs.values())
        print("%s: reference %.1fs, oracle %.1fs, worst rel err oracle-vs-reference %.3e" % (name, t_ref, t_orc, worst))
        assert worst < 2e-5, errs
        report[name] = {"t_reference_s": t_ref, "t_oracle_s": t_orc, "oracle_vs_reference_max_rel": errs,
                        "cpu_threads": torch.get_num_threads()}
        gold = sample_tokens(captured["toks"])
        gold["tok_absmean"] = np.array([float(captured["toks"][l].abs().mean()) for l in range(24)], dtype=np.float64)
        gold["pose_enc"] = ref["pose_enc"].numpy()
        gold["pose_enc_list"] = torch.stack(ref["pose_enc_list"]).numpy()
        gold["depth"] = ref["depth"][0, :, ::37, ::37, 0].contiguous().numpy()
        gold["depth_conf"] = ref["depth_conf"][0, :, ::37, ::37].contiguous().numpy()
        gold["world_points"] = ref["world_points"][0, :, ::37, ::37].contiguous().numpy()
        gold["world_points_conf"] = ref["world_points_conf"][0, :, ::37, ::37].contiguous().numpy()
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **gold)
    json.dump(report, open(rep_path, "w"), indent=1)
    print("done")


if __name__ == "__main__":
    main()
