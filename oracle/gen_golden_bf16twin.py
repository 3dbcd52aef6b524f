"""Same-precision twin for the 16-bit throughput modes (SURVEY.md section 8c Gate 2; build container only).

Runs the REAL reference (oracle/ref_shim.py, seeded synthetic state dict, seeded synthetic inputs -- the same as
oracle/gen_golden.py) under `torch.autocast('cpu', dtype=torch.bfloat16)` and stores the same sub-sampled outputs as
tests/golden/<case>_bf16twin.npz, plus the twin's OWN error against the f32 reference golden
(tests/golden/bf16twin_report.json).  The GPU tests compare the HIP bf16 / f16 modes with both: their error against
the f32 reference must stay within 2x the twin's (tests/test_gpu_aggregator.py::test_low_precision_modes_vs_twin).

    python oracle/gen_golden_bf16twin.py [case ...]          # ~2-4 min per case on 8 cores
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
import gen_golden as gg  # noqa: E402
import ref_shim  # noqa: E402
from omnivggt_official_amd import weights  # noqa: E402

TWIN_CASES = ("s2_images_only", "s3_partial_aux", "s2_392x518_aux")


def max_rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def rms_rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp(min=1e-30))


def main():
    torch.set_num_threads(os.cpu_count())
    manifest = json.load(open(os.path.join(gg.GOLD, "state_dict_manifest.json")))
    sd = weights.synthetic_state_dict(manifest, seed=2)
    model = ref_shim.build_reference_model()
    model.load_state_dict(sd, strict=True)
    only = sys.argv[1:]
    rep_path = os.path.join(gg.GOLD, "bf16twin_report.json")
    report = json.load(open(rep_path)) if os.path.exists(rep_path) else {}
    for name in TWIN_CASES:
        if only and name not in only:
            continue
        case = gg.CASES[name]
        S, dgi, cgi = case[:3]
        inp = orc.synthetic_inputs(S, hw=case[3] if len(case) > 3 else 518)
        captured = {}

        def hook(mod, args, out, captured=captured):
            captured["toks"] = [t.float() for t in out[0]]
        t0 = time.time()
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            h = model.aggregator.register_forward_hook(hook)
            ref = model(inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], list(dgi), list(cgi))
            h.remove()
        dt = time.time() - t0
        gold = gg.sample_tokens(captured["toks"])
        gold["pose_enc"] = ref["pose_enc"].float().numpy()
        gold["depth"] = ref["depth"][0, :, ::37, ::37, 0].float().contiguous().numpy()
        gold["depth_conf"] = ref["depth_conf"][0, :, ::37, ::37].float().contiguous().numpy()
        gold["world_points"] = ref["world_points"][0, :, ::37, ::37].float().contiguous().numpy()
        gold["world_points_conf"] = ref["world_points_conf"][0, :, ::37, ::37].float().contiguous().numpy()
        np.savez_compressed(os.path.join(gg.GOLD, name + "_bf16twin.npz"), **gold)
        f32 = dict(np.load(os.path.join(gg.GOLD, name + ".npz")))
        errs = {k: {"max_rel": max_rel(gold[k], f32[k]), "rms_rel": rms_rel(gold[k], f32[k])}
                for k in list(gold) if k in f32}
        report[name] = {"seconds": dt, "twin_vs_f32_reference": errs, "torch": torch.__version__}
        print(name, "%.0fs" % dt, {k: "%.2e" % v["max_rel"] for k, v in errs.items()}, flush=True)
    json.dump(report, open(rep_path, "w"), indent=1)


if __name__ == "__main__":
    main()
