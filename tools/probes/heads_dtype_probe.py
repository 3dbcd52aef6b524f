"""16-bit aggregator with 16-bit heads (default) vs exact-f32 heads (OmniVGGT(head_dtype=torch.float32), the reference's arrangement --
it disables autocast around the heads, omnivggt.py:45): prediction error against the REFERENCE goldens (tests/golden) and end-to-end time.

    python tools/probes/heads_dtype_probe.py        (on an MI355X)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))  # tests/common.py (the goldens checker) imports the oracle; this probe is a test-side measurement, not product
import common  # noqa: E402
from omnivggt_official_amd.model import OmniVGGT  # noqa: E402

DEV = "cuda"


def build(dtype, head_dtype):
    with torch.device("meta"):
        m = OmniVGGT(compute_dtype=dtype, head_dtype=head_dtype)
    m = m.to_empty(device="cpu")
    m.load_state_dict(common.full_state_dict(), strict=True)
    return m.to(DEV).eval()


def main():
    print("# max-rel vs the f32 REFERENCE golden (full depth); heads: the aggregator's dtype (default) vs exact f32 (head_dtype=torch.float32)")
    for dtype in (torch.bfloat16, torch.float16):
        for hd in (None, torch.float32):
            m = build(dtype, hd)
            for name in ("s3_partial_aux", "s2_392x518_aux"):
                S, dgi, cgi, hw = common.case(name)
                inp = common.inputs_for(S, DEV, hw=hw)
                with torch.no_grad():
                    out = m(inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], dgi, cgi)
                gold = common.load_golden(name)
                errs = {"pose_enc": common.max_rel(out["pose_enc"].cpu(), gold["pose_enc"]),
                        "depth": common.max_rel(out["depth"][0, :, ::37, ::37, 0].cpu(), gold["depth"]),
                        "depth_conf": common.max_rel(out["depth_conf"][0, :, ::37, ::37].cpu(), gold["depth_conf"]),
                        "world_points": common.max_rel(out["world_points"][0, :, ::37, ::37].cpu(), gold["world_points"])}
                print("%-8s aggregator, %-7s heads, %-15s %s" % (str(dtype).replace("torch.", ""), "f32" if hd else "16-bit", name,
                                                                   "  ".join("%s %.2e" % kv for kv in errs.items())), flush=True)
            inp = common.inputs_for(8, DEV)
            run = lambda: m(inp["images"], inp["extrinsics"], inp["intrinsics"], inp["depth"], inp["mask"], [], [])
            with torch.no_grad():
                run(); run()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    run()
                torch.cuda.synchronize()
            print("%-8s aggregator, %-7s heads: 8-view OmniVGGT.forward %.1f ms" % (str(dtype).replace("torch.", ""), "f32" if hd else "16-bit",
                                                                                     (time.perf_counter() - t0) / 5 * 1e3), flush=True)
            del m
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
