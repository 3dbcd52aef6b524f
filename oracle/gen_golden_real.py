"""Real-input fixtures + goldens (VERDICT r1 item 7 / SURVEY.md section 8c; build container only).

Runs the restated loaders (oracle/loader_oracle.py) on the reference's own example scenes, feeds the result to the
REAL reference model (oracle/ref_shim.py, seeded synthetic state dict -- no checkpoint offline) and stores
  tests/golden/real/<scene>_<i>.png      the resized uint8 frames (lossless; ToTensor(frame) is exactly the loader's tensor)
  tests/golden/real/<case>_inputs.npz    cameras / depth / index lists as the loader produced them
  tests/golden/real/<case>.npz           sub-sampled reference outputs (same layout as gen_golden.py)
for three cases:
  office_pad518      BASELINE configs[0]: example/office first 4 frames, load_and_preprocess_images(mode="pad") -> 518 x 518,
                     images only (zero placeholders, empty index lists)
  office_392_cams    the same frames as inference.py loads them (load_images_and_cameras): 392 x 518, cameras on all 4
  infinigen_294_aux  example/infinigen: 4 frames 294 x 518 with depth (.npy, sky = 1e10 filtered) and cameras on all 4

    python oracle/gen_golden_real.py        # ~3 min on 8 cores
"""
import json
import os
import sys
import time

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import gen_golden as gg  # noqa: E402
import loader_oracle as lo  # noqa: E402
import ref_shim  # noqa: E402
from omnivggt_official_amd import weights  # noqa: E402

REAL = os.path.join(gg.GOLD, "real")
EX = os.path.join(ref_shim.REFERENCE_ROOT, "example")


def save_frames(scene, images):
    for i, img in enumerate(images):
        a = (img.permute(1, 2, 0) * 255).round().to(torch.uint8).numpy()
        assert torch.equal(torch.from_numpy(a).permute(2, 0, 1).float().div(255), img)      # lossless round trip
        Image.fromarray(a, "RGB").save(os.path.join(REAL, "%s_%d.png" % (scene, i)), optimize=True)


def cases():
    office = os.path.join(EX, "office")
    inf = os.path.join(EX, "infinigen")
    paths = sorted(p for p in os.listdir(os.path.join(office, "images")))[:4]
    pad = lo.load_and_preprocess_images_pad([os.path.join(office, "images", p) for p in paths])
    S = pad.shape[0]
    z = torch.zeros
    yield "office_pad518", "office", None, (pad, z(1, S, 3, 4), z(1, S, 3, 3), z(1, S, 518, 518, 1), z(1, S, 518, 518), [], [])
    o = lo.load_images_and_cameras(os.path.join(office, "images"), os.path.join(office, "cameras"), None, limit=4)
    yield "office_392_cams", "office", o[0], o
    f = lo.load_images_and_cameras(os.path.join(inf, "images"), os.path.join(inf, "cameras"), os.path.join(inf, "depths"), limit=4)
    yield "infinigen_294_aux", "infinigen", f[0], f


def main():
    os.makedirs(REAL, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    manifest = json.load(open(os.path.join(gg.GOLD, "state_dict_manifest.json")))
    sd = weights.synthetic_state_dict(manifest, seed=2)
    model = ref_shim.build_reference_model()
    model.load_state_dict(sd, strict=True)
    report = {}
    for name, scene, frames, (images, ext, intr, depth, mask, dgi, cgi) in cases():
        if frames is not None:
            save_frames(scene, frames)
        captured = {}

        def hook(mod, args, out, captured=captured):
            captured["toks"] = out[0]
        t0 = time.time()
        with torch.no_grad():
            h = model.aggregator.register_forward_hook(hook)
            ref = model(images, ext, intr, depth, mask, list(dgi), list(cgi))
            h.remove()
        dt = time.time() - t0
        gold = gg.sample_tokens(captured["toks"])
        gold["tok_absmean"] = np.array([float(t.abs().mean()) for t in captured["toks"]], dtype=np.float64)
        gold["pose_enc"] = ref["pose_enc"].numpy()
        gold["depth"] = ref["depth"][0, :, ::37, ::37, 0].contiguous().numpy()
        gold["depth_conf"] = ref["depth_conf"][0, :, ::37, ::37].contiguous().numpy()
        gold["world_points"] = ref["world_points"][0, :, ::37, ::37].contiguous().numpy()
        gold["world_points_conf"] = ref["world_points_conf"][0, :, ::37, ::37].contiguous().numpy()
        np.savez_compressed(os.path.join(REAL, name + ".npz"), **gold)
        np.savez_compressed(os.path.join(REAL, name + "_inputs.npz"), extrinsics=ext.numpy(), intrinsics=intr.numpy(),
                            depth=depth[0, ..., 0].numpy(), depth_gt_index=np.array(dgi, dtype=np.int64),
                            camera_gt_index=np.array(cgi, dtype=np.int64), hw=np.array(images.shape[-2:]))
        report[name] = {"seconds_reference_cpu": dt, "cpu_threads": torch.get_num_threads(), "views": int(images.shape[0]),
                        "hw": [int(images.shape[-2]), int(images.shape[-1])], "depth_gt_index": list(dgi), "camera_gt_index": list(cgi),
                        "frames_per_s_reference_cpu": images.shape[0] / dt}
        print(name, "%.1fs" % dt, images.shape, "depth", dgi, "cams", cgi, flush=True)
    json.dump(report, open(os.path.join(REAL, "report.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
