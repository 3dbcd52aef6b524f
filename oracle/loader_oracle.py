"""CPU restatement of the reference's input loaders -- TEST INFRASTRUCTURE ONLY (same rules as aggregator_oracle.py).

The reference's loaders cannot be imported in the build container (visual_util.py:1-40 and
omnivggt/utils/load_fn.py need cv2 / torchvision, both absent), so the two functions the hot path's inputs come from
are restated here with PIL + numpy, line by line:

  load_and_preprocess_images_pad   omnivggt/utils/load_fn.py:53-118 (mode="pad": BASELINE configs[0], 518 x 518 with white borders)
  load_images_and_cameras          visual_util.py:679-845 (what inference.py:335 calls: width 518, height a multiple of 14,
                                   optional depth .npy + camera .txt per image)
  load_camera_from_txt             visual_util.py:847-893
  closed_form_inverse_se3          omnivggt/utils/geometry.py:269-318 (numpy branch)

Third-party arithmetic restated from its published definition (the packages are not in /root/reference and not
installed here; requirements.txt pins no version):
  torchvision.transforms.ToTensor  uint8 HWC -> float32 CHW / 255           (omnivggt/utils/image.py:26 `ImgNorm`)
  cv2.resize(..., INTER_NEAREST)   dst(y, x) = src(min(floor(y * sh / dh), sh - 1), min(floor(x * sw / dw), sw - 1)),
                                   scale factors evaluated in double (OpenCV imgproc resize.cpp, resizeNN; INTER_NEAREST
                                   has no half-pixel shift)                  (visual_util.py:785)
PIL's own bicubic resize is the same library the reference calls, so the image path is exact; the depth path is pinned
only by this restatement ("parity unpinned" for cv2's nearest resize -- the real one cannot be run here).
"""
import glob
import os
from pathlib import Path

import numpy as np
import torch
from PIL import Image


def to_tensor(img):
    """torchvision ToTensor on a PIL RGB image: float32 CHW in [0, 1]."""
    a = np.asarray(img, dtype=np.uint8)
    return torch.from_numpy(a.copy()).permute(2, 0, 1).to(torch.float32).div(255)


def open_rgb(path):
    """visual_util.py:722-729 / load_fn.py:75-85: RGBA is composited onto white, everything ends as RGB."""
    img = Image.open(path)
    if img.mode == "RGBA":
        background = Image.new("RGBA", img.size, (255, 255, 255, 255))
        img = Image.alpha_composite(background, img)
    return img.convert("RGB")


def resize_nearest_cv2(src, new_width, new_height):
    """cv2.resize(src, (new_width, new_height), interpolation=cv2.INTER_NEAREST) for a 2-D array."""
    sh, sw = src.shape
    ys = np.minimum(np.floor(np.arange(new_height) * (sh / new_height)).astype(np.int64), sh - 1)
    xs = np.minimum(np.floor(np.arange(new_width) * (sw / new_width)).astype(np.int64), sw - 1)
    return src[ys[:, None], xs[None, :]]


def closed_form_inverse_se3(se3):
    """geometry.py:269-318, numpy branch: float32 R^T and -R^T t written into a float64 identity."""
    R = se3[:, :3, :3]
    T = se3[:, :3, 3:]
    Rt = np.transpose(R, (0, 2, 1))
    top_right = -np.matmul(Rt, T)
    inv = np.tile(np.eye(4), (len(R), 1, 1))
    inv[:, :3, :3] = Rt
    inv[:, :3, 3:] = top_right
    return inv


def load_camera_from_txt(camera_path):
    """visual_util.py:847-893: 3 lines of a 3x4 cam-to-world matrix, 3 lines of a 3x3 intrinsic matrix."""
    with open(camera_path, "r") as f:
        lines = f.readlines()
    lines = [l.strip() for l in lines if l.strip() and not l.strip().startswith("#")]
    if len(lines) < 6:
        return None, None
    ext = []
    for i in range(3):
        values = [float(x) for x in lines[i].split()]
        if len(values) != 4:
            return None, None
        ext.append(values)
    intr = []
    for i in range(3, 6):
        values = [float(x) for x in lines[i].split()]
        if len(values) != 3:
            return None, None
        intr.append(values)
    return np.array(ext, dtype=np.float32), np.array(intr, dtype=np.float32)


def resized_geometry(width, height, target_size=518):
    """visual_util.py:731-747: (new_width, new_height, crop_start_y, final_height)."""
    new_width = target_size
    new_height = round(height * (new_width / width) / 14) * 14
    crop_start_y, final_height = 0, new_height
    if new_height > target_size:
        crop_start_y = (new_height - target_size) // 2
        final_height = target_size
    return new_width, new_height, crop_start_y, final_height


def load_images_and_cameras(image_folder, camera_folder=None, depth_folder=None, target_size=518, max_depth=100, limit=None):
    """visual_util.py:679-845. `limit`: only the first `limit` sorted files (the fixtures use 4 views).
    Returns (images (S,3,H,W), extrinsics (1,S,3,4) w2c, intrinsics (1,S,3,3), depthmaps (1,S,H,W,1), masks (1,S,H,W),
    depth_indices, camera_indices) exactly like the reference."""
    image_paths = sorted(glob.glob(os.path.join(image_folder, "*")))
    image_paths = [p for p in image_paths if p.lower().endswith((".png", ".jpg", ".jpeg"))]
    if limit is not None:
        image_paths = image_paths[:limit]
    img_list, extrinsics_list, intrinsics_list, depths_list, masks_list = [], [], [], [], []
    depth_indices, camera_indices = [], []
    for idx, img_path in enumerate(image_paths):
        basename = Path(img_path).stem
        img = open_rgb(img_path)
        width, height = img.size
        new_width, new_height, crop_start_y, final_height = resized_geometry(width, height, target_size)
        scale_x = new_width / width
        scale_y = new_height / height
        img = img.resize((new_width, new_height), Image.Resampling.BICUBIC)
        if new_height > target_size:
            img = img.crop((0, crop_start_y, new_width, crop_start_y + target_size))
        img_list.append(to_tensor(img))

        depthmap = None
        if depth_folder is not None:
            depth_path = os.path.join(depth_folder, f"{basename}.npy")        # (.png depth needs cv2.imread: not restated)
            if os.path.exists(depth_path):
                depthmap = np.load(depth_path).astype(np.float32)
                depthmap[~np.isfinite(depthmap)] = 0
                depthmap[depthmap > max_depth] = 0
                depthmap[depthmap < 1e-5] = 0
        if depthmap is not None:
            depth_indices.append(idx)
            depthmap = resize_nearest_cv2(depthmap, new_width, new_height)
            if new_height > target_size:
                depthmap = depthmap[crop_start_y: crop_start_y + target_size, :]
            mask = depthmap > 1e-5
        else:
            depthmap = np.zeros((final_height, new_width), dtype=np.float32)
            mask = np.zeros_like(depthmap, dtype=bool)
        depths_list.append(depthmap)
        masks_list.append(mask)

        extrinsic = intrinsic = None
        if camera_folder is not None:
            cam = os.path.join(camera_folder, f"{basename}.txt")
            if os.path.exists(cam):
                extrinsic, intrinsic = load_camera_from_txt(cam)
        if extrinsic is not None and intrinsic is not None:
            camera_indices.append(idx)
            intrinsic[0, 0] *= scale_x
            intrinsic[1, 1] *= scale_y
            intrinsic[0, 2] *= scale_x
            intrinsic[1, 2] *= scale_y
            if new_height > target_size:
                intrinsic[1, 2] -= crop_start_y
            extrinsic = closed_form_inverse_se3(extrinsic[None])[0][:3]
        else:
            extrinsic = np.zeros((3, 4), dtype=np.float32)
            intrinsic = np.zeros((3, 3), dtype=np.float32)
        extrinsics_list.append(extrinsic)
        intrinsics_list.append(intrinsic)

    images = torch.stack(img_list, dim=0)
    depthmaps = torch.from_numpy(np.array(depths_list))[None, ..., None].float()
    masks = torch.from_numpy(np.array(masks_list))[None, ...].float()
    extrinsics = torch.from_numpy(np.array(extrinsics_list))[None, ...].float()
    intrinsics = torch.from_numpy(np.array(intrinsics_list))[None, ...].float()
    return images, extrinsics, intrinsics, depthmaps, masks, depth_indices, camera_indices


def load_and_preprocess_images_pad(image_path_list, target_size=518):
    """load_fn.py:53-118 with mode="pad": largest side -> 518 (other side rounded to a multiple of 14), white borders
    to 518 x 518. Returns (S, 3, 518, 518)."""
    images = []
    for image_path in sorted(image_path_list):
        img = open_rgb(image_path)
        width, height = img.size
        if width >= height:
            new_width = target_size
            new_height = round(height * (new_width / width) / 14) * 14
        else:
            new_height = target_size
            new_width = round(width * (new_height / height) / 14) * 14
        img = to_tensor(img.resize((new_width, new_height), Image.Resampling.BICUBIC))
        h_padding = target_size - img.shape[1]
        w_padding = target_size - img.shape[2]
        if h_padding > 0 or w_padding > 0:
            pad_top = h_padding // 2
            pad_left = w_padding // 2
            img = torch.nn.functional.pad(img, (pad_left, w_padding - pad_left, pad_top, h_padding - pad_top), mode="constant", value=1.0)
        images.append(img)
    return torch.stack(images)


def pad_to_square(images, target_size=518):
    """The padding step of load_and_preprocess_images(mode="pad") on already-resized (S,3,h,w) tensors (used by the
    GPU-box tests, which only have the resized fixtures)."""
    h_padding = target_size - images.shape[-2]
    w_padding = target_size - images.shape[-1]
    pad_top, pad_left = h_padding // 2, w_padding // 2
    return torch.nn.functional.pad(images, (pad_left, w_padding - pad_left, pad_top, h_padding - pad_top), mode="constant", value=1.0)
