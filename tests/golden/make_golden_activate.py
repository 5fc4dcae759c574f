#!/usr/bin/env python
"""Generate tests/golden/activate_vectors.npz by running the reference's OWN `activate_stream` and `lerf_localization`
(/root/reference/evaluate_iou_loc.py:93-226) and `smooth` (eval/utils.py:55-64) in this container.  Only data is stored.

    python tests/golden/make_golden_activate.py

The reference functions run unmodified; what is replaced is what this container lacks or what writes files:
  * cv2.filter2D -> `filter2d_box` below: the same correlation (anchor = kernel centre ksize // 2, BORDER_REFLECT_101, a
    [H,W,1] input comes back as [H,W]), evaluated as an exact float64 box sum rounded to float32.  (OpenCV itself
    evaluates a 30x30 kernel through a DFT: its output differs from the exact mean by ~1e-7.)
  * colormap_saving / show_result / vis_mask_save / colormaps.apply_colormap: recording stand-ins -- they capture the
    tensors the reference would have written to disk (heat map, final mask, localisation coordinates).
  * clip_model: an object with `positives` and a `get_max_across` that returns the stored relevancy maps (the relevancy
    itself is pinned by next_vectors.npz rel_*).
"""
import os
import sys
import types
from pathlib import Path

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "activate_vectors.npz")


def filter2d_box(src, ddepth, kernel):
    src = np.asarray(src)
    squeeze = src.ndim == 3 and src.shape[2] == 1
    a = src[..., 0] if squeeze else src
    kh, kw = kernel.shape
    ay, ax = kh // 2, kw // 2
    assert np.allclose(kernel, kernel.flat[0])
    p = np.pad(a.astype(np.float64), ((ay, kh - 1 - ay), (ax, kw - 1 - ax)), mode="reflect")  # numpy 'reflect' = REFLECT_101
    c = np.zeros((p.shape[0] + 1, p.shape[1] + 1))
    c[1:, 1:] = p.cumsum(0).cumsum(1)
    h, w = a.shape
    s = c[kh:kh + h, kw:kw + w] - c[:h, kw:kw + w] - c[kh:kh + h, :w] + c[:h, :w]
    return (s * float(kernel.flat[0])).astype(a.dtype)


def main():
    class _Sub:
        def __class_getitem__(cls, item):
            return cls
    stubs = ("plyfile", "cv2", "simple_knn", "simple_knn._C", "gsplat", "open_clip", "torchvision", "torchvision.transforms",
             "matplotlib", "matplotlib.pyplot", "matplotlib.patches", "mediapy", "jaxtyping", "tqdm")
    for name in stubs:
        sys.modules[name] = types.ModuleType(name)
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
    sys.modules["simple_knn._C"].distCUDA2 = lambda *a, **k: None
    sys.modules["gsplat"].rasterization = None
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
    sys.modules["matplotlib"].patches = sys.modules["matplotlib.patches"]
    sys.modules["matplotlib"].use = lambda *a, **k: None
    sys.modules["matplotlib"].colormaps = {}
    sys.modules["jaxtyping"].Bool = sys.modules["jaxtyping"].Float = _Sub
    sys.modules["tqdm"].tqdm = lambda x, *a, **k: x
    sys.modules["cv2"].filter2D = filter2d_box
    sys.path.insert(0, REF)
    import evaluate_iou_loc as E

    rec = {}
    E.colormap_saving = lambda img, opts, path: rec.setdefault("heat", []).append(img[..., 0].clone())
    E.vis_mask_save = lambda mask, path: rec.setdefault("mask", []).append(np.array(mask).copy())
    E.colormaps.apply_colormap = lambda x, *a, **k: (None, torch.zeros(x.shape[0], x.shape[1], 3))

    def show_result(image, save_path, point=None, bbox=None):
        if point is not None:
            rec.setdefault("coords", []).append(np.array(point).copy())
    E.show_result = show_result
    Path.mkdir = lambda self, *a, **k: None  # (the reference creates its output folders)

    g = torch.Generator().manual_seed(777)
    k, h, w = 3, 70, 94
    yy, xx = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
    maps = []
    for j in range(k):
        m = 0.35 + 0.05 * torch.randn(h, w, generator=g)
        for _ in range(2 + j):
            cy, cx = torch.rand(2, generator=g) * torch.tensor([h, w])
            sg = 6 + 10 * torch.rand(1, generator=g)
            m = m + 0.5 * torch.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * sg ** 2))
        maps.append(m.clamp(0, 1))
    valid = torch.stack(maps)                                   # [k, h, w]
    gt = (valid > 0.6).numpy().astype(np.uint8)
    boxes = [np.array([[10, 8, 60, 50], [5, 40, 30, 69]], np.float32), np.array([[0, 0, 93, 69]], np.float32),
             np.array([[80, 60, 93, 69]], np.float32)]
    names = ("alpha", "beta", "gamma")
    clip = types.SimpleNamespace(positives=names, get_max_across=lambda sem: valid.clone()[None])
    ann = {n: {"mask": gt[j], "bboxes": boxes[j]} for j, n in enumerate(names)}
    image = torch.rand(h, w, 3, generator=g)

    out = {"act_valid_map": valid.numpy(), "act_gt_mask": gt, "act_thresh": np.array(0.4, np.float32)}
    for j in range(k):
        out[f"act_boxes{j}"] = boxes[j]
    iou = E.activate_stream(None, image, clip, Path("/nonexistent"), ann, thresh=0.4, colormap_options=None)
    out["act_iou"] = np.array(iou, np.float64)
    out["act_heatmap"] = torch.stack(rec["heat"]).numpy()
    out["act_mask"] = np.stack(rec["mask"]).astype(np.uint8)
    rec.clear()
    acc = E.lerf_localization(None, image, clip, Path("/nonexistent"), ann)
    out["act_loc_acc"] = np.array(acc)
    for j in range(k):
        out[f"act_loc_coords{j}"] = rec["coords"][j].reshape(-1, 2)
    # the majority filter alone, on a random mask, incl. its border rule (eval/utils.py:55-64)
    from eval.utils import smooth
    rm = (torch.rand(37, 41, generator=g) > 0.45).numpy().astype(np.uint8)
    out["act_smooth_in"], out["act_smooth_out"] = rm, smooth(rm)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(out), "arrays; iou", iou, "acc", acc)


if __name__ == "__main__":
    main()
