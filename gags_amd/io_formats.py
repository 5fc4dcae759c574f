"""On-disk formats of the reference that carry the tensors of the hot path, without the
`plyfile` dependency:

* point_cloud.ply -- binary little-endian PLY, one `vertex` element of float32 properties in the order of
  scene/gaussian_model.py:222-237 (`construct_list_of_attributes`): x y z nx ny nz f_dc_* f_rest_* opacity scale_*
  rot_* semantic_*; written / read as :240-259 / :266-318 do (f_dc / f_rest stored channel-major: `transpose(1,2)`).
* chkpnt*.pth     -- `torch.save((gaussians.capture(), iteration))`, capture() being the 13-tuple of
  scene/gaussian_model.py:63-78 (a 12-tuple for an RGB checkpoint without features: :80-113, train.py:82-94).
* <image>_f.npy [n_seg, 512] / <image>_s.npy [4, h, w] -- per-image segment embeddings and the four segment-id
  maps (default + s / m / l granularity, -1 = none) written by preprocess.py:332-336.
"""
import os

import numpy as np
import torch

_PLY_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1",
              "char": "i1", "int8": "i1", "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2",
              "int": "<i4", "int32": "<i4", "uint": "<u4", "uint32": "<u4"}


def ply_attribute_names(n_dc, n_rest, n_scale, n_rot, n_semantic):
    """Property names in file order (scene/gaussian_model.py:222-237)."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(n_dc)]
    names += [f"f_rest_{i}" for i in range(n_rest)]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(n_scale)]
    names += [f"rot_{i}" for i in range(n_rot)]
    names += [f"semantic_{i}" for i in range(n_semantic)]
    return names


def write_ply(path, xyz, features_dc, features_rest, opacity, scaling, rotation, semantic_feature=None):
    """Raw (pre-activation) tensors in the reference layout -> point_cloud.ply."""
    def a(t):
        return t.detach().cpu().numpy().astype(np.float32) if torch.is_tensor(t) else np.asarray(t, np.float32)

    xyz = a(xyz)
    n = xyz.shape[0]
    f_dc = a(features_dc).transpose(0, 2, 1).reshape(n, -1)       # [N,1,3] -> [N,3,1] -> [N,3]
    f_rest = a(features_rest).transpose(0, 2, 1).reshape(n, -1)   # [N,15,3] -> [N,3,15] -> [N,45]
    sem = np.zeros((n, 0), np.float32) if semantic_feature is None else a(semantic_feature)
    cols = [xyz, np.zeros_like(xyz), f_dc, f_rest, a(opacity).reshape(n, 1), a(scaling), a(rotation), sem]
    table = np.ascontiguousarray(np.concatenate(cols, axis=1), dtype="<f4")
    names = ply_attribute_names(f_dc.shape[1], f_rest.shape[1], cols[5].shape[1], cols[6].shape[1], sem.shape[1])
    assert len(names) == table.shape[1]
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n
    header += "".join(f"property float {nm}\n" for nm in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(table.tobytes())


def read_ply_table(path):
    """(names, structured array) of the first element of a binary little-endian (or ascii) PLY."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_first, seen = None, None, [], False, 0
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: no end_header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                seen += 1
                in_first = seen == 1
                if in_first:
                    count = int(tok[2])
            elif tok[0] == "property" and in_first:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        dt = np.dtype(props)
        if fmt == "binary_little_endian":
            data = np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)
        elif fmt == "ascii":
            data = np.loadtxt(f, dtype=np.float64, max_rows=count, ndmin=2)
            out = np.empty(count, dtype=dt)
            for i, (nm, _) in enumerate(props):
                out[nm] = data[:, i]
            data = out
        else:
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
    return [p[0] for p in props], data


def read_ply(path, max_sh_degree=3):
    """point_cloud.ply -> dict of float32 arrays in the reference's tensor layout (scene/gaussian_model.py:266-318):
    xyz [N,3], features_dc [N,1,3], features_rest [N,(deg+1)^2-1,3], opacity [N,1], scaling [N,3], rotation [N,4],
    semantic_feature [N,D] or None."""
    names, d = read_ply_table(path)

    def col(nm):
        return np.asarray(d[nm], np.float32)

    def numbered(prefix):
        return sorted([nm for nm in names if nm.startswith(prefix)], key=lambda s: int(s.split("_")[-1]))

    xyz = np.stack([col("x"), col("y"), col("z")], axis=1)
    n = xyz.shape[0]
    f_dc = np.stack([col(f"f_dc_{i}") for i in range(3)], axis=1).reshape(n, 3, 1)
    rest_names = numbered("f_rest_")
    if len(rest_names) != 3 * (max_sh_degree + 1) ** 2 - 3:
        raise ValueError(f"{path}: {len(rest_names)} f_rest_* columns, expected {3 * (max_sh_degree + 1) ** 2 - 3}")
    f_rest = np.stack([col(nm) for nm in rest_names], axis=1).reshape(n, 3, (max_sh_degree + 1) ** 2 - 1)
    sem_names = numbered("semantic_")
    return {
        "xyz": xyz,
        "features_dc": np.ascontiguousarray(f_dc.transpose(0, 2, 1)),
        "features_rest": np.ascontiguousarray(f_rest.transpose(0, 2, 1)),
        "opacity": col("opacity")[:, None],
        "scaling": np.stack([col(nm) for nm in numbered("scale_")], axis=1),
        "rotation": np.stack([col(nm) for nm in numbered("rot")], axis=1),
        "semantic_feature": np.stack([col(nm) for nm in sem_names], axis=1) if sem_names else None,
    }


def load_language_features(prefix, render_hw=None, device="cpu"):
    """`<prefix>_f.npy` [n_seg, 512] and `<prefix>_s.npy` [4, h, w] (preprocess.py:332-336) -> (img_embed, seg_map);
    with render_hw=(H, W) the segment maps are nearest-resized to the render resolution as utils/camera_utils.py:61
    does (ids must not be interpolated)."""
    img_embed = torch.from_numpy(np.load(prefix + "_f.npy").astype(np.float32))
    seg_map = torch.from_numpy(np.load(prefix + "_s.npy").astype(np.float32))
    if render_hw is not None and tuple(seg_map.shape[1:]) != tuple(render_hw):
        seg_map = torch.nn.functional.interpolate(seg_map[None], size=tuple(render_hw), mode="nearest")[0]
    return img_embed.to(device), seg_map.contiguous().to(device)


def save_checkpoint(path, gaussians, iteration):
    """train.py:230-232: torch.save((gaussians.capture(), iteration), path)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save((gaussians.capture(), iteration), path)


def load_checkpoint(path, map_location="cpu"):
    """-> (model_args tuple of 12 or 13 entries, iteration); train.py:82-94 tells the two apart by length."""
    model_args, iteration = torch.load(path, map_location=map_location, weights_only=False)
    if len(model_args) not in (12, 13):
        raise ValueError(f"{path}: {len(model_args)}-tuple, expected 12 (RGB field) or 13 (feature field)")
    return model_args, iteration
