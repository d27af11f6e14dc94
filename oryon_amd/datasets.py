"""Test-time readers of the reference's fixed evaluation splits (REAL275 / NOCS and TOYL), i.e. what `system.get_test_dataloader()`
feeds `FPM_Pipeline.test_step` in the reference's run_test.py:41-42.

Restated from the reference's on-disk contract (datasets.py:369-544 NOCSDataset, :546-714 TOYLDataset, utils/data/nocs.py:163-278,
utils/data/toyl.py:95-196) for `eval=True` only - no training split, no augmentations, no correspondence sampling beyond passing the
stored ground-truth correspondences through.  One `FixedSplit` object serves both datasets; the two layouts differ in where the
per-image annotations live:

    <root>/<name>/templates.json                  80 prompt templates ("a photo of a {}." ...)
    <root>/<name>/object_splits.json              {"<obj>": [category ids ...]}        which categories a run evaluates
    <root>/<name>/fixed_split/<split>/instance_list.txt
          NOCS line:  "<part>, <scene_a> <img_a>, <scene_q> <img_q>, <cat_id> <obj_name>"
          TOYL line:  "<part>, <scene_a> <img_a>, <scene_q> <img_q>, <cls_id>"
    <root>/<name>/fixed_split/<split>/annots.pkl  {"<sa>_<ia>_<sq>_<iq>_<cat>[_<obj_name>]": {"gt": 4x4 (translation mm), "corrs": [n,4]}}
    NOCS:  split/real_test/scene_<s>/<img:04d>_{color,mask,depth}.png, _meta.txt ("<mask_id> <cls_id> <obj_name>"), _detection.txt,
           gts/real_test/results_real_test_scene_<s>_<img:04d>.pkl ({"gt_RTs": [k,4,4]}, scaled rotations), obj_names.json,
           obj_models/real_test/{models_info.json, <obj>_vertices.txt}
    TOYL:  split/test/<scene:06d>/{rgb,mask_visib,depth}/<img:06d>.png, scene_gt.json, scene_gt_info.json, models_name.json,
           models_bop/{models_info.json, obj_<id:06d>.ply}

Items come out in the shape `oryon_amd.data.preprocess_item` / `DeviceCollate` expect (the reference's `get_item_data` dict), so
`DeviceCollate(max_corrs)([split[i] for i in idx])` is the batch `Pipeline.test_step*` consumes.  PNG decoding runs on the host through
PIL, exactly as in the reference's dataloader workers.  Nothing here touches the GPU.
"""
from __future__ import annotations

import json
import os
import pickle
import struct
from os.path import join
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .data import check_validity, get_mask_type, preprocess_item

NOCS_K = np.asarray([[591.0125, 0, 322.525], [0, 590.16775, 244.11084], [0, 0, 1]])           # datasets.py:398
TOYL_K = np.asarray([[572.4114, 0.0, 325.2611], [0.0, 573.5704, 242.0489], [0.0, 0.0, 1.0]])   # datasets.py:573


def _png(path: str, mode: Optional[str]) -> np.ndarray:
    from PIL import Image
    img = Image.open(path)
    return np.asarray(img.convert(mode) if mode else img)


def read_ply_vertices(path: str) -> np.ndarray:
    """x, y, z of a PLY file's vertex element (ascii or binary_little_endian; what the reference reads through `plyfile`)."""
    with open(path, "rb") as f:
        header, fmt, n_vert, props, in_vertex = [], None, 0, [], False
        while True:
            line = f.readline().decode("ascii", "replace").strip()
            header.append(line)
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n_vert = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                props.append((tok[1], tok[2]))
            elif tok[0] == "end_header":
                break
        names = [p[1] for p in props]
        ix, iy, iz = names.index("x"), names.index("y"), names.index("z")
        if fmt == "ascii":
            rows = [f.readline().split() for _ in range(n_vert)]
            return np.asarray([[float(r[ix]), float(r[iy]), float(r[iz])] for r in rows], dtype=np.float64)
        if fmt != "binary_little_endian":
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
        code = {"float": "f", "float32": "f", "double": "d", "float64": "d", "uchar": "B", "uint8": "B", "char": "b", "int8": "b",
                "short": "h", "int16": "h", "ushort": "H", "uint16": "H", "int": "i", "int32": "i", "uint": "I", "uint32": "I"}
        rec = struct.Struct("<" + "".join(code[t] for t, _ in props))
        raw = f.read(rec.size * n_vert)
        out = np.empty((n_vert, 3), dtype=np.float64)
        for i in range(n_vert):
            v = rec.unpack_from(raw, i * rec.size)
            out[i] = (v[ix], v[iy], v[iz])
        return out


def unique_matches(matches: torch.Tensor) -> torch.Tensor:
    """Distinct rows of an [N,4] match list as float (utils/misc.py:146-164).  The reference builds them through a Python set of
    strings, so its row ORDER changes from process to process (string hashing); here the distinct rows come in lexicographic order."""
    return torch.unique(matches.to(torch.int), dim=0).to(torch.float)


def sample_correspondences(corrs: torch.Tensor, max_corrs: int) -> torch.Tensor:
    """datasets.py:116-136 (test-time branch): distinct matches, then exactly max_corrs of them drawn from the global torch generator
    (torch.multinomial over a uniform weight vector, with replacement only when fewer exist - utils/misc.py:242-254).  Equal to the
    reference in distribution; row-for-row equality is not defined (see unique_matches)."""
    if corrs.shape[0] == 0:
        return torch.zeros((0, 4))
    u = unique_matches(corrs.clone())
    w = torch.ones(u.shape[0], dtype=torch.float64)
    return u[torch.multinomial(w, max_corrs, replacement=max_corrs > u.shape[0])]


def extent_diameter(pts: np.ndarray) -> float:
    """The ADD diameter of the reference: largest side of the axis-aligned bounding box (utils/pcd.py:16-20), not the BOP diameter."""
    xyz = pts[:, :3]
    return float(np.max(xyz.max(axis=0) - xyz.min(axis=0)))


class FixedSplit:
    """`NOCSDataset(args, eval=True)` / `TOYLDataset(args, eval=True)` as one index over a fixed test split.

    kind: 'nocs' | 'toyl';  root/name/split/obj follow configs/config.yaml (dataset.root, dataset.test.{name,split,obj});
    mask_type is `test.mask` ('predicted' and 'oracle' both read the ground-truth instance mask, datasets.py:27-46; 'ovseg' reads the
    stored predicted masks)."""

    def __init__(self, kind: str, root: str, name: str, split: str, obj: str = "all", mask_type: str = "oracle", max_corrs: int = 500,
                 add_description: str = "no"):
        if kind not in ("nocs", "toyl"):
            raise ValueError(f"unknown dataset kind {kind!r}")
        self.kind, self.base = kind, join(root, name)
        self.mask_type = get_mask_type(mask_type, True)
        self.max_corrs = max_corrs
        self.add_description = add_description
        self.K = NOCS_K if kind == "nocs" else TOYL_K
        with open(join(self.base, "templates.json")) as f:
            self.prompt_templates = json.load(f)
        with open(join(self.base, "object_splits.json")) as f:
            wanted = {int(c) for c in json.load(f)[str(obj)]}
        split_dir = join(self.base, "fixed_split", split)
        with open(join(split_dir, "annots.pkl"), "rb") as f:
            annots = pickle.load(f)
        self.instances: List[Tuple] = []
        self.poses: List[np.ndarray] = []
        self.corrs: List[np.ndarray] = []
        with open(join(split_dir, "instance_list.txt")) as f:
            for line in f:
                if not line.strip():
                    continue
                _, ida, idq, cat = [t.strip() for t in line.split(",")]
                sa, ia = (int(n) for n in ida.split())
                sq, iq = (int(n) for n in idq.split())
                if kind == "nocs":
                    cat_id, obj_name = cat.split()
                    cat_id, obj_key = int(cat_id), obj_name
                    annot_key = f"{sa}_{ia}_{sq}_{iq}_{cat_id}_{obj_name}"
                else:
                    cat_id = int(cat)
                    obj_key = cat_id
                    annot_key = f"{sa}_{ia}_{sq}_{iq}_{cat_id}"
                if cat_id not in wanted:
                    continue
                gt = np.array(annots[annot_key]["gt"], dtype=np.float64, copy=True)
                gt[:3, 3] /= 1000.0                                           # stored in millimetres (datasets.py:439 / :611)
                self.instances.append((sa, ia, sq, iq, cat_id, obj_key))
                self.poses.append(gt)
                self.corrs.append(np.asarray(annots[annot_key]["corrs"]))
        if kind == "nocs":
            with open(join(self.base, "obj_names.json")) as f:
                self.obj_names = json.load(f)
            self._nocs_poses: Dict[str, np.ndarray] = {}
        else:
            with open(join(self.base, "models_name.json")) as f:
                self.obj_names = json.load(f)
            self._toyl_scenes: Dict[int, Dict] = {}
        self._models: Dict = {}

    def __len__(self) -> int:
        return len(self.instances)

    # ------------------------------------------------------------------ per-image annotations
    def _nocs_image_poses(self, scene: int, img: int) -> np.ndarray:
        key = f"{scene}_{img}"
        if key not in self._nocs_poses:
            with open(join(self.base, "gts", "real_test", f"results_real_test_scene_{scene}_{img:04d}.pkl"), "rb") as f:
                self._nocs_poses[key] = np.asarray(pickle.load(f)["gt_RTs"], dtype=np.float64)
        return self._nocs_poses[key]

    def _nocs_item(self, scene: int, img: int, obj_name: str) -> dict:
        stem = join(self.base, "split", "real_test", f"scene_{scene}", f"{img:04d}")
        all_poses = self._nocs_image_poses(scene, img)
        meta = {"cls_ids": [], "mask_ids": [], "cls_names": [], "cls_descs": [], "poses": [], "boxes": []}
        with open(stem + "_meta.txt") as fm, open(stem + "_detection.txt") as fd:
            for i, (ml, dl) in enumerate(zip(fm.readlines(), fd.readlines())):
                mask_id, cls_id, name = ml.split()
                if name != obj_name:
                    continue
                pose = all_poses[i].copy()
                pose[:3, :3] = pose[:3, :3] / np.linalg.norm(pose[:3, :3], axis=1)      # NOCS poses carry the object scale (nocs.py:173-176)
                meta["cls_ids"].append(int(cls_id))
                meta["mask_ids"].append(int(mask_id))
                meta["cls_names"].append(self.obj_names[name][0])
                meta["cls_descs"].append(self.obj_names[name][1:])
                meta["poses"].append(pose)
                meta["boxes"].append(tuple(int(v) for v in dl.split()[1:]))
        mask_file = {"oracle": "_mask.png", "ovseg": "_pred_mask.png"}.get(self.mask_type)
        if mask_file is not None:
            mask = _png(stem + mask_file, "L")
        elif self.mask_type in ("san", "oryon"):
            folder = "san_name" if self.mask_type == "san" else "oryon"
            m = _png(join(self.base, folder, f"{scene} {img} {obj_name}.png"), "L")
            mask = np.where(m == 1, meta["mask_ids"][0], 255)
        else:
            raise RuntimeError(f"Mask type {self.mask_type} not implemented.")
        return {"rgb": _png(stem + "_color.png", "RGB"), "mask": mask, "depth": _png(stem + "_depth.png", None), "metadata": meta,
                "instance_id": f"{scene} {img} {obj_name}"}

    def _toyl_scene(self, scene: int) -> Dict:
        if scene not in self._toyl_scenes:
            d = join(self.base, "split", "test", f"{scene:06d}")
            with open(join(d, "scene_gt.json")) as fa, open(join(d, "scene_gt_info.json")) as fi:
                self._toyl_scenes[scene] = (json.load(fa), json.load(fi))
        return self._toyl_scenes[scene]

    def _toyl_item(self, scene: int, img: int, cls_id: int) -> dict:
        gts, infos = self._toyl_scene(scene)
        meta = {"cls_ids": [], "mask_ids": [], "cls_names": [], "cls_descs": [], "poses": [], "boxes": []}
        for i, (g, info) in enumerate(zip(gts[str(img)], infos[str(img)])):
            if int(g["obj_id"]) != int(cls_id):
                continue
            pose = np.eye(4)
            pose[:3, :3] = np.asarray(g["cam_R_m2c"], dtype=np.float64).reshape(3, 3)
            pose[:3, 3] = np.asarray(g["cam_t_m2c"], dtype=np.float64) / 1000.0
            names = self.obj_names[str(int(cls_id))]
            for v in meta.values():                                           # toyl.py:125-130: the per-image dict is keyed by the object
                v.clear()                                                     # id, so of several annotations of one object the LAST wins
            meta["cls_ids"].append(int(cls_id))
            meta["mask_ids"].append(i + 1)                                    # toyl.py:123: masks are numbered by annotation order
            meta["cls_names"].append(names[0])
            meta["cls_descs"].append(names[1:])
            meta["poses"].append(pose)
            meta["boxes"].append(info["bbox_visib"])
        d = join(self.base, "split", "test", f"{scene:06d}")
        sub = {"oracle": "mask_visib", "ovseg": "mask_pred"}.get(self.mask_type)
        if sub is not None:
            mask = _png(join(d, sub, f"{img:06d}.png"), "L")
        elif self.mask_type in ("san", "oryon"):
            folder = "san_name" if self.mask_type == "san" else "oryon"
            m = _png(join(self.base, folder, f"{scene} {img} {cls_id}.png"), "L")
            mask = np.where(m == 1, meta["mask_ids"][0], 255)
        else:
            raise RuntimeError(f"Mask type {self.mask_type} not implemented.")
        return {"rgb": _png(join(d, "rgb", f"{img:06d}.png"), "RGB"), "mask": mask, "depth": _png(join(d, "depth", f"{img:06d}.png"), None),
                "metadata": meta, "instance_id": f"{scene} {img} {cls_id}"}

    def get_item(self, scene: int, img: int, obj_key) -> dict:
        item = self._nocs_item(scene, img, obj_key) if self.kind == "nocs" else self._toyl_item(scene, img, obj_key)
        item["camera"] = self.K
        return item

    # ------------------------------------------------------------------ dataset protocol
    def prompts_for(self, item: dict) -> List[str]:
        """Bare name + the 80 templates (datasets.py:515-532); the bare name is dropped by the text tower (models/vlm.py:67)."""
        name = item["metadata"]["cls_names"][0]
        descs = item["metadata"]["cls_descs"][0]
        if self.add_description == "yes":
            name = f"{descs[0]} {name}"
        elif self.add_description == "wrong":
            name = f"{descs[1]} {name}"
        elif self.add_description == "desconly":
            name = f"{descs[0]} object"
        return [name] + [t.format(name) for t in self.prompt_templates]

    def __getitem__(self, index: int) -> Tuple:
        sa, ia, sq, iq, cat_id, obj_key = self.instances[index]
        instance_id = f"{sa}_{ia}_{sq}_{iq}_{obj_key}"
        item_a = preprocess_item(self.get_item(sa, ia, obj_key))
        item_q = preprocess_item(self.get_item(sq, iq, obj_key))
        prompt = self.prompts_for(item_a)
        corrs = torch.as_tensor(self.corrs[index])
        sampled = sample_correspondences(corrs, self.max_corrs)               # consumed by the training loss / FMR only, not by the pose path
        valid = check_validity(item_a) and check_validity(item_q) and corrs.shape[0] > 0
        return item_a, item_q, prompt, sampled, corrs, self.poses[index], obj_key, instance_id, valid

    # ------------------------------------------------------------------ object models (evaluation)
    def object_info(self, obj_key) -> Dict:
        """{'pts' [N,3] millimetres, 'diameter' (BOP, mm), 'syms' [S,3,4] the BOP symmetry set, 'symmetric' bool}: what the evaluator
        needs (utils/evaluator.py:246-275: ADD-S iff the symmetry set has more than the identity; MSSD / MSPD minimise over it)."""
        if obj_key not in self._models:
            if self.kind == "nocs":
                d = join(self.base, "obj_models", "real_test")
                with open(join(d, "models_info.json")) as f:
                    info = json.load(f)[str(obj_key)]
                with open(join(d, f"{obj_key}_vertices.txt")) as f:
                    pts = np.asarray([[float(v) for v in line.split()[:3]] for line in f if line.strip()]) * 1000.0
            else:
                d = join(self.base, "models_bop")
                with open(join(d, "models_info.json")) as f:
                    info = json.load(f)[str(int(obj_key))]
                pts = read_ply_vertices(join(d, f"obj_{int(obj_key):06d}.ply"))
            from .evaluation import format_sym_set, get_symmetry_transformations
            # the symmetry set the reference's evaluator holds (utils/data/nocs.py:139, utils/data/toyl.py:233: max_sym_disc_step=0.05)
            syms = format_sym_set(get_symmetry_transformations(info, max_sym_disc_step=0.05))
            self._models[obj_key] = {"pts": np.asarray(pts, dtype=np.float64), "diameter": float(info["diameter"]), "syms": syms,
                                     "symmetric": syms.shape[0] > 1}
        return self._models[obj_key]
