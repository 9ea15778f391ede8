"""On-disk formats of the reference, so real checkpoints / results can be dropped in without code changes
(SURVEY.md section 8 f3).

* training checkpoints ``models.pth`` / ``epoch_N.pth`` -- dict written by ``save_models``
  (src/trainer_unified_nuscenes.py:476-490): ``model_params`` (state-dict), ``shape_code_params`` /
  ``texture_code_params`` (nn.Embedding state-dicts), ``niter``, ``nepoch``, ``instoken2idx``, ``optimized_idx``;
* optimisation results ``codes+poses.pth`` -- dict written by ``save_opts_w_pose`` (src/optimizer_nuscenes.py:1463-1476),
  consumed by ``scripts/eval_saved_result.py`` / ``collect_eval_results`` (src/utils.py:786).
Pure host code (torch.save / torch.load)."""
import os
from typing import Dict, Optional

import torch


def save_checkpoint(path: str, model: torch.nn.Module, shape_codes: torch.Tensor, texture_codes: torch.Tensor, niter: int, nepoch: int,
                    instoken2idx: Optional[Dict[str, int]] = None, optimized_idx: Optional[torch.Tensor] = None):
    """Write a checkpoint with the reference's keys (embedding tables as ``{'weight': (n_inst, 256)}``)."""
    n = shape_codes.shape[0]
    d = {"model_params": {k: v.detach().cpu() for k, v in model.state_dict().items()},
         "shape_code_params": {"weight": shape_codes.detach().cpu()},
         "texture_code_params": {"weight": texture_codes.detach().cpu()},
         "niter": int(niter), "nepoch": int(nepoch),
         "instoken2idx": instoken2idx if instoken2idx is not None else {str(i): i for i in range(n)},
         "optimized_idx": optimized_idx if optimized_idx is not None else torch.ones(n)}
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(d, path)


def load_checkpoint(path: str, model: torch.nn.Module, strict: bool = False):
    """Load ``model_params`` into ``model`` (non-strict like ``resume_from_epoch`` :500, so encoder keys the caller's
    module lacks are skipped) and derive the mean codes exactly as ``load_model`` does
    (src/optimizer_nuscenes.py:1799-1808): mean over the instances whose ``optimized_idx > 0`` when that key exists.
    Returns (mean_shape (1,256), mean_texture (1,256), saved_data)."""
    saved = torch.load(path, map_location=torch.device("cpu"), weights_only=False)
    missing = model.load_state_dict(saved["model_params"], strict=strict)
    sw, tw = saved["shape_code_params"]["weight"], saved["texture_code_params"]["weight"]
    if "optimized_idx" in saved:
        keep = torch.as_tensor(saved["optimized_idx"]).numpy() > 0
        mean_shape, mean_texture = sw[keep].mean(0).reshape(1, -1), tw[keep].mean(0).reshape(1, -1)
    else:
        mean_shape, mean_texture = sw.mean(0).reshape(1, -1), tw.mean(0).reshape(1, -1)
    return mean_shape, mean_texture, saved, missing


def save_opts_w_pose(save_dir: str, num_obj: int, shapecodes: dict, texturecodes: dict, poses: dict, psnr_eval: dict, depth_err_mean: dict,
                     R_eval: dict, T_eval: dict, ssim_eval: Optional[dict] = None, lidar_pts_cnt: Optional[dict] = None) -> str:
    """``codes+poses.pth`` with the reference's schema; the dicts are keyed like the reference's
    (``optimized_*[anntoken][cam_id]`` tensors, metric dicts keyed ``f'{anntoken}_{cam_id}'`` -> list per iteration)."""
    d = {"num_obj": num_obj, "optimized_shapecodes": shapecodes, "optimized_texturecodes": texturecodes, "optimized_poses": poses,
         "psnr_eval": psnr_eval, "ssim_eval": ssim_eval or {}, "depth_err_mean": depth_err_mean, "lidar_pts_cnt": lidar_pts_cnt or {},
         "R_eval": R_eval, "T_eval": T_eval}
    os.makedirs(save_dir, exist_ok=True)
    path = os.path.join(save_dir, "codes+poses.pth")
    torch.save(d, path)
    return path


def metric_rows_to_eval_dicts(metrics: torch.Tensor, ids, cam_id: int = 0, n_lidar=64):
    """(n_objects, num_opts*4) rows of ``driver.optimize_objects`` -> the per-iteration metric dicts of the reference, with the element
    types its reader relies on (``collect_eval_results``, src/utils.py:786-880): ``psnr_eval`` / ``depth_err_mean`` lists of floats
    (``np.array(list)[:max_iter]``), ``R_eval`` / ``T_eval`` lists of 0-dim tensors (``torch.stack(list)``, they come from
    ``calc_pose_err``, src/optimizer_nuscenes.py:1694-1702), ``lidar_pts_cnt`` one count per key (the weights of the depth-error mean:
    ``n_lidar`` is the per-object count array the driver reports -- ``info["lidar_count"]`` / ``optimize_objects(return_counts=True)`` --
    or one integer for all).  Returns (psnr_eval, depth_err_mean, R_eval, T_eval, lidar_pts_cnt)."""
    ids = list(ids)
    m = metrics.view(metrics.shape[0], -1, 4).cpu()
    counts = [int(n_lidar)] * len(ids) if isinstance(n_lidar, (int, float)) else [int(v) for v in torch.as_tensor(n_lidar).reshape(-1).tolist()]
    if len(counts) != len(ids) or m.shape[0] != len(ids):
        raise ValueError(f"metric_rows_to_eval_dicts: {m.shape[0]} metric rows, {len(ids)} ids, {len(counts)} lidar counts")
    psnr, depth, R, T, cnt = {}, {}, {}, {}, {}
    for row, i, n_i in zip(m, ids, counts):
        key = f"{i}_{cam_id}"
        psnr[key], depth[key] = row[:, 0].tolist(), row[:, 1].tolist()
        R[key], T[key] = list(row[:, 2].clone().unbind(0)), list(row[:, 3].clone().unbind(0))
        cnt[key] = n_i
    return psnr, depth, R, T, cnt


def save_driver_results(save_dir: str, metrics: torch.Tensor, ids, shapecodes=None, texturecodes=None, poses=None, cam_id: int = 0,
                        n_lidar=64) -> str:
    """``driver.optimize_objects`` output -> ``codes+poses.pth`` that ``scripts/eval_saved_result.py`` plots unchanged.  ``n_lidar``: the
    per-object depth-pixel counts the driver reports (or one integer for all)."""
    ids = list(ids)                          # (a generator would be empty after the first pass)
    psnr, depth, R, T, cnt = metric_rows_to_eval_dicts(metrics, ids, cam_id, n_lidar)
    return save_opts_w_pose(save_dir, len(ids), shapecodes or {}, texturecodes or {}, poses or {}, psnr, depth, R, T, lidar_pts_cnt=cnt)
