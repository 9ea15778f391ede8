#!/usr/bin/env python3
"""Round-2 golden vectors, generated from the REFERENCE itself (build container only; needs /root/reference).

Adds, without touching the round-1 fixtures (``gen_golden.py``):

* ``kitti.npz`` / ``kitti.json`` -- the host geometry of the cross-domain KITTI loop (BASELINE config 4):
  ``obj_pose_kitti2nusc`` (src/utils.py:1354-1366) on synthetic KITTI labels incl. the in-place write,
  ``roi_process`` (src/utils.py:1392-1415) with the nuScenes (5) and KITTI (15) margins on boxes that stay inside / leave the image,
  ``sample_from_rays_v2`` (src/utils.py:170-184), ``calc_pose_err`` / ``rot_dist`` (src/utils.py:675-722), one end-to-end
  ``render_rays_v2`` call on a KITTI object whose crop is NOT im_sz^2 (so the bilinear resize and the int32 mask truncation run),
  and the values of jsonfiles/supnerf.kitti.car.json / supnerf.nusc.vehicle.car.json that the loop reads;
* ``formats.npz`` -- the on-disk formats seen from the reference's side: a ``codes+poses.pth`` written by
  ``supnerf_amd.io.save_driver_results`` is read by the reference's own ``collect_eval_results`` (src/utils.py:786) and the curves it
  plots are stored; a checkpoint built like ``save_models`` (src/trainer_unified_nuscenes.py:476-490) with real
  ``nn.Embedding.state_dict()``s is read by ``supnerf_amd.io.load_checkpoint``; one written by ``io.save_checkpoint`` is loaded back
  into the reference's modules the way ``resume_from_epoch`` / ``load_model`` do (:492-513, src/optimizer_nuscenes.py:1790-1808).

Every reference output is also checked against the CPU oracle's restatement.  Fixtures are numbers only.
Usage:  python tests/golden/gen_golden_r2.py
"""
import io as _io
import json
import os
import sys
import tempfile
from contextlib import redirect_stdout

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import gen_golden as G          # noqa: E402  (import_reference, RandTap, assert_same, save, make_model)
from oracle import supnerf_oracle as O  # noqa: E402

REF = G.REF


def kitti_fixture(RU, model, params):
    import supnerf_amd as A
    syn = A.synthetic
    out = {}
    # ---- pose convention
    objs = [syn.synthetic_kitti_object(i) for i in range(6)]
    poses = torch.stack([o["obj_pose"] for o in objs])
    h = torch.tensor([float(o["wlh"][2]) for o in objs])
    src = poses.clone()
    ref = RU.obj_pose_kitti2nusc(src, h)
    G.assert_same("kitti2nusc", O.obj_pose_kitti2nusc(poses, h), ref)
    # (the inverse, obj_pose_nuse2kitti :1369-1381, broadcasts a (B,) height against a (B,2) slice and raises for B != 2; no caller)
    out.update(k2n_in=poses, k2n_h=h, k2n_out=ref, k2n_in_after=src)
    # ---- roi_process: (box, H, W, margin, sq_pad)
    cases = []
    boxes = [o["box2d"] for o in objs] + [torch.tensor(b, dtype=torch.int32) for b in
                                          ([3, 100, 250, 260], [1100, 150, 1241, 300], [500, 2, 600, 80], [400, 300, 520, 374], [10, 10, 11, 300],
                                           [601, 171, 640, 190], [0, 0, 1241, 374])]
    for b in boxes:
        for margin, H, W in ((15, 375, 1242), (5, 900, 1600), (0, None, None)):
            for sq in (True, False):
                r = RU.roi_process(b, H, W, margin, sq_pad=sq)
                G.assert_same(f"roi_process {b.tolist()} {margin} {sq}", O.roi_process(b, H, W, margin, sq), r)
                cases.append((b, -1 if H is None else H, -1 if W is None else W, margin, int(sq), r))
    fb = torch.tensor([100.5, 50.25, 180.0, 90.75])
    rf = RU.roi_process(fb, 375, 1242, 15, sq_pad=True)
    out.update(roi_in=torch.stack([c[0] for c in cases]), roi_H=np.array([c[1] for c in cases]), roi_W=np.array([c[2] for c in cases]),
               roi_margin=np.array([c[3] for c in cases]), roi_sq=np.array([c[4] for c in cases]), roi_out=torch.stack([c[5] for c in cases]),
               roi_float_in=fb, roi_float_out=rf)
    # ---- sample_from_rays_v2
    g = torch.Generator().manual_seed(31)
    rays = torch.cat([torch.randn(9, 3, generator=g), torch.randn(9, 3, generator=g), torch.rand(9, 1, generator=g) * 3 + 2,
                      torch.rand(9, 1, generator=g) * 3 + 6], -1)
    torch.manual_seed(32)
    with G.RandTap() as tap:
        z = RU.sample_from_rays_v2(rays, 16)
    G.assert_same("sample_from_rays_v2", O.unit_interval_samples(rays[:, 6:7], rays[:, 7:8], 16, tap.draws[0]), z)
    out.update(sfr2_rays=rays, sfr2_jitter=tap.draws[0], sfr2_z=z)
    # ---- pose errors
    est, tgt = ref[:4].clone(), ref[1:5].clone()
    eR, eT = RU.calc_pose_err(est, tgt)
    out.update(perr_est=est, perr_tgt=tgt, perr_R=eR, perr_T=eT)
    # ---- one KITTI object end to end through the reference's render_rays_v2: crop of roi size -> resize to im_sz
    hp = A.driver.load_hpams(dataset="kitti")
    ob = A.driver.make_kitti_objects([3], hp)[0]              # truncated car: non-square crop
    assert ob["img"].shape[0] != ob["img"].shape[1]
    gg = torch.Generator().manual_seed(33)
    sc, tc = torch.randn(1, 256, generator=gg) * 0.3, torch.randn(1, 256, generator=gg) * 0.3
    torch.manual_seed(34)
    with G.RandTap() as tap, torch.no_grad():
        r = RU.render_rays_v2(model, "cpu", ob["img"], ob["mask"], ob["cam_pose"], ob["obj_diag"], ob["K"], ob["roi"], 64, sc, tc, 1, 0, im_sz=16)
    with torch.no_grad():
        o = O.render_rays_v2(params, ob["img"], ob["mask"], ob["cam_pose"], ob["obj_diag"], ob["K"], ob["roi"], 64, sc, tc, True, im_sz=16,
                             jitter=tap.draws[0])
    for a, b, n in zip(o, r, ("rgb", "depth", "acc", "tgt", "occ")):
        G.assert_same("kitti render_rays_v2 " + n, a, b)
    assert 0 < int((r[4] > 0).sum()) < r[4].numel() and int((r[4] == 0).sum()) > 0
    out.update(e2e_index=np.array(3), e2e_roi=ob["roi"], e2e_cam_pose=ob["cam_pose"], e2e_obj_diag=ob["obj_diag"], e2e_shapecode=sc,
               e2e_texturecode=tc, e2e_jitter=tap.draws[0], e2e_rgb=r[0], e2e_depth=r[1], e2e_acc=r[2], e2e_tgt=r[3], e2e_occ=r[4])
    G.save("kitti", **out)
    # ---- the config values the loop reads (data of the reference's json files)
    keep = ("n_samples", "render_im_sz", "roi_margin", "shapenet_obj_cood", "sym_aug", "loss_occ_coef", "optimize")
    cfg = {}
    for tag, f in (("kitti", "supnerf.kitti.car.json"), ("nusc", "supnerf.nusc.vehicle.car.json")):
        j = json.load(open(os.path.join(REF, "jsonfiles", f)))
        cfg[tag] = {k: j[k] for k in keep}
        cfg[tag]["net_hyperparams"] = {k: j["net_hyperparams"][k] for k in ("shape_blocks", "texture_blocks", "latent_dim", "num_xyz_freq", "num_dir_freq")}
        cfg[tag]["dataset"] = {k: v for k, v in j["dataset"].items() if k in ("name", "img_h", "img_w", "mask_pixels", "max_dist", "min_depth")}
    with open(os.path.join(HERE, "kitti.json"), "w") as f:
        json.dump(cfg, f, indent=1, sort_keys=True)
    print("  wrote kitti.json")


def formats_fixture(RU, codenerf, model, params):
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    import supnerf_amd as A
    out = {}
    tmp = tempfile.mkdtemp(prefix="snr_formats_")
    # ---- codes+poses.pth written by the product, read by the reference's reader
    g = torch.Generator().manual_seed(41)
    n_obj, n_it = 5, 12
    rows = torch.rand(n_obj, n_it, 4, generator=g)
    rows[:, :, 0] = rows[:, :, 0] * 20 - 2          # some negative PSNRs: the reader zeroes them
    rows[1, 3, 0] = float("inf")                    # and infinities -- by ROW: it indexes with argwhere's (row, col) pairs, so objects 1 AND 3
    rows[2, 4, 2] = float("nan")                    # are zeroed whole (:820); likewise a NaN rotation error zeroes objects 2 and 4 (:866)
    ids = [7, 8, 9, 12, 40]
    path = A.io.save_driver_results(os.path.join(tmp, "res"), rows.reshape(n_obj, -1), ids, n_lidar=64)
    fig, axes = plt.subplots(2, 2)
    buf = _io.StringIO()
    with redirect_stdout(buf):
        lines = RU.collect_eval_results(path, n_it, axes, "m", True, None, print_iters=[0, 3, 5, 10], rot_outlier_ignore=False)
    curves = [np.asarray(l.get_ydata(), dtype=np.float64) for l in lines]
    assert len(curves) == 4
    saved = torch.load(path, map_location="cpu")           # default weights_only load must work too (plain containers + tensors)
    mine = O.eval_curves(saved, n_it)
    for a, b, n in zip(mine, curves, ("psnr", "depth", "R", "T")):
        G.assert_same("collect_eval_results " + n, a, b, tol=1e-12)
    keys = sorted(saved["psnr_eval"].keys())
    assert keys == sorted(f"{i}_0" for i in ids)
    out.update(eval_rows=rows, eval_ids=np.array(ids), eval_psnr=curves[0], eval_depth=curves[1], eval_R_deg=curves[2], eval_T=curves[3])
    plt.close(fig)
    # ---- a checkpoint as save_models writes it (real nn.Embedding state dicts) -> the product's loader
    n_inst = 9
    torch.manual_seed(43)
    shape_codes, texture_codes = torch.nn.Embedding(n_inst, 256), torch.nn.Embedding(n_inst, 256)
    optimized_idx = torch.tensor([1., 1., 0., 1., 0., 0., 1., 1., 0.])
    save_dict = {"model_params": model.state_dict(), "shape_code_params": shape_codes.state_dict(), "texture_code_params": texture_codes.state_dict(),
                 "niter": 1234, "nepoch": 5, "instoken2idx": {f"tok{i}": i for i in range(n_inst)}, "optimized_idx": optimized_idx}
    ck = os.path.join(tmp, "models.pth")
    torch.save(save_dict, ck)
    # what load_model derives from it (src/optimizer_nuscenes.py:1799-1808)
    sd = torch.load(ck, map_location=torch.device("cpu"))
    oi = sd["optimized_idx"].numpy()
    mean_shape = torch.mean(sd["shape_code_params"]["weight"][oi > 0], dim=0).reshape(1, -1)
    mean_texture = torch.mean(sd["texture_code_params"]["weight"][oi > 0], dim=0).reshape(1, -1)
    mine_model = A.CodeNeRF(shape_blocks=3, texture_blocks=1)
    ms, mt, saved2, missing = A.io.load_checkpoint(ck, mine_model, strict=True)
    assert torch.equal(ms, mean_shape) and torch.equal(mt, mean_texture)
    for k, v in model.state_dict().items():
        assert torch.equal(mine_model.state_dict()[k], v), k
    out.update(ck_seed=np.array(43), ck_optimized_idx=optimized_idx, ck_mean_shape=mean_shape, ck_mean_texture=mean_texture,
               ck_shape_row0=sd["shape_code_params"]["weight"][0], ck_niter=np.array(1234), ck_nepoch=np.array(5))
    # ---- a checkpoint written by the product -> the reference's modules, the way resume_from_epoch / load_model read it
    mine_model2 = A.CodeNeRF(shape_blocks=3, texture_blocks=1)
    mine_model2.load_state_dict(params)
    p2 = os.path.join(tmp, "ours", "models.pth")
    A.io.save_checkpoint(p2, mine_model2, shape_codes.weight, texture_codes.weight, niter=77, nepoch=3,
                         instoken2idx={f"tok{i}": i for i in range(n_inst)}, optimized_idx=optimized_idx)
    sd2 = torch.load(p2, map_location=torch.device("cpu"))          # reference call, default arguments
    ref_model = codenerf.CodeNeRF(shape_blocks=3, texture_blocks=1, W=256, latent_dim=256)
    ref_model.load_state_dict(sd2["model_params"])                  # strict, like load_model :1796
    e1, e2 = torch.nn.Embedding(n_inst, 256), torch.nn.Embedding(n_inst, 256)
    e1.load_state_dict(sd2["shape_code_params"]); e2.load_state_dict(sd2["texture_code_params"])      # resume_from_epoch :510-511
    assert torch.equal(e1.weight, shape_codes.weight) and sd2["niter"] + 1 == 78 and sd2["nepoch"] + 1 == 4
    oi2 = sd2["optimized_idx"].numpy()
    assert torch.equal(torch.mean(sd2["shape_code_params"]["weight"][oi2 > 0], dim=0).reshape(1, -1), mean_shape)
    g2 = torch.Generator().manual_seed(44)
    xyz, vd = torch.rand(4, 8, 3, generator=g2) - 0.5, torch.nn.functional.normalize(torch.randn(4, 8, 3, generator=g2), dim=-1)
    with torch.no_grad():
        s_ref, c_ref = ref_model(xyz, vd, mean_shape, mean_texture)
        s_or, c_or = O.decoder_forward(params, xyz, vd, mean_shape, mean_texture)
    G.assert_same("decoder from our checkpoint sigma", s_or, s_ref); G.assert_same("decoder from our checkpoint rgb", c_or, c_ref)
    out.update(ck_probe_xyz=xyz, ck_probe_viewdir=vd, ck_probe_sigma=s_ref, ck_probe_rgb=c_ref)
    G.save("formats", **out)


def main():
    torch.manual_seed(0); np.random.seed(0)
    codenerf, RU, RR = G.import_reference()
    model, params = G.make_model(codenerf)
    kitti_fixture(RU, model, params)
    formats_fixture(RU, codenerf, model, params)
    print("round-2 reference-vs-oracle checks passed; fixtures written to", HERE)


if __name__ == "__main__":
    main()
