"""Round-3 fixtures taken from the reference (run from the repository root in the build container, where /root/reference exists):

* ``waymo.json`` -- the values of jsonfiles/supnerf.waymo.car.json that the optimise loop reads (``driver.load_hpams(dataset="waymo")`` must
  agree with them; optimize_waymo.py runs the KITTI-convention loop with these).
Data only: configuration values, no reference source text."""
import json
import os

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
LOOP_KEYS = ("n_samples", "render_im_sz", "roi_margin", "shapenet_obj_cood", "sym_aug", "loss_occ_coef", "optimize")


def main():
    with open(os.path.join(REF, "jsonfiles", "supnerf.waymo.car.json")) as f:
        hp = json.load(f)
    out = {k: hp[k] for k in LOOP_KEYS}
    out["dataset"] = {k: hp["dataset"][k] for k in ("name", "mask_pixels", "max_dist", "min_depth", "min_lidar_cnt")}
    out["net_hyperparams"] = {k: hp["net_hyperparams"][k] for k in ("shape_blocks", "texture_blocks", "latent_dim", "num_xyz_freq", "num_dir_freq")}
    with open(os.path.join(HERE, "waymo.json"), "w") as f:
        json.dump({"waymo": out}, f, indent=1, sort_keys=True)
    print("wrote waymo.json")


if __name__ == "__main__":
    main()
