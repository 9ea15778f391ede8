"""CPU tests of the host side: the C-ABI library loads and exports what the header declares, the host geometry
matches the oracle, the product path refuses CPU tensors (no fallback), sharding / metric gather logic incl. a
world_size-2 gloo run."""
import inspect
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import supnerf_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def amd():
    import supnerf_amd
    return supnerf_amd


def test_library_exports_every_declared_symbol(amd):
    hdr = open(os.path.join(ROOT, "include", "supnerf_hip.h")).read()
    declared = set(re.findall(r"\b(snr_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(amd._lib.exported_symbols()), declared ^ set(amd._lib.exported_symbols())
    lib = amd._lib.lib()                      # loads without a GPU; no compute call is made here
    for s in declared:
        assert hasattr(lib, s)
    assert lib.snr_abi_version() == amd._lib.header_abi_version() == int(re.search(r"SNR_ABI_VERSION (\d+)", hdr).group(1))
    assert lib.snr_packed_bytes(3, 1) == 3615264 + (2 * 32768 + 5 * 8 * 32768 + 8 * 36864 + 4 * 32768) * 2 and lib.snr_packed_bytes(9, 1) == 0
    assert lib.snr_precision_supported(1, 3, 1, 4096 * 64) == 1 and lib.snr_precision_supported(1, 5, 5, 4096 * 64) == 0
    assert lib.snr_precision_supported(1, 3, 1, 35) == 0 and lib.snr_precision_supported(0, 5, 5, 35) == 1
    assert lib.snr_mask_bytes(262144, 3, 1) == 262144 // 32 * 7 * 1024


def test_graft_entry_build(amd):
    """__graft_entry__.build() is what the driver runs on the CPU box: it must compile (or find up to date) the library, load it,
    resolve every symbol and agree with the header on the ABI version."""
    import __graft_entry__ as G
    G.build()


def test_sample_from_rays_v2_is_the_renderer_method(amd, golden):
    """utils.sample_from_rays_v2 (src/utils.py:170-184, imported by scripts/demo.py:14) against the reference's numbers."""
    g = golden("twins")
    amd.utils.JITTER_OVERRIDE = g["sfr_jitter"]
    try:
        z = amd.utils.sample_from_rays_v2(g["sfr_rays"], 32)
    finally:
        amd.utils.JITTER_OVERRIDE = None
    assert z.shape == g["sfr_z"].shape and float((z - g["sfr_z"]).abs().max()) < 1e-6
    assert list(inspect.signature(amd.utils.sample_from_rays_v2).parameters) == ["rays", "n_samples"]


def test_api_caches_follow_their_inputs(amd):
    """The public render functions cache the camera-frame pixel table per (K, roi, grid) and the resized targets per (crop, mask, size):
    a changed input must give changed outputs (in-place edits bump the tensor version; new tensors are new keys)."""
    U = amd.utils
    g = torch.Generator().manual_seed(0)
    img, mask = torch.rand(20, 30, 3, generator=g), (torch.randint(0, 3, (20, 30, 1), generator=g) - 1).float()
    a = U._resize_to(img, mask, 8, "cpu")
    b = U._resize_to(img, mask, 8, "cpu")
    assert a[0] is b[0] and a[1] is b[1]                                   # served from the cache
    ref_t, ref_o = U._resize(img, mask, 8)
    assert torch.equal(a[0], ref_t.reshape(-1, 3)) and torch.equal(a[1], ref_o.reshape(-1, 1))
    img.mul_(0.5)                                                          # in-place edit of the caller's crop
    c = U._resize_to(img, mask, 8, "cpu")
    assert c[0] is not a[0] and torch.equal(c[0], U._resize(img, mask, 8)[0].reshape(-1, 3))
    d = U._resize_to(img.clone(), mask, 8, "cpu")                          # another tensor with the same numbers: its own entry
    assert d[0] is not c[0] and torch.equal(d[0], c[0])
    assert U._resize_to(img, mask, 4, "cpu")[0].shape == (16, 3)           # another size
    # a caller that edits the RETURNED targets in place (reference-style ``occ_pixels[occ_pixels < 0] = 0``) must not poison later calls
    e = U._resize_to(img, mask, 8, "cpu")
    want_occ = e[1].clone()
    e[1][e[1] < 0] = 0
    f = U._resize_to(img, mask, 8, "cpu")
    assert f[1] is not e[1] and torch.equal(f[1], want_occ)
    U.clear_caches()
    assert U._resize_to(img, mask, 8, "cpu")[0] is not f[0]
    K = torch.tensor([[1000., 0., 500.], [0., 1000., 300.], [0., 0., 1.]])
    pose = torch.cat([torch.eye(3), torch.tensor([[0.1], [0.2], [5.0]])], 1)
    o1, d1 = U.get_rays(K, pose, [100, 50, 164, 114], uv_steps=[8, 8])
    o2, d2 = U.get_rays(K, pose, [100, 50, 164, 114], uv_steps=[8, 8])
    assert torch.equal(d1, d2)
    K2 = K.clone(); K2[0, 2] = 510.
    assert not torch.equal(U.get_rays(K2, pose, [100, 50, 164, 114], uv_steps=[8, 8])[1], d1)      # other intrinsics: other table
    assert not torch.equal(U.get_rays(K, pose, [101, 50, 165, 114], uv_steps=[8, 8])[1], d1)       # other roi
    want = O.pixel_rays(K, pose, torch.tensor([100, 50, 164, 114]), uv_steps=[8, 8])
    assert torch.equal(o1, want[0]) and torch.equal(d1, want[1])


def test_product_path_has_no_cpu_fallback(amd):
    p = O.init_decoder_params()
    with pytest.raises(amd.SnrError):
        amd.ops.pack_weights(p, 3, 1)                      # CPU tensors
    with pytest.raises(amd.SnrError):
        amd.ops.composite_fwd(torch.rand(4, 8), torch.rand(4, 8, 3), torch.rand(8), amd.ops.Z_SHARED, False)
    m = amd.CodeNeRF(3, 1)
    with pytest.raises(amd.SnrError):
        m(torch.rand(2, 4, 3), torch.rand(2, 4, 3), torch.rand(1, 256), torch.rand(1, 256))
    with pytest.raises(amd.SnrError):
        amd.CodeNeRF(3, 1, W=128)                          # unsupported width fails loudly


def test_state_dict_names_match_reference(amd):
    m = amd.CodeNeRF(shape_blocks=3, texture_blocks=1)
    assert list(m.state_dict().keys()) == list(O.decoder_param_names(3, 1))
    m.load_state_dict(O.init_decoder_params(), strict=True)
    s = amd.SUPNeRF(shape_blocks=3, texture_blocks=1, pose_blocks=3, regress_blocks=3, img_encoder=False)
    keys = list(s.state_dict().keys())
    assert keys[:28] == list(O.decoder_param_names(3, 1)) and "out_delta_layer.weight" in keys and "pose_layer_0.0.weight" in keys
    # pose head is stock torch
    assert s.pose_update(torch.rand(2, 256), torch.rand(2, 16)).shape == (2, 6)


def test_synthetic_matches_oracle_copy(amd):
    a, b = amd.synthetic.init_decoder_params(), O.init_decoder_params()
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
    for i in (0, 3, 17):
        x, y = amd.synthetic.synthetic_object(i), O.synthetic_object(i)
        assert torch.equal(x["cam_pose"], y["cam_pose"]) and torch.equal(x["roi"], y["roi"]) and x["obj_diag"] == y["obj_diag"]
        assert all(torch.equal(p, q) for p, q in zip(amd.synthetic.synthetic_targets(i, 16), O.synthetic_targets(i, 16)))


def test_rays_and_depths_match_oracle(amd, golden):
    g = golden("rays")
    U = amd.utils
    o, d = U.get_rays(g["K"], g["cam_pose"], g["roi"], uv_steps=[8, 8])
    assert torch.equal(o, g["rays_o"]) and torch.equal(d, g["viewdir"])
    o, d = U.get_rays(g["K"], g["cam_pose"], g["roi_small"])
    assert torch.equal(d, g["viewdir_small"])
    o, d = U.get_rays_specified(g["K"], g["cam_pose"], g["x_vec"].numpy() + int(g["roi"][0]), g["y_vec"].numpy() + int(g["roi"][1]))
    assert torch.equal(d, g["viewdir_spec"])
    jit = torch.rand(64, generator=torch.Generator().manual_seed(1))
    near, far = U._sphere_bounds(g["cam_pose"], np.float32(5.3))
    assert torch.equal(U._shared_depths(near, far, 64, "cpu", jitter=jit), O.shared_depth_samples(*O.sphere_bounds(g["cam_pose"], np.float32(5.3)), 64, jit))
    # tensor end points (device-side bounds, no host sync) agree with torch.linspace to fp32 round-off
    a = U._linspace(torch.tensor(9.25), torch.tensor(14.5), 64, "cpu")
    assert float((a - torch.linspace(9.25, 14.5, 64)).abs().max()) < 2e-6


def test_precision_names_pairs_and_triples(amd):
    """``precision``: one name for every launch, a pair (forward, backward) or a training triple (forward chain, backward chain,
    products); 'auto' falls back to exact fp32 where the split kernels do not take the shape, an explicit 'bf16x3' raises there."""
    ops = amd.ops
    import warnings
    ok = dict(shape_blocks=3, texture_blocks=1, points_per_obj=4096)
    assert ops.resolve_precision("fp32", **ok) == ops.FP32 and ops.resolve_precision("bf16x3", **ok) == ops.BF16X3
    assert ops.resolve_precision("auto", **ok) == ops.BF16X3 and ops.resolve_precision(None, **ok) == ops.BF16X3
    pair = ("fp32", "bf16x3")
    assert ops.resolve_precision(pair, **ok) == ops.FP32 and ops.resolve_precision(pair, backward=True, **ok) == ops.BF16X3
    triple = ("auto", "fp32", "bf16x3")
    assert ops.resolve_precision(triple, **ok) == ops.BF16X3 and ops.resolve_precision(triple, backward=True, **ok) == ops.FP32
    with pytest.raises(amd.SnrError):
        ops.resolve_precision(("fp32",), **ok)
    with pytest.raises(amd.SnrError):
        ops.resolve_precision("fp16", **ok)
    big = dict(shape_blocks=5, texture_blocks=5, points_per_obj=4096)          # SUPNeRF's default decoder: ten blocks
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert ops.resolve_precision("auto", **big) == ops.FP32
        assert ops.resolve_precision(("auto", "auto"), backward=True, **big) == ops.FP32
    with pytest.raises(amd.SnrError):
        ops.resolve_precision("bf16x3", **big)
    with pytest.raises(amd.SnrError):
        ops.resolve_precision("bf16x3", shape_blocks=3, texture_blocks=1, points_per_obj=35)      # a partial 32-point tile per object


def test_frame_matrices(amd):
    U = amd.utils
    g = torch.Generator().manual_seed(0)
    xyz, vd = torch.randn(5, 7, 3, generator=g), torch.randn(5, 7, 3, generator=g)
    for flip in (False, True):
        for kitti in (False, True):
            for shapenet in (False, True):
                m = torch.tensor(U._frame(flip, kitti, shapenet)).view(3, 3)
                xo, vo = O.object_frame_transforms(xyz, vd, flip, kitti, shapenet)
                assert torch.equal(xyz @ m.T, xo) and torch.equal(vd @ m.T, vo)


def test_prepare_pixel_samples_cpu_and_resize(amd, golden):
    g = golden("prepare_pixel_samples")
    amd.utils.JITTER_OVERRIDE = g["jitter"]
    try:
        np.random.seed(9)
        out = amd.utils.prepare_pixel_samples(g["img"], g["mask_occ"], g["cam_pose"], np.float32(g["obj_diag"]), g["K"], g["roi"], 20, 64,
                                              1, 0, im_sz=8)
    finally:
        amd.utils.JITTER_OVERRIDE = None
    for a, k in zip(out, ("xyz", "viewdir", "z_vals", "rgb_tgt", "occ")):
        assert float((a - g[k]).abs().max()) < 2e-6, k
    r = golden("resize_targets")
    im, mk = amd.utils._resize(r["img"], r["mask_occ"], 8)
    assert torch.equal(im.reshape(-1, 3), r["rgb_tgt"]) and torch.equal(mk.reshape(-1, 1), r["occ"])


def test_box_bounds_match_oracle(amd, golden):
    g = golden("render_b_hit")
    ro, vd = O.pixel_rays(g["K"], g["cam_pose"], g["roi"], uv_steps=[8, 8])
    wlh = g["wlh"].numpy()
    diag = np.linalg.norm(wlh).astype(np.float32)
    near, far, hit = amd.renderer._box_bounds(ro / (diag / 2), vd, wlh, diag)
    assert torch.equal(hit, g["hit"].bool())
    rend = amd.NeRFRenderer(n_samples=64)
    amd.utils.JITTER_OVERRIDE = g["jitter"]
    try:
        xyz, vdir, z_vals, hit2 = rend.prepare_sampled_rays(ro, vd, wlh)        # CPU tensors: plain torch arithmetic
    finally:
        amd.utils.JITTER_OVERRIDE = None
    assert float((z_vals - g["z_vals"]).abs().max()) < 1e-6
    zi, zo, hm = amd.utils.ray_box_intersection_tensor(ro / (diag / 2), vd, -torch.tensor([wlh[1], wlh[0], wlh[2]]) / diag,
                                                       torch.tensor([wlh[1], wlh[0], wlh[2]]) / diag)
    assert torch.equal(hm, hit) and zi.shape[0] == int(hit.sum())
    assert amd.utils.ray_box_intersection_tensor(torch.empty(0, 3), torch.empty(0, 3)) == (None, None, None)


def test_rotation_helpers(amd):
    D = amd.driver
    g = torch.Generator().manual_seed(0)
    v = torch.randn(64, 3, generator=g)
    v = v / v.norm(dim=-1, keepdim=True) * (torch.rand(64, 1, generator=g) * 3.1)
    R = D.axis_angle_to_matrix(v)
    eye = torch.eye(3).expand(64, 3, 3)
    assert float((R @ R.transpose(-1, -2) - eye).abs().max()) < 1e-5 and float((torch.linalg.det(R) - 1).abs().max()) < 1e-5
    assert float((D.matrix_to_axis_angle(R) - v).abs().max()) < 2e-4
    assert float(D.axis_angle_to_matrix(torch.zeros(3)).sub(torch.eye(3)).abs().max()) == 0
    # against scipy (independent implementation)
    from scipy.spatial.transform import Rotation
    assert float((R - torch.from_numpy(Rotation.from_rotvec(v.numpy()).as_matrix()).float()).abs().max()) < 1e-5
    assert float((D.rot_dist(R, R)).abs().max()) < 1e-3


def test_shard_slices_cover_everything(amd):
    D = amd.driver
    for n in (0, 1, 7, 64, 65):
        for w in (1, 2, 3, 8):
            got = [i for r in range(w) for i in D.shard_slice(n, w, r)]
            assert got == list(range(n))
            sizes = [len(D.shard_slice(n, w, r)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1


def test_hpams_reader(amd, tmp_path):
    h = amd.driver.load_hpams()
    assert h["n_samples"] == 64 and h["optimize"]["num_opts"] == 100 and h["net_hyperparams"]["shape_blocks"] == 3
    p = tmp_path / "c.json"
    p.write_text('{"n_samples": 32, "optimize": {"num_opts": 5}}')
    assert amd.driver.load_hpams(str(p))["n_samples"] == 32
    # the shipped defaults of every dataset hold the values of the reference's json files (fixtures written from those files by
    # tests/golden/gen_golden_r2.py / gen_golden_r3.py)
    import json, os
    for name, fixture in (("kitti", "kitti.json"), ("nusc", "kitti.json"), ("waymo", "waymo.json")):
        want = json.load(open(os.path.join(os.path.dirname(__file__), "golden", fixture)))[name]
        got = amd.driver.load_hpams(dataset=name)
        for k, v in want.items():
            if isinstance(v, dict):
                for kk, vv in v.items():
                    assert got[k][kk] == vv, (name, k, kk)
            else:
                assert got[k] == v, (name, k)
    with pytest.raises(ValueError):
        amd.driver.load_hpams(dataset="argoverse")


def test_result_file_carries_per_object_lidar_counts(amd, tmp_path):
    """codes+poses.pth: ``lidar_pts_cnt`` is the weight of every object's depth error in the reference's evaluation
    (collect_eval_results, src/utils.py:786): the writer takes the per-object counts the driver reports, and a generator of ids."""
    rows = torch.rand(3, 2 * 4)
    psnr, depth, R, T, cnt = amd.io.metric_rows_to_eval_dicts(rows, (i for i in (7, 8, 9)), n_lidar=torch.tensor([40, 53, 66]))
    assert cnt == {"7_0": 40, "8_0": 53, "9_0": 66} and len(psnr) == 3
    path = amd.io.save_driver_results(str(tmp_path / "r"), rows, (i for i in (7, 8, 9)), n_lidar=[40, 53, 66])
    saved = torch.load(path, weights_only=False)
    assert saved["num_obj"] == 3 and saved["lidar_pts_cnt"] == cnt
    with pytest.raises(ValueError):
        amd.io.metric_rows_to_eval_dicts(rows, [1, 2, 3], n_lidar=[1, 2])


_GLOO_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
import supnerf_amd
from supnerf_amd import driver as D
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
rank = dist.get_rank()
n = 5
mine = list(D.shard_slice(n, 2, rank))
rows = torch.tensor([[float(i), 10.0 * i + 1, 10.0 * i + 2] for i in mine]).reshape(len(mine), 3)
out = D.gather_metric_rows(rows, torch.tensor(mine, dtype=torch.float32), n)
exp = torch.tensor([[float(i), 10.0 * i + 1, 10.0 * i + 2] for i in range(n)])
assert torch.equal(out, exp), (rank, out)
dist.destroy_process_group()
print("ok", rank)
"""


def test_metric_gather_two_ranks_gloo(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    script = tmp_path / "w.py"
    script.write_text(_GLOO_WORKER.format(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)


_TRAIN_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
import supnerf_amd
from supnerf_amd import trainer as T
from oracle import supnerf_oracle as O
rank = int(sys.argv[1])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=rank, world_size=2)
torch.set_num_threads(2)
hp = dict(lr_schedule=[dict(lr=1e-3, interval=100), dict(lr=1e-2, interval=100)])


N_INST = {n_inst}


def world(seed_codes=3):
    m = supnerf_amd.CodeNeRF(shape_blocks=1, texture_blocks=1)          # parameters only; the CPU forward below is the oracle's
    m.load_state_dict(O.init_decoder_params(seed=0, shape_blocks=1, texture_blocks=1), strict=True)
    codes = T.CodeTables(N_INST, 256, seed=seed_codes)
    fwd = lambda xyz, vd, sc, tc: O.decoder_forward(dict(m.named_parameters()), xyz, vd, sc, tc)
    params = list(m.parameters()) + list(codes.parameters())
    return m, codes, fwd, params


g = torch.Generator().manual_seed(11)
B, n, S = 2, 6, 8
full = dict(code_idx=torch.tensor({code_idx}), xyz=torch.rand(B, n, S, 3, generator=g) - 0.5,
            viewdir=torch.nn.functional.normalize(torch.randn(B, n, 1, 3, generator=g), dim=-1).repeat(1, 1, S, 1),
            z_vals=torch.sort(torch.rand(B, S, generator=g) * 4 + 9, dim=-1)[0], rgb_tgt=torch.rand(B, n, 3, generator=g),
            occ_pixels=torch.tensor([1., -1., 0., 1., 1., -1.]).view(1, n, 1).repeat(B, 1, 1))
mine = {{k: v[rank:rank + 1] for k, v in full.items()}}

# two ranks, one object each, two iterations
m, codes, fwd, params = world()
bucket = T.GradBucket(params, row_sparse=list(codes.parameters()) if {row_sparse} else ())
opt = T.make_optimizer(m, codes, hp)
for it in range(2):
    out = T.train_step(fwd, codes, opt, bucket, mine, 0.1, composite=O.volume_rendering_batch)

# the same two iterations in one process on the whole batch (what DataParallel's loss.mean() computes)
m1, codes1, fwd1, params1 = world()
opt1 = T.make_optimizer(m1, codes1, hp)
for it in range(2):
    sc, tc = codes1(full["code_idx"])
    total = O.training_losses(dict(m1.named_parameters()), full["xyz"], full["viewdir"], sc, tc, full["z_vals"], full["rgb_tgt"],
                              full["occ_pixels"], 0.1)[0]
    opt1.zero_grad()
    total.backward()
    opt1.step()
worst = max(float((a - b).abs().max()) for a, b in zip(params, params1))
assert worst < 2e-5, worst          # AdamW normalises: rounding of near-zero gradients is amplified to ~1e-4 of a step
flat = torch.cat([p.detach().flatten() for p in params])
other = [torch.empty_like(flat) for _ in range(2)]
dist.all_gather(other, flat)
assert torch.equal(other[0], other[1])                      # replicas stay bit-identical
assert float(bucket.flat.abs().max()) == 0.0                # zeroed for the next iteration, views intact
assert all(float(p.grad.abs().max()) == 0.0 for p in bucket.rows)
if {row_sparse}:
    assert bucket.flat.numel() == sum(p.numel() for p in m.parameters())      # the tables are NOT in the all-reduced bucket
bucket.check_views()
dist.destroy_process_group()
print("ok", rank, worst)
"""


@pytest.mark.parametrize("n_inst,code_idx,row_sparse", [(4, [2, 0], False), (4, [2, 0], True), (10000, [9731, 12], True), (10000, [77, 77], True)])
def test_training_step_two_ranks_gloo(tmp_path, n_inst, code_idx, row_sparse):
    """DDP-style step: per-rank batch slice + one bucket all-reduce (+ the row exchange of the code tables) == single-process
    step on the full batch -- with the tables inside the dense bucket (small table) and with the row-sparse exchange, at 4 and at 10 000
    instances (src/trainer_unified_nuscenes.py:276-285: a step touches B rows; :414-422: AdamW still decays every row), and with both
    ranks touching the SAME row."""
    port = 31500 + (os.getpid() % 2000) + (7 if row_sparse else 0) + (13 if n_inst > 4 else 0) + code_idx[0] % 5
    script = tmp_path / "t.py"
    script.write_text(_TRAIN_WORKER.format(root=ROOT, port=port, n_inst=n_inst, code_idx=code_idx, row_sparse=row_sparse))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)


def test_grad_bucket_and_lr_schedule(amd):
    T = amd.trainer
    lin = torch.nn.Linear(4, 3)
    b = T.GradBucket(lin.parameters())
    lin(torch.ones(2, 4)).sum().backward()
    assert b.flat.numel() == 15 and torch.equal(b.flat[:12].view(3, 4), torch.full((3, 4), 2.0)) and torch.equal(b.flat[12:], torch.full((3,), 2.0))
    b.allreduce_mean()                                           # no process group: a no-op
    lin.zero_grad(set_to_none=True)
    with pytest.raises(RuntimeError):
        b.check_views()
    hp = dict(lr_schedule=[dict(lr=1e-4, interval=10), dict(lr=1e-3, interval=4)])
    assert T.learning_rates(hp, 0) == (1e-4, 1e-3) and T.learning_rates(hp, 25) == (1e-4 / 4, 1e-3 / 64)
    c = T.CodeTables(3, 256, mean_shape=torch.ones(1, 256), mean_texture=torch.zeros(1, 256))
    assert float(c.shape_codes.weight.min()) == 1.0 and float(c.texture_codes.weight.abs().max()) == 0.0


def test_scene_ray_table_matches_oracle(amd, golden):
    """Host side of vis_scene (scripts/demo.py:437-523): per-object rois, rays, box entry/exit depths, valid-pixel mask."""
    g = golden("scene")
    H, W = int(g["H"]), int(g["W"])
    tab, valid, diags = amd.scene.scene_rays(g["obj_poses"], g["obj_wlh"], g["K"], H, W)
    o_tab, o_valid, o_diags = O.scene_rays(g["obj_poses"], g["obj_wlh"], g["K"], H, W)
    assert torch.equal(valid, g["valid"].bool()) and torch.equal(tab, o_tab) and torch.equal(diags, o_diags)
    tab2, valid2, _ = amd.scene.scene_rays(g["obj_poses"], g["obj_wlh"], g["K"], H, W, manipulation=(0.5, 0.0, 2.0), rend_aabb=False)
    o2 = O.scene_rays(g["obj_poses"], g["obj_wlh"], g["K"], H, W, manipulation=(0.5, 0.0, 2.0), rend_aabb=False)
    assert torch.equal(tab2, o2[0]) and torch.equal(valid2, o2[1])
    with pytest.raises(amd.SnrError):
        amd.scene.vis_scene(None, "cpu", g["obj_poses"], g["obj_wlh"][:2], g["shapecodes"], g["texturecodes"], g["K"], H, W, 8)


def test_small_utilities_match_reference(amd, golden):
    """get_rays_srn (src/utils.py:94-104), the numpy slab test ray_box_intersection (:236-280): host functions, reference outputs."""
    g = golden("twins")
    so, sd = amd.utils.get_rays_srn(6, 5, 40.0, g["srn_c2w"])
    assert torch.equal(so, g["srn_o"]) and torch.equal(sd, g["srn_d"])
    z_in, z_out, hit = amd.utils.ray_box_intersection(g["box_o"].numpy(), g["box_d"].numpy(), aabb_min=-g["box_max"].numpy(), aabb_max=g["box_max"].numpy())
    assert np.array_equal(hit, g["box_hit"].numpy().astype(bool)) and np.array_equal(z_in, g["box_z_in"].numpy()) and np.array_equal(z_out, g["box_z_out"].numpy())
    # CPU twin of NeRFRenderer.prepare_pixel_samples (the datasets call it in CPU workers): plain torch, same random streams
    rend = amd.NeRFRenderer(n_samples=32, white_bkgd=True)
    np.random.seed(77)
    amd.utils.JITTER_OVERRIDE = g["pps_jitter"]
    try:
        out = rend.prepare_pixel_samples(g["img"], g["mask_occ"], g["cam_pose"], g["wlh"].numpy(), g["K"], g["roi"], 40, im_sz=8)
    finally:
        amd.utils.JITTER_OVERRIDE = None
    for a, k in zip(out, ("pps_xyz", "pps_viewdir", "pps_z", "pps_tgt", "pps_occ")):
        assert float((a - g[k]).abs().max()) < 1e-6, k


def test_m0_is_written_only_by_the_weight_ring(tmp_path):
    """The bf16x3 kernels issue their LDS-DMA through inline asm that sets M0 itself and does not restore it (csrc/snr_bf16.hip,
    ring_piece).  That is only sound while the compiler keeps no value of its own in M0: every mention of m0 in the generated code
    must be the ring's own `s_mov_b32 m0, sN`."""
    import re, shutil, subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sup-nerf_amd", "csrc", "snr_bf16.hip")
    out = tmp_path / "snr_bf16.s"
    subprocess.run([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "--offload-device-only", "-S", src, "-o", str(out)],
                   check=True, capture_output=True)
    lines = [l.strip() for l in open(out) if re.search(r"\bm0\b", l.split(";")[0])]
    assert lines, "the ring's DMA instructions are gone?"
    bad = [l for l in lines if not re.fullmatch(r"s_mov_b32 m0, s\d+", l.split(";")[0].strip())]
    assert not bad, bad[:5]
    n_dma = sum(1 for l in open(out) if "global_load_lds_dwordx4" in l)
    assert n_dma == len(lines)


def test_reference_file_formats_round_trip(amd, tmp_path):
    """models.pth / codes+poses.pth with the reference's keys (src/trainer_unified_nuscenes.py:476-490, src/optimizer_nuscenes.py:1463-1476)."""
    m = amd.SUPNeRF(shape_blocks=3, texture_blocks=1, img_encoder=False)
    g = torch.Generator().manual_seed(0)
    sc, tc = torch.randn(5, 256, generator=g), torch.randn(5, 256, generator=g)
    opt = torch.tensor([1., 0., 1., 1., 0.])
    p = str(tmp_path / "ck" / "models.pth")
    amd.io.save_checkpoint(p, m, sc, tc, niter=7, nepoch=2, optimized_idx=opt)
    saved = torch.load(p, weights_only=False)
    assert set(saved) == {"model_params", "shape_code_params", "texture_code_params", "niter", "nepoch", "instoken2idx", "optimized_idx"}
    m2 = amd.CodeNeRF(shape_blocks=3, texture_blocks=1)          # decoder-only module: extra pose-head keys are skipped
    ms, mt, _, missing = amd.io.load_checkpoint(p, m2)
    assert torch.equal(ms, sc[[0, 2, 3]].mean(0, keepdim=True)) and torch.equal(mt, tc[[0, 2, 3]].mean(0, keepdim=True))
    assert all(torch.equal(a, b) for a, b in zip(m2.state_dict().values(), list(m.state_dict().values())[:28]))
    rows = torch.arange(2 * 3 * 4, dtype=torch.float32).view(2, 12)
    ps, de, R, T, cnt = amd.io.metric_rows_to_eval_dicts(rows, [10, 11])
    out = amd.io.save_opts_w_pose(str(tmp_path / "res"), 2, {}, {}, {}, ps, de, R, T, lidar_pts_cnt=cnt)
    d = torch.load(out)                  # plain containers + tensors: loads under the default (weights_only) reader, like the reference's call
    assert d["psnr_eval"]["10_0"] == [0.0, 4.0, 8.0] and [float(t) for t in d["T_eval"]["11_0"]] == [15.0, 19.0, 23.0]
    assert all(torch.is_tensor(t) and t.dim() == 0 for t in d["R_eval"]["10_0"]) and d["lidar_pts_cnt"]["11_0"] == 64
    assert set(d) >= {"num_obj", "optimized_shapecodes", "optimized_texturecodes", "optimized_poses", "psnr_eval", "depth_err_mean", "R_eval", "T_eval"}


def test_file_formats_against_the_reference_reader(amd, golden, tmp_path):
    """tests/golden/formats.npz holds what the REFERENCE's code made of our files (gen_golden_r2.py): the curves its
    collect_eval_results (src/utils.py:786) plotted from a codes+poses.pth written by io.save_driver_results, and the mean codes its
    load_model formulas (src/optimizer_nuscenes.py:1799-1808) derive from a checkpoint built like save_models
    (src/trainer_unified_nuscenes.py:476-490) with real nn.Embedding state dicts."""
    g = golden("formats")
    rows, ids = g["eval_rows"], g["eval_ids"].tolist()
    path = amd.io.save_driver_results(str(tmp_path / "res"), rows.reshape(rows.shape[0], -1), ids, n_lidar=64)
    saved = torch.load(path, map_location=torch.device("cpu"))
    assert sorted(saved["psnr_eval"]) == sorted(f"{i}_0" for i in ids)
    for a, k in zip(O.eval_curves(saved, rows.shape[1]), ("eval_psnr", "eval_depth", "eval_R_deg", "eval_T")):
        assert np.allclose(a, g[k].numpy(), rtol=0, atol=1e-12), k
    # checkpoint in the reference's own shape -> our loader
    n_inst = g["ck_optimized_idx"].shape[0]
    torch.manual_seed(int(g["ck_seed"]))
    shape_codes, texture_codes = torch.nn.Embedding(n_inst, 256), torch.nn.Embedding(n_inst, 256)
    assert torch.equal(shape_codes.weight[0], g["ck_shape_row0"])
    ref_like = amd.CodeNeRF(3, 1); ref_like.load_state_dict(O.init_decoder_params())
    ck = str(tmp_path / "models.pth")
    torch.save({"model_params": ref_like.state_dict(), "shape_code_params": shape_codes.state_dict(), "texture_code_params": texture_codes.state_dict(),
                "niter": int(g["ck_niter"]), "nepoch": int(g["ck_nepoch"]), "instoken2idx": {f"tok{i}": i for i in range(n_inst)},
                "optimized_idx": g["ck_optimized_idx"]}, ck)
    m = amd.CodeNeRF(3, 1)
    ms, mt, saved, _ = amd.io.load_checkpoint(ck, m, strict=True)
    assert torch.equal(ms, g["ck_mean_shape"]) and torch.equal(mt, g["ck_mean_texture"]) and saved["niter"] == 1234
    with torch.no_grad():      # the decoder those weights + mean codes define (the reference's own forward on our checkpoint)
        s_or, c_or = O.decoder_forward(dict(m.state_dict()), g["ck_probe_xyz"], g["ck_probe_viewdir"], ms, mt)
    assert torch.equal(s_or, g["ck_probe_sigma"]) and torch.equal(c_or, g["ck_probe_rgb"])


def test_kitti_host_geometry(amd, golden):
    """utils.obj_pose_kitti2nusc / roi_process / sample_from_rays_v2 / calc_pose_err against the reference's numbers, the KITTI object
    generator against the fixture, and the shipped hyper-parameters against the values of the reference's json files."""
    import json
    g = golden("kitti")
    src = g["k2n_in"].clone()
    assert torch.equal(amd.utils.obj_pose_kitti2nusc(src, g["k2n_h"]), g["k2n_out"])
    assert torch.equal(src, g["k2n_in_after"])                      # same in-place write as the reference
    for b, H, W, mg, sq, want in zip(g["roi_in"], g["roi_H"], g["roi_W"], g["roi_margin"], g["roi_sq"], g["roi_out"]):
        got = amd.utils.roi_process(b, None if H < 0 else int(H), None if W < 0 else int(W), int(mg), sq_pad=bool(sq))
        assert torch.equal(got, want) and got.dtype == want.dtype
    assert torch.equal(amd.utils.roi_process(g["roi_float_in"], 375, 1242, 15, sq_pad=True), g["roi_float_out"])
    amd.utils.JITTER_OVERRIDE = g["sfr2_jitter"]
    try:
        assert torch.equal(amd.utils.sample_from_rays_v2(g["sfr2_rays"], 16), g["sfr2_z"])
    finally:
        amd.utils.JITTER_OVERRIDE = None
    eR, eT = amd.utils.calc_pose_err(g["perr_est"], g["perr_tgt"])
    assert float((eR - g["perr_R"]).abs().max()) < 1e-6 and float((eT - g["perr_T"]).abs().max()) < 1e-6
    ob = amd.driver.make_kitti_objects([int(g["e2e_index"])], amd.driver.load_hpams(dataset="kitti"))[0]
    assert torch.equal(ob["roi"], g["e2e_roi"]) and torch.equal(ob["cam_pose"], g["e2e_cam_pose"])
    ref_cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "kitti.json")))
    for tag in ("kitti", "nusc"):
        hp = amd.driver.load_hpams(dataset=tag)
        for k, v in ref_cfg[tag].items():
            if isinstance(v, dict):
                assert all(hp[k][kk] == vv for kk, vv in v.items() if kk in hp[k]), (tag, k)
                assert k != "optimize" or set(v) == set(hp[k])
            else:
                assert hp[k] == v, (tag, k, hp[k], v)


def test_bench_self_launch_parent_stays_off_the_gpu(monkeypatch, capsys):
    """`python bench.py --gpus N` without WORLD_SIZE (how the driver calls it): the parent must start the ranks as CHILD processes with the
    contract's torch.distributed.run command, relay exactly rank 0's JSON line, pass the exit code on -- and never initialise the GPU
    itself (an exec / fork after HIP initialisation takes the box down)."""
    import importlib.util
    import subprocess as sp
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_run(cmd, **kw):
        seen["cmd"], seen["kw"] = cmd, kw
        return sp.CompletedProcess(cmd, 0, stdout='RCCL banner\n{"metric": "m", "value": 1.0, "n_gpus": 4}\n')
    monkeypatch.setattr(sp, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    out = capsys.readouterr().out.strip().splitlines()
    assert out == ['{"metric": "m", "value": 1.0, "n_gpus": 4}']                       # one line, the JSON, nothing else
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1].isdigit()
    assert cmd[-7:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["kw"]["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert not torch.cuda.is_initialized()                                                # the parent never touched the GPU
    # a failing child: its code comes back, nothing is invented on stdout
    monkeypatch.setattr(sp, "run", lambda cmd, **kw: sp.CompletedProcess(cmd, 3, stdout=""))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 3 and capsys.readouterr().out == ""
