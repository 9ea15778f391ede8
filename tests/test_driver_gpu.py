"""The optimise loop (supnerf_amd.driver) on the GPU against the same loop run with the CPU oracle renderer:
per-iteration PSNR / pose-error traces over the first iterations (same seeds, same jitter stream)."""
import numpy as np
import pytest
import torch

from oracle import supnerf_oracle as O

pytestmark = pytest.mark.gpu


def oracle_loop(params, obj, hpams, sc0, tc0, seed, reg_iters, pose_noise, D):
    """The reference iteration (src/optimizer_nuscenes.py:674-783) with the oracle renderer, CPU."""
    opt = hpams["optimize"]
    rs = np.random.RandomState(seed)
    R_gt = obj["cam_pose"][:, :3].T
    t_gt = -R_gt @ obj["cam_pose"][:, 3:]
    rot_vec = (D.matrix_to_axis_angle(R_gt[None]) + torch.from_numpy(rs.randn(1, 3).astype(np.float32)) * pose_noise[0]).requires_grad_()
    trans_vec = (t_gt.T + torch.from_numpy(rs.randn(1, 3).astype(np.float32)) * pose_noise[1]).requires_grad_()
    sc, tc = sc0.clone().requires_grad_(), tc0.clone().requires_grad_()
    optim = D.make_optimizer(sc, tc, rot_vec, trans_vec, {k: opt[k] for k in ("lr_shape", "lr_texture", "lr_pose")})
    ys, xs = np.where(obj["mask"][:, :, 0].numpy() > 0)
    pick = rs.permutation(len(ys))[:64]
    y_vec, x_vec = ys[pick], xs[pick]
    psnr, rot_err = [], []
    for it in range(opt["num_opts"]):
        optim.zero_grad()
        R = D.axis_angle_to_matrix(rot_vec[0]); t = trans_vec[0].unsqueeze(-1)
        Rc = R.transpose(-2, -1)
        cam2opt = torch.cat([Rc, -Rc @ t], -1)
        out = O.render_rays_v2(params, obj["img"], obj["mask"], cam2opt, obj["obj_diag"], obj["K"], obj["roi"], hpams["n_samples"], sc, tc,
                               True, im_sz=hpams["render_im_sz"])
        loss, _, _, ps = O.optimise_losses(out[0], out[2], out[3], out[4], hpams["loss_occ_coef"])
        loss.backward()
        with torch.no_grad():
            O.render_rays_specified(params, obj["img"], obj["mask"], cam2opt.detach(), obj["obj_diag"], obj["K"], obj["roi"], x_vec, y_vec,
                                    hpams["n_samples"], sc, tc, True)           # consumes the same jitter draw
        psnr.append(float(ps)); rot_err.append(float(D.rot_dist(cam2opt[:, :3].detach().T, R_gt)))
        if it > reg_iters:
            optim.step()
    return np.array(psnr), np.array(rot_err)


def test_optimise_loop_trace_matches_oracle_loop():
    import supnerf_amd as A
    D = A.driver
    dev = torch.device("cuda:0")
    params = O.init_decoder_params()
    model = A.CodeNeRF(3, 1); model.load_state_dict(params); model = model.to(dev)
    hp = D.load_hpams()
    hp["render_im_sz"] = 16
    hp["optimize"]["num_opts"] = 8
    obj = D.make_objects([21], 16)[0]
    g = torch.Generator().manual_seed(5)
    sc0, tc0 = torch.randn(1, 256, generator=g) * 0.3, torch.randn(1, 256, generator=g) * 0.3
    torch.manual_seed(123)
    ps_ref, rot_ref = oracle_loop(params, obj, hp, sc0, tc0, seed=9, reg_iters=1, pose_noise=(0.05, 0.3), D=D)
    torch.manual_seed(123)
    m, sc, tc, pose = D.optimize_object(model, dev, obj, hp, sc0, tc0, pose_noise=(0.05, 0.3), reg_iters=1, seed=9)
    ps, rot = m[:, 0].numpy(), m[:, 2].numpy()
    # identical until the first optimiser step, then fp32-level drift amplified by Adam's normalisation
    assert np.abs(ps[:3] - ps_ref[:3]).max() < 1e-3
    assert np.abs(ps - ps_ref).max() < 0.05, (ps, ps_ref)
    assert np.abs(rot - rot_ref).max() < 2e-3
    assert ps[-1] > ps[0]                       # and the optimisation makes progress


def test_sharded_objects_single_process():
    import supnerf_amd as A
    D = A.driver
    dev = torch.device("cuda:0")
    model = A.CodeNeRF(3, 1); model.load_state_dict(O.init_decoder_params()); model = model.to(dev)
    hp = D.load_hpams(); hp["render_im_sz"] = 8; hp["optimize"]["num_opts"] = 3
    torch.manual_seed(0)
    full = D.optimize_objects(model, dev, 3, hp, rank=0, world_size=1, seed=1)
    assert full.shape == (3, 12) and bool(torch.isfinite(full).all())
    # rank 1 of 2 owns object 2 only (tail kept); same seeds -> same rows as in the full run up to the jitter stream
    assert list(D.shard_slice(3, 2, 1)) == [2]


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_training_step_matches_oracle(oracle_params, precision):
    """supnerf_amd.trainer.train_step on the HIP training path: losses and EVERY gradient of the first
    iteration against the oracle's autograd on the CPU, then three more iterations must keep lowering the loss.  Both arithmetics
    (fp32: exact, the training default; bf16x3: split-bf16 chains and weight-gradient products, opt-in).  Mask-matched (tests/relu_bits.py): the oracle differentiates with
    the ReLU bits the forward chain saved, so a flipped unit cannot move an entry by a percent and both arithmetics are held to 2e-4 of
    each tensor's largest entry (round 2 accepted 5e-3 for split-bf16)."""
    import supnerf_amd
    from relu_bits import relu_bits_of
    T = supnerf_amd.trainer
    dev = torch.device("cuda:0")
    rel = 2e-4
    m = supnerf_amd.CodeNeRF(shape_blocks=3, texture_blocks=1)
    m.load_state_dict(oracle_params, strict=True)
    m.precision = precision
    m = m.to(dev)
    m.train_decoder_weights = True
    codes = T.CodeTables(5, 256, seed=4).to(dev)
    g = torch.Generator().manual_seed(21)
    B, n, S = 2, 32, 64
    batch = dict(code_idx=torch.tensor([3, 1]), xyz=torch.rand(B, n, S, 3, generator=g) - 0.5,
                 viewdir=torch.nn.functional.normalize(torch.randn(B, n, 1, 3, generator=g), dim=-1).repeat(1, 1, S, 1),
                 z_vals=torch.sort(torch.rand(B, S, generator=g) * 4 + 9, dim=-1)[0], rgb_tgt=torch.rand(B, n, 3, generator=g),
                 occ_pixels=(torch.randint(0, 3, (B, n, 1), generator=g) - 1).float())
    cpu_batch = batch
    # product path
    batch = {k: v.to(dev) for k, v in batch.items()}
    params = list(m.parameters()) + list(codes.parameters())
    bucket = T.GradBucket(params)
    sc, tc = codes(batch["code_idx"])
    seen = {}

    def capturing(*a):
        out = m(*a)
        seen["masks"] = relu_bits_of(out[0], 3, 1)          # (before backward frees what the operator saved)
        return out
    losses_all, total = T.nerf_losses(capturing, batch["xyz"], batch["viewdir"], sc, tc, batch["z_vals"], batch["rgb_tgt"], batch["occ_pixels"], 0.1)
    total.backward()
    bucket.check_views()
    # oracle gradients (CPU autograd) on the same piecewise-linear function
    p_cpu = {k: v.detach().cpu().clone().requires_grad_() for k, v in m.named_parameters()}
    w_sc = codes.shape_codes.weight.detach().cpu().clone().requires_grad_()
    w_tc = codes.texture_codes.weight.detach().cpu().clone().requires_grad_()
    with O.given_relu_masks(seen["masks"]):
        ref = O.training_losses(p_cpu, cpu_batch["xyz"], cpu_batch["viewdir"], w_sc[cpu_batch["code_idx"]], w_tc[cpu_batch["code_idx"]],
                                cpu_batch["z_vals"], cpu_batch["rgb_tgt"], cpu_batch["occ_pixels"], 0.1)
        ref[0].backward()
    assert abs(float(total) - float(ref[0])) < 1e-5 and abs(float(losses_all["psnr"]) - float(ref[4])) < 0.01
    assert abs(float(losses_all["loss_reg"]) - float(ref[3])) < 1e-5
    worst = 0.0
    for (name, p) in list(m.named_parameters()) + [("shape_codes", codes.shape_codes.weight), ("texture_codes", codes.texture_codes.weight)]:
        want = {"shape_codes": w_sc, "texture_codes": w_tc}.get(name, p_cpu.get(name)).grad
        err = float((p.grad.cpu() - want).abs().max())
        worst = max(worst, err / (float(want.abs().max()) + 1e-30))
        assert err <= rel * float(want.abs().max()) + 1e-7, (name, err, float(want.abs().max()))
    print(f"[training step, {precision}] worst gradient entry, relative to its tensor's largest: {worst:.2e}")
    bucket.zero()
    hp = dict(lr_schedule=[dict(lr=1e-4, interval=100), dict(lr=1e-3, interval=100)])
    opt = T.make_optimizer(m, codes, hp)
    trace = [float(T.train_step(m, codes, opt, bucket, batch, 0.1)["loss_total"]) for _ in range(4)]
    assert all(b < a for a, b in zip(trace, trace[1:])), trace


def test_training_outcome_fp32_and_bf16x3_track_the_oracle(oracle_params):
    """Which arithmetic may config 5 train in?  (VERDICT r2 #4b)  60 steps of ``trainer.train_step`` over four fixed mini-batches (2 objects
    x 32 rays x 64 samples, decoder + codes trained, the reference's AdamW) with the exact-fp32 kernels and with the split kernels ("auto" =
    "bf16x3": fp16 pieces in the forward chain, bf16 pieces in the backward chain and the weight-gradient products),
    against the SAME 60 steps on the CPU oracle in float64 (the truth) and in float32 (the reference's arithmetic: its distance from the
    truth is the floor).  Loss curve and final decoder weights, the latter relative to how far training moved each tensor."""
    import supnerf_amd
    T = supnerf_amd.trainer
    dev = torch.device("cuda:0")
    STEPS, B, n, S = 60, 2, 32, 64
    g = torch.Generator().manual_seed(5)
    batches = [dict(code_idx=torch.tensor([(2 * k) % 6, (2 * k + 3) % 6]), xyz=torch.rand(B, n, S, 3, generator=g) - 0.5,
                    viewdir=torch.nn.functional.normalize(torch.randn(B, n, 1, 3, generator=g), dim=-1).repeat(1, 1, S, 1),
                    z_vals=torch.sort(torch.rand(B, S, generator=g) * 4 + 9, dim=-1)[0], rgb_tgt=torch.rand(B, n, 3, generator=g),
                    occ_pixels=(torch.randint(0, 3, (B, n, 1), generator=g) - 1).float()) for k in range(4)]
    hp = dict(lr_schedule=[dict(lr=1e-4, interval=40000), dict(lr=1e-4, interval=40000)])       # the reference's rates (jsonfiles/*.json)

    def oracle_run(dtype):
        c = lambda t: t.to(dtype) if t.is_floating_point() else t
        p = {k: c(v).clone().requires_grad_() for k, v in oracle_params.items()}
        codes = T.CodeTables(6, 256, seed=4)
        w_sc, w_tc = c(codes.shape_codes.weight.detach()).clone().requires_grad_(), c(codes.texture_codes.weight.detach()).clone().requires_grad_()
        opt = torch.optim.AdamW([{"params": list(p.values()), "lr": 1e-4}, {"params": [w_sc], "lr": 1e-4}, {"params": [w_tc], "lr": 1e-4}])
        curve = []
        for it in range(STEPS):
            b = {k: c(v) for k, v in batches[it % 4].items()}
            opt.zero_grad()
            total = O.training_losses(p, b["xyz"], b["viewdir"], w_sc[b["code_idx"]], w_tc[b["code_idx"]], b["z_vals"], b["rgb_tgt"], b["occ_pixels"], 0.1)[0]
            total.backward()
            opt.step()
            curve.append(float(total))
        return np.array(curve), {k: v.detach().double() for k, v in p.items()}

    def gpu_run(precision):
        m = supnerf_amd.CodeNeRF(3, 1); m.load_state_dict(oracle_params, strict=True); m.precision = precision
        m = m.to(dev); m.train_decoder_weights = True
        codes = T.CodeTables(6, 256, seed=4).to(dev)
        bucket = T.GradBucket(list(m.parameters()) + list(codes.parameters()), row_sparse=list(codes.parameters()))
        opt = T.make_optimizer(m, codes, hp)
        dev_batches = [{k: v.to(dev) for k, v in b.items()} for b in batches]
        curve = [float(T.train_step(m, codes, opt, bucket, dev_batches[it % 4], 0.1)["loss_total"]) for it in range(STEPS)]
        return np.array(curve), {k: v.detach().double().cpu() for k, v in m.named_parameters()}

    c64, w64 = oracle_run(torch.float64)
    c32, w32 = oracle_run(torch.float32)
    init = {k: v.double() for k, v in oracle_params.items()}

    def weight_dev(w, entrywise=False):
        """Worst tensor: distance from the float64 run's weights relative to how far that run moved the tensor (L2; entry-wise maxima are
        printed only: Adam turns a gradient entry at rounding level into a full-size step of random sign, so single dead entries differ
        by their whole movement between any two runs)."""
        if entrywise:
            return max(float((w[k] - w64[k]).abs().max()) / (float((w64[k] - init[k]).abs().max()) + 1e-12) for k in w64)
        return max(float((w[k] - w64[k]).norm()) / (float((w64[k] - init[k]).norm()) + 1e-12) for k in w64)
    floor_c, floor_w = float(np.abs(c32 - c64).max()), weight_dev(w32)
    assert c64[-1] < c64[0] - 0.01                                      # the 60 steps do train
    res = {}
    for precision in ("fp32", "auto", "bf16x3"):
        c, w = gpu_run(precision)
        dc, dw = float(np.abs(c - c64).max()), weight_dev(w)
        res[precision] = (c, w)
        print(f"[training outcome, {precision}] loss {c[0]:.4f} -> {c[-1]:.4f}; vs the float64 oracle run: loss curve {dc:.2e} (fp32 oracle {floor_c:.2e}), "
              f"final weights {dw:.2e} of each tensor's training movement in L2 (fp32 oracle {floor_w:.2e}); worst single entry "
              f"{weight_dev(w, True):.2e} ({weight_dev(w32, True):.2e})")
        # every arithmetic: as close to the truth as the reference's own fp32 arithmetic is, within a factor of three.  (The split kernels
        # with BF16 pieces in the forward chain, rounds 1-2, measured 3.9e-5 / 1.1e-2 here -- 27x / 8x the floor; with FP16 pieces: 2.1e-6 / 1.21e-3.)
        assert dc < 3 * floor_c + 1e-5 and dw < 3 * floor_w + 1e-3, (precision, dc, floor_c, dw, floor_w)
    # "auto" in training mode = the split kernels throughout (fp16 pieces in the forward chain): the same launches as "bf16x3"
    assert np.array_equal(res["auto"][0], res["bf16x3"][0]) and all(torch.equal(res["auto"][1][k], res["bf16x3"][1][k]) for k in res["bf16x3"][1])


def test_batched_loop_equals_per_object_loop():
    """BASELINE config 3 in miniature: B objects per launch must follow the same trajectories as the one-object loop when
    both get the same jitter draws (exact fp32 kernels; the only difference is the batching of launches and optimiser rows)."""
    import supnerf_amd as A
    D = A.driver
    dev = torch.device("cuda:0")
    model = A.CodeNeRF(3, 1); model.load_state_dict(O.init_decoder_params()); model = model.to(dev)
    model.precision = "fp32"
    hp = D.load_hpams(); hp["render_im_sz"] = 16; hp["optimize"]["num_opts"] = 7
    ids, T_, S = [3, 8, 21], 7, hp["n_samples"]
    objs = D.make_objects(ids, 16)
    g = torch.Generator().manual_seed(2)
    sc0, tc0 = torch.randn(3, 256, generator=g) * 0.3, torch.randn(3, 256, generator=g) * 0.3
    jit = torch.rand(T_, 2, 3, S, generator=g)
    seeds = [100 + i for i in ids]
    mb, scb, tcb, poseb = D.optimize_objects_batched(model, dev, objs, hp, sc0, tc0, seeds, reg_iters=1, jitter=jit)
    assert mb.shape == (3, T_, 4)
    for b, ob in enumerate(objs):
        m1, sc1, tc1, pose1 = D.optimize_object(model, dev, ob, hp, sc0[b:b + 1], tc0[b:b + 1], reg_iters=1, seed=seeds[b], jitter=jit[:, :, b])
        d = (mb[b].cpu() - m1).abs()
        assert float(d[:2].max()) < 1e-4, (b, d[:2])                           # before the first optimiser step: same numbers
        assert float(d[:, 0].max()) < 0.05 and float(d[:, 2:].max()) < 2e-3, (b, d)   # afterwards fp32 drift through Adam's normalisation
        assert float((poseb[b].cpu() - pose1.cpu()).abs().max()) < 2e-3 and float((scb[b].cpu() - sc1[0].cpu()).abs().max()) < 2e-2
    assert bool((mb[:, -1, 0] > mb[:, 0, 0]).all())                            # every object improves its PSNR


def test_optimize_objects_batches_and_shards():
    import supnerf_amd as A
    D = A.driver
    dev = torch.device("cuda:0")
    model = A.CodeNeRF(3, 1); model.load_state_dict(O.init_decoder_params()); model = model.to(dev)
    hp = D.load_hpams(); hp["render_im_sz"] = 8; hp["optimize"]["num_opts"] = 3
    full = D.optimize_objects(model, dev, 5, hp, seed=1, batch=2)              # batches of 2, 2, 1
    one = D.optimize_objects(model, dev, 5, hp, seed=1, batch=64)
    assert full.shape == (5, 12) and bool(torch.isfinite(full).all())
    assert float((full - one).abs().max()) < 1e-3                              # the batch size does not change an object's numbers


def test_depth_metric_on_measured_lidar_returns():
    """Objects that carry lidar returns (``lidar_xy``, ``lidar_depth``; every object its own count): the depth column of the metric rows is
    the reference's depth L1 against the measurements over ALL of the object's returns (src/optimizer_nuscenes.py:751-765,1736-1741) --
    the fused batched loop, the one-object loop and the API-structured loop agree, the batch composition does not change an object's number,
    and the counts the driver reports are the per-object ones (they weight the depth-error mean in the reference's evaluation)."""
    import supnerf_amd as A
    D = A.driver
    dev = torch.device("cuda:0")
    model = A.CodeNeRF(3, 1); model.load_state_dict(O.init_decoder_params()); model = model.to(dev)
    model.precision = "fp32"
    hp = D.load_hpams(); hp["render_im_sz"] = 16; hp["optimize"]["num_opts"] = 4
    ids, T_, S = [3, 8, 21, 30], 4, hp["n_samples"]
    objs = D.make_objects(ids, 16, lidar=True)
    want_cnt = [len(ob["lidar_depth"]) for ob in objs]
    assert len(set(want_cnt)) > 1                                                   # different counts in one launch
    g = torch.Generator().manual_seed(2)
    sc0, tc0 = torch.randn(4, 256, generator=g) * 0.3, torch.randn(4, 256, generator=g) * 0.3
    jit = torch.rand(T_, 2, 4, S, generator=g)
    seeds = [100 + i for i in ids]
    info = {}
    mb, *_ = D.optimize_objects_batched(model, dev, objs, hp, sc0, tc0, seeds, reg_iters=1, jitter=jit, info=info)
    assert info["lidar_count"].tolist() == want_cnt
    for b, ob in enumerate(objs):
        i1, i2 = {}, {}
        m1, *_ = D.optimize_object(model, dev, ob, hp, sc0[b:b + 1], tc0[b:b + 1], reg_iters=1, seed=seeds[b], jitter=jit[:, :, b], info=i1)
        m2, *_ = D.optimize_object_api(model, dev, ob, hp, sc0[b:b + 1], tc0[b:b + 1], reg_iters=1, seed=seeds[b], jitter=jit[:, :, b], info=i2)
        assert i1["lidar_count"].tolist() == [want_cnt[b]] and i2["lidar_count"].tolist() == [want_cnt[b]]
        assert float((mb[b, :, 1].cpu() - m1[:, 1]).abs().max()) < 2e-3, (b, mb[b, :, 1], m1[:, 1])      # batched == one object
        assert float((m1[:2, 1] - m2[:2, 1]).abs().max()) < 1e-4, (b, m1[:, 1], m2[:, 1])               # fused == API loop (before Adam's drift)
        # the number itself at iteration 0: depth L1 of the API render at the lidar pixels against the measurements
        xy, gt = ob["lidar_xy"], torch.from_numpy(ob["lidar_depth"])
        assert float(m1[0, 1]) > 0.05                                                # metres off the measured surface, not a change against itself
    rows = D.optimize_objects(model, dev, 3, hp, seed=1, batch=2, return_counts=True, lidar=True)
    assert rows.shape == (3, 4 * 4 + 1) and rows[:, -1].tolist() == [float(len(ob["lidar_depth"])) for ob in D.make_objects([0, 1, 2], 16, lidar=True)]


def test_waymo_loop_runs():
    """optimize_waymo.py's configuration (jsonfiles/supnerf.waymo.car.json): the KITTI-convention loop on 1920 x 1280 images with Waymo
    intrinsics; poses converted by obj_pose_kitti2nusc, roi_margin 15 (src/optimizer_waymo.py:125,140)."""
    import supnerf_amd as A
    D = A.driver
    dev = torch.device("cuda:0")
    model = A.CodeNeRF(3, 1); model.load_state_dict(O.init_decoder_params()); model = model.to(dev)
    hp = D.load_hpams(dataset="waymo")
    hp["optimize"]["num_opts"] = 6
    objs = D.make_kitti_objects([2, 5], hp)
    assert all(float(ob["K"][0, 0]) > 2000 and int(ob["roi"][2]) < 1920 and int(ob["roi"][3]) < 1280 for ob in objs)
    g = torch.Generator().manual_seed(4)
    m, sc, tc, pose = D.optimize_objects_batched(model, dev, objs, hp, torch.randn(2, 256, generator=g) * 0.3, torch.randn(2, 256, generator=g) * 0.3,
                                                 [0, 1], reg_iters=1)
    assert m.shape == (2, 6, 4) and bool(torch.isfinite(m).all()) and bool((m[:, -1, 0] > m[:, 0, 0]).all())


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` as the driver calls it (no launcher, no WORLD_SIZE): the script starts its two ranks itself and prints ONE
    JSON line for the job.  On this one-GPU box the ranks rendezvous over gloo and share the card (SNR_BENCH_BACKEND=gloo: a rehearsal of
    the N > 1 branch, the numbers mean nothing); on an 8-GPU node the same command runs one rank per GPU over RCCL."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SNR_BENCH_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--headline-only"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["unit"] == "rays/s" and line["value"] > 0 and line["scaling"] == "weak"
    # one record per rank, gathered over the communicator: which device every rank rendered on and at what rate on its own clock
    assert [r["rank"] for r in line["ranks"]] == [0, 1] and all(r["rays_per_s"] > 0 and r["pci_bus_id"] for r in line["ranks"])
    assert line["distinct_devices"] == min(2, torch.cuda.device_count())          # (gloo rehearsal on one card: both ranks report the same device)
    assert line["roofline"]["traffic_measured_in_run"] is False


@pytest.mark.parametrize("prec", ["fp32", "bf16x3"])
def test_optimise_loop_is_bit_reproducible(prec):
    """Two runs of the fused loop from the same seeds (torch's global CPU generator feeds the jitter, like the reference's torch.rand(S))
    give the same bits in every metric of every iteration, the final codes and the pose: no launch of the iteration -- forward, backward,
    the latent-gradient reductions, loss tail, AdamW -- depends on scheduling order (no atomics, fixed reduction trees)."""
    import supnerf_amd as A
    D = A.driver
    dev = torch.device("cuda:0")
    model = A.CodeNeRF(3, 1); model.load_state_dict(O.init_decoder_params()); model = model.to(dev)
    model.precision = prec
    hp = D.load_hpams()
    hp["render_im_sz"] = 32
    hp["optimize"]["num_opts"] = 12
    objs = D.make_objects([31, 32], 32)
    g = torch.Generator().manual_seed(7)
    sc0, tc0 = torch.randn(2, 256, generator=g) * 0.3, torch.randn(2, 256, generator=g) * 0.3
    runs = []
    for _ in range(2):
        torch.manual_seed(77)
        one = D.optimize_object(model, dev, objs[0], hp, sc0[:1], tc0[:1], seed=3)
        both = D.optimize_objects_batched(model, dev, objs, hp, sc0, tc0, seeds=[3, 4])
        torch.cuda.synchronize()
        runs.append([t.cpu() for t in one] + [t.cpu() for t in both])
    for a, b in zip(*runs):
        assert torch.equal(a, b), float((a - b).abs().max())
