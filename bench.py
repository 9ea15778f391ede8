#!/usr/bin/env python3
"""bench.py -- rays/s of the fused render hot path at BASELINE.json's configuration.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

One *step* = one fused forward pass (sample -> frame transform -> positional encoding -> decoder -> composite) over
one object's ray batch: 4096 rays x 64 samples (BASELINE configs[1]: supnerf.nusc.vehicle.car.json hyper-parameters,
im_sz 64, synthetic nuScenes-like car, random-init decoder with the sigma bias of SURVEY 8d), inputs resident in
HBM.  Every rank renders its own object (objects are independent: weak scaling, no data-path collective); the only
collective is the max-reduce of the elapsed time.  Rank 0 prints ONE JSON line.

Extra legs that are reported but are NOT `value`:
  * forward+backward (codes + pose gradients) rays/s, the optimiser's inner iteration;
  * `roofline`: algorithmic FLOPs (57.56 MFLOP/ray) / kernel time from device events on the
    launch stream, against the fp32-matrix peak of MI355X_MICROARCH.md (157.3 TFLOP/s);
  * `cpu_baseline`: the CPU oracle (pure PyTorch restatement of the reference, oracle/) timed on this host's cores
    over a bounded sample, plus the parity numbers (PSNR delta, depth L1) of the GPU output against it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_RAY = 57.56e6          # forward, S = 64
# HBM bytes per 4096x64 forward launch from the committed counter profile (cannot be collected from inside this process)
HBM_TRAFFIC = {"bf16x3": 16020480.0, "fp32": None}
PEAK_TFLOPS = {"fp32": 157.3,    # MI355X_MICROARCH.md "Peak FP32 (matrix)": v_mfma_f32_32x32x2_f32
               "bf16x3": 2500.0}  # dense BF16 MFMA peak; the split-bf16 path issues 3 MFMAs per algorithmic product
N_RAYS, N_SAMPLES, IM_SZ = 4096, 64, 64


def make_workload(dev, seed):
    """One synthetic object: rays of a 64x64 grid over its roi, shared stratified depths, codes, decoder weights."""
    import supnerf_amd as A
    from supnerf_amd import synthetic as O
    params = O.init_decoder_params(seed=0)
    model = A.CodeNeRF(shape_blocks=3, texture_blocks=1)
    model.load_state_dict(params)
    model = model.to(dev)
    ob = O.synthetic_object(seed)
    img, mask = O.synthetic_targets(seed, IM_SZ)
    gen = torch.Generator().manual_seed(seed)
    sc = (torch.randn(1, 256, generator=gen) * 0.3).to(dev)
    tc = (torch.randn(1, 256, generator=gen) * 0.3).to(dev)
    jit = torch.rand(N_SAMPLES, generator=gen)
    return dict(model=model, params=params, ob=ob, img=img, mask=mask, sc=sc, tc=tc, jit=jit)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the headline launches (no other-precision / forward+backward / optimise-loop / HBM-kernel legs): the form "
                         "profiled for profiles/*_bench_kernel_stats.csv, so the per-kernel average there is the headline kernel alone")
    ap.add_argument("--precision", default="bf16x3", choices=["bf16x3", "fp32"],
                    help="decoder GEMM arithmetic of the headline run (the other mode is timed too and reported under 'extra')")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                         "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)     # RCCL over xGMI

    import supnerf_amd as A
    from supnerf_amd import ops, utils as U
    w = make_workload(dev, seed=100 + rank)
    model, ob = w["model"], w["ob"]

    # ---- device-resident inputs of the hot path (ray generation is outside the path's timed region: it is part of the
    # caller-side glue and costs microseconds)
    with torch.no_grad():
        rays_o, viewdir = U.get_rays(ob["K"], ob["cam_pose"].to(dev), ob["roi"], uv_steps=[IM_SZ, IM_SZ])
        near, far = U._sphere_bounds(ob["cam_pose"], ob["obj_diag"])
        z = U._shared_depths(near, far, N_SAMPLES, dev, jitter=w["jit"])
        lat = model.latent_terms(w["sc"], w["tc"])
    packed = model.packed_weights()
    div = torch.full((1,), float(ob["obj_diag"]), device=dev)
    cfg = ops.RenderCfg(N_SAMPLES, ops.Z_SHARED, N_RAYS, 3, 1, frame=U._frame(False, False, True), precision=args.precision)
    other = "fp32" if args.precision == "bf16x3" else "bf16x3"
    cfg_other = ops.RenderCfg(N_SAMPLES, ops.Z_SHARED, N_RAYS, 3, 1, frame=U._frame(False, False, True), precision=other)
    # like model.fused_render: the latent terms also folded into the next layers' biases (prepared beside the latent terms, outside the path)
    cfg.latent_bias = cfg_other.latent_bias = model.latent_biases(lat)
    model.precision = args.precision
    assert rays_o.shape[0] == N_RAYS

    def step(c=cfg):
        return ops.render_fwd(rays_o, viewdir, z, div, None, lat, packed, c)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    value = world * N_RAYS * args.steps / elapsed

    # ---- kernel time from device events on the launch stream (torch's current stream IS the launch stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(args.steps):
        out = step()
    e1.record()
    torch.cuda.synchronize()
    kern_ms = e0.elapsed_time(e1) / args.steps
    achieved = N_RAYS * FLOP_PER_RAY / (kern_ms * 1e-3) / 1e12
    other_ms, other_tf, fwd_bwd_rays, fb_elapsed, n_fb, loop, hbm = None, None, None, None, 1, {}, {}
    if not args.headline_only:
        # the other arithmetic mode, same launches (reported, not the headline)
        for _ in range(5):
            step(cfg_other)
        torch.cuda.synchronize()
        e0.record()
        n_other = max(10, args.steps // 4)
        for _ in range(n_other):
            out_other = step(cfg_other)
        e1.record()
        torch.cuda.synchronize()
        other_ms = e0.elapsed_time(e1) / n_other
        other_tf = N_RAYS * FLOP_PER_RAY / (other_ms * 1e-3) / 1e12

        # ---- forward + backward leg (the optimiser's inner iteration: gradients wrt codes and pose)
        sc_g, tc_g = w["sc"].clone().requires_grad_(), w["tc"].clone().requires_grad_()
        pose_g = ob["cam_pose"].to(dev).requires_grad_()
        tgt = w["img"].reshape(-1, 3).to(dev)

        def step_fb():
            ro, vd = U.get_rays(ob["K"], pose_g, ob["roi"], uv_steps=[IM_SZ, IM_SZ])
            rgb, depth, acc = model.fused_render(ro, vd, z, div, None, sc_g, tc_g, cfg)
            loss = ((rgb - tgt) ** 2).mean() + 0.1 * acc.mean()
            sc_g.grad = tc_g.grad = pose_g.grad = None
            loss.backward()

        for _ in range(3):
            step_fb()
        barrier()
        n_fb = max(10, args.steps // 4)
        t0 = time.perf_counter()
        for _ in range(n_fb):
            step_fb()
        barrier()
        fb_elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([fb_elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            fb_elapsed = float(t.item())
        fwd_bwd_rays = world * N_RAYS * n_fb / fb_elapsed

        # ---- BASELINE config 3 in one process: the optimise loop over 64 objects, all of them in one launch per iteration
        loop = {}
        if rank == 0 and world == 1:      # (single-GPU runs only: at N > 1 every rank does the same work and leaves together)
            from supnerf_amd import driver as D
            hp = D.load_hpams()
            hp["render_im_sz"] = IM_SZ
            n_obj, n_it = 64, 4
            hp["optimize"]["num_opts"] = n_it
            objs = D.make_objects(list(range(200, 200 + n_obj)), IM_SZ)
            gl = torch.Generator().manual_seed(3)
            sc_l, tc_l = torch.randn(n_obj, 256, generator=gl) * 0.3, torch.randn(n_obj, 256, generator=gl) * 0.3
            D.optimize_objects_batched(model, dev, objs[:8], hp, sc_l[:8], tc_l[:8], list(range(8)))          # warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            D.optimize_objects_batched(model, dev, objs, hp, sc_l, tc_l, list(range(n_obj)))
            torch.cuda.synchronize()
            t_b = time.perf_counter() - t0
            n_it1 = 50                                   # the reference runs 100 iterations per object: amortise the set-up alike
            hp["optimize"]["num_opts"] = n_it1
            t0 = time.perf_counter()
            D.optimize_object(model, dev, objs[0], hp, sc_l[:1], tc_l[:1], seed=0)
            torch.cuda.synchronize()
            t_1 = time.perf_counter() - t0
            t0 = time.perf_counter()                     # the same object through the batched loop replayed as a HIP graph
            D.optimize_objects_batched(model, dev, objs[:1], hp, sc_l[:1], tc_l[:1], [0], graph=True)
            torch.cuda.synchronize()
            t_g = time.perf_counter() - t0
            loop = {"objects": n_obj, "iterations": n_it, "rays_per_object": N_RAYS,
                    "batched_ms_per_iteration": t_b / n_it * 1e3, "batched_object_iterations_per_s": n_obj * n_it / t_b,
                    "batched_rays_per_s_fwd_bwd_plus_depth_render": n_obj * n_it * N_RAYS / t_b,
                    "one_object_loop_ms_per_iteration": t_1 / n_it1 * 1e3, "one_object_loop_object_iterations_per_s": n_it1 / t_1,
                    "one_object_loop_iterations": n_it1,
                    "one_object_hip_graph_ms_per_iteration": t_g / n_it1 * 1e3, "one_object_hip_graph_object_iterations_per_s": n_it1 / t_g,
                    "note": "iteration = fused forward + backward (codes, pose) + 64-pixel depth render + AdamW; the batched loop runs all objects "
                            "in one launch each and never syncs with the host; the one-object loop is the reference's structure; 'hip_graph' = the batched loop at one object with the iteration "
                            "recorded once (two graphs) and replayed (set-up and recording included everywhere)"}
            del objs

        # ---- the two HBM-bound stand-alone kernels (encode with PE output, composite) at 16 objects x 4096 x 64: achieved GB/s
        hbm = {}
        if rank == 0 and world == 1:
            Bh = 64
            ro_h, vd_h = rays_o.repeat(Bh, 1), viewdir.repeat(Bh, 1)
            z_h = z[None].repeat(Bh, 1).contiguous()
            div_h = div.repeat(Bh)
            cfg_h = ops.RenderCfg(N_SAMPLES, ops.Z_PER_OBJECT, N_RAYS, 3, 1, frame=U._frame(False, False, True))
            P_h = Bh * N_RAYS * N_SAMPLES

            def timed(fn, n=10):
                for _ in range(2):
                    fn()
                torch.cuda.synchronize()
                e0.record()
                for _ in range(n):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / n * 1e-3
            t_enc = timed(lambda: ops.encode(ro_h, vd_h, z_h, div_h, None, cfg_h, want_pe=True))
            enc_bytes = P_h * (12 + 12 + 4 + 63 * 4) + Bh * N_RAYS * (24 + 27 * 4)          # xyz, viewdir, z, PE(xyz) per point; rays in, PE(dir) out
            sig_h = torch.rand(Bh * N_RAYS, N_SAMPLES, device=dev)
            rgb_h = torch.rand(Bh * N_RAYS, N_SAMPLES, 3, device=dev)
            t_cmp = timed(lambda: ops.composite_fwd(sig_h, rgb_h, z_h, ops.Z_PER_OBJECT, False, N_RAYS))
            cmp_bytes = P_h * 16 + Bh * N_RAYS * 20
            # scene composite (vis_scene): 131072 pixels x 3 objects x 64 samples, depth merge + composite
            P_s, n_s = 131072, 3 * N_SAMPLES
            z_s = (torch.rand(P_s, 3, 1, device=dev) * 20 + 2 + torch.sort(torch.rand(P_s, 3, N_SAMPLES, device=dev), dim=-1)[0] * 4).view(P_s, n_s)
            sig_s, rgb_s = torch.rand(P_s, n_s, device=dev), torch.rand(P_s, n_s, 3, device=dev)
            t_scn = timed(lambda: ops.scene_composite(sig_s, rgb_s, z_s, True))
            scn_bytes = P_s * n_s * 20 + P_s * 20
            del z_s, sig_s, rgb_s
            hbm = {"encode": {"GB_per_s": enc_bytes / t_enc / 1e9, "ms": t_enc * 1e3, "bytes": enc_bytes, "frac_of_8TBps": enc_bytes / t_enc / 8e12},
                   "scene_composite": {"GB_per_s": scn_bytes / t_scn / 1e9, "ms": t_scn * 1e3, "bytes": scn_bytes, "frac_of_8TBps": scn_bytes / t_scn / 8e12,
                                       "shape": f"{P_s} pixels x 3 objects x {N_SAMPLES} samples"},
                   "composite_fwd": {"GB_per_s": cmp_bytes / t_cmp / 1e9, "ms": t_cmp * 1e3, "bytes": cmp_bytes, "frac_of_8TBps": cmp_bytes / t_cmp / 8e12},
                   "shape": f"{Bh} objects x {N_RAYS} rays x {N_SAMPLES} samples"}
            del sig_h, rgb_h

    result = {
        "metric": "rays/sec at 4096 rays x 64 samples (fused render forward)",
        "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.precision == "fp32" else "bf16x3 (fp32 operands split into bf16 hi+lo, 3 bf16 MFMAs per product, fp32 accumulate)",
        "data": "synthetic",
        "config": {"workload": "supnerf.nusc.vehicle.car.json decoder (shape_blocks 3, texture_blocks 1, W 256), 1 object per GPU, "
                               "4096 rays x 64 samples, family-A render (render_rays_v2 tail)", "precision": args.precision,
                   "rays": N_RAYS, "samples": N_SAMPLES, "objects_per_gpu": 1, "sharding": "objects across ranks, no data-path collective"},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS[args.precision], "unit": "TFLOP/s",
                     "frac": achieved / PEAK_TFLOPS[args.precision], "traffic": HBM_TRAFFIC.get(args.precision),
                     "traffic_unit": "bytes per launch", "traffic_source": "profiles/r01_v7_fwd_pmc.json: rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) "
                                                                          "+ WRITE_SIZE in separate passes; 8 XCD L2s fetch the 1.74 MB weight stream once each",
                     "kernel": "bf16_fwd_kernel<1,false,true>" if args.precision == "bf16x3" else "decoder_fwd_kernel<1>",
                     "kernel_ms": kern_ms, "flop_per_launch": N_RAYS * FLOP_PER_RAY,
                     "note": "achieved = ALGORITHMIC flops (57.56 MFLOP/ray) / kernel time; bf16x3 issues 3x that on the bf16 MFMA pipe "
                             "(mfma_pipe_frac), fp32 issues 1x on the fp32 MFMA pipe",
                     "mfma_pipe_frac": (3.0 if args.precision == "bf16x3" else 1.0) * achieved / PEAK_TFLOPS[args.precision]},
        "extra": None if args.headline_only else {
                  other + "_mode": {"rays_per_s": N_RAYS / (other_ms * 1e-3), "kernel_ms": other_ms, "achieved_tflops": other_tf,
                                    "frac_of_peak": other_tf / PEAK_TFLOPS[other]},
                  "hbm_bound_kernels": hbm, "optimise_loop": loop,
                  "fwd_bwd_rays_per_s": fwd_bwd_rays, "fwd_bwd_ms_per_iter": fb_elapsed / n_fb * 1e3,
                  "fwd_bwd_note": "forward + backward to shape/texture codes and camera pose, incl. ray generation and loss in torch"},
    }

    # ---- CPU baseline + parity (rank 0, N = 1 only): the oracle is the checker and the baseline, never the product
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import supnerf_oracle as O
        # the box's CPU share for one GPU is 16 cores whatever os.cpu_count() says; more threads only thrash
        cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
        torch.set_num_threads(cores)
        n_cpu = 1024                       # a quarter of the workload: same rays, same codes, same weights
        ro_c, vd_c, z_c = rays_o[:n_cpu].cpu(), viewdir[:n_cpu].cpu(), z.cpu()
        sc_c, tc_c = w["sc"].cpu(), w["tc"].cpu()

        def cpu_pass():
            with torch.no_grad():
                xyz, vd = O.points_on_rays(ro_c, vd_c, z_c)
                xyz = xyz / ob["obj_diag"]
                xyz, vd = O.object_frame_transforms(xyz, vd, False, False, True)
                sig, rgb = O.decoder_forward(w["params"], xyz, vd, sc_c, tc_c)
                return O.volume_rendering2(sig, rgb, z_c)
        ref = cpu_pass()
        times = []
        t_start = time.perf_counter()
        while len(times) < 3 or (time.perf_counter() - t_start < 10 and len(times) < 30):
            t0 = time.perf_counter(); ref = cpu_pass(); times.append(time.perf_counter() - t0)
        cpu_rays = n_cpu / float(np.median(times))
        rgb_g, depth_g, acc_g = [t[:n_cpu].cpu() for t in out[:3]]
        tgt_c, occ_c = w["img"].reshape(-1, 3)[:n_cpu], w["mask"].reshape(-1, 1)[:n_cpu]
        fg = occ_c.clone(); fg[occ_c < 0] = 0
        ps = lambda rgb: float(-10 * torch.log10(((rgb - tgt_c) ** 2 * fg).sum() / (fg.sum() + 1e-9)))
        result["cpu_baseline"] = {"value": cpu_rays, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
                                  "sample": f"{n_cpu} of the 4096 rays x 64 samples, forward, oracle/supnerf_oracle.py "
                                            f"(PyTorch {torch.__version__} CPU), median of {len(times)} passes"}
        result["parity"] = {"psnr_delta_db": abs(ps(rgb_g) - ps(ref[0])), "depth_l1_mean_m": float((depth_g - ref[1]).abs().mean()),
                            "rgb_max_abs": float((rgb_g - ref[0]).abs().max()), "acc_max_abs": float((acc_g - ref[2]).abs().max()),
                            "bound": "north_star: PSNR delta <= 0.01 dB, depth L1 <= 1e-4"}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
