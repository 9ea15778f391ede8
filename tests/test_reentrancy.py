"""Re-entrancy of the drop-in (SURVEY 8b "ownership / threading"): under ``nn.DataParallel`` the decoder and the render functions are
entered concurrently from one Python thread per GPU (src/trainer_unified_nuscenes.py:227-229), each on its device's current stream.
The C side holds no state (thread-local error string, no allocation, the stream is an argument); these tests enter the PYTHON side
from two threads on two streams and demand the serial run's bits."""
import copy
import threading

import numpy as np
import pytest
import torch

from oracle import supnerf_oracle as O

pytestmark = pytest.mark.gpu
ITERS = 20


def _work(A, model, dev, ob, img, mask, sc0, tc0, jit, stream, out, key, precision_check=None):
    """20 x (render_rays_v2 + loss + backward + a plain SGD step on codes and pose) for one object, on ``stream``."""
    try:
        with torch.cuda.device(dev), torch.cuda.stream(stream):
            sc, tc = sc0.to(dev).requires_grad_(), tc0.to(dev).requires_grad_()
            pose = ob["cam_pose"].to(dev).requires_grad_()
            rows = []
            for it in range(ITERS):
                with A.utils.jitter_override(jit[it]):                  # thread-local injection: the other thread sees its own
                    rgb, depth, acc, tgt, occ = A.utils.render_rays_v2(model, dev, img, mask, pose, ob["obj_diag"], ob["K"], ob["roi"], 64, sc, tc,
                                                                       1, 0, im_sz=16)
                loss = O.optimise_losses(rgb, acc, tgt, occ, 0.1)[0]
                g_sc, g_tc, g_pose = torch.autograd.grad(loss, [sc, tc, pose])
                with torch.no_grad():
                    sc -= 0.05 * g_sc; tc -= 0.05 * g_tc; pose -= 1e-3 * g_pose
                rows.append(torch.cat([rgb.detach().reshape(-1), depth.detach(), acc.detach(), g_sc.reshape(-1), g_pose.reshape(-1)]))
            stream.synchronize()
            out[key] = torch.stack(rows).cpu()
    except BaseException as e:          # noqa: BLE001  (surface the thread's failure in the main thread)
        out[key] = e


def _objects(n):
    obs = []
    for i in range(n):
        ob = O.synthetic_object(31 + i)
        img, mask = O.synthetic_targets(31 + i, 16)
        g = torch.Generator().manual_seed(100 + i)
        obs.append((ob, img, mask, torch.randn(1, 256, generator=g) * 0.3, torch.randn(1, 256, generator=g) * 0.3, torch.rand(ITERS, 64, generator=g)))
    return obs


def _run(A, models, devs, objs, concurrent):
    out = {}
    streams = [torch.cuda.Stream(device=d) for d in devs]
    threads = [threading.Thread(target=_work, args=(A, models[i], devs[i], *objs[i], streams[i], out, i)) for i in range(len(objs))]
    if concurrent:
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    else:
        for t in threads:
            t.start(); t.join()
    for v in out.values():
        if isinstance(v, BaseException):
            raise v
    return [out[i] for i in range(len(objs))]


def _model(A, dev, precision):
    m = A.CodeNeRF(shape_blocks=3, texture_blocks=1)
    m.load_state_dict(O.init_decoder_params())
    m.precision = precision
    return m.to(dev)


@pytest.mark.parametrize("precision", ["fp32", "auto"])
def test_two_threads_two_streams_one_module_bit_equal_to_serial(precision):
    """ONE module shared by two threads, each on its own stream, different objects: what two DataParallel workers would do to a module
    whose caches (packed weights, stacked latent layers, camera tables, resized targets) were process-global in round 3."""
    import supnerf_amd as A
    dev = torch.device("cuda:0")
    model = _model(A, dev, precision)
    objs = _objects(2)
    A.utils.clear_caches()
    serial = _run(A, [model, model], [dev, dev], objs, concurrent=False)
    A.utils.clear_caches()
    model2 = _model(A, dev, precision)                      # fresh caches on the module too
    conc = _run(A, [model2, model2], [dev, dev], objs, concurrent=True)
    for a, b in zip(serial, conc):
        assert torch.equal(a, b)
    assert not torch.equal(serial[0], serial[1])            # (the two objects really differ)


@pytest.mark.parametrize("precision", ["fp32", "auto"])
def test_replicas_in_threads_bit_equal_to_serial(precision):
    """Two replicas of the module the way ``nn.DataParallel`` makes them (``torch.nn.parallel.replicate``; ``copy.deepcopy`` if this
    torch refuses two replicas on one device), one thread and one stream each."""
    import supnerf_amd as A
    dev = torch.device("cuda:0")
    model = _model(A, dev, precision)
    objs = _objects(2)
    serial = _run(A, [model, model], [dev, dev], objs, concurrent=False)
    try:
        with torch.no_grad():
            reps = torch.nn.parallel.replicate(model, [0, 0], detach=True)
        how = "replicate"
    except Exception:                                       # noqa: BLE001
        reps, how = [copy.deepcopy(model), copy.deepcopy(model)], "deepcopy"
    assert all(isinstance(r, A.model._DecoderBase) for r in reps), how
    conc = _run(A, reps, [dev, dev], objs, concurrent=True)
    for a, b in zip(serial, conc):
        assert torch.equal(a, b), how


def test_dataparallel_wrapper_forward_matches_module():
    """``nn.DataParallel(model)(xyz, viewdir, codes...)`` -- the trainer's wrap (src/trainer_unified_nuscenes.py:227-229) -- on the GPUs
    that exist: scatter on dim 0 is object-major, exactly the decoder's batching."""
    import supnerf_amd as A
    n_dev = torch.cuda.device_count()
    dev = torch.device("cuda:0")
    model = _model(A, dev, "fp32")
    B = 2 * max(n_dev, 1)
    g = torch.Generator().manual_seed(3)
    xyz = (torch.rand(B * 32, 16, 3, generator=g) - 0.5).to(dev)
    vd = torch.nn.functional.normalize(torch.randn(B * 32, 16, 3, generator=g), dim=-1).to(dev)
    sc, tc = (torch.randn(B, 256, generator=g) * 0.3).to(dev), (torch.randn(B, 256, generator=g) * 0.3).to(dev)
    with torch.no_grad():
        s0, c0 = model(xyz, vd, sc, tc)
    dp = torch.nn.DataParallel(model, device_ids=list(range(max(n_dev, 1))))
    with torch.no_grad():
        # DataParallel scatters every positional input along dim 0: (B*32 rays) and (B codes) split into the same objects
        s1, c1 = dp(xyz, vd, sc, tc)
    assert torch.equal(s0, s1.to(dev)) and torch.equal(c0, c1.to(dev))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_two_devices_two_threads_bit_equal_to_serial():
    """The cuda:1 twin: a replica per device, a thread per device (skipped on the one-GPU test box; runs on a multi-GPU lease)."""
    import supnerf_amd as A
    devs = [torch.device("cuda:0"), torch.device("cuda:1")]
    models = [_model(A, d, "auto") for d in devs]
    objs = _objects(2)
    serial = _run(A, models, devs, objs, concurrent=False)
    conc = _run(A, models, devs, objs, concurrent=True)
    for a, b in zip(serial, conc):
        assert torch.equal(a, b)
    ref0 = _run(A, [models[0]], [devs[0]], [objs[1]], concurrent=False)[0]       # the same object on the other device: same bits
    assert torch.equal(ref0, serial[1])


@pytest.mark.gpu
def test_raw_stream_follows_the_stream_context():
    """Every cache key and every launch takes the current stream through ops.raw_stream (the private raw getter, no Python Stream object per
    call): it must be the stream torch itself would launch on, inside a torch.cuda.stream() context and on a non-default device index alike."""
    import supnerf_amd as A
    dev = torch.device("cuda:0")
    assert A.ops.raw_stream(dev) == torch.cuda.current_stream(dev).cuda_stream
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        assert A.ops.raw_stream(dev) == side.cuda_stream
        assert A.ops.raw_stream("cuda") == side.cuda_stream
        assert A.utils._stream_key(dev) == (str(dev), side.cuda_stream)
    assert A.ops.raw_stream(dev) == torch.cuda.current_stream(dev).cuda_stream
