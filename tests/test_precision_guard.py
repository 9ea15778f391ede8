"""``precision = "auto"`` is a recorded, range-checked decision (``model.last_precision``): the split kernels' forward carries fp16
pieces (activations clamp at +-65504, weights are packed clamped), so the library itself must notice a decoder that leaves that range
-- no real checkpoint exists offline -- and must say which arithmetic a call ran in."""
import warnings

import numpy as np
import pytest
import torch

from oracle import supnerf_oracle as O

pytestmark = pytest.mark.gpu


def _points(dev, P=4096, seed=3):
    g = torch.Generator().manual_seed(seed)
    xyz = (torch.rand(P // 64, 64, 3, generator=g) - 0.5).to(dev)
    vd = torch.nn.functional.normalize(torch.randn(P // 64, 64, 3, generator=g), dim=-1).to(dev)
    sc, tc = (torch.randn(1, 256, generator=g) * 0.3).to(dev), (torch.randn(1, 256, generator=g) * 0.3).to(dev)
    return xyz, vd, sc, tc


def _scaled(amd, dev, oracle_params, scale, weight_scale=None):
    params = {k: v.clone() for k, v in oracle_params.items()}
    params["encoding_xyz.0.weight"] *= scale                       # first-layer activations of order ``scale`` ...
    params["shape_layer_1.0.weight"] /= scale                      # ... brought back by the next layer
    if weight_scale is not None:
        params["rgb.0.weight"][0, 0] = weight_scale                # one weight beyond the fp16 range
    m = amd.CodeNeRF(3, 1); m.load_state_dict(params)
    return m.to(dev)


def _count_launches(fn):
    """How many times the decoder / render forward entry points were called inside fn (the probes are extra calls of these)."""
    import supnerf_amd as A
    calls = {"n": 0}
    real_dec, real_ren = A.ops.decoder_fwd, A.ops.render_fwd

    def dec(*a, **k):
        calls["n"] += 1
        return real_dec(*a, **k)

    def ren(*a, **k):
        calls["n"] += 1
        return real_ren(*a, **k)

    A.ops.decoder_fwd, A.ops.render_fwd = dec, ren
    try:
        fn()
    finally:
        A.ops.decoder_fwd, A.ops.render_fwd = real_dec, real_ren
    return calls["n"]


def test_auto_on_a_healthy_decoder_runs_the_split_kernels_and_says_so(oracle_params):
    import supnerf_amd as A
    dev = torch.device("cuda:0")
    m = _scaled(A, dev, oracle_params, 1.0)
    xyz, vd, sc, tc = _points(dev)
    assert m.last_precision is None
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        with torch.no_grad():
            n_first = _count_launches(lambda: m(xyz, vd, sc, tc))
            n_steady = _count_launches(lambda: m(xyz, vd, sc, tc))
    assert n_first == 3 and n_steady == 1                          # one probe pair on the first call of a weight version, then nothing
    rec = m.last_precision
    assert rec["requested"] == "auto" and rec["forward"] == "bf16x3" and rec["backward"] == "bf16x3" and "range guard passed" in rec["reason"]
    # an in-place weight change is a new weight version: checked again, once
    with torch.no_grad():
        m.encoding_shape.weight.mul_(1.0)
        assert _count_launches(lambda: m(xyz, vd, sc, tc)) == 3 and _count_launches(lambda: m(xyz, vd, sc, tc)) == 1
    # explicit requests are recorded as such and never probed
    m.precision = "fp32"
    with torch.no_grad():
        assert _count_launches(lambda: m(xyz, vd, sc, tc)) == 1
    assert m.last_precision == {"requested": "fp32", "forward": "fp32", "backward": "fp32", "reason": "requested"}
    m.precision = ("fp32", "bf16x3")
    with torch.no_grad():
        m(xyz, vd, sc, tc)
    assert (m.last_precision["forward"], m.last_precision["backward"]) == ("fp32", "bf16x3")


@pytest.mark.parametrize("how", ["activations", "weight"])
def test_auto_downgrades_a_decoder_beyond_the_fp16_range_to_exact_fp32(oracle_params, how):
    """Activations (first layer scaled by 3e5) or one weight (1e5) beyond +-65504: ``auto`` notices on the first call, warns once, and
    every output is the exact-fp32 kernels' -- bit for bit -- from that call on, with no extra launch in the steady state."""
    import supnerf_amd as A
    dev = torch.device("cuda:0")
    m = _scaled(A, dev, oracle_params, 3.0e5) if how == "activations" else _scaled(A, dev, oracle_params, 1.0, weight_scale=1.0e5)
    exact = _scaled(A, dev, oracle_params, 3.0e5) if how == "activations" else _scaled(A, dev, oracle_params, 1.0, weight_scale=1.0e5)
    exact.precision = "fp32"
    xyz, vd, sc, tc = _points(dev)
    with torch.no_grad():
        want = exact(xyz, vd, sc, tc)
        with pytest.warns(RuntimeWarning, match="exact fp32 kernels"):
            got = m(xyz, vd, sc, tc)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    rec = m.last_precision
    assert rec["requested"] == "auto" and rec["forward"] == "fp32" and rec["backward"] == "fp32" and "range guard" in rec["reason"]
    detail = m._guard["detail"]
    assert (detail["values_out_of_tolerance"] > 0) if how == "activations" else (detail["weights_beyond_fp16_range"] == 1)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                              # said once
        with torch.no_grad():
            assert _count_launches(lambda: m(xyz, vd, sc, tc)) == 1
            got2 = m(xyz, vd, sc, tc)
    assert torch.equal(got2[1], want[1])
    # the fused render takes the same decision (same module, same weight version: no new probe)
    ob = O.synthetic_object(11)
    img, mask = O.synthetic_targets(11, 16)
    with torch.no_grad():
        n = _count_launches(lambda: A.utils.render_rays_v2(m, dev, img, mask, ob["cam_pose"].to(dev), ob["obj_diag"], ob["K"], ob["roi"], 64, sc, tc, 1, 0,
                                                           im_sz=16))
    assert n == 1 and m.last_precision["forward"] == "fp32"


def test_auto_records_the_shape_fallback(oracle_params):
    """Ragged objects (35 points per object, no latent gradient wanted): the split kernels do not take the shape; ``auto`` = fp32, recorded."""
    import supnerf_amd as A
    dev = torch.device("cuda:0")
    m = _scaled(A, dev, oracle_params, 1.0)
    g = torch.Generator().manual_seed(1)
    xyz = (torch.rand(10, 7, 3, generator=g) - 0.5).to(dev)
    vd = torch.nn.functional.normalize(torch.randn(10, 7, 3, generator=g), dim=-1).to(dev)
    sc, tc = (torch.randn(2, 256, generator=g) * 0.3).to(dev), (torch.randn(2, 256, generator=g) * 0.3).to(dev)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m(xyz, vd, sc, tc)
    assert m.last_precision["forward"] == "fp32" and m.last_precision["reason"].startswith("shape")
    # with the codes' gradient wanted the operators pad every object to whole tiles: the split kernels run, and the record says so
    sc.requires_grad_()
    m(xyz, vd, sc, tc)[1].sum().backward()
    assert m.last_precision["forward"] == "bf16x3" and sc.grad is not None


def test_training_mode_checks_on_a_sparse_schedule(oracle_params):
    """Training changes the weights every step: the guard re-checks on steps 1, 16, 256 and every 1024th, not every step."""
    import supnerf_amd as A
    dev = torch.device("cuda:0")
    m = _scaled(A, dev, oracle_params, 1.0)
    m.train_decoder_weights = True
    xyz, vd, sc, tc = _points(dev, P=2048)
    probes = []
    for step in range(1, 18):
        n = _count_launches(lambda: m(xyz, vd, sc, tc)[1].sum().backward())
        probes.append(n - 1)
        with torch.no_grad():
            for p in m.parameters():
                p.add_(1e-6)
                p.grad = None
    assert probes == [2] + [0] * 14 + [2, 0], probes
    assert m.last_precision["forward"] == "bf16x3"
