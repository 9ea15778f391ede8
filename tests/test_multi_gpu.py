"""RCCL (backend "nccl" on ROCm) tests of the two places the path talks between GPUs: the all_gather of the per-object metric rows
behind object sharding and the one all-reduce of the training step's gradient bucket (8 f2).  They need two GPUs and are
skipped on a one-GPU box; the same logic runs on gloo / CPU in tests/test_host_logic.py.  One process per GPU, rendezvous on 127.0.0.1."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs_two = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL)")

_PRELUDE = """
import os, sys
sys.path.insert(0, {root!r})
rank = int(sys.argv[1])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="{port}", RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                  HSA_ENABLE_IPC_MODE_LEGACY="0")
import torch
import torch.distributed as dist
torch.cuda.set_device(rank)
dev = torch.device("cuda", rank)
dist.init_process_group("nccl", rank=rank, world_size=2, device_id=dev)
solo = [dist.new_group([0]), dist.new_group([1])][rank]      # a group of one (every rank creates both, as new_group requires)
import supnerf_amd as A
from oracle import supnerf_oracle as O
"""

_SHARD_WORKER = _PRELUDE + """
D = A.driver
model = A.CodeNeRF(3, 1); model.load_state_dict(O.init_decoder_params()); model = model.to(dev)
model.precision = "fp32"
hp = D.load_hpams(); hp["render_im_sz"] = 16; hp["optimize"]["num_opts"] = 4
sharded = D.optimize_objects(model, dev, 5, hp, rank=rank, world_size=2, seed=1, batch=64)      # ranks own objects [0,1,2] and [3,4]
whole = D.optimize_objects(model, dev, 5, hp, rank=0, world_size=1, seed=1, batch=64, group=solo)
assert sharded.shape == (5, 16) and bool(torch.isfinite(sharded).all())
d = float((sharded - whole).abs().max())
assert d < 1e-3, d                         # an object's numbers do not depend on which rank or batch it ran in
both = [torch.empty_like(sharded) for _ in range(2)]
dist.all_gather(both, sharded)
assert torch.equal(both[0], both[1])       # every rank ends with the same table
dist.destroy_process_group()
print("ok", rank, d)
"""

_TRAIN_WORKER = _PRELUDE + """
T = A.trainer
hp = dict(lr_schedule=[dict(lr=1e-4, interval=100), dict(lr=1e-3, interval=100)])
def world():
    m = A.CodeNeRF(3, 1); m.load_state_dict(O.init_decoder_params()); m = m.to(dev); m.train_decoder_weights = True
    codes = T.CodeTables(2000, 256, seed=3).to(dev)
    return m, codes, list(m.parameters()) + list(codes.parameters())
g = torch.Generator().manual_seed(11)
B, n, S = 2, 32, 64
full = dict(code_idx=torch.tensor([1742, 9]), xyz=torch.rand(B, n, S, 3, generator=g) - 0.5,
            viewdir=torch.nn.functional.normalize(torch.randn(B, n, 1, 3, generator=g), dim=-1).repeat(1, 1, S, 1),
            z_vals=torch.sort(torch.rand(B, S, generator=g) * 4 + 9, dim=-1)[0], rgb_tgt=torch.rand(B, n, 3, generator=g),
            occ_pixels=(torch.randint(0, 3, (B, n, 1), generator=g) - 1).float())
full = {{k: v.to(dev) for k, v in full.items()}}
mine = {{k: v[rank:rank + 1] for k, v in full.items()}}
m, codes, params = world()
bucket = T.GradBucket(params, row_sparse=list(codes.parameters()))      # the tables travel as touched rows, not in the all-reduced bucket
opt = T.make_optimizer(m, codes, hp)
for it in range(2):                                   # two ranks, one object each
    T.train_step(m, codes, opt, bucket, mine, 0.1)
m1, codes1, params1 = world()                         # the same two iterations on the whole batch, no exchange (a group of one)
bucket1 = T.GradBucket(params1, group=solo, row_sparse=list(codes1.parameters()))
opt1 = T.make_optimizer(m1, codes1, hp)
for it in range(2):
    T.train_step(m1, codes1, opt1, bucket1, full, 0.1)
worst = max(float((a - b).abs().max()) for a, b in zip(params, params1))
assert worst < 5e-5, worst
flat = torch.cat([p.detach().flatten() for p in params])
other = [torch.empty_like(flat) for _ in range(2)]
dist.all_gather(other, flat)
assert torch.equal(other[0], other[1])                # replicas stay bit-identical
dist.destroy_process_group()
print("ok", rank, worst)
"""


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run_two(tmp_path, text):
    script = tmp_path / "w.py"
    script.write_text(text.format(root=ROOT, port=_free_port()))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = ["", ""]
    try:
        for i, p in enumerate(procs):
            outs[i] = p.communicate(timeout=600)[0]
    finally:                                   # a rank that hangs (rendezvous, a collective) must not keep a GPU after the test
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs), outs


@needs_two
def test_sharded_optimise_equals_unsharded_rccl(tmp_path):
    """driver.optimize_objects over 2 ranks (object shards + all_gather over RCCL) == the same 5 objects on one rank."""
    _run_two(tmp_path, _SHARD_WORKER)


@needs_two
def test_training_step_two_ranks_rccl(tmp_path):
    """trainer.train_step: per-rank batch slice + one bucket all-reduce over RCCL == the full batch on one rank."""
    _run_two(tmp_path, _TRAIN_WORKER)
