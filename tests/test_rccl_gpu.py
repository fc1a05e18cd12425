"""GPU: RCCL sanity on ONE GPU.  north_star's exchange step is "an RCCL all-gather of logits over xGMI"; the multi-GPU
node is the driver's to launch, so this proves what one box can: in a FRESH child process (no GPU call before the
process group exists) `init_process_group("nccl", world_size=1, device_id=...)` succeeds -- RCCL loads and a
communicator is built -- and both collective branches of `gather_logits` (all_gather_into_tensor for equal shards,
the size exchange + list all_gather otherwise) run on the device through it and return the rank's own rows."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29631")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from quantize_amd import dist as qdist
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)     # first GPU use of this process
    try:
        assert dist.get_backend() == "nccl"
        g = torch.Generator(device=dev); g.manual_seed(3)
        logits = torch.randn(256, 1000, generator=g, device=dev)
        a = qdist.gather_logits(logits, equal_shards=True, force_collective=True)      # all_gather_into_tensor (ncclAllGather)
        b = qdist.gather_logits(logits, force_collective=True)                          # size exchange + all_gather_into_tensor
        torch.cuda.synchronize()
        assert a.shape == (256, 1000) and torch.equal(a, logits) and torch.equal(b, logits)
        assert a.data_ptr() != logits.data_ptr()                                        # a real output buffer, not the early return
        assert torch.equal(qdist.top1(a), logits.argmax(1))
        print("RCCL_OK")
    finally:
        dist.destroy_process_group()
""") % (REPO,)


def test_rccl_all_gather_on_one_gpu():
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, "-c", CHILD], cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "RCCL_OK" in r.stdout, r.stdout
