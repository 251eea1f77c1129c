"""The N>1 path on CPU: two ranks over gloo (torch.distributed.run, 127.0.0.1), each aligning its
shard of the job table with the emulated kernels; rank 0 gathers and compares with the unsharded run.
No data-path collective exists in the product (jobs are independent) -- this covers the sharding,
the rank/launch plumbing bench.py relies on, and the host-side gather."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_gloo_shard_and_gather(emu, tmp_path):
    out = tmp_path / "result.txt"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "tests", "dist_worker.py"), str(out)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert out.read_text().startswith("OK 14")
