"""The N>1 path on CPU: two ranks over gloo (torch.distributed.run, 127.0.0.1).

1. bench.py --sharded --backend emu: the real bench driver -- every rank generating and aligning the chunk of the config-4
   table it owns (emulated kernels in place of the GPU), rank 0 gathering records and CIGARs with dist.gather of uint8 tensors
   inside the timed loop -- must print one JSON line whose job total scales with the number of ranks.
2. a bespoke worker that compares the gathered records with the unsharded run, job for job.
No data-path collective exists in the product (jobs are independent)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(args, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port)] + args
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)


def test_two_rank_gloo_bench_sharded(emu, hip_lib):
    p = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--sharded", "--backend", "emu"], 29519)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, p.stdout[-2000:]
    rec = json.loads(line[0])
    assert rec["mode"] == "sharded" and rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["config"]["jobs_total"] == 2 * 48 and rec["config"]["jobs_per_gpu"] == 48 and rec["sum_score1"] > 0      # 48 jobs per rank (emu rehearsal)
    assert "f16" in rec["dtype"]


def test_two_rank_gloo_shard_and_gather(emu, tmp_path):
    out = tmp_path / "result.txt"
    p = _torchrun([os.path.join(ROOT, "tests", "dist_worker.py"), str(out)], 29517)
    assert p.returncode == 0, p.stderr[-2000:]
    assert out.read_text().startswith("OK 14")


def test_single_process_bench_drives_every_gpu_from_its_own_host_thread(emu, hip_lib):
    """plain `python bench.py --gpus N` (no torchrun): one host thread per GPU issues that GPU's steps; rehearsed on the emulator.  The
    line names the threads and the measured enqueue cost per GPU, and shard 0's results are those of the one-GPU run."""
    def run(n):
        env = dict(os.environ, PYTHONPATH=ROOT)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--backend", "emu", "--steps", "2", "--warmup", "1",
                            "--reads-per-gpu", "48", "--no-cpu-baseline", "--no-subconfigs", "--streams", "2"], env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(line) == 1, p.stdout[-2000:]
        return json.loads(line[0])
    one, three = run(1), run(3)
    assert three["n_gpus"] == 3 and three["host_threads"] == 3 and len(three["host_enqueue_ms_per_step"]) == 3
    assert one["host_threads"] == 1 and three["sum_score1"] == one["sum_score1"] > 0
