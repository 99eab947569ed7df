"""The driver runs `pytest tests -x -q -m gpu` and stops at the first failure: every oracle comparison must be collected before the
tests about throughput plumbing, shared-GPU shapes and misuse (one such test, collected first, voided a round's
parity record).  CPU: collection only."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parity_files_are_collected_before_plumbing_tests():
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "--collect-only", "-q", "-m", "gpu"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, cwd=ROOT)
    ids = [ln for ln in r.stdout.splitlines() if "::" in ln]
    assert len(ids) >= 150, r.stdout[-2000:]
    files = []
    for i in ids:
        f = os.path.basename(i.split("::")[0])
        if not files or files[-1] != f:
            files.append(f)
    assert len(files) == len(set(files)), files                 # every file in one block
    assert files[0] == "test_gpu_ops.py" and files[-1] == "test_gpu_bench_ranks.py", files
    parity = ["test_gpu_ops.py", "test_gpu_ops_sweep.py", "test_gpu_model.py", "test_gpu_configs_c4_c5.py", "test_gpu_synthesize.py",
              "test_gpu_pipeline.py", "test_gpu_properties.py", "test_gpu_edge_cases.py", "test_gpu_decode_modes.py"]
    plumbing = ["test_gpu_misuse.py", "test_gpu_second_client.py", "test_gpu_bench_ranks.py"]
    assert max(files.index(f) for f in parity) < min(files.index(f) for f in plumbing), files
