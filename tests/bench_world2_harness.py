"""Run bench.py as ONE RANK of a job whose ranks share GPU 0 over gloo (launched by tests/test_gpu_bench_world2.py under
torch.distributed.run with MH_BENCH_SHARED_GPU=1): stages gloo's missing device all-to-all through the host, then hands over to
bench.py's own main().  Test infrastructure."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from test_gpu_world2 import _patch_gloo_for_device_tensors  # noqa: E402

_patch_gloo_for_device_tensors()
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
