"""bench.py / __graft_entry__.py contract checks that need no GPU: the scripts parse, expose the documented flags, and
the JSON line carries every field the driver reads (a typo here only shows on the GPU box otherwise)."""
import ast
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def test_bench_parses_and_lists_its_flags():
    src = (ROOT / "bench.py").read_text()
    ast.parse(src)
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    for flag in ("--gpus", "--steps", "--warmup", "--mode", "--workload", "--extra-table-rows"):
        assert flag in out.stdout
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"steps"', '"warmup"', '"ms_per_step"', '"higher_is_better"',
                '"scaling"', '"vs_baseline"', '"dtype"', '"data"', '"config"', '"roofline"', '"cpu_baseline"'):
        assert key in src, key


def test_no_statement_hides_behind_a_comment():
    """The failure mode that once broke the default run: an edit leaving code after a `#` on the same line."""
    for name in ("bench.py", "__graft_entry__.py"):
        for i, line in enumerate((ROOT / name).read_text().splitlines(), 1):
            code, _, comment = line.partition("#")
            if comment and any(tok in comment for tok in (".cpu().numpy()", "].cpu()", ")[:")):
                raise AssertionError(f"{name}:{i}: code after a comment marker: {line.strip()[:120]}")


def test_graft_entry_exposes_build_and_smoke():
    tree = ast.parse((ROOT / "__graft_entry__.py").read_text())
    names = {n.name for n in tree.body if isinstance(n, ast.FunctionDef)}
    assert {"build", "smoke"} <= names


def _bench_module():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_under_test", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gpus_flag_becomes_ranks_or_fails_loudly():
    """`--gpus N` is never a no-op (round-3 review): without a launcher the script re-executes itself under
    torch.distributed.run with N ranks; with a launcher WORLD_SIZE must equal N; fewer GPUs than ranks is an error."""
    import pytest

    b = _bench_module()
    assert b.resolve_launch(1, {}, 1, []) == ("run", 1)
    assert b.resolve_launch(1, {}, 0, []) == ("run", 1)  # N = 1 needs no probing here: torch.cuda.set_device fails by itself
    act, cmd = b.resolve_launch(4, {}, 8, ["--gpus", "4", "--steps", "7"], free_port=lambda: 12345)
    assert act == "spawn"
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "12345"
    assert cmd[-5] == str(ROOT / "bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "7"]
    # launched by the driver's torch.distributed.run: this process is one of the ranks
    assert b.resolve_launch(8, {"WORLD_SIZE": "8", "RANK": "3", "LOCAL_RANK": "3"}, 8, []) == ("run", 8)
    with pytest.raises(SystemExit) as e:   # 1-GPU box, `python bench.py --gpus 2`
        b.resolve_launch(2, {}, 1, [])
    assert "only 1 visible GPU" in str(e.value)
    with pytest.raises(SystemExit) as e:   # launcher and flag disagree: the line would carry the wrong n_gpus
        b.resolve_launch(1, {"WORLD_SIZE": "2"}, 2, [])
    assert "WORLD_SIZE=2" in str(e.value)
    with pytest.raises(SystemExit):
        b.resolve_launch(2, {"WORLD_SIZE": "2"}, 1, [])
    with pytest.raises(SystemExit):
        b.resolve_launch(0, {}, 1, [])


def test_gpus_2_on_a_box_without_two_gpus_exits_nonzero():
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env={k: v for k, v in __import__("os").environ.items()
                                                                          if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    import torch

    if torch.cuda.device_count() < 2:
        assert out.returncode != 0 and "visible GPU" in out.stderr and '"n_gpus"' not in out.stdout
