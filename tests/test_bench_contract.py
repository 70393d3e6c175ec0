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


def test_multi_gpu_secondaries_run_in_one_order_and_contain_their_failures(monkeypatch):
    """The N > 1 secondaries are collectives: every rank must walk the SAME list in the SAME order, and a (rank-symmetric)
    failure of one entry must cost that entry only."""
    import argparse

    b = _bench_module()
    calls = []
    monkeypatch.setattr(b, "run_c4_sharded", lambda args, device, tm, rank, rows: calls.append(("c4", rows)) or {"value": 1.0, "ms_per_step": 1.0})

    def tt(args, device, tm, steps, warmup, sustain, batch=None):
        calls.append(("tt", batch))
        if batch == 65536:
            raise RuntimeError("boom")
        return {"value": 2.0, "ms_per_step": 2.0, "metric": "m", "junk": 1}

    monkeypatch.setattr(b, "run_twotower", tt)
    monkeypatch.setattr(b, "run_dcn", lambda args, device, tm: calls.append(("dcn", args.steps)) or {"value": 3.0, "ms_per_step": 3.0})
    monkeypatch.setattr(b.torch.cuda, "empty_cache", lambda: None)
    args = argparse.Namespace(c4_rows=100, tt_batches="32768,65536", steps=20, warmup=5, sustain=5.0, batches=8, mode="train")
    tm = argparse.Namespace(world=4)
    sec = b.run_multi_gpu_secondaries(args, None, tm, 0, {})
    assert calls == [("c4", 100), ("tt", 32768), ("tt", 65536), ("dcn", 6)]
    assert list(sec) == ["c4", "twotower_train", "twotower_train_b64k", "dcn_train"]
    assert sec["twotower_train_b64k"]["error"].startswith("RuntimeError") and "junk" not in sec["twotower_train"]
    assert all(sec[k]["n_gpus"] == 4 for k in ("c4", "twotower_train", "dcn_train"))


def test_deadline_prints_the_line_it_has_and_ends_the_process():
    code = ("import sys, time; sys.path.insert(0, %r)\n"
            "import importlib.util as u\n"
            "sp = u.spec_from_file_location('b', %r); b = u.module_from_spec(sp); sp.loader.exec_module(b)\n"
            "with b.Deadline(0.5, 0, lambda: {'metric': 'm', 'value': 1.0}):\n"
            "    time.sleep(60)\n" % (str(ROOT), str(ROOT / "bench.py")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    import json

    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert out.returncode == 0 and line["value"] == 1.0 and "deadline" in line["secondary_aborted"]
