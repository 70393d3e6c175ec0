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


def _fat_result():
    """A result object of the size round 5 printed (18 secondaries, long-form rooflines) with non-finite values sprinkled in."""
    long = "x" * 400
    sec = {f"sec{i}": {"metric": long, "value": 1.0 + i, "ms_per_step": float("nan") if i == 3 else 2.0, "kernels_ms": {f"k{j}": j for j in range(40)}}
           for i in range(18)}
    sec["twotower_train_b64k"] = {"value": 3.1e6, "ms_per_step": 21.0, "steps": 8, "kernels_ms": {}}
    sec["broken"] = {"error": "RuntimeError: boom"}
    rl = {"kernel": long, "op": "embedding_bwd", "bound": "hbm", "achieved": 5028.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.6285, "traffic": 1.13e9,
          "traffic_source": long, "algorithmic_bytes_per_launch": 1.04e9, "avg_launch_ms": 0.2077, "definition": long,
          "whole_update": {"frac": 0.39, "avg_launch_ms": 0.335, "definition": long}}
    return {"metric": "samples/sec at batch 64K (DLRM)", "value": 6.7e7, "unit": "samples/s", "n_gpus": 1, "steps": 20, "warmup": 5,
            "ms_per_step": 0.97, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": long, "global_batch": 65536, "per_gpu_batch": 65536, "mode": "train", "optimizer": "adagrad",
                       "launch": "eager + side streams", "launch_probe": {"a": float("inf")}, "parallelism": "dp1", "input_staging": long},
            "roofline": rl, "roofline_fused_fwd": {"frac": 0.7, "note": long}, "mfma": {"linear_415x128": {"tflops": 91.3}},
            "kernels_ms": {f"k{j}": j * 0.1 for j in range(40)}, "parity_notes": {"pinned": long, "unpinned": long},
            "cpu_baseline": {"value": 63245.9, "unit": "samples/s", "cores": 16, "kind": "port", "sample": long},
            "secondary": sec, "max_abs_err_vs_oracle": 5.96e-8, "sustained": {"value": 6.8e7, "ms_per_step": 0.96, "seconds": 5.0, "steps": 5000}}


def test_last_stdout_line_is_a_small_strict_json_headline(tmp_path, monkeypatch):
    """Round-5 review, item 1: the driver could not parse a 21.6 KB single object.  The LAST line must be < 4 KB of strict JSON with
    the contract keys + roofline + cpu_baseline + the TwoTower-64K half of the metric; every other line strict JSON too."""
    import io
    import json

    b = _bench_module()
    monkeypatch.setattr(b, "ROOT", tmp_path)
    buf = io.StringIO()
    b.emit(_fat_result(), buf)
    lines = buf.getvalue().splitlines()
    strict = lambda l: json.loads(l, parse_constant=lambda c: (_ for _ in ()).throw(AssertionError(f"non-finite {c}")))
    objs = [strict(l) for l in lines]
    last = lines[-1]
    assert len(last) < 4096, len(last)
    h = objs[-1]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in h, k
    assert set(h["config"]) == {"workload", "global_batch", "per_gpu_batch", "mode", "optimizer", "launch", "parallelism"}
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms", "whole_update_frac"):
        assert k in h["roofline"], k
    assert h["roofline"]["frac"] == 0.6285 and h["roofline"]["whole_update_frac"] == 0.39
    assert {"value", "unit", "cores", "kind"} <= set(h["cpu_baseline"]) and h["cpu_baseline"]["value"] == 63245.9
    assert h["twotower_b64k"]["value"] == 3.1e6 and h["twotower_b64k"]["ms_per_step"] == 21.0
    assert h["max_abs_err_vs_oracle"] == 5.96e-8 and h["secondary_lines"]["errors"] == ["broken"]
    # the secondaries ride on their own lines, before the headline; the full object is on disk
    names = [o["secondary"] for o in objs[:-1] if "secondary" in o]
    assert len(names) == 20 and "twotower_train_b64k" in names
    assert objs[names.index("sec3")]["ms_per_step"] is None  # NaN -> null
    full = json.loads((tmp_path / h["full_object"]).read_text())
    assert full["config"]["launch_probe"]["a"] is None and len(full["secondary"]) == 20
    # and the inverse used by the tests / tools
    r = b.reassemble(buf.getvalue())
    assert r["secondary"]["sec5"]["value"] == 6.0 and r["roofline"]["whole_update"]["frac"] == 0.39 and r["value"] == 6.7e7


def test_headline_survives_missing_objects():
    import json

    b = _bench_module()
    h = b.headline({"metric": "m", "value": float("nan"), "n_gpus": 2, "roofline": None, "cpu_baseline": None,
                    "exchange": {"communicator": "RCCL", "groups": [{"dedup": True, "window_slots": 5, "local_rows": 7, "features": ["a"] * 100}]}})
    s = json.dumps(h, allow_nan=False)
    assert h["value"] is None and h["roofline"] is None and h["cpu_baseline"] is None and h["exchange"]["groups"] == [{"dedup": True, "window_slots": 5, "local_rows": 7}]
    assert len(s) < 4096
