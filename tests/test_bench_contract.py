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
