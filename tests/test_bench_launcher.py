"""`python bench.py --gpus N` must start N ranks itself or fail loudly — never print a 1-GPU line for N GPUs
(SURVEY.md §8(e)).  CPU-only: the launcher never touches a GPU, the ranks here are a stub script."""
import importlib.util
import json
import os
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_mod_launcher", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_rank_environments(bench):
    envs = bench.rank_environments(4, 23456, base_env={"PATH": "/usr/bin"})
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "23456" for e in envs)
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)


def test_more_gpus_than_devices_fails_loudly(bench, monkeypatch, capsys):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("VITVS_BENCH_SHARE_GPU", raising=False)
    monkeypatch.setattr(bench, "visible_devices", lambda: 1)
    with pytest.raises(SystemExit) as exc:
        bench.main(["--gpus", "8", "--steps", "2", "--warmup", "1"])
    assert exc.value.code == 2
    out = capsys.readouterr()
    assert "{" not in out.out and "only 1 HIP device" in out.err


def test_world_size_mismatch_is_an_error(bench, monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as exc:
        bench.main(["--gpus", "4"])
    assert "WORLD_SIZE=2" in str(exc.value.code)


def _stub(tmp_path, body):
    path = tmp_path / "rank_stub.py"
    path.write_text(textwrap.dedent(body))
    return str(path)


def test_launcher_starts_n_ranks_and_relays_rank0(bench, monkeypatch, tmp_path, capsys):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench, "visible_devices", lambda: 4)
    marks = tmp_path / "marks"
    marks.mkdir()
    script = _stub(tmp_path, f"""
        import json, os, sys
        open(os.path.join({str(marks)!r}, os.environ["RANK"]), "w").write(os.environ["LOCAL_RANK"] + " " + " ".join(sys.argv[1:]))
        if os.environ["RANK"] == "0":
            print("noise before the line")
            print(json.dumps(dict(metric="servo_updates_per_sec", n_gpus=int(os.environ["WORLD_SIZE"]), value=1.0)))
    """)
    args = bench.parse_args(["--gpus", "4", "--steps", "3"])
    rc = bench.launch_ranks(args, ["--gpus", "4", "--steps", "3"], script=script)
    assert rc == 0
    assert sorted(os.listdir(marks)) == ["0", "1", "2", "3"]
    assert (marks / "2").read_text() == "2 --gpus 4 --steps 3"
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert line["n_gpus"] == 4


def test_launcher_reports_failing_rank_and_mislabelled_line(bench, monkeypatch, tmp_path, capsys):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench, "visible_devices", lambda: 2)
    bad = _stub(tmp_path, """
        import os, sys
        sys.exit(3 if os.environ["RANK"] == "1" else 0)
    """)
    assert bench.launch_ranks(bench.parse_args(["--gpus", "2"]), ["--gpus", "2"], script=bad) == 1
    lying = _stub(tmp_path, """
        import json, os
        if os.environ["RANK"] == "0":
            print(json.dumps(dict(n_gpus=1, value=1.0)))
    """)
    assert bench.launch_ranks(bench.parse_args(["--gpus", "2"]), ["--gpus", "2"], script=lying) == 1
    assert "n_gpus=1" in capsys.readouterr().err
