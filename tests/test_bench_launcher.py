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


def test_crashing_rank_ends_the_run_in_seconds_and_names_itself(bench, monkeypatch, tmp_path, capsys):
    """Rank 2 dies during start-up while the others would wait in a rendezvous for minutes: the launcher must end within
    seconds, terminate the waiting ranks, and print the dead rank with the tail of ITS stderr (ranks > 0 used to have none)."""
    import time
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench, "visible_devices", lambda: 4)
    pids = tmp_path / "pids"
    pids.mkdir()
    script = _stub(tmp_path, f"""
        import os, sys, time
        open(os.path.join({str(pids)!r}, os.environ["RANK"]), "w").write(str(os.getpid()))
        if os.environ["RANK"] == "2":
            time.sleep(0.5)
            print("hipErrorNoDevice: rank 2 found no GPU", file=sys.stderr)
            sys.exit(7)
        time.sleep(600)          # the others: stuck in init_process_group
    """)
    t0 = time.monotonic()
    rc = bench.launch_ranks(bench.parse_args(["--gpus", "4"]), ["--gpus", "4"], script=script)
    took = time.monotonic() - t0
    err = capsys.readouterr().err
    assert rc == 1 and took < 20.0
    assert "rank 2 exited with code 7" in err and "rank 2 found no GPU" in err
    for r in ("0", "1", "3"):                                   # nobody is left behind holding a GPU
        pid = int((pids / r).read_text())
        with pytest.raises(ProcessLookupError):
            os.kill(pid, 0)


def test_overall_deadline_terminates_hung_ranks(bench, monkeypatch, tmp_path, capsys):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench, "visible_devices", lambda: 2)
    script = _stub(tmp_path, """
        import time
        time.sleep(600)
    """)
    rc = bench.launch_ranks(bench.parse_args(["--gpus", "2"]), ["--gpus", "2"], script=script, deadline_s=1.0)
    assert rc == 1 and "no result after 1 s" in capsys.readouterr().err


def test_device_count_reads_the_kfd_topology_without_torch(bench, monkeypatch, tmp_path):
    """The launcher counts GPUs from /sys/class/kfd (nodes with SIMDs), narrowed by *_VISIBLE_DEVICES, and never imports
    torch or opens HIP for it."""
    topo = tmp_path / "nodes"
    for i, simd in enumerate([0, 0, 1024, 1024, 1024]):          # two CPU nodes, three GPUs
        d = topo / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\nmem_banks_count 1\n")
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    had_torch = "torch" in sys.modules
    assert bench.visible_devices(str(topo)) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.visible_devices(str(topo)) == 2
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1")
    assert bench.visible_devices(str(topo)) == 1
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_devices(str(topo)) == 0
    assert ("torch" in sys.modules) == had_torch


def test_launcher_parent_never_imports_torch():
    """`python bench.py --gpus N` (no WORLD_SIZE) must reach its refusal without torch in the parent process."""
    import subprocess
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '64']\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n    rc = e.code\n"
            "print('torch' in sys.modules, rc)") % os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "VITVS_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert r.stdout.strip().splitlines()[-1] == "False 2", (r.stdout, r.stderr)


def test_sigterm_to_the_launcher_leaves_no_rank_behind(tmp_path):
    """The ranks run in sessions of their own, so a signal to the launcher's process group does not reach them: the launcher
    itself must end them when it is told to stop (harness time limit, Ctrl-C, job cancel) — and remove nothing it still
    needs for the post-mortem.  A real parent process here: SIGTERM to it, then every rank's pid must be gone."""
    import signal
    import subprocess
    import time
    pids = tmp_path / "pids"
    pids.mkdir()
    stub = _stub(tmp_path, f"""
        import os, time
        open(os.path.join({str(pids)!r}, os.environ["RANK"]), "w").write(str(os.getpid()))
        time.sleep(600)
    """)
    driver = tmp_path / "driver.py"
    driver.write_text(textwrap.dedent(f"""
        import importlib.util, sys
        spec = importlib.util.spec_from_file_location("bench_mod_sig", {os.path.join(ROOT, "bench.py")!r})
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        bench.visible_devices = lambda: 3
        sys.exit(bench.launch_ranks(bench.parse_args(["--gpus", "3"]), ["--gpus", "3"], script={stub!r}))
    """))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "VITVS_BENCH_SHARE_GPU")}
    parent = subprocess.Popen([sys.executable, str(driver)], env=env, stderr=subprocess.PIPE, text=True)
    t_end = time.monotonic() + 60.0
    while len(os.listdir(pids)) < 3 and time.monotonic() < t_end:
        time.sleep(0.1)
    assert len(os.listdir(pids)) == 3, "the ranks never started"
    time.sleep(0.3)                                             # (a pid file exists a moment before it has its content)
    rank_pids = [int((pids / r).read_text()) for r in ("0", "1", "2")]
    parent.send_signal(signal.SIGTERM)
    _, err = parent.communicate(timeout=30)
    assert parent.returncode == 128 + signal.SIGTERM and "terminating the ranks" in err
    for pid in rank_pids:
        with pytest.raises(ProcessLookupError):
            os.kill(pid, 0)
