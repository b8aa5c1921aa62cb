"""CPU: the host logic of bench.py - the self-launcher for --gpus N (the driver starts bench.py either
under torch.distributed.run or bare), the marker-cut reduction of a rocprofv3 PMC trace, and the
byte accounting of SURVEY.md section 8d.  No GPU call is made."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_launch_command_is_the_drivers_form():
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "5"], port=29511, python="python3")
    assert cmd[:3] == ["python3", "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "5"]
    assert 1024 < bench.free_port() < 65536


def test_self_launch_relays_rank0_line_and_exit_code(monkeypatch, capfd):
    line = json.dumps({"metric": "m", "value": 1.0, "n_gpus": 2})
    child = f"import sys; print('noise'); print({line!r}); sys.exit(0)"
    monkeypatch.setattr(bench, "launch_command", lambda n, argv: [sys.executable, "-c", child])
    assert bench.self_launch(2, []) == 0
    out, err = capfd.readouterr()
    assert out.strip() == line          # exactly ONE line on stdout: rank 0's JSON
    assert "noise" in err               # anything else the ranks print goes to stderr
    monkeypatch.setattr(bench, "launch_command", lambda n, argv: [sys.executable, "-c", "import sys; sys.exit(3)"])
    assert bench.self_launch(2, []) == 3
    monkeypatch.setattr(bench, "launch_command", lambda n, argv: [sys.executable, "-c", "print('no json')"])
    assert bench.self_launch(2, []) == 1  # exit 0 without a line is a failure


def test_gpus_n_without_world_size_self_launches_before_any_gpu_call(monkeypatch):
    """`python bench.py --gpus 2` (WORLD_SIZE unset) must reach self_launch - with torch.cuda untouched."""
    import torch

    seen = {}
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    monkeypatch.setattr(bench, "self_launch", lambda n, argv: seen.setdefault("call", (n, list(argv))) and 0)
    monkeypatch.setattr(torch.cuda, "set_device", lambda *a: (_ for _ in ()).throw(AssertionError("GPU touched")))
    try:
        bench.main()
    except SystemExit as e:
        assert e.code in (0, None)
    assert seen["call"] == (2, ["--gpus", "2", "--steps", "3"])


def test_real_torchrun_child_of_the_launcher_on_cpu(tmp_path):
    """The launcher's command line works end to end: two ranks of a tiny stand-in script under
    torch.distributed.run, gloo, rank 0 prints the line."""
    script = tmp_path / "fake_bench.py"
    script.write_text(
        "import os, json, torch.distributed as d\n"
        "d.init_process_group('gloo')\n"
        "import torch; t = torch.ones(1); d.all_reduce(t)\n"
        "if d.get_rank() == 0: print(json.dumps({'metric': 'x', 'n_gpus': d.get_world_size(), 'sum': t.item()}))\n"
        "d.destroy_process_group()\n")
    cmd = bench.launch_command(2, [])
    cmd[cmd.index(os.path.join(ROOT, "bench.py"))] = str(script)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rec == {"metric": "x", "n_gpus": 2, "sum": 2.0}


def _run_selftest(mode, timeout_env="20"):
    import time

    cmd = bench.launch_command(2, ["--gpus", "2", "--steps", "8", "--selftest-ranks", mode])
    env = dict(os.environ, RECMAN_BENCH_TIMEOUT_S=timeout_env)
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    return r.returncode, lines, time.time() - t0, r.stderr


def test_rank_runtime_prints_one_line_when_every_rank_finishes():
    rc, lines, _, err = _run_selftest("none")
    assert rc == 0, err[-2000:]
    assert len(lines) == 1 and lines[0]["selftest"] is True and lines[0]["n_gpus"] == 2


def test_a_rank_that_dies_mid_run_ends_the_job_quickly_with_a_nonzero_code():
    """Rank 1 exits abruptly before its 4th collective: rank 0 must not sit in all_reduce for the backend's
    default timeout (10-30 min, past the driver's limit) - the job ends non-zero well within 150 s, and if rank 0
    got to report at all, its line carries an "error" field, never a value."""
    rc, lines, dt, err = _run_selftest("die:1:3")
    assert rc != 0 and dt < 150, (rc, dt, err[-2000:])
    assert all(l.get("value") is None and "error" in l for l in lines), lines


def test_a_rank_that_raises_mid_run_ends_the_job_with_an_error_line_or_code():
    rc, lines, dt, err = _run_selftest("raise:0:2")
    assert rc != 0 and dt < 150, (rc, dt, err[-2000:])
    assert len(lines) == 1 and lines[0]["value"] is None and "selftest" in lines[0]["error"], lines


def _row(i, name, counter, v):
    return {"Dispatch_Id": str(i), "Kernel_Name": name, "Counter_Name": counter, "Counter_Value": str(v)}


def test_reduce_counter_cuts_the_trace_at_the_markers():
    M = "(anonymous namespace)::rm_profile_marker_kernel(int, int*)"
    E = "void (anonymous namespace)::embed_fwd_fused_kernel<8, 4, false, true>(long const*)"
    C = "void (anonymous namespace)::cin_fwd_kernel<4, 1>(float const*)"
    rows, i = [], 0
    seq = [(E, 999.0), (M, 0), (E, 10.0), (E, 12.0), (E, 14.0), (M, 0),      # deepfm probe
           (E, 777.0), (C, 5.0), (M, 0), (C, 100.0), (C, 101.0), (C, 102.0), (M, 0)]  # xdeepfm: its own embed first
    for name, v in seq:
        rows.append(_row(i, name, "FETCH_SIZE", v))
        rows.append(_row(i, name, "GRBM_GUI_ACTIVE", 1))  # another counter of the same dispatch
        i += 1
    out = {}
    plan = [("deepfm", "embed_fwd_fused_kernel"), ("xdeepfm", "cin_fwd_kernel")]
    assert bench.reduce_counter(list(reversed(rows)), plan, "FETCH_SIZE", out) is None
    assert out == {"deepfm": {"embed_fwd_fused_kernel": {"FETCH_SIZE": 12.0}},
                   "xdeepfm": {"cin_fwd_kernel": {"FETCH_SIZE": 101.0}}}
    assert "marker" in bench.reduce_counter(rows[:-2], plan, "FETCH_SIZE", {})
    assert "launches" in bench.reduce_counter([r for r in rows if r["Counter_Value"] != "12.0"], plan, "FETCH_SIZE", {})


def test_attach_traffic_doubles_fetch_and_prices_the_launch():
    roofs = [{"symbol": "embed_fwd_fused_kernel", "bound": "hbm", "avg_launch_us": 100.0, "traffic": None}]
    live = {"deepfm": {"embed_fwd_fused_kernel": {"FETCH_SIZE": 100000.0, "WRITE_SIZE": 50000.0}}}
    bench.attach_traffic(roofs, "deepfm", live, {})
    r = roofs[0]
    assert r["traffic"] == (2 * 100000 + 50000) * 1024 and r["traffic_live"] is True
    assert abs(r["frac_traffic"] - 256e6 / 100e-6 / 1e9 / 8000.0) < 1e-4
    roofs = [{"symbol": "k", "bound": "mfma", "avg_launch_us": 1.0, "traffic": None}]
    bench.attach_traffic(roofs, "dcn", {}, {"dcn": {"k": {"bytes": 5.0, "source": "profiles/x.md"}}})
    assert roofs[0]["traffic"] == 5.0 and roofs[0]["traffic_live"] is False and "frac_traffic" not in roofs[0]
    roofs = [{"symbol": "k", "bound": "hbm", "avg_launch_us": 1.0, "traffic": None}]
    bench.attach_traffic(roofs, "dcn", {}, {})
    assert roofs[0]["traffic"] is None and roofs[0]["traffic_live"] is False


def test_step_hbm_is_surveys_586_7_mb():
    w = bench.WORKLOADS["deepfm"]
    rec = bench.step_hbm(w, w["B"], 0.2)
    assert rec["algorithmic_bytes"] == 65536 * 26 * 76 + 65536 * 4 + 65536 * 26 * 64 + 65536 * 26 * 204 + 65536 * 4
    assert abs(rec["algorithmic_bytes"] / 1e6 - 586.7) < 0.1
    assert abs(rec["frac"] - rec["algorithmic_bytes"] / 0.2e-3 / 8e12) < 1e-4


def test_default_run_is_the_contract_line_plus_sub_records():
    a = bench.parse([])
    assert a.gpus == 1 and a.workload is None and not a.only and a.steps > 0 and a.warmup > 0
    assert bench.parse(["--workload", "dcn"]).workload == "dcn"
    assert set(bench.CONFIG_NO) == {"deepfm", "xdeepfm", "dcn", "xdeepfm_100m"}
