"""CPU: `python3 bench.py --gpus 2` with no external launcher — the form the driver uses — starts its own two rank
processes, they meet in rendezvous.FileRendezvous, rank 0's JSON line comes back on the launcher's stdout, and a failing
rank makes the launcher fail.  The HIP front end is replaced by tests/stub_backend.py (no GPU here); everything else is
bench.py's real code path: argument handling, launch_ranks, the environment of the ranks, the file rendezvous and its
session handshake, ChunkPipeline, the gather bookkeeping, the barrier / max-over-ranks timing."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--steps", "3", "--warmup", "1", "--width", "160", "--height", "120", "--distinct-frames", "8", "--pairs-per-step", "4",
         "--contexts", "2", "--no-cpu-baseline", "--no-profile", "--no-stream-pass", "--no-sustain", "--no-faithful-pass"]


def _run(gpus, extra_env=None, tmp=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "VO_RENDEZVOUS_KEY")}
    env.update(VO_BENCH_STUB="tests.stub_backend", PYTHONPATH=ROOT, VO_RENDEZVOUS_DIR=str(tmp))
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus)] + FLAGS, env=env, cwd=ROOT,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)


def test_bench_starts_its_own_ranks(tmp_path):
    p = _run(2, tmp=tmp_path)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout                                   # ONE JSON line: rank 0's
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak"
    assert line["config"]["n_ranks_in_communicator"] == 2
    assert line["config"]["trajectory_poses_gathered"] == 2 * 4 + 1     # both ranks' records arrived
    assert line["config"]["gathered_equals_local"] is True
    assert "STUB" in line["data"]
    assert line["value"] == round(2 * 4 * 3 / (line["ms_per_step"] * 3 / 1000), 2) or line["value"] > 0
    assert not [d for d in os.listdir(tmp_path) if d.startswith("vo_rdv_")]     # rank 0 removed the rendezvous directory


def test_single_gpu_line_needs_no_launcher(tmp_path):
    p = _run(1, tmp=tmp_path)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["config"]["n_ranks_in_communicator"] is None


def test_failing_rank_fails_the_launch(tmp_path):
    t0 = time.time()
    p = _run(2, {"VO_STUB_FAIL_RANK": "1"}, tmp=tmp_path)
    assert p.returncode == 7, (p.returncode, p.stderr[-2000:])
    assert "rank 1 exited with 7" in p.stderr
    assert time.time() - t0 < 120                                       # rank 0 was ended, not left waiting for the rendezvous


def test_stale_rendezvous_directory_is_not_trusted(tmp_path):
    """A crashed launch leaves its session and files behind; the next launch with the same key must not read them."""
    from visual_odometry_amd.rendezvous import FileRendezvous
    d = tmp_path / "vo_rdv_k"
    d.mkdir()
    (d / "session").write_text("deadbeefdeadbeef 999999999 12345")      # rank 0 of that launch no longer exists
    (d / "deadbeefdeadbeef_go").write_text("")
    (d / "deadbeefdeadbeef_rccl_id_1.bin").write_bytes(b"stale")
    code = ("import sys; sys.path.insert(0, %r); from visual_odometry_amd.rendezvous import FileRendezvous\n"
            "r = int(sys.argv[1]); rdv = FileRendezvous(r, 2, key='k', root=sys.argv[2], timeout=60)\n"
            "print(rdv.broadcast(b'fresh' if r == 0 else b'', 'rccl_id')); rdv.barrier(); rdv.close()\n") % ROOT
    follower = subprocess.Popen([sys.executable, "-c", code, "1", str(tmp_path)], stdout=subprocess.PIPE, text=True)
    time.sleep(1.0)                                                      # the follower is up first and sees only the stale session
    assert follower.poll() is None
    leader = subprocess.Popen([sys.executable, "-c", code, "0", str(tmp_path)], stdout=subprocess.PIPE, text=True)
    assert leader.communicate(timeout=90)[0].strip() == "b'fresh'" and follower.communicate(timeout=90)[0].strip() == "b'fresh'"
    assert leader.returncode == 0 and follower.returncode == 0
    assert FileRendezvous is not None and not d.exists()


def test_bench_quotes_counters_only_from_a_pass_on_these_sources(tmp_path, monkeypatch):
    """roofline.traffic / valu come from profiles/<tag>_pmc_traffic.json only when the file carries the hash of the sources the library
    is built from (tools/source_hash.py): counters of an older kernel are reported as null, not replayed (round-3 review, weak #4)."""
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    from tools import source_hash as SH
    prof = tmp_path / "profiles"; prof.mkdir()
    rec = {"k_fast<false>": {"hbm_bytes_per_launch": 865000000, "valu_wave_insts_per_launch": 398000000},
           "_meta": {"frames_per_launch": 257, "source_hash": SH.source_hash()}}
    (prof / f"{bench.PROFILE_TAG}_pmc_traffic.json").write_text(json.dumps(rec))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench._PMC_CACHE.clear()
    assert bench.pmc_sum("orb", "fast_score_nms", "hbm_bytes_per_launch", 257) == 865000000.0
    assert bench.pmc_sum("orb", "fast_score_nms", "hbm_bytes_per_launch", 129) is None            # another chunk size: not this pass
    rec["_meta"]["source_hash"] = "0" * 40
    (prof / f"{bench.PROFILE_TAG}_pmc_traffic.json").write_text(json.dumps(rec))
    bench._PMC_CACHE.clear()
    assert bench.pmc_sum("orb", "fast_score_nms", "hbm_bytes_per_launch", 257) is None
    assert bench.pmc_sum("orb", "fast_score_nms", "valu_wave_insts_per_launch", 257) is None
    bench._PMC_CACHE.clear()


def test_traffic_tool_doubles_fetch_size_for_streaming_kernels_only():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import collect_traffic as CT
    for k in ("k_fast<false>", "k_resize_direct", "k_blur_direct", "k_sb_sweep<27, false>", "k_sb_extrema<5>", "k_jpeg_idct", "k_nn_fp4<false>"):
        assert CT.streaming(k), k
    for k in ("k_harris", "k_angle", "k_brief", "k_sb_descriptor", "k_sb_orient", "k_ransac", "k_pose", "k_sel_rows<true>"):
        assert not CT.streaming(k), k


def test_stopping_the_launcher_stops_the_ranks(tmp_path):
    """SIGTERM to `bench.py --gpus 2` (a driver's time limit) reaches the rank processes: nothing is left behind."""
    import signal
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "VO_RENDEZVOUS_KEY")}
    env.update(VO_BENCH_STUB="tests.stub_backend", PYTHONPATH=ROOT, VO_RENDEZVOUS_DIR=str(tmp_path), VO_STUB_HANG_RANK="1")
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + FLAGS, env=env, cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    marker = tmp_path / "hung_rank_pid"
    t0 = time.time()
    while not marker.exists() and time.time() - t0 < 120:
        time.sleep(0.1)
    assert marker.exists(), "the hanging rank never started"
    pid = int(marker.read_text())
    p.send_signal(signal.SIGTERM)
    p.communicate(timeout=60)
    assert p.returncode == 130
    t0 = time.time()
    while time.time() - t0 < 20:
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            break
        time.sleep(0.1)
    else:
        os.kill(pid, signal.SIGKILL)
        raise AssertionError("the rank survived its launcher")
