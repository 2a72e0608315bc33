"""bench.py --gpus N must start N fresh rank processes itself when no launcher has (no
WORLD_SIZE), before anything touches the GPU, and relay rank 0's line (the
driver's contract for N = 1).  CPU only: the children here are stand-ins or fail at the
"needs a GPU" check, which is exactly what shows that they were started as ranks."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_spawn_ranks_starts_one_process_per_rank_with_the_rendezvous_environment():
    import bench
    child = [sys.executable, "-c",
             "import os, json; print(json.dumps({k: os.environ[k] for k in "
             "('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}))"]
    procs = bench.spawn_ranks(3, child, port=29417)
    assert len(procs) == 3
    out, _ = procs[0].communicate()
    for p in procs[1:]:
        assert p.wait() == 0
    env = json.loads(out)
    assert env == {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "3", "MASTER_ADDR": "127.0.0.1",
                   "MASTER_PORT": "29417"}


def test_bench_gpus_2_launches_two_ranks_and_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        return  # on a GPU box the real thing runs in test_gpu_configs
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu", "--steps", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    # one refusal per rank: both children were started, neither fell back to the CPU
    assert r.stderr.count("bench.py needs a GPU") == 2


def test_bench_under_an_outer_launcher_does_not_spawn_again():
    import torch
    if torch.cuda.is_available():
        return
    env = dict(os.environ, WORLD_SIZE="2", RANK="1", LOCAL_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29418")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stderr.count("bench.py needs a GPU") == 1


def test_launcher_reports_a_rank_killed_by_a_signal_and_ends_its_siblings():
    """a rank that dies by a signal has a NEGATIVE Popen return code (a GPU abort is
    -6); max() with healthy siblings' 0 used to turn that into exit status 0 with empty stdout,
    and the siblings were left to wait for the rendezvous timeout.  Stand-in ranks: rank 0 kills
    itself with SIGABRT, rank 1 would sleep for a minute."""
    import time
    import bench
    child = [sys.executable, "-c",
             "import os, signal, time\n"
             "if os.environ['RANK'] == '0':\n"
             "    os.kill(os.getpid(), signal.SIGABRT)\n"
             "time.sleep(60)\n"]
    t0 = time.perf_counter()
    rc = bench.run_as_launcher(2, [], child_cmd=child)
    assert rc == 6, rc
    assert time.perf_counter() - t0 < 30, "the surviving rank was not terminated"


def test_launcher_fails_when_a_rank_other_than_zero_fails(capsys):
    import bench
    child = [sys.executable, "-c",
             "import os, sys\n"
             "print('{\"rank0\": true}') if os.environ['RANK'] == '0' else sys.exit(3)\n"]
    rc = bench.run_as_launcher(2, [], child_cmd=child)
    assert rc == 3
    child_ok = [sys.executable, "-c", "import os\nif os.environ['RANK'] == '0': print('{\"ok\": 1}')\n"]
    assert bench.run_as_launcher(3, [], child_cmd=child_ok) == 0
    assert '{"ok": 1}' in capsys.readouterr().out
