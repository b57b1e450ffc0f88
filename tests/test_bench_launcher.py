"""bench.py --gpus N started without a launcher: the parent must start N rank processes itself,
relay rank 0's line, fail when a rank fails and refuse to run when fewer than N GPUs are visible
(rather than measure one GPU and call it N).  The ranks here are a stand-in program: no GPU."""
import json
import os
import subprocess
import sys
import textwrap

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _stub(tmp_path, body):
    path = tmp_path / "stub_rank.py"
    path.write_text(textwrap.dedent(body))
    return str(path)


def test_parent_starts_n_ranks_and_relays_rank0_line(tmp_path, capfd):
    stub = _stub(tmp_path, """
        import json, os, sys
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["MASTER_ADDR"] == "127.0.0.1"
        open(os.path.join(os.path.dirname(__file__), f"seen{rank}"), "w").write(" ".join(sys.argv[1:]))
        print("noise that is not the result")
        if rank == 0:
            print(json.dumps({"n_gpus": world, "local_rank": int(os.environ["LOCAL_RANK"])}))
    """)
    rc = bench.launch_ranks(2, ["--gpus", "2", "--steps", "1"], script=stub, device_count=2, timeout=300)
    assert rc == 0
    lines = [ln for ln in capfd.readouterr().out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"n_gpus": 2, "local_rank": 0}
    for rank in (0, 1):                                            # both ranks ran, with the parent's arguments
        assert (tmp_path / f"seen{rank}").read_text() == "--gpus 2 --steps 1"


def test_parent_fails_when_a_rank_fails(tmp_path, capfd):
    stub = _stub(tmp_path, """
        import os, sys
        if os.environ["RANK"] == "1":
            sys.exit(3)
        print('{"n_gpus": 2}')
    """)
    assert bench.launch_ranks(2, [], script=stub, device_count=2, timeout=300) != 0
    assert not [ln for ln in capfd.readouterr().out.splitlines() if ln.startswith("{")]   # no result relayed


def test_parent_fails_without_a_result_line(tmp_path):
    stub = _stub(tmp_path, "print('nothing useful')\n")
    assert bench.launch_ranks(2, [], script=stub, device_count=2, timeout=300) != 0


def test_too_few_gpus_is_an_error_not_a_one_gpu_run(tmp_path):
    marker = tmp_path / "ran"
    stub = _stub(tmp_path, f"open({str(marker)!r}, 'w').write('x')\n")
    assert bench.launch_ranks(2, [], script=stub, device_count=1) != 0
    assert not marker.exists()                                     # no rank was started


def _run_bench(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=300)


def test_command_line_refuses_more_gpus_than_visible():
    """The shape of the driver's single-process command with N > 1 (no GPU in the CPU test run: 0 visible)."""
    import torch
    if torch.cuda.device_count() >= 2:
        return
    done = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert done.returncode != 0 and b"visible GPUs" in done.stderr and not done.stdout.strip()


def test_world_size_must_match_gpus():
    done = _run_bench(["--gpus", "4"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2"})
    assert done.returncode != 0 and b"WORLD_SIZE=2" in done.stderr and not done.stdout.strip()
