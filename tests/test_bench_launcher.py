"""bench.py's `--gpus N` logic on CPU (no GPU work): the function that turns `--gpus N` into either "this process is the job / one rank of it"
or the child command line under torch.distributed.run, and the refusals that keep an N-GPU line from being printed by fewer GPUs."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import bench  # noqa: E402


def test_single_gpu_runs_in_process():
    assert bench.launch_plan(1, {}, 1, ["--steps", "5"]) == ("inprocess", None)
    assert bench.launch_plan(1, {}, 8, []) == ("inprocess", None)


def test_gpus_n_without_launcher_spawns_one_rank_per_gpu():
    mode, cmd = bench.launch_plan(4, {}, 8, ["--gpus", "4", "--steps", "20", "--warmup", "3"], port=29511)
    assert mode == "spawn"
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(REPO, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "20", "--warmup", "3"]      # the child sees the caller's flags unchanged
    # a free port is picked when none is given
    _, cmd2 = bench.launch_plan(2, {}, 2, ["--gpus", "2"])
    assert 1024 < int(cmd2[cmd2.index("--master-port") + 1]) < 65536


def test_refuses_more_gpus_than_visible():
    with pytest.raises(SystemExit) as e:
        bench.launch_plan(2, {}, 1, ["--gpus", "2"])
    assert "only 1 HIP device" in str(e.value.code)
    with pytest.raises(SystemExit):
        bench.launch_plan(8, {}, 0, ["--gpus", "8"])
    with pytest.raises(SystemExit):
        bench.launch_plan(1, {}, 0, [])
    with pytest.raises(SystemExit):
        bench.launch_plan(0, {}, 8, [])


def test_rank_of_a_launched_job_checks_world_size_against_gpus():
    env = {"WORLD_SIZE": "8", "RANK": "3", "LOCAL_RANK": "3"}
    assert bench.launch_plan(8, env, 8, ["--gpus", "8"]) == ("inprocess", None)
    with pytest.raises(SystemExit) as e:                       # torchrun with 8 ranks but bench.py told --gpus 4 (or left at the default 1)
        bench.launch_plan(4, env, 8, ["--gpus", "4"])
    assert "WORLD_SIZE" in str(e.value.code)
    with pytest.raises(SystemExit):
        bench.launch_plan(1, env, 8, [])
    with pytest.raises(SystemExit):                            # a rank without a device of its own
        bench.launch_plan(8, env, 2, ["--gpus", "8"])


def test_command_line_exits_nonzero_without_enough_gpus():
    # here (no GPU at all) `python bench.py --gpus 2` must fail loudly instead of printing a line
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "HIP device" in (r.stderr + r.stdout)
    assert "\"metric\"" not in r.stdout


def test_ranks_of_a_node_tell_the_wait_policy_about_each_other():
    """One process per GPU: rade_batch_rx's wait policy (spin while engines <= CPUs) only sees its own process's engines; bench.py hands it the number
    of ranks that share the node's CPUs ($RADE_SYNC_PEERS), and the library multiplies (rade_engine.c: sync_blocking_now): 8 ranks x 3 engines under a
    16-core quota sleep on a blocking event, a single rank spins."""
    import bench
    assert bench.sync_peers(1, {}) is None
    assert bench.sync_peers(8, {}) == "8"
    assert bench.sync_peers(8, {"LOCAL_WORLD_SIZE": "4"}) == "4"
    assert bench.sync_peers(8, {"LOCAL_WORLD_SIZE": "4", "RADE_SYNC_PEERS": "2"}) == "2"
    import ctypes as C
    from radae_amd import engine
    lib = engine.load_library()
    lib.rade_sync_policy.argtypes = [C.c_int, C.c_double]
    assert lib.rade_sync_policy(3 * 8, 16.0) == 1 and lib.rade_sync_policy(3 * 1, 16.0) == 0


def test_oversubscribe_developer_mode_plan():
    """`bench.py --gpus 8 --oversubscribe-device 0`: the eight ranks of an 8-GPU job on ONE device (host-contention proxy): spawned under
    torch.distributed.run although a single device is visible; refused when the device does not exist; a rank of it checks WORLD_SIZE against --gpus."""
    mode, cmd = bench.launch_plan(8, {}, 1, ["--gpus", "8", "--oversubscribe-device", "0"], port=29512, oversubscribe=0)
    assert mode == "spawn" and "--nproc-per-node=8" in cmd and cmd[-4:] == ["--gpus", "8", "--oversubscribe-device", "0"]
    assert bench.launch_plan(8, {"WORLD_SIZE": "8", "RANK": "5", "LOCAL_RANK": "5"}, 1, [], oversubscribe=0) == ("inprocess", None)
    with pytest.raises(SystemExit):
        bench.launch_plan(8, {"WORLD_SIZE": "4"}, 1, [], oversubscribe=0)
    with pytest.raises(SystemExit):
        bench.launch_plan(8, {}, 1, [], oversubscribe=1)
    with pytest.raises(SystemExit):
        bench.launch_plan(8, {}, 0, [], oversubscribe=0)
    assert bench.launch_plan(1, {}, 1, [], oversubscribe=0) == ("inprocess", None)
