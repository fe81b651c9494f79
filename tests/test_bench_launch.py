"""CPU: `bench.py --gpus N` starts N ranks itself (one process per GPU through torch.distributed.run) when it is
not already one rank of a launch; `--gpus 1` stays a plain single process.  No GPU is needed: the child command is
captured instead of run."""
import argparse
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _args(n):
    return argparse.Namespace(gpus=n)


def test_gpus_1_runs_in_process():
    ran = []
    assert bench.maybe_self_launch(_args(1), ['--gpus', '1'], environ={}, device_count=8, run=ran.append) is None
    assert ran == []


def test_gpus_2_launches_two_ranks():
    ran = []

    def run(cmd):
        ran.append(cmd)
        return 7
    argv = ['--gpus', '2', '--steps', '3', '--warmup', '1']
    assert bench.maybe_self_launch(_args(2), argv, environ={}, device_count=8, run=run) == 7   # the children's code
    (cmd,) = ran
    assert cmd[0] == sys.executable and cmd[1:3] == ['-m', 'torch.distributed.run']
    assert cmd[cmd.index('--nproc-per-node') + 1] == '2'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert 0 < int(cmd[cmd.index('--master-port') + 1]) < 65536
    script = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[script + 1:] == argv                      # own arguments passed through unchanged


def test_inside_a_launch_nothing_is_started():
    ran = []
    env = {'WORLD_SIZE': '4', 'RANK': '1', 'LOCAL_RANK': '1'}
    assert bench.maybe_self_launch(_args(4), ['--gpus', '4'], environ=env, device_count=8, run=ran.append) is None
    assert ran == []
    with pytest.raises(SystemExit, match='WORLD_SIZE=4'):
        bench.maybe_self_launch(_args(8), ['--gpus', '8'], environ=env, device_count=8, run=ran.append)


def test_more_gpus_than_visible_is_refused():
    with pytest.raises(SystemExit, match='2 GPU'):
        bench.maybe_self_launch(_args(8), ['--gpus', '8'], environ={}, device_count=2, run=lambda c: 0)


def test_roofline_kernel_comes_from_a_committed_summary():
    dom = bench.dominant_kernel('adv')
    assert dom is not None
    label, src = dom
    assert label in dict((lab, 1) for _, lab in bench._KERNEL_LABELS)
    assert src.startswith('profiles/') and os.path.exists(os.path.join(ROOT, src))
