"""GPU: the data-parallel step over RCCL (torch.distributed backend "nccl") in a group of ONE rank.

More ranks need more GPUs than the test box has (RCCL refuses two ranks on one device), but a group of one still runs
the real thing end to end: the process group on RCCL, flat device buckets all-reduced asynchronously on RCCL's stream
from inside backward(), stream-ordered `work.wait()`, the device-side token weight and the scalar all-reduce — with
the host never synchronising inside the step.  Averaging over one rank is the identity, so the result must equal the
plain single-process step bit for bit (embedding gradients, which are atomic scatter-adds, to rounding).  Runs in a child process (the process group is per process)."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_zz_dp_gpu import T, _inputs, _models, _step          # same shapes and inputs as the two-rank gloo test


def _child(port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    import a4nt_amd as impl
    from a4nt_amd import ops
    ops.BWD_MAX_WGS = 256            # bench.py's data-parallel setting: backward sweeps queued beside a collective: one per CU
    src, tgt, gb, noise = _inputs()
    res = {}
    for mode in ('plain', 'rccl'):
        G, D = _models(impl)
        reds = (None, None)
        if mode == 'rccl':
            reds = (impl.GradAllReducer(list(G.parameters()), bucket_mb=0.01, force=True),
                    impl.GradAllReducer(list(D.parameters()), bucket_mb=0.01, force=True))
            assert reds[0].on and not reds[0].host_staged
        for it in range(2):                                   # second pass: buckets go out under backward()
            G.zero_grad(); D.zero_grad()
            out = _step(impl, G, D, src, tgt, gb, [u.clone() for u in noise], reds)
        if mode == 'rccl':
            cnt = reds[1].all_reduce_scalars(torch.tensor([3.0, 4.0], device='cuda'))
            assert cnt.tolist() == [3.0, 4.0]
            # the budget switch follows the collectives: sweeps queued in front of the first complete bucket kept the
            # device's own budget, those queued behind it took BWD_MAX_WGS, and nothing is left in flight
            assert ops.COLLECTIVES_IN_FLIGHT == 0, ops.COLLECTIVES_IN_FLIGHT
            assert ops.BWD_BUDGET_STATS['switched'] > 0 and ops.BWD_BUDGET_STATS['kept'] > 0, ops.BWD_BUDGET_STATS
        else:
            assert ops.BWD_BUDGET_STATS['switched'] == 0, ops.BWD_BUDGET_STATS      # no collective, no switch
        impl.check_timeouts()
        torch.cuda.synchronize()
        res[mode] = ({n: p.grad.cpu().numpy() for n, p in G.named_parameters()},
                     {n: p.grad.cpu().numpy() for n, p in D.named_parameters()}, float(out[1]['grad_norm']),
                     float(out[1]['cyc_loss']))
        # two STEPPING updates: under a process group the discriminator's exchange + optimizer step are deferred into
        # the generator update (behind the reconstruction pass's first products); the weights must come out as in the
        # plain step, where the discriminator is stepped before the generator update starts
        G2, D2 = _models(impl)
        reds2 = (None, None)
        if mode == 'rccl':
            reds2 = (impl.GradAllReducer(list(G2.parameters()), bucket_mb=0.01, force=True),
                     impl.GradAllReducer(list(D2.parameters()), bucket_mb=0.01, force=True))
        optG, optD = impl.make_optimizers(G2, D2)
        for it in range(2):
            G2.set_noise([u.cuda() for u in noise])
            impl.fused_update(G2, D2, optG, optD, src, tgt, gb, 0, -1, T, 0.5, 2, cycle_loss_type='ml', step=True,
                              reducers=reds2)
        impl.check_timeouts()
        torch.cuda.synchronize()
        assert ops.COLLECTIVES_IN_FLIGHT == 0
        res[mode] = res[mode] + ({'G.' + n: p.detach().cpu().numpy() for n, p in G2.named_parameters()},
                                 {'D.' + n: p.detach().cpu().numpy() for n, p in D2.named_parameters()})
    dist.destroy_process_group()
    q.put(res)


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_rccl_group_of_one_equals_plain_step():
    import numpy as np
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_child, args=(29900 + (os.getpid() % 90), q))
    p.start()
    import queue
    import time
    res, t0 = None, time.time()
    while res is None:
        try:
            res = q.get(timeout=5)
        except queue.Empty:
            if not p.is_alive():        # the child raised: fail now, with its exit code (its traceback is on stderr)
                raise AssertionError('the RCCL single-rank child exited with code %s without a result' % p.exitcode)
            if time.time() - t0 > 200:
                p.kill()
                p.join(timeout=30)
                raise AssertionError('the RCCL single-rank step produced no result within 200 s')
    p.join(timeout=60)
    assert p.exitcode == 0
    (gG, gD, gn, cyc, wG, wD), (rG, rD, rn, rcyc, vG, vD) = res['plain'], res['rccl']
    for a, b in ((wG, vG), (wD, vD)):           # weights after two stepping updates (a misplaced step shows as ~1e-2)
        assert set(a) == set(b)
        for k in a:
            assert np.abs(a[k] - b[k]).max() <= 2e-5 * max(np.abs(a[k]).max(), 1e-30), k
    assert cyc == rcyc and abs(gn - rn) <= 1e-6 * abs(gn)
    for a, b in ((gG, rG), (gD, rD)):
        assert set(a) == set(b)
        for k in a:
            if 'emb' in k or 'encoder.weight' in k:
                # embedding gradients are scatter-added with fp32 atomics: the order of the adds varies run to run
                assert np.abs(a[k] - b[k]).max() <= 2e-6 * max(np.abs(a[k]).max(), 1e-30), k
            else:
                assert np.array_equal(a[k], b[k]), k
