"""GPU, two processes sharing cuda:0 over gloo (RCCL refuses two ranks on one device; the collective
backend is the only thing that differs from the 8-GPU run): the data-parallel adversarial step
(fused_update with the overlapped GradAllReducer, deferred discriminator step, token-weighted reconstruction
loss) on 2 ranks x b sentences equals the single-process step on 2b sentences.

Small shapes on purpose: two processes' persistent sweeps must be co-resident on the one GPU.
Collected last (file name): a gloo hiccup here must not hide the HIP parity tests behind `pytest -x`."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
V, E, H, T, Bfull = 40, 16, 64, 6, 8
GP = dict(vocabulary_size=V, embedding_size=E, enc_hidden_size=H, dec_hidden_size=H, enc_hidden_depth=1,
          dec_hidden_depth=1, en_residual_conn=0, pad_auth_vec=4, num_output_layers=2, batch_size=Bfull,
          gumbel_hard=True, softmax_scale=3.0, drop_prob_emb=0., drop_prob_encoder=0., drop_prob_decoder=0.)
EP = dict(vocabulary_size=V, embedding_size=E, hidden_size=H, hidden_depth=2, num_output_layers=2, mode='classifier',
          drop_prob_encoder=0., drop_prob_decoder=0.)


def _inputs():
    import a4nt_amd as impl
    g = torch.Generator().manual_seed(7)
    lens = [6, 6, 5, 4, 6, 3, 3, 2]                       # rank 0 gets the first four rows: ragged token counts
    mk = lambda a: (lambda b: (b[0], b[1], torch.full((Bfull,), a, dtype=torch.long), b[2]))(
        impl.synthetic_batch(g, T, Bfull, V, lens))
    src, tgt, gb = mk(0), mk(1), mk(0)
    noise = [torch.rand(2 * Bfull, V + 1, generator=g) for _ in range(T)]      # rows: [src rows ; gb rows]
    return src, tgt, gb, noise


def _shard(batch, r0, r1):
    return (batch[0][:, r0:r1].contiguous(), batch[1][:, r0:r1].contiguous(), batch[2][r0:r1], batch[3][r0:r1])


def _models(impl):
    torch.manual_seed(0)
    G, D = impl.CharTranslator(GP), impl.CharLstm(EP)
    with torch.no_grad():
        G.decoder_W.mul_(300.0)
    return G.train(), D.train()


def _step(impl, G, D, src, tgt, gb, noise, reducers):
    optG, optD = impl.make_optimizers(G, D)
    G.set_noise([u.cuda() for u in noise])
    return impl.fused_update(G, D, optG, optD, src, tgt, gb, 0, -1, T, 0.5, 2, cycle_loss_type='ml', step=False,
                             reducers=reducers)


def _worker(rank, world, port, q, dump_path):
    sys.path.insert(0, ROOT)
    import faulthandler
    dump = open(dump_path, 'w')
    faulthandler.dump_traceback_later(75, exit=False, file=dump)      # where a stuck rank is, for the parent's report
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import a4nt_amd as impl
    impl.lstm_set_max_workgroups(256)
    G, D = _models(impl)
    src, tgt, gb, noise = _inputs()
    b = Bfull // world
    r0, r1 = rank * b, (rank + 1) * b
    my_noise = [torch.cat([u[r0:r1], u[Bfull + r0:Bfull + r1]], 0) for u in noise]
    # before_wait: this rank's persistent sweeps are checked for a hand-off timeout (workgroups of two processes not
    # co-resident) BEFORE it blocks in a collective -- such a stall fails here with its own message instead of looking
    # like a stall of gloo
    reds = (impl.GradAllReducer(list(G.parameters()), bucket_mb=0.01, before_wait=impl.check_timeouts),
            impl.GradAllReducer(list(D.parameters()), bucket_mb=0.01, before_wait=impl.check_timeouts))
    out = None
    for it in range(2):                                   # second pass: buckets go out under backward()
        G.zero_grad(); D.zero_grad()
        out = _step(impl, G, D, _shard(src, r0, r1), _shard(tgt, r0, r1), _shard(gb, r0, r1),
                    [u.clone() for u in my_noise], reds)
    impl.check_timeouts()
    torch.cuda.synchronize()
    if rank == 0:
        q.put(({n: p.grad.cpu().numpy() for n, p in G.named_parameters()},
               {n: p.grad.cpu().numpy() for n, p in D.named_parameters()}, float(out[1]['grad_norm'])))   # by value
    faulthandler.cancel_dump_traceback_later()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_two_ranks_on_one_gpu_equal_single_process():
    sys.path.insert(0, ROOT)
    import a4nt_amd as impl
    from golden_util import assert_close
    G, D = _models(impl)
    src, tgt, gb, noise = _inputs()
    rd, rg = _step(impl, G, D, src, tgt, gb, noise, (None, None))
    impl.check_timeouts()
    torch.cuda.synchronize()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 1000)
    import tempfile
    dumps = [os.path.join(tempfile.gettempdir(), 'a4nt_dp_rank%d_%d.txt' % (r, os.getpid())) for r in range(2)]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, dumps[r])) for r in range(2)]
    for p in procs:
        p.start()
    import queue
    import time
    got, t0 = None, time.time()
    while got is None and time.time() - t0 < 90:
        try:
            got = q.get(timeout=3)
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead:                    # a rank raised (its traceback is on stderr): not a hang, fail at once
                for p in procs:
                    p.kill()
                raise AssertionError('a data-parallel rank exited with code %s without a result' % dead)
    if got is not None:
        gG, gD, gnorm = got
    else:
        # A hang is a failure: a deadlock in GradAllReducer, two processes' persistent sweeps that are not co-resident,
        # or gloo not completing async all-reduces of CUDA tensors on a busy GPU (seen with plain torch matmuls,
        # DESIGN.md section 7) all look the same from here; the ranks' stacks (faulthandler, 75 s) tell them apart.
        for p in procs:
            p.kill()
        for p in procs:
            p.join(timeout=30)
        stacks = ''
        for r, d in enumerate(dumps):
            try:
                stacks += '\n--- rank %d ---\n%s' % (r, open(d).read()[-3000:])
            except OSError:
                pass
        raise AssertionError('two data-parallel ranks over gloo on one GPU produced no result within 90 s '
                             '(exit codes %s); stacks of the ranks at 75 s:%s' % ([p.exitcode for p in procs], stacks))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert_close(torch.tensor(gnorm), rg['grad_norm'].cpu(), 5e-5, 'global grad norm')
    for n, p in D.named_parameters():
        assert_close(torch.from_numpy(gD[n]), p.grad.cpu(), 5e-5, 'D grad ' + n)
    for n, p in G.named_parameters():
        assert_close(torch.from_numpy(gG[n]), p.grad.cpu(), 5e-5, 'G grad ' + n)
