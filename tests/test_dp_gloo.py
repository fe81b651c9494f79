"""CPU, world_size 2 over gloo: the data-parallel gradient exchange (a4nt_amd.GradAllReducer, the same
code that runs over RCCL on the GPUs) makes N ranks x batch b equal to 1 rank x batch N*b.

The compute inside each rank is the CPU oracle (the HIP modules need a GPU); what is under test is
the bucketing / async all-reduce / averaging / scalar agreement logic."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

EP = dict(vocabulary_size=40, embedding_size=12, hidden_size=16, hidden_depth=2, num_output_layers=2,
          mode='classifier', drop_prob_encoder=0.0, drop_prob_decoder=0.0)
T, B = 6, 8


def _data():
    g = torch.Generator().manual_seed(3)
    x = torch.randint(1, 41, (T, B), generator=g)
    y = torch.randint(0, 2, (B,), generator=g)
    return x, y


def _worker(rank, world, port, out_q, overlap):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import oracle
    from a4nt_amd import GradAllReducer, ops
    torch.manual_seed(0)
    m = oracle.CharLstm(EP)
    x, y = _data()
    sh = slice(rank * B // world, (rank + 1) * B // world)
    red = GradAllReducer(list(m.parameters()), bucket_mb=0.001, overlap=overlap)   # tiny buckets: several in flight
    issued_early = []
    for it in range(3):         # overlap: step 0 learns the gradient-ready order, steps 1-2 reduce under backward()
        m.zero_grad()
        logits, _ = m.forward_classify(x[:, sh], lens=[T] * (B // world))
        red.arm()
        F.cross_entropy(logits, y[sh]).backward()
        issued_early.append(len(red._pending))
        # the count that steers the backward sweeps' workgroup budget (ops.BWD_MAX_WGS) follows the exchange
        assert ops.COLLECTIVES_IN_FLIGHT == len(red._pending)
        red.launch()
        assert ops.COLLECTIVES_IN_FLIGHT == len(red._pending) > 0
        red.wait()
        assert ops.COLLECTIVES_IN_FLIGHT == 0
    cnt = red.all_reduce_scalars(torch.tensor([1.0 + rank, 10.0]))
    w_tok = float(red.token_weight(10 if rank == 0 else 30))
    if rank == 0:
        # by value (numpy): tensors travel as shared-memory handles that die with this process
        out_q.put(({n: p.grad.numpy().copy() for n, p in m.named_parameters()}, cnt.tolist(), issued_early, w_tok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize('overlap', [False, True])
def test_two_ranks_equal_one_big_batch(overlap):
    import oracle
    torch.manual_seed(0)
    ref = oracle.CharLstm(EP)
    x, y = _data()
    logits, _ = ref.forward_classify(x, lens=[T] * B)
    F.cross_entropy(logits, y).backward()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    port += 7 * int(overlap)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, overlap)) for r in range(2)]
    for p in procs:
        p.start()
    grads, cnt, issued_early, w_tok = q.get(timeout=100)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.allclose(torch.tensor(cnt), torch.tensor([3.0, 20.0]))
    assert abs(w_tok - 10 * 2 / 40.0) < 1e-6          # rank 0 holds 10 of the 40 valid tokens
    if overlap:
        # nothing can be issued while the order is being learnt; afterwards most buckets go out inside backward()
        assert issued_early[0] == 0 and issued_early[1] > 0 and issued_early[2] > 0, issued_early
    else:
        assert issued_early == [0, 0, 0]
    for n, p in ref.named_parameters():
        assert torch.allclose(torch.from_numpy(grads[n]), p.grad, rtol=1e-5, atol=1e-7), n


def test_single_process_is_noop():
    from a4nt_amd import GradAllReducer
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.full((3,), 2.0)
    red = GradAllReducer([p])
    red.launch()
    red.wait()
    assert torch.equal(p.grad, torch.full((3,), 2.0))


# ----------------------------------------------------------------------------------------------
# two models, two reducers in flight one after the other, ragged per-token loss (token_weight) and the scalar
# all-reduce -- the reducer usage pattern of adv_step.fused_update (discriminator step, then generator step with a
# per-token reconstruction CE), with CPU oracle models standing in for the HIP modules
GP2 = dict(vocabulary_size=40, embedding_size=12, enc_hidden_size=16, dec_hidden_size=16, enc_hidden_depth=1,
           dec_hidden_depth=1, en_residual_conn=0, pad_auth_vec=4, num_output_layers=2, batch_size=8,
           drop_prob_emb=0.0, drop_prob_encoder=0.0, drop_prob_decoder=0.0)
LENS2 = [6, 6, 5, 5, 4, 3, 2, 2]          # sorted; rank 0 holds 22 of the 33 tokens


def _data2():
    g = torch.Generator().manual_seed(5)
    x = torch.randint(1, 41, (T, B), generator=g)
    y = torch.randint(1, 41, (T, B), generator=g)
    for b, l in enumerate(LENS2):
        x[l:, b] = 0
        y[l:, b] = 0
    lab = torch.randint(0, 2, (B,), generator=g)
    return x, y, lab


def _two_model_step(oracle, G, D, x, y, lab, lens, redG, redD, n_glob_rows):
    """D-like step (mean over sentences) then G-like step (mean over valid tokens, token-weighted)."""
    from torch.nn.utils.rnn import pack_padded_sequence as pps
    D.zero_grad(); G.zero_grad()
    logits, _ = D.forward_classify(x, lens=lens)
    if redD is not None:
        redD.arm()
    F.cross_entropy(logits, lab).backward()
    if redD is not None:
        redD.launch()                                        # in flight while the generator-side forward runs
    scores, _ = G.forward_mltrain(x, lens, x, lens, auths=torch.zeros(len(lens), dtype=torch.long))
    ce = F.cross_entropy(pps(scores, lens)[0], pps(y, lens)[0])
    w = redG.token_weight(sum(lens)) if redG is not None else 1.0
    if redD is not None:
        redD.wait()
    if redG is not None:
        redG.arm()
    (ce * w).backward()
    if redG is not None:
        redG.launch()
        redG.wait()


def _worker2(rank, world, port, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import oracle
    from a4nt_amd import GradAllReducer
    torch.manual_seed(0)
    G, D = oracle.CharTranslator(GP2), oracle.CharLstm(EP)
    x, y, lab = _data2()
    b = B // world
    sh = slice(rank * b, (rank + 1) * b)
    lens = LENS2[sh]
    redG = GradAllReducer(list(G.parameters()), bucket_mb=0.001)
    redD = GradAllReducer(list(D.parameters()), bucket_mb=0.001)
    for it in range(3):                      # step 0 learns the bucket order of BOTH reducers, 1-2 overlap
        _two_model_step(oracle, G, D, x[:max(lens), sh], y[:max(lens), sh], lab[sh], lens, redG, redD, B)
    cnt = redD.all_reduce_scalars(torch.tensor([float(sum(lens)), 1.0]))
    if rank == 0:
        out_q.put(({n: p.grad.numpy().copy() for n, p in G.named_parameters()},
                   {n: p.grad.numpy().copy() for n, p in D.named_parameters()}, cnt.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_reducers_ragged_tokens_equal_one_big_batch():
    import oracle
    torch.manual_seed(0)
    G, D = oracle.CharTranslator(GP2), oracle.CharLstm(EP)
    x, y, lab = _data2()
    _two_model_step(oracle, G, D, x, y, lab, LENS2, None, None, B)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker2, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gG, gD, cnt = q.get(timeout=150)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert cnt == [float(sum(LENS2)), 2.0]
    for n, p in D.named_parameters():
        assert torch.allclose(torch.from_numpy(gD[n]), p.grad, rtol=1e-5, atol=1e-7), 'D ' + n
    for n, p in G.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        assert torch.allclose(torch.from_numpy(gG[n]), g, rtol=2e-5, atol=1e-7), 'G ' + n


# ------------------------------------------------------------------------------------------------
# persistent flat buckets: what happens to the parameters' .grad
# ------------------------------------------------------------------------------------------------
def _worker3(rank, world, port, out_q, overlap):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from a4nt_amd import GradAllReducer
    torch.manual_seed(1)
    used = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
    unused = torch.nn.Parameter(torch.randn(4))            # never part of the graph: its gradient must stay None
    extra = torch.nn.Parameter(torch.randn(3))             # gets a gradient from an EARLIER backward of the step
    params = [used[0], unused, used[1], extra]
    red = GradAllReducer(params, bucket_mb=1.0, overlap=overlap)
    x = torch.arange(3.0) + rank
    facts = {}
    for it in range(3):
        for p in params:
            p.grad = None if it < 2 else (p.grad.zero_() if p.grad is not None else None)   # step 2: set_to_none=False
        (extra * (rank + 1.0)).sum().backward()            # e.g. an ml_update pass before the armed backward
        red.arm()
        ((used[0] @ x).sum() * (rank + 1.0) + (used[1] ** 2).sum() + extra.sum()).backward()
        red.launch()
        red.wait()
        facts[it] = dict(unused_none=unused.grad is None,
                         in_bucket=[bool(red._is_view(red._index[id(p)])) for p in (used[0], used[1], extra)],
                         g0=used[0].grad.clone().numpy(), g1=used[1].grad.clone().numpy(), ge=extra.grad.clone().numpy())
    if rank == 0:
        out_q.put((facts, used[1].detach().numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize('overlap', [False, True])
def test_flat_buckets_keep_gradient_semantics(overlap):
    """Gradients live in the persistent flat buckets after the exchange; a parameter no pass touched keeps a None
    gradient (the optimizer must skip it as in a single-process run); a gradient accumulated by an earlier backward
    of the same step is carried into the bucket; in-place zeroing (set_to_none=False) between steps works."""
    import numpy as np
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29650 + (os.getpid() % 300) + (7 if overlap else 0)
    procs = [ctx.Process(target=_worker3, args=(r, 2, port, q, overlap)) for r in range(2)]
    for p in procs:
        p.start()
    facts, w1 = q.get(timeout=100)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # averages over ranks 0 and 1: d/dW0 of sum(W0 x_r)(r+1) = outer(1, x_r)(r+1); x_0 = [0,1,2], x_1 = [1,2,3]
    g0 = (np.tile(np.arange(3.0), (5, 1)) * 1.0 + np.tile(np.arange(3.0) + 1.0, (5, 1)) * 2.0) / 2.0
    ge = np.full(3, ((1.0 + 1.0) + (2.0 + 1.0)) / 2.0)       # earlier backward (r + 1) + the armed one (1)
    for it in range(3):
        f = facts[it]
        assert f['unused_none'], 'step %d: untouched parameter got a gradient' % it
        assert all(f['in_bucket']), 'step %d: gradients are not slices of the flat buckets' % it
        np.testing.assert_allclose(f['g0'], g0, rtol=1e-6)
        np.testing.assert_allclose(f['g1'], 2.0 * w1, rtol=1e-6)
        np.testing.assert_allclose(f['ge'], ge, rtol=1e-6)
