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
    from a4nt_amd import GradAllReducer
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
        red.launch()
        red.wait()
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
