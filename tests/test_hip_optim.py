"""GPU: the multi-tensor optimizer step (csrc/optim.hip: ops.clip_grad_norm_, ops.RMSprop) against
torch.nn.utils.clip_grad_norm_ + torch.optim.RMSprop on the same tensors (the reference's optimizers,
train_adversarial_translator.py:306-312, :730)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from golden_util import assert_close

pytestmark = pytest.mark.gpu
SHAPES = [(10001, 512), (2048, 1034), (2048,), (7,), (16385,), (3, 5, 11), (512, 10001), (1,)]


def _params(seed, scale):
    g = torch.Generator().manual_seed(seed)
    ps = []
    for i, sh in enumerate(SHAPES):
        p = torch.randn(*sh, generator=g).cuda().requires_grad_(True)
        p.grad = (torch.randn(*sh, generator=g) * scale * (10.0 ** (-(i % 3)))).cuda()
        ps.append(p)
    # a gradient that is a view into a flat bucket at an odd offset (the data-parallel reducer's layout)
    flat = (torch.randn(1000 + 333, generator=g) * scale).cuda()
    p = torch.randn(333, generator=g).cuda().requires_grad_(True)
    p.grad = flat[1000:]
    ps.append(p)
    return ps


@pytest.mark.parametrize('scale,max_norm', [(1e-4, 10.0), (1.0, 10.0), (0.0, 10.0)])
def test_clip_grad_norm(scale, max_norm):
    import a4nt_amd
    from a4nt_amd import ops
    a, b = _params(1, scale), _params(1, scale)
    na = torch.nn.utils.clip_grad_norm_(a, max_norm)
    nb = ops.clip_grad_norm_(b, max_norm)
    assert_close(nb, na, 2e-6, 'total norm')
    for x, y in zip(a, b):
        assert_close(y.grad, x.grad, 2e-6, 'clipped gradient')
    if scale == 1.0:
        assert float(na) > max_norm          # the clipping branch ran
    if scale == 1e-4:
        assert float(na) < max_norm and all(torch.equal(x.grad, y.grad) for x, y in zip(a, b))   # untouched


@pytest.mark.parametrize('wd', [0.0, 0.01])
def test_rmsprop_steps(wd):
    import a4nt_amd
    from a4nt_amd import ops
    a, b = _params(2, 0.1), _params(2, 0.1)
    oa = torch.optim.RMSprop([{'params': a[:4]}, {'params': a[4:], 'weight_decay': wd}], lr=1e-3, alpha=0.99, eps=1e-8)
    ob = ops.RMSprop([{'params': b[:4]}, {'params': b[4:], 'weight_decay': wd}], lr=1e-3, alpha=0.99, eps=1e-8)
    g = torch.Generator().manual_seed(9)
    for it in range(3):
        for x, y in zip(a, b):
            d = (torch.randn(x.shape, generator=g) * 0.1).cuda()
            x.grad.copy_(d)
            y.grad.copy_(d)
        if it == 2:
            a[1].grad, b[1].grad = None, None            # a parameter the pass did not touch keeps its state
        oa.step()
        ob.step()
    for i, (x, y) in enumerate(zip(a, b)):
        assert_close(y.detach(), x.detach(), 2e-6, 'parameter %d' % i)
        assert_close(ob.state[y]['square_avg'], oa.state[x]['square_avg'], 2e-6, 'square_avg %d' % i)
        assert float(ob.state[y]['step']) == float(oa.state[x]['step'])
    # same state_dict layout: a torch optimizer loads it
    oc = torch.optim.RMSprop([{'params': a[:4]}, {'params': a[4:], 'weight_decay': wd}], lr=1e-3, alpha=0.99, eps=1e-8)
    oc.load_state_dict(ob.state_dict())


def test_rmsprop_falls_back_for_other_options():
    import a4nt_amd
    from a4nt_amd import ops
    a, b = _params(3, 0.1), _params(3, 0.1)
    oa = torch.optim.RMSprop(a, lr=1e-3, momentum=0.9, centered=True)
    ob = ops.RMSprop(b, lr=1e-3, momentum=0.9, centered=True)
    oa.step()
    ob.step()
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize('N,C,col,target,weighted', [(128, 2, 1, 1.0, False), (128, 2, 0, 0.0, False), (37, 3, 2, 1.0, True),
                                                     (2, 2, 0, 0.0, True), (700, 2, 1, 0.0, False)])
def test_bce_with_logits_col(N, C, col, target, weighted):
    """ops.bce_with_logits_col against F.binary_cross_entropy_with_logits on the column view (loss and gradient)."""
    import a4nt_amd
    from a4nt_amd import ops
    g = torch.Generator().manual_seed(N + col)
    x0 = (torch.randn(N, C, generator=g) * 4.0)
    x0[0, col] = 30.0                      # saturated either way
    x0[N - 1, col] = -30.0
    w = torch.rand(N, generator=g).cuda() if weighted else None
    a = x0.clone().cuda().requires_grad_(True)
    b = x0.clone().cuda().requires_grad_(True)
    la = torch.nn.functional.binary_cross_entropy_with_logits(a[:, col], torch.full((N,), target, device='cuda'), weight=w)
    lb = ops.bce_with_logits_col(b, col, target, w)
    (3.0 * la).backward()
    (3.0 * lb).backward()
    assert_close(lb, la, 2e-6, 'loss')
    assert_close(b.grad, a.grad, 2e-6, 'gradient')
