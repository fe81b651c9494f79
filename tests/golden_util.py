"""Shared helpers for the golden-fixture tests (CPU oracle tests and GPU parity tests)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


class Fixture(object):
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
        self.params = json.loads(str(self.z['params_json']))

    def __contains__(self, k):
        return k in self.z.files

    def np(self, k):
        return self.z[k]

    def t(self, k, device=None):
        a = self.z[k]
        t = torch.from_numpy(np.array(a, copy=True, order='C'))
        return t.to(device) if device is not None else t

    def scalar(self, k):
        return self.z[k].item()

    def json(self, k):
        return json.loads(str(self.z[k]))

    def group(self, prefix, device=None):
        out = {}
        for k in self.z.files:
            if k.startswith(prefix):
                out[k[len(prefix):]] = self.t(k, device)
        return out

    def noise_list(self, device=None):
        return [self.t('u%d' % i, device) for i in range(int(self.z['n_noise']))]


def rel_err(a, b):
    """max|a-b| / max(|b|, tiny): the max-normalised error SURVEY.md Appendix B uses."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, (tuple(a.shape), tuple(b.shape))
    den = max(b.abs().max().item(), 1e-30)
    return (a - b).abs().max().item() / den


def assert_close(a, b, tol, what=''):
    e = rel_err(a, b)
    assert e <= tol, '%s: max-normalised error %.3e > %.1e' % (what, e, tol)


def assert_grads_close(named_params, ref_grads, tol, what=''):
    worst = (0.0, None)
    for n, p in named_params:
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        r = ref_grads[n]
        if r.abs().max().item() == 0.0:
            assert g.abs().max().item() <= 1e-12, '%s grad %s should be zero' % (what, n)
            continue
        e = rel_err(g, r)
        if e > worst[0]:
            worst = (e, n)
    assert worst[0] <= tol, '%s: worst grad %s max-normalised error %.3e > %.1e' % (what, worst[1], worst[0], tol)
