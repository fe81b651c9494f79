"""GPU: properties that do not depend on the size, checked at the benchmark's full size (V=10001, E=512, H=512,
T=100, 128 sentences; BASELINE.json configs[2]) where the oracle would take minutes:

  * rows are independent: generating for the first half of a batch alone gives, bit for bit, the tokens, values and
    lengths the full batch gives for those rows (the Philox stream is keyed by row, column and step);
  * padding does not matter: a ragged batch padded to T=100 and the same batch trimmed to its longest sentence give
    the same reconstruction loss and gradients;
  * batch-split linearity of the step: the token-weighted gradients of two half batches add up to the gradient of
    the whole batch (what the data-parallel all-reduce relies on).
"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from golden_util import assert_close

pytestmark = pytest.mark.gpu
V, E, H, T, B, A = 10000, 512, 512, 100, 128, 10
GP = dict(vocabulary_size=V, embedding_size=E, enc_hidden_size=H, dec_hidden_size=H, enc_hidden_depth=1,
          dec_hidden_depth=1, en_residual_conn=0, pad_auth_vec=A, num_output_layers=2, batch_size=B, gumbel_hard=True,
          softmax_scale=3.0, maxpoolrnn=0, drop_prob_emb=0., drop_prob_encoder=0., drop_prob_decoder=0., noise_seed=5)


@pytest.fixture(scope='module')
def impl():
    import a4nt_amd
    return a4nt_amd


@pytest.fixture(scope='module')
def gen(impl):
    torch.manual_seed(3)
    return impl.CharTranslator(GP).cuda().train()


def _ragged(B_, seed):
    g = torch.Generator().manual_seed(seed)
    lens = sorted([int(v) for v in torch.randint(4, T + 1, (B_,), generator=g)], reverse=True)
    lens[0] = T
    x = torch.randint(1, V, (T, B_), generator=g)
    for b, l in enumerate(lens):
        x[l:, b] = 0
    return x, lens


@pytest.mark.timeout(300)
def test_generation_rows_are_independent(impl, gen):
    x, lens = _ragged(B, 1)
    auths = torch.zeros(B, dtype=torch.long)
    res = []
    for nb in (B, B // 2):
        gen._noise_step = 0
        with torch.no_grad():
            s, gl, co = gen.forward_advers_gen(x[:, :nb].contiguous(), lens[:nb], soft_samples=True, temp=0.5, n_max=T,
                                               end_c=-1, auths=auths[:nb], compact=True)
        res.append((s.tok.clone(), s.val.clone(), gl.clone()))
    impl.check_timeouts()
    h = B // 2
    assert torch.equal(res[0][0][:, :h], res[1][0]), 'tokens of the first half change with the batch size'
    assert torch.equal(res[0][1][:, :h], res[1][1])
    assert torch.equal(res[0][2][:h], res[1][2])
    assert int(res[0][0].max()) <= V and int(res[0][0].min()) >= 0
    assert len(torch.unique(res[0][0])) > 100        # the Philox stream is not degenerate


@pytest.mark.timeout(300)
def test_padding_does_not_matter(impl, gen):
    g = torch.Generator().manual_seed(2)
    lens = sorted([int(v) for v in torch.randint(3, 41, (B,), generator=g)], reverse=True)     # longest sentence 40 < T
    x = torch.randint(1, V, (T, B), generator=g)
    y = torch.randint(1, V, (T, B), generator=g)
    for b, l in enumerate(lens):
        x[l:, b] = 0
        y[l:, b] = 0
    auths = torch.arange(B) % 2
    out = []
    for Tt in (T, max(lens)):
        gen.zero_grad()
        loss = gen.forward_mltrain_loss(x[:Tt], lens, x[:Tt], lens, y[:Tt], auths=auths)
        loss.backward()
        out.append((loss.detach().clone(), {n: p.grad.clone() for n, p in gen.named_parameters()}))
    impl.check_timeouts()
    assert_close(out[1][0], out[0][0], 1e-6, 'loss')
    for n in out[0][1]:
        assert_close(out[1][1][n], out[0][1][n], 2e-5, 'grad ' + n)


@pytest.mark.timeout(300)
def test_half_batches_add_up(impl, gen):
    x, lens = _ragged(B, 4)
    y, _ = _ragged(B, 5)
    auths = torch.arange(B) % 2
    ntok = float(sum(lens))

    def grads(r0, r1):
        gen.zero_grad()
        # sentences sorted by length inside each half, as the data provider hands them over
        loss = gen.forward_mltrain_loss(x[:, r0:r1].contiguous(), lens[r0:r1], x[:, r0:r1].contiguous(), lens[r0:r1],
                                        y[:, r0:r1].contiguous(), auths=auths[r0:r1])
        (loss * (sum(lens[r0:r1]) / ntok)).backward()        # mean over tokens -> share of the whole batch's tokens
        return {n: p.grad.clone() for n, p in gen.named_parameters()}

    whole, a, b = grads(0, B), grads(0, B // 2), grads(B // 2, B)
    impl.check_timeouts()
    for n in whole:
        assert_close(a[n] + b[n], whole[n], 2e-5, 'grad ' + n)


@pytest.mark.timeout(300)
@pytest.mark.parametrize('end_c', [-1, 7])
def test_native_loops_equal_python_loops(impl, gen, end_c):
    """The library's C++ step loops (sampler finalisation merged into the next decoder step's prologue, one finalize
    launch per chunk) against the Python per-step loops (one finalize launch per step): same tokens and lengths, bit
    for bit; straight-through values, loss and gradients to rounding (the merge adds a row's strips in another
    order).  With an end token the generation runs in chunks of CHECK_EVERY steps and some rows stop early."""
    from a4nt_amd import gen_decode
    x, lens = _ragged(B, 11)
    auths = torch.zeros(B, dtype=torch.long)
    n_max = 37
    res = []
    if end_c >= 0:
        with torch.no_grad():
            gen.decoder_b[end_c] += 1.0        # ~4 % per step: rows end at different steps, a few run to n_max
    for py in (True, False):
        gen_decode.PY_LOOPS = py
        try:
            gen.zero_grad()
            gen._noise_step = 0
            s, gl, co = gen.forward_advers_gen(x, lens, soft_samples=True, temp=0.5, n_max=n_max, end_c=end_c, auths=auths,
                                               compact=True)
            emb = gen_decode.sample_embed(s, gen.char_emb.weight)   # [n, B, E]: what a consumer of the samples computes
            w = torch.linspace(0.5, 1.5, emb.shape[-1], device=emb.device)
            loss = (emb * w).sum() / emb.shape[1]
            loss.backward()
            res.append((s.tok.clone(), gl.clone(), s.val.clone(), loss.detach().clone(),
                        {k: p.grad.clone() for k, p in gen.named_parameters() if p.grad is not None}))
        finally:
            gen_decode.PY_LOOPS = False
    if end_c >= 0:
        with torch.no_grad():
            gen.decoder_b[end_c] -= 1.0
    impl.check_timeouts()
    (tok_p, gl_p, val_p, loss_p, g_p), (tok_n, gl_n, val_n, loss_n, g_n) = res
    assert torch.equal(tok_p, tok_n) and torch.equal(gl_p, gl_n)
    if end_c >= 0:
        assert int(gl_n.min()) < n_max, 'no row ended early: the end-token branch was not exercised'
    assert_close(val_n, val_p, 1e-6, 'straight-through values')
    assert_close(loss_n, loss_p, 1e-6, 'loss')
    assert set(g_p) == set(g_n)
    for k in g_p:
        assert_close(g_n[k], g_p[k], 2e-5, 'grad ' + k)
