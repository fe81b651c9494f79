"""GPU: the generation loop as one persistent launch (a4nt_gen_persist_fwd) against the per-step launches
(a4nt_gen_loop_fwd, itself pinned to the oracle by the golden and baseline-dims tests): the same arithmetic per
element, so tokens, straight-through values, lengths and every saved tensor must agree bit for bit."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from golden_util import assert_close

pytestmark = pytest.mark.gpu


def _model(impl, V, E, H, B, seed=3):
    gp = dict(vocabulary_size=V, embedding_size=E, enc_hidden_size=H, dec_hidden_size=H, enc_hidden_depth=1,
              dec_hidden_depth=1, en_residual_conn=0, pad_auth_vec=10, num_output_layers=2, batch_size=B, gumbel_hard=True,
              softmax_scale=3.0, maxpoolrnn=0, drop_prob_emb=0., drop_prob_encoder=0., drop_prob_decoder=0., noise_seed=5)
    torch.manual_seed(seed)
    return impl.CharTranslator(gp).cuda().train()


def _batch(V, T, B, seed):
    g = torch.Generator().manual_seed(seed)
    lens = sorted([int(v) for v in torch.randint(2, T + 1, (B,), generator=g)], reverse=True)
    lens[0] = T
    x = torch.randint(1, V, (T, B), generator=g)
    for b, l in enumerate(lens):
        x[l:, b] = 0
    return x, lens


@pytest.mark.timeout(300)
@pytest.mark.parametrize('V,E,H,B,n_max,end_c,soft', [
    (300, 32, 64, 5, 9, -1, True),
    (300, 32, 64, 40, 21, 7, True),
    (1000, 64, 128, 200, 12, -1, True),
    (1000, 64, 128, 33, 12, -1, False),
    (997, 48, 256, 129, 18, 11, True),
    (10000, 512, 512, 256, 24, -1, True),
    (10000, 512, 512, 256, 37, 7, True),
    (10000, 512, 512, 100, 10, -1, False),
])
def test_persistent_generation_equals_step_launches(V, E, H, B, n_max, end_c, soft):
    import a4nt_amd as impl
    from a4nt_amd import gen_decode
    gen = _model(impl, V, E, H, B)
    T = 14
    x, lens = _batch(V, T, B, 1)
    auths = torch.zeros(B, dtype=torch.long)
    if end_c >= 0:
        with torch.no_grad():
            gen.decoder_b[end_c] += 1.5
    res = []
    for mode in (False, True):
        gen_decode.PERSIST = mode
        try:
            gen.zero_grad()
            gen._noise_step = 0
            s, gl, co = gen.forward_advers_gen(x, lens, soft_samples=soft, temp=0.5, n_max=n_max, end_c=end_c, auths=auths,
                                               compact=True)
            grads = {}
            if soft:
                emb = gen_decode.sample_embed(s, gen.char_emb.weight)
                w = torch.linspace(0.5, 1.5, emb.shape[-1], device=emb.device)
                ((emb * w).sum() / emb.shape[1]).backward()
                grads = {k: p.grad.clone() for k, p in gen.named_parameters() if p.grad is not None}
                res.append((s.tok.clone(), gl.clone(), s.val.clone(), grads))
            else:
                res.append((torch.stack(list(co)).clone(), gl.clone(), None, grads))
        finally:
            gen_decode.PERSIST = False
    impl.check_timeouts()
    (tok_a, gl_a, val_a, g_a), (tok_b, gl_b, val_b, g_b) = res
    assert torch.equal(tok_a, tok_b), 'tokens differ'
    assert torch.equal(gl_a, gl_b)
    if soft:
        assert torch.equal(val_a, val_b), 'straight-through values differ'
        if end_c >= 0:
            assert int(gl_b.min()) < n_max
        assert set(g_a) == set(g_b)
        for k in g_a:
            assert_close(g_b[k], g_a[k], 1e-6, 'grad ' + k)      # embed_bwd scatter-adds with atomics: order varies
