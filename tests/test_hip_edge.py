"""GPU: degenerate shapes -- one sentence, one step, every sentence of length 1, one very short sentence next to
a full one, a generation cut to a single step -- on the HIP-backed modules against the oracle restatement with
the same weights.  (The reference's driver never builds such batches, so no fixture recorded from it covers
them; the oracle itself is pinned by tests/test_oracle_golden.py.)  Same tolerances as test_hip_golden.py."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence as pps

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from golden_util import assert_close, assert_grads_close

pytestmark = pytest.mark.gpu
TOL_OUT, TOL_GRAD = 1e-5, 5e-5
V, E, H = 50, 24, 64
GP = dict(vocabulary_size=V, embedding_size=E, enc_hidden_size=H, dec_hidden_size=H, enc_hidden_depth=1,
          dec_hidden_depth=1, en_residual_conn=0, pad_auth_vec=10, num_output_layers=2, batch_size=999, gumbel_hard=True,
          softmax_scale=3.0, maxpoolrnn=0, drop_prob_emb=0., drop_prob_encoder=0., drop_prob_decoder=0.)
EP = dict(vocabulary_size=V, embedding_size=E, hidden_size=H, hidden_depth=2, en_residual_conn=0, num_output_layers=2,
          mode='classifier', maxpoolrnn=0, bidir=False, drop_prob_encoder=0., drop_prob_decoder=0.)
SHAPES = [(1, [1]), (1, [5]), (3, [1, 1, 1]), (2, [6, 1]), (17, [4] + [1] * 16)]     # (B, lens sorted descending)


def _pair(cls_name, params):
    import a4nt_amd
    import oracle
    torch.manual_seed(11)
    ref = getattr(oracle, cls_name)(params).train()
    hip = getattr(a4nt_amd, cls_name)(params)
    missing = hip.load_state_dict(ref.state_dict())
    assert not missing.missing_keys and not missing.unexpected_keys
    return ref, hip.cuda().train(), a4nt_amd


def _grads(m):
    return {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in m.named_parameters()}


def _ids(B, lens, seed):
    g = torch.Generator().manual_seed(seed)
    T = max(lens)
    x = torch.randint(1, V, (T, B), generator=g)
    for b, l in enumerate(lens):
        x[l:, b] = 0
    return x


@pytest.mark.parametrize('B,lens', SHAPES)
def test_mltrain_edge(B, lens):
    ref, hip, impl = _pair('CharTranslator', GP)
    inps, targs = _ids(B, lens, 1), _ids(B, lens, 2)
    auths = torch.arange(B) % 2
    out = {}
    for tag, m in (('ref', ref), ('hip', hip)):
        scores, _ = m.forward_mltrain(inps, lens, inps, lens, auths=auths)
        loss = F.cross_entropy(pps(scores, lens)[0], pps(targs.to(scores.device), lens)[0])
        loss.backward()
        out[tag] = (scores.detach().cpu(), loss.detach().cpu())
    impl.check_timeouts()
    assert_close(out['hip'][0], out['ref'][0], TOL_OUT, 'scores')
    assert_close(out['hip'][1], out['ref'][1], TOL_OUT, 'loss')
    assert_grads_close(hip.named_parameters(), _grads(ref), TOL_GRAD, 'mltrain edge')


@pytest.mark.parametrize('B,lens', SHAPES)
def test_classifier_edge(B, lens):
    ref, hip, impl = _pair('CharLstm', EP)
    x = _ids(B, lens, 3)
    labels = torch.arange(B) % 2
    out = {}
    for tag, m in (('ref', ref), ('hip', hip)):
        logits, feat = m.forward_classify(x, lens=lens)
        loss = F.cross_entropy(logits, labels.to(logits.device))
        loss.backward()
        out[tag] = (logits.detach().cpu(), feat.detach().cpu())
    impl.check_timeouts()
    assert_close(out['hip'][0], out['ref'][0], TOL_OUT, 'logits')
    assert_close(out['hip'][1], out['ref'][1], TOL_OUT, 'features')
    assert_grads_close(hip.named_parameters(), _grads(ref), TOL_GRAD, 'classifier edge')


@pytest.mark.parametrize('B,lens,n_max,end_c', [(1, [1], 1, -1), (1, [3], 4, -1), (3, [2, 1, 1], 1, 7), (5, [4, 2, 1, 1, 1], 6, 7)])
def test_generation_edge(B, lens, n_max, end_c):
    """Free-running generation of n_max steps (1 = a single sampling step) with the Gumbel uniforms pinned."""
    ref, hip, impl = _pair('CharTranslator', GP)
    with torch.no_grad():                       # sharpen the output layer so that tokens do not hinge on rounding
        ref.decoder_W.mul_(30.0)
        hip.decoder_W.mul_(30.0)
    inps = _ids(B, lens, 4)
    auths = torch.arange(B) % 2
    g = torch.Generator().manual_seed(9)
    noise = [torch.rand(B, ref.decoder_W.shape[1], generator=g) for _ in range(n_max)]
    R = torch.randn(n_max, B, ref.decoder_W.shape[1], generator=g)
    out = {}
    for tag, m, dev in (('ref', ref, 'cpu'), ('hip', hip, 'cuda')):
        m.set_noise([u.clone().to(dev) for u in noise])
        s, gl, co = m.forward_advers_gen(inps, lens, soft_samples=True, temp=0.5, n_max=n_max, end_c=end_c, auths=auths)
        samp = torch.stack(list(s), 0)
        (samp * R[:samp.shape[0]].to(dev)).sum().backward()
        out[tag] = (samp.detach().cpu(), torch.stack(list(co), 0).cpu().long(), gl.cpu().long())
    impl.check_timeouts()
    assert torch.equal(out['hip'][1], out['ref'][1]), 'sampled tokens differ'
    assert torch.equal(out['hip'][2], out['ref'][2]), 'generated lengths differ'
    assert_close(out['hip'][0], out['ref'][0], TOL_OUT, 'samples')
    assert_grads_close(hip.named_parameters(), _grads(ref), TOL_GRAD, 'generation edge')


def test_generation_ends_long_before_n_max():
    """Every sentence emits the end token within a few steps of a 60-step budget: the loop stops at one of its
    periodic looks (one period late) and the outputs are trimmed to what the reference's per-step check gives."""
    ref, hip, impl = _pair('CharTranslator', GP)
    end_c = 7
    with torch.no_grad():
        for m in (ref, hip):
            m.decoder_W.mul_(30.0)
            m.decoder_b[end_c] += 2.0          # the end token wins early, not necessarily at once
    B, lens, n_max = 5, [4, 3, 3, 2, 1], 60
    inps = _ids(B, lens, 6)
    auths = torch.arange(B) % 2
    g = torch.Generator().manual_seed(10)
    Vp = ref.decoder_W.shape[1]
    noise = [torch.rand(B, Vp, generator=g) for _ in range(n_max)]
    R = torch.randn(n_max, B, Vp, generator=g)
    out = {}
    for tag, m, dev in (('ref', ref, 'cpu'), ('hip', hip, 'cuda')):
        m.set_noise([u.clone().to(dev) for u in noise])
        s, gl, co = m.forward_advers_gen(inps, lens, soft_samples=True, temp=0.5, n_max=n_max, end_c=end_c, auths=auths)
        samp = torch.stack(list(s), 0)
        (samp * R[:samp.shape[0]].to(dev)).sum().backward()      # (a plain sum has a zero gradient: softmax rows sum to 1)
        out[tag] = (samp.detach().cpu(), torch.stack(list(co), 0).cpu().long(), gl.cpu().long())
    impl.check_timeouts()
    assert int(out['ref'][2].max()) < 30, 'scenario no longer ends early: %s' % out['ref'][2]
    assert out['hip'][0].shape == out['ref'][0].shape
    assert torch.equal(out['hip'][1], out['ref'][1]) and torch.equal(out['hip'][2], out['ref'][2])
    assert_close(out['hip'][0], out['ref'][0], TOL_OUT, 'samples')
    assert_grads_close(hip.named_parameters(), _grads(ref), TOL_GRAD, 'early end')


# ------------------------------------------------------------------------------------------
# hidden sizes the kernels are not instantiated for (nn.LSTM in the reference takes any:
# models/char_translator.py:74-90, models/char_lstm.py:55-59): run zero-padded at the next instantiated size
@pytest.mark.parametrize('Hx', [96, 300, 20])
def test_any_hidden_size(Hx):
    gp = dict(GP, enc_hidden_size=Hx, dec_hidden_size=Hx)
    ep = dict(EP, hidden_size=Hx)
    B, lens = 6, [7, 7, 5, 4, 2, 1]
    # classifier, two stacked layers
    ref, hip, impl = _pair('CharLstm', ep)
    assert hip.state_dict()['rec_layers.weight_hh_l1'].shape == (4 * Hx, Hx)         # state_dict keeps the real size
    x = _ids(B, lens, 13)
    labels = torch.arange(B) % 2
    out = {}
    for tag, m in (('ref', ref), ('hip', hip)):
        logits, feat = m.forward_classify(x, lens=lens)
        F.cross_entropy(logits, labels.to(logits.device)).backward()
        out[tag] = (logits.detach().cpu(), feat.detach().cpu())
    impl.check_timeouts()
    assert out['hip'][1].shape == (B, Hx)
    assert_close(out['hip'][0], out['ref'][0], TOL_OUT, 'H=%d logits' % Hx)
    assert_close(out['hip'][1], out['ref'][1], TOL_OUT, 'H=%d features' % Hx)
    assert_grads_close(hip.named_parameters(), _grads(ref), TOL_GRAD, 'H=%d classifier' % Hx)
    # translator: teacher forced, then free-running generation with pinned noise (fused decode loop)
    ref, hip, impl = _pair('CharTranslator', gp)
    inps, targs = _ids(B, lens, 14), _ids(B, lens, 15)
    auths = torch.arange(B) % 2
    for tag, m in (('ref', ref), ('hip', hip)):
        scores, _ = m.forward_mltrain(inps, lens, inps, lens, auths=auths)
        F.cross_entropy(pps(scores, lens)[0], pps(targs.to(scores.device), lens)[0]).backward()
        out[tag] = scores.detach().cpu()
    impl.check_timeouts()
    assert_close(out['hip'], out['ref'], TOL_OUT, 'H=%d scores' % Hx)
    assert_grads_close(hip.named_parameters(), _grads(ref), TOL_GRAD, 'H=%d mltrain' % Hx)
    with torch.no_grad():
        ref.decoder_W.mul_(30.0)
        hip.decoder_W.mul_(30.0)
    ref.zero_grad(); hip.zero_grad()
    g = torch.Generator().manual_seed(19)
    n_max, Vp = 6, ref.decoder_W.shape[1]
    noise = [torch.rand(B, Vp, generator=g) for _ in range(n_max)]
    R = torch.randn(n_max, B, Vp, generator=g)
    for tag, m, dev in (('ref', ref, 'cpu'), ('hip', hip, 'cuda')):
        m.set_noise([u.clone().to(dev) for u in noise])
        s, gl, co = m.forward_advers_gen(inps, lens, soft_samples=True, temp=0.5, n_max=n_max, end_c=7, auths=auths)
        samp = torch.stack(list(s), 0)
        (samp * R[:samp.shape[0]].to(dev)).sum().backward()
        out[tag] = (samp.detach().cpu(), torch.stack(list(co), 0).cpu().long(), gl.cpu().long())
    impl.check_timeouts()
    assert torch.equal(out['hip'][1], out['ref'][1]) and torch.equal(out['hip'][2], out['ref'][2])
    assert_close(out['hip'][0], out['ref'][0], TOL_OUT, 'H=%d samples' % Hx)
    assert_grads_close(hip.named_parameters(), _grads(ref), TOL_GRAD, 'H=%d generation' % Hx)
    with torch.no_grad():
        toks = []
        for m in (ref, hip):
            m.eval()
            _, gl, co = m.forward_advers_gen(inps, lens, soft_samples=False, n_max=n_max, end_c=-1, auths=auths)
            toks.append(torch.stack(list(co), 0).cpu().long())
    assert torch.equal(toks[0], toks[1]), 'H=%d greedy tokens' % Hx
