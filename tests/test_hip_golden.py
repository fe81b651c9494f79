"""GPU: the HIP-backed modules (a4nt_amd) replay the golden fixtures recorded from the reference's
own model files (tests/golden, oracle/gen_golden.py), through the C-ABI kernels.

Stated tolerance: max-normalised error <= 1e-5 on logits / scores / losses,
<= 5e-5 on every parameter gradient; sampled and greedy tokens and generated lengths exact.
"""
import pytest
import torch

import scenarios as sc

pytestmark = pytest.mark.gpu
TOL_OUT, TOL_GRAD = 1e-5, 5e-5


@pytest.fixture(scope='module')
def impl():
    import a4nt_amd
    return a4nt_amd


@pytest.mark.parametrize('name', sc.MLTRAIN)
def test_mltrain(impl, name):
    sc.run_mltrain(impl, 'cuda', name, TOL_OUT, TOL_GRAD)
    impl.check_timeouts()


def test_mltrain_soft_sortenc(impl):
    sc.run_mltrain_soft(impl, 'cuda', TOL_OUT, TOL_GRAD)


@pytest.mark.parametrize('name', sc.GEN_SOFT)
def test_gen_soft(impl, name):
    sc.run_gen_soft(impl, 'cuda', name, TOL_OUT, TOL_GRAD)
    impl.check_timeouts()


@pytest.mark.parametrize('name', sc.ENCODE)
def test_encode(impl, name):
    sc.run_encode(impl, 'cuda', name, TOL_OUT, TOL_GRAD)
    impl.check_timeouts()


def test_gen_greedy(impl):
    sc.run_gen_greedy(impl, 'cuda')


@pytest.mark.parametrize('name', sc.CLF_LSTM + sc.CLF_CNN)
def test_classifier(impl, name):
    sc.run_classifier(impl, 'cuda', name, TOL_OUT, TOL_GRAD)
    impl.check_timeouts()


@pytest.mark.parametrize('name', sc.ADV_STEP)
def test_adv_step(impl, name):
    sc.run_adv_step(impl, 'cuda', name, TOL_OUT, TOL_GRAD, tol_weights=2e-3)
    impl.check_timeouts()


@pytest.mark.parametrize('name', sc.CLF_HEADS)
def test_classifier_heads(impl, name):
    sc.run_classifier_heads(impl, 'cuda', name, TOL_OUT, TOL_GRAD)
    impl.check_timeouts()


def test_lm_multihead(impl):
    sc.run_lm_multihead(impl, 'cuda', TOL_OUT, TOL_GRAD)


@pytest.mark.parametrize('name', sc.LOSS_VARIANTS)
def test_loss_variants(impl, name):
    sc.run_loss_variant(impl, 'cuda', name, TOL_OUT, TOL_GRAD)
    impl.check_timeouts()


@pytest.mark.parametrize('name', sc.GSTEP)
def test_gstep_variants(impl, name):
    sc.run_gstep_variant(impl, 'cuda', name, TOL_OUT, TOL_GRAD)
    impl.check_timeouts()


def test_mltrain_advtarg(impl):
    sc.run_mltrain_advtarg(impl, 'cuda', TOL_OUT, TOL_GRAD)


def test_gstep_langloss(impl):
    sc.run_gstep_langloss(impl, 'cuda', TOL_OUT, TOL_GRAD)
    impl.check_timeouts()


def test_eval_translate(impl):
    sc.run_eval_translate(impl, 'cuda', impl.eval_utils.eval_forward_pass, TOL_OUT)
    impl.check_timeouts()


def test_mltrain_loss_matches_scores_path(impl):
    """forward_mltrain_loss (fused projection + NLL) == CrossEntropy over forward_mltrain scores."""
    from golden_util import Fixture, assert_close
    import torch.nn.functional as F
    from torch.nn.utils.rnn import pack_padded_sequence as pps
    fx = Fixture('mltrain_plain')
    m = impl.CharTranslator(fx.params)
    m.load_state_dict({k: v.cuda() for k, v in fx.group('sd.').items()})
    m.train()
    lens = fx.np('lens').tolist()
    loss = m.forward_mltrain_loss(fx.t('inps'), lens, fx.t('inps'), lens, fx.t('targs'), auths=fx.t('auths'))
    loss.backward()
    assert_close(loss, fx.t('loss'), TOL_OUT, 'fused CE loss')
    from golden_util import assert_grads_close
    assert_grads_close(m.named_parameters(), fx.group('grad.'), TOL_GRAD, 'fused CE')


@pytest.mark.parametrize('name', sc.ADV_STEP)
def test_fused_update_matches_separate_updates(impl, name):
    """fused_update (one generation of 2B rows for both updates, BPTT kept for the generator update's rows
    only) == d_update followed by g_update, on the fixture's weights, batches and Gumbel noise."""
    from golden_util import Fixture, assert_close
    fx = Fixture(name)
    ep = fx.json('eval_params_json')
    B = fx.t('src.inps').shape[1]
    c_aid = int(fx.scalar('c_aid'))
    end_c, maxlen, temp = int(fx.scalar('end_c')), int(fx.scalar('maxlen')), float(fx.scalar('temp'))
    mk = lambda nm, a: (fx.t(nm + '.inps'), fx.t(nm + '.targs'), torch.LongTensor([a] * B), fx.np(nm + '.lens').tolist())
    noise = fx.noise_list('cuda')
    n_d = int(fx.scalar('n_noise_d'))
    u_d, u_g = noise[:n_d], noise[n_d:]
    filler = torch.full_like(noise[0], 0.5)
    merged = [torch.cat([u_d[t] if t < len(u_d) else filler, u_g[t] if t < len(u_g) else filler], 0)
              for t in range(max(len(u_d), len(u_g)))]
    append = torch.zeros(1, B, fx.params['vocabulary_size'] + 1, device='cuda')
    append[:, :, 2] = 1.0

    def build():
        G = impl.CharTranslator(fx.params)
        G.load_state_dict({k: v.cuda() for k, v in fx.group('sdG0.').items()})
        E = impl.CharLstm(ep) if ep.get('modeltype', 'lstm') == 'lstm' else impl.CharCNN(ep)
        E.load_state_dict({k: v.cuda() for k, v in fx.group('sdE0.').items()})
        return (G.train(), E.train()) + tuple(impl.make_optimizers(G, E))

    G1, E1, oG1, oE1 = build()
    G1.set_noise([u.clone() for u in noise])
    rd1 = impl.d_update(G1, E1, oE1, mk('src', c_aid), mk('tgt', 1 - c_aid), c_aid, end_c, maxlen, temp)
    rg1 = impl.g_update(G1, E1, oG1, mk('gb', c_aid), c_aid, end_c, maxlen, temp, append, cycle_loss_type='ml')
    G2, E2, oG2, oE2 = build()
    G2.set_noise(merged)
    rd2, rg2 = impl.fused_update(G2, E2, oG2, oE2, mk('src', c_aid), mk('tgt', 1 - c_aid), mk('gb', c_aid), c_aid, end_c,
                                 maxlen, temp, append, cycle_loss_type='ml')
    impl.check_timeouts()
    assert torch.equal(torch.stack(list(rd1['char_outs'])), torch.stack(list(rd2['char_outs'])))
    assert torch.equal(torch.stack(list(rg1['char_outs'])), torch.stack(list(rg2['char_outs'])))
    assert torch.equal(rg1['gen_lens'].cpu(), rg2['gen_lens'].cpu())
    for k in ('loss', 'fake_logits', 'real_logits'):
        assert_close(rd2[k], rd1[k], TOL_OUT, name + ' D ' + k)
    for k in ('loss_gan', 'cyc_loss', 'grad_norm'):
        assert_close(rg2[k], rg1[k], TOL_GRAD, name + ' G ' + k)
    for (n1, p1), (_, p2) in zip(E1.named_parameters(), E2.named_parameters()):
        assert_close(p2, p1, TOL_OUT, name + ' D weight ' + n1)
    for (n1, p1), (_, p2) in zip(G1.named_parameters(), G2.named_parameters()):
        assert_close(p2.grad, p1.grad, TOL_GRAD, name + ' G grad ' + n1)
        assert_close(p2, p1, TOL_OUT, name + ' G weight ' + n1)


@pytest.mark.parametrize('name', sc.ADV_STEP)
@pytest.mark.parametrize('py_loops', [False, True])
def test_folded_sample_gradient_matches_time_batched(impl, name, py_loops):
    """Generator update with the reconstruction encoder's d(embedding) folded into the reverse loop's own product
    ((a + b) E^T, gen_decode.FOLD_SAME_EMBEDDING, the default) against every consumer contracted time-batched
    (a E^T + b E^T, the r02 arithmetic): same losses and tokens, gradients to the parity tolerance; with the library's
    C++ loop and with the Python per-step loop."""
    from golden_util import Fixture, assert_close
    from a4nt_amd import gen_decode
    fx = Fixture(name)
    ep = fx.json('eval_params_json')
    B = fx.t('gb.inps').shape[1]
    c_aid = int(fx.scalar('c_aid'))
    end_c, maxlen, temp = int(fx.scalar('end_c')), int(fx.scalar('maxlen')), float(fx.scalar('temp'))
    gb = (fx.t('gb.inps'), fx.t('gb.targs'), torch.LongTensor([c_aid] * B), fx.np('gb.lens').tolist())
    noise = fx.noise_list('cuda')[int(fx.scalar('n_noise_d')):]
    append = torch.zeros(1, B, fx.params['vocabulary_size'] + 1, device='cuda')
    append[:, :, 2] = 1.0
    res = []
    for fold in (False, True):
        G = impl.CharTranslator(fx.params)
        G.load_state_dict({k: v.cuda() for k, v in fx.group('sdG0.').items()})
        E = impl.CharLstm(ep) if ep.get('modeltype', 'lstm') == 'lstm' else impl.CharCNN(ep)
        E.load_state_dict({k: v.cuda() for k, v in fx.group('sdE0.').items()})
        G.train(), E.train()
        oG, _ = impl.make_optimizers(G, E)
        G.set_noise([u.clone() for u in noise])
        gen_decode.FOLD_SAME_EMBEDDING, gen_decode.PY_LOOPS = fold, py_loops
        try:
            rg = impl.g_update(G, E, oG, gb, c_aid, end_c, maxlen, temp, append, cycle_loss_type='ml', step=False)
        finally:
            gen_decode.FOLD_SAME_EMBEDDING, gen_decode.PY_LOOPS = True, False
        res.append((rg, {n: p.grad.clone() for n, p in G.named_parameters() if p.grad is not None}))
    impl.check_timeouts()
    (r0, g0), (r1, g1) = res
    assert torch.equal(torch.stack(list(r0['char_outs'])), torch.stack(list(r1['char_outs'])))
    for k in ('loss', 'loss_gan', 'cyc_loss'):
        assert_close(r1[k], r0[k], TOL_OUT, name + ' ' + k)
    assert_close(r1['grad_norm'], r0['grad_norm'], TOL_GRAD, name + ' grad norm')
    assert set(g0) == set(g1)
    for n in g0:
        assert_close(g1[n], g0[n], TOL_GRAD, name + ' grad ' + n)
