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
