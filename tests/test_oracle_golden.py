"""CPU: the oracle restatement (oracle/) reproduces the golden vectors that oracle/gen_golden.py
recorded from the reference's own model files.  This is what pins the oracle."""
import numpy as np
import pytest
import torch

import oracle
import scenarios as sc
from golden_util import Fixture

TOL_OUT = 2e-6      # same torch CPU kernels on both sides: only op-ordering noise is expected
TOL_GRAD = 2e-5


@pytest.mark.parametrize('name', ['gumbel_hard0', 'gumbel_hard1'])
def test_gumbel(name):
    sc.run_gumbel(oracle, 'cpu', name, TOL_OUT)


@pytest.mark.parametrize('name', sc.MLTRAIN)
def test_mltrain(name):
    sc.run_mltrain(oracle, 'cpu', name, TOL_OUT, TOL_GRAD)


def test_mltrain_soft_sortenc():
    sc.run_mltrain_soft(oracle, 'cpu', TOL_OUT, TOL_GRAD)


@pytest.mark.parametrize('name', sc.GEN_SOFT)
def test_gen_soft(name):
    sc.run_gen_soft(oracle, 'cpu', name, TOL_OUT, TOL_GRAD)


def test_gen_greedy():
    sc.run_gen_greedy(oracle, 'cpu')


@pytest.mark.parametrize('name', sc.CLF_LSTM + sc.CLF_CNN)
def test_classifier(name):
    sc.run_classifier(oracle, 'cpu', name, TOL_OUT, TOL_GRAD)


@pytest.mark.parametrize('name', sc.ADV_STEP)
def test_adv_step(name):
    sc.run_adv_step(oracle, 'cpu', name, TOL_OUT, TOL_GRAD)


@pytest.mark.parametrize('name', sc.CLF_HEADS)
def test_classifier_heads(name):
    sc.run_classifier_heads(oracle, 'cpu', name, TOL_OUT, TOL_GRAD)


def test_lm_multihead():
    sc.run_lm_multihead(oracle, 'cpu', TOL_OUT, TOL_GRAD)


@pytest.mark.parametrize('name', sc.LOSS_VARIANTS)
def test_loss_variants(name):
    sc.run_loss_variant(oracle, 'cpu', name, TOL_OUT, TOL_GRAD)


@pytest.mark.parametrize('name', sc.GSTEP)
def test_gstep_variants(name):
    sc.run_gstep_variant(oracle, 'cpu', name, TOL_OUT, TOL_GRAD)


def test_mltrain_advtarg():
    sc.run_mltrain_advtarg(oracle, 'cpu', TOL_OUT, TOL_GRAD)


def test_gstep_langloss():
    sc.run_gstep_langloss(oracle, 'cpu', TOL_OUT, TOL_GRAD)


def test_eval_translate():
    sc.run_eval_translate(oracle, 'cpu', oracle.eval_forward_pass, TOL_OUT)


@pytest.mark.parametrize('name', sc.ENCODE)
def test_encode(name):
    sc.run_encode(oracle, 'cpu', name, TOL_OUT, TOL_GRAD)


def test_init_full():
    sc.run_init(oracle, 'oracle')


def test_init_quirks():
    """Forget-gate bias fill covers indices H+1..2H-1 only (index H untouched), value 1.0 for the
    translator and 2.0 for the classifier; decoder_W ranges (char_translator.py:121-161,
    char_lstm.py:100-134)."""
    fx = Fixture('init_bias')
    H = fx.params['enc_hidden_size']
    for key, val in (('g_enc_bias_ih', 1.0), ('g_dec_bias_hh', 1.0), ('e_bias_ih_l1', 2.0)):
        b = fx.np(key)
        assert np.all(b[H + 1:2 * H] == val) and b[H] != val
    torch.manual_seed(0)
    m = oracle.CharTranslator(fx.params)
    b = m.enc_rec_layers.bias_ih_l0.detach().numpy()
    assert np.all(b[H + 1:2 * H] == 1.0) and b[H] != 1.0
    assert m.decoder_W.abs().max().item() <= 1e-3 and fx.scalar('g_decoder_W_absmax') <= 1e-3
    ep = dict(vocabulary_size=50, embedding_size=24, hidden_size=H, hidden_depth=2, num_output_layers=2,
              mode='classifier')
    e = oracle.CharLstm(ep)
    b = e.rec_layers.bias_hh_l1.detach().numpy()
    assert np.all(b[H + 1:2 * H] == 2.0) and b[H] != 2.0
    assert e.decoder_W.abs().max().item() <= 1.73 / np.sqrt(H) + 1e-6
