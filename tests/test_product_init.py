"""CPU: the PRODUCT's constructors (a4nt_amd.CharTranslator / CharLstm / CharCNN -- parameter containers only, no
kernel runs) initialise exactly like the reference's: same state_dict keys and shapes, and under the same
torch.manual_seed the same values bit for bit (nn.LSTM / nn.Embedding / nn.Conv1d default draws in the reference's
construction order, then init_weights: decoder_W ~ U(+-1e-3) resp. U(+-1.73/sqrt(fan_in)), zero biases, forget-gate
bias fill over indices H+1..2H-1 with index H untouched).  Fixture: tests/golden/init_full.npz and init_bias.npz,
recorded from the reference's own classes."""
import numpy as np
import torch

import scenarios as sc
from golden_util import Fixture


def test_product_init_matches_reference_bit_for_bit():
    import a4nt_amd
    sc.run_init(a4nt_amd, 'product', extra={'device': 'cpu'})      # parameters only: no kernel runs


def test_product_init_bias_fixture():
    import a4nt_amd
    fx = Fixture('init_bias')
    H = fx.params['enc_hidden_size']
    torch.manual_seed(10)                                   # oracle/gen_golden.py fx_init
    m = a4nt_amd.CharTranslator(dict(fx.params, device='cpu'))
    e = a4nt_amd.CharLstm(dict(vocabulary_size=50, embedding_size=24, hidden_size=H, hidden_depth=2, en_residual_conn=0,
                               num_output_layers=2, mode='classifier', maxpoolrnn=0, bidir=False, drop_prob_encoder=0.0,
                               drop_prob_decoder=0.0, device='cpu'))
    for key, got, val in (('g_enc_bias_ih', m.enc_rec_layers.bias_ih_l0, 1.0), ('g_dec_bias_hh', m.dec_rec_layers.bias_hh_l0, 1.0),
                          ('e_bias_ih_l1', e.rec_layers.bias_ih_l1, 2.0)):
        want = fx.np(key)
        got = got.detach().numpy()
        assert np.array_equal(got, want), key
        assert np.all(got[H + 1:2 * H] == val) and got[H] != val          # the reference's off-by-one is kept
    assert abs(float(m.decoder_W.abs().max()) - float(fx.scalar('g_decoder_W_absmax'))) == 0.0
    assert abs(float(e.decoder_W.abs().max()) - float(fx.scalar('e_decoder_W_absmax'))) == 0.0
