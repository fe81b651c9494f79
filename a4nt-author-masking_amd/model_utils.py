"""Helpers shared by the HIP-backed model classes (mirror of the reference's models/model_utils.py).

The reference runs its LSTMs over PackedSequence objects (pack_padded_sequence /
pad_packed_sequence, model_utils.py:14-28).  The MI355X kernels instead take the padded,
time-major batch plus a per-row length vector and freeze rows past their length, so "packing" here
is only the conversion of the reference's length list into a device int32 vector.
"""
import math

import torch
import torch.nn as nn

from . import ops


class RecParams(nn.Module):
    """Parameter container with nn.LSTM's names, shapes, gate order (i,f,g,o) and default init
    (uniform +-1/sqrt(H), drawn in nn.LSTM's parameter order so that a given torch seed yields the
    same weights as the reference's nn.LSTM)."""

    def __init__(self, input_size, hidden_size, num_layers):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        stdv = 1.0 / math.sqrt(hidden_size) if hidden_size > 0 else 0
        for k in range(num_layers):
            isz = input_size if k == 0 else hidden_size
            for name, shape in (('weight_ih_l%d' % k, (4 * hidden_size, isz)),
                                ('weight_hh_l%d' % k, (4 * hidden_size, hidden_size)),
                                ('bias_ih_l%d' % k, (4 * hidden_size,)), ('bias_hh_l%d' % k, (4 * hidden_size,))):
                p = nn.Parameter(torch.empty(*shape))
                nn.init.uniform_(p, -stdv, stdv)
                self.register_parameter(name, p)

    def layer(self, k):
        return (getattr(self, 'weight_ih_l%d' % k), getattr(self, 'weight_hh_l%d' % k),
                getattr(self, 'bias_ih_l%d' % k), getattr(self, 'bias_hh_l%d' % k))


def build_rec(in_size, hsz, depth, residual):
    """nn.LSTM(in,H,L) or, with residual connections, a ModuleList of 1-layer LSTMs
    (char_translator.py:74-78, char_lstm.py:55-59): same state_dict keys."""
    if residual:
        sizes = [in_size] + [hsz] * (depth - 1)
        return nn.ModuleList([RecParams(sizes[i], hsz, 1) for i in range(depth)])
    return RecParams(in_size, hsz, depth)


def rec_layer_params(rec, residual, i):
    return rec[i].layer(0) if residual else rec.layer(i)


def fill_forget_bias(rec, residual, depth, hsz, value):
    """Reference quirk kept: index range arange(H+1, 2H), index H untouched
    (char_translator.py:138-161, char_lstm.py:121-134)."""
    with torch.no_grad():
        for i in range(depth):
            _, _, b_ih, b_hh = rec_layer_params(rec, residual, i)
            b_ih[hsz + 1:2 * hsz] = value
            b_hh[hsz + 1:2 * hsz] = value


def lens_to_device(lens, T, device):
    """Length list (sorted or not) -> int32 device vector, or None when every row runs T steps."""
    if lens is None:
        return None
    if torch.is_tensor(lens):
        return lens.to(device=device, dtype=torch.int32)
    if all(int(l) >= T for l in lens):
        return None
    return torch.tensor([int(l) for l in lens], dtype=torch.int32, device=device)


def run_stack(rec, x, lens_dev, depth, residual, h_prev=None, gc0=None, w_ih0=None):
    """Stacked (optionally residual) LSTM over a padded time-major batch
    (CharTranslator._my_recurrent_layer, char_translator.py:175-198; CharLstm: char_lstm.py:142-164).

    gc0 / w_ih0: optional loop-invariant gate term and replacement input weight for layer 0 (the
    decoder's [ctxt, emb] input split into its constant and per-step halves).
    Returns (out [T,B,H], (h [L,B,H], c [L,B,H]))."""
    cur = x
    hs, cs = [], []
    for i in range(depth):
        w_ih, w_hh, b_ih, b_hh = rec_layer_params(rec, residual, i)
        if i == 0 and w_ih0 is not None:
            w_ih = w_ih0
        h0 = h_prev[0][i] if h_prev is not None else None
        c0 = h_prev[1][i] if h_prev is not None else None
        out, hN, cN = ops.lstm_layer(cur, w_ih, w_hh, b_ih, b_hh, lens_dev, h0, c0, gc0 if i == 0 else None)
        cur = (cur + out) if (residual and i > 0) else out
        hs.append(hN)
        cs.append(cN)
    return cur, (torch.stack(hs, 0), torch.stack(cs, 0))


def packed_mean(out, lens, T):
    """model_utils.packed_mean: sum over the (zero padded) time axis / length."""
    if lens is None:
        return out.sum(dim=0) / float(T)
    l = torch.as_tensor(lens, dtype=torch.float32, device=out.device).view(-1, 1)
    return out.sum(dim=0) / l


def get_classifier(params):
    """model_utils.get_classifier (model_utils.py:5-12)."""
    from .char_lstm import CharLstm
    from .char_cnn import CharCNN
    if params.get('modeltype', 'lstm') == 'lstm':
        return CharLstm(params)
    return CharCNN(params)


def check_hidden(h, what):
    if h not in (64, 128, 256, 512, 1024):
        raise ops.A4NTError('%s = %d: the gfx950 persistent LSTM kernels are instantiated for hidden sizes '
                            '64, 128, 256, 512 and 1024' % (what, h))
