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

    def __init__(self, input_size, hidden_size, num_layers, bidirectional=False):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        self.bidirectional = bool(bidirectional)
        stdv = 1.0 / math.sqrt(hidden_size) if hidden_size > 0 else 0
        for k in range(num_layers):
            isz = input_size if k == 0 else hidden_size * (2 if self.bidirectional else 1)
            for sfx in (['', '_reverse'] if self.bidirectional else ['']):
                for name, shape in (('weight_ih_l%d%s' % (k, sfx), (4 * hidden_size, isz)),
                                    ('weight_hh_l%d%s' % (k, sfx), (4 * hidden_size, hidden_size)),
                                    ('bias_ih_l%d%s' % (k, sfx), (4 * hidden_size,)),
                                    ('bias_hh_l%d%s' % (k, sfx), (4 * hidden_size,))):
                    p = nn.Parameter(torch.empty(*shape))
                    nn.init.uniform_(p, -stdv, stdv)
                    self.register_parameter(name, p)

    def layer(self, k, reverse=False):
        sfx = '_reverse' if reverse else ''
        return (getattr(self, 'weight_ih_l%d%s' % (k, sfx)), getattr(self, 'weight_hh_l%d%s' % (k, sfx)),
                getattr(self, 'bias_ih_l%d%s' % (k, sfx)), getattr(self, 'bias_hh_l%d%s' % (k, sfx)))


def build_rec(in_size, hsz, depth, residual, bidirectional=False):
    """nn.LSTM(in,H,L) or, with residual connections, a ModuleList of 1-layer LSTMs
    (char_translator.py:74-78, char_lstm.py:55-59): same state_dict keys."""
    if residual:
        sizes = [in_size] + [hsz] * (depth - 1)
        return nn.ModuleList([RecParams(sizes[i], hsz, 1, bidirectional) for i in range(depth)])
    return RecParams(in_size, hsz, depth, bidirectional)


def rec_layer_params(rec, residual, i, reverse=False):
    return rec[i].layer(0, reverse) if residual else rec.layer(i, reverse)


def fill_forget_bias(rec, residual, depth, hsz, value, bidirectional=False):
    """Reference quirk kept: index range arange(H+1, 2H), index H untouched
    (char_translator.py:138-161, char_lstm.py:121-134)."""
    with torch.no_grad():
        for i in range(depth):
            for rev in ([False, True] if bidirectional else [False]):
                _, _, b_ih, b_hh = rec_layer_params(rec, residual, i, rev)
                b_ih[hsz + 1:2 * hsz] = value
                b_hh[hsz + 1:2 * hsz] = value


def reverse_time(x, lens, T):
    """Per-row time reversal of the valid prefix of a padded time-major batch (row b: steps 0..len_b-1 are
    flipped, the padding stays where it is).  Its own inverse.  lens None: every row has T steps."""
    if lens is None:
        return x.flip(0)
    l = ops.h2d([int(v) for v in lens], x.device).view(1, -1)
    t = torch.arange(T, device=x.device).view(-1, 1)
    idx = torch.where(t < l, l - 1 - t, t)                                     # [T,B]
    return x.gather(0, idx.unsqueeze(-1).expand(-1, -1, x.shape[2]))


def run_stack_bidir(rec, x, lens, depth, h_prev=None):
    """Bidirectional stacked LSTM (nn.LSTM(bidirectional=True), char_lstm.py:57-59): the reverse direction is
    the same persistent forward sweep over the per-row time-reversed input.  Returns (out [T,B,2H],
    (h [2L,B,H], c [2L,B,H])) with nn.LSTM's state order (layer 0 fwd, layer 0 reverse, layer 1 fwd, ...)."""
    T = x.shape[0]
    lens_dev = lens_to_device(lens, T, x.device)
    cur, hs, cs = x, [], []
    for i in range(depth):
        outs = []
        for d, rev in enumerate((False, True)):
            w_ih, w_hh, b_ih, b_hh = rec.layer(i, rev)
            h0 = h_prev[0][2 * i + d] if h_prev is not None else None
            c0 = h_prev[1][2 * i + d] if h_prev is not None else None
            inp = reverse_time(cur, lens, T) if rev else cur
            out, hN, cN = ops.lstm_layer(inp, w_ih, w_hh, b_ih, b_hh, lens_dev, h0, c0)
            outs.append(reverse_time(out, lens, T) if rev else out)
            hs.append(hN)
            cs.append(cN)
        cur = torch.cat(outs, dim=-1)
    return cur, (torch.stack(hs, 0), torch.stack(cs, 0))


def ids_to_device(x, device):
    """Token ids (CPU int64, as the callers hand them over: char_translator.py:208, char_lstm.py:273) -> device.
    Asynchronous (ops.h2d): from the pinned staging tensors of data_provider.prepare_data a DMA the host does not wait
    for; pageable tensors are staged through a pinned copy first."""
    return ops.h2d(x, device)


def lens_to_device(lens, T, device):
    """Length list (sorted or not) -> int32 device vector, or None when every row runs T steps."""
    if lens is None:
        return None
    if torch.is_tensor(lens):
        return ops.h2d(lens, device, torch.int32)
    if all(int(l) >= T for l in lens):
        return None
    return ops.h2d([int(l) for l in lens], device, torch.int32)


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
    l = ops.h2d([float(v) for v in lens] if not torch.is_tensor(lens) else lens, out.device, torch.float32).view(-1, 1)
    return out.sum(dim=0) / l


def get_classifier(params):
    """model_utils.get_classifier (model_utils.py:5-12)."""
    from .char_lstm import CharLstm
    from .char_cnn import CharCNN
    if params.get('modeltype', 'lstm') == 'lstm':
        return CharLstm(params)
    return CharCNN(params)


def check_hidden(h, what):
    """Any hidden size up to the largest instantiation (1024): sizes in between run zero-padded (ops.lstm_layer)."""
    if not (1 <= int(h) <= ops.HIDDEN_SIZES[-1]):
        raise ops.A4NTError('%s = %d: the gfx950 persistent LSTM kernels serve hidden sizes 1..%d'
                            % (what, h, ops.HIDDEN_SIZES[-1]))
