"""HIP-backed CharLstm attribution classifier / discriminator (reference models/char_lstm.py:32-164,
263-340): same constructor, params dict, forward_classify signature and state_dict keys.

Scope: classifier mode with the default linear head and the
maxpoolrnn 0/1/2/3 pooling variants, hard-id and soft (`adv_inp`) inputs.  The bidirectional /
compression / MLP / CNN-over-hidden / generic heads and the generative multi-head LM modes are
section-8(f) "next" rows and raise NotImplementedError.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .gen_decode import SoftSamples, sample_embed
from .model_utils import build_rec, fill_forget_bias, lens_to_device, run_stack, packed_mean, check_hidden

_UNSUPPORTED = ['bidir', 'compression_layer', 'decoder_mlp', 'decoder_cnn', 'generic_classifier']


class CharLstm(nn.Module):
    def __init__(self, params):
        super().__init__()
        for k in _UNSUPPORTED:
            if params.get(k, 0):
                raise NotImplementedError('a4nt_amd.CharLstm: option %r is outside the hot-path scope' % k)
        if params.get('mode', 'generative') != 'classifier':
            raise NotImplementedError('a4nt_amd.CharLstm: only mode="classifier" is on the hot path')
        self.output_size = params.get('vocabulary_size', -1) + 1
        self.num_output_layers = params.get('num_output_layers', 1)
        self.num_rec_layers = params.get('hidden_depth', -1)
        self.emb_size = params.get('embedding_size', -1)
        self.hidden_size = params.get('hidden_size', -1)
        self.en_residual = params.get('en_residual_conn', 0)
        self.max_pool_rnn = params.get('maxpoolrnn', 0)
        check_hidden(self.hidden_size, 'hidden_size')
        self.encoder = nn.Embedding(self.output_size, self.emb_size, padding_idx=0)
        self.p_enc = params.get('drop_prob_encoder', 0.25)
        self.rec_layers = build_rec(self.emb_size, self.hidden_size, self.num_rec_layers, self.en_residual)
        decoder_size = self.hidden_size * (1 + int(self.max_pool_rnn in (1, 3)))
        self.decoder_W = nn.Parameter(torch.zeros(decoder_size, self.num_output_layers))
        self.decoder_b = nn.Parameter(torch.zeros(self.num_output_layers))
        self.p_dec = params.get('drop_prob_decoder', 0.25)
        self.init_weights()
        self.to(params.get('device', 'cuda'))
        ops._lib.init_device(self.decoder_W.device.index or 0)

    def init_weights(self):
        a = math.sqrt(float(self.decoder_W.shape[0]))
        with torch.no_grad():
            self.decoder_W.uniform_(-1.73 / a, 1.73 / a)
            self.decoder_b.zero_()
        fill_forget_bias(self.rec_layers, self.en_residual, self.num_rec_layers, self.hidden_size, 2.0)

    def init_hidden(self, bsz):
        w = self.decoder_W
        return (w.new_zeros(self.num_rec_layers, bsz, self.hidden_size),
                w.new_zeros(self.num_rec_layers, bsz, self.hidden_size))

    def _drop(self, x, p, drop):
        return F.dropout(x, p, True) if (drop and self.training and p > 0.0) else x

    def forward(self, *a, **k):
        raise NotImplementedError('generative multi-head LM mode (char_lstm.py:166-261) is a section-8(f) row')

    forward_eval = forward
    forward_gen = forward

    def _embed_any(self, x, adv_inp, drop, dev):
        T, B = x.shape[0], x.shape[1]
        if isinstance(x, SoftSamples):
            emb = sample_embed(x, self.encoder.weight)
        elif adv_inp:
            emb = ops.linear_kn(x.reshape(T * B, -1), self.encoder.weight).view(T, B, -1)
        else:
            emb = ops.embedding(x.to(dev), self.encoder.weight, 0)
        return self._drop(emb, self.p_enc, drop)

    def _classify_emb(self, emb, lens, h_prev, compute_softmax, drop):
        dev = emb.device
        T, B = emb.shape[0], emb.shape[1]
        lens_dev = lens_to_device(lens, T, dev)
        rnn_out, hidden = run_stack(self.rec_layers, emb, lens_dev, self.num_rec_layers, self.en_residual, h_prev)
        t_eff = max(int(l) for l in lens) if lens is not None else T
        if self.max_pool_rnn == 1:
            ctxt = torch.cat([packed_mean(rnn_out, lens, T), hidden[0][-1]], dim=-1)
        elif self.max_pool_rnn == 2:
            ctxt = rnn_out[:t_eff].max(dim=0)[0]          # zero pads take part (char_lstm.py:303-306)
        elif self.max_pool_rnn == 3:
            ctxt = torch.cat([rnn_out[:t_eff].max(dim=0)[0],
                              hidden[0].transpose(0, 1).contiguous().view(B, -1)], dim=-1)
        else:
            ctxt = hidden[0][-1]
        enc_out = self._drop(ctxt, self.p_dec, drop)
        dec_out = ops.linear_kn(enc_out, self.decoder_W, self.decoder_b)
        if compute_softmax:
            dec_out = F.softmax(dec_out, dim=-1)
        return dec_out, enc_out

    def forward_classify(self, x, h_prev=None, compute_softmax=False, predict_mode=False, adv_inp=False, lens=None,
                         drop=True):
        dev = self.decoder_W.device
        ctxmgr = torch.no_grad() if predict_mode else torch.enable_grad()
        with ctxmgr:
            emb = self._embed_any(x, adv_inp, drop, dev)
            return self._classify_emb(emb, lens, h_prev, compute_softmax, drop)

    def forward_classify_multi(self, inputs, drop=True):
        """Several independent forward_classify calls as ONE pass: the batches are concatenated
        along the batch axis so that each persistent LSTM sweep serves all of them (twice the rows per
        inter-workgroup synchronisation).  `inputs` = [(x, lens, adv_inp), ...]; rows are independent, so
        every result equals the separate call's.  Pooling variants that look at the padded length
        (maxpoolrnn 2/3) fall back to separate calls."""
        dev = self.decoder_W.device
        if self.max_pool_rnn in (2, 3) or len(inputs) == 1:
            return [self.forward_classify(x, adv_inp=a, lens=l, drop=drop) for x, l, a in inputs]
        embs, lens_all, sizes = [], [], []
        T = max(x.shape[0] for x, _, _ in inputs)
        for x, lens, adv in inputs:
            e = self._embed_any(x, adv, drop, dev)
            if e.shape[0] < T:
                e = F.pad(e, (0, 0, 0, 0, 0, T - e.shape[0]))
            embs.append(e)
            lens_all += [int(l) for l in (lens if lens is not None else [x.shape[0]] * x.shape[1])]
            sizes.append(x.shape[1])
        dec_out, enc_out = self._classify_emb(torch.cat(embs, dim=1), lens_all, None, False, drop)
        return list(zip(dec_out.split(sizes, dim=0), enc_out.split(sizes, dim=0)))
