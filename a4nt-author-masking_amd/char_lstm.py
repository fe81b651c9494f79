"""HIP-backed CharLstm attribution classifier / discriminator / multi-head language model (reference
models/char_lstm.py:32-340): same constructor, params dict, method signatures and state_dict keys.

Classifier mode: the default linear head with the maxpoolrnn 0/1/2/3 pooling variants, the MLP head
(`decoder_mlp`), the CNN-over-hidden head (`decoder_cnn`), the generic real/fake head
(`generic_classifier`), a bidirectional LSTM (`bidir`; the reverse direction is the persistent forward
sweep over per-row time-reversed input) and the compression layer; hard-id and soft (`adv_inp`) inputs.
Generative mode: one [H,V] output layer per author (`forward`, `forward_eval`, `forward_gen`).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .gen_decode import SoftSamples, sample_embed
from .model_utils import (ids_to_device, build_rec, fill_forget_bias, lens_to_device, run_stack, run_stack_bidir, packed_mean,
                          check_hidden)


class CharLstm(nn.Module):
    def __init__(self, params):
        super().__init__()
        self.output_size = params.get('vocabulary_size', -1) + 1
        self.num_output_layers = params.get('num_output_layers', 1)
        self.num_rec_layers = params.get('hidden_depth', -1)
        self.emb_size = params.get('embedding_size', -1)
        self.hidden_size = params.get('hidden_size', -1)
        self.en_residual = params.get('en_residual_conn', 0)
        self.bidir = bool(params.get('bidir', 0))
        self.compression_layer = params.get('compression_layer', 0)
        self.max_pool_rnn = params.get('maxpoolrnn', 0)
        self.classifier = params.get('mode', 'generative') == 'classifier'
        check_hidden(self.hidden_size, 'hidden_size')
        if self.bidir and self.en_residual and self.num_rec_layers > 1:
            # the reference builds residual layers with H-wide inputs but a bidirectional layer emits 2H (:56-57)
            raise ops.A4NTError('bidir with en_residual_conn needs hidden_depth 1 (the reference layers do not compose)')
        H, D = self.hidden_size, 1 + int(self.bidir)
        # construction order follows the reference so that a torch seed gives the same weights
        if self.compression_layer:
            self.compression_W = nn.Embedding(self.output_size, self.compression_layer)
            self.encoder = nn.Embedding(self.compression_layer, self.emb_size)
        else:
            self.encoder = nn.Embedding(self.output_size, self.emb_size, padding_idx=0)
        self.p_enc = params.get('drop_prob_encoder', 0.25)
        self.rec_layers = build_rec(self.emb_size, H, self.num_rec_layers, self.en_residual, self.bidir)
        self.head_mlp = self.head_cnn = False
        if self.classifier:
            decoder_size = H * (1 + int(self.max_pool_rnn in (1, 3))) * D
            if params.get('decoder_mlp', 0):
                self.decoder_W_mlp = nn.Parameter(torch.zeros(decoder_size, params['decoder_mlp']))
                self.decoder_b_mlp = nn.Parameter(torch.zeros(params['decoder_mlp']))
                decoder_size = params['decoder_mlp']
                self.head_mlp = True
            elif params.get('decoder_cnn', 0):
                from .char_cnn import _ConvParams
                self.cnn_ks = params['decoder_cnn_ks']
                self.cnn_nfilt = params['decoder_cnn_nfilt']
                self.decoder_cnn = nn.ModuleList([_ConvParams(H, self.cnn_nfilt, K, padding=K) for K in self.cnn_ks])
                self.head_cnn = True
                decoder_size = self.cnn_nfilt * len(self.cnn_ks)
            self.decoder_W = nn.Parameter(torch.zeros(decoder_size, self.num_output_layers))
            self.decoder_b = nn.Parameter(torch.zeros(self.num_output_layers))
            if params.get('generic_classifier', False):
                self.generic_W = nn.Parameter(torch.zeros(H * (1 + int(self.max_pool_rnn == 1)), 1))
                self.generic_b = nn.Parameter(torch.zeros(1))
        else:
            self.decoder_W = nn.Parameter(torch.zeros(self.num_output_layers, H, self.output_size))
            self.decoder_b = nn.Parameter(torch.zeros(self.num_output_layers, self.output_size))
        self.p_dec = params.get('drop_prob_decoder', 0.25)
        self.init_weights()
        self.to(params.get('device', 'cuda'))
        if self.decoder_W.is_cuda:            # device='cpu' builds a parameter container only (init / checkpoint tools);
            ops._lib.init_device(self.decoder_W.device.index or 0)      # every forward still needs the GPU

    def init_weights(self):
        a = math.sqrt(float(self.decoder_W.shape[0]))
        with torch.no_grad():
            if hasattr(self, 'generic_W'):
                self.generic_W.uniform_(-a, a)
                self.generic_b.zero_()
            if self.head_mlp:
                n_in = math.sqrt(float(self.decoder_W_mlp.shape[0]))
                self.decoder_W_mlp.uniform_(-1.73 / n_in, 1.73 / n_in)
                self.decoder_b_mlp.zero_()
            self.decoder_W.uniform_(-1.73 / a, 1.73 / a)
            self.decoder_b.zero_()
            if self.compression_layer:
                w = self.compression_W.weight
                w.div_(w.abs().sum(dim=1, keepdim=True))
        fill_forget_bias(self.rec_layers, self.en_residual, self.num_rec_layers, self.hidden_size, 2.0, self.bidir)

    def init_hidden(self, bsz):
        w = self.decoder_W
        n = self.num_rec_layers * (1 + int(self.bidir))
        return (w.new_zeros(n, bsz, self.hidden_size), w.new_zeros(n, bsz, self.hidden_size))

    def _drop(self, x, p, drop):
        return F.dropout(x, p, True) if (drop and self.training and p > 0.0) else x

    def _rec(self, emb, lens, h_prev):
        T = emb.shape[0]
        if self.bidir:
            rec = self.rec_layers[0] if self.en_residual else self.rec_layers
            return run_stack_bidir(rec, emb, lens, self.num_rec_layers, h_prev)
        return run_stack(self.rec_layers, emb, lens_to_device(lens, T, emb.device), self.num_rec_layers,
                         self.en_residual, h_prev)

    # -- generative multi-head language model (char_lstm.py:166-261) ---------------------------
    def _need_generative(self, what):
        if self.classifier:
            raise ops.A4NTError('%s needs mode="generative" (this CharLstm was built as a classifier)' % what)

    def forward(self, x, lengths, h_prev, target_head, compute_softmax=False):
        """Scores [T,B,V] of every position under each row's own author head (baddbmm with decoder_W[target_head],
        :180-185): rows are grouped by head and each group is one MFMA GEMM."""
        self._need_generative('forward')
        dev = self.decoder_W.device
        T, B = x.shape
        emb = self._drop(ops.embedding(ids_to_device(x, dev), self.encoder.weight, 0), self.p_enc, True)
        rnn_out, hidden = self._rec(emb, lengths, h_prev)
        n_t = max(int(l) for l in lengths)
        rnn_out = self._drop(rnn_out[:n_t], self.p_dec, True)
        head = target_head.to(dev)
        out = None
        for c in range(self.num_output_layers):
            rows = (head == c).nonzero().view(-1)
            if rows.numel() == 0:
                continue
            sc = ops.linear_kn(rnn_out.index_select(1, rows).reshape(-1, self.hidden_size), self.decoder_W[c],
                               self.decoder_b[c]).view(n_t, rows.numel(), self.output_size)
            out = sc.new_zeros(n_t, B, self.output_size) if out is None else out
            out = out.index_copy(1, rows, sc)
        if compute_softmax:
            out = F.softmax(out, dim=-1)
        return out, hidden

    def forward_eval(self, x, h_prev, compute_softmax=True):
        """One sequence ([T,1] ids) scored by every author's output layer -> [T, n_auth, V] (:194-222)."""
        self._need_generative('forward_eval')
        dev = self.decoder_W.device
        with torch.no_grad():
            emb = ops.embedding(ids_to_device(x, dev), self.encoder.weight, 0)
            rnn_out, hidden = self._rec(emb, None, h_prev)
            h2 = rnn_out.reshape(-1, self.hidden_size)
            out = torch.stack([ops.linear_kn(h2, self.decoder_W[c], self.decoder_b[c])
                               for c in range(self.num_output_layers)], dim=1)
            if compute_softmax:
                out = F.softmax(out, dim=-1)
        return out, hidden

    def forward_gen(self, x, h_prev, target_auth, n_max=100, end_c=-1):
        """Greedy continuation of the seed x ([T,1]) under target_auth's head; like the reference it never stops
        early (:252).  The reference's own loop cannot run on current torch (2-D LSTM input, :255-258), so this is
        checked against forward_eval instead of a golden vector."""
        self._need_generative('forward_gen')
        dev = self.decoder_W.device
        c = int(target_auth.view(-1)[0])
        char_out = []
        with torch.no_grad():
            emb = ops.embedding(ids_to_device(x, dev), self.encoder.weight, 0)
            rnn_out, hidden = self._rec(emb, None, h_prev)
            p_rnn = rnn_out[-1]
            for _ in range(n_max):
                logits = ops.linear_kn(p_rnn, self.decoder_W[c], self.decoder_b[c])
                pred_c = torch.empty(logits.shape[0], dtype=torch.int64, device=dev)
                ops.gumbel_rows(logits.contiguous(), None, 0, 0, 1.0, 1.0, 1, 1, None, None, pred_c, None, None, None,
                                None, -1)
                char_out.append(pred_c)
                emb = ops.embedding(pred_c.view(1, -1), self.encoder.weight, 0)
                rnn_out, hidden = self._rec(emb, None, hidden)
                p_rnn = rnn_out[-1]
        return char_out

    # -- classifier (char_lstm.py:263-340) -------------------------------------------------------
    def _embed_any(self, x, adv_inp, drop, dev):
        T, B = x.shape[0], x.shape[1]
        if self.compression_layer:
            # vocabulary -> compression classes (L1-normalised per position) -> embedding (:275-278, :284-287)
            if isinstance(x, SoftSamples):
                comp = sample_embed(x, self.compression_W.weight).view(T * B, -1)
            elif adv_inp:
                comp = ops.linear_kn(x.reshape(T * B, -1), self.compression_W.weight)
            else:
                comp = ops.embedding(ids_to_device(x, dev), self.compression_W.weight, -1).view(T * B, -1)
            qn = comp.abs().sum(dim=1, keepdim=True)
            if adv_inp or isinstance(x, SoftSamples):
                qn = qn.detach()
            emb = ops.linear_kn(comp / (qn + 1e-8), self.encoder.weight).view(T, B, -1)
        elif isinstance(x, SoftSamples):
            emb = sample_embed(x, self.encoder.weight)
        elif adv_inp:
            emb = ops.linear_kn(x.reshape(T * B, -1), self.encoder.weight).view(T, B, -1)
        else:
            emb = ops.embedding(ids_to_device(x, dev), self.encoder.weight, 0)
        return self._drop(emb, self.p_enc, drop)

    def _classify_emb(self, emb, lens, h_prev, compute_softmax, drop):
        from .char_cnn import _Conv1dTM, _LreluMaxTime
        if not self.classifier:
            raise ops.A4NTError('forward_classify needs mode="classifier"')
        T, B = emb.shape[0], emb.shape[1]
        rnn_out, hidden = self._rec(emb, lens, h_prev)
        t_eff = max(int(l) for l in lens) if lens is not None else T
        if self.head_cnn:
            # Conv1d(padding=K) over the zero-padded hidden states, leaky_relu, max over time (:321-327); this
            # dropout follows train()/eval() only, whatever `drop` says
            pooled = [_LreluMaxTime.apply(_Conv1dTM.apply(rnn_out[:t_eff], cv.weight, cv.bias, cv.padding), 0, 0.01)
                      for cv in self.decoder_cnn]
            dec_in = self._drop(torch.cat(pooled, dim=1), self.p_dec, True)
            enc_out = dec_in
        else:
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
            dec_in = enc_out
            if self.head_mlp:
                dec_in = torch.tanh(ops.linear_kn(enc_out, self.decoder_W_mlp, self.decoder_b_mlp))
        dec_out = ops.linear_kn(dec_in, self.decoder_W, self.decoder_b)
        if compute_softmax:
            dec_out = F.softmax(dec_out, dim=-1)
        if hasattr(self, 'generic_W'):
            # real/fake head; the reference returns the LSTM state, not the features, in this case (:335-338)
            generic = torch.sigmoid(ops.linear_kn(enc_out, self.generic_W, self.generic_b))
            return dec_out, hidden, generic
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
        if self.max_pool_rnn in (2, 3) or len(inputs) == 1 or self.head_cnn or self.bidir or hasattr(self, 'generic_W'):
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
