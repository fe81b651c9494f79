"""HIP-backed CharTranslator: the reference's seq2seq generator (models/char_translator.py:30-401)
with the same constructor, params dict, method signatures and state_dict keys, computing through
liba4nt_hip.so (persistent LSTM sweeps, MFMA GEMMs, fused Gumbel / NLL row kernels).

Inputs follow the reference's conventions: token ids arrive as CPU int64 [T,B] tensors, lengths as
Python lists, `auths` as a CPU LongTensor; outputs are CUDA tensors attached to autograd.
Extensions (used by this package's own adversarial step, not required by reference callers):
  * forward_advers_gen(..., compact=True) returns the samples as a SoftSamples (token, value) pair
    instead of a list of dense [B,V] rows;
  * every method that takes a dense soft input [T,B,V] also takes a SoftSamples;
  * forward_mltrain_loss fuses the vocab projection with log-softmax/NLL (the reference's callers
    compute CrossEntropyLoss on the returned [T,B,V] scores themselves);
  * set_noise(list) pins the Gumbel uniforms for parity tests (the reference draws from the global
    CUDA RNG, char_translator.py:11); otherwise an on-device Philox stream keyed by `noise_seed`.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .gen_decode import SoftSamples, sample_embed, gen_decode
from .model_utils import (ids_to_device, build_rec, rec_layer_params, fill_forget_bias, lens_to_device, run_stack, packed_mean,
                          check_hidden)

class CharTranslator(nn.Module):
    def __init__(self, params, encoder_only=False):
        super().__init__()
        self.encoder_only = encoder_only
        self.no_encoder = params.get('no_encoder', 0)
        self.enc_noise = params.get('apply_noise', 0)
        self.learn_gumbel = params.get('learn_gumbel', False)
        self.split_gen = params.get('split_generators', 0)
        self.encoder_mean_vec = params.get('encoder_mean_vec', 0)
        self.vocab_size = params.get('vocabulary_size', -1) + 1
        self.enc_num_rec_layers = params.get('enc_hidden_depth', -1)
        self.emb_size = params.get('embedding_size', -1)
        self.enc_hidden_size = params.get('enc_hidden_size', -1)
        self.dec_num_rec_layers = params.get('dec_hidden_depth', -1)
        self.dec_hidden_size = params.get('dec_hidden_size', -1)
        self.en_residual = params.get('en_residual_conn', 1)
        self.pad_auth_vec = params.get('pad_auth_vec', 0)
        self.gumb_type = params.get('gumbel_hard', False)
        self.softmax_scale = params.get('softmax_scale', 2.0)
        self.max_pool_rnn = params.get('maxpoolrnn', 0)
        self.temp = None
        if not self.no_encoder:
            check_hidden(self.enc_hidden_size, 'enc_hidden_size')

        # construction order follows the reference so that a torch seed gives the same weights
        if self.learn_gumbel:
            self.gumbel_W = nn.Parameter(torch.zeros(self.dec_hidden_size, 1))
            self.gumbel_b = nn.Parameter(torch.zeros(1))
        if self.pad_auth_vec:
            self.auth_emb = nn.Embedding(params.get('num_output_layers', 2), self.pad_auth_vec)
        self.char_emb = nn.Embedding(self.vocab_size, self.emb_size, padding_idx=0)
        self.p_emb = params.get('drop_prob_emb', 0.25)
        if not self.no_encoder:
            self.enc_rec_layers = build_rec(self.emb_size, self.enc_hidden_size, self.enc_num_rec_layers,
                                            self.en_residual)
        self.p_enc = params.get('drop_prob_encoder', 0.25)
        if not encoder_only:
            check_hidden(self.dec_hidden_size, 'dec_hidden_size')
            enc_out = self.enc_hidden_size * (1 + self.max_pool_rnn) if not self.no_encoder else 0
            self.ctx_size = enc_out + self.pad_auth_vec
            self.dec_rec_layers = build_rec(self.ctx_size + self.emb_size, self.dec_hidden_size,
                                            self.dec_num_rec_layers, self.en_residual)
            self.decoder_W = nn.Parameter(torch.zeros(self.dec_hidden_size, self.vocab_size))
            self.decoder_b = nn.Parameter(torch.zeros(self.vocab_size))
            self.p_dec = params.get('drop_prob_decoder', 0.25)
            if self.split_gen:
                # second decoder + output layer for batches of author 1 (char_translator.py:99-108)
                self.dec_rec_layers_2 = build_rec(self.ctx_size + self.emb_size, self.dec_hidden_size,
                                                  self.dec_num_rec_layers, self.en_residual)
                self.decoder_W_2 = nn.Parameter(torch.zeros(self.dec_hidden_size, self.vocab_size))
                self.decoder_b_2 = nn.Parameter(torch.zeros(self.vocab_size))
        self.init_weights()
        self.zero_hidden_bsz = params['batch_size']
        self._noise = None
        self.noise_seed = int(params.get('noise_seed', 0))
        self._noise_step = 0
        dev = params.get('device', 'cuda')
        self.to(dev)
        if self.char_emb.weight.is_cuda:            # device='cpu' builds a parameter container only (init / checkpoint tools);
            ops._lib.init_device(self.char_emb.weight.device.index or 0)      # every forward still needs the GPU

    def init_weights(self):
        with torch.no_grad():
            if not self.encoder_only:
                self.decoder_W.uniform_(-0.001, 0.001)
                self.decoder_b.zero_()
                if self.split_gen:
                    self.decoder_W_2.uniform_(-0.001, 0.001)
                    self.decoder_b_2.zero_()
            if self.learn_gumbel:
                self.gumbel_W.uniform_(-0.01, 0.01)
        if not self.no_encoder:
            fill_forget_bias(self.enc_rec_layers, self.en_residual, self.enc_num_rec_layers, self.enc_hidden_size, 1.0)
        if not self.encoder_only:
            fill_forget_bias(self.dec_rec_layers, self.en_residual, self.dec_num_rec_layers, self.dec_hidden_size, 1.0)
            if self.split_gen:
                fill_forget_bias(self.dec_rec_layers_2, self.en_residual, self.dec_num_rec_layers,
                                 self.dec_hidden_size, 1.0)

    def _decoder(self, auths):
        """(rec layers, W, b) of the decoder in use: the second set when split_generators is on and the
        batch's first author id is not 0 (char_translator.py:265,273-275,359-366)."""
        if not self.split_gen or int(auths[0]) == 0:
            return self.dec_rec_layers, self.decoder_W, self.decoder_b
        return self.dec_rec_layers_2, self.decoder_W_2, self.decoder_b_2

    # ------------------------------------------------------------------------------------------
    def set_noise(self, uniforms):
        """Pin the Gumbel uniforms: one [B,V] CUDA tensor per sampling step, consumed in order."""
        self._noise = list(uniforms) if uniforms is not None else None

    def _dev(self):
        return self.char_emb.weight.device

    def init_hidden(self, bsz):
        w = self.char_emb.weight
        return (w.new_zeros(self.enc_num_rec_layers, bsz, self.enc_hidden_size),
                w.new_zeros(self.enc_num_rec_layers, bsz, self.enc_hidden_size))

    def init_hidden_dec(self, bsz):
        w = self.char_emb.weight
        return (w.new_zeros(self.dec_num_rec_layers, bsz, self.dec_hidden_size),
                w.new_zeros(self.dec_num_rec_layers, bsz, self.dec_hidden_size))

    def _need_encoder(self, what):
        if self.no_encoder:
            raise ops.A4NTError('%s needs the encoder, but this translator was built with no_encoder=1 (the reference '
                                'has no enc_rec_layers there either, char_translator.py:73,335)' % what)

    def _drop(self, x, p):
        return F.dropout(x, p, True) if (self.training and p > 0.0) else x

    def _embed_input(self, x, adv_inp, drop):
        """Hard ids -> nn.Embedding (+ emb dropout when `drop`); soft input -> samples * char_emb.weight
        (no dropout, char_translator.py:212-214)."""
        if isinstance(x, SoftSamples):
            return sample_embed(x, self.char_emb.weight)
        if adv_inp:
            T, B = x.shape[0], x.shape[1]
            return ops.linear_kn(x.reshape(T * B, -1), self.char_emb.weight).view(T, B, -1)
        emb = ops.embedding(ids_to_device(x, self._dev()), self.char_emb.weight, 0)
        return self._drop(emb, self.p_emb) if drop else emb

    def _encode(self, emb, lengths, h_prev):
        T = emb.shape[0]
        lens_dev = lens_to_device(lengths, T, emb.device)
        out, hidden = run_stack(self.enc_rec_layers, emb, lens_dev, self.enc_num_rec_layers, self.en_residual, h_prev)
        return out, hidden

    # -- teacher forced (char_translator.py:200-286) -------------------------------------------
    def _mltrain_hidden(self, inp, lengths_inp, targ, lengths_targ, h_prev, auths, adv_inp, sort_enc, adv_targ=False):
        dev = self._dev()
        soft = adv_inp or isinstance(inp, SoftSamples)
        T_in, B = inp.shape[0], inp.shape[1]
        enc_hidden, ctxt = None, None
        if not self.no_encoder:
            emb = self._embed_input(inp, soft, drop=True)
            enc_out, enc_hidden = self._encode(emb, lengths_inp, h_prev)
            if self.max_pool_rnn:
                # reference: plain mean over the padded time axis of the UNPACKED output, whose length is
                # max(lengths_inp) (char_translator.py:223-228)
                t_eff = max(int(l) for l in lengths_inp)
                ctxt = torch.cat([enc_out[:t_eff].mean(dim=0), enc_hidden[0][-1]], dim=-1)
            else:
                ctxt = enc_hidden[0][-1]
            if sort_enc is not None:
                ctxt = ctxt.index_select(0, ops.h2d(sort_enc, dev))
            if self.pad_auth_vec:
                ctxt = torch.cat([ctxt, self.auth_emb(ops.h2d(auths, dev))], dim=-1)
            if self.enc_noise:
                ctxt = ctxt + torch.randn(ctxt.shape, device=dev) / 20.       # char_translator.py:240-241
            else:
                ctxt = self._drop(ctxt, self.p_enc)
        elif self.ctx_size:
            raise ops.A4NTError('no_encoder with pad_auth_vec > 0: the decoder input would be narrower than its '
                                'weights (the reference fails the same way, char_translator.py:86,260)')

        if adv_targ or isinstance(targ, SoftSamples):
            # soft target rows (the language-model loss feeds generated samples, char_translator.py:252-254)
            if isinstance(targ, SoftSamples):
                t_emb = sample_embed(targ, self.char_emb.weight)
            else:
                t_emb = ops.linear_kn(targ.reshape(targ.shape[0] * targ.shape[1], -1), self.char_emb.weight).view(
                    targ.shape[0], targ.shape[1], -1)
            targ_emb = self._drop(t_emb, self.p_emb)
        else:
            targ = ids_to_device(targ, dev)
            targ_emb = self._drop(ops.embedding(targ, self.char_emb.weight, 0), self.p_emb)
        n_t = targ.shape[0]
        lens_dev = lens_to_device(lengths_targ, n_t, dev)
        dec_rec, W_out, b_out = self._decoder(auths)
        if ctxt is not None:
            # decoder input is [ctxt, emb_t]: the ctxt half of the input contraction is loop invariant
            w_ih0 = rec_layer_params(dec_rec, self.en_residual, 0)[0]
            gc = ops.linear_nk(ctxt, w_ih0[:, :self.ctx_size])
            dec_out, dec_hidden = run_stack(dec_rec, targ_emb, lens_dev, self.dec_num_rec_layers, self.en_residual,
                                            None, gc0=gc, w_ih0=w_ih0[:, self.ctx_size:].contiguous())
        else:
            dec_out, dec_hidden = run_stack(dec_rec, targ_emb, lens_dev, self.dec_num_rec_layers, self.en_residual, None)
        n_eff = max(int(l) for l in lengths_targ)
        dec_out = self._drop(dec_out[:n_eff], self.p_dec)
        return dec_out, n_eff, B, (enc_hidden, dec_hidden), W_out, b_out

    def forward_mltrain(self, inp, lengths_inp, targ, lengths_targ, h_prev=None, compute_softmax=False, auths=None,
                        adv_inp=False, sort_enc=None, adv_targ=False):
        dec_out, n_t, B, hidden, W_out, b_out = self._mltrain_hidden(inp, lengths_inp, targ, lengths_targ, h_prev, auths,
                                                                     adv_inp, sort_enc, adv_targ)
        score = ops.linear_kn(dec_out.reshape(n_t * B, -1), W_out, b_out)
        if compute_softmax:
            score = F.log_softmax(score, dim=1)
        return score.view(n_t, B, self.vocab_size), hidden

    def forward_mltrain_loss(self, inp, lengths_inp, targ_in, lengths_targ, targ_out, auths=None, adv_inp=False,
                             sort_enc=None, h_prev=None):
        """CrossEntropyLoss(pack(forward_mltrain(...)[0], lens), pack(targ_out, lens)) with the vocab
        projection fused into the log-softmax/NLL kernel (train_translator.py:111-113;
        train_adversarial_translator.py:176-177,661-664)."""
        dec_out, n_t, B, _, W_out, b_out = self._mltrain_hidden(inp, lengths_inp, targ_in, lengths_targ, h_prev, auths,
                                                                adv_inp, sort_enc)
        dev = self._dev()
        tgt = ops.h2d(targ_out[:n_t], dev).clone()
        steps = torch.arange(n_t, device=dev).view(-1, 1)
        lens_t = ops.h2d([int(l) for l in lengths_targ], dev).view(1, -1)
        tgt.masked_fill_(steps >= lens_t, -1)
        n_valid = int(sum(min(int(l), n_t) for l in lengths_targ))
        return ops.proj_nll(dec_out.reshape(n_t * B, -1), W_out, b_out, tgt.view(-1), n_valid)

    # -- encoder only (char_translator.py:288-312) ----------------------------------------------
    def forward_encode(self, x, lengths_inp, h_prev=None, adv_inp=False):
        self._need_encoder('forward_encode')
        emb = self._embed_input(x, adv_inp, drop=False)
        enc_out, enc_hidden = self._encode(emb, lengths_inp, h_prev)
        if self.encoder_mean_vec:
            return torch.cat([packed_mean(enc_out, lengths_inp, emb.shape[0]), enc_hidden[0][-1]], dim=-1)
        return enc_hidden[0][-1]

    # -- free running generation (char_translator.py:314-401) -----------------------------------
    def _gen_context(self, x, lengths_inp, h_prev, soft_in, auths):
        """Encoder pass of forward_advers_gen (char_translator.py:321-345) -> ctxt [B, ctx_size]."""
        emb = self._embed_input(x, soft_in, drop=False)
        enc_out, enc_hidden = self._encode(emb, lengths_inp, h_prev)
        if self.max_pool_rnn:
            ctxt = torch.cat([packed_mean(enc_out, lengths_inp, emb.shape[0]), enc_hidden[0][-1]], dim=-1)
        else:
            ctxt = enc_hidden[0][-1]
        if self.pad_auth_vec:
            ctxt = torch.cat([ctxt, self.auth_emb(ops.h2d(auths, self._dev()))], dim=-1)
        return ctxt

    def forward_advers_gen(self, x, lengths_inp, h_prev=None, n_max=100, end_c=-1, soft_samples=False, temp=0.1,
                           auths=None, adv_inp=False, n_samples=1, compact=False, grad_rows=None):
        """grad_rows = (r0, r1) (extension, fused path with compact=True only): only batch columns [r0, r1)
        will be differentiated; the others are encoded without a graph and share the decoding loop."""
        self._need_encoder('forward_advers_gen')
        dev = self._dev()
        soft_in = adv_inp or isinstance(x, SoftSamples)
        B = x.shape[1]
        if grad_rows is None:
            ctxt = self._gen_context(x, lengths_inp, h_prev, soft_in, auths)
        else:
            r0, r1 = grad_rows
            assert compact and not soft_in and h_prev is None and 0 <= r0 < r1 <= B
            outer = torch.is_grad_enabled()

            def part(a_, b_, grad):
                with torch.set_grad_enabled(grad and outer):
                    return self._gen_context(x[:, a_:b_], lengths_inp[a_:b_], None, False, auths[a_:b_])
            spans = [(a_, b_, g_) for a_, b_, g_ in ((0, r0, False), (r0, r1, True), (r1, B, False)) if b_ > a_]
            if len(spans) == 2:
                # the two encoder passes (rows encoded without a graph, rows that will be differentiated) do not use
                # each other's results: their LSTM layers share launches (ops.paired)
                parts = ops.paired(lambda: part(*spans[0]), lambda: part(*spans[1]))
            else:
                parts = [part(*sp) for sp in spans]
            ctxt = torch.cat(list(parts), dim=0)
        if soft_samples:
            self.temp = temp
        dec = self._decoder(auths)
        # a SoftSamples input whose first row is a constant start token (prepend_token) needs no gradient for
        # the decoder's first input, so it can take the fused loop too (the reverse translation of the cycle losses)
        first_const = isinstance(x, SoftSamples) and x.const_rows >= 1
        fused = (self.dec_num_rec_layers == 1 and (not soft_in or first_const) and n_samples == 1 and
                 not self.learn_gumbel and (self.gumb_type or not soft_samples))
        if not fused:
            if grad_rows is not None:
                raise NotImplementedError('grad_rows needs the fused generation path')
            return self._advers_gen_generic(x, soft_in, ctxt, n_max, end_c, soft_samples, temp, n_samples, compact, dec)
        w_ih, w_hh, b_ih, b_hh = rec_layer_params(dec[0], self.en_residual, 0)
        W_out = dec[1]
        Hd, Hp = self.dec_hidden_size, ops.padded_hidden(self.dec_hidden_size)
        if Hp != Hd:
            # a hidden size between two kernel instantiations: zero-weight extra units (ops.pad_lstm_params) and
            # zero rows of the output projection for them
            w_ih, w_hh, b_ih, b_hh = ops.pad_lstm_params(w_ih, w_hh, b_ih, b_hh, Hd, Hp)
            W_out = F.pad(W_out, [0, 0, 0, Hp - Hd])
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        cfg = dict(n_max=n_max, end_c=int(end_c), tau=float(temp), scale=float(self.softmax_scale),
                   hard=bool(self.gumb_type), greedy=not soft_samples, need_grad=need_grad, noise=self._noise,
                   seed=self.noise_seed, step0=self._noise_step, grad_rows=grad_rows)
        tok0 = x.tok[0] if isinstance(x, SoftSamples) else ids_to_device(x[0], dev)
        samples, gen_lens, tok = gen_decode(ctxt, tok0, self.char_emb.weight, w_ih, w_hh, b_ih, b_hh,
                                            W_out, dec[2], cfg)
        self._noise_step += n_max
        char_out = list(tok.unbind(0))
        if not soft_samples:
            return [], gen_lens, char_out
        if compact:
            return samples, gen_lens, char_out
        return list(samples.to_dense().unbind(0)), gen_lens, char_out

    def _advers_gen_generic(self, x, soft_in, ctxt, n_max, end_c, soft_samples, temp, n_samples, compact, dec):
        """Step-by-step composition of the same HIP operators (stacked / residual decoders, soft
        (non straight-through) samples, soft encoder input, n_samples > 1, learnt temperature): follows
        the reference loop literally, including its per-step host check."""
        dec_rec, W_out, b_out = dec
        if compact:
            raise NotImplementedError('compact samples need the fused path (1-layer decoder, gumbel_hard, hard input)')
        dev = self._dev()
        B = x.shape[1]
        if isinstance(x, SoftSamples):
            targ_emb = x.val[0].view(-1, 1) * ops.embedding(x.tok[0], self.char_emb.weight, -1)
        elif soft_in:
            targ_emb = ops.linear_kn(x[0], self.char_emb.weight)
        else:
            targ_emb = ops.embedding(ids_to_device(x[0], dev), self.char_emb.weight, 0)
        dec_bsz = B * n_samples
        ctxt_rep = ctxt.repeat(n_samples, 1)
        dec_inp = torch.cat([ctxt, targ_emb], dim=-1).repeat(n_samples, 1)
        out, dec_hidden = run_stack(dec_rec, dec_inp.view(1, dec_bsz, -1), None, self.dec_num_rec_layers,
                                    self.en_residual, None)
        p_rnn = out.view(dec_bsz, -1)
        char_out, samp_out = [], []
        gen_lens = torch.zeros(dec_bsz, dtype=torch.int32, device=dev)
        prev_done = torch.zeros(dec_bsz, dtype=torch.bool, device=dev)
        for _ in range(n_max):
            dec_out = ops.linear_kn(p_rnn, W_out, b_out)
            if soft_samples:
                U = self._noise.pop(0) if self._noise is not None else None
                if self.learn_gumbel:
                    samp, pred_c = self._gumbel_learnt_tau(dec_out * self.softmax_scale, p_rnn, U)
                else:
                    samp, pred_c = ops.gumbel_dense(dec_out * self.softmax_scale, U, self.noise_seed, self._noise_step,
                                                    float(temp), bool(self.gumb_type))
                    if not self.gumb_type:
                        samp = _DropDoneRowGrads.apply(samp, prev_done)
                self._noise_step += 1
                emb = ops.linear_kn(samp, self.char_emb.weight)
                char_out.append(pred_c)
                samp.data.masked_fill_(prev_done.view(-1, 1), 0.)      # char_translator.py:384
                samp_out.append(samp)
            else:
                pred_c = _greedy_argmax(dec_out)
                char_out.append(pred_c)
                emb = ops.embedding(pred_c, self.char_emb.weight, 0)
            gen_lens += (~prev_done).int()
            prev_done = prev_done | (pred_c == end_c)
            if bool(prev_done.all()):
                break
            dec_inp = torch.cat([ctxt_rep, emb], dim=-1).view(1, dec_bsz, -1)
            out, dec_hidden = run_stack(dec_rec, dec_inp, None, self.dec_num_rec_layers, self.en_residual, dec_hidden)
            p_rnn = out.view(dec_bsz, -1)
        return samp_out, gen_lens, char_out

    def _gumbel_learnt_tau(self, x, p_rnn, U):
        """Gumbel-softmax with one learnt temperature per row, tau = softplus(p_rnn*gumbel_W + b) + 1
        (char_translator.py:379).  softmax((log_softmax(x)+g)/tau) = softmax(log(softmax(x+g))/tau): the
        tau = 1 row kernel supplies softmax(x+g) (and its backward); the per-row rescale, which carries the
        gradient to gumbel_W / gumbel_b, is elementwise torch on that row."""
        tau = F.softplus(ops.linear_kn(p_rnn, self.gumbel_W, self.gumbel_b)) + 1.0           # [B,1]
        self.temp = tau
        y1, _ = ops.gumbel_dense(x, U, self.noise_seed, self._noise_step, 1.0, False)
        ysft = F.softmax(torch.log(y1.clamp_min(1e-37)) / tau, dim=1)
        pred_c = ysft.detach().argmax(dim=1)
        if self.gumb_type:
            hard = torch.zeros_like(ysft).scatter_(1, pred_c.view(-1, 1), 1.0)
            return (hard - ysft.detach()) + ysft, pred_c
        return ysft, pred_c

    def forward_gen(self, x, h_prev=None, auths=None, n_max=100, end_c=-1, soft_samples=False, temp=0.1):
        """B = 1 sampler of the reference's demo script (char_translator.py:403-469); greedy only (the
        reference's soft branch is broken: pred_c undefined, :452-461)."""
        if soft_samples:
            raise NotImplementedError('forward_gen(soft_samples=True) is broken in the reference; not provided')
        was_training = self.training
        self.eval()
        with torch.no_grad():
            _, gl, co = self.forward_advers_gen(x, [x.shape[0]] * x.shape[1], n_max=n_max, end_c=end_c,
                                                soft_samples=False, auths=auths)
        self.train(was_training)
        n = int(gl.max().item())
        return co[:n]


class _DropDoneRowGrads(torch.autograd.Function):
    """Soft (non straight-through) samples are the softmax output itself, and the reference zeroes the
    rows of finished sequences through .data (char_translator.py:384), i.e. in the tensor softmax's
    backward reads: those rows get no gradient at all.  The row kernel keeps its own saved state, so the
    same effect is applied to the incoming gradient here."""

    @staticmethod
    def forward(ctx, samp, done):
        ctx.save_for_backward(done.clone())
        return samp.clone()

    @staticmethod
    def backward(ctx, g):
        (done,) = ctx.saved_tensors
        return g.masked_fill(done.view(-1, 1), 0.), None


def _greedy_argmax(logits):
    B, V = logits.shape
    tok = torch.empty(B, dtype=torch.int64, device=logits.device)
    ops.gumbel_rows(logits.contiguous(), None, 0, 0, 1.0, 1.0, 1, 1, None, None, tok, None, None, None, None, -1)
    return tok
