#!/usr/bin/env python3
"""Time of one free-running generation (a4nt_gen_loop_fwd: 100 steps, H = 512, V = 10001, Philox noise) at 256 and
128 rows; with A4NT_HIP_LIB pointing at a diagnostic build it prices single parts of the decoder step."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import a4nt_amd  # noqa: F401
from a4nt_amd import ops, _lib

lib = _lib.load()
dev = torch.device('cuda')
H, V, n_max = 512, 10001, 100


def make(B, seed):
    g = torch.Generator(device='cuda').manual_seed(seed)
    r = lambda *s, sc=0.1: (torch.randn(*s, device=dev, generator=g) * sc)
    W = r(H, V)
    keep = dict(cg=r(B, 4 * H), ptab=r(V, 4 * H), tok0=torch.full((B,), 2, dtype=torch.int64, device=dev),
                ones=torch.ones(B, device=dev), whh=r(4 * H, H, sc=0.05), hs_all=torch.zeros(n_max + 1, B, H, device=dev),
                gates=torch.empty(n_max, B, 4 * H, device=dev), cs=torch.empty(n_max, B, H, device=dev),
                xch=torch.zeros(int(lib.a4nt_lstm_xch_floats(B, H, 0)), device=dev), Wp=ops.proj_gumbel_pack(W),
                bias=r(V), z=torch.empty(n_max, B, V, device=dev),
                part=torch.empty(int(lib.a4nt_proj_gumbel_part_floats(B, V)), device=dev),
                stats=torch.empty(n_max, B, 2, device=dev), tok=torch.zeros(n_max, B, dtype=torch.int64, device=dev),
                val_in=torch.ones(n_max, B, device=dev), val_out=torch.zeros(n_max, B, device=dev),
                gen_lens=torch.zeros(B, dtype=torch.int32, device=dev), done=torch.zeros(B, dtype=torch.uint8, device=dev))
    fa = _lib.GenFwdArgs()
    fa.B, fa.H, fa.V, fa.n_max = B, H, V, n_max
    p = ops._p
    fa.cg, fa.ptab, fa.tok0, fa.ones, fa.whh = p(keep['cg']), p(keep['ptab']), p(keep['tok0']), p(keep['ones']), p(keep['whh'])
    fa.hs_all, fa.gates, fa.cs, fa.xch = p(keep['hs_all']), p(keep['gates']), p(keep['cs']), p(keep['xch'])
    fa.W, fa.ldw, fa.bias = p(keep['Wp']), 0, p(keep['bias'])
    fa.scale, fa.tau, fa.hard, fa.greedy = 3.0, 0.5, 1, 0
    fa.U, fa.n_noise, fa.ldu = None, 0, 0
    fa.seed, fa.step0 = 0, 0
    fa.z, fa.ldz, fa.zrow0, fa.zrows = p(keep['z']), V, 0, B
    fa.part, fa.stats, fa.tok, fa.val_in, fa.val_out = p(keep['part']), p(keep['stats']), p(keep['tok']), p(keep['val_in']), p(keep['val_out'])
    fa.gen_lens, fa.done, fa.end_c = p(keep['gen_lens']), p(keep['done']), -1
    return fa, keep


def run(fa, stream):
    keep['xch'].zero_()
    _lib.check(lib.a4nt_gen_loop_fwd(ctypes.byref(fa), 0, n_max, stream.cuda_stream), 'a4nt_gen_loop_fwd')


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def run_persist(fa, ws, stream):
    ws[0].zero_()
    _lib.check(lib.a4nt_gen_persist_fwd(ctypes.byref(fa), 0, n_max, ops._p(ws[0]), ops._p(ws[1]), stream.cuda_stream),
               'a4nt_gen_persist_fwd')


ops._ensure_init()
main = torch.cuda.current_stream()
for B in (256, 128):
    fa, keep = make(B, B)
    ms = timed(lambda: run(fa, main), reps=10)
    print('generation forward, %d rows: %.3f ms (%.1f us per step)' % (B, ms, ms * 1000.0 / n_max))
    tok_steps = keep['tok'].clone()
    ws = (torch.zeros(int(lib.a4nt_gen_persist_xch_floats(B, H)), device=dev),
          torch.zeros(int(lib.a4nt_gen_persist_sync_words()), device=dev, dtype=torch.int32))
    keep['xch'].zero_()
    ms = timed(lambda: run_persist(fa, ws, main), reps=10)
    print('  one persistent launch, %d rows: %.3f ms (%.1f us per step), tokens equal: %s, timeout word %d'
          % (B, ms, ms * 1000.0 / n_max, bool(torch.equal(tok_steps, keep['tok'])), int(ws[1][32 * 16])))
    st = ws[1][32 * 16 + 8: 32 * 16 + 64].cpu().numpy().astype('uint32')
    if st.any():            # -DGP_STAMP build (tools/gp_stamp_build.sh): phase sums of workgroup 0's P wave 0 and L wave 0
        for name, off, labels in (('P wave', 0, ['barrier 1', 'poll h_t', 'A loads + k loop', 'barriers 2-3', 'epilogue',
                                                  'drain', 'barrier 4', 'noise']),
                                  ('L wave', 24, ['poll strips + barrier 1', 'merge', 'gather + cell update',
                                                  'barrier 2, publish, barrier 3', 'output stores', 'poll h_t', 'recurrent product',
                                                  'barrier 4'])):
            v = [int(st[off + 2 * k]) | (int(st[off + 2 * k + 1]) << 32) for k in range(8)]
            tot = float(sum(v)) or 1.0
            print('   %s (share of its step, x %.1f us):' % (name, ms * 1000.0 / n_max))
            for lb, x in zip(labels, v):
                print('      %-32s %5.1f %%  %6.2f us' % (lb, 100.0 * x / tot, x / tot * ms * 1000.0 / n_max))

# The same loop replayed from a captured graph (hipGraph through torch.cuda.CUDAGraph): does the GPU run the 200
# dependent launches of a generation closer together when they come from one graph launch?
if os.environ.get('A4NT_GRAPH', '1') != '0':
    for B in (256, 128):
        fa, keep = make(B, B)
        run(fa, main)
        torch.cuda.synchronize()
        ref_tok = keep['tok'].clone()
        ms_direct = timed(lambda: run(fa, main), reps=10)
        side = torch.cuda.Stream()
        side.wait_stream(main)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            run(fa, side)                                # warm-up on the capture stream
            side.synchronize()
            with torch.cuda.graph(graph, stream=side):
                run(fa, torch.cuda.current_stream())
        main.wait_stream(side)
        keep['tok'].zero_()
        graph.replay()
        torch.cuda.synchronize()
        same = bool(torch.equal(ref_tok, keep['tok']))
        ms_graph = timed(graph.replay, reps=10)
        print('generation forward, %d rows: direct %.3f ms, graph replay %.3f ms (%.1f vs %.1f us per step), tokens equal: %s'
              % (B, ms_direct, ms_graph, ms_direct * 10.0, ms_graph * 10.0, same))
