#!/usr/bin/env python3
"""bench.py -- A4NT adversarial training step on MI355X (BASELINE.json configs[2] / configs[3]).

One "step" = one full adversarial step at steady state (iters_eval = iters_gen = 1):
  D update: generator free-run (Gumbel-softmax, B sentences) -> discriminator on the B generated +
            B real sentences -> BCE -> backward -> RMSprop;
  G update: generator free-run with graph -> discriminator -> 5*BCE + 'ml' reconstruction CE through
            the teacher-forced pass -> BPTT -> clip_grad_norm(10) -> RMSprop.
Workload: translator enc/dec 1x512, CharLstm discriminator 2x512, V = 10001, E = 512,
T = 100, B = 128 per GPU, all lens = T, end_c = -1 (every generation runs T steps), synthetic ids,
random-init weights (torch.manual_seed(0)), dropout at the reference CLI defaults, fp32.

Launch: `python bench.py --gpus N` (N > 1 and no WORLD_SIZE in the environment: this process starts N ranks
itself through `torch.distributed.run`, before anything touches the GPU, and exits with their return code) or
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (one rank per GPU, RCCL).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

V, E, H, A, T, B = 10000, 512, 512, 10, 100, 128       # BASELINE.json configs[2] (the metric's configuration)
FP32_MFMA_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, chip-level table (v_mfma_f32_*_f32)

# BASELINE.json configs by name.  'adv' (configs[2]/[3]) is what the metric is quoted on and the default;
# the others are reported in DESIGN.md and never replace the default line.
WORKLOADS = {
    'adv': dict(V=10000, E=512, H=512, T=100, B=128, disc='lstm',
                text='full A4NT adversarial step (D update + G update with Gumbel-softmax sampling, char_lstm '
                     'discriminator, ml reconstruction loss); translator enc/dec 1x512, discriminator 2x512'),
    'cfg1': dict(V=10000, E=512, H=512, T=100, B=64, disc='lstm',
                 text='char_lstm attribution classifier 2x512 forward+backward+RMSprop step'),
    'cfg2': dict(V=10000, E=512, H=512, T=100, B=64, disc=None,
                 text='char_translator seq2seq enc/dec 1x512 teacher-forced training step (fused projection+NLL)'),
    'cfg5': dict(V=30000, E=512, H=1024, T=200, B=256, disc='cnn',
                 text='adversarial step, stress shape: translator enc/dec 1x1024, char_cnn discriminator'),
}


def set_workload(name):
    global V, E, H, T, B
    w = WORKLOADS[name]
    V, E, H, T, B = w['V'], w['E'], w['H'], w['T'], w['B']
    return w


def gen_params(batch):
    return dict(vocabulary_size=V, embedding_size=E, enc_hidden_size=H, dec_hidden_size=H, enc_hidden_depth=1,
                dec_hidden_depth=1, en_residual_conn=0, pad_auth_vec=A, num_output_layers=2, batch_size=batch,
                gumbel_hard=True, softmax_scale=3.0, maxpoolrnn=0, drop_prob_emb=0.25, drop_prob_encoder=0.5,
                drop_prob_decoder=0.5)


def eval_params(kind='lstm'):
    if kind == 'cnn':
        # train.py:214-216 defaults: 1 layer of 100 filters for each of the kernel sizes 2, 3, 4
        return dict(vocabulary_size=V, embedding_size=E, num_output_layers=2, modeltype='cnn', drop_prob_encoder=0.5,
                    drop_prob_decoder=0.5, decoder_cnn_ks=[2, 3, 4], decoder_cnn_nfilt=100, decoder_cnn_nlayers=1)
    return dict(vocabulary_size=V, embedding_size=E, hidden_size=512, hidden_depth=2, en_residual_conn=0,
                num_output_layers=2, mode='classifier', maxpoolrnn=0, bidir=False, drop_prob_encoder=0.5,
                drop_prob_decoder=0.5)


def ragged_lens(gen, batch):
    """Speech-shaped lengths: clip(round(LogNormal(ln 18, 0.5)) + 2, 4, T), sorted descending."""
    z = torch.randn(batch, generator=gen) * 0.5 + 2.8903717578961645
    return sorted([int(min(max(round(float(v)) + 2, 4), T)) for v in z.exp()], reverse=True)


def make_batches(impl, seed, batch, n, ragged=False):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        b = impl.synthetic_batch(g, T, batch, V, lens=ragged_lens(g, batch) if ragged else None)
        # pinned, as DataProvider.prepare_data stages them: the modules' id copies are then asynchronous DMAs
        out.append((b[0].pin_memory(), b[1].pin_memory(), b[2]))
    return out


def clf_step(impl, D, optD, batches, it, batch):
    """configs[0]: classifier forward + cross-entropy + backward + RMSprop (train.py:94-139)."""
    import torch.nn.functional as F
    _, targs, lens = batches[it % len(batches)]
    labels = torch.full((batch,), it % 2, dtype=torch.long, device='cuda')
    optD.zero_grad()
    logits, _ = D.forward_classify(targs, lens=lens)
    loss = F.cross_entropy(logits, labels)
    loss.backward()
    optD.step()
    return {'loss': loss.detach()}, {'loss': loss.detach()}


def ml_step(impl, G, optG, batches, it, batch):
    """configs[1]: teacher-forced translator step (train_translator.py:93-125)."""
    inps, targs, lens = batches[it % len(batches)]
    auths = torch.full((batch,), it % 2, dtype=torch.long)
    optG.zero_grad()
    loss = G.forward_mltrain_loss(inps, lens, inps, lens, targs, auths=auths)
    loss.backward()
    impl.ops.clip_grad_norm_(G.parameters(), 10.0)
    optG.step()
    return {'loss': loss.detach()}, {'loss': loss.detach()}


def adv_step(impl, G, D, optG, optD, batches, it, batch, reducers, fused=True):
    c_aid = it % 2
    mk = lambda b, a: (b[0], b[1], torch.full((batch,), a, dtype=torch.long), b[2])
    src, tgt, gb = batches[(3 * it) % len(batches)], batches[(3 * it + 1) % len(batches)], batches[(3 * it + 2) % len(batches)]
    if fused:
        # one free-running generation for the 2B source sentences of the two updates (adv_step.fused_update)
        return impl.fused_update(G, D, optG, optD, mk(src, c_aid), mk(tgt, 1 - c_aid), mk(gb, c_aid), c_aid, -1, T, 0.5,
                                 2, cycle_loss_type='ml', cycle_loss_w=1.0, grad_clip=10.0, reducers=reducers)
    # the discriminator's gradient exchange + optimizer step are deferred into the generator update, where they
    # overlap the free-running generation (which does not read the discriminator)
    rd = impl.d_update(G, D, optD, mk(src, c_aid), mk(tgt, 1 - c_aid), c_aid, -1, T, 0.5, reducer=reducers[1],
                       defer=True)
    rg = impl.g_update(G, D, optG, mk(gb, c_aid), c_aid, -1, T, 0.5, 2, cycle_loss_type='ml', cycle_loss_w=1.0,
                       grad_clip=10.0, reducer=reducers[0], before_disc=rd.pop('finish'))
    return rd, rg


def step_flops(workload, batch):
    """Algorithmic FLOPs of one step (forward = 2 per multiply-add, backward with weight gradients =
    2 x forward, data gradient only = 1 x forward; the hard Gumbel sample makes every soft-embedding FORWARD a gather,
    its backward is the dense d(emb) x E^T).  Per token and sentence:
      LSTM layer 8 H (I + H); vocabulary projection 2 H V; soft-embedding backward 2 V E; CharCNN sum_K 2 E F K.
    Adversarial step: the D update runs the generation forward (enc + dec + proj) and the discriminator forward +
    backward on the generated and the real batch; the G update runs the generation forward, the discriminator
    forward + data gradient, its soft-embedding backward, the reconstruction pass (enc + dec + proj, forward + full
    backward, + the soft-embedding backward of its input) and the generation's full backward with the per-step
    soft-embedding backward: 7 gen + 8 disc + 3 softemb in all."""
    Vv, A_ = V + 1, A
    enc = 8.0 * H * (E + H)
    dec = 8.0 * H * ((H + E + A_) + H)
    proj = 2.0 * H * Vv
    softemb = 2.0 * Vv * E
    gen = enc + dec + proj
    if WORKLOADS[workload]['disc'] == 'cnn':
        disc = sum(2.0 * E * 100 * k for k in (2, 3, 4))
    else:
        disc = 2 * 8.0 * 512 * (512 + 512)                 # CharLstm 2 x 512 on E = 512 inputs
    per_token = {'cfg1': 3 * disc, 'cfg2': 3 * gen}.get(workload, 7 * gen + 8 * disc + 3 * softemb)
    return per_token * T * batch


def cpu_baseline(batch_cpu=B):
    """The oracle (pure-PyTorch CPU restatement pinned to the reference by tests/golden) timed on
    this host's cores on a bounded sample: ONE adversarial step at the benchmark's batch (about 10-25 s
    on the GPU box's 16-thread CPU share)."""
    import oracle
    torch.manual_seed(0)
    threads = min(os.cpu_count() or 1, int(os.environ.get('OMP_NUM_THREADS', os.cpu_count() or 1)))
    torch.set_num_threads(threads)
    gp, ep = gen_params(batch_cpu), eval_params()
    G, D = oracle.CharTranslator(gp), oracle.CharLstm(ep)
    G.train(); D.train()
    optG, optD = oracle.make_optimizers(G, D)
    g = torch.Generator().manual_seed(99)
    mk = lambda a: (lambda b: (b[0], b[1], torch.full((batch_cpu,), a, dtype=torch.long), b[2]))(
        _synth(g, batch_cpu))
    append = torch.zeros(1, batch_cpu, V + 1)
    append[:, :, 2] = 1.0
    t0 = time.time()
    oracle.d_update(G, D, optD, mk(0), mk(1), 0, -1, T, 0.5)
    oracle.g_update(G, D, optG, mk(0), 0, -1, T, 0.5, append, cycle_loss_type='ml', cycle_loss_w=1.0)
    dt = time.time() - t0
    return {'value': batch_cpu / dt, 'unit': 'sentences/s', 'cores': threads, 'kind': 'port',
            'sample': '1 adversarial step (D update + G update with ml reconstruction) at batch %d of %d, T=%d, '
                      'V=%d, oracle/ on torch CPU fp32, %.1f s' % (batch_cpu, B, T, V + 1, dt)}


def _latest_profile(stem, workload='adv'):
    """profiles/rNN_<stem>.json (the metric's workload) or profiles/rNN_<workload>_<stem>.json of the latest round
    that has one."""
    import glob
    if workload != 'adv':
        stem = '%s_%s' % (workload, stem)
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_%s.json' % stem)))
    return files[-1] if files else None


_KERNEL_LABELS = (                       # rocprofv3 kernel name prefix -> ops._Timed label
    ('void sgemm_kernel<true, false, 2, 2, 2, 1, false, true>', 'sgemm_tn_128_colsum'),
    ('void sgemm_kernel<true, false, 2, 2,', 'sgemm_tn_128'), ('void sgemm_kernel<false, true, 2, 2,', 'sgemm_nt_128'),
    ('void sgemm_kernel<false, false, 2, 2,', 'sgemm_nn_128'), ('void sgemm_kernel<false, true, 1, 1,', 'sgemm_nt_64'),
    ('void proj_gumbel_kernel', 'proj_gumbel'), ('void lstm_fwd', 'lstm_seq_fwd'), ('void lstm_bwd', 'lstm_seq_bwd'))


def dominant_kernel(workload='adv'):
    """(label, source file) of the kernel with the largest share of the step in the latest committed rocprofv3
    summary of this workload (profiles/rNN_bench_kernel_stats*.csv for the metric's workload,
    profiles/rNN_<workload>_kernel_stats*.csv otherwise); None when there is no summary or its top kernel is not
    one the event timer knows."""
    import csv
    import glob
    stem = 'bench' if workload == 'adv' else workload
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_%s_kernel_stats*.csv' % stem)),
                   key=lambda f: (os.path.basename(f)[:3], len(f), f))
    if not files:
        return None
    with open(files[-1]) as f:
        rows = list(csv.DictReader(f))
    if not rows:
        return None
    top = max(rows, key=lambda r: float(r['Percentage']))
    for prefix, label in _KERNEL_LABELS:
        if top['Name'].startswith(prefix):
            return label, os.path.relpath(files[-1], ROOT)
    return None


def pmc_traffic(label, workload='adv'):
    """HBM bytes per launch of the roofline kernel from the committed rocprofv3 --pmc passes
    (profiles/rNN_pmc_traffic.json, written by tools/pmc_traffic.py from separate FETCH_SIZE and
    WRITE_SIZE runs of this same command); None when that kernel has no entry."""
    path = _latest_profile('pmc_traffic', workload)
    if path is None:
        return None
    with open(path) as f:
        tab = json.load(f)
    ent = tab.get('kernels', {}).get(label)
    return ent['hbm_bytes_per_launch'] if ent else None


def lstm_mfma_util(workload='adv'):
    """BASELINE.json's second metric: matrix-pipe utilisation of the LSTM kernels (MFMA-busy cycles / (active cycles
    x 1024 SIMDs)) from the committed counter pass over this same command (profiles/rNN_pmc_mfma.json,
    tools/pmc_mfma.py).  Counters cannot be read from inside the timed run, so these are the committed profile's
    numbers, named with their source; None when no profile is committed."""
    path = _latest_profile('pmc_mfma', workload)
    if path is None:
        return None
    with open(path) as f:
        tab = json.load(f)
    out = {'source': os.path.relpath(path, ROOT)}
    for key, lab in (('fwd_sweep', 'lstm_fwd'), ('fwd_sweep_pair', 'lstm_fwd_pair'), ('bwd_sweep', 'lstm_bwd'),
                     ('bwd_sweep_pair', 'lstm_bwd_pair'), ('decoder_step', 'lstm_dec_step'),
                     ('fwd_sweeps_and_decoder_steps', 'lstm_fwd (sweeps and decoder steps)')):
        if lab in tab:
            out[key] = round(tab[lab]['mfma_util'], 4)
    return out


def _synth(g, batch):
    inps = torch.zeros(T, batch, dtype=torch.long)
    targs = torch.zeros(T, batch, dtype=torch.long)
    body = torch.randint(3, V + 1, (T, batch), generator=g)
    inps[0] = 2
    inps[1:] = body[:T - 1]
    targs[:T - 1] = body[:T - 1]
    targs[T - 1] = 1
    return inps, targs, [T] * batch


def self_launch_cmd(n_gpus, argv, port=None):
    """Command line that runs this script as `n_gpus` ranks of one node (one process per GPU, rendezvous on
    127.0.0.1: the container's hostname may not resolve).  `argv` = this process's own arguments, passed through."""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(int(n_gpus)),
            '--master-addr', '127.0.0.1', '--master-port', str(int(port)), os.path.abspath(__file__)] + list(argv)


def maybe_self_launch(args, argv, environ=None, device_count=None, run=None):
    """`--gpus N` with N > 1 outside a torch.distributed launch: start the N ranks as fresh child processes and
    return their exit code (None: nothing to do, this process is the single rank or already one of N).  Runs
    before any HIP call of this process -- torch.cuda.device_count() does not initialise the GPU -- and never
    replaces the running program: the children are started with subprocess and this process only waits."""
    environ = os.environ if environ is None else environ
    if 'WORLD_SIZE' in environ:
        world = int(environ['WORLD_SIZE'])
        if world != args.gpus:
            raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world))
        return None
    if args.gpus <= 1:
        return None
    have = torch.cuda.device_count() if device_count is None else device_count
    if getattr(args, 'share_gpu', False):
        have = args.gpus                 # rehearsal: every rank on device 0
    if args.gpus > have:
        raise SystemExit('bench.py: --gpus %d asked for, %d GPU(s) visible on this node' % (args.gpus, have))
    import subprocess
    cmd = self_launch_cmd(args.gpus, argv)
    if run is None:
        run = lambda c: subprocess.run(c).returncode
    return run(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='adv', choices=sorted(WORKLOADS), help='default: the metric\'s configuration')
    ap.add_argument('--ragged', action='store_true', help='speech-shaped sentence lengths instead of all = T')
    ap.add_argument('--batch', type=int, default=None, help='sentences per GPU (default: the workload\'s)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--separate-gen', action='store_true',
                    help='generate for the D update and for the G update in two passes of B rows (the reference\'s '
                         'loop structure) instead of one pass of 2B rows')
    ap.add_argument('--roofline-kernel', default=None,
                    help='kernel timed with HIP events inside the timed region (default: the kernel with the '
                         'largest share of the step in the latest committed rocprofv3 summary under profiles/)')
    ap.add_argument('--timed-launches', type=int, default=1024,
                    help='HIP event pairs made before the timed region for the roofline kernel (launches beyond go untimed)')
    ap.add_argument('--no-extra-rooflines', action='store_true')
    ap.add_argument('--no-pack-w', action='store_true', help='fused projection reads the [H,V] matrix as stored')
    ap.add_argument('--py-loops', action='store_true',
                    help='drive the decoder\'s step loops from Python instead of the library\'s C++ loops (A/B)')
    ap.add_argument('--lstm-workgroups', type=int, default=0, choices=(0, 256, 512),
                    help='persistent LSTM sweep residency (0: 512 on one GPU, 256 beside RCCL kernels when N > 1)')
    ap.add_argument('--dp-rehearsal', action='store_true',
                    help='one GPU, but with the data-parallel machinery live: RCCL process group of one rank, gradient '
                         'buckets all-reduced under backward(), one LSTM workgroup per CU (what every rank of an N > 1 run does)')
    ap.add_argument('--share-gpu', action='store_true',
                    help='rehearsal of the N > 1 path on a one-GPU box: every rank runs on device 0 and the collectives go '
                         'over gloo (RCCL refuses two ranks on one device); use a small --batch so that the ranks\' '
                         'persistent sweeps are co-resident.  The number is not a benchmark result.')
    args = ap.parse_args()
    rc = maybe_self_launch(args, sys.argv[1:])
    if rc is not None:
        sys.exit(rc)
    # stdout carries exactly one line, the result: whatever libraries print on the way (RCCL announces its version on
    # stdout when a communicator is created) goes to stderr instead
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    wl = set_workload(args.workload)
    roofline_source = 'command line'
    if args.roofline_kernel is None:
        dom = dominant_kernel(args.workload) or dominant_kernel('adv')
        args.roofline_kernel, roofline_source = dom if dom else ('sgemm_nt_128', 'default (no summary committed)')
    if args.batch is None:
        args.batch = B

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    if world > 1 and args.share_gpu:
        dist.init_process_group('gloo')
    elif world > 1:
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    elif args.dp_rehearsal:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', local_rank))
    import a4nt_amd as impl
    from a4nt_amd import ops
    if args.no_pack_w:
        from a4nt_amd import gen_decode
        gen_decode.PACK_W = False
    if args.py_loops:
        from a4nt_amd import gen_decode
        gen_decode.PY_LOOPS = True
    if args.share_gpu:
        impl.lstm_set_max_workgroups(256)
    elif args.lstm_workgroups:
        impl.lstm_set_max_workgroups(args.lstm_workgroups)
    elif world > 1 or args.dp_rehearsal:
        # leave half of every CU's registers to the RCCL kernels that run beside the persistent LSTM sweeps.  On the
        # fused path every exchange has been waited for before the next forward pass starts (the gradient buckets
        # travel under the BACKWARD passes only), so the forward sweeps keep two workgroups per CU, and so do the
        # backward sweeps queued before the pass's first bucket goes out (ops.BWD_BUDGET_DYNAMIC: the budget applies
        # while a collective is in flight); with two generation passes the discriminator's exchange runs under the
        # generator's encoder sweeps: one per CU throughout
        if args.separate_gen:
            impl.lstm_set_max_workgroups(256)
        else:
            ops.BWD_MAX_WGS = 256

    torch.manual_seed(0)
    G = impl.CharTranslator(gen_params(args.batch)) if args.workload != 'cfg1' else None
    D = None
    if wl['disc']:
        D = impl.CharCNN(eval_params('cnn')) if wl['disc'] == 'cnn' else impl.CharLstm(eval_params())
    if G is not None:
        G.noise_seed = 1000 + rank
        G.train()
    if D is not None:
        D.train()
    if G is not None and D is not None:
        optG, optD = impl.make_optimizers(G, D)
    elif G is not None:
        optG, optD = impl.ops.RMSprop(G.parameters(), lr=1e-4, alpha=0.99, eps=1e-8), None
    else:
        optG, optD = None, impl.ops.RMSprop(D.parameters(), lr=1e-3, alpha=0.99, eps=1e-8)
    force = bool(args.dp_rehearsal)
    reducers = (impl.GradAllReducer(list(G.parameters()), force=force) if G is not None else None,
                impl.GradAllReducer(list(D.parameters()), force=force) if D is not None else None)
    batches = make_batches(impl, 1234 + rank, args.batch, 6, ragged=args.ragged)
    if args.workload == 'cfg1':
        run_step = lambda it, red: clf_step(impl, D, optD, batches, it, args.batch)
    elif args.workload == 'cfg2':
        run_step = lambda it, red: ml_step(impl, G, optG, batches, it, args.batch)
    else:
        run_step = lambda it, red: adv_step(impl, G, D, optG, optD, batches, it, args.batch, red,
                                            fused=not args.separate_gen)
    if world > 1:
        dist.barrier()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for it in range(args.warmup):
        run_step(it, reducers)
    ops.check_timeouts()
    ops.TIMER = ops.make_timer(args.roofline_kernel, max_launches=args.timed_launches)
    sync()
    t0 = time.perf_counter()
    for it in range(args.steps):
        rd, rg = run_step(args.warmup + it, reducers)
    sync()
    dt = time.perf_counter() - t0
    timer, ops.TIMER = ops.TIMER, None
    ops.check_timeouts()
    coll_dev = 'cpu' if args.share_gpu else 'cuda'
    tmax = torch.tensor([dt], device=coll_dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    # how many ranks the collective library really joined (1 without a process group)
    seen = torch.ones(1, device=coll_dev)
    if world > 1 or args.dp_rehearsal:
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)
    ranks_seen = int(seen.item())
    loss_ok = bool(torch.isfinite(rd['loss']).item() and torch.isfinite(rg['loss']).item())

    def summarise(tm):
        flops = sum(f for _, _, f in tm['events'])
        secs = sum(e0.elapsed_time(e1) for e0, e1, _ in tm['events']) / 1000.0
        n = max(1, len(tm['events']))
        ach = flops / secs / 1e12 if secs > 0 else 0.0
        return {'kernel': tm['name'], 'achieved': ach, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': ach / FP32_MFMA_PEAK_TFLOPS, 'launches': len(tm['events']),
                'avg_launch_ms': secs * 1000.0 / n, 'gflop_per_launch': flops / n / 1e9}

    # the other MFMA kernels of the step, one untimed extra step each (outside the timed region)
    others = []
    if world == 1 and not args.no_extra_rooflines:
        for name in ('sgemm_tn_128', 'sgemm_nn_128', 'lstm_seq_fwd', 'lstm_seq_bwd', 'proj_gumbel', 'sgemm_nt_64'):
            if name == timer['name']:
                continue
            ops.TIMER = {'name': name, 'events': []}
            run_step(args.warmup + args.steps, (None, None))
            torch.cuda.synchronize()
            tm, ops.TIMER = ops.TIMER, None
            others.append(summarise(tm))

    if rank == 0:
        ms = dt * 1000.0 / args.steps
        main_rl = summarise(timer)
        line = {
            'metric': ('adversarial train sentences/sec (node)' if args.workload in ('adv', 'cfg5')
                       else 'train sentences/sec (node)'), 'value': world * args.batch / (dt / args.steps),
            'unit': 'sentences/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s, V=%d, E=%d, T=%d, batch %d/GPU, %s' % (
                           wl['text'], V + 1, E, T, args.batch,
                           'speech-shaped ragged lens' if args.ragged else 'all lens=T'),
                       'global_batch': world * args.batch, 'parallelism': 'dp%d' % world, 'losses_finite': loss_ok,
                       **({'rehearsal': 'all ranks share GPU 0, gloo collectives: not a benchmark result'}
                          if args.share_gpu else {})},
            'ranks_seen': ranks_seen,
            # whole step against the fp32 matrix peak: algorithmic FLOPs of the step (step_flops) / wall time per step
            'end_to_end': {'tflop_per_step_per_gpu': step_flops(args.workload, args.batch) / 1e12,
                           'achieved_tflops_per_gpu': step_flops(args.workload, args.batch) / (dt / args.steps) / 1e12,
                           'frac_of_fp32_mfma_peak': step_flops(args.workload, args.batch) / (dt / args.steps) / 1e12 / FP32_MFMA_PEAK_TFLOPS},
            'roofline': dict({'bound': 'mfma', 'traffic': pmc_traffic(timer['name'], args.workload),
                              'kernel_chosen_from': roofline_source}, **main_rl),
        }
        util = lstm_mfma_util(args.workload)
        if util is not None and args.workload in ('adv', 'cfg5'):
            line['lstm_mfma_util'] = util
        if others:
            line['roofline']['other_kernels'] = others
        if world == 1 and not args.no_cpu_baseline and args.workload == 'adv' and not args.ragged:
            line['cpu_baseline'] = cpu_baseline()
        os.write(result_fd, (json.dumps(line) + '\n').encode())
    if world > 1 or args.dp_rehearsal:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
