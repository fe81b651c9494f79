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

Launch: `python bench.py --gpus 1` or
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

V, E, H, A, T, B = 10000, 512, 512, 10, 100, 128
FP32_MFMA_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, chip-level table (v_mfma_f32_*_f32)


def gen_params(batch):
    return dict(vocabulary_size=V, embedding_size=E, enc_hidden_size=H, dec_hidden_size=H, enc_hidden_depth=1,
                dec_hidden_depth=1, en_residual_conn=0, pad_auth_vec=A, num_output_layers=2, batch_size=batch,
                gumbel_hard=True, softmax_scale=3.0, maxpoolrnn=0, drop_prob_emb=0.25, drop_prob_encoder=0.5,
                drop_prob_decoder=0.5)


def eval_params():
    return dict(vocabulary_size=V, embedding_size=E, hidden_size=H, hidden_depth=2, en_residual_conn=0,
                num_output_layers=2, mode='classifier', maxpoolrnn=0, bidir=False, drop_prob_encoder=0.5,
                drop_prob_decoder=0.5)


def make_batches(impl, seed, batch, n):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        b = impl.synthetic_batch(g, T, batch, V)
        out.append(b)
    return out


def adv_step(impl, G, D, optG, optD, batches, it, batch, reducers):
    c_aid = it % 2
    mk = lambda b, a: (b[0], b[1], torch.full((batch,), a, dtype=torch.long), b[2])
    src, tgt, gb = batches[(3 * it) % len(batches)], batches[(3 * it + 1) % len(batches)], batches[(3 * it + 2) % len(batches)]
    # the discriminator's gradient exchange + optimizer step are deferred into the generator update, where they
    # overlap the free-running generation (which does not read the discriminator)
    rd = impl.d_update(G, D, optD, mk(src, c_aid), mk(tgt, 1 - c_aid), c_aid, -1, T, 0.5, reducer=reducers[1],
                       defer=True)
    rg = impl.g_update(G, D, optG, mk(gb, c_aid), c_aid, -1, T, 0.5, 2, cycle_loss_type='ml', cycle_loss_w=1.0,
                       grad_clip=10.0, reducer=reducers[0], before_disc=rd.pop('finish'))
    return rd, rg


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


def pmc_traffic(label):
    """HBM bytes per launch of the roofline kernel from the committed rocprofv3 --pmc passes
    (profiles/r01_pmc_traffic.json, written by tools/pmc_traffic.py from separate FETCH_SIZE and
    WRITE_SIZE runs of this same command); None when that kernel has no entry."""
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')
    if not os.path.exists(path):
        return None
    with open(path) as f:
        tab = json.load(f)
    ent = tab.get('kernels', {}).get(label)
    return ent['hbm_bytes_per_launch'] if ent else None


def _synth(g, batch):
    inps = torch.zeros(T, batch, dtype=torch.long)
    targs = torch.zeros(T, batch, dtype=torch.long)
    body = torch.randint(3, V + 1, (T, batch), generator=g)
    inps[0] = 2
    inps[1:] = body[:T - 1]
    targs[:T - 1] = body[:T - 1]
    targs[T - 1] = 1
    return inps, targs, [T] * batch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=B, help='sentences per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--roofline-kernel', default='sgemm_tn_128',
                    help='kernel timed with HIP events inside the timed region (default: the kernel with the '
                         'largest share of the step in profiles/: sgemm_kernel<true,false,2,2,2>)')
    ap.add_argument('--no-extra-rooflines', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    import a4nt_amd as impl
    from a4nt_amd import ops
    if world > 1:
        # leave half of every CU's registers to the RCCL kernels that run beside the persistent LSTM sweeps
        impl.lstm_set_max_workgroups(256)

    torch.manual_seed(0)
    G = impl.CharTranslator(gen_params(args.batch))
    D = impl.CharLstm(eval_params())
    G.noise_seed = 1000 + rank
    G.train(); D.train()
    optG, optD = impl.make_optimizers(G, D)
    reducers = (impl.GradAllReducer(list(G.parameters())), impl.GradAllReducer(list(D.parameters())))
    batches = make_batches(impl, 1234 + rank, args.batch, 6)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for it in range(args.warmup):
        adv_step(impl, G, D, optG, optD, batches, it, args.batch, reducers)
    ops.check_timeouts()
    ops.TIMER = {'name': args.roofline_kernel, 'events': []}
    sync()
    t0 = time.perf_counter()
    for it in range(args.steps):
        rd, rg = adv_step(impl, G, D, optG, optD, batches, args.warmup + it, args.batch, reducers)
    sync()
    dt = time.perf_counter() - t0
    timer, ops.TIMER = ops.TIMER, None
    ops.check_timeouts()
    tmax = torch.tensor([dt], device='cuda')
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
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
        for name in ('lstm_seq_fwd', 'lstm_seq_bwd', 'proj_gumbel', 'sgemm_nt_128', 'sgemm_nn_128', 'sgemm_nt_64'):
            if name == timer['name']:
                continue
            ops.TIMER = {'name': name, 'events': []}
            adv_step(impl, G, D, optG, optD, batches, args.warmup + args.steps, args.batch, (None, None))
            torch.cuda.synchronize()
            tm, ops.TIMER = ops.TIMER, None
            others.append(summarise(tm))

    if rank == 0:
        ms = dt * 1000.0 / args.steps
        main_rl = summarise(timer)
        line = {
            'metric': 'adversarial train sentences/sec (node)', 'value': world * args.batch / (dt / args.steps),
            'unit': 'sentences/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'full A4NT adversarial step (D update + G update with Gumbel-softmax sampling, '
                                   'char_lstm discriminator, ml reconstruction loss); translator enc/dec 1x512, '
                                   'discriminator 2x512, V=10001, E=512, T=100, batch %d/GPU, all lens=T' % args.batch,
                       'global_batch': world * args.batch, 'parallelism': 'dp%d' % world, 'losses_finite': loss_ok},
            'roofline': dict({'bound': 'mfma', 'traffic': pmc_traffic(timer['name'])}, **main_rl),
        }
        if others:
            line['roofline']['other_kernels'] = others
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline()
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
