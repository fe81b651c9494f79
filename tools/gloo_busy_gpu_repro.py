#!/usr/bin/env python3
"""Stand-alone reproduction attempt for the stall DESIGN.md section 7 describes: two processes that share ONE GPU,
each with work queued on the device (plain torch matmuls -- no kernel of this repository is loaded), issue an ASYNC
gloo all-reduce of a CUDA tensor and wait for it.  gloo stages device tensors through the host itself; the question
is whether those async collectives complete while the GPU has queued work.

Every wait is bounded (work.wait(timeout)); a rank that does not finish is killed by the parent, which prints what
each rank logged.  Usage (GPU box): python tools/gloo_busy_gpu_repro.py [trials] [matmuls queued per trial]
The parent never touches the GPU."""
import datetime
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, trials, depth, log_path, host_staged):
    log = open(log_path, 'w', buffering=1)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    x = torch.randn(4096, 4096, device='cuda')
    worst = 0.0
    for trial in range(trials):
        for _ in range(depth):                      # ~1 ms of GPU work each, queued ahead of the collective
            y = x @ x
        t = torch.full((1 << 20,), float(rank + 1), device='cuda')
        t0 = time.time()
        try:
            if host_staged:
                h = t.cpu()                         # what GradAllReducer.host_staged does: drains this rank's queue
                work = dist.all_reduce(h, async_op=True)
                work.wait(timeout=datetime.timedelta(seconds=30))
                t.copy_(h)
            else:
                work = dist.all_reduce(t, async_op=True)
                work.wait(timeout=datetime.timedelta(seconds=30))
            torch.cuda.synchronize()
        except Exception as e:                      # a timeout raises here
            log.write('rank %d trial %d: all-reduce did NOT complete within 30 s: %s\n' % (rank, trial, str(e)[:200]))
            log.write('RESULT rank %d: STALL at trial %d\n' % (rank, trial))
            os._exit(3)
        dt = time.time() - t0
        worst = max(worst, dt)
        ok = abs(float(t[0]) - 3.0) < 1e-6
        log.write('rank %d trial %d: %.3f s, sum %s\n' % (rank, trial, dt, 'ok' if ok else 'WRONG %r' % float(t[0])))
    log.write('RESULT rank %d: all %d trials completed, slowest %.3f s\n' % (rank, trials, worst))
    dist.barrier()
    dist.destroy_process_group()


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    depth = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    ctx = mp.get_context('spawn')
    for host_staged in (False, True):
        print('=== %s, %d trials, %d matmuls queued per trial' % (
            'host tensors handed to gloo (GradAllReducer.host_staged)' if host_staged else
            'CUDA tensors handed to gloo (its own staging)', trials, depth), flush=True)
        port = 29800 + (os.getpid() % 100) + (50 if host_staged else 0)
        logs = ['/tmp/gloo_repro_%d_%d_%d.log' % (os.getpid(), int(host_staged), r) for r in range(2)]
        procs = [ctx.Process(target=worker, args=(r, 2, port, trials, depth, logs[r], host_staged)) for r in range(2)]
        for p in procs:
            p.start()
        deadline = time.time() + 150
        for p in procs:
            p.join(timeout=max(1.0, deadline - time.time()))
        for p in procs:
            if p.is_alive():
                print('rank still running at the deadline: killed (pid %d)' % p.pid, flush=True)
                p.kill()
                p.join(timeout=20)
        for r, lg in enumerate(logs):
            try:
                lines = open(lg).read().splitlines()
            except OSError:
                lines = []
            for ln in lines[-6:]:
                print('   ' + ln, flush=True)
        print('exit codes %s' % [p.exitcode for p in procs], flush=True)


if __name__ == '__main__':
    main()
