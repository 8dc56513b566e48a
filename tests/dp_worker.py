"""One data-parallel rank of tests/test_multirank_gpu.py (started as a FRESH process, never imported by pytest).

    RANK=r WORLD_SIZE=n MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/dp_worker.py <out_dir> <backend>

Builds the tiny model with the shared weights, takes its contiguous shard of the shared 4-row batch, runs ONE
model.train_step with parallel.GradAllReduce attached (the model's own buckets, the side-stream join before
bucket_ready, the 1/world scaling inside Adam) and writes the summed flat gradient and the stepped parameters.
Several ranks share ONE GPU here, so the backend is gloo (RCCL refuses two ranks on one device).

Optional third argument:
  guard   the same at a width where the guarded fp16x3 engine is active (R = S = 256, two layers, F = 256, T = 512), with a
          preprocess kernel scaled so that the residual stream of a LOUD input leaves fp16's range under the start-up scales
          while a SILENT input stays inside: rank 1 gets the loud rows, rank 0 silence.  Only rank 1 raises the range flag; the
          flag's MAX all-reduce must make BOTH ranks repeat the step on the fp32 engine.
  rccl1   ONE rank on the real RCCL communicator (backend nccl, WORLD_SIZE=1) with GradAllReduce(force=True): the bucket slices,
          the side-stream joins and the flag exchange run on RCCL; a 1-rank sum is the identity, so every bucket must come back
          bit-identical and the buckets must tile the flat gradient exactly once.
"""
import importlib
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))


def shared_problem():
    from oracle import ref_model as M
    import make_golden
    m, w = make_golden.tiny_cfg()
    P = M.init_params(m, w, 10, seed=5, randomize_all=True)
    x, spk, _ = M.synthetic_batch(4, 512, 10, 9)
    return m, w, P, x[:, :, 0].contiguous(), spk


def guard_problem():
    """Reference-width channels, two layers: every decoder contraction and encoder layers 1-2 run on the fp16x3 engine."""
    from oracle import ref_model as M
    w = dict(M.DEFAULT_WAVENET)
    w.update(dilation_rates=[1, 2], num_cycles=1, num_cycle_layers=2)
    m = dict(M.DEFAULT_MODEL, encoder_filters=256)
    P = M.init_params(m, w, 10, seed=7, randomize_all=True)
    P['decoder/preprocess/kernel'] *= 3e5            # |net| ~ 3e5 for a loud input, = the bias for silence
    for name in P:
        if name.endswith('/gated/kernel'):
            P[name] *= 1e-5                           # (keeps the gates unsaturated)
    x, spk, _ = M.synthetic_batch(4, 512, 10, 9)
    x = x[:, :, 0].contiguous()
    x[:2] = 0.0                                       # rank 0: silence
    return m, w, P, x, spk


def rccl_one_rank(pkg, out_dir, deferred=False):
    """See the module docstring: mode rccl1 (rccl1_deferred: model.defer_guard, the flag all-reduce stays on the device)."""
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    m, w, P, x, spk = guard_problem()
    x, spk = x[2:].contiguous().cuda(), spk[2:].contiguous().cuda()        # the loud rows: the step is flagged and repeated
    model = pkg.model.VQVAE(m, w, 10, device='cuda', seed=0)
    model.load_named(P)
    sync = model.grad_sync = pkg.parallel.GradAllReduce(model.grad, force=True)
    model.defer_guard = deferred
    assert sync.active and sync.world == 1
    seen = {'buckets': [], 'identical': True, 'max_calls': 0}
    real_all_reduce = dist.all_reduce

    def checked_all_reduce(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
        if op == dist.ReduceOp.MAX:
            seen['max_calls'] += 1
            return real_all_reduce(t, op=op, group=group, async_op=async_op)
        before = t.clone()                             # (on the stream the collective is queued on)
        r = real_all_reduce(t, op=op, group=group, async_op=async_op)
        # bit patterns, not values: the first pass of this step overflows on purpose and its gradients hold NaNs
        seen['identical'] = seen['identical'] and bool(torch.equal(before.view(torch.int32), t.view(torch.int32)))
        return r
    dist.all_reduce = checked_all_reduce
    try:
        ws = model.train_step(x, spk)
        if deferred:
            assert len(model._pending) == 1 and model.x3_fallbacks == 0
            model.finish_steps()
    finally:
        dist.all_reduce = real_all_reduce
    torch.cuda.synchronize()
    plain = pkg.model.VQVAE(m, w, 10, device='cuda', seed=0)             # the same step without any grad_sync
    plain.load_named(P)
    wp = plain.train_step(x, spk)
    torch.cuda.synchronize()
    torch.save({'buckets': torch.tensor(sync.last_buckets), 'identical': torch.tensor(seen['identical']),
                'max_calls': torch.tensor(seen['max_calls']), 'n_flat': torch.tensor(model.n_flat),
                'fallbacks': torch.tensor([model.x3_fallbacks, plain.x3_fallbacks]),
                'grad': model.grad.cpu(), 'grad_plain': plain.grad.cpu(), 'flat': model.flat.cpu(), 'flat_plain': plain.flat.cpu(),
                'loss': torch.tensor([model.losses(ws)[0], plain.losses(wp)[0]]),
                'backend': torch.tensor([ord(c) for c in dist.get_backend()])}, os.path.join(out_dir, 'rccl1.pt'))
    dist.destroy_process_group()


def main():
    out_dir, backend = sys.argv[1], sys.argv[2]
    mode = sys.argv[3] if len(sys.argv) > 3 else 'tiny'
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    pkg = importlib.import_module('vq-vae-wavenet_amd')
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count())
    if mode in ('rccl1', 'rccl1_deferred'):
        return rccl_one_rank(pkg, out_dir, deferred=(mode == 'rccl1_deferred'))
    dist.init_process_group(backend, rank=rank, world_size=world)
    m, w, P, x, spk = guard_problem() if mode.startswith('guard') else shared_problem()
    model = pkg.model.VQVAE(m, w, 10, device='cuda', seed=0)
    model.load_named(P)
    model.grad_sync = pkg.parallel.GradAllReduce(model.grad)
    model.defer_guard = mode == 'guard_deferred'       # the flag is MAX-all-reduced on the device and read by finish_steps()
    per = x.shape[0] // world
    rows = slice(rank * per, (rank + 1) * per)
    ws = model.train_step(x[rows].contiguous().cuda(), spk[rows].contiguous().cuda())
    if model.defer_guard:
        assert len(model._pending) == 1 and model.x3_fallbacks == 0
        model.finish_steps()               # (ws is the cached workspace: it now holds the repeated step)
    torch.cuda.synchronize()
    torch.save({'grad_sum': model.grad.cpu(), 'flat': model.flat.cpu(), 'ema': model.ema.cpu(),
                'loss': torch.tensor(model.losses(ws)), 'fallbacks': torch.tensor(model.x3_fallbacks),
                'x3_guard': torch.tensor(bool(model.x3_guard)), 'x3_used': torch.tensor(bool(ws.get('x3_used'))),
                'own_flag': torch.tensor(getattr(model, 'x3_own_flag', -1))}, os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
