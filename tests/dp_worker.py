"""One data-parallel rank of tests/test_multirank_gpu.py (started as a FRESH process, never imported by pytest).

    RANK=r WORLD_SIZE=n MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/dp_worker.py <out_dir> <backend>

Builds the tiny model with the shared weights, takes its contiguous shard of the shared 4-row batch, runs ONE
model.train_step with parallel.GradAllReduce attached (the model's own buckets, the side-stream join before
bucket_ready, the 1/world scaling inside Adam) and writes the summed flat gradient and the stepped parameters.
Several ranks share ONE GPU here, so the backend is gloo (RCCL refuses two ranks on one device).
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


def main():
    out_dir, backend = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    pkg = importlib.import_module('vq-vae-wavenet_amd')
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count())
    dist.init_process_group(backend, rank=rank, world_size=world)
    m, w, P, x, spk = shared_problem()
    model = pkg.model.VQVAE(m, w, 10, device='cuda', seed=0)
    model.load_named(P)
    model.grad_sync = pkg.parallel.GradAllReduce(model.grad)
    per = x.shape[0] // world
    rows = slice(rank * per, (rank + 1) * per)
    ws = model.train_step(x[rows].contiguous().cuda(), spk[rows].contiguous().cuda())
    torch.cuda.synchronize()
    torch.save({'grad_sum': model.grad.cpu(), 'flat': model.flat.cpu(), 'ema': model.ema.cpu(),
                'loss': torch.tensor(model.losses(ws))}, os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
