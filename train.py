#!/usr/bin/env python3
"""train.py -- same command line as the reference's train.py:12-37, running on MI355X.

    python3 train.py -dataset VCTK -length 6656 -batch 8 -step 100000 -save saved_model/weights
    python3 -m torch.distributed.run --nproc-per-node 8 train.py ...      # data parallel (RCCL)

Differences from the reference: `-dataset synthetic` needs no files; checkpoints are torch
files `<save>-<global_step>.pt` (the reference writes TF checkpoints `<save>-<global_step>`);
TF summaries are replaced by the console line.
"""
import importlib
import json
import os
import sys
import time
from argparse import ArgumentParser

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def display_time(t, second):
    """Console suffix in the reference's format (utils.py:49-67): batch time, then the ETA as seconds, `Nm S.SSSs` above
    one minute, `Nh Nm S.SSSs` above one hour of minutes (the reference's thresholds are strict: 60 s prints as seconds)."""
    eta = '%.3fs' % second
    if second > 60:
        minute, second = divmod(second, 60)
        eta = '%dm %.3fs' % (minute, second)
        if minute > 60:
            eta = '%dh %dm %.3fs' % (minute // 60, minute % 60, second)
    return ' [BATCH %.3fs / ETA %s]     ' % (t, eta)


def main():
    parser = ArgumentParser()
    parser.add_argument('-dataset', default='VCTK', type=str, help='VCTK or LibriSpeech or Aishell (or synthetic)', metavar='DATASET')
    parser.add_argument('-length', default=6656, type=int, dest='max_len', metavar='int', help='number of samples one audio will contain')
    parser.add_argument('-step', default=1000000, type=int, dest='num_steps', metavar='int', help='number of steps to train')
    parser.add_argument('-batch', default=8, type=int, dest='batch_size', metavar='int', help='batch size (per GPU)')
    parser.add_argument('-interval', default=200, type=int, dest='interval', metavar='int', help='log every interval step')
    parser.add_argument('-restore', dest='restore_path', metavar='string', help='path to restore weights')
    parser.add_argument('-save', default='saved_model/weights', dest='save_path', metavar='string', help='path to save weights')
    parser.add_argument('-params', default='model_parameters.json', dest='parameter_path', metavar='str', help='path to parameters file')
    args = parser.parse_args()

    pkg = importlib.import_module('vq-vae-wavenet_amd')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    D = pkg.data
    dargs = dict(batch_size=args.batch_size, max_len=args.max_len, device=dev)
    if args.dataset == 'VCTK':
        dataset = D.VCTK(relative_path='data/', rank=rank, world=world, **dargs)
    elif args.dataset == 'LibriSpeech':
        dataset = D.LibriSpeech(relative_path='data/', rank=rank, world=world, **dargs)
    elif args.dataset == 'Aishell':
        dataset = D.Aishell(relative_path='data/', rank=rank, world=world, **dargs)
    elif args.dataset == 'synthetic':
        dataset = D.Synthetic(seed=1234 + rank, **dargs)
    else:
        raise NotImplementedError('dataset %s not implemented' % args.dataset)

    dataset = D.Prefetcher(dataset, depth=3, device=dev)       # WAV reading / resampling off the step loop (dataset.py:75-84)
    parameters, wavenet_parameters = pkg.model.load_configs(args.parameter_path)
    if parameters['encoder'] not in ('64', 'Magenta', '2019'):                      # train.py:52-60
        raise NotImplementedError('encoder %s not implemented' % parameters['encoder'])
    model = pkg.model.VQVAE(parameters, wavenet_parameters, dataset.num_speakers, device=dev, seed=0)
    if args.restore_path is not None:
        if args.restore_path.endswith(('.safetensors', '.npz')):      # TF variable names (checkpoint.py)
            pkg.checkpoint.load(model, args.restore_path)
        else:
            model.load_state_dict(torch.load(args.restore_path, map_location='cpu', weights_only=True))
    if world > 1:
        model.grad_sync = pkg.parallel.GradAllReduce(model.grad)
    model.defer_guard = os.environ.get('VQW_DEFER_GUARD', '1') != '0'     # the engine's range flag is read one step late (no host sync per step)
    gs, lr = model.global_step, model.lr_at(model.global_step)
    if rank == 0:
        print('[restore] last global step: %d, learning rate: %.5f' % (gs, lr))
    save_dir, save_name = args.save_path.split('/')
    if rank == 0 and not os.path.isdir(save_dir):
        os.mkdir(save_dir)

    for step in range(1, 1 + args.num_steps):
        t = time.time()
        x, spk = dataset.next()
        ws = model.train_step(x, spk)
        gs = model.global_step
        if rank == 0 and (gs % args.interval == 0 or step == args.num_steps):
            loss, rl, vq, commit = model.losses(ws)          # synchronises: only every `interval` steps
            t = time.time() - t
            progress = '\r[step %d] %.2f' % (gs, step / args.num_steps * 100) + '%'
            msg = ' [recons %.5f] [vq %.5f] [lr %.5f]' % (rl, vq, model.lr_at(gs - 1))
            print(progress + msg + display_time(t, (args.num_steps - step) * t), end='', flush=True)
            # the reference writes its merged summaries to a TensorBoard event file here (train.py:104-109); without
            # TensorFlow the same tags go to <save_dir>/summaries.jsonl, one line per logged step
            with open(os.path.join(save_dir, 'summaries.jsonl'), 'a') as f:
                f.write(json.dumps({'global_step': gs, 'learning_rate': model.lr_at(gs - 1), **model.summaries(ws)}) + '\n')
    if rank == 0:
        torch.cuda.synchronize()
        path = '%s-%d.pt' % (args.save_path, model.global_step)
        torch.save(model.state_dict(), path)
        # the same state under the reference's TF variable names (+ EMA shadows, Adam slots): checkpoint.py
        pkg.checkpoint.save(model, '%s-%d.safetensors' % (args.save_path, model.global_step))
        with open(os.path.join(save_dir, save_name + '.json'), 'w') as f:
            json.dump({'model': parameters, 'wavenet': wavenet_parameters, 'num_speakers': dataset.num_speakers}, f)
        print('\nsaved', path)
    dataset.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
