#!/usr/bin/env python3
"""Benchmark of the VQ-VAE-WaveNet hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one full training step (forward + backward + gradient all-reduce over RCCL for
N > 1 + TF-Adam + EMA) on synthetic 16 kHz segments, len=6656, batch=8 PER GPU (weak scaling),
default config (encoder '64', K=512, 30-layer WaveNet), fp32 -- BASELINE.json configs[1].
Rank 0 prints ONE JSON line: training audio-samples/s for the whole job, the roofline of the
dominant kernel (the gated dilated conv, MFMA-bound: SURVEY.md 8(d)) measured live with HIP
events on the launch stream, the CPU baseline (the oracle restatement timed on this box's host
cores; rank 0 at N=1 only) and the fast-generation rate.
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_F16_MFMA_TFLOPS = 2500.0   # dense fp16/bf16 matrix peak, same guide (the fp16x3 engine's pipe)
PEAK_HBM_GBS = 8000.0


def synthetic_batch(B, T, num_speakers, seed, device):
    """int16 PCM of 4 sinusoids + envelope + noise -> x=(pcm+0.5)/32767.5."""
    import math
    g = torch.Generator().manual_seed(seed)
    n = torch.arange(T, dtype=torch.float64)
    f = 80 + (3400 - 80) * torch.rand(B, 4, generator=g, dtype=torch.float64)
    ph = 2 * math.pi * torch.rand(B, 4, generator=g, dtype=torch.float64)
    env = 0.6 + 0.4 * torch.sin(2 * math.pi * n / T * (1 + 3 * torch.rand(B, 1, generator=g, dtype=torch.float64)))
    s = torch.sin(2 * math.pi * f[:, :, None] * n / 16000.0 + ph[:, :, None]).sum(1)
    noise = torch.randn(B, T, generator=g, dtype=torch.float64)
    wav = torch.clamp(0.25 * s * env + 0.02 * noise, -1, 1)
    pcm = torch.round(wav * 32767).to(torch.int16)
    x = ((pcm.to(torch.float32) + 0.5) / 32767.5)
    spk = torch.randint(0, num_speakers, (B,), generator=g)
    return x.to(device), spk.to(device)


def default_configs():
    m = {"encoder": "64", "use_vq": True, "speaker_embedding": 64, "k": 512, "latent_dim": 64, "beta": 0.25,
         "learning_rate_schedule": {"0": 8e-5, "80000": 6e-5, "160000": 4e-5, "240000": 2e-5, "320000": 1e-5,
                                    "400000": 8e-6}}
    w = {"quantization_channels": 256, "num_cycles": 3, "num_cycle_layers": 10,
         "dilation_rates": [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3, "kernel_size": 3,
         "dilation_filters": 256, "skip_filters": 512, "residual_filters": 256,
         "preprocess": {"kernel_size": 32, "filters": 256}}
    for name, cfg in (('model_parameters.json', m), ('wavenet_parameters.json', w)):
        p = os.path.join(ROOT, name)
        if os.path.exists(p):
            with open(p) as fh:
                cfg.update(json.load(fh))
    return m, w


class KernelTimers:
    """HIP events around every launch of the kernel families on the roofline lines, recorded on the stream the launch goes
    to (the weight gradients run on the model's side stream: torch.cuda.Event.record() uses the stream that is current
    at the call, which is that one).  'gate': the dilated gate conv (fp32 engine: conv_gemm with the GATE epilogue).
    'wgrad': every engine weight-gradient launch, with its ALGORITHMIC flop count 2 * B * T * Cp * taps * (Q0 + Q1)."""

    def __init__(self, kernels_mod):
        self.K = kernels_mod
        self.orig = {n: getattr(kernels_mod, n) for n in ('conv_gemm', 'f16x3_gate_conv', 'f16x3_wgrad', 'f16x3_wgrad_batch', 'wgrad_gemm')}
        self.x3 = False          # the plane engine's gate kernel was the one launched
        self.events = {'gate': [], 'wgrad': []}
        self.on = False

    def _timed(self, family, fn, kw, flops):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(**kw)
        e1.record()
        self.events[family].append((e0, e1, flops))

    def __enter__(self):
        K, o = self.K, self.orig

        def conv_gemm(**kw):
            if self.on and kw.get('epilogue') == K.EPI_GATE:
                self._timed('gate', o['conv_gemm'], kw, 0.0)
            else:
                o['conv_gemm'](**kw)

        def gate_conv(**kw):
            if self.on:
                self._timed('gate', o['f16x3_gate_conv'], kw, 0.0)
                self.x3 = True
            else:
                o['f16x3_gate_conv'](**kw)

        def wgrad_x3(**kw):
            if self.on:
                self._timed('wgrad', o['f16x3_wgrad'], kw, 2.0 * kw['B'] * kw['T'] * kw['Cp'] * len(kw['taps']) * (kw['Q0'] + kw.get('Q1', 0)))
            else:
                o['f16x3_wgrad'](**kw)

        def wgrad_batch(problems, **kw):      # the weight gradients of several layers in one launch
            if self.on:
                fl = 0.0
                for pr in problems:
                    q = dict(kw, **pr)
                    fl += 2.0 * q['B'] * q['T'] * q['Cp'] * len(q['taps']) * (q['Q0'] + q.get('Q1', 0))
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                o['f16x3_wgrad_batch'](problems, **kw)
                e1.record()
                self.events['wgrad'].append((e0, e1, fl))
            else:
                o['f16x3_wgrad_batch'](problems, **kw)

        def wgrad_fp32(**kw):
            if self.on:
                self._timed('wgrad', o['wgrad_gemm'], kw, 2.0 * kw['B'] * kw['T_q'] * kw['Cp'] * len(kw['taps']) * (kw['Q0'] + kw.get('Q1', 0)))
            else:
                o['wgrad_gemm'](**kw)
        K.conv_gemm, K.f16x3_gate_conv, K.f16x3_wgrad, K.wgrad_gemm, K.f16x3_wgrad_batch = conv_gemm, gate_conv, wgrad_x3, wgrad_fp32, wgrad_batch
        return self

    def __exit__(self, *a):
        for n, f in self.orig.items():
            setattr(self.K, n, f)

    def reset(self):
        self.events = {'gate': [], 'wgrad': []}

    def mean_ms(self, family='gate'):
        ev = self.events[family]
        return sum(a.elapsed_time(b) for a, b, _ in ev) / max(len(ev), 1)

    def family(self, family):
        """(launches, total ms, total algorithmic flop) of one family over the recorded region."""
        ev = self.events[family]
        return len(ev), sum(a.elapsed_time(b) for a, b, _ in ev), sum(f for _, _, f in ev)


def hbm_traffic(name='round1_gate_conv_traffic.json'):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/<name>, made by tools/pmc_traffic.py); None if absent."""
    p = os.path.join(ROOT, 'profiles', name)
    if not os.path.exists(p):
        return None
    with open(p) as fh:
        return json.load(fh).get('hbm_bytes')


def log(*a):
    print('[bench]', *a, file=sys.stderr, flush=True)


def host_cores():
    """Cores this process may actually use (the GPU box gives a 16-core share of a big host)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get('VQW_CPU_THREADS', '16'))))


def cpu_baseline(B, T, steps):
    """The oracle (a torch-CPU fp32 restatement of the reference graph: 'port') timed on the
    host cores of this box: full training steps at batch 1 (BASELINE.json configs[0])."""
    from oracle import ref_model as M
    cores = host_cores()
    torch.set_num_threads(cores)
    log('cpu baseline: %d threads' % cores)
    m, w = dict(M.DEFAULT_MODEL), dict(M.DEFAULT_WAVENET)
    P = M.init_params(m, w, 109, seed=0)
    x, spk, _ = M.synthetic_batch(B, T, 109, 1234)
    st = {'t': 0, 'm': {}, 'v': {}, 'ema': {}}
    M.train_step(x, spk, P, m, w, st, 0)            # warm-up
    log('cpu baseline: warm-up step done')
    t0 = time.time()
    for i in range(steps):
        M.train_step(x, spk, P, m, w, st, i + 1)
    dt = (time.time() - t0) / steps
    return {"value": B * T / dt, "unit": "audio-samples/s", "cores": cores, "kind": "port",
            "sample": "%d full training steps (fwd+bwd+Adam+EMA) of the torch-CPU oracle at batch=%d len=%d, "
                      "after 1 warm-up step" % (steps, B, T), "ms_per_step": dt * 1e3}


def cpu_ar_baseline(steps):
    """The oracle's FIFO-queue generator (wavenet_ops.py:147-267 restated in torch-CPU fp32) timed on the host
    cores: greedy samples/s of one utterance at the default widths."""
    from oracle import ref_model as M
    m, w = dict(M.DEFAULT_MODEL), dict(M.DEFAULT_WAVENET)
    P = M.init_params(m, w, 109, seed=0)
    g = M.FastGenerator(P, w, 1)
    cond = torch.zeros(1, m['latent_dim'] + m['speaker_embedding'])
    a = torch.zeros(1, 1)
    with torch.no_grad():
        g.step(a, cond)                                  # warm-up
        t0 = time.time()
        for _ in range(steps):
            pr = g.step(a, cond)
            a = torch.from_numpy(M.R.mu_law_decode_np(pr.argmax(-1).numpy().astype('float32'))).reshape(1, 1)
        dt = time.time() - t0
    return {"value": steps / dt, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d greedy steps of the torch-CPU oracle generator, batch 1" % steps}


def spawn_ranks(n):
    """One process per GPU under torch.distributed.run (RCCL rendezvous on 127.0.0.1, a free port); returns its exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC: RCCL needs it on this driver
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log('launching %d ranks: %s' % (n, ' '.join(cmd)))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)     # (the first steps allocate workspaces and settle the guard scales)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--length', type=int, default=6656)
    ap.add_argument('--gen-steps', type=int, default=8192, help='AR samples to generate for the generation rate (SURVEY 8(d): L >= 8192)')
    ap.add_argument('--encoder', default=None, help="override model_parameters.json's encoder ('64', 'Magenta', '2019'); "
                    "'2019' needs --length 6400 (T %% 320 == 0): BASELINE.json configs[4] in fp32")
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'], help="bf16: BASELINE.json configs[4] -- the decoder's contraction operands as "
                    "bf16 planes, bf16 MFMA, fp32 accumulate (use with --encoder 2019 --length 6400)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-gen', action='store_true')
    ap.add_argument('--no-other-engine', action='store_true', help='skip the extra timing of the same step on the fp32-MFMA engine')
    ap.add_argument('--no-config4', action='store_true', help="skip the 5-step '2019' encoder / bf16 leg (BASELINE.json configs[4])")
    ap.add_argument('--probe-ranks', action='store_true', help='only start the ranks and report how many joined (no GPU work)')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the N>1 code path)")
    a = ap.parse_args()

    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as FRESH child processes of
        # torch.distributed.run, before this process has made any GPU call (it never makes one), and pass the one
        # JSON line of rank 0 through.
        raise SystemExit(spawn_ranks(a.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus:
        raise SystemExit('WORLD_SIZE=%d does not match --gpus %d' % (world, a.gpus))
    if a.probe_ranks:     # launcher check without a GPU: every rank joins a gloo group, rank 0 prints what the job looks like
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group('gloo', rank=rank, world_size=world)
        t = torch.ones(1)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"probe": "ranks", "n_gpus": world, "ranks_seen": int(t.item())}), flush=True)
        dist.destroy_process_group()
        return
    ndev = torch.cuda.device_count()
    if world > max(ndev, 1) and a.backend == 'nccl':
        raise SystemExit('--gpus %d needs %d GPUs, this node shows %d (RCCL cannot place two ranks on one GPU; '
                         '--backend gloo rehearses the N>1 code path on fewer)' % (a.gpus, a.gpus, ndev))
    local = local % max(ndev, 1)   # gloo rehearsal: several ranks on one GPU
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if a.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)

    if a.dtype == 'bf16':
        os.environ['VQW_DTYPE'] = 'bf16'
    pkg = importlib.import_module('vq-vae-wavenet_amd')
    K = pkg.kernels
    m, w = default_configs()
    if a.encoder:
        m['encoder'] = a.encoder
    S = 109
    model = pkg.model.VQVAE(m, w, S, device=dev, seed=0)           # identical weights on every rank
    B, T = a.batch, a.length
    x, spk = synthetic_batch(B, T, S, 1234 + rank, dev)             # per-rank data

    if world > 1:   # RCCL sum over xGMI of the 140.6 MB flat gradient, 2 buckets overlapped with backward
        model.grad_sync = pkg.parallel.GradAllReduce(model.grad)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log('model built, %d parameters; warm-up' % model.n_flat)
    # the guarded engine's range flag of step k is read after step k + 1 has been enqueued (model._train_step_deferred, as
    # train.py runs it); finish_steps() inside the timed region resolves the last step's flag before the clock stops
    model.defer_guard = os.environ.get('VQW_DEFER_GUARD', '1') != '0'
    with KernelTimers(K) as gt:
        for _ in range(a.warmup):
            model.train_step(x, spk)
        model.finish_steps()
        barrier()
        log('timing %d steps' % a.steps)
        gt.on = True
        t0 = time.perf_counter()
        for _ in range(a.steps):
            ws = model.train_step(x, spk)
        model.finish_steps()
        barrier()
        dt = time.perf_counter() - t0
        gt.on = False
        gate_ms = gt.mean_ms('gate')
        gate_x3 = gt.x3
        wg_n, wg_ms, wg_flop = gt.family('wgrad')
        # the same kernel families once more with the step on ONE stream (outside the timed region): in the timed region the
        # weight gradients share the chip with the backward chain's kernels, so their event brackets include the other
        # stream's work; these are the stand-alone durations profiles/*_single_stream.csv shows
        ss = None
        if world == 1 and model.overlap_wgrad:
            gt.reset()
            model.overlap_wgrad = False
            model.train_step(x, spk)
            torch.cuda.synchronize()
            gt.on = True
            for _ in range(3):
                model.train_step(x, spk)
            torch.cuda.synchronize()
            gt.on = False
            model.overlap_wgrad = True
            ss = {'gate_ms': gt.mean_ms('gate'), 'wgrad': gt.family('wgrad')}
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    loss = model.losses(ws)[0]
    log('train: %.2f ms/step, loss %.5f' % (dt / a.steps * 1e3, loss))

    # Beside the headline: the same step on the other engine (VQW_ENGINE=fp32: the fp32-MFMA engine everywhere)
    exp = None
    if world == 1 and not a.no_other_engine and model.x3_guard and a.dtype == 'f32':
        try:
            os.environ['VQW_ENGINE'] = 'fp32'
            try:
                model_x = pkg.model.VQVAE(m, w, S, device=dev, seed=0)
            finally:
                del os.environ['VQW_ENGINE']
            for _ in range(a.warmup):
                model_x.train_step(x, spk)
            torch.cuda.synchronize()
            tx = time.perf_counter()
            for _ in range(a.steps):
                ws_x = model_x.train_step(x, spk)
            torch.cuda.synchronize()
            dtx = time.perf_counter() - tx
            exp = {"switch": "VQW_ENGINE=fp32", "what": "every contraction of the step on the fp32 MFMA (v_mfma_f32_32x32x2_f32)",
                   "value": B * T * a.steps / dtx, "unit": "audio-samples/s", "ms_per_step": dtx / a.steps * 1e3,
                   "loss": model_x.losses(ws_x)[0]}
            log('fp32 engine: %.2f ms/step, loss %.5f' % (exp["ms_per_step"], exp["loss"]))
            del model_x, ws_x
            torch.cuda.empty_cache()
        except Exception as e:   # the headline must not depend on the comparison leg
            exp = {"switch": "VQW_ENGINE=fp32", "error": '%s: %s' % (type(e).__name__, e)}
            log('fp32-engine leg failed: %s' % exp['error'])

    gen = None
    if not a.no_gen:
        gen_mod = pkg.generator
        enc = model.encode(x[:1].contiguous(), spk[:1].contiguous())          # one utterance per GPU
        g = gen_mod.FastGenerator(model, batch=1)
        g.generate(enc, 64)                                                    # warm-up + graph capture
        torch.cuda.synchronize()
        g.reset()
        t1 = time.perf_counter()
        g.generate(enc, a.gen_steps)
        torch.cuda.synchronize()
        gdt = time.perf_counter() - t1
        gsum = torch.tensor([a.gen_steps / gdt], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(gsum)
        gen = {"metric": "AR-gen samples/sec", "value": float(gsum.item()), "unit": "samples/s",
               "utterances": world, "batch_per_gpu": 1, "steps": a.gen_steps, "mode": "greedy",
               "us_per_sample": gdt / a.gen_steps * 1e6}
        g.close()
        log('generation: %.1f us/sample' % (gdt / a.gen_steps * 1e6))
        if world == 1:   # BASELINE.json configs[3] folded onto ONE GPU: 8 speakers x 1 utterance, rows as concurrent handles
            enc8 = model.encode(x.contiguous(), spk.contiguous())
            g8 = gen_mod.FastGenerator(model, batch=B)
            g8.generate(enc8, 64)
            torch.cuda.synchronize()
            g8.reset()
            t1 = time.perf_counter()
            g8.generate(enc8, a.gen_steps)
            torch.cuda.synchronize()
            g8dt = time.perf_counter() - t1
            g8.close()
            gen["eight_utterances_one_gpu"] = {"value": B * a.gen_steps / g8dt, "unit": "samples/s", "utterances": B,
                                               "us_per_step": g8dt / a.gen_steps * 1e6}
            log('generation, %d utterances on one GPU: %.1f us/step' % (B, g8dt / a.gen_steps * 1e6))

    # BASELINE.json configs[4] on this GPU: '2019' encoder, T=6400, bf16 storage + fp32 accumulate, 5 timed steps
    cfg4 = None
    if world == 1 and not a.no_config4 and a.dtype == 'f32' and (a.encoder or m['encoder']) == '64':
        model.free_workspaces()
        ws = None
        torch.cuda.empty_cache()
        try:
            os.environ['VQW_DTYPE'] = 'bf16'
            try:
                m4 = dict(m, encoder='2019')
                model4 = pkg.model.VQVAE(m4, w, S, device=dev, seed=0)
            finally:
                del os.environ['VQW_DTYPE']
            T4 = 6400
            x4, spk4 = synthetic_batch(B, T4, S, 1234, dev)
            for _ in range(3):
                model4.train_step(x4, spk4)
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            for _ in range(5):
                ws4 = model4.train_step(x4, spk4)
            torch.cuda.synchronize()
            d4 = (time.perf_counter() - t4) / 5
            cfg4 = {"workload": "LibriSpeech '2019' encoder, K=512, len=%d batch=%d, bf16 storage + fp32 accumulate, full train step" % (T4, B),
                    "value": B * T4 / d4, "unit": "audio-samples/s", "ms_per_step": d4 * 1e3, "steps": 5, "warmup": 3,
                    "dtype": "bf16 planes (v_mfma_f32_32x32x16_bf16), fp32 accumulate / master weights / optimiser",
                    "steps_on_engine": model4.x3_steps, "loss": model4.losses(ws4)[0]}
            log('configs[4] bf16: %.2f ms/step' % (d4 * 1e3))
            del model4, ws4
            torch.cuda.empty_cache()
        except Exception as e:
            cfg4 = {"error": '%s: %s' % (type(e).__name__, e)}
            log('configs[4] leg failed: %s' % cfg4['error'])

    if rank == 0:
        R, ks = model.R, model.ks
        flops_gate = 2.0 * B * T * (ks * R) * (2 * R)       # K1: causal dilated conv 256 -> 512, k=3 (ALGORITHMIC: SURVEY 8(d))
        ach = flops_gate / (gate_ms * 1e-3) / 1e12
        step_tflops = 118.14e6 * B * T * a.steps / dt / 1e12   # 118.14 MFLOP per audio sample, fwd + bwd

        def wgrad_line(n, ms, flop, peak, note):
            if not n or ms <= 0:
                return None
            ach_w = flop / (ms * 1e-3) / 1e12
            return {"bound": "mfma", "kernel": note, "achieved": ach_w, "peak": peak, "unit": "TFLOP/s", "frac": ach_w / peak,
                    "launches_per_step": n / a.steps, "ms_per_step": ms / a.steps, "flop_per_step": flop / a.steps, "traffic": None}
        rec = {
            "metric": "training audio-samples/sec", "value": B * T * a.steps * world / dt,
            "unit": "audio-samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "VCTK '%s' encoder, len=%d batch=%d per GPU, fp32, full train step "
                                   "(fwd+bwd+allreduce+Adam+EMA)" % (m['encoder'], T, B),
                       "global_batch": B * world, "seq_len": T, "parallelism": "dp%d" % world},
            "loss": loss,
            "roofline": {"bound": "mfma", "kernel": "conv_gemm_kernel<2,2,GATE> (dilated k=3 conv 256->512 + cond-add + tanh*sigmoid gate; LDS-DMA pipeline, half-width tail tiles)",
                         "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / PEAK_FP32_MFMA_TFLOPS, "traffic": hbm_traffic(),
                         "ms_per_launch": gate_ms, "flop_per_launch": flops_gate,
                         "whole_step_tflops": step_tflops, "whole_step_frac": step_tflops / PEAK_FP32_MFMA_TFLOPS},
        }
        rec["roofline_wgrad"] = wgrad_line(wg_n, wg_ms, wg_flop, PEAK_FP32_MFMA_TFLOPS, "wgrad_kernel<2,2> (fp32 MFMA, atomics)")
        if gate_x3 and model.bf16:   # BASELINE.json configs[4]: one bf16 MFMA per product
            rec["dtype"] = ("bf16 (decoder contraction operands stored as bf16 planes, v_mfma_f32_32x32x16_bf16, fp32 accumulate; master "
                            "weights, optimiser, residual stream, encoder, VQ and losses fp32)")
            rec["config"]["workload"] = rec["config"]["workload"].replace("fp32,", "bf16 storage + fp32 accumulate,")
            rec["roofline"].update({"kernel": "gate_f16x3_kernel<bf16> (dilated k=3 conv 256->512 + cond-add + tanh*sigmoid gate; one bf16 plane per operand)",
                                    "achieved": ach, "peak": PEAK_F16_MFMA_TFLOPS, "frac": ach / PEAK_F16_MFMA_TFLOPS, "traffic": None,
                                    "whole_step_frac": step_tflops / PEAK_F16_MFMA_TFLOPS})
            rec["roofline_wgrad"] = wgrad_line(wg_n, wg_ms, wg_flop, PEAK_F16_MFMA_TFLOPS, "wgrad_f16x3_kernel<bf16> + fp32-engine launches")
            rec["engine"] = {"name": "bf16", "steps_on_engine": model.x3_steps}
        elif gate_x3:
            # the fp16x3 engine ran.  frac = ALGORITHMIC flop / time / the peak of the pipe the kernel runs on (the
            # fp16 MFMA); the engine issues three fp16 MFMA terms per algorithmic product, so the pipe's issue rate is 3x that:
            # reported separately as pipe_utilisation, never as achieved work
            rec["dtype"] = ("f32 via fp16x3 (storage and accumulation fp32; the engine's contraction operands are each fp32 value split "
                            "into two fp16 planes = 22-23 significand bits, products h1h1 + h1h2 + h2h1 on the fp16 MFMA, device-side "
                            "range guards -- forward, input and weight gradients of the residual stack, of the 1x1 convs around it and "
                            "of encoder layers 1-5; the Cin=1 convs, encoder layer 6 (1x1) and the condition projection on the fp32 "
                            "MFMA / VALU; the exact-fp32 engine's number for the same step is engine_fp32)")
            rec["config"]["workload"] = rec["config"]["workload"].replace("fp32,", "fp32 (fp16x3 engine),")
            rec["roofline"].update({"kernel": "gate_f16x3_kernel<%d-row blocks> (dilated k=3 conv 256->512 + cond-add + tanh*sigmoid gate; operands as two fp16 planes, 3 MFMA terms)" % (128 if model.x3_mode_fwd & 2 else 256),
                                    "achieved": ach, "peak": PEAK_F16_MFMA_TFLOPS, "frac": ach / PEAK_F16_MFMA_TFLOPS,
                                    "pipe_utilisation": 3 * ach / PEAK_F16_MFMA_TFLOPS, "mfma_terms_per_product": 3,
                                    "traffic": hbm_traffic('round3_gate_f16x3_traffic.json' if model.x3_mode_fwd & 2 else 'round2_gate_f16x3_traffic_256row_blocks.json'),
                                    "whole_step_frac": step_tflops / PEAK_F16_MFMA_TFLOPS})
            rec["roofline_wgrad"] = wgrad_line(wg_n, wg_ms, wg_flop, PEAK_F16_MFMA_TFLOPS,
                                               "wgrad_f16x3_kernel + wgrad_reduce_kernel (every engine weight-gradient launch of the step: the decoder's "
                                               "layers in batches -- all skip halves, gate kernels by six, residual halves -- with both operands read as "
                                               "planes, the convs around the stack, encoder layers 1-5) + the 2 fp32-engine launches")
            if rec["roofline_wgrad"]:
                rec["roofline_wgrad"]["pipe_utilisation"] = 3 * rec["roofline_wgrad"]["frac"]
            rec["engine"] = {"name": "f16x3", "steps_on_engine": model.x3_steps, "steps_repeated_on_fp32": model.x3_fallbacks,
                             "range_flag": "read one step late (model.defer_guard)" if model.defer_guard else "read before the optimiser (host sync per step)",
                             "host_enqueue_ms_per_step": model.host_enqueue_ms}
        if ss:      # single-stream durations of the same families (3 extra steps after the timed region)
            n1, ms1, fl1 = ss['wgrad']
            pk = rec["roofline"]["peak"]
            rec["roofline"]["single_stream"] = {"ms_per_launch": ss['gate_ms'], "frac": flops_gate / (ss['gate_ms'] * 1e-3) / 1e12 / pk}
            if rec["roofline_wgrad"] and n1:
                rec["roofline_wgrad"]["single_stream"] = {"ms_per_step": ms1 / 3, "achieved": fl1 / (ms1 * 1e-3) / 1e12,
                                                          "frac": fl1 / (ms1 * 1e-3) / 1e12 / pk}
        if cfg4:
            rec["config4_bf16"] = cfg4
        if exp:
            rec["engine_fp32"] = exp
        if gen:
            rec["ar_gen"] = gen
        if world == 1 and not a.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(1, T, 12)          # ~10 s of host work
            if gen:
                rec["ar_gen"]["cpu_baseline"] = cpu_ar_baseline(128)
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
