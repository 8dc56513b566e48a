"""Input pipeline: mirror of dataset.py:33-84,114-144 and utils.py:93-100 of the reference.

`-dataset synthetic` (what bench.py measures) needs no files.  The real datasets read 16 kHz
mono WAVs listed in `<relative_path>/<name>_train.txt` with speaker ids from
`<relative_path>/<name>_speakers.txt` ("<id>, <int>" per line), pick a random file and a random
crop of max_len samples per example and scale int16 PCM as (pcm + 0.5) / 32767.5
(dataset.py:41).  librosa is not available; files at other sample rates are resampled with
scipy.signal.resample_poly.
"""
import math
import os
import queue
import threading

import numpy as np
import torch


def get_speaker_to_int(speaker_path):
    """utils.py:93-100."""
    speaker_to_int = {}
    with open(speaker_path) as f:
        for line in f:
            if line.strip():
                speaker, number = line.strip().split(', ')
                speaker_to_int[speaker] = int(number)
    return speaker_to_int


class Synthetic:
    """SURVEY.md 8(d) synthetic int16-PCM segments (4 sinusoids x slow envelope + noise)."""

    def __init__(self, batch_size, max_len, num_speakers=109, seed=1234, device='cuda'):
        self.B, self.T, self.num_speakers, self.dev = batch_size, max_len, num_speakers, device
        self.g = torch.Generator().manual_seed(seed)

    def next(self):
        x, spk = self.next_host()
        return x.to(self.dev, non_blocking=True), spk.to(self.dev, non_blocking=True)

    def next_host(self):
        B, T, g = self.B, self.T, self.g
        n = torch.arange(T, dtype=torch.float64)
        f = 80 + (3400 - 80) * torch.rand(B, 4, generator=g, dtype=torch.float64)
        ph = 2 * math.pi * torch.rand(B, 4, generator=g, dtype=torch.float64)
        env = 0.6 + 0.4 * torch.sin(2 * math.pi * n / T * (1 + 3 * torch.rand(B, 1, generator=g, dtype=torch.float64)))
        s = torch.sin(2 * math.pi * f[:, :, None] * n / 16000.0 + ph[:, :, None]).sum(1)
        wav = torch.clamp(0.25 * s * env + 0.02 * torch.randn(B, T, generator=g, dtype=torch.float64), -1, 1)
        pcm = torch.round(wav * 32767).to(torch.int16)
        x = (pcm.to(torch.float32) + 0.5) / 32767.5
        spk = torch.randint(0, self.num_speakers, (B,), generator=g)
        return x, spk


class Prefetcher:
    """The reference feeds its graph from a tf.data pipeline with prefetch (dataset.py:75-84); here a background thread
    reads / crops / resamples the NEXT batches into pinned host buffers while the GPU runs the current step, and `next()`
    only starts the asynchronous host-to-device copy (8 x 6656 fp32 = 213 KB per batch).  `source.next_host()` must return
    (x float32 [B, T], speaker ids int64 [B]) as CPU tensors.  `waits` counts the calls that found no batch ready."""

    def __init__(self, source, depth=3, device='cuda'):
        self.source, self.dev = source, torch.device(device)
        self.num_speakers = getattr(source, 'num_speakers', None)
        self.pin = self.dev.type == 'cuda' and torch.cuda.is_available()
        self.slots, self.free, self.ready = [], queue.Queue(), queue.Queue()
        self.waits, self.served, self.error = 0, 0, None
        self.depth = depth
        self._stop = threading.Event()
        self.thread = threading.Thread(target=self._produce, name='vqw-prefetch', daemon=True)
        self.thread.start()

    def _slot(self, x, spk):
        mk = (lambda t: torch.empty_like(t).pin_memory()) if self.pin else torch.empty_like
        return {'x': mk(x), 'spk': mk(spk), 'event': torch.cuda.Event() if self.pin else None, 'used': False}

    def _produce(self):
        try:
            while not self._stop.is_set():
                x, spk = self.source.next_host()
                if len(self.slots) < self.depth:
                    self.slots.append(self._slot(x, spk))
                    i = len(self.slots) - 1
                else:
                    while True:
                        try:
                            i = self.free.get(timeout=0.1)
                            break
                        except queue.Empty:
                            if self._stop.is_set():
                                return
                slot = self.slots[i]
                if slot['used'] and slot['event'] is not None:
                    slot['event'].synchronize()          # the copy that last read this pinned buffer has finished
                slot['x'].copy_(x)
                slot['spk'].copy_(spk)
                self.ready.put(i)
        except BaseException as e:      # surfaced by next()
            self.error = e
            self.ready.put(None)

    def next(self):
        if self.ready.empty():
            self.waits += 1
        i = self.ready.get()
        if i is None:
            raise RuntimeError('input pipeline failed') from self.error
        slot = self.slots[i]
        x = slot['x'].to(self.dev, non_blocking=True)
        spk = slot['spk'].to(self.dev, non_blocking=True)
        if not self.pin:                                  # (CPU target: .to() may alias the buffer)
            x, spk = x.clone(), spk.clone()
        else:
            slot['event'].record()
        slot['used'] = True
        self.free.put(i)
        self.served += 1
        return x, spk

    def close(self):
        self._stop.set()
        self.thread.join(timeout=2.0)


class WavDataset:
    """dataset.py:14-84: random file, random crop, one-hot speaker (here: speaker index)."""
    speaker_of = staticmethod(lambda rel: rel.split('/')[0])

    def __init__(self, batch_size, max_len, relative_path, list_name, speakers_name, data_dir='', sr=16000,
                 seed=0, device='cuda', rank=0, world=1):
        self.B, self.T, self.sr, self.dev = batch_size, max_len, sr, device
        self.root = relative_path
        self.data_dir = data_dir
        with open(os.path.join(relative_path, list_name)) as f:
            self.files = [ln.strip() for ln in f if ln.strip()]
        self.speaker_to_int = get_speaker_to_int(os.path.join(relative_path, speakers_name))
        self.num_speakers = len(self.speaker_to_int)
        bad = {k: v for k, v in self.speaker_to_int.items() if not 0 <= v < self.num_speakers}
        if bad:      # the id indexes the [num_speakers][Cs] embedding table (model.py:19-27)
            raise ValueError('%s: speaker ids outside 0..%d: %s' % (speakers_name, self.num_speakers - 1, sorted(bad.items())[:5]))
        self.rng = np.random.RandomState(seed + 7919 * rank)

    def _read(self, rel):
        """-> float32 samples in [-1,1].  16 kHz int16 files: (pcm + 0.5) / 32767.5 (dataset.py:41);
        other rates are resampled to 16 kHz from pcm / 32768 (what librosa.load returns, :50)."""
        from scipy.io import wavfile
        sr, wav = wavfile.read(os.path.join(self.root, self.data_dir, rel))
        if wav.ndim > 1:
            wav = wav[:, 0]
        if sr == self.sr and wav.dtype == np.int16:
            return ((wav.astype(np.float32) + 0.5) / 32767.5).astype(np.float32)
        f = wav.astype(np.float64) / 32768.0 if wav.dtype == np.int16 else wav.astype(np.float64)
        if sr != self.sr:
            from scipy.signal import resample_poly
            g = math.gcd(sr, self.sr)
            f = resample_poly(f, self.sr // g, sr // g)
        return f.astype(np.float32)

    def next(self):
        x, spk = self.next_host()
        return x.to(self.dev, non_blocking=True), spk.to(self.dev, non_blocking=True)

    def next_host(self):
        xs, ss = [], []
        while len(xs) < self.B:
            rel = self.files[self.rng.randint(len(self.files))]
            wav = self._read(rel)
            if len(wav) <= self.T:                      # dataset.py:44-45: too short -> another file
                continue
            start = self.rng.randint(0, len(wav) - self.T)
            xs.append(wav[start:start + self.T])
            ss.append(self.speaker_to_int[self.speaker_of(rel)])
        return torch.from_numpy(np.stack(xs)), torch.tensor(ss, dtype=torch.int64)


class VCTK(WavDataset):       # dataset.py:125-133
    def __init__(self, batch_size, max_len, relative_path='data/', **kw):
        super().__init__(batch_size, max_len, relative_path, 'vctk_train.txt', 'vctk_speakers.txt',
                         data_dir='VCTK-Corpus/wav48/', **kw)


class LibriSpeech(WavDataset):  # dataset.py:114-122
    speaker_of = staticmethod(lambda rel: rel.split('/')[-1].split('-', 1)[0])

    def __init__(self, batch_size, max_len, relative_path='data/', **kw):
        super().__init__(batch_size, max_len, relative_path, 'librispeech_train_clean_100.txt', 'librispeech_speakers.txt', **kw)


class Aishell(WavDataset):     # dataset.py:136-144
    speaker_of = staticmethod(lambda rel: rel.split('/train/')[1].split('/')[0])

    def __init__(self, batch_size, max_len, relative_path='data/', **kw):
        super().__init__(batch_size, max_len, relative_path, 'aishell_train.txt', 'aishell_speakers.txt', **kw)
