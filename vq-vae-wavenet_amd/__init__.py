"""vq-vae-wavenet_amd: MI355X-native hot path of VQ-VAE-WaveNet (training + fast generation).

The directory name is not a Python identifier; import it with
    importlib.import_module('vq-vae-wavenet_amd')
(the repo root on sys.path), which is what train.py / generate.py / bench.py / tests do.
"""
from . import _lib, checkpoint, data, encoders, generator, graph, kernels, model, ops, parallel, utils  # noqa: F401

__all__ = ['_lib', 'checkpoint', 'data', 'encoders', 'generator', 'graph', 'kernels', 'model', 'ops', 'parallel', 'utils']
