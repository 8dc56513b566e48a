"""CPU oracle for the VQ-VAE-WaveNet hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy / torch-CPU fp32 / plain C) of the
reference algorithm (StanislavParovoy/VQ-VAE-WaveNet, TensorFlow 1.x).  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it; the product package `vq-vae-wavenet_amd/` never does.

PARITY UNPINNED: TensorFlow is not installed here and the reference ships no
tests or golden vectors, so equality with real TF output could not be checked.
The only reference artefact that pins anything is the set of result WAVs
(results/VCTK/p225_001/*.wav), which pin the mu-law decode level table to a
few ulp (tests/test_oracle.py::test_decode_levels_match_reference_wavs, on a
committed fixture of their unique sample values).
"""
