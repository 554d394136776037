"""Seeded synthetic workloads of the benchmarks and parity tests (TEST / BENCH INFRASTRUCTURE, not part of the product package).

There are no pretrained checkpoints and no datasets offline: `synth.py` holds the seeded weight recipes (throughput, conditioned,
spread, linear-regime) and image generators, `data/` the committed BatchNorm calibrations they load.  Imported by bench.py, tests/, tools/, oracle/make_synth_bn.py
and tests/golden/make_golden.py; nothing under yolort_amd/ imports it.
"""
