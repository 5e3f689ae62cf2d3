"""pypore_amd -- MI355X-native per-read QC statistics for `pypore seqstats`.

Only what the hot path needs lives here: ``csrc/`` (HIP kernels + C ABI -> libpypore_qc.so), the
ctypes binding (``_lib``), the per-GPU driver (``qc``), the synthetic workload generator (``synth``)
and the host-side mirror of the reference's seqstats interface (``seqstats``, ``seq_routines``).
"""
__version__ = "0.1.0"
