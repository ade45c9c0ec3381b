"""CPU oracle for the UniRec hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the arithmetic of the reference path
(microsoft/UniRec @ 2024_08_07) that `unirec_amd` re-implements in HIP:

  * ``oracle.model_ref``   -- torch fp32 restatement of embedding lookup, SASRec / GRU / MF
                              user encoders, dot-product scorer, BPR / softmax / BCE / CCL /
                              fullsoftmax losses, global-norm clip and *dense* Adam
                              (reference: unirec/model/**, unirec/facility/trainer.py).
  * ``oracle.data_ref``    -- integer restatement of CPython's MT19937 ``random`` stream,
                              AddNegSamples, AddUserHistory and the left-padding rule
                              (reference: unirec/data/transform/*.py, seqrecdataset.py).
  * ``oracle.philox_ref``  -- numpy restatement of the counter-based (Philox4x32-10) device
                              sampler that the HIP sampler must match bit for bit.

Rules (enforced by tests/test_no_oracle_in_product.py):
  only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
  import anything from here, and only as the checker / the timed CPU baseline.  Nothing in
  ``unirec_amd/`` imports it; the product path raises when the HIP library is missing.

Pinning: the floating-point restatement is pinned against golden vectors captured from the
*imported reference itself* (tools/capture_goldens.py -> tests/golden/*.npz, run in the build
container where /root/reference exists); the integer restatement additionally against the
known-answer values of SURVEY.md Appendix C and against CPython's own ``random`` module.
"""
