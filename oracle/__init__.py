"""CPU oracle for the V-AURA generation hot path.  TEST INFRASTRUCTURE — NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import this package, and only as the checker.  Nothing under ``vaura_amd/`` imports it;
the product path fails loudly when the HIP library is missing instead of falling back here.

Every function restates, in plain fp32 torch-CPU (the reference's own arithmetic library) or
numpy (integer bookkeeping), what the reference computes on this path and cites the
reference file:line it follows (paths relative to /root/reference).

Pinning status
--------------
* decoder / pattern / sampling / generate loop: PINNED — ``tests/golden/*.npz`` were produced by
  importing and running the reference itself (``tests/golden/make_golden.py``); the
  ``-m "not gpu"`` suite checks this oracle against them.
* DAC codec decode (``dac_oracle.py``): PARITY UNPINNED by the reference — the arithmetic lives
  in the un-vendored ``descript-audio-codec==1.0.0`` (conda_env_cuda12.1.yaml:298) and the
  reference holds no test vectors for it.  The restatement follows the published DAC-44k
  architecture and is cross-checked for structure against the independent
  ``transformers.models.dac`` implementation present in this image (golden made from that).
"""
