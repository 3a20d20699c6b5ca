"""CPU oracle for the CLIP-Event hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product path
(``clip_event_amd``) never imports this package and fails loudly when its
HIP extension is missing.

Pinning: every function here is a plain-PyTorch fp32 restatement of the
reference file:line it cites, and is checked in ``tests/test_oracle_golden.py``
against golden vectors captured by *importing the reference itself* in the
build container (``tests/golden/make_golden.py``; the reference never travels).
"""
