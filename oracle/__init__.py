"""CPU oracle for the EveryVoice TTS hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker.  The product path (``everyvoice_amd``) never
imports this package and fails loudly when its HIP library is missing.

Pinning status:

* ``heavy_ref`` (expand / collate / get_segments / dynamic-range compression),
  ``attention_prior_ref`` and ``model_utils_ref`` are PINNED: they are checked
  against golden vectors produced by importing the reference itself
  (``tests/golden/make_golden.py``, run in the build container where
  ``/root/reference`` is mounted).
* ``mel_ref`` is anchored by the ming024 mel array that the reference's test data
  holds for ``LJ010-0008.wav`` (a sanity anchor: mean |diff| ~1.5e-4, the array comes
  from a sibling code base and the reference's own tests do not assert on it).
* ``hifigan_ref`` is "PARITY UNPINNED" numerically: the reference's model code
  lives in un-vendored git submodules (``.gitmodules:1-6``), no test of the
  reference pins any output tensor.  It is pinned STRUCTURALLY by the exact
  parameter counts the reference's tests assert (``everyvoice/tests/test_cli.py:340,363``)
  and by the hyper-parameter defaults frozen in
  ``everyvoice/.schema/everyvoice-spec-to-wav-0.5.json``.
"""
