"""CPU oracle for the cough-detector hot path -- TEST INFRASTRUCTURE ONLY.

Nothing in ``cough_detector_amd`` (the product) may import this package.  The
only legal importers are ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``, and there only as the checker / the timed
CPU baseline -- never as the thing shipped.

Contents
--------
``featurizer``  torch-CPU restatement of ``/root/reference/src/preprocessing.py``
                (shipped configuration: log-mel(64) + MFCC(13) + delta(13)) with
                the torchaudio transforms it calls restated op-for-op.
``dft64``       independent float64 numpy re-derivation (direct framing + rFFT,
                analytic tables) used to cross-check ``featurizer``.
``resnet``      functional restatement of ``CoughDetectorResidual`` /
                ``ResidualBlock`` (``/root/reference/src/model.py:210-293``).
``engine``      restatement of the sliding-window loop
                (``/root/reference/src/inference.py:165-247``).
``cnn``         functional restatement of ``CoughDetector`` / ``CoughDetectorSmall``
                (``/root/reference/src/model.py:11-207``).
``augmentation`` restatement of ``SpecAugment`` (``/root/reference/src/augmentation.py:271-331``).

Pinning status
--------------
* ``resnet``: PINNED.  ``/root/reference/src/model.py`` imports in the build
  container (torch only); ``oracle/make_golden.py`` runs it and commits
  weights / inputs / per-layer activations / logits under ``tests/golden/``.
* ``cnn``: PINNED the same way (``oracle/make_golden_cnn.py`` -> ``tests/golden/cnn_golden.npz``).
* ``augmentation``, and the resampler / PCEN / spectral-contrast / centroid parts of ``featurizer``:
  **PARITY UNPINNED** (torchaudio algorithms restated from their published form; no reference vectors).
* ``featurizer``: **PARITY UNPINNED against torchaudio.**  The arithmetic lives
  in ``torchaudio`` (``requirements.txt:3``: ``torchaudio>=2.0.0``, no lock
  file) which is neither under ``/root/reference`` nor installed nor
  installable here, and the reference has no tests or golden vectors.  The
  restatement follows torchaudio's published algorithm (functional.spectrogram,
  melscale_fbanks[htk, norm=None], amplitude_to_DB[top_db per clip],
  create_dct[ortho]) and is cross-checked against ``dft64``,
  ``transformers.audio_utils.mel_filter_bank`` and ``scipy.fft.dct`` plus
  analytic known-answer cases (see ``tests/test_oracle_featurizer.py``).
"""
