"""CPU oracle for the musicFPaugment hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, in numpy / torch-CPU, the algorithms of the reference
(deezer/musicFPaugment, pure Python) for the one hot path this repo accelerates:

    waveform -> STFT magnitude -> [UNet denoiser] -> spectral-peak picking -> peak-mask metrics

Every function cites the reference ``file:line`` it follows.  The reference is
Python, so the restatement is Python too (numpy float64 for the signal /
peak-picking arithmetic, torch-CPU float32 for the UNet).

Pinning: the oracle is checked against golden vectors produced by importing the
real reference in the build container (``tools/make_goldens.py`` ->
``tests/golden/*.npz``; see tests/test_oracle_golden.py).  Third-party arithmetic
used by the reference on this path (numpy ``rfft``/``log``/``mean``, scipy
``lfilter``/``maximum_filter``/``binary_erosion``, matplotlib ``mlab.specgram``)
is restated explicitly here and cross-checked against those libraries where they
are installed.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package -- and only as the checker / the reported CPU
baseline.  The product package ``musicfpaugment_amd`` never imports it and has
no CPU fallback: it raises when the HIP library is missing.
"""
