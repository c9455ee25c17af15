"""CPU oracle for the BDS-3 B1C/B2a acquisition + tracking correlator path.

TEST INFRASTRUCTURE ONLY.  This package is a float64 NumPy/SciPy restatement of
the reference's MATLAB algorithm (lyf8118/BDS-3-B1C-B2a-SDR-receiver), written
by reading the ``.m`` files; every function cites the reference file:line it
follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it -- and only as the checker / reported CPU
baseline, never as the thing measured or shipped.  The product path
(``bds_amd`` -> ``libbds_mi355x.so`` -> HIP kernels) never imports this package
and fails loudly when the HIP library is missing.

PARITY UNPINNED: the reference is pure MATLAB, ships no tests, no golden
vectors and no IF recordings, and neither MATLAB nor Octave exists in this
image, so the reference itself can not be executed here.  What pins this
restatement instead (tests/test_oracle_*.py):
  * structural known-answers for the ranging codes (Weil/Legendre window
    property, balance, LFSR period/reset behaviour, data/pilot table identity
    for PRN 1-60) and the code digests of SURVEY.md Appendix E;
  * synthetic-IF round trips (injected PRN / Doppler / code delay recovered by
    acquisition, tracking loops lock with the documented I/Q conventions);
  * agreement with the independent C++/HIP implementation behind the C ABI;
  * agreement of two separately written restatements with each other: ``oracle/c/`` restates the hot loops -- the Doppler rows of
    the coarse search (acq_oracle.c: own mixed-radix transform, OpenMP over the bins) and one tracking epoch's sample loop
    (trk_oracle.c) -- in C from the same .m lines (``oracle/cfast.py``; tests/test_oracle_c.py holds them against the NumPy
    statements they replace), which is also what makes the oracle fast enough for whole-grid / whole-horizon checks at the
    BASELINE sizes.
One exception is pinned by the reference itself: the packed-sample converter (oracle/unpack.py) is
checked against the four literal look-up tables of B2a/include/unpack_cplx.m:32-35
(tests/golden/unpack_cplx_lut.npz).

Path abbreviations used in citations (all under /root/reference/BDS3_B1C_B2a):
  B1C/...    = BDS-3_B1C/...
  B2a/...    = BDS-3_B2a/...
  Common/... = Common/...
"""
