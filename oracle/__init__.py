"""ORACLE — test infrastructure only.

CPU restatement (plain ``torch`` CPU ops / numpy / C) of the reference's
CenterNet hot path: backbones, heads, losses, decode.  Every function cites the
reference file:line it follows.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import anything from here; the
product package (``centernet-pytorch-lightning_amd/``) never does.

Pinning status
--------------
* decode / losses / heads / PoseResNet / DLA dataflow: **pinned** against the
  imported reference (``oracle/gen_golden.py`` -> ``tests/golden/*.npz``) and the
  reference's own known-answer test (tests/test_sample_encode_decode.py:51-56).
* DCNv2 arithmetic: **parity unpinned** — the extension
  (``DCNv2 @ git+https://github.com/tteepe/DCNv2``, requirements.txt:1, no version
  pin) is absent from /root/reference and cannot be fetched; ``oracle/dcn_ref.py``
  restates the published DCNv2 algorithm and is pinned by
  known-answer tests only (tests/test_oracle_dcn.py).
"""
