"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement (torch fp32 / numpy integer) of the arithmetic behind the
reference's hot path ``pipe(**pipe_args)`` (run_aug/run_aug.py:278) and
``cv2.Canny`` (all_utils/utils.py:83).

PARITY UNPINNED: the arithmetic of this path lives in un-vendored third-party
packages pinned by the reference's environment.yml (diffusers==0.32.2,
transformers==4.48.3, opencv-python==4.8.0.74, torch==2.6.0).  None of them is
installed in the build container or on the GPU box, there are no weights, and
the reference has no tests / golden vectors for this path.  The oracle is
therefore a restatement of the published algorithms, anchored on the
reference's own call sites (run_aug/run_aug.py:221, :235-241, :268-269, :278)
and cross-checked where an independent implementation is importable
(``transformers.CLIPTextModel`` -> tests/test_oracle_clip.py).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package.  The product path
(``saspa-aug_amd/``) never does.
"""
