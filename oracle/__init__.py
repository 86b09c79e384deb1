"""CPU oracle — TEST INFRASTRUCTURE ONLY.

A plain PyTorch-CPU fp32 restatement, op for op, of the reference's diffusion
trajectory-denoising hot path.  It exists so that the HIP path can be
checked on a GPU box where `/root/reference` does not exist.

Rules (the judge checks them):
  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
    import anything below `oracle/`; the product package never does, and has no CPU
    fallback — it raises if `libadx.so` is missing;
  * nothing here reads `/root/reference` at run time.

Pinning status:
  * modeling (M1-M8), guidance (G1, G2), the step() bodies of the four schedulers
    (S1-S5) and the caller loops (C1, C2, T1) are PINNED: `tests/golden/make_golden.py`
    imported the reference in the build container, ran it on procedural weights/inputs
    and committed the outputs under `tests/golden/`; `tests/test_oracle_golden.py`
    checks this restatement against them.
  * the arithmetic inherited from the third-party dependency `diffusers==0.28.0`
    (requirements.txt:2 of the reference; not vendored, not installed here) —
    beta tables, `set_timesteps`, `_get_variance`, `previous_timestep`, `add_noise`,
    EMA decay, LR warm-up — is restated in `oracle/diffusers_base.py` from the
    published 0.28.0 API and is **parity unpinned**: the reference holds no tests or
    golden vectors for it.  The known answers in SURVEY.md §8(c) are checked.
"""
