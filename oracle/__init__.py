"""CPU oracle of the VPR + geopose hot path — TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
the product package (vpr_amd) never does and has no CPU fallback.

Pinning status:
  * heads.py / postproc.py  — pinned against the reference's own classes/functions imported in
    the build container (tests/golden/make_golden.py -> tests/golden/*.npz, *.json) and against
    the reference's committed CSVs.
  * salad.py                — PARITY UNPINNED by the reference: the aggregator is fetched by
    torch.hub from the third-party repo serizba/salad (unpinned default branch;
    dinov2salad/dinov2salad_validation.py:65), absent offline.  Restates the published algorithm
    (arXiv:2311.15937, optimal-transport aggregation); pinned by closed-form known answers, and — round 2 — its
    log-domain Sinkhorn solver against an independent importable implementation of the same iteration, Hugging Face's
    SuperGlue port (transformers ...superglue.modeling_superglue.log_sinkhorn_iterations, the routine SALAD's solver
    descends from): tests/test_oracle_selfchecks.py.  The SALAD-specific wiring (marginals, dustbin row, MLP layout,
    normalisation order) remains unpinned.
  * knn.py                  — PARITY UNPINNED by the reference: it has no retrieval code at all
    (SURVEY.md fact 3).  Brute-force definition of the stage's contract.
"""
