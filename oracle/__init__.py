"""CPU oracle for the ViMoCLIP hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain PyTorch-fp32 / numpy restatement of the reference algorithms on the path named by
BASELINE.json.  Every function cites the reference file:line it
follows (paths relative to the reference checkout).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package, and only as the checker.  Nothing under ``vimo_clip_amd/`` imports it; the product path fails
loudly when the HIP library is missing instead of falling back to this code.

Pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md §4), so the oracle is pinned against outputs of the reference itself run in the build
container — ``losses.py`` and ``TFAM/models/AMO_CLIP.py`` are imported directly, the ViT arithmetic
(third-party OpenAI ``clip`` / HF ``transformers``) against ``transformers.CLIPModel`` built from a
config — by ``oracle/make_golden.py``, which writes the fixtures under ``tests/golden/``.
"""
