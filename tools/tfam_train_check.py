"""Per-parameter gradient error of one TFAM train step (fused chains and per-op path) against torch autograd through the CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import test_gpu_tfam_train as t

name = sys.argv[1] if len(sys.argv) > 1 else "b8_d768"
c = next(x for x in t.CASES_ALL if x["name"] == name)
ref_loss, ref_logits, ref = t._oracle_grads(c)
res = {}
for fused in (True, False):
    m = t._model(c)
    loss, logits, grads = t._step(m, c, fused)
    res[fused] = (loss, {k: ((grads[k] - r).norm() / (r.norm() + 1e-20)).item() for k, r in ref.items()})
print("loss ref %.6f fused %.6f per-op %.6f" % (ref_loss, res[True][0], res[False][0]))
for k in ref:
    print("%-44s fused %.3e   per-op %.3e" % (k, res[True][1][k], res[False][1][k]))
