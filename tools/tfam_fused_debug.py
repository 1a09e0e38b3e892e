"""Stage-by-stage check of the fused TFAM chain against the CPU oracle (builder tool; needs a GPU).
Runs kv + layer 0 through the C ABI and compares every buffer the six launches leave in the workspace."""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from oracle import make_golden as mg  # noqa: E402
from vimo_clip_amd import synth, tfam_fused as tf  # noqa: E402
from vimo_clip_amd._lib import check, dt, lib, ptr, stream  # noqa: E402
from vimo_clip_amd.TFAM.models import AMO_CLIP  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cross_d512"
dtype = torch.float16
c = next(x for x in mg.TFAM_CASES if x["name"] == name)
kw = mg.tfam_mode_kwargs(c["mode"])
m = AMO_CLIP(d_model=c["D"], nhead=c["H"], num_layers=c["L"], dim_feedforward=c["ff"], num_classes=c["C"], use_pe=False,
             dropout=0.0, mlp_dropout=0.0, device="cuda", compute_dtype=dtype, **kw).cuda().eval()
sd = synth.tfam_state_dict(c["D"], c["H"], c["L"], c["ff"], c["C"], c["seed"])
m.load_state_dict(sd, strict=True)
rgb, mot, mr, mf = mg.tfam_inputs(c)
B, T, D = rgb.shape
Tk, H, ff, L, C = mot.shape[1], c["H"], c["ff"], c["L"], c["C"]
pack = tf.get_pack(m, dtype).refresh()
ws = pack.workspace(B, T, Tk, True)
ws.zero_()
x, mo = rgb.cuda().contiguous(), mot.cuda().contiguous()
m8, f8 = mr.cuda().to(torch.uint8).contiguous(), mf.cuda().to(torch.uint8).contiguous()
dims = (B, T, Tk, D, H, ff, L, C)
check(lib.vmc_tfam_kv_fwd(ptr(mo), ptr(pack.wpack), ptr(pack.ppack), ptr(ws), ws.numel(), *dims, dt(dtype), stream()), "kv")
check(lib.vmc_tfam_layer_fwd(ptr(x), ptr(m8), ptr(f8), ptr(pack.wpack), ptr(pack.ppack), 0, ptr(ws), ws.numel(), *dims, 1, dt(dtype),
                             stream()), "layer")
torch.cuda.synchronize()
al = lambda n: (n + 255) // 256 * 256
M, Mk = B * T, B * Tk
o = 0


def take(nbytes, dt_, shape):
    global o
    t = ws[o:o + nbytes].view(dt_).view(*shape).float().cpu()
    o += al(nbytes)
    return t


y = take(M * D * 4, torch.float32, (M, D))
xa = take(M * D * 4, torch.float32, (M, D))
xb = take(M * D * 4, torch.float32, (M, D))
qkv = take(M * 3 * D * 2, dtype, (M, 3 * D))
q = take(M * D * 2, dtype, (M, D))
h = take(M * ff * 2, dtype, (M, ff))
kv = take(Mk * L * 2 * D * 2, dtype, (Mk, L * 2 * D))

# oracle, layer 0
p = "layers.0."
xr = rgb.reshape(M, D)
Wi, bi = sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"]
r_qkv = xr @ Wi.t() + bi


def rep(tag, got, ref):
    e = (got - ref).abs()
    print(f"{tag:>8}: max err {e.max().item():.3e}  (|ref|max {ref.abs().max().item():.2f})  worst row {int(e.max(dim=1).values.argmax())} "
          f"col {int(e.max(dim=0).values.argmax())}  rows>1e-2: {(e.max(dim=1).values > 1e-2).nonzero().flatten().tolist()[:20]}")


rep("qkv", qkv, r_qkv)
Wc, bc = sd[p + "cross_attn.in_proj_weight"], sd[p + "cross_attn.in_proj_bias"]
r_kv = torch.cat([mot.reshape(Mk, D) @ sd[f"layers.{l}.cross_attn.in_proj_weight"][D:].t() + sd[f"layers.{l}.cross_attn.in_proj_bias"][D:]
                  for l in range(L)], dim=1)
rep("kv", kv, r_kv)
from oracle import tfam as ot  # noqa: E402
x3 = rgb
a1 = ot.mha(sd, p + "self_attn.", x3, x3, H, ~mr)
x1 = F.layer_norm(x3 + a1, (D,), sd[p + "norm_self.weight"], sd[p + "norm_self.bias"], 1e-5)
rep("xb=x1", xb, x1.reshape(M, D))
r_q = x1.reshape(M, D) @ Wc[:D].t() + bc[:D]
rep("q", q, r_q)
a2 = ot.mha(sd, p + "cross_attn.", x1, mot, H, ~mf)
x2 = F.layer_norm(x1 + a2, (D,), sd[p + "norm_cross.weight"], sd[p + "norm_cross.bias"], 1e-5)
rep("xa=x2", xa, x2.reshape(M, D))
r_h = torch.relu(x2.reshape(M, D) @ sd[p + "ffn.0.weight"].t() + sd[p + "ffn.0.bias"])
rep("h", h, r_h)
r_y = x2.reshape(M, D) + r_h @ sd[p + "ffn.3.weight"].t() + sd[p + "ffn.3.bias"]
rep("y", y, r_y)
