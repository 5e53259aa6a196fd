"""(test infrastructure, not collected by pytest: it uses the oracle, so it lives under tests/)  debug: per-tensor gradient errors of the tower units on the small preset (emulated-bf16 oracle vs all-fp32 oracle), for a few gscale values"""
import sys
from pathlib import Path
root = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(root), str(root / "vla-from-fastvlm_amd"), str(root / "tests")]   # root = the repo (this file sits in tests/)
import torch
from gpu_util import DEV, rel_l2
from fastvla_hip import FastVLAEngine, arch, weights
from oracle import fastvit_hd, train_tower
import test_gpu_train_tower as T

model = arch.preset("small")
w, eng = T._engine(model, 2, 16)
tc = T._tcfg(model)
tensors, total, nb = eng.train_layout()
pf = train_tower.fold_tower(w, tc)
qf = fastvit_hd.strip_prefix(pf)
g = torch.Generator().manual_seed(7)
img = torch.rand(2, 3, 336, 336, generator=g)
pix = eng.preprocess(img.to(DEV))
_, tout, taps = eng.vision_forward_unit_taps(pix)
units = fastvit_hd.tower_units(tc) + [("head", len(tc.layers) - 1, None, None)]
tws = eng.train_tower_workspace(2)
grads = torch.zeros(total, dtype=torch.float32, device=DEV)
VT = fastvit_hd.VT
sel = [int(a) for a in sys.argv[1:]] or list(range(len(units)))
for n in sel:
    unit = units[n]
    x_in = pix if n == 0 else taps[n - 1]
    x_ref = (pix.float().cpu()[..., :3] if n == 0 else x_in.float().cpu()).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        y_shape = (fastvit_hd.tower_head_forward(qf, x_ref, tc) if unit[0] == "head" else fastvit_hd.unit_forward(qf, x_ref, unit, tc)).shape
    g_ref = torch.randn(y_shape, generator=g) * 1e-2
    g_nhwc = g_ref if unit[0] == "head" else g_ref.permute(0, 2, 3, 1).contiguous()
    y_e, gx_e, gw_e = train_tower.unit_backward(qf, x_ref, unit, g_ref, tc, emulate_bf16=True)
    y_f, gx_f, gw_f = train_tower.unit_backward(qf, x_ref, unit, g_ref, tc, emulate_bf16=False)
    for gscale in (1024.0, 16.0):
        grads.zero_()
        y, g_in = eng.train_tower_unit(n, x_in, g_nhwc.contiguous(), tws, grads, gscale)
        torch.cuda.synchronize()
        named = eng.train_named_tensors(grads / gscale)
        yy = y.float().cpu()
        ye = y_e if unit[0] == "head" else y_e.permute(0, 2, 3, 1)
        line = [f"unit {n} {unit} gscale {gscale:g}: fwd(emul) {rel_l2(yy.reshape(ye.shape), ye):.2e}"]
        if n > 0:
            line.append(f"gin emul {rel_l2(g_in.cpu(), gx_e.permute(0, 2, 3, 1)):.2e} fp32 {rel_l2(g_in.cpu(), gx_f.permute(0, 2, 3, 1)):.2e} | oracle emul-vs-fp32 {rel_l2(gx_e, gx_f):.2e}")
        print(" ".join(line))
        for k in gw_e:
            got = named[VT + k].cpu().reshape(gw_e[k].shape)
            print(f"      {k:50s} emul {rel_l2(got, gw_e[k]):.2e}  fp32 {rel_l2(got, gw_f[k]):.2e}  | oracle emul-vs-fp32 {rel_l2(gw_e[k], gw_f[k]):.2e}")
