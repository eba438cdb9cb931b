import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import cases as C, recipe as R, vmg_oracle as O
from tests.util import build_product, psnr
for name in ["vmg_tiny_few", "vmg_reds_few_cfg1"]:
    case = C.CASES[name]; cfg = case["cfg"]
    shapes, _ = C.load_fixture(f"tests/golden/{name}.npz")
    inp = case["inputs"](); tgt = R.synthetic_target(inp["x"])
    for kind in ["recipe", "init"]:
        torch.manual_seed(0)
        m32 = build_product(cfg, torch.float32); mbf = build_product(cfg, torch.bfloat16)
        sd = C.case_state_dict(case, shapes) if kind == "recipe" else {k: v.detach().cpu().clone() for k, v in m32.state_dict().items()}
        m32.load_state_dict(sd); mbf.load_state_dict(sd); m32.eval(); mbf.eval()
        with torch.no_grad():
            g32 = m32(inp["x"].cuda()).cpu(); gbf = mbf(inp["x"].cuda()).cpu()
            w = O.vmg_forward({k: v.clone() for k, v in sd.items()}, cfg, inp["x"])
        up = torch.nn.functional.interpolate(inp["x"][0], scale_factor=4, mode="bilinear")
        print(name, kind, "psnr(fp32,oracle)=%.2f psnr(bf16,oracle)=%.2f maxerr32=%.2e maxerrbf=%.2e residual_rms=%.4f dpsnr_tgt=%.4f" % (
            psnr(g32, w), psnr(gbf, w), (g32-w).abs().max(), (gbf-w).abs().max(), (w[0]-up).pow(2).mean().sqrt(), psnr(gbf,tgt)-psnr(w,tgt)), flush=True)
