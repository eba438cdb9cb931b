"""What does the REFERENCE do when a feature map is smaller than the (wt, 8, 8) attention window?  (build container only; test infrastructure)

    python -m oracle.check_window_shrink

models/swin_3d.py:88-101 (get_window_size) shrinks the window to the map and zeroes the shift, but rWindowAttention (swin_3d.py:120-252) slices
queries and keys with `total_seq` / `interval`, fixed at CONSTRUCTION from the full window size: with fewer tokens than wt * 64 the key gather
`k[..., other_id, :]` indexes past the window.  This script runs the unmodified DecoderLayer on maps of 8 x 8 (a full window), 4 x 4 and 8 x 4 and
records what happens; the result is kept in tests/golden/window_shrink.json and compared with the product's behaviour by
tests/test_host_logic.py::test_small_feature_maps_raise_like_the_reference."""
import json
import os

import torch

from .gen_golden import GOLD, _import_reference


def main():
    Fn, L, S3, Tj, V, Ls = _import_reference()
    out = {}
    for (h, w) in ((8, 8), (4, 4), (8, 4), (16, 6)):
        torch.manual_seed(0)
        layer = S3.DecoderLayer(dim=32, input_resolution=4, depth=2, num_heads=4, window_size=[2, 8, 8], shift_size=None, mlp_ratio=2, qkv_bias=True,
                                is_train=True, if_unfold=False)
        x = torch.randn(1, 4, 32, h, w)  # (B, T, C, H, W), as gen_golden.py feeds it
        try:
            y = layer(x)
            out[f"{h}x{w}"] = {"ok": True, "shape": list(y.shape)}
        except Exception as e:  # noqa: BLE001 -- the point is to record WHICH error the reference raises, and where
            import traceback
            tb = traceback.extract_tb(e.__traceback__)[-1]
            out[f"{h}x{w}"] = {"ok": False, "error": type(e).__name__, "where": "models/%s:%d" % (os.path.basename(tb.filename), tb.lineno)}
    with open(os.path.join(GOLD, "window_shrink.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
