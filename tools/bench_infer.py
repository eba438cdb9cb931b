"""Sliding-window inference benchmark (BASELINE configs[3] = cfg4 of SURVEY 8d): a (1,100,3,180,320) sequence through
vmg_amd.infer.test_clips with temporal windows 50/25 and spatial tiles 128/20 = 18 calls of (1,50,3,128,128), few_levels
network, bf16, random-init weights, synthetic frames.  Prints one JSON line (sequence-level LR-frames/s) and the
accumulate kernel's achieved HBM rate.   python tools/bench_infer.py [--frames 100] [--reps 2]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--reps", type=int, default=2)
    args = ap.parse_args()
    import vmg_amd
    from vmg_amd import infer
    from vmg_amd.data import REDS_FEW_LEVELS, synthetic_clip
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    torch.backends.cudnn.benchmark = True
    m = vmg_amd.VMG(num_frames=50, image_size=[128, 128], is_train=False, spynet_pretrained=None, compute_dtype=torch.bfloat16, **REDS_FEW_LEVELS)
    m.spynet = vmg_amd.SPyNet(None)
    m = m.to(dev).eval()
    x = synthetic_clip(1, args.frames, 180, 320, seed=7, device=dev)
    calls = len(infer.tile_starts(args.frames, 50, 25)) * len(infer.tile_starts(180, 128, 20)) * len(infer.tile_starts(320, 128, 20))
    out = infer.test_clips(m, x, 50, 25, [128, 128], 20, 4)  # warm-up (MIOpen search, weight packs)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.reps):
        out = infer.test_clips(m, x, 50, 25, [128, 128], 20, 4)
        u8 = infer.to_uint8(out)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / args.reps
    # accumulate kernel alone: one (1,50,3,512,512) tile into the canvases, algorithmic bytes = patch read + 2 x (read + write) of the region
    patch = torch.rand(1, 50, 3, 512, 512, device=dev)
    E = torch.zeros(1, 50, 3, 720, 1280, device=dev)
    Wt = torch.zeros_like(E)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        infer._accumulate(patch, E, Wt, 0, 0, (0, 10, 0, 10))
    ev0.record()
    for _ in range(20):
        infer._accumulate(patch, E, Wt, 0, 0, (0, 10, 0, 10))
    ev1.record()
    torch.cuda.synchronize()
    us = ev0.elapsed_time(ev1) / 20 * 1e3
    gb = patch.numel() * 4 * 5 / 1e9
    print(json.dumps({"metric": "LR-frames/s (sliding-window inference, 180x320 -> 720x1280, few_levels, bf16)", "value": round(args.frames / dt, 3),
                      "unit": "LR-frames/s", "seconds_per_sequence": round(dt, 3), "network_calls": calls, "output_shape": list(u8.shape),
                      "accumulate_kernel": {"us": round(us, 1), "algorithmic_GB": round(gb, 3), "GB_per_s": round(gb / (us * 1e-6), 1), "peak_GB_per_s": 8000},
                      "finite": bool(torch.isfinite(out.float()).all())}))


if __name__ == "__main__":
    main()
