"""World-model inference forward at the reference's rollout configuration: B clips of 16x256^2, 7-d actions and
states, predict_horizon 4, encoder ViT-L (24 x 1024, 16 heads), predictor 12 x 1024 (8 heads, hd 128).
usage: rollout_bench.py [B] [horizon]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import surgical_world_model as swm

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
horizon = int(sys.argv[2]) if len(sys.argv) > 2 else 4
torch.manual_seed(42)
model = swm.SurgicalActionConditionedWorldModel(img_size=256, num_frames=16, encoder_depth=24, encoder_heads=16,
                                                predictor_depth=12, predictor_heads=8).eval().cuda()
video = torch.randn(B, 3, 16, 256, 256, device="cuda")
actions = torch.randn(B, 16, 7, device="cuda"); states = torch.randn(B, 16, 7, device="cuda")
with torch.no_grad():
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        preds, cur, fut = model(video, actions, states, predict_horizon=horizon)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"iter {it}: {dt*1e3:.1f} ms  ({B/dt:.1f} clips/s), horizon {len(preds)}, finite {bool(torch.isfinite(cur).all())}", flush=True)
