"""The stage-1 training step of the reference (stage1_surgvu_pretraining.py:418-470) at ViT-L size: order
classification + InfoNCE + masked-mean denoising, three gradient-carrying encoder passes fused into one of 3B
clips plus one no-grad pass, global-norm clip 1.0 and AdamW as multi-tensor kernels.
usage: stage1_bench.py [B] [steps] [frozen]
`frozen`: the fine-tuning loop's frozen-encoder step instead (stage1_vjepa2_finetuning.py:281-339): four no-grad encoder
passes, the reference's per-sub-module optimizer groups, only the heads receive gradients."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vit_amd import surgical_world_model as swm
from vit_amd import train_step as ts
from vit_amd.optim import FusedAdamW

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.manual_seed(42)
model = swm.SurgicalActionConditionedWorldModel(img_size=256, num_frames=16, encoder_depth=24, encoder_heads=16,
                                                predictor_depth=1, predictor_heads=8).cuda()
model.encoder.eval()
heads = ts.Stage1Heads().cuda()
frozen = len(sys.argv) > 3 and sys.argv[3] == "frozen"
if frozen:
    opt = FusedAdamW(ts.finetune_param_groups(model, heads, 4e-4, encoder_frozen=True), weight_decay=0.1, max_grad_norm=1.0)
else:
    params = list(model.encoder.parameters()) + list(heads.parameters())
    opt = FusedAdamW(params, lr=1e-4, weight_decay=0.05, max_grad_norm=1.0)
g = torch.Generator().manual_seed(7)
batch = {k: torch.randn(B, 3, 16, 256, 256, generator=g).cuda() for k in ("video", "video_augmented", "shuffled_video")}
batch["order_label"] = torch.randint(0, 2, (B,), generator=g).cuda()
for it in range(steps + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss, parts = ts.stage1_step(model, heads, opt, batch, max_norm=None, encoder_grad=not frozen)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"step {it}: {dt*1e3:.1f} ms  loss {loss:.4f}  ({B/dt:.1f} samples/s)  "
          f"grad norm {opt.grad_norm().item():.3f}  peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
